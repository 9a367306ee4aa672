#!/usr/bin/env python3
"""Generate the committed fixtures under tests/golden/ from the reference's DATA files.

Run once in the build container (the reference tree does not exist on the GPU box):

    python tests/golden/make_fixtures.py [/root/reference]

Only data is derived here -- the reference's datasets under src/CUDA/csv_files/ -- never its
source text.  Outputs (all little-endian, a few hundred KB in total):

  hall_ranges_u32.bin      16384 x uint32   OS1-16 ranges in mm, scan order (64 packets x 16 azimuth
                                            blocks x 16 beams), decoded from Donut_1024x16.csv with an
                                            independent byte-offset parser (format: SURVEY.md 2.4)
  hall_meta.json           encoder count of the first azimuth block, counts, zero-range count
  beam_intrinsics.csv      verbatim copy of the 131-line data file (64 altitude + 64 azimuth angles)
  os1_two_packets.csv      first 2 x 12608 lines of Donut_1024x16.csv (one byte value per line) --
                           raw-format sample for the reader tests
  bunny_res_xyz_f32.bin    8171 x 3 float32   Bunny_res.csv parsed (space separated)
  bunny_xyz_f32.bin        35947 x 3 float32  Bunny.csv parsed (semicolon separated)
  bunny_res_head.csv / bunny_head.csv   first 64 lines of each, verbatim (CRLF kept) -- raw-format samples
"""
import json
import os
import sys

import numpy as np

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
CSV = os.path.join(REF, "src", "CUDA", "csv_files")
OUT = os.path.dirname(os.path.abspath(__file__))

PACKET_BYTES = 12608
BLOCK_BYTES = 788
N_PACKETS = 64


def main():
    # ---- hall -------------------------------------------------------------------------------
    with open(os.path.join(CSV, "Donut_1024x16.csv"), "rb") as f:
        raw_lines = f.read().split(b"\n")
    vals = np.array([int(x) for x in raw_lines if x.strip() != b""], dtype=np.int64)
    assert vals.size == N_PACKETS * PACKET_BYTES, vals.size
    assert vals.min() >= 0 and vals.max() <= 255
    b = vals.astype(np.uint32)
    ranges = np.zeros(N_PACKETS * 16 * 16, dtype=np.uint32)
    o = 0
    for p in range(N_PACKETS):
        for blk in range(16):
            base = p * PACKET_BYTES + blk * BLOCK_BYTES + 16
            for ch in range(2, 64, 4):
                w = base + 12 * ch
                ranges[o] = b[w] | (b[w + 1] << 8) | ((b[w + 2] & 0xF) << 16)
                o += 1
    enc = int(b[12] | (b[13] << 8))
    ranges.tofile(os.path.join(OUT, "hall_ranges_u32.bin"))
    with open(os.path.join(OUT, "hall_meta.json"), "w") as f:
        json.dump({"encoder_count0": enc, "n_ranges": int(ranges.size), "n_zero_ranges": int((ranges == 0).sum()),
                   "source": "src/CUDA/csv_files/Donut_1024x16.csv"}, f, indent=1)
    with open(os.path.join(OUT, "os1_two_packets.csv"), "wb") as f:
        f.write(b"\n".join(raw_lines[: 2 * PACKET_BYTES]) + b"\n")
    with open(os.path.join(CSV, "beam_intrinsics.csv"), "rb") as f:
        data = f.read()
    with open(os.path.join(OUT, "beam_intrinsics.csv"), "wb") as f:
        f.write(data)

    # ---- bunny ------------------------------------------------------------------------------
    for name, out, sep, npts in (("Bunny_res.csv", "bunny_res", None, 8171), ("Bunny.csv", "bunny", ";", 35947)):
        with open(os.path.join(CSV, name), "rb") as f:
            data = f.read()
        lines = data.split(b"\n")
        pts = []
        for ln in lines:
            ln = ln.strip()
            if not ln:
                continue
            tok = ln.split(sep.encode() if sep else None)
            pts.append([np.float32(float(t)) for t in tok])
        xyz = np.array(pts, dtype=np.float32)
        assert xyz.shape == (npts, 3), xyz.shape
        xyz.tofile(os.path.join(OUT, out + "_xyz_f32.bin"))
        with open(os.path.join(OUT, out + "_head.csv"), "wb") as f:
            f.write(b"\n".join(lines[:64]) + b"\n")
    print("fixtures written to", OUT)


if __name__ == "__main__":
    main()
