#!/usr/bin/env python3
"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs as the
MI355X guide prescribes).  Counter unit: KiB.  gfx950 correction: FETCH_SIZE reads 1/2 of the bytes of a wide
(16 B/lane) coalesced stream -> the read side is reported raw and doubled."""
import csv, glob, json, sys, collections

def load(d, name):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == name:
            acc[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    return acc

fetch = load(sys.argv[1], "FETCH_SIZE")
write = load(sys.argv[2], "WRITE_SIZE")
out = {}
for k in sorted(set(fetch) | set(write)):
    f = fetch.get(k, [0]); w = write.get(k, [0])
    fa, wa = sum(f) / len(f) * 1024, sum(w) / len(w) * 1024
    out[k] = dict(launches=len(f), fetch_bytes_raw=fa, fetch_bytes_x2=2 * fa, write_bytes=wa, hbm_bytes_corrected=2 * fa + wa)
    print(f"{k[-44:]:45s} n={len(f):4d} FETCH {fa/1e3:10.1f} KB (x2: {2*fa/1e3:10.1f})  WRITE {wa/1e3:10.1f} KB")
if len(sys.argv) > 3:
    json.dump(out, open(sys.argv[3], "w"), indent=1)
