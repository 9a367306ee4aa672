#!/usr/bin/env python3
"""Phase log of a launch of nn_match_sparse (flat search; ICP_NN_PHASES=file): per block its role (row, part of parts), the
hits on its list and how long its waves took from 'bounds seeded' to 'scan done' -- the slowest blocks first.
usage: share_report.py ph.bin [waves_per_block=8]"""
import sys
import numpy as np
a = np.fromfile(sys.argv[1], dtype=np.int64)
nw = int(sys.argv[2]) if len(sys.argv) > 2 else 8
a = a[: len(a) // (10 * nw) * 10 * nw].reshape(-1, nw, 10)
live = (a[:, :, 3] > 0).any(axis=1) & (a[:, 1, 6] != 0)
# (the log is not cleared between launches: blocks without a role in the last launch still show an earlier one -- keep the last 150 us)
live &= a[:, :, 3].max(axis=1) > a[:, :, 3].max() - 15000
a = a[live]
role = a[:, 1, 6]; hits = a[:, 1, 7]; prev = a[:, 1, 8] & 0xffffffff; T = a[:, 1, 8] >> 32
row, part, parts = role >> 32, (role >> 16) & 0xffff, role & 0xffff
scan = (a[:, :, 3] - a[:, :, 2]).max(axis=1) / 100.0
t0 = a[:, :, 1][a[:, :, 1] > 0].min()
end = (a[:, :, 3].max(axis=1) - t0) / 100.0
tail_end = (a[:, 0, 9][a[:, 0, 9] > 0].max() - t0) / 100.0
print(f"pass in the kernel (first message seen -> last row tagged): {tail_end:.2f} us; front (entry -> points + transform) median {np.median((a[:, :, 1] - a[:, :, 0])[a[:, :, 0] > 0]) / 100.0:.2f} us")
w2 = a[:, 2, :].astype(np.float64)
ok = (w2[:, 6] > 0) & (w2[:, 7] > 0) & (w2[:, 2] > 0) & (w2[:, 3] > 0)
if ok.any():
    f = lambda x: f"median {np.median(x) / 100.0:.2f} p90 {np.percentile(x, 90) / 100.0:.2f} max {x.max() / 100.0:.2f}"
    print(f"   wave 2: bounds -> list complete {f(w2[ok, 6] - w2[ok, 2])};  -> first batch fetched {f(w2[ok, 7] - w2[ok, 6])};  -> scan done {f(w2[ok, 3] - w2[ok, 7])} us")
if "--brief" in sys.argv:
    print(f"   blocks {len(a)} hits/block mean {hits.mean():.0f} max {hits.max()}  scan median {np.median(scan):.2f} max {scan.max():.2f} us")
    sys.exit(0)
print(f"blocks with a role: {len(a)}; rows {len(np.unique(row))}; parts: max {parts.max()}, rows split {len(np.unique(row[parts > 1]))}")
print(f"hits per block: mean {hits.mean():.0f} median {np.median(hits):.0f} p90 {np.percentile(hits, 90):.0f} max {hits.max()};  scan us: median {np.median(scan):.2f} p90 {np.percentile(scan, 90):.2f} max {scan.max():.2f}")
o = np.argsort(-scan)[:12]
for k in o:
    print(f"  block row {row[k]:4d} part {part[k]:2d}/{parts[k]:2d}  hits {hits[k]:5d}  (the row last time: {prev[k]:5d}, target per block {T[k]:4d})  scan {scan[k]:6.2f} us  scan done at {end[k]:6.2f} us")
c = np.corrcoef(hits, scan)[0, 1]
ok_ = np.isfinite(hits) & np.isfinite(scan) & (np.abs(scan) < 1e6)   # (stamps of an older launch among the last one's: wipe = 1 in the ICP_NN_PHASES spec avoids them)
fit = np.polyfit(hits[ok_], scan[ok_], 1) if ok_.sum() > 2 else (float('nan'), float('nan'))
print(f"scan time vs hits: {fit[1]:.2f} us + {1e3 * fit[0]:.1f} ns per hit (correlation {c:.2f})")
first = part == 0
print(f"sum of the rows' hits last time (blocks with part 0): {prev[first].sum()}; this time: {hits.sum()}")
o = np.argsort(-prev[first])[:10]
print("heaviest rows last time:", [(int(row[first][k]), int(prev[first][k]), int(parts[first][k])) for k in o])
print("targets per block seen by the blocks:", sorted(set(int(v) for v in T)))
