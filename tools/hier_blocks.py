#!/usr/bin/env python3
"""Phase log of a launch of the hierarchical 16-wave kernel (ICP_NN_PHASES=file:pass:slots:wipe, slots large enough for every block):
how long the blocks took, what that adds up to per CU, and how long the launch lasted -- the difference is imbalance.
usage: hier_blocks.py ph.bin [cus=256] [waves per block=16] [blocks of the launch: the log is not wiped between launches, and
an earlier launch of more or larger blocks leaves its stamps behind the last one's]"""
import sys, numpy as np
NW = int(sys.argv[3]) if len(sys.argv) > 3 else 16
a = np.fromfile(sys.argv[1], dtype=np.int64); a = a[: len(a) // (NW * 10) * (NW * 10)].reshape(-1, NW, 10)
cus = int(sys.argv[2]) if len(sys.argv) > 2 else 256
if len(sys.argv) > 4: a = a[: int(sys.argv[4])]
# (run with wipe = 1 in the ICP_NN_PHASES spec: the log then holds the last launch only.  A spare block leaves nothing; a part of a split row
# that does not close it ends at its ticket, phase 7; the closing block at phase 9)
live = a[:, 0, 0] > 0
a = a[live]
start = a[:, :, 0].min(1)
end = np.maximum(a[:, 0, :].max(1), a[:, :, :6].max(axis=(1, 2)))   # (slots 6..9 of the other waves hold durations and counts, not stamps)
dur = (end - start) / 100.0
hits = a[:, 1, 9] & 0xffffffff
t0 = start.min(); span = (end.max() - t0) / 100.0
print(f"blocks {len(a)}; launch span {span:.0f} us; block time: median {np.median(dur):.1f} mean {dur.mean():.1f} p99 {np.percentile(dur, 99):.1f} max {dur.max():.0f} us; "
      f"sum / {cus} CUs = {dur.sum() / cus:.0f} us")
print(f"chunk hits per block: median {np.median(hits):.0f} mean {hits.mean():.0f} p99 {np.percentile(hits, 99):.0f} max {hits.max()}; blocks above 10x the median: {(hits > 10 * np.median(hits)).sum()}, "
      f"they hold {100.0 * hits[hits > 10 * np.median(hits)].sum() / hits.sum():.0f} % of the hits")
o = np.argsort(-dur)[:10]
for k in o:
    print(f"  block {k:6d}: {dur[k]:7.0f} us, chunk hits {hits[k]:7d}, started at {(start[k] - t0) / 100.0:7.0f} us, ended at {(end[k] - t0) / 100.0:7.0f} us")
late = (end - t0) / 100.0 > 0.5 * span
print(f"blocks still running after half the span: {late.sum()}")
