#!/usr/bin/env python3
"""From an ICP_NN_PHASES dump of a resident hall pass: how evenly a block's waves share its scan.
   ICP_NN_WAVES=16 ICP_NN_PHASES=ph.bin:6 python3 tools/phase_run.py 9; python3 tools/block_balance.py ph.bin 16"""
import sys, numpy as np
a = np.fromfile(sys.argv[1], dtype=np.int64)
nw = int(sys.argv[2]) if len(sys.argv) > 2 else 8
a = a[: len(a) // (10 * nw) * (10 * nw)].reshape(-1, nw, 10).astype(np.float64)
live = (a[:, :, 3] > 0).all(axis=1) & (a[:, :, 2] > 0).all(axis=1)
a = a[live]
t0 = a[:, :, 1][a[:, :, 1] > 0].min()
scan = (a[:, :, 3] - a[:, :, 2]) / 100.0          # per wave: bounds seeded -> scan done (us)
end = (a[:, :, 3] - t0) / 100.0
print(f"blocks {len(a)}, waves per block {nw}")
print(f"scan time of a wave: mean {scan.mean():.2f}  median {np.median(scan):.2f}  max {scan.max():.2f} us")
bm, bx = scan.mean(axis=1), scan.max(axis=1)
print(f"per block: mean of its waves {bm.mean():.2f} (largest block {bm.max():.2f});  slowest wave {bx.mean():.2f} (largest {bx.max():.2f})")
print(f"-> a block waits for its slowest wave {np.mean(bx - bm):.2f} us longer than an even split would take (worst block {np.max(bx - bm):.2f})")
print(f"scan done, latest wave of each block: median {np.median(end.max(axis=1)):.2f}  last block {end.max():.2f} us after the first block saw the message")
i = int(np.argmax(end.max(axis=1)))
print(f"the last block: its waves' scans {np.round(scan[i], 2).tolist()}")
