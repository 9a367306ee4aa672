#!/usr/bin/env python3
"""Which CUs does a resident launch of the 64-point-row kernel occupy?  Reads an ICP_NN_PHASES log: wave 1 of every block
leaves XCC_ID << 32 | HW_ID in its slot 8.   usage: python tools/cu_usage.py ph.bin [waves_per_block = 8]"""
import sys, numpy as np
a = np.fromfile(sys.argv[1], dtype=np.int64)
a = a[: len(a) // 10 * 10].reshape(-1, 10)
nw = int(sys.argv[2]) if len(sys.argv) > 2 else 8
nb = len(a) // nw
a = a[: nb * nw].reshape(nb, nw, 10)
live = a[:, 0, 1] > 0
a = a[live]
hw = a[:, 1, 8]
xcc = (hw >> 32) & 0xF
hwid = hw & 0xFFFFFFFF
cu = (hwid >> 8) & 0xF; sh = (hwid >> 12) & 1; se = (hwid >> 13) & 7
key = xcc * 1000 + se * 100 + sh * 20 + cu
per = np.bincount(np.unique(key, return_inverse=True)[1])
print(f"blocks {len(a)}: {len(per)} distinct CUs in use over {len(np.unique(xcc))} XCDs; blocks per occupied CU min {per.min()} max {per.max()}; "
      f"blocks per XCD {np.bincount(xcc.astype(int)).tolist()}")
