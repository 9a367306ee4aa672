#!/usr/bin/env python3
"""sub-stamps of the row64 kernel's scan phase (waves other than 0): 2 bounds -> 6 coverage decided -> 7 box stage done -> 3 scan done"""
import sys, numpy as np
a = np.fromfile(sys.argv[1], dtype=np.int64)
a = a[: len(a) // 10 * 10].reshape(-1, 10).astype(np.float64)
nw = int(sys.argv[2]) if len(sys.argv) > 2 else 8
w = np.arange(len(a)) % nw
m = (w != 0) & (a[:, 2] > 0) & (a[:, 3] > 0)
b = a[m]
for name, lo, hi in (("bounds -> coverage decided", 2, 6), ("coverage -> box stage done", 6, 7), ("box stage -> scan done", 7, 3), ("bounds -> scan done", 2, 3)):
    ok = (b[:, lo] > 0) & (b[:, hi] > 0)
    d = (b[ok, hi] - b[ok, lo]) / 100.0
    if len(d):
        print(f"{name:30s} n={len(d):5d} median {np.median(d):5.2f} p90 {np.percentile(d, 90):5.2f} max {d.max():5.2f} us")
print("waves with a coverage stamp:", int((b[:, 6] > 0).sum()), "of", len(b))
