#!/usr/bin/env python3
"""Read an ICP_NN_PHASES dump (last matching launch of a context): per-phase timing of the packed matching kernel.

    ICP_NN_PHASES=/tmp/ph.bin python tools/phase_run.py && python tools/phase_report.py /tmp/ph.bin

Stamps are s_memrealtime ticks (100 MHz -> 10 ns).  Phases: 0 entry, 1 points loaded (+ fused transform/error),
2 seeds gathered, 3 scan done, 4 index recovered, 5 block met, 6 keys merged (atomics drained), 7 ticket drawn,
8 row stored, 9 fenced + tagged (closing wave only)."""
import sys
import numpy as np

a = np.fromfile(sys.argv[1], dtype=np.int64)
a = a[: len(a) // 10 * 10].reshape(-1, 10)
live = a[:, 0] > 0
a = a[live].astype(np.float64)
t0 = a[:, 0].min()
us = lambda x: (x - t0) / 100.0
print(f"waves {len(a)}; kernel span (first entry -> last stamp) {us(a[a > 0].max()):.2f} us")
print(f"wave entry: median {np.median(us(a[:, 0])):.2f} us, p90 {np.percentile(us(a[:, 0]), 90):.2f}, last {us(a[:, 0]).max():.2f}")
names = ["entry", "points(+transform)", "seeds", "scan", "recover", "block met", "keys merged", "ticket", "row stored", "tagged"]
last = np.zeros(len(a))
for ph in range(1, 10):
    last = np.where(a[:, ph - 1] > 0, a[:, ph - 1], last)  # most recent earlier stamp of the wave
    m = (a[:, ph] > 0) & (last > 0)
    if not m.any():
        continue
    prev = last[m]
    d = (a[m, ph] - prev) / 100.0
    print(f"phase {ph} {names[ph]:20s} n={m.sum():6d}  dt median {np.median(d):6.2f}  p90 {np.percentile(d, 90):6.2f}  max {d.max():6.2f} us;"
          f"  reached at median {np.median(us(a[m, ph])):6.2f}  last {us(a[m, ph]).max():6.2f} us")
