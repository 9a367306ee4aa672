#!/usr/bin/env python3
"""Read an ICP_NN_PHASES dump (last matching launch of a context): per-phase timing of the packed matching kernel.

    ICP_NN_PHASES=/tmp/ph.bin python tools/phase_run.py && python tools/phase_report.py /tmp/ph.bin [waves_per_block]

Stamps are s_memrealtime ticks (100 MHz -> 10 ns).  Phases: 0 entry, 1 points loaded (+ message received, fused
transform/error), 2 bounds seeded, 3 scan done, 4 results handed in, 5 block met, 6 results merged (or keys merged, atomics
drained), 7 moments accumulated (or ticket drawn), 8 row stores issued, 9 drained + tagged (closing wave only).  With ICP_NN_PHASES=file:p only pass p of a
resident launch is stamped (phase 0 then belongs to the launch, not to the pass)."""
import sys
import numpy as np

a = np.fromfile(sys.argv[1], dtype=np.int64)
a = a[: len(a) // 10 * 10].reshape(-1, 10)
for arg in sys.argv:   # --waves=N: only the first N waves of the log (the log is not wiped between launches)
    if arg.startswith("--waves="): a = a[: int(arg[8:])]
if "--hier" in sys.argv:
    # two-level search: waves other than wave 0 of a block leave, in slots 6..9, the ticks spent in the super-box find,
    # the chunk find and the hit processing (barriers included) and (super hits << 32 | chunk hits) of the block
    nw = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else 16
    w = np.arange(len(a)) % nw
    r = a[(w == 1) & (a[:, 1] > 0)]
    sf, cf, pr = r[:, 6] / 100.0, r[:, 7] / 100.0, r[:, 8] / 100.0
    sh, h = r[:, 9] >> 32, r[:, 9] & 0xffffffff
    for name, v in (("super find us", sf), ("chunk find us", cf), ("process us", pr), ("super hits", sh), ("chunk hits", h)):
        print(f"{name:14s} blocks {len(v):6d}  median {np.median(v):9.2f}  mean {v.mean():9.2f}  p90 {np.percentile(v, 90):9.2f}  max {v.max():9.2f}")
    a[w != 0, 6:] = 0
live = (a[:, 1:] > 0).any(axis=1)
a = a[live].astype(np.float64)
first = 0 if (a[:, 0] > 0).all() and a[:, 0].max() <= a[:, 1][a[:, 1] > 0].min() + 1e7 and (a[:, 1][a[:, 1] > 0].min() - a[:, 0].min()) < 1e4 else 1
t0 = a[:, first][a[:, first] > 0].min()
us = lambda x: (x - t0) / 100.0
print(f"waves {len(a)}; span (first phase-{first} stamp -> last stamp) {us(a[:, first:][a[:, first:] > 0].max()):.2f} us")
names = ["entry", "points/message(+transform)", "bounds seeded", "scan", "handed in", "block met", "merged", "accumulated", "row stores issued", "tagged"]
last = np.zeros(len(a))
for ph in range(first, 10):
    if ph > first:
        last = np.where(a[:, ph - 1] >= t0, a[:, ph - 1], last)  # most recent earlier stamp of the wave (of this launch)
    m = a[:, ph] > 0
    if not m.any():
        continue
    # (stamps from before this launch's first one are leftovers of an earlier launch or pass -- the log is not wiped: masked)
    m = m & (a[:, ph] >= t0)
    if not m.any():
        continue
    line = f"phase {ph} {names[ph]:27s} n={m.sum():6d}  reached at median {np.median(us(a[m, ph])):6.2f}  p90 {np.percentile(us(a[m, ph]), 90):6.2f}  last {us(a[m, ph]).max():6.2f} us"
    mm = m & (last > 0)
    if ph > first and mm.any():
        d = (a[mm, ph] - last[mm]) / 100.0
        line += f";  dt median {np.median(d):6.2f}  p90 {np.percentile(d, 90):6.2f}  max {d.max():6.2f}"
    print(line)
