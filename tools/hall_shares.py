#!/usr/bin/env python3
"""The hall cloud split over N ranks, what each rank's share costs ALONE on one GPU: 13 fixed iterations per registration (the pair's
own count), back to back -- us per iteration of every rank's share; a sharded registration is as slow as its slowest rank (+ the exchange).
usage: hall_shares.py [world=8] [contiguous|rows]   (rows: rows of 64 points along a Hilbert curve dealt to the ranks)"""
import os, sys, time, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package
pkg = load_package()
g = os.path.join(ROOT, "tests", "golden")
world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
form = sys.argv[2] if len(sys.argv) > 2 else "contiguous"
with pkg.Context(0) as c0:
    r = np.fromfile(os.path.join(g, "hall_ranges_u32.bin"), dtype=np.uint32)
    alt, az = pkg.datasets.read_os1_intrinsics(os.path.join(g, "beam_intrinsics.csv"))
    P, Q = pkg.datasets.hall_clouds(c0, r, 33616, alt, az)
order = pkg.distributed.curve_order(P) if form == "rows" else None
out = []
for rank in range(world):
    if form == "rows":
        Ps = np.ascontiguousarray(P[pkg.distributed.shard_cyclic_index(len(P), rank, world, 64, order)])
    else:
        lo, cnt = pkg.shard_range(len(P), rank, world); Ps = np.ascontiguousarray(P[lo:lo + cnt])
    with pkg.Context(0) as ctx:
        ctx.set_model(Q); ctx.set_moving(Ps)
        def run(n):
            for _ in range(n):
                ctx.reset_moving(); ctx.loop_begin(pkg.ICP_POINT_TO_POINT, max_iter=13, tol=0.0, fixed_iterations=True)
                ctx.loop_run(1 << 20)
        run(20)
        t0 = time.perf_counter(); run(300); dt = time.perf_counter() - t0
        out.append(1e6 * dt / (300 * 13))
print(f"hall, {world} ranks, {form}: us per iteration of each rank's share alone: " + " ".join(f"{v:.2f}" for v in out) + f"   max {max(out):.2f} mean {np.mean(out):.2f}")
