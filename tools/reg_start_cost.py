#!/usr/bin/env python3
"""What a registration costs before its first seeded pass (hall pair): loop_begin alone, and registrations cut after 1, 2, 3 steps."""
import os, sys, time, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package
pkg = load_package()
g = os.path.join(ROOT, "tests", "golden")
with pkg.Context(0) as ctx:
    r = np.fromfile(os.path.join(g, "hall_ranges_u32.bin"), dtype=np.uint32)
    alt, az = pkg.datasets.read_os1_intrinsics(os.path.join(g, "beam_intrinsics.csv"))
    P, Q = pkg.datasets.hall_clouds(ctx, r, 33616, alt, az)
    if os.environ.get("EXCL") == "1": ctx.set_exclusive(True)
    ctx.set_model(Q); ctx.set_moving(P)
    def reg(steps):
        ctx.reset_moving(); ctx.loop_begin(pkg.ICP_POINT_TO_POINT, max_iter=100, tol=1e-6)
        return ctx.loop_run(steps)[0]
    for _ in range(200): reg(13)
    n = 5000
    t0 = time.perf_counter()
    for _ in range(n): ctx.reset_moving(); ctx.loop_begin(pkg.ICP_POINT_TO_POINT, max_iter=100, tol=1e-6)
    tb = (time.perf_counter() - t0) / n
    print(f"reset_moving + loop_begin (through ctypes): {1e6 * tb:.2f} us")
    prev = 0.0
    for steps in (1, 2, 3, 4, 13):
        n = 3000
        t0 = time.perf_counter()
        for _ in range(n): reg(steps)
        t = (time.perf_counter() - t0) / n
        print(f"registration cut after {steps:2d} steps: {1e6 * t:7.2f} us  (+{1e6 * (t - prev):6.2f})")
        prev = t
