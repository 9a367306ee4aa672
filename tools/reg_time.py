#!/usr/bin/env python3
"""Back-to-back hall registrations (the bench's regime, no per-launch kernel timing): iterations/s as the loop runs by default.
usage: reg_time.py [iterations] [plane]"""
import os, sys, time, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package
pkg = load_package()
g = os.path.join(ROOT, "tests", "golden")
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
plane = len(sys.argv) > 2 and sys.argv[2] == "plane"
with pkg.Context(0) as ctx:
    r = np.fromfile(os.path.join(g, "hall_ranges_u32.bin"), dtype=np.uint32)
    alt, az = pkg.datasets.read_os1_intrinsics(os.path.join(g, "beam_intrinsics.csv"))
    P, Q = pkg.datasets.hall_clouds(ctx, r, 33616, alt, az)
    ctx.set_model(Q); ctx.set_moving(P)
    if plane:
        ctx.estimate_normals()
    def run(count):
        left, regs = count, 0
        while left > 0:
            ctx.reset_moving()
            ctx.loop_begin(pkg.ICP_POINT_TO_PLANE if plane else pkg.ICP_POINT_TO_POINT, max_iter=50 if plane else 100, tol=1e-6, fixed_iterations=False)
            k, _ = ctx.loop_run(left)
            left -= k; regs += 1
        return regs
    run(100)
    t0 = time.perf_counter(); regs = run(steps); dt = time.perf_counter() - t0
    st = ctx.loop_state()
    print(f"{steps} iterations in {regs} registrations: {1e6*dt/steps:.2f} us/iteration, {steps/dt:.0f} it/s; last err {st['err'][-1]:.6g}")
