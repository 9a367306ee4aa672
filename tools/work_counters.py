#!/usr/bin/env python3
"""one full hall registration (tol 1e-6) with the work counters on: speculation statistics and executed work per pass"""
import os, sys, json, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package
pkg = load_package()
g = os.path.join(ROOT, "tests", "golden")
with pkg.Context(0) as ctx:
    r = np.fromfile(os.path.join(g, "hall_ranges_u32.bin"), dtype=np.uint32)
    alt, az = pkg.datasets.read_os1_intrinsics(os.path.join(g, "beam_intrinsics.csv"))
    P, Q = pkg.datasets.hall_clouds(ctx, r, 33616, alt, az)
    ctx.set_model(Q); ctx.set_moving(P)
    ctx.set_work_counting(True)
    ctx.reset_moving()
    ctx.loop_begin(pkg.ICP_POINT_TO_POINT, max_iter=100, tol=1e-6)
    k = 0
    prev = None
    while True:
        kk, done = ctx.loop_run(1)      # one pass per call: a new resident kernel each time -> no speculation; so also run whole below
        k += kk
        if done: break
    w1 = ctx.get_work_counters()
    ctx.reset_moving()
    ctx.loop_begin(pkg.ICP_POINT_TO_POINT, max_iter=100, tol=1e-6)
    done = False
    while not done:
        _, done = ctx.loop_run(1 << 20)
    w2 = ctx.get_work_counters()
    info = ctx.nn_launch_info()
    print(json.dumps(dict(info=info, stepped=w1, whole=w2, passes=ctx.loop_state()["passes"])))
