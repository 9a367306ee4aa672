#!/usr/bin/env python3
"""How long do icp_set_model / icp_set_moving / a whole icp_point_to_point call take (host preprocessing included)?"""
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
    B = np.fromfile(os.path.join(g, "bunny_xyz_f32.bin"), dtype=np.float32).reshape(-1, 3)
    BM = pkg.datasets.make_model_gpu(B, *pkg.datasets.BUNNY)
    for name, D, M in (("hall", P, Q), ("bunny", B, BM)):
        ctx.set_model(M); ctx.set_moving(D)
        t = []
        for _ in range(5):
            t0 = time.perf_counter(); ctx.set_model(M); t1 = time.perf_counter(); ctx.set_moving(D); t2 = time.perf_counter()
            res = ctx.point_to_point(D, M, max_iter=100, tol=1e-6); t3 = time.perf_counter()
            t.append((t1 - t0, t2 - t1, t3 - t2, res.iterations))
        a = np.array(t)
        print(f"{name}: set_model {1e3*a[:,0].min():.3f} ms, set_moving {1e3*a[:,1].min():.3f} ms, point_to_point (uploads + {int(a[0,3])} iterations) {1e3*a[:,2].min():.3f} ms")
