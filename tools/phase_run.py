#!/usr/bin/env python3
"""Run a short hall registration so that the last matching launch is a seeded, fused one (for ICP_NN_PHASES)."""
import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package
pkg = load_package()
g = os.path.join(ROOT, "tests", "golden")
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 6
with pkg.Context(0) as ctx:
    r = np.fromfile(os.path.join(g, "hall_ranges_u32.bin"), dtype=np.uint32)
    alt, az = pkg.datasets.read_os1_intrinsics(os.path.join(g, "beam_intrinsics.csv"))
    P, Q = pkg.datasets.hall_clouds(ctx, r, 33616, alt, az)
    res = ctx.point_to_point(P, Q, max_iter=iters, tol=1e-6, fixed_iterations=True)
    print("iterations", res.iterations, "err", res.err[-1])
