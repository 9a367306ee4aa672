#!/usr/bin/env python3
"""One stand-alone launch of the dense packed kernel on the hall pair, for its phase log:
   ICP_NN_SPARSE=0 ICP_NN_CULL=0 ICP_NN_PHASES=ph.bin python tools/dense_phase.py && python tools/phase_report.py ph.bin"""
import os, sys, numpy as np
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
    for _ in range(3):
        ctx.nn_match_resident()
    print("launch", ctx.nn_launch_info(), "ms", ctx.nn_match_resident(timed=True))
