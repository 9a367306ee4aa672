#!/usr/bin/env python3
"""Bunny registration cut after K passes, so that the last matching launch (what ICP_NN_PHASES logs) is pass K.
usage: ICP_NN_PHASES=ph.bin python tools/bunny_phase.py [K] && python tools/phase_report.py ph.bin 16"""
import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package
pkg = load_package()
g = os.path.join(ROOT, "tests", "golden")
K = int(sys.argv[1]) if len(sys.argv) > 1 else 3
with pkg.Context(0) as ctx:
    B = np.fromfile(os.path.join(g, "bunny_xyz_f32.bin"), dtype=np.float32).reshape(-1, 3)
    BM = pkg.datasets.make_model_gpu(B, *pkg.datasets.BUNNY)
    for _ in range(int(os.environ.get("REPEAT", "1"))):   # (REPEAT=2: the second registration's first pass goes by the first one's counts)
        r = ctx.point_to_point(B, BM, max_iter=K, tol=1e-6, fixed_iterations=True)
    print("passes", r.passes, "err", r.err[-1], "loop ms", 1e3 * r.seconds_total)
