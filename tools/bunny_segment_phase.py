"""Bunny.csv run as segments (see bunny_segments.py), four registrations; the last one stops after its FIRST segment, so that an ICP_NN_PHASES log
of pass p < the segment's length holds that segment's pass p.  usage: ICP_NN_PHASES=ph.bin:1 python tools/bunny_segment_phase.py 3"""
import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package
pkg = load_package()
g = os.path.join(ROOT, "tests", "golden")
first = int(sys.argv[1]) if len(sys.argv) > 1 else 3
with pkg.Context(0) as ctx:
    B = np.fromfile(os.path.join(g, "bunny_xyz_f32.bin"), dtype=np.float32).reshape(-1, 3)
    BM = pkg.datasets.make_model_gpu(B, *pkg.datasets.BUNNY)
    ctx.set_model(BM); ctx.set_moving(B)
    for k in range(4):
        ctx.reset_moving(); ctx.loop_begin(pkg.ICP_POINT_TO_POINT, max_iter=100, tol=1e-6)
        it, done = ctx.loop_run(first)
        if k < 3:
            while not done: it, done = ctx.loop_run(1 << 20)
    print("stopped after", first, "passes of the fourth registration")
