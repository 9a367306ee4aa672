#!/usr/bin/env python3
"""configs[4] (10 M x 10 M on one GPU): the kernel's work counters pass by pass -- one registration of fixed iterations, run one pass
per icp_loop_run call with the instrumented instantiation, the counters read (and differenced) after every pass.
usage: python tools/s5_work_per_pass.py [passes=12] [points=10000000]"""
import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package
pkg = load_package()
K = int(sys.argv[1]) if len(sys.argv) > 1 else 12
N = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000_000
W = int(np.ceil(np.sqrt(N)))
D = pkg.datasets.synthetic_grid(W, np.float32)[:N]
M = pkg.datasets.make_model_gpu(D, *pkg.datasets.P2P_GPU)
with pkg.Context(0) as ctx:
    ctx.set_model(M); ctx.set_moving(D)
    rows = ctx.nn_launch_info()["blocks"]
    ctx.set_work_counting(True)
    ctx.loop_begin(pkg.ICP_POINT_TO_POINT, max_iter=K, tol=0.0, fixed_iterations=True)
    ctx.get_work_counters(reset=True)
    for p in range(1, K + 1):
        ctx.loop_run(1)
        d = ctx.get_work_counters(reset=True)   # (zeroed by the read: one pass's work)
        print(f"pass {p:2d}: per row: boxes tested in the find {d['find_boxes'] / rows:8.0f} (+ upper levels {d['upper_boxes'] / rows:6.0f}), chunks listed {d['hits_box'] / rows:7.0f}, "
              f"past the per-point box test {d['hits_xy'] / rows:6.0f}, evaluated in full {d['hits_full'] / rows:6.0f}, sample groups {d['sample_groups'] / rows:5.0f}, block passes {d['block_passes'] / rows:5.2f}", flush=True)
        if os.environ.get("S5_RAW"):   # (an experimental build that reuses slots 8..11: see the file that names it)
            print("         raw slots 8..11:", d['spec_lists'], d['spec_covered'], d['spec_hits'], d['list_hits'], " hits_full", d['hits_full'], flush=True)
