"""synthetic grids of several sizes against a 4096-point / a full-size model: us per iteration of a short registration"""
import os, sys, time, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package
pkg = load_package()
for width, full in [(int(w), True) for w in os.environ["GRID_WIDTHS"].split(",")] if os.environ.get("GRID_WIDTHS") else ((192, False), (224, True), (256, True), (300, True), (362, True), (512, True)):
    D = pkg.datasets.synthetic_grid(width, np.float32)
    M = pkg.datasets.make_model_gpu(D if full else D[:4096], *pkg.datasets.P2P_GPU)
    with pkg.Context(0) as ctx:
        ctx.set_model(M); ctx.set_moving(D)
        def run():
            ctx.reset_moving(); ctx.loop_begin(pkg.ICP_POINT_TO_POINT, max_iter=12, tol=0.0, fixed_iterations=True)
            k, d = ctx.loop_run(1 << 20); return k
        run()
        t0 = time.perf_counter(); ks = [run() for _ in range(5)]; dt = time.perf_counter() - t0
        info = ctx.nn_launch_info()
        print(f"grid {width}x{width} = {len(D)} points, model {len(M)}: {1e6 * dt / sum(ks):.1f} us per iteration; blocks {info['blocks']} x {info['threads']} threads", flush=True)
