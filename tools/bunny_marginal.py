"""Bunny.csv: what a late pass of a registration costs -- registrations of K fixed iterations, the difference between two K"""
import os, sys, time, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package
pkg = load_package()
g = os.path.join(ROOT, "tests", "golden")
with pkg.Context(0) as ctx:
    B = np.fromfile(os.path.join(g, "bunny_xyz_f32.bin"), dtype=np.float32).reshape(-1, 3)
    BM = pkg.datasets.make_model_gpu(B, *pkg.datasets.BUNNY)
    ctx.set_model(BM); ctx.set_moving(B)
    def run(K):
        ctx.reset_moving(); ctx.loop_begin(pkg.ICP_POINT_TO_POINT, max_iter=K, tol=0.0, fixed_iterations=True)
        k, d = ctx.loop_run(1 << 20); return k
    res = {}
    for K in (2, 6, 11, 16, 21):
        run(K)
        t0 = time.perf_counter()
        for _ in range(20): run(K)
        res[K] = (time.perf_counter() - t0) / 20 * 1e6
    ks = sorted(res)
    print("registration of K iterations:", ", ".join(f"K={k}: {res[k]:.0f} us" for k in ks))
    print("per pass between:", ", ".join(f"{a}->{b}: {(res[b] - res[a]) / (b - a):.1f} us" for a, b in zip(ks, ks[1:])))
