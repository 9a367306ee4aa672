"""configs[0] on the GPU: src/ICP_CPU.c's run (fp64, tol 1e-5, MAX_ITER 200) at WIDTH 32 and 100 through the fp64 path."""
import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
pkg = load_package()
for W in (32, 100):
    D = pkg.datasets.synthetic_grid(W, np.float64); M = pkg.datasets.make_model_cpu(D)
    with pkg.Context(0) as ctx:
        r = ctx.point_to_point(D, M, max_iter=200, tol=1e-5)
        t0 = time.perf_counter(); r = ctx.point_to_point(D, M, max_iter=200, tol=1e-5); dt = time.perf_counter() - t0
    print(f"fp64 W={W} ({W*W} points): {r.iterations} iterations, E={r.err[-1]:.5f}, {1e3*dt:.2f} ms whole call, {1e6*r.seconds_total/max(1,r.passes):.1f} us per pass")
