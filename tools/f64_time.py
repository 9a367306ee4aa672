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
# the hall pair widened to fp64 (clouds that are nearly aligned: the case the sparse structure is made for)
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
g = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
with pkg.Context(0) as ctx:
    r = np.fromfile(os.path.join(g, "hall_ranges_u32.bin"), dtype=np.uint32)
    alt, az = pkg.datasets.read_os1_intrinsics(os.path.join(g, "beam_intrinsics.csv"))
    P, Q = pkg.datasets.hall_clouds(ctx, r, 33616, alt, az)
    P, Q = P.astype(np.float64), Q.astype(np.float64)
    res = ctx.point_to_point(P, Q, max_iter=200, tol=1e-5)
    t0 = time.perf_counter(); res = ctx.point_to_point(P, Q, max_iter=200, tol=1e-5); dt = time.perf_counter() - t0
print(f"fp64 hall (16384 points): {res.iterations} iterations, E={res.err[-1]:.3g}, {1e3*dt:.2f} ms whole call, {1e6*res.seconds_total/max(1,res.passes):.1f} us per pass")
