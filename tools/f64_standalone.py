"""configs[0]'s clouds (src/ICP_CPU.c: WIDTH 32 and 100, a far-apart pose): the stand-alone matching launch of the fp64 path against the same
clouds in fp32 (events around every launch, 2 warm-ups), and the fp32 loop's time per pass -- what of an fp64 pass is the precision."""
import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
pkg = load_package()
for W in (32, 100):
    D = pkg.datasets.synthetic_grid(W, np.float64); M = pkg.datasets.make_model_cpu(D)
    for dt in (np.float64, np.float32):
        with pkg.Context(0) as c:
            c.set_model(M.astype(dt)); c.set_moving(D.astype(dt))
            a = c.nn_match_bench_launches(10, 2, 0); b = c.nn_match_bench_launches(10, 2, 1)
            info = c.nn_launch_info()
            r = c.point_to_point(D.astype(dt), M.astype(dt), max_iter=200, tol=1e-5)
            r = c.point_to_point(D.astype(dt), M.astype(dt), max_iter=200, tol=1e-5)
        print(f"W={W} {np.dtype(dt).name}: stand-alone matching launch seeded {1e3*a.mean():.1f} us (min {1e3*a.min():.1f}), cold {1e3*b.mean():.1f} us; "
              f"loop {r.passes} passes, {1e6*r.seconds_total/r.passes:.1f} us per pass; blocks {info['blocks']} x {info['threads']}", flush=True)
