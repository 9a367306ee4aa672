"""Bunny.csv: a registration run as SEGMENTS of passes -- icp_loop_run(k) calls, each a resident launch of its own whose shared rows are
dealt by the counts of the segment before -- against one launch for the whole registration.  usage: python tools/bunny_segments.py "6,8" "4,6,6" ... """
import os, sys, time, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package
pkg = load_package()
g = os.path.join(ROOT, "tests", "golden")
plans = [[int(v) for v in a.split(",") if v] for a in sys.argv[1:]] or [[], [6], [4, 8], [3, 5, 8], [2, 4, 6, 8]]
with pkg.Context(0) as ctx:
    B = np.fromfile(os.path.join(g, "bunny_xyz_f32.bin"), dtype=np.float32).reshape(-1, 3)
    BM = pkg.datasets.make_model_gpu(B, *pkg.datasets.BUNNY)
    ctx.set_model(BM); ctx.set_moving(B)
    def run(plan):
        ctx.reset_moving(); ctx.loop_begin(pkg.ICP_POINT_TO_POINT, max_iter=100, tol=1e-6)
        total, done = 0, False
        for k in plan + [1 << 20]:
            if done: break
            it, done = ctx.loop_run(k); total += it
        return total
    for plan in plans:
        for _ in range(3): run(plan)
        t0 = time.perf_counter(); ks = [run(plan) for _ in range(20)]; dt = time.perf_counter() - t0
        st = ctx.loop_state()
        print(f"segments {plan or 'none'}: {1e6 * dt / sum(ks):.2f} us per iteration ({sum(ks) // 20} iterations, err {st['err'][-1]:.9g})", flush=True)
