"""Bunny.csv (35 947 x 35 947, the reference's second dataset): us per iteration of whole registrations, and a digest of the
result (iterations, transform, error series, correspondences) to compare the forms of the loop by."""
import hashlib, os, sys, time, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package
pkg = load_package()
g = os.path.join(ROOT, "tests", "golden")
with pkg.Context(0) as ctx:
    B = np.fromfile(os.path.join(g, "bunny_xyz_f32.bin"), dtype=np.float32).reshape(-1, 3)
    BM = pkg.datasets.make_model_gpu(B, *pkg.datasets.BUNNY)
    r = ctx.point_to_point(B, BM, max_iter=100, tol=1e-6)
    h = hashlib.sha256(); h.update(np.ascontiguousarray(r.T).tobytes()); h.update(np.ascontiguousarray(r.err).tobytes()); h.update(np.ascontiguousarray(r.idx).tobytes())
    info = ctx.nn_launch_info()
    ctx.set_model(BM); ctx.set_moving(B)
    def run():
        ctx.reset_moving(); ctx.loop_begin(pkg.ICP_POINT_TO_POINT, max_iter=100, tol=1e-6)
        k, d = ctx.loop_run(1 << 20); return k
    run()
    t0 = time.perf_counter(); ks = [run() for _ in range(20)]; dt = time.perf_counter() - t0
    print(f"bunny: {sum(ks)} iterations in 20 registrations: {1e6*dt/sum(ks):.2f} us/iteration; threads/block {info['threads']}; "
          f"result: {r.iterations} iterations, err {r.err[-1]:.9g}, digest {h.hexdigest()[:16]}")
