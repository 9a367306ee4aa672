"""Bunny.csv soak: registrations for a number of seconds through one loop form (the caller's environment); every one must end
with the same iteration count, transform, error series and correspondences as the first."""
import hashlib, os, sys, time, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package
pkg = load_package()
seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 30.0
B = np.fromfile(os.path.join(ROOT, "tests", "golden", "bunny_xyz_f32.bin"), dtype=np.float32).reshape(-1, 3)
BM = pkg.datasets.make_model_gpu(B, *pkg.datasets.BUNNY)
def digest(r):
    h = hashlib.sha256()
    for a in (r.T, r.err, r.idx):
        h.update(np.ascontiguousarray(a).tobytes())
    return r.iterations, h.hexdigest()[:16]
with pkg.Context(0) as ctx:
    first = digest(ctx.point_to_point(B, BM, max_iter=100, tol=1e-6))
    t0 = time.perf_counter(); regs = its = 0
    while time.perf_counter() - t0 < seconds:
        d = digest(ctx.point_to_point(B, BM, max_iter=100, tol=1e-6))
        if d != first:
            print(f"MISMATCH after {regs} registrations: {d} != {first}"); sys.exit(1)
        regs += 1; its += d[0]
    print(f"{regs} registrations, {its} iterations in {time.perf_counter() - t0:.0f} s, all {first}")
