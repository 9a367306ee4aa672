import os, sys, time, hashlib, numpy as np
ROOT = "/root/repo" if os.path.isdir("/root/repo/tests") else os.environ["GRAFT_REPO_ROOT"]
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package
pkg = load_package()
B = np.fromfile(os.path.join(ROOT, "tests", "golden", "bunny_xyz_f32.bin"), dtype=np.float32).reshape(-1, 3)
BM = pkg.datasets.make_model_gpu(B, *pkg.datasets.BUNNY)
for name, env in (("shared", {}), ("waves16", {"ICP_NN_WAVES128": "16"}), ("resident_shared_2nd", {"ICP_RESIDENT": "2"}), ("stepwise", {"ICP_RESIDENT": "0", "ICP_ARMED": "0"})):
    for k in ("ICP_NN_WAVES128", "ICP_RESIDENT", "ICP_ARMED"):
        os.environ.pop(k, None)
    os.environ.update(env)
    with pkg.Context(0) as c:
        for _ in range(2):
            t0 = time.perf_counter(); r = c.point_to_plane(B, BM, max_iter=30, tol=1e-6); dt = time.perf_counter() - t0
        h = hashlib.sha256()
        for a in (r.T, r.err, r.idx): h.update(np.ascontiguousarray(a).tobytes())
        print(f"{name:22s} iterations {r.iterations} loop {1e6 * r.seconds_total / max(1, r.passes):.1f} us per pass, err {r.err[-1]:.6g}, digest {h.hexdigest()[:16]}")
