import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from __graft_entry__ import load_package
import oracle_lib
pkg = load_package(); orc = oracle_lib.Oracle(); orc.set_threads(16)
g = os.path.join(ROOT, "tests", "golden")
B = np.fromfile(os.path.join(g, "bunny_xyz_f32.bin"), dtype=np.float32).reshape(-1, 3)
BM = pkg.datasets.make_model_gpu(B, *pkg.datasets.BUNNY)
def run(env, K):
    for k in ("ICP_NN_SHARE", "ICP_RESIDENT", "ICP_ARMED", "ICP_NN_WAVES128", "ICP_SHARE_RESIDENT_AFTER", "ICP_NN_SHARE_RESIDENT"):
        os.environ.pop(k, None)
    os.environ.update(env)
    with pkg.Context(0) as c:
        r = c.point_to_point(B, BM, max_iter=K, tol=0.0, fixed_iterations=True)
    return r
K = 9
want = {k: orc.icp_p2p_f32x(B, BM, k, 0.0, fixed=True)["idx"] for k in (7, 8, 9, 10)}
for name, env in (("armed", {"ICP_SHARE_RESIDENT_AFTER": "-1"}), ("mixed", {}), ("resident", {"ICP_RESIDENT": "2"})):
    r = run(env, K)
    print(name, {k: int((r.idx != w).sum()) for k, w in want.items()})
