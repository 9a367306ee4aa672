import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package
pkg = load_package()
D = pkg.datasets.synthetic_grid(192, np.float32)
M = pkg.datasets.make_model_gpu(D[:4096], *pkg.datasets.P2P_GPU)
forms = {"shared": {}, "unshared": {"ICP_NN_SHARE": "0"}, "resident": {"ICP_RESIDENT": "2"}, "stepwise": {"ICP_RESIDENT": "0", "ICP_ARMED": "0"},
         "stepwise_unshared": {"ICP_RESIDENT": "0", "ICP_ARMED": "0", "ICP_NN_SHARE": "0"}, "waves16": {"ICP_NN_WAVES128": "16"}, "waves16_stepwise": {"ICP_NN_WAVES128": "16", "ICP_RESIDENT": "0", "ICP_ARMED": "0"}}
res = {}
for name, env in forms.items():
    for k in ("ICP_NN_SHARE", "ICP_RESIDENT", "ICP_ARMED", "ICP_NN_WAVES128"):
        os.environ.pop(k, None)
    os.environ.update(env)
    with pkg.Context(0) as c:
        res[name] = c.point_to_point(D, M, max_iter=6, tol=1e-6)
ref = res["waves16_stepwise"]
for name, r in res.items():
    print(f"{name:20s} it {r.iterations} T equal {np.array_equal(r.T, ref.T)} idx equal {np.array_equal(r.idx, ref.idx)} err diff (ulps of the value) {[(float(a - b) / np.spacing(b)) if b else 0.0 for a, b in zip(r.err, ref.err)]}")
