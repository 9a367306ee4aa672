import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
pkg = load_package()
ctx = pkg.Context(0)
print("ctx ok", flush=True)
D = pkg.datasets.synthetic_grid(32, np.float32)
M = pkg.datasets.make_model_standard(D)
ctx.set_model(M); print("model ok", flush=True)
ctx.set_moving(D); print("moving ok", flush=True)
ctx.nn_match_resident(); print("nn ok", flush=True)
print(ctx.get_indices()[:10], flush=True)
