import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from __graft_entry__ import load_package
import oracle_lib
pkg = load_package(); orc = oracle_lib.Oracle()
ctx = pkg.Context(0)
D = pkg.datasets.synthetic_grid(40, np.float32)
M = pkg.datasets.make_model_gpu(D, *pkg.datasets.P2P_GPU)
ctx.set_model(M)
_, nbr = ctx.estimate_normals(want_neighbours=True)
want = orc.knn4(M)
bad = np.where((nbr != want).any(1))[0]
print("mismatching rows:", len(bad), bad[:10])
for i in bad[:5]:
    d = ((M - M[i]) ** 2)
    dd = (d[:, 0] + d[:, 1]) + d[:, 2]
    print(i, "gpu", nbr[i], dd[nbr[i]], "oracle", want[i], dd[want[i]])
