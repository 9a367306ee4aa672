"""Registrations at the borders of the plan's size classes (rows of 64 / 128, flat / hierarchical search, shared / ordered rows,
resident / one launch per pass): two fixed iterations through the plan the library picks, against the dense kernel that executes
every pair (a context created under ICP_NN_SPARSE=0) -- same correspondences, same transform.
usage: python tools/size_sweep.py [max pairs per pass = 3e10] [--plane | --f64]   (point-to-plane with normals estimated on the device; the fp64 path)"""
import os, sys, time, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package
pkg = load_package()
ARGS = [a for a in sys.argv[1:] if not a.startswith("--")]
PLANE, F64 = "--plane" in sys.argv, "--f64" in sys.argv
LIMIT = float(ARGS[0]) if ARGS else 3e10
NS = [130, 8192, 16384, 32768, 32769, 33000, 57344, 57345, 65536, 70000, 131072, 200000]
MS = [4096, 65535, 65536, 65537, 131071, 131072, 131073, 524287, 524288, 524289]
if PLANE: NS, MS = [130, 16384, 32768, 33000, 57345, 65536, 70000], [4096, 65536, 65537, 131072]
if F64: NS, MS = [130, 8192, 16384, 16385, 33000], [4096, 65536, 131071, 131072]
DT = np.float64 if F64 else np.float32
METRIC = pkg.ICP_POINT_TO_PLANE if PLANE else pkg.ICP_POINT_TO_POINT
W = 725   # 525 625 grid points
G = pkg.datasets.synthetic_grid(W, np.float32)
rng = np.random.default_rng(5)
def run(P, M, dense):
    if dense: os.environ["ICP_NN_SPARSE"] = "0"
    else: os.environ.pop("ICP_NN_SPARSE", None)
    with pkg.Context(0) as c:
        c.set_model(M.astype(DT)); c.set_moving(P.astype(DT))
        if PLANE: c.estimate_normals()
        info = c.nn_launch_info()
        c.loop_begin(METRIC, max_iter=2, tol=0.0, fixed_iterations=True)
        done = False
        while not done: _, done = c.loop_run(1 << 20)
        return c.loop_state(), c.loop_indices(), info
bad = 0
for m in MS:
    M = pkg.datasets.make_model_gpu(np.ascontiguousarray(G[:m]), *pkg.datasets.P2P_GPU)
    for n in NS:
        if float(n) * m > LIMIT: continue
        P = np.ascontiguousarray(G[np.sort(rng.choice(len(G), n, replace=False))] if n != m else G[:n])
        try:
            t0 = time.perf_counter(); st, idx, info = run(P, M, False); dt = time.perf_counter() - t0
            sd, idd, _ = run(P, M, True)
            ok = st["iterations"] == sd["iterations"] == 2 and np.array_equal(idx, idd) and np.allclose(st["T"], sd["T"], rtol=0, atol=1e-5)
            print(f"n {n:7d} m {m:7d}: blocks {info['blocks']:5d} x {info['threads']:4d} threads  {'ok' if ok else 'MISMATCH'}  ({(idx != idd).sum()} indices differ)", flush=True)
        except Exception as e:  # noqa: BLE001
            ok = False
            print(f"n {n:7d} m {m:7d}: FAILED {str(e)[:160]}", flush=True)
        bad += 0 if ok else 1
print("mismatches or failures:", bad)
