import os, sys, time, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package
pkg = load_package()
W = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
with pkg.Context(0) as ctx:
    D = pkg.datasets.synthetic_grid(W, np.float32)
    t0 = time.perf_counter(); ctx.set_model(D); ctx.set_moving(D); t1 = time.perf_counter()
    ms = ctx.nn_match_resident()
    idx = ctx.get_indices()
    print(f"{W*W} x {W*W}: set-up {1e3*(t1-t0):.1f} ms, cold matching kernel {ms:.3f} ms, self-match ok: {bool(np.array_equal(idx, np.arange(W*W)))}")
    M = pkg.datasets.make_model_gpu(D, *pkg.datasets.P2P_GPU)
    ctx.set_model(M)
    ms = ctx.nn_match_resident(); i1 = ctx.get_indices()
    ms2 = ctx.nn_match_bench(3, seeded=True) / 3
    print(f"moved copy: cold {ms:.3f} ms, seeded {ms2:.3f} ms; idx range ok: {int(i1.min()) >= 0 and int(i1.max()) < W*W}")
    # spot check 200 random points against brute force on the host
    rng = np.random.default_rng(1); s = rng.integers(0, W*W, 200)
    d = ((D[s, None, :].astype(np.float32) - M[None, :, :].astype(np.float32)) ** 2)
    dd = (d[:, :, 0] + d[:, :, 1]) + d[:, :, 2]
    print("spot check vs numpy argmin:", bool(np.array_equal(dd.argmin(1), i1[s])))
