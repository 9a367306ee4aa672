"""BASELINE configs[4] on ONE GPU: the share of one rank of 8 -- a 1.25 M-point slice of the 10 M-point synthetic
cloud (W = 3163 grid truncated to 1e7 points, model = moved copy with the point-to-point constants)
matched against the whole 10 M-point model.

    python tools/s5_time.py [rank 0..7] [world 8] [iterations 3] [--dense]

prints set-up, cold and seeded matching passes, a spot check against numpy brute force, (--dense) a full index comparison
against the kernel that executes every pair, and a short registration."""
import os, sys, time, numpy as np
DENSE = "--dense" in sys.argv   # the kernel choice is read once per process: the dense comparison is a run of its own
if DENSE:
    os.environ["ICP_NN_SPARSE"] = "0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package
pkg = load_package()
args = [a for a in sys.argv[1:] if not a.startswith("--")]
rank = int(args[0]) if len(args) > 0 else 0
world = int(args[1]) if len(args) > 1 else 8
iters = int(args[2]) if len(args) > 2 else 3
NPTS = int(os.environ.get("S5_POINTS", 10_000_000))
W = int(np.ceil(np.sqrt(NPTS)))
D = pkg.datasets.synthetic_grid(W, np.float32)[:NPTS]
M = pkg.datasets.make_model_gpu(D, *pkg.datasets.P2P_GPU)
lo, cnt = pkg.shard_range(NPTS, rank, world); hi = lo + cnt
P = np.ascontiguousarray(D[lo:hi])
if os.environ.get("S5_SHARD_BLOCK"):   # (dealt in blocks of that many points, round-robin: distributed.shard_cyclic_index; S5_SHARD_CURVE=1: along a Z-order curve)
    order = pkg.distributed.curve_order(D, bits=int(os.environ.get("S5_CURVE_BITS", 10))) if os.environ.get("S5_SHARD_CURVE") else None
    P = np.ascontiguousarray(D[pkg.distributed.shard_cyclic_index(NPTS, rank, world, int(os.environ["S5_SHARD_BLOCK"]), order)])
print(f"model {len(M)} points, moving shard [{lo}, {hi}) = {len(P)} points (rank {rank} of {world})", flush=True)
SAVE = os.path.join(ROOT, "gpurun_out", f"s5_idx_{NPTS}_{rank}_{world}.npy")
if DENSE:
    with pkg.Context(0) as c2:
        c2.set_model(M); c2.set_moving(P)
        dms = c2.nn_match_resident(timed=True); i_d = c2.get_indices()
    print(f"dense kernel (every pair executed) {dms:.1f} ms = {len(P)*len(M)/dms/1e9:.2f} T pairs/s", flush=True)
    if os.path.exists(SAVE):
        print(f"all {len(P)} indices identical to the sparse kernel's: {bool(np.array_equal(i_d, np.load(SAVE)))}", flush=True)
    sys.exit(0)
with pkg.Context(0) as ctx:
    t0 = time.perf_counter(); ctx.set_model(M); t1 = time.perf_counter(); ctx.set_moving(P); t2 = time.perf_counter()
    print(f"set-up: model {1e3*(t1-t0):.1f} ms, moving {1e3*(t2-t1):.1f} ms; launch {ctx.nn_launch_info()}", flush=True)
    cold = ctx.nn_match_resident(timed=True); i_cold = ctx.get_indices()
    print(f"cold matching pass {cold:.3f} ms", flush=True)
    seeded = ctx.nn_match_bench(3, seeded=True) / 3
    i_seed = ctx.get_indices()
    print(f"seeded matching pass {seeded:.3f} ms; same indices as cold: {bool(np.array_equal(i_cold, i_seed))}", flush=True)
    rng = np.random.default_rng(1); s = rng.integers(0, len(P), 64)
    ok = True
    for i in s:
        d = (P[i][None, :] - M) ** 2
        dd = (d[:, 0] + d[:, 1]) + d[:, 2]
        ok &= int(dd.argmin()) == int(i_cold[i])
    print("spot check of 64 points against numpy brute force:", bool(ok), flush=True)
    os.makedirs(os.path.dirname(SAVE), exist_ok=True); np.save(SAVE, i_cold)
    if iters > 0:
        t0 = time.perf_counter()
        r = ctx.point_to_point(P, M, max_iter=iters, tol=1e-6, fixed_iterations=True)
        t1 = time.perf_counter()
        print(f"{iters} iterations: {1e3*(t1-t0):.1f} ms wall (incl. set-up), loop {1e3*r.seconds_total:.1f} ms, matching {1e3*r.seconds_nn:.1f} ms,"
              f" err {np.asarray(r.err)[:iters]}", flush=True)
