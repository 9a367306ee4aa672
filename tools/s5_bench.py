#!/usr/bin/env python3
"""BASELINE configs[4] through bench.py's launch contract: the synthetic 10 M-point cloud (W = 3163 grid
z = x^2 - y^2 truncated to 1e7 points; model = the moved copy, point-to-point constants), the MOVING cloud sharded over
the ranks, the model replicated, one sum of 32 doubles per iteration (shared host memory on one node).

    python tools/s5_bench.py [--gpus 1] [--steps 30] [--warmup 5] [--points 10000000]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P tools/s5_bench.py --gpus N ...

A step is one ICP iteration of the WHOLE cloud (strong scaling: the total work is fixed as N grows); the timed region is
`steps` iterations of one registration from the initial pose (fixed iteration count, cold first pass), bracketed by a
barrier + device synchronisation, max over ranks.  Rank 0 prints one JSON line.  Not run by the driver (bench.py is the
hall workload the metric is quoted on); ICP_BENCH_ONE_DEVICE=1 rehearses several ranks on one GPU (gloo)."""
import argparse, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--points", type=int, default=10_000_000)
    args = ap.parse_args()
    import torch
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    one_device = os.environ.get("ICP_BENCH_ONE_DEVICE") == "1"
    if one_device:
        local_rank = 0
    if args.gpus != world and world == 1 and args.gpus > 1:
        raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run --nproc-per-node N")
    dist, saved_stdout = None, None
    if world > 1:
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)        # (RCCL prints a banner on stdout)
        import torch.distributed as dist
        os.environ.setdefault("MASTER_PORT", "29534")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="gloo" if one_device else "nccl", rank=rank, world_size=world)
    from __graft_entry__ import load_package
    pkg = load_package()
    N = args.points
    W = int(np.ceil(np.sqrt(N)))
    D = pkg.datasets.synthetic_grid(W, np.float32)[:N]
    M = pkg.datasets.make_model_gpu(D, *pkg.datasets.P2P_GPU)
    lo, cnt = pkg.shard_range(N, rank, world)
    P = np.ascontiguousarray(D[lo:lo + cnt])
    del D
    ctx = pkg.Context(local_rank)
    t0 = time.perf_counter()
    ctx.set_model(M)
    ctx.set_moving(P)
    setup_s = time.perf_counter() - t0
    if world > 1:
        pkg.distributed.attach_local_comm(ctx, dist)

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def run(iters):
        ctx.reset_moving()
        ctx.loop_begin(pkg.ICP_POINT_TO_POINT, max_iter=iters, tol=1e-6, fixed_iterations=True)
        k, done = 0, False
        while not done:
            kk, done = ctx.loop_run(1 << 20)
            k += kk
        return k

    if args.warmup > 0:
        run(args.warmup)
    sync()
    t0 = time.perf_counter()
    k = run(args.steps)
    sync()
    dt = time.perf_counter() - t0
    st = ctx.loop_state()
    t_max = dt
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device="cpu" if one_device else f"cuda:{local_rank}")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        t_max = float(tt.item())
    if saved_stdout is not None:
        sys.stdout.flush()
        os.dup2(saved_stdout, 1)
        os.close(saved_stdout)
    if rank == 0:
        print(json.dumps({
            "metric": "ICP iterations/sec, synthetic 10M-point cloud (BASELINE configs[4])",
            "value": args.steps / t_max, "unit": "iterations/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * t_max / args.steps, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic z = x^2 - y^2 grid, W = %d truncated to %d points; model = moved copy" % (W, N),
            "config": {"workload": "synthetic 10M-point cloud point-to-point ICP, moving cloud sharded over the ranks",
                       "model_points": N, "moving_points_per_gpu": cnt, "global_moving_points": N,
                       "regime": "one registration of `steps` fixed iterations from the initial pose (cold first pass)",
                       "iterations_run": k, "set_up_ms_rank0": 1e3 * setup_s,
                       "collective": "sum of 32 doubles per iteration through shared host memory (icp_comm_init_local)" if world > 1 else "none"},
            "pairs_per_s_algorithmic": float(N) * float(N) * args.steps / t_max,
            "rms_error_series_head": [float(e) for e in st["err"][:6]], "final_rms_error": float(st["err"][-1]),
        }), flush=True)
    if world > 1:
        ctx.comm_destroy()
    ctx.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
