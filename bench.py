#!/usr/bin/env python3
"""bench.py -- the hot path's headline measurement on MI355X.

    python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run)

Workload (BASELINE.json `metric`, configs[2]): point-to-point ICP on the hall LiDAR scan, 16 384 moving x
16 384 model points, fp32, clouds resident in HBM.  A *step* is one full ICP iteration of the loop in
libicp_mi355x.so (icp_loop_run): ONE resident kernel per registration; per iteration the host sends a mailbox
message (command, R, t), the kernel does [transform + error of the previous pass] -> matching (exact, sparse) ->
moment rows into pinned host memory, the host adds the rows in fixed order as they arrive and solves the 3x3 SVD.
The K timed steps are the iterations of back-to-back REAL registrations of the pair (tol 1e-6, MAX_ITER 100 as in
src/CUDA/GPU_point_to_point_real.cu): each one restarts from the pristine moving cloud, pays its kernel launch and
its cold first matching pass and stops by the reference's rule -- not K iterations of an already converged pose.

N > 1 (weak scaling): every rank holds a hall-sized shard of the moving cloud (the global moving cloud is
N x 16 384 points) and the full model; the only data that crosses ranks is the 32-double moment vector, summed once
per iteration -- by default through shared host memory (icp_comm_init_local: the vector is already on the host, the
ranks of one node exchange 256 bytes in ~1 us and keep their resident kernels); ICP_BENCH_COMM=rccl uses ONE RCCL
all-reduce issued by the library on the loop's stream (one kernel launch per pass), ICP_BENCH_COMM=torch the same
through torch.distributed.  `value` = (N x K shard-iterations) / max-over-ranks time.

One JSON line on stdout (rank 0).  Extra objects: `roofline` (the loop's kernel timed with HIP events inside the
timed region + the stand-alone matching kernel), `cpu_baseline` (the CPU oracle on this box's host cores, bounded
sample).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

FP32_PEAK_TFLOPS = 157.3   # MI355X fp32 vector == fp32-input MFMA peak (MI355X_MICROARCH.md)
HBM_PEAK_GBPS = 8000.0     # HBM3E spec peak (MI355X_MICROARCH.md)


def hall_fixture():
    """(P, Q, source) -- the hall pair in metres, fp32 AoS, built by the PRODUCT path"""
    from __graft_entry__ import load_package
    pkg = load_package()
    g = os.path.join(ROOT, "tests", "golden")
    ranges = np.fromfile(os.path.join(g, "hall_ranges_u32.bin"), dtype=np.uint32)
    enc = json.load(open(os.path.join(g, "hall_meta.json")))["encoder_count0"]
    alt, az = pkg.datasets.read_os1_intrinsics(os.path.join(g, "beam_intrinsics.csv"))
    return pkg, ranges, enc, alt, az


def cpu_baseline(P, Q, budget_s=12.0):
    """the CPU oracle (scalar C restatement of src/ICP_CPU.c's loop in fp32) on this host, bounded sample"""
    import oracle_lib
    orc = oracle_lib.Oracle()
    t0 = time.perf_counter()
    orc.icp_p2p(P, Q, 1, 0.0, fixed=True)
    one = time.perf_counter() - t0
    iters = max(2, min(40, int(budget_s / max(one, 1e-3))))
    t0 = time.perf_counter()
    r = orc.icp_p2p(P, Q, iters, 0.0, fixed=True)
    dt = time.perf_counter() - t0
    assert r["passes"] == iters
    out = {"value": iters / dt, "unit": "iterations/s", "cores": 1, "kind": "port",
           "sample": f"{iters} fixed point-to-point iterations of the same hall workload (16384x16384, fp32), "
                     f"oracle/icp_oracle.c single thread, {dt:.1f} s",
           "host_cpus": os.cpu_count()}
    # beside it: the same port with its matching loop spread over the cores this process may use (OpenMP over the
    # moving points; the minimisation stays scalar) -- the strongest thing the host can do with the reference's algorithm
    cores = len(os.sched_getaffinity(0))
    try:   # a container's CPU quota counts, not the CPUs it can see (threads beyond it only queue up)
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            cores = max(1, min(cores, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    if cores > 1:
        orc.set_threads(cores)
        try:
            orc.icp_p2p(P, Q, 2, 0.0, fixed=True)          # (thread pool start-up)
            it2 = 200
            t0 = time.perf_counter()
            r = orc.icp_p2p(P, Q, it2, 0.0, fixed=True)
            dt2 = time.perf_counter() - t0
            out["all_cores"] = {"value": it2 / dt2, "unit": "iterations/s", "cores": cores,
                                "sample": f"{it2} fixed iterations, matching loop over {cores} OpenMP threads, {dt2:.1f} s"}
        finally:
            orc.set_threads(1)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # rehearsal on a one-GPU box: ICP_BENCH_ONE_DEVICE=1 puts every rank on device 0 (gloo carries the set-up messages,
    # RCCL refuses two ranks on one device); never used by the driver
    one_device = os.environ.get("ICP_BENCH_ONE_DEVICE") == "1"
    if one_device:
        local_rank = 0
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run --nproc-per-node N")
    dist = None
    # ICP_BENCH_FORCE_DIST=1 drives the multi-GPU code path (RCCL all-reduce, torch stream, device finalize)
    # with a single rank, so that it can be rehearsed on a one-GPU box
    use_dist = world > 1 or os.environ.get("ICP_BENCH_FORCE_DIST") == "1"
    saved_stdout = None
    if use_dist:
        # RCCL prints a version banner on STDOUT when its first communicator is created; the contract is ONE JSON
        # line on stdout, so fd 1 points at stderr until the result is printed
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
        import torch.distributed as dist
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="gloo" if one_device else "nccl", rank=rank, world_size=world)

    pkg, ranges, enc, alt, az = hall_fixture()
    ctx = pkg.Context(local_rank)          # raises when the HIP library / device is missing
    P, Q = pkg.datasets.hall_clouds(ctx, ranges, enc, alt, az)
    n, m = P.shape[0], Q.shape[0]
    K, W = args.steps, args.warmup

    ctx.set_model(Q)
    ctx.set_moving(P)
    mom = None
    stream_ctx = None
    native_comm = False
    local_comm = False
    comm_kind = os.environ.get("ICP_BENCH_COMM", "local")   # local (shared host memory, default) | rccl | torch
    if use_dist and comm_kind == "local":
        # one node: the loop's 32-double vector is already in host memory when the rows have been added, so the ranks
        # exchange it through shared memory (~1 us; a 256-byte RCCL all-reduce costs more than the whole iteration)
        # and every rank keeps its resident kernel.  torch.distributed only carries the 128-byte segment id.
        try:
            pkg.distributed.attach_local_comm(ctx, dist)
            local_comm = native_comm = True
        except Exception as e:  # noqa: BLE001
            print(f"[bench] host-memory communicator unavailable ({e}); using RCCL", file=sys.stderr)
    if use_dist and not native_comm and comm_kind != "torch":
        # preferred: the library issues the all-reduce itself (RCCL bound at run time, icp_comm_init); torch only
        # broadcasts the 128-byte communicator id.  Any failure falls back to the torch.distributed collective.
        try:
            torch.cuda.set_device(local_rank)
            pkg.distributed.attach_native_comm(ctx, dist)
            native_comm = True
        except Exception as e:  # noqa: BLE001
            print(f"[bench] native RCCL communicator unavailable ({e}); using torch.distributed", file=sys.stderr)
    if use_dist and not native_comm:
        # The loop writes its moment vector straight into a torch tensor and runs on a torch-owned (non-default)
        # stream that is also torch's CURRENT stream while the loop runs, so the RCCL all-reduce is ordered behind
        # the finalize kernel, and the D2H behind the all-reduce, without any host synchronisation in between.
        torch.cuda.set_device(local_rank)
        mom = torch.zeros(pkg.ICP_NMOM, dtype=torch.float64, device=f"cuda:{local_rank}")
        torch.cuda.synchronize()
        side = torch.cuda.Stream(device=local_rank)
        stream_ctx = torch.cuda.stream(side)
        stream_ctx.__enter__()
        ctx.set_stream(side.cuda_stream)
        ctx.loop_set_moments_dev(mom.data_ptr())

    def step():
        ctx.loop_enqueue()          # native_comm: finalize kernel + ncclAllReduce on the loop's stream
        if use_dist and not native_comm:
            dist.all_reduce(mom)
        return ctx.loop_complete()

    def sync():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    STRIDE = int(os.environ.get("ICP_BENCH_TIMING_STRIDE", "7"))
    ctx.set_profiling(STRIDE)   # HIP events around every 7th launch of the loop's kernel (with a resident kernel: every 7th registration)
    in_library = not (use_dist and not native_comm)   # nothing Python has to do between the steps
    TOL, MAX_ITER = 1e-6, 100   # src/CUDA/GPU_point_to_point_real.cu:18,404-405
    stats = {"registrations": 0, "iterations": 0}

    def run_steps(count):
        """`count` ICP iterations, executed as back-to-back REAL registrations of the hall pair: every registration
        starts from the pristine moving cloud (device-to-device reset, inside the timed region), begins with a cold
        matching pass and iterates until the reference's stop rule fires; the last one is cut when `count` is reached."""
        left = count
        while left > 0:
            ctx.reset_moving()
            ctx.loop_begin(pkg.ICP_POINT_TO_POINT, max_iter=MAX_ITER, tol=TOL, fixed_iterations=False)
            stats["registrations"] += 1
            if in_library:
                k, _ = ctx.loop_run(left)                 # (enqueue + complete) x k inside libicp_mi355x
            else:
                k, done = 0, False
                while not done and k < left:
                    done = step()
                    k += 1
            left -= k
            stats["iterations"] += k

    run_steps(W)
    ctx.set_profiling(STRIDE)    # restart the kernel-time accumulators for the timed region
    stats = {"registrations": 0, "iterations": 0}
    sync()
    t0 = time.perf_counter()
    run_steps(K)
    sync()
    dt = time.perf_counter() - t0
    sec1, cnt1 = ctx.loop_timing()
    passes1 = ctx.loop_timing_passes()
    sec0, cnt0 = 0.0, 0
    st = ctx.loop_state()

    t_max = dt
    if use_dist:
        tt = torch.tensor([dt], dtype=torch.float64, device="cpu" if one_device else f"cuda:{local_rank}")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        t_max = float(tt.item())

    if saved_stdout is not None:
        sys.stdout.flush()
        os.dup2(saved_stdout, 1)
        os.close(saved_stdout)
    if rank == 0:
        info = ctx.nn_launch_info()
        nn_launches = max(1, cnt1 - cnt0)
        nn_avg_s = (sec1 - sec0) / nn_launches
        passes_per_launch = max(1, passes1) / nn_launches    # 1 when every pass is its own launch; a resident kernel runs a whole registration
        flops_pass = 8.0 * n * m                             # 3 sub + 3 mul + 2 add per pair
        flops = flops_pass * passes_per_launch
        alg_bytes = (12.0 * n + 12.0 * m + 4.0 * n) * passes_per_launch   # read P, read Q, write idx (fp32), per pass
        # back-to-back launches of the stand-alone matching kernel, no other work between: the kernel-quality figure
        b2b_ms = ctx.nn_match_bench(50) / 50.0
        traffic, traffic_src = None, None
        pmc = os.path.join(ROOT, "profiles", "r1", "17_pmc_hbm_traffic_sparse.json")
        if world == 1 and os.path.exists(pmc):
            # HBM bytes per launch of the seeded stand-alone matching kernel from the committed rocprofv3 PMC passes
            # (FETCH_SIZE and WRITE_SIZE in separate runs, KiB units, FETCH doubled: gfx950 correction), scaled to the
            # passes one timed launch runs
            rec = next((v for k, v in json.load(open(pmc)).items() if "nn_match_sparse" in k), None)
            if rec:
                traffic = rec["hbm_bytes_corrected"] * passes_per_launch
                traffic_src = ("profiles/r1/17_pmc_hbm_traffic_sparse.json: per pass FETCH_SIZE %.0f B raw (x2 corrected) + WRITE_SIZE %.0f B, "
                               "times %.2f passes per launch" % (rec["fetch_bytes_raw"], rec["write_bytes"], passes_per_launch))
        out = {
            "metric": "ICP iterations/sec + NN HBM GB/s (% roofline), hall cloud",
            "value": world * K / t_max,
            "unit": "iterations/s",
            "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": 1e3 * t_max / K,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "hall LiDAR scan fixture (tests/golden/hall_ranges_u32.bin, decoded from the reference's "
                    "Donut_1024x16.csv; polar->Cartesian by the device kernel)",
            "config": {"workload": "hall LiDAR scan point-to-point ICP (BASELINE configs[2])", "moving_points_per_gpu": n,
                       "model_points": m, "global_moving_points": n * world,
                       "regime": "back-to-back full registrations from the initial pose (cold first pass, tol 1e-6, stop rule on); "
                                 "a step = one iteration of such a registration",
                       "registrations_timed": stats["registrations"],
                       "iterations_per_registration": stats["iterations"] / max(1, stats["registrations"]),
                       "collective": (("sum of the loop's moment vector (the 19 doubles point-to-point uses) per iteration over the node's ranks through shared host memory "
                                       "(icp_comm_init_local), rank order, every rank keeps its resident kernel") if local_comm else
                                      ("1 RCCL all-reduce of 32 doubles per iteration, issued by "
                                       + ("libicp_mi355x on the loop's stream" if native_comm else "torch.distributed"))) if use_dist else "none"},
            "roofline": {
                "kernel": ("nn_match_sparse<1>, resident: ONE launch per REGISTRATION, every block on the machine (%.2f matching passes on average); every "
                           "pass = mailbox message from the host (command, R, t) -> [transform + error of the previous pass] -> "
                           "lane-parallel chunk-box search -> hit processing (packed fp32, exact arithmetic) -> LDS key merge -> moment row "
                           "to the host.  The duration INCLUDES the host round trips between the passes (the kernel waits for every solve)."
                           % passes_per_launch) if passes_per_launch > 1.5 else
                          "nn_match_sparse<1>: one launch per iteration = [transform + error of the previous pass] + matching + moment rows",
                "bound": "mfma",
                "bound_detail": "compute roof, fp32 dense peak 157.3 TFLOP/s -- on gfx950 the same figure for MFMA and for packed vector FMA.  "
                                "The kernel issues packed VALU ops (v_pk_add/mul_f32): MFMA cannot evaluate (dx*dx + dy*dy) + dz*dz "
                                "bit-exactly, and the contract forbids FMA, which caps EXECUTED arithmetic at 0.5 of that roof.",
                "bound_note": "brute-force NN is 4681 flop/B on this cloud (ridge ~20): compute-bound, not HBM-bound.  'achieved' counts the ALGORITHMIC 8*N*M flop of every pass; the kernel skips most of them "
                              "(exactly: results are bit-identical to the full scan), so frac measures time-to-solution against the "
                              "brute-force roofline, not executed instructions.",
                "executed_fraction_estimate": {"pairs": 0.0105, "source": "tools/pair_stats.py (numpy, hall pair at a steady pass): 28 of 2048 chunks "
                                               "survive the group-box test per block, 21.6 are evaluated in full"},
                "achieved": flops / nn_avg_s / 1e12, "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": flops / nn_avg_s / 1e12 / FP32_PEAK_TFLOPS,
                "traffic": traffic, "traffic_source": traffic_src,
                "flops_per_launch": flops, "avg_launch_us": 1e6 * nn_avg_s, "launches_timed": nn_launches,
                "passes_per_launch": passes_per_launch, "avg_pass_us": 1e6 * nn_avg_s / passes_per_launch,
                "pairs_per_s": n * m * passes_per_launch / nn_avg_s,
                "matching_only": {"what": "the stand-alone matching kernel (no transform, no moment rows; icp_nn_match_bench_ex: 50 back-to-back "
                                          "seeded launches of nn_match_sparse<0>), i.e. the part the 8*N*M flop belong to",
                                  "avg_launch_us": 1e3 * b2b_ms, "achieved": flops_pass / (1e-3 * b2b_ms) / 1e12,
                                  "frac": flops_pass / (1e-3 * b2b_ms) / 1e12 / FP32_PEAK_TFLOPS},
                "launch": info,
                "hbm": {"algorithmic_bytes_per_launch": alg_bytes, "achieved_GBps": alg_bytes / nn_avg_s / 1e9,
                        "peak_GBps": HBM_PEAK_GBPS, "frac": alg_bytes / nn_avg_s / 1e9 / HBM_PEAK_GBPS},
            },
            "final_rms_error": float(st["err"][-1]),
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(P, Q)
        print(json.dumps(out), flush=True)
    if stream_ctx is not None:
        ctx.loop_set_moments_dev(0)
        ctx.set_stream(0)
        stream_ctx.__exit__(None, None, None)
    if native_comm:
        ctx.comm_destroy()
    ctx.close()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
