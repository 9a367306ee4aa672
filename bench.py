#!/usr/bin/env python3
"""bench.py -- the hot path's headline measurement on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config hall|s5]

N > 1: either launched by `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...` (RANK / LOCAL_RANK /
WORLD_SIZE / MASTER_* from the environment), or plainly as `python bench.py --gpus N`, in which case this process starts
the N ranks itself as fresh child processes (before anything here has touched a GPU) and relays rank 0's line.

Workload `hall` (default; BASELINE.json `metric`, configs[2]): point-to-point ICP on the hall LiDAR scan, 16 384 moving x
16 384 model points, fp32, clouds resident in HBM.  A *step* is one full ICP iteration of the loop in libicp_mi355x.so
(icp_loop_run): ONE resident kernel per registration; per iteration the host sends a mailbox message (command, R, t),
the kernel does [transform + error of the previous pass] -> matching (exact, sparse) -> moment rows into pinned host
memory, the host adds the rows in fixed order as they arrive and solves the 3x3 SVD.  The K timed steps are the
iterations of back-to-back REAL registrations of the pair (tol 1e-6, MAX_ITER 100 as in
src/CUDA/GPU_point_to_point_real.cu): each one restarts from the pristine moving cloud, pays its kernel launch and its
cold first matching pass and stops by the reference's rule -- not K iterations of an already converged pose.
N > 1 is weak scaling: every rank holds a hall-sized shard of the moving cloud (the global moving cloud is N x 16 384
points) and the full model; the only data that crosses ranks is the 32-double moment vector, summed once per iteration.
`value` uses the node-local route (icp_comm_init_local: the vector is already in host memory, the ranks exchange it
through shared memory in ~1-2 us and keep their resident kernels); the same K steps are then repeated with ONE RCCL
all-reduce per iteration issued by the library on the loop's stream (icp_comm_init) and reported beside it as `rccl`.

Workload `s5` (BASELINE configs[4]): synthetic z = x^2 - y^2 grid truncated to --points (default 10 M) points, model = the
moved copy; the MOVING cloud is sharded over the ranks (strong scaling), the model replicated.  A step is one iteration
of the whole cloud; the timed region is one registration of K fixed iterations from the initial pose.

One JSON line on stdout (rank 0).  Extra objects: `roofline` (the loop's kernel timed with HIP events on its own stream
inside the timed region, the EXECUTED arithmetic counted by the kernel's instrumented instantiation, the stand-alone
matching kernel by the reference's min-of-10 method and the dense kernel that executes every pair), `cpu_baseline` (the
CPU oracle on this box's host cores, bounded sample).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

FP32_PEAK_TFLOPS = 157.3   # MI355X fp32 vector (packed FMA) == fp32 MFMA peak (MI355X_MICROARCH.md)
HBM_PEAK_GBPS = 8000.0     # HBM3E peak (MI355X_MICROARCH.md)
TOL, MAX_ITER = 1e-6, 100  # src/CUDA/GPU_point_to_point_real.cu:18,404-405

# fp32 operations behind each tally of the sparse kernel's instrumented instantiation (include/icp_mi355x.h,
# icp_get_work_counters; sub/mul/add counted, compare/select/min not -- the convention of SURVEY 8d's 8 flop per pair):
#   box test of one point or one group box against one chunk box: 6 sub + 3 mul + 2 add + 1 mul (safety factor) = 12
#   xy half of one pair: 2 sub + 2 mul + 1 add = 5;  z half: 1 sub + 1 mul + 1 add = 3;  a full pair: 8
#   transform of one point: 9 mul + 9 add = 18 (every one of a block's 16 waves re-derives the block's 128 points)
FLOP_BOX, FLOP_XY, FLOP_Z, FLOP_PAIR, FLOP_RT = 12, 5, 3, 8, 18
PTS_PER_HIT = 128          # one hit = one wave (64 lanes x 2 moving points) against one 8-point model chunk
WAVES_PER_BLOCK = 16


def executed_flop(work, pts_per_hit=PTS_PER_HIT, waves_per_block=WAVES_PER_BLOCK):
    """fp32 flop the sparse kernel EXECUTED, from its work counters (the fp64 moment sums of the row tail -- ~45 flop per
    point and pass -- are left out: another unit, <2 % of the total).  pts_per_hit: the moving points one wave holds
    (128 with rows of 128 points, 64 with rows of 64); waves_per_block: every wave re-derives the block's points"""
    parts = {
        "find (group box vs chunk boxes)": (work["find_boxes"] + work["upper_boxes"]) * FLOP_BOX,
        "per-point box tests": work["hits_box"] * pts_per_hit * FLOP_BOX,
        "xy halves": work["hits_xy"] * pts_per_hit * 8 * FLOP_XY,
        "z halves": work["hits_full"] * pts_per_hit * 8 * FLOP_Z,
        "cold-start samples": work["sample_groups"] * pts_per_hit * 8 * FLOP_PAIR,
        "seed distances": work["block_passes"] * waves_per_block * pts_per_hit * FLOP_PAIR,
        "transforms": work["block_transforms"] * waves_per_block * pts_per_hit * FLOP_RT,
    }
    return float(sum(parts.values())), {k: float(v) for k, v in parts.items()}


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


# ---- N > 1 without a launcher: start the ranks ourselves -------------------------------------------------------------
def spawn_ranks(n):
    """`python bench.py --gpus N` with no WORLD_SIZE: N fresh child processes, one rank each.  This (parent) process has
    made no GPU call -- torch is not even imported yet -- and never will: it only relays rank 0's line."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr))
    out0, _ = procs[0].communicate()
    rcs = [procs[0].returncode] + [p.wait() for p in procs[1:]]
    sys.stdout.write(out0.decode())
    sys.stdout.flush()
    bad = [rc for rc in rcs if rc != 0]
    return bad[0] if bad else 0


class Ranks:
    """the control plane of a multi-rank run: torch.distributed over gloo (barriers, the max over ranks, the 128-byte
    communicator ids).  The data path never goes through it."""

    def __init__(self, rank, world):
        self.rank, self.world = rank, world
        self.dist = None
        if world > 1:
            import torch.distributed as dist
            os.environ.setdefault("MASTER_PORT", "29533")
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)
            self.dist = dist

    def barrier(self):
        if self.dist:
            self.dist.barrier()

    def max(self, v):
        if not self.dist:
            return v
        import torch
        t = torch.tensor([v], dtype=torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def min_int(self, v):
        if not self.dist:
            return v
        import torch
        t = torch.tensor([v], dtype=torch.int64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MIN)
        return int(t.item())

    def close(self):
        if self.dist:
            self.dist.barrier()
            self.dist.destroy_process_group()


STUCK = []   # helper threads that never came back (a communicator attempt that hangs)


def guarded(fn, seconds):
    """run fn() on a helper thread; (True, result) or (False, reason) when it raised or did not return in time -- the
    stuck thread is abandoned (the process leaves through os._exit)"""
    box = {}

    def body():
        try:
            box["ok"] = fn()
        except Exception as e:  # noqa: BLE001
            box["err"] = f"{type(e).__name__}: {e}"
    t = threading.Thread(target=body, daemon=True)
    t.start()
    t.join(seconds)
    if t.is_alive():
        STUCK.append(t)
        return False, f"no answer within {seconds} s"
    if "err" in box:
        return False, box["err"]
    return True, box.get("ok")


# ---- the hall workload -----------------------------------------------------------------------------------------------
def run_hall(args, rank, local_rank, world):
    import numpy as np
    import torch
    from __graft_entry__ import load_package
    pkg = load_package()
    ranks = Ranks(rank, world)
    K, W = args.steps, args.warmup

    g = os.path.join(ROOT, "tests", "golden")
    ranges = np.fromfile(os.path.join(g, "hall_ranges_u32.bin"), dtype=np.uint32)
    enc = json.load(open(os.path.join(g, "hall_meta.json")))["encoder_count0"]
    alt, az = pkg.datasets.read_os1_intrinsics(os.path.join(g, "beam_intrinsics.csv"))
    ctx = pkg.Context(local_rank)          # raises when the HIP library / a gfx950 device is missing
    P, Q = pkg.datasets.hall_clouds(ctx, ranges, enc, alt, az)   # polar -> Cartesian by the device kernel
    n, m = P.shape[0], Q.shape[0]
    ctx.set_model(Q)
    ctx.set_moving(P)

    def sync():
        ranks.barrier()
        torch.cuda.synchronize(local_rank)

    def run_steps(count, stats=None):
        """`count` ICP iterations, executed as back-to-back REAL registrations of the hall pair: every registration
        starts from the pristine moving cloud (reset inside the timed region), begins with a cold matching pass and
        iterates until the reference's stop rule fires; the last one is cut when `count` is reached."""
        left = count
        while left > 0:
            ctx.reset_moving()
            ctx.loop_begin(pkg.ICP_POINT_TO_POINT, max_iter=MAX_ITER, tol=TOL, fixed_iterations=False)
            k, _ = ctx.loop_run(left)                 # (enqueue + complete) x k inside libicp_mi355x
            left -= k
            if stats is not None:
                stats["registrations"] += 1
                stats["iterations"] += k

    def timed(count, stride):
        stats = {"registrations": 0, "iterations": 0}
        ctx.set_profiling(stride)     # (also restarts the stride: the first launch after this is a timed one)
        sync()
        t0 = time.perf_counter()
        run_steps(count, stats)
        sync()
        dt = ranks.max(time.perf_counter() - t0)
        sec, cnt = ctx.loop_timing()
        passes = ctx.loop_timing_passes()
        ctx.set_profiling(0)
        return dt, stats, sec, cnt, passes

    use_local = world > 1 or os.environ.get("ICP_BENCH_FORCE_DIST") == "1"
    if use_local and ranks.dist:
        pkg.distributed.attach_local_comm(ctx, ranks.dist)

    # set-up, untimed: one complete registration (loads the code objects, sizes the work buffers) whose result is the
    # bench's own sanity check -- every rank registers the same pair, so it must end like the single-GPU run does
    ctx.reset_moving()
    ctx.loop_begin(pkg.ICP_POINT_TO_POINT, max_iter=MAX_ITER, tol=TOL, fixed_iterations=False)
    done = False
    while not done:
        _, done = ctx.loop_run(1 << 20)
    st = ctx.loop_state()
    passes_full = int(st["passes"])
    if not (2 <= passes_full <= MAX_ITER and float(st["err"][-1]) < 1e-3):
        raise SystemExit(f"[bench] the hall registration did not converge: {passes_full} passes, rms error {st['err'][-1]}")

    run_steps(W)
    # HIP events around the loop's kernel inside the timed region: with a resident kernel a launch is a whole registration,
    # so every 7th is timed when the region holds many of them (1-2 % of overhead).  A region of a few registrations is
    # not bracketed at all -- the two event records and the wait for the kernel's end would be a tenth of what is being
    # measured; the roofline leg then comes from the fixed block of 20 registrations run right after it
    regs_expected = max(1, K // max(1, passes_full))
    stride = 7 if regs_expected >= 70 else 0
    dt, stats, sec1, cnt1, passes1 = timed(K, stride)
    out = None
    if rank == 0:
        out = {
            "metric": "ICP iterations/sec + NN HBM GB/s (% roofline), hall cloud",
            "value": world * K / dt, "unit": "iterations/s",
            "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": 1e3 * dt / K,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "hall LiDAR scan fixture (tests/golden/hall_ranges_u32.bin, decoded from the reference's "
                    "Donut_1024x16.csv; polar->Cartesian by the device kernel)",
            "config": {"workload": "hall LiDAR scan point-to-point ICP (BASELINE configs[2])", "moving_points_per_gpu": n,
                       "model_points": m, "global_moving_points": n * world,
                       "regime": "back-to-back full registrations from the initial pose (cold first pass, tol 1e-6, stop rule on); "
                                 "a step = one iteration of such a registration",
                       "registrations_timed": stats["registrations"],
                       "iterations_per_registration": stats["iterations"] / max(1, stats["registrations"]),
                       "passes_of_a_full_registration": passes_full,
                       "collective": ("sum of the loop's moment vector (19 doubles) per iteration over the node's ranks through shared "
                                      "host memory (icp_comm_init_local), rank order, every rank keeps its resident kernel; the RCCL "
                                      "route is measured beside it (`rccl`)") if use_local else "none"},
            "final_rms_error": float(st["err"][-1]),
        }

    if use_local and ranks.dist:
        ctx.comm_destroy()     # rank 0 measures its roofline leg alone: no communicator may wait for the others
    if rank == 0:
        info = ctx.nn_launch_info()
        flops_pass = float(FLOP_PAIR) * n * m                 # the brute-force scan's arithmetic
        alg_bytes_pass = 12.0 * n + 12.0 * m + 4.0 * n        # read P, read Q, write idx (fp32)
        # (1) the loop's kernel, timed inside the timed region (events on the loop's own stream)
        region = None
        if cnt1 > 0:
            region = {"launches_timed": cnt1, "avg_launch_us": 1e6 * sec1 / cnt1, "passes_per_launch": passes1 / cnt1,
                      "timing_stride": stride}
        # ... and over a fixed block of 20 registrations after it (K-independent: the figure of a short region is noisy)
        ctx.set_profiling(1)
        for _ in range(20):
            run_steps(passes_full)
        sec2, cnt2 = ctx.loop_timing()
        passes2 = ctx.loop_timing_passes()
        ctx.set_profiling(0)
        block = {"launches_timed": cnt2, "avg_launch_us": 1e6 * sec2 / max(1, cnt2), "passes_per_launch": passes2 / max(1, cnt2)}
        prim, prim_src = (region, "timed region") if region else (block, "block of 20 registrations after the timed region")
        t_launch = 1e-6 * prim["avg_launch_us"]
        ppl = max(1.0, prim["passes_per_launch"])
        # (2) what the kernel EXECUTES: one full registration with the instrumented instantiation
        ctx.set_work_counting(True)
        ctx.reset_moving()
        ctx.loop_begin(pkg.ICP_POINT_TO_POINT, max_iter=MAX_ITER, tol=TOL, fixed_iterations=False)
        done = False
        while not done:
            _, done = ctx.loop_run(1 << 20)
        work = ctx.get_work_counters()
        ctx.set_work_counting(False)
        blocks = info["blocks"]
        passes_counted = work["block_passes"] / max(1, blocks)
        row64 = info["threads"] == 512          # rows of 64 points: 8 waves per block, one point per lane
        pts_hit, nwaves = (64, 8) if row64 else (128, 16)
        flop_reg, flop_parts = executed_flop(work, pts_hit, nwaves)
        flop_launch = flop_reg * (ppl / max(1.0, passes_counted))      # scaled to the passes an average timed launch ran
        pairs_full = work["hits_full"] * pts_hit * 8
        # (3) the stand-alone matching kernel by the reference's method: min (and mean) of 10 launches after 2 warm-ups
        seeded = ctx.nn_match_bench_launches(10, 2, 0)
        dense = ctx.nn_match_bench_launches(10, 2, 2)
        dinfo = ctx.nn_launch_info_ex(dense=True)
        dense_flop = float(FLOP_PAIR) * dinfo["n_pad"] * dinfo["m_pad"]
        t_dense = 1e-3 * float(dense.mean())
        # HBM traffic: PMC counters need rocprofv3 (separate --pmc passes), so the figure comes from this round's committed
        # profile of the stand-alone kernel; left null when that profile is missing
        traffic, traffic_src = None, None
        pmc = os.path.join(ROOT, "profiles", "r2", "pmc_hbm_traffic_sparse.json")
        if world == 1 and os.path.exists(pmc):
            rec = json.load(open(pmc))
            k = next((v for kk, v in rec.items() if "nn_match_" in kk), None)
            if k:
                traffic = k["hbm_bytes_corrected"]
                traffic_src = ("profiles/r2/pmc_hbm_traffic_sparse.json (%s): ONE stand-alone seeded pass, FETCH_SIZE %.0f B raw (x2: gfx950 "
                               "correction) + WRITE_SIZE %.0f B; a resident launch keeps its chunk boxes in registers, so its later passes "
                               "read less" % (rec.get("_build", "build unknown"), k["fetch_bytes_raw"], k["write_bytes"]))
        out["roofline"] = {
            "kernel": ("nn_match_row64<1> (rows of 64 points, 8 waves per block)" if row64 else "nn_match_sparse<1> (rows of 128 points, 16 waves per block)") +
                      ", resident: ONE launch per REGISTRATION, every block on the machine (%.2f matching passes per "
                      "timed launch); every pass = mailbox message from the host (command, R, t) -> [transform + error of the previous "
                      "pass] -> lane-parallel chunk-box search -> hit processing (packed fp32, exact arithmetic) -> LDS key merge -> "
                      "moment row to the host.  The duration INCLUDES the host round trips between the passes." % ppl,
            "bound": "valu",
            "bound_detail": "fp32 vector roof 157.3 TFLOP/s (FMA-counted; the same figure as fp32 MFMA on gfx950).  The kernel issues packed "
                            "VALU ops only (v_pk_add/mul_f32, no MFMA: (dx*dx + dy*dy) + dz*dz must round every operation separately, and "
                            "the contract forbids FMA, which caps executed arithmetic at half that roof).  Brute-force NN is 4681 "
                            "flop/B on this cloud (ridge ~20): not HBM-bound.",
            "achieved": flop_launch / t_launch / 1e12, "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": flop_launch / t_launch / 1e12 / FP32_PEAK_TFLOPS,
            "traffic": traffic, "traffic_source": traffic_src,
            "what_frac_counts": "EXECUTED fp32 flop (kernel-side tallies of one full registration, icp_get_work_counters) / launch time; the "
                                "kernel returns the brute-force answer bit for bit but proves for ~98 % of the pairs that they cannot win, "
                                "so it is bound by a chain of dependent latencies, not by arithmetic",
            "avg_launch_us": prim["avg_launch_us"], "launches_timed": prim["launches_timed"], "passes_per_launch": ppl,
            "avg_pass_us": prim["avg_launch_us"] / ppl, "timed_in": prim_src,
            "timed_region": region, "post_region_block": block,
            "executed": {"flop_per_launch": flop_launch, "flop_per_pass": flop_reg / max(1.0, passes_counted),
                         "flop_by_part_one_registration": flop_parts, "work_counters_one_registration": work,
                         "passes_counted": passes_counted,
                         "pairs_evaluated_in_full_fraction": pairs_full / max(1.0, passes_counted * n * m)},
            "time_to_solution": {"brute_force_flop_per_pass": flops_pass,
                                 "brute_force_equivalent_TFLOPs": flops_pass * ppl / t_launch / 1e12,
                                 "note": "the arithmetic the answer stands for (8 flop x N x M per pass) over the measured time: a speed-up "
                                         "figure, not a roofline fraction",
                                 "speedup_vs_dense_kernel_per_pass": t_dense / (t_launch / ppl)},
            "matching_only": {"what": "stand-alone seeded launches of the same kernel without its fused front end and tail (no transform, no moment rows), events around "
                                      "every launch, 2 warm-ups: the reference's method (src/CUDA/Matching_opt.cu:213-226)",
                              "min_launch_us": 1e3 * float(seeded.min()), "avg_launch_us": 1e3 * float(seeded.mean()), "launches": 10},
            "dense_kernel": {"what": "nn_match_f32_v2<2,8,0,0>: the LDS-tiled packed kernel that EXECUTES every pair (no boxes, no early-out) on "
                                     "the same resident clouds; executed flop = 8 x n_pad x m_pad",
                             "min_launch_us": 1e3 * float(dense.min()), "avg_launch_us": 1e3 * float(dense.mean()), "launches": 10,
                             "flop_per_launch": dense_flop, "achieved": dense_flop / t_dense / 1e12,
                             "frac": dense_flop / t_dense / 1e12 / FP32_PEAK_TFLOPS, "launch": dinfo},
            "launch": info,
            "hbm": {"algorithmic_bytes_per_pass": alg_bytes_pass, "achieved_GBps": alg_bytes_pass * ppl / t_launch / 1e9,
                    "peak_GBps": HBM_PEAK_GBPS, "frac": alg_bytes_pass * ppl / t_launch / 1e9 / HBM_PEAK_GBPS},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(P, Q)
    # ---- the same K steps with the iteration's collective done by RCCL (library-issued ncclAllReduce) -----------------
    if use_local:
        ranks.barrier()
        ok, why = guarded(lambda: pkg.distributed.attach_native_comm(ctx, ranks.dist) if ranks.dist
                          else ctx.comm_init(ctx.comm_unique_id(), 0, 1), 90)
        all_ok = ranks.min_int(1 if ok else 0) == 1
        rccl = {"route": "icp_comm_init: ONE ncclAllReduce(sum, 32 doubles, in place) per iteration, issued by libicp_mi355x "
                         "right behind the finalize kernel on the loop's stream; one kernel launch per pass (no resident kernel)",
                "ranks": world}
        if all_ok:
            run_steps(min(W, 50))
            dt_r, stats_r, _, _, _ = timed(K, 0)
            rccl.update({"value": world * K / dt_r, "unit": "iterations/s", "us_per_iteration": 1e6 * dt_r / K,
                         "registrations_timed": stats_r["registrations"]})
            ctx.comm_destroy()
        else:
            rccl["error"] = why if not ok else "another rank could not create its communicator"
            if ok:
                ctx.comm_destroy()
        if out is not None:
            out["rccl"] = rccl

    ranks.barrier()
    return out, ctx, ranks


# ---- configs[4]: the synthetic 10 M-point cloud, moving cloud sharded over the ranks ------------------------------------
def run_s5(args, rank, local_rank, world):
    import numpy as np
    import torch
    from __graft_entry__ import load_package
    pkg = load_package()
    ranks = Ranks(rank, world)
    N = args.points
    Wd = int(np.ceil(np.sqrt(N)))
    D = pkg.datasets.synthetic_grid(Wd, np.float32)[:N]
    M = pkg.datasets.make_model_gpu(D, *pkg.datasets.P2P_GPU)
    lo, cnt = pkg.shard_range(N, rank, world)
    P = np.ascontiguousarray(D[lo:lo + cnt])
    del D
    ctx = pkg.Context(local_rank)
    t0 = time.perf_counter()
    ctx.set_model(M)
    ctx.set_moving(P)
    setup_s = time.perf_counter() - t0
    if ranks.dist:
        pkg.distributed.attach_local_comm(ctx, ranks.dist)

    def sync():
        ranks.barrier()
        torch.cuda.synchronize(local_rank)

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
    ctx.set_profiling(1)
    sync()
    t0 = time.perf_counter()
    k = run(args.steps)
    sync()
    dt = ranks.max(time.perf_counter() - t0)
    sec, cnt_l = ctx.loop_timing()
    st = ctx.loop_state()
    out = None
    if rank == 0:
        out = {
            "metric": "ICP iterations/sec, synthetic 10M-point cloud (BASELINE configs[4])",
            "value": args.steps / dt, "unit": "iterations/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic z = x^2 - y^2 grid, W = %d truncated to %d points; model = moved copy" % (Wd, N),
            "config": {"workload": "synthetic 10M-point cloud point-to-point ICP, moving cloud sharded over the ranks (BASELINE configs[4])",
                       "model_points": N, "moving_points_per_gpu": cnt, "global_moving_points": N,
                       "regime": "one registration of `steps` fixed iterations from the initial pose (cold first pass)",
                       "iterations_run": k, "set_up_ms_rank0": 1e3 * setup_s,
                       "collective": "sum of 32 doubles per iteration through shared host memory (icp_comm_init_local)" if world > 1 else "none"},
            "pairs_per_s_algorithmic": float(N) * float(N) * args.steps / dt,
            "matching_kernel": {"launches_timed": cnt_l, "avg_launch_ms": 1e3 * sec / max(1, cnt_l)},
            "rms_error_series_head": [float(e) for e in st["err"][:6]], "final_rms_error": float(st["err"][-1]),
        }
    if ranks.dist:
        ctx.comm_destroy()
    return out, ctx, ranks


def run_bunny(args, rank, local_rank, world):
    """BASELINE configs[1]: Bunny.csv (35 947 points) against its moved copy, point-to-point, the whole cloud on every rank
    (replicas: a cloud of this size does not shard usefully).  A step is one ICP iteration of real registrations from the
    initial pose (tolerance 1e-6, cold first pass and launches included), as in the hall configuration."""
    import numpy as np
    import torch
    from __graft_entry__ import load_package
    pkg = load_package()
    ranks = Ranks(rank, world)
    B = np.fromfile(os.path.join(ROOT, "tests", "golden", "bunny_xyz_f32.bin"), dtype=np.float32).reshape(-1, 3)
    M = pkg.datasets.make_model_gpu(B, *pkg.datasets.BUNNY)
    ctx = pkg.Context(local_rank)
    ctx.set_model(M)
    ctx.set_moving(B)

    def registration():
        ctx.reset_moving()
        ctx.loop_begin(pkg.ICP_POINT_TO_POINT, max_iter=100, tol=1e-6)
        k, done = 0, False
        while not done:
            kk, done = ctx.loop_run(1 << 20)
            k += kk
        return k

    def sync():
        ranks.barrier()
        torch.cuda.synchronize(local_rank)

    first = time.perf_counter()
    k_first = registration()                     # a context's first registration: its first pass has no counts to share the rows by
    first = time.perf_counter() - first
    done_it = 0
    while done_it < args.warmup:
        done_it += registration()
    sync()
    t0 = time.perf_counter()
    steps = regs = 0
    while steps < args.steps:
        steps += registration()
        regs += 1
    sync()
    dt = ranks.max(time.perf_counter() - t0)
    st = ctx.loop_state()
    info = ctx.nn_launch_info()
    out = None
    if rank == 0:
        out = {
            "metric": "ICP iterations/sec, Bunny.csv 35 947-point cloud (BASELINE configs[1])",
            "value": world * steps / dt, "unit": "iterations/s", "n_gpus": world, "steps": steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "tests/golden/bunny_xyz_f32.bin (the reference's Bunny.csv as float32) and its moved copy",
            "config": {"workload": "Bunny.csv point-to-point ICP (BASELINE configs[1]), one replica per rank",
                       "moving_points": int(B.shape[0]), "model_points": int(M.shape[0]), "registrations_timed": regs,
                       "iterations_per_registration": steps / regs, "steps_requested": args.steps,
                       "matching_blocks": info["blocks"], "threads_per_block": info["threads"],
                       "first_registration_of_the_context_us_per_iteration": 1e6 * first / max(1, k_first)},
            "final_rms_error": float(st["err"][-1]),
        }
    return out, ctx, ranks


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--config", choices=("hall", "s5", "bunny"), default="hall")
    ap.add_argument("--points", type=int, default=10_000_000, help="s5: size of the synthetic cloud")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()
    if args.steps is None:
        args.steps = {"hall": 2000, "bunny": 420}.get(args.config, 30)
    if args.warmup is None:
        args.warmup = {"hall": 200, "bunny": 42}.get(args.config, 5)
    if args.gpus < 1 or args.steps < 1 or args.warmup < 0:
        raise SystemExit("--gpus >= 1, --steps >= 1, --warmup >= 0")

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(spawn_ranks(args.gpus))          # (nothing in this process has touched a GPU)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", str(rank)))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} does not match WORLD_SIZE {world}")
    # rehearsal on a one-GPU box: ICP_BENCH_ONE_DEVICE=1 puts every rank on device 0 (RCCL refuses two ranks on one
    # device, so the `rccl` leg reports that); never used by the driver
    if os.environ.get("ICP_BENCH_ONE_DEVICE") == "1":
        local_rank = 0

    # the contract is ONE JSON line on stdout and RCCL prints a version banner there when its first communicator is
    # created: fd 1 points at stderr until the result is printed
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)
    out, ctx, ranks = {"hall": run_hall, "s5": run_s5, "bunny": run_bunny}[args.config](args, rank, local_rank, world)
    sys.stdout.flush()
    os.dup2(saved_stdout, 1)
    os.close(saved_stdout)
    if rank == 0:
        print(json.dumps(out), flush=True)
    ctx.close()
    ranks.close()
    sys.stdout.flush()
    sys.stderr.flush()
    if STUCK:
        os._exit(0)    # a communicator attempt never returned: its helper thread would keep the interpreter from ending


if __name__ == "__main__":
    main()
