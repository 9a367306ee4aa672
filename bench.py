#!/usr/bin/env python3
"""bench.py -- the hot path's measurements on MI355X, one JSON line per run.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config hall|hall_plane|bunny|s5|cpu_f64] [--shard strong|weak]
                    [--s5-shard blocks|contiguous] [--repeats R] [--no-s5] [--no-rccl] [--no-cpu-baseline] [--no-fresh-pair]

N > 1: either launched by `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...` (RANK / LOCAL_RANK /
WORLD_SIZE / MASTER_* from the environment), or plainly as `python bench.py --gpus N`, in which case this process starts
the N ranks itself and relays rank 0's line.

Processes.  The process the launcher starts is a SUPERVISOR: it makes no GPU call, ever.  It runs the measurement (`--leg
main`) in a fresh child process and -- for N > 1 -- the RCCL leg (`--leg rccl`) in a second one, each under a time limit.
A child that does not answer in time is killed (its exact PID), the line is still printed, and the supervisor exits
non-zero: a stuck communicator can neither hold the line back nor end as rc 0.  To profile, put the measuring process
itself behind the profiler: `rocprofv3 --kernel-trace --stats -- python3 bench.py --leg main [--config ...]`.

What `value` is (round 4).  The K-step region -- EXACTLY K steps between barrier + synchronize on both sides, the max over the
ranks -- is repeated R times (R >= 25 where a region is short: the driver's `--steps 20` is 0.2 ms of work), each repeat
between its own barriers; `ms_per_step` is the MEDIAN region / K, `value` = K / median region, the spread beside it
(`ms_per_step_min` / `_max`, `repeats`).  For N > 1 `value` is the iteration rate of what the ranks compute TOGETHER: K / dt.
Ranks that shard one cloud (hall, s5) run ONE registration; ranks that run replicas (hall_plane, bunny, cpu_f64) each run
their own, and the sum over replicas is reported beside it as `aggregate_iterations_per_s`, never as `value`.

Workloads (`--config`; every BASELINE.json config has one; each line carries `roofline` and `cpu_baseline`):
  hall        configs[2], the one BASELINE.json's metric is quoted on (default): point-to-point ICP on the hall LiDAR scan,
              16 384 x 16 384 points, fp32.  A step = one ICP iteration of back-to-back REAL registrations of the pair
              (tol 1e-6, MAX_ITER 100, src/CUDA/GPU_point_to_point_real.cu:18,404-405): each restarts from the pristine
              cloud, pays its launch and its cold first matching pass and stops by the reference's rule.
              N > 1 (`--shard strong`, the default: north_star's split on the metric's own cloud): the 16 384 moving points are
              sharded over the ranks (icp_shard_range), the full model on every rank, one sum per iteration; `scaling` "strong".
              The line carries beside it `weak_shards` (`--shard weak`: a hall-sized shard per rank, i.e. ONE registration of an
              N x 16 384-point cloud against the hall model), `rccl` (the same sharded steps over the library-issued
              ncclAllReduce) and -- every N, unless --no-s5 -- `s5`: the config north_star shards (configs[4]) in brief.
  hall_plane  configs[3]: the same pair, point-to-plane (kNN(4) + PCA normals of the model on the device, 6x6 solve on the
              host; MAX_ITER 50, tol 1e-6 as src/ICP_point_to_plane.cu).  Same regime.
  bunny       configs[1]: Bunny.csv 35 947 points against its moved copy, point-to-point, fp32, same regime; one replica
              per rank.
  s5          configs[4]: synthetic z = x^2 - y^2 grid truncated to --points (10 M) points against its moved copy; the
              MOVING cloud is sharded over the ranks (strong scaling), the model replicated.  A step = one iteration of
              the whole cloud; the timed region is ONE registration of K fixed iterations from the initial pose.
              N > 1 (`--s5-shard blocks`, the default): the cloud is DEALT to the ranks in blocks of 16 384 points along a Hilbert
              curve -- the work per point varies over this pair, contiguous eighths (`--s5-shard contiguous`, icp_shard_range)
              take 16 to 30 ms per registration and the registration waits for the slowest; dealt blocks take 20 to 24.5.
  cpu_f64     configs[0]: src/ICP_CPU.c's own run -- synthetic grid WIDTH x WIDTH (--width, 32 -> 1024 points), fp64,
              tol 1e-5, MAX_ITER 200 -- through the fp64 path (ICP_F64); same regime as hall.
The only data that crosses ranks is the loop's 32-double moment vector, summed once per iteration.  hall: `value` is measured
with the node-local route (icp_comm_init_local: the vector is already in host memory; shared memory, ~1-2 us -- an RCCL
all-reduce of 256 bytes costs more than the 9 us iteration) and the RCCL route is reported beside it (`rccl`).  s5, the config
north_star shards: `value` is measured on the RCCL route north_star names (ONE library-issued ncclAllReduce per iteration,
icp_comm_init; 0.5 % of a 5 ms iteration), the node-local route beside it (`local`); if RCCL refuses (two ranks rehearsed on
one device) the line says so and `value` falls back to the node-local route.

`roofline`: the dominant kernel of the config, its average launch duration by HIP events on the loop's own stream, the
arithmetic it EXECUTED (kernel-side tallies of its instrumented instantiation, icp_get_work_counters) over that time,
against the vector peak of the precision (fp32 157.3, fp64 78.6 TFLOP/s).  `cpu_baseline`: the CPU oracle on this box's
host cores, a bounded sample per BASELINE.md section 3.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

FP32_PEAK_TFLOPS = 157.3   # MI355X fp32 vector (packed FMA) == fp32 MFMA peak (MI355X_MICROARCH.md)
FP64_PEAK_TFLOPS = 78.6    # MI355X fp64 vector (half the fp32 rate)
HBM_PEAK_GBPS = 8000.0     # HBM3E peak (MI355X_MICROARCH.md)
CONFIGS = ("hall", "hall_plane", "bunny", "s5", "cpu_f64")

# operations behind each tally of a sparse kernel's instrumented instantiation (include/icp_mi355x.h,
# icp_get_work_counters; sub/mul/add counted, compare/select/min not -- the convention of SURVEY 8d's 8 flop per pair):
#   box test of one point or one group box against one chunk box: 6 sub + 3 mul + 2 add + 1 mul (safety factor) = 12
#   xy half of one pair: 2 sub + 2 mul + 1 add = 5;  z half: 1 sub + 1 mul + 1 add = 3;  a full pair: 8
#   transform of one point: 9 mul + 9 add = 18 (every one of a block's waves re-derives the block's points)
FLOP_BOX, FLOP_XY, FLOP_Z, FLOP_PAIR, FLOP_RT = 12, 5, 3, 8, 18
PTS_PER_HIT = 128          # one hit = one wave (64 lanes x 2 moving points) against one 8-point model chunk
WAVES_PER_BLOCK = 16


def executed_flop(work, pts_per_hit=PTS_PER_HIT, waves_per_block=WAVES_PER_BLOCK):
    """flop a sparse kernel EXECUTED, from its work counters (the fp64 moment sums of the row tail -- ~45 flop per point
    and pass -- are left out: <2 % of the total).  pts_per_hit: the moving points one wave holds (128 with rows of 128
    points, 64 with rows of 64); waves_per_block: every wave re-derives the block's points"""
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


def cpu_model():
    """the host CPU's model string (both CPU figures are labelled with it)"""
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.lower().startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def usable_cores():
    cores = len(os.sched_getaffinity(0))
    try:   # a container's CPU quota counts, not the CPUs it can see (threads beyond it only queue up)
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            cores = max(1, min(cores, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return cores


# ======================================================================================================================
# workloads
# ======================================================================================================================
class Workload:
    """one BASELINE config: the clouds, how a registration is started, what its numbers mean"""
    name = ""
    metric_id = ""
    dtype = "f32"
    peak = FP32_PEAK_TFLOPS
    scaling = "weak"
    regime = "registrations"     # back-to-back real registrations, cut at K steps | "fixed": one registration of K iterations
    max_iter, tol = 100, 1e-6
    block_regs = 20              # registrations of the kernel-timing block that follows the timed region
    elem = 4

    def __init__(self, pkg, args, rank, local_rank, world):
        self.pkg, self.args, self.rank, self.local_rank, self.world = pkg, args, rank, local_rank, world
        self.metric = pkg.ICP_POINT_TO_POINT
        self.ctx = pkg.Context(local_rank)     # raises when the HIP library / a gfx950 device is missing
        self.setup_s = 0.0
        self.build()

    # -- the loop ---------------------------------------------------------------------------------------------------
    def begin(self, iters=None):
        self.ctx.reset_moving()
        self.ctx.loop_begin(self.metric, max_iter=iters if iters else self.max_iter, tol=self.tol, fixed_iterations=self.regime == "fixed")

    def registration(self, iters=None):
        """one whole registration from the initial pose; returns its iterations"""
        self.begin(iters)
        k, done = 0, False
        while not done:
            kk, done = self.ctx.loop_run(1 << 20)
            k += kk
        return k

    def run_steps(self, count, stats=None):
        """`count` ICP iterations.  registrations: back-to-back REAL registrations, every one from the pristine moving cloud
        (reset inside the timed region), a cold matching pass first, iterating until the reference's stop rule fires; the last
        one is cut when `count` is reached.  fixed: ONE registration of `count` iterations from the initial pose."""
        if self.regime == "fixed":
            k = self.registration(count)
            if stats is not None:
                stats["registrations"] += 1
                stats["iterations"] += k
            return
        left = count
        while left > 0:
            self.begin()
            k, _ = self.ctx.loop_run(left)
            left -= k
            if stats is not None:
                stats["registrations"] += 1
                stats["iterations"] += k

    def sanity(self):
        """one complete registration, untimed (loads the code objects, sizes the work buffers); the bench refuses to print a
        rate for a loop that does not converge"""
        if self.regime == "fixed":
            return self.args.steps
        self.registration()
        st = self.ctx.loop_state()
        passes = int(st["passes"])
        self.final_err = float(st["err"][-1])      # (of a COMPLETE registration: the timed region's last one is cut at K steps)
        self.check_converged(passes, self.final_err)
        return passes

    def check_converged(self, passes, err):
        if not (2 <= passes <= self.max_iter and err < 1e-3):
            raise SystemExit(f"[bench] the {self.name} registration did not converge: {passes} passes, rms error {err}")

    # -- a pair the context has not seen -------------------------------------------------------------------------------
    fresh_reps, fresh_warm = 10, 2
    variant_base = None            # (angles, t) of the config's own model; variant k scales them

    def variant_model(self, k):
        ang, t = self.variant_base
        f, g = 1.0 + 0.06 * (k + 1), 1.0 - 0.05 * (k + 1)
        return self.pkg.datasets.make_model_gpu(self.P, tuple(a * f for a in ang), tuple(x * g for x in t))

    def upload_pair(self, Q):
        self.ctx.set_model(Q)
        self.ctx.set_moving(self.P)

    def fresh_pair(self):
        """What a sensor's NEXT pair costs: icp_set_model (+ normals) + icp_set_moving + one registration of a pair this context has
        not seen (the same moving cloud against a differently moved copy, another one every time), host buffers in, clouds not
        resident.  The reference's method (src/CUDA/Matching_opt.cu:213-226): the minimum of `fresh_reps` after `fresh_warm`."""
        models = [self.variant_model(k) for k in range(3)]
        setup, total, iters = [], [], []
        for r in range(-self.fresh_warm, self.fresh_reps):
            Q = models[(r + self.fresh_warm) % len(models)]
            t0 = time.perf_counter()
            self.upload_pair(Q)
            t1 = time.perf_counter()
            self.ctx.loop_begin(self.metric, max_iter=self.args.steps if self.regime == "fixed" else self.max_iter, tol=self.tol,
                                fixed_iterations=self.regime == "fixed")
            k, done = 0, False
            while not done:
                kk, done = self.ctx.loop_run(1 << 20)
                k += kk
            t2 = time.perf_counter()
            if r >= 0:
                setup.append(t1 - t0); total.append(t2 - t0); iters.append(k)
        import statistics
        return {"fresh_pair_ms": 1e3 * min(total), "setup_ms": 1e3 * min(setup),
                "fresh_pair": {"what": "icp_set_model (+ normals) + icp_set_moving + ONE registration of a pair the context has not seen (host buffers in; "
                                       "the moving cloud against a differently moved copy, another copy every repeat); minimum of %d after %d "
                                       "(the reference's method, src/CUDA/Matching_opt.cu:213-226)" % (self.fresh_reps, self.fresh_warm),
                               "min_ms": 1e3 * min(total), "median_ms": 1e3 * statistics.median(total), "setup_min_ms": 1e3 * min(setup),
                               "setup_median_ms": 1e3 * statistics.median(setup), "iterations": iters,
                               "registration_alone_min_ms": 1e3 * min(t - u for t, u in zip(total, setup))}}

    # -- numbers ----------------------------------------------------------------------------------------------------
    def alg_flop_pass(self):
        return float(FLOP_PAIR) * self.n * self.m            # the brute-force scan's arithmetic

    def alg_bytes_pass(self):
        return (3.0 * self.elem) * (self.n + self.m) + 4.0 * self.n   # read P, read Q, write idx

    def geometry(self, info):
        """(moving points a wave holds = points of a row, waves per block) of the loop's matching kernel"""
        return info["n_pad"] // max(1, info["blocks"] // max(1, info["splits"])), info["threads"] // 64


def _hall_pair(pkg, ctx):
    import numpy as np
    g = os.path.join(ROOT, "tests", "golden")
    ranges = np.fromfile(os.path.join(g, "hall_ranges_u32.bin"), dtype=np.uint32)
    enc = json.load(open(os.path.join(g, "hall_meta.json")))["encoder_count0"]
    alt, az = pkg.datasets.read_os1_intrinsics(os.path.join(g, "beam_intrinsics.csv"))
    return pkg.datasets.hall_clouds(ctx, ranges, enc, alt, az)   # polar -> Cartesian by the device kernel


class Hall(Workload):
    name = "hall"

    def geometry(self, info):
        pts, nw = super().geometry(info)
        return pts, (16 if (self.exclusive and pts == 64 and info["blocks"] <= 256) else nw)
    pmc_kernel = "nn_match_row64<1, false, false"
    metric_id = "ICP iterations/sec + NN HBM GB/s (% roofline), hall cloud"
    data = ("hall LiDAR scan fixture (tests/golden/hall_ranges_u32.bin, decoded from the reference's Donut_1024x16.csv; "
            "polar->Cartesian by the device kernel)")
    workload = "hall LiDAR scan point-to-point ICP (BASELINE configs[2])"
    kernel = "nn_match_row64<1> (rows of 64 points, one per lane; 16 waves per block with the device to itself, else 8; point-to-point row tail)"

    def build(self):
        import numpy as np
        self.P, self.Q = _hall_pair(self.pkg, self.ctx)
        self.shard = getattr(self.args, "shard", "strong") if self.world > 1 else "none"
        if self.shard == "strong":
            # north_star's split on the metric's own cloud: the 16 384 moving points sharded over the ranks, the model replicated
            lo, cnt = self.pkg.shard_range(self.P.shape[0], self.rank, self.world)
            self.n_global = self.P.shape[0]
            self.P = np.ascontiguousarray(self.P[lo:lo + cnt])
            self.scaling = "strong"
        self.n, self.m = self.P.shape[0], self.Q.shape[0]
        if self.shard != "strong":
            self.n_global = self.n * self.world      # (weak: a hall-sized shard per rank = one registration of an N x 16 384-point cloud)
        # one process per GPU and nothing else on it: the bench owns its device and says so (icp_set_exclusive: the rows of 64
        # points run as 16-wave blocks, one to a CU; same bits).  Not in the one-device rehearsal, where two ranks share a GPU.
        self.exclusive = os.environ.get("ICP_BENCH_ONE_DEVICE") != "1" and os.environ.get("ICP_BENCH_EXCLUSIVE", "1") != "0"
        if self.exclusive:
            self.ctx.set_exclusive(True)
        ang, t_mm = self.pkg.datasets.HALL_MM
        self.variant_base = (ang, tuple(x / 1000.0 for x in t_mm))     # (the fixture's pair is built in mm and scaled: the variants in metres)
        self.ctx.set_model(self.Q)
        self.ctx.set_moving(self.P)

    def cpu_baseline(self, budget_s=12.0):
        """the CPU oracle (scalar C restatement of src/ICP_CPU.c's loop in fp32) on this host, bounded sample"""
        import oracle_lib
        orc = oracle_lib.Oracle()
        P, Q = self.P, self.Q
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
               "host_cpus": os.cpu_count(), "cpu_model": cpu_model()}
        # beside it: the same port with its matching loop spread over the cores this process may use (OpenMP over the
        # moving points; the minimisation stays scalar) -- the strongest thing the host can do with the reference's algorithm
        cores = usable_cores()
        if cores > 1:
            orc.set_threads(cores)
            try:
                orc.icp_p2p(P, Q, 2, 0.0, fixed=True)          # (thread pool start-up)
                it2 = 200
                t0 = time.perf_counter()
                orc.icp_p2p(P, Q, it2, 0.0, fixed=True)
                dt2 = time.perf_counter() - t0
                out["all_cores"] = {"value": it2 / dt2, "unit": "iterations/s", "cores": cores,
                                    "sample": f"{it2} fixed iterations, matching loop over {cores} OpenMP threads, {dt2:.1f} s"}
            finally:
                orc.set_threads(1)
        return out


class HallPlane(Hall):
    name = "hall_plane"
    pmc_kernel = "nn_match_row64<2, false"
    metric_id = "ICP iterations/sec, hall cloud, point-to-plane (BASELINE configs[3])"
    workload = "hall LiDAR scan point-to-plane ICP, 6x6 solve on the host (BASELINE configs[3])"
    kernel = "nn_match_row64<2> (rows of 64 points, one per lane; 16 waves per block with the device to itself, else 8; point-to-plane row tail: 21 + 6 sums)"
    max_iter, tol = 50, 1e-6      # src/ICP_point_to_plane.cu

    def upload_pair(self, Q):
        self.ctx.set_model(Q)
        self.ctx.estimate_normals()      # (kNN(4) + PCA of the NEW model on the device: part of what a fresh pair costs)
        self.ctx.set_moving(self.P)

    def build(self):
        super().build()
        self.metric = self.pkg.ICP_POINT_TO_PLANE
        t0 = time.perf_counter()
        self.normals = self.ctx.estimate_normals()     # kNN(4) + PCA on the device; the context keeps them
        self.normals_s = time.perf_counter() - t0

    def cpu_baseline(self, budget_s=12.0):
        """src/CUDA/CPU_ICP_point_to-plane.cpp restated (oracle/icp_oracle.c), fixed iterations on the normals of the device"""
        import oracle_lib
        orc = oracle_lib.Oracle()
        P, Q, Nr = self.P, self.Q, self.normals
        t0 = time.perf_counter()
        orc.icp_p2plane(P, Q, Nr, 1, 0.0, fixed=True)
        one = time.perf_counter() - t0
        iters = max(2, min(40, int(budget_s / max(one, 1e-3))))
        t0 = time.perf_counter()
        orc.icp_p2plane(P, Q, Nr, iters, 0.0, fixed=True)
        dt = time.perf_counter() - t0
        out = {"value": iters / dt, "unit": "iterations/s", "cores": 1, "kind": "port",
               "sample": f"{iters} fixed point-to-plane iterations of the same hall workload (16384x16384, fp32), oracle/icp_oracle.c "
                         f"single thread, {dt:.1f} s; the normals are not in the rate",
               "host_cpus": os.cpu_count(), "cpu_model": cpu_model()}
        t0 = time.perf_counter()
        nbr = orc.knn4(Q)
        orc.normals(Q, nbr)
        out["normals_once_s"] = time.perf_counter() - t0     # (the one-off front end: brute-force kNN(4) + eigenvectors on one core)
        out["normals_once_device_s"] = self.normals_s
        cores = usable_cores()
        if cores > 1:
            orc.set_threads(cores)
            try:
                orc.icp_p2plane(P, Q, Nr, 2, 0.0, fixed=True)
                it2 = 100
                t0 = time.perf_counter()
                orc.icp_p2plane(P, Q, Nr, it2, 0.0, fixed=True)
                dt2 = time.perf_counter() - t0
                out["all_cores"] = {"value": it2 / dt2, "unit": "iterations/s", "cores": cores,
                                    "sample": f"{it2} fixed iterations, matching loop over {cores} OpenMP threads, {dt2:.1f} s"}
            finally:
                orc.set_threads(1)
        return out


class Bunny(Workload):
    name = "bunny"
    pmc_kernel = "nn_match_sparse<1, false, true, false, 8>"
    metric_id = "ICP iterations/sec, Bunny.csv 35 947-point cloud (BASELINE configs[1])"
    data = "tests/golden/bunny_xyz_f32.bin (the reference's Bunny.csv as float32) and its moved copy"
    workload = "Bunny.csv point-to-point ICP (BASELINE configs[1]), one replica per rank"
    kernel = ("nn_match_sparse<1, ..., NWS = 8> (rows of 128 points, 8-wave blocks, shared rows: a grid of 2 x CUs blocks, the spare "
              "ones dealt to the heaviest rows; Morton-ordered views of both clouds; a context's first registration runs one armed "
              "launch per pass, every later one is a single launch whose blocks keep the roles the counts of the registration before give them)")
    block_regs = 6

    def build(self):
        import numpy as np
        B = np.fromfile(os.path.join(ROOT, "tests", "golden", "bunny_xyz_f32.bin"), dtype=np.float32).reshape(-1, 3)
        self.P = B
        self.variant_base = self.pkg.datasets.BUNNY
        self.Q = self.pkg.datasets.make_model_gpu(B, *self.pkg.datasets.BUNNY)
        self.n, self.m = self.P.shape[0], self.Q.shape[0]
        self.n_global = self.n
        self.ctx.set_model(self.Q)
        self.ctx.set_moving(self.P)

    def cpu_baseline(self, budget_s=12.0):
        """Bunny.csv, ONE iteration x 3, the minimum"""
        import oracle_lib
        orc = oracle_lib.Oracle()
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            orc.icp_p2p(self.P, self.Q, 1, 0.0, fixed=True)
            ts.append(time.perf_counter() - t0)
        out = {"value": 1.0 / min(ts), "unit": "iterations/s", "cores": 1, "kind": "port",
               "sample": "ONE point-to-point iteration of the same Bunny.csv workload (35947x35947, fp32) x 3, the minimum "
                         f"({min(ts):.2f} s; oracle/icp_oracle.c single thread, {sum(ts):.1f} s in all)",
               "host_cpus": os.cpu_count(), "cpu_model": cpu_model()}
        cores = usable_cores()
        if cores > 1:
            orc.set_threads(cores)
            try:
                orc.icp_p2p(self.P, self.Q, 1, 0.0, fixed=True)
                it2 = 20
                t0 = time.perf_counter()
                orc.icp_p2p(self.P, self.Q, it2, 0.0, fixed=True)
                dt2 = time.perf_counter() - t0
                out["all_cores"] = {"value": it2 / dt2, "unit": "iterations/s", "cores": cores,
                                    "sample": f"{it2} fixed iterations, matching loop over {cores} OpenMP threads, {dt2:.1f} s"}
            finally:
                orc.set_threads(1)
        return out


class S5(Workload):
    name = "s5"
    pmc_kernel = "nn_match_sparse<1, false, true, true, 4>"     # (the seeded passes; a registration's cold first pass runs the 8-wave form)
    metric_id = "ICP iterations/sec, synthetic 10M-point cloud (BASELINE configs[4])"
    workload = "synthetic 10M-point cloud point-to-point ICP, moving cloud sharded over the ranks (BASELINE configs[4])"
    kernel = ("nn_match_sparse<1, ..., HIER, 4> (rows of 128 points, 4 waves per block and four blocks to a CU -- 8 waves for the cold first "
              "pass of a context without history --, three-level box hierarchy over the Hilbert-ordered view of the model, rows taken "
              "heaviest weight class first -- a class keeps the curve's order, sixteen consecutive rows to an XCD --, the heaviest split over 2..64 blocks, a refinement round over local samples before the chunks are listed, a chunk asked for by one half of a row evaluated for that half), "
              "ONE launch per pass")
    scaling = "strong"
    regime = "fixed"
    fresh_reps, fresh_warm = 2, 1      # (a fresh 10 M-point pair is seconds of set-up: three in all)

    def variant_model(self, k):
        ang, t = self.pkg.datasets.P2P_GPU
        f, g = 1.0 + 0.06 * (k + 1), 1.0 - 0.05 * (k + 1)
        return self.pkg.datasets.make_model_gpu(self.P, tuple(a * f for a in ang), tuple(x * g for x in t))   # (N = 1: P is the whole cloud)

    def build(self):
        import numpy as np
        N = self.args.points
        self.Wd = int(np.ceil(np.sqrt(N)))
        D = self.pkg.datasets.synthetic_grid(self.Wd, np.float32)[:N]
        self.Q = self.pkg.datasets.make_model_gpu(D, *self.pkg.datasets.P2P_GPU)
        if self.world > 1 and getattr(self.args, "s5_shard", "blocks") == "blocks":
            # N > 1: the cloud is dealt to the ranks in blocks of 16 384 points along a Hilbert curve (128 rows of neighbours each), not
            # cut into contiguous eighths -- the work per point varies over this pair, and the eighths take 16 to 30 ms per registration
            # against 20 to 25 for dealt blocks (profiles/r4/r4_14_s5_shares_of_eight.txt); a registration is as slow as its slowest
            # rank.  Any partition of the moving points is the same registration.  (--s5-shard contiguous: icp_shard_range, as before)
            idx = self.pkg.distributed.shard_cyclic_index(N, self.rank, self.world, 16384, self.pkg.distributed.curve_order(D))
            self.P = np.ascontiguousarray(D[idx]); cnt = len(idx)
            self.shard_form = "blocks of 16 384 points along a Hilbert curve, dealt to the ranks (distributed.shard_cyclic_index)"
        else:
            lo, cnt = self.pkg.shard_range(N, self.rank, self.world)
            self.P = np.ascontiguousarray(D[lo:lo + cnt])
            self.shard_form = "contiguous ranges (icp_shard_range)" if self.world > 1 else "one rank: the whole cloud"
        self.slice_D = np.ascontiguousarray(D[: min(N, 100_000)])
        del D
        self.n, self.m, self.n_global = cnt, N, N
        self.data = "synthetic z = x^2 - y^2 grid, W = %d truncated to %d points; model = moved copy" % (self.Wd, N)
        t0 = time.perf_counter()
        self.ctx.set_model(self.Q)
        self.ctx.set_moving(self.P)
        self.setup_s = time.perf_counter() - t0

    def alg_flop_pass(self):
        return float(FLOP_PAIR) * self.n_global * self.m

    def alg_bytes_pass(self):
        return 12.0 * (self.n_global + self.m) + 4.0 * self.n_global

    def cpu_baseline(self, budget_s=12.0):
        """1e14 pairs per iteration are out of a CPU's reach; a 100 k-point slice is timed and the
        iteration time scaled by (N / slice)^2 -- a brute-force scan costs the same per pair whatever the geometry"""
        import oracle_lib
        orc = oracle_lib.Oracle()
        D = self.slice_D
        s = D.shape[0]
        M = self.Q[:s]
        t0 = time.perf_counter()
        orc.icp_p2p(D, M, 1, 0.0, fixed=True)
        one = time.perf_counter() - t0
        scale = (float(self.n_global) / s) * (float(self.m) / s)
        out = {"value": 1.0 / (one * scale), "unit": "iterations/s", "cores": 1, "kind": "port", "extrapolated": True,
               "sample": f"EXTRAPOLATED: one point-to-point iteration of a {s} x {s}-point slice of the same clouds ({one:.2f} s, "
                         f"oracle/icp_oracle.c single thread) x (N / slice)^2 = {scale:.0f} -> {one * scale:.0f} s per iteration of the "
                         f"{self.n_global} x {self.m} workload",
               "slice_points": s, "slice_iteration_s": one, "host_cpus": os.cpu_count(), "cpu_model": cpu_model()}
        cores = usable_cores()
        if cores > 1:
            orc.set_threads(cores)
            try:
                orc.icp_p2p(D, M, 1, 0.0, fixed=True)
                t0 = time.perf_counter()
                orc.icp_p2p(D, M, 3, 0.0, fixed=True)
                dt2 = (time.perf_counter() - t0) / 3
                out["all_cores"] = {"value": 1.0 / (dt2 * scale), "unit": "iterations/s", "cores": cores, "extrapolated": True,
                                    "sample": f"EXTRAPOLATED the same way from {dt2:.2f} s per slice iteration with the matching loop over {cores} OpenMP threads"}
            finally:
                orc.set_threads(1)
        return out


class CpuF64(Workload):
    name = "cpu_f64"
    pmc_kernel = "nn_match_row64_f64<1, 16, false>"
    metric_id = "ICP iterations/sec, synthetic z=x^2-y^2 cloud, fp64 (src/ICP_CPU.c's run; BASELINE configs[0])"
    dtype = "f64"
    peak = FP64_PEAK_TFLOPS
    elem = 8
    max_iter, tol = 200, 1e-5     # src/ICP_CPU.c:267-269
    kernel = ("nn_match_row64_f64<1, NW> (rows of 64 points, one per lane, 16 waves per block while every row has its own CU, chunk boxes "
              "+ exact scalar (dx*dx + dy*dy) + dz*dz in double), resident: ONE launch per registration")

    def build(self):
        import numpy as np
        W = self.args.width
        self.P = self.pkg.datasets.synthetic_grid(W, np.float64)
        self.Q = self.pkg.datasets.make_model_cpu(self.P)
        self.n, self.m = self.P.shape[0], self.Q.shape[0]
        self.n_global = self.n
        self.data = f"synthetic z = x^2 - y^2 grid, WIDTH {W} ({W * W} points), model by src/ICP_CPU.c:100-149; float64"
        self.workload = f"synthetic z=x^2-y^2 cloud ({W * W} points) point-to-point ICP in fp64, the run of src/ICP_CPU.c (BASELINE configs[0])"
        self.ctx.set_model(self.Q)
        self.ctx.set_moving(self.P)

    def variant_model(self, k):
        ang, t = self.pkg.datasets.CPU_F64
        f, g = 1.0 + 0.03 * (k + 1), 1.0 - 0.05 * (k + 1)
        return self.pkg.datasets.make_model_cpu(self.P, tuple(a * f for a in ang), tuple(x * g for x in t))

    def geometry(self, info):
        return 64, (16 if info["blocks"] <= 256 else 8)    # (launch_row64_f64: 16 waves while every row has a CU of its own)

    def check_converged(self, passes, err):
        # (this pair is a radian apart and ends at E = 0.83 by the |dE| rule: src/ICP_CPU.c prints the same)
        if not (2 <= passes <= self.max_iter):
            raise SystemExit(f"[bench] the fp64 registration did not stop by its rule: {passes} passes")

    def cpu_baseline(self, budget_s=10.0):
        """the fp64 oracle (oracle/icp_oracle.c, orc_icp_p2p_f64 = src/ICP_CPU.c:217-271) to convergence on the same pair"""
        import oracle_lib
        orc = oracle_lib.Oracle()
        t0 = time.perf_counter()
        r = orc.icp_p2p(self.P, self.Q, self.max_iter, self.tol)
        one = time.perf_counter() - t0
        reps = max(1, min(400, int(budget_s / max(one, 1e-4))))
        t0 = time.perf_counter()
        its = 0
        for _ in range(reps):
            its += orc.icp_p2p(self.P, self.Q, self.max_iter, self.tol)["passes"]
        dt = time.perf_counter() - t0
        out = {"value": its / dt, "unit": "iterations/s", "cores": 1, "kind": "port",
               "sample": f"{reps} whole registrations of the same pair to convergence ({r['passes']} passes each, tol 1e-5), fp64 oracle single thread, {dt:.1f} s",
               "iterations_per_registration": r["passes"], "final_rms_error": float(r["err"][-1]), "host_cpus": os.cpu_count(), "cpu_model": cpu_model()}
        cores = usable_cores()
        if cores > 1 and self.n >= 4096:
            orc.set_threads(cores)
            try:
                orc.icp_p2p(self.P, self.Q, self.max_iter, self.tol)
                t0 = time.perf_counter()
                its2 = sum(orc.icp_p2p(self.P, self.Q, self.max_iter, self.tol)["passes"] for _ in range(3))
                dt2 = time.perf_counter() - t0
                out["all_cores"] = {"value": its2 / dt2, "unit": "iterations/s", "cores": cores,
                                    "sample": f"3 registrations, matching loop over {cores} OpenMP threads, {dt2:.1f} s"}
            finally:
                orc.set_threads(1)
        return out


WORKLOADS = {"hall": Hall, "hall_plane": HallPlane, "bunny": Bunny, "s5": S5, "cpu_f64": CpuF64}
DEFAULT_STEPS = {"hall": (2000, 200), "hall_plane": (1000, 100), "bunny": (420, 42), "s5": (30, 5), "cpu_f64": (2000, 200)}


# ======================================================================================================================
# the control plane of a multi-rank run
# ======================================================================================================================
class Ranks:
    """torch.distributed over gloo (barriers, the max over ranks, the 128-byte communicator ids).  The data path never
    goes through it."""

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

    def max_int(self, v):
        if not self.dist:
            return v
        import torch
        t = torch.tensor([v], dtype=torch.int64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return int(t.item())

    def close(self):
        if self.dist:
            self.dist.barrier()
            self.dist.destroy_process_group()


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


# ======================================================================================================================
# --leg main: the measurement
# ======================================================================================================================
def timed_steps(wl, ranks, K, stride, repeats=None):
    """The K-step region -- EXACTLY K steps between barrier + synchronize on both sides, the max over the ranks -- R times, every
    repeat between its own barriers.  R: `--repeats`, else at least 25 where a region is short (the driver's `--steps 20` is 0.2 ms
    of work: ONE such sample says nothing, and the boxes of the pool differ by up to 25 %), as many as make a quarter of a second
    (at most 400), and 3..7 for regions that are long by themselves (s5: one registration of 30 iterations, 0.15 s).
    Returns the R durations (every rank holds the same list) and the loop's event timing summed over all repeats."""
    import torch
    ctx = wl.ctx

    def sync():
        ranks.barrier()
        torch.cuda.synchronize(wl.local_rank)

    def region(stats):
        sync()
        t0 = time.perf_counter()
        wl.run_steps(K, stats)
        sync()
        return ranks.max(time.perf_counter() - t0)

    stats = {"registrations": 0, "iterations": 0}
    ctx.set_profiling(stride)     # (also restarts the stride: the first launch after this is a timed one)
    dts = [region(stats)]
    if repeats is None or repeats < 1:
        d0 = dts[0]
        if d0 < 0.05:
            repeats = max(25, min(400, int(0.25 / max(d0, 1e-6)) + 1))
        else:
            repeats = max(3, min(25, int(1.0 / d0) + 1))
        if os.environ.get("ICP_BENCH_REPEATS"):
            repeats = max(1, int(os.environ["ICP_BENCH_REPEATS"]))
    for _ in range(repeats - 1):
        dts.append(region(stats))
    sec, cnt = ctx.loop_timing()
    passes = ctx.loop_timing_passes()
    ctx.set_profiling(0)
    return dts, stats, sec, cnt, passes


def region_stats(dts, K):
    import statistics
    med = statistics.median(dts)
    return {"value": K / med, "ms_per_step": 1e3 * med / K, "ms_per_step_min": 1e3 * min(dts) / K, "ms_per_step_max": 1e3 * max(dts) / K,
            "repeats": len(dts), "region_ms_median": 1e3 * med, "region_ms_min": 1e3 * min(dts), "region_ms_max": 1e3 * max(dts)}


def roofline_leg(wl, K, passes_full, region, stride):
    """rank 0, alone: kernel durations by HIP events, executed arithmetic by the kernel's own tallies"""
    pkg, ctx = wl.pkg, wl.ctx
    info = ctx.nn_launch_info()
    n, m = wl.n, wl.m
    # (1) the kernel's launches of a fixed block of registrations right after the timed region (K-independent); where the
    #     timed region itself was bracketed by events (every launch of s5, every 7th registration of a long hall run) that
    #     figure comes first
    block = None
    if wl.regime != "fixed":
        ctx.set_profiling(1)
        for _ in range(wl.block_regs):
            wl.registration()
        sec2, cnt2 = ctx.loop_timing()
        passes2 = ctx.loop_timing_passes()
        ctx.set_profiling(0)
        block = {"launches_timed": cnt2, "avg_launch_us": 1e6 * sec2 / max(1, cnt2), "passes_per_launch": passes2 / max(1, cnt2),
                 "registrations": wl.block_regs}
    prim, prim_src = (region, "timed region") if region else (block, f"block of {wl.block_regs} registrations after the timed region")
    t_launch = 1e-6 * prim["avg_launch_us"]
    ppl = max(1.0, prim["passes_per_launch"])
    per_pass = ppl < 1.5      # one launch per matching pass (s5); the others: ONE resident launch per registration
    # (2) what the kernel EXECUTES: the same registration once more with the instrumented instantiation (every launch timed,
    #     so that the loop's own counters say how many launches and matching passes it was)
    ctx.set_work_counting(True)
    ctx.set_profiling(1)
    wl.registration(K if wl.regime == "fixed" else None)
    launches_reg = ctx.loop_timing()[1]
    passes_counted = float(max(1, ctx.loop_timing_passes()))
    ctx.set_profiling(0)
    work = ctx.get_work_counters()
    ctx.set_work_counting(False)
    pts_hit, nwaves = wl.geometry(info)
    flop_reg, flop_parts = executed_flop(work, pts_hit, nwaves)
    flop_pass = flop_reg / max(1.0, passes_counted)
    flop_launch = flop_pass * ppl
    pairs_full = work["hits_full"] * pts_hit * 8
    alg_flop, alg_bytes = wl.alg_flop_pass(), wl.alg_bytes_pass()
    share = float(wl.n) / float(wl.n_global)      # (s5, N > 1: rank 0 holds this share of the moving cloud)
    achieved = flop_launch / t_launch / 1e12
    roof = {
        "kernel": wl.kernel + (", resident: ONE launch per REGISTRATION, every block on the machine (%.2f matching passes per timed launch); "
                               "the duration INCLUDES the host round trips between the passes" % ppl if not per_pass
                               else "; one launch per matching pass (plain launches while they are timed)"),
        "bound": "valu",
        "bound_detail": ("%s vector roof %.1f TFLOP/s (FMA-counted).  The kernels issue %s only -- (dx*dx + dy*dy) + dz*dz must round "
                         "every operation separately, the contract forbids FMA (which caps executed arithmetic at half that roof) and an MFMA "
                         "cannot evaluate a per-pair subtraction.  Brute-force NN is %.0f flop/B on this cloud (ridge ~20): not HBM-bound."
                         % (wl.dtype, wl.peak, "packed VALU ops (v_pk_add/mul_f32)" if wl.dtype == "f32" else "scalar fp64 VALU ops",
                            alg_flop / alg_bytes)),
        "achieved": achieved, "peak": wl.peak, "unit": "TFLOP/s", "frac": achieved / wl.peak,
        "traffic": None, "traffic_from_committed_profile": None,
        "what_frac_counts": "EXECUTED flop (kernel-side tallies of one whole registration, icp_get_work_counters) per matching pass x passes per "
                            "timed launch / average launch time; the kernel returns the brute-force answer bit for bit but proves for most pairs "
                            "that they cannot win",
        "avg_launch_us": prim["avg_launch_us"], "launches_timed": prim["launches_timed"], "passes_per_launch": ppl,
        "avg_pass_us": prim["avg_launch_us"] / ppl, "timed_in": prim_src, "timing_stride": stride,
        "timed_region": region, "post_region_block": block,
        "executed": {"flop_per_launch": flop_launch, "flop_per_pass": flop_pass,
                     "flop_by_part_one_registration": flop_parts, "work_counters_one_registration": work,
                     "passes_counted": passes_counted, "launches_of_that_registration": launches_reg,
                     "moving_points_per_wave": pts_hit, "waves_per_block": nwaves,
                     "pairs_evaluated_in_full_fraction": pairs_full / max(1.0, passes_counted * n * m)},
        "time_to_solution": {"brute_force_flop_per_pass": alg_flop * share,
                             "brute_force_equivalent_TFLOPs": alg_flop * share * ppl / t_launch / 1e12,
                             "note": "the arithmetic the answer stands for (8 flop x N x M per pass) over the measured time: a speed-up "
                                     "figure, not a roofline fraction"},
        "launch": info,
        "hbm": {"algorithmic_bytes_per_pass": alg_bytes * share, "achieved_GBps": alg_bytes * share * ppl / t_launch / 1e9,
                "peak_GBps": HBM_PEAK_GBPS, "frac": alg_bytes * share * ppl / t_launch / 1e9 / HBM_PEAK_GBPS},
    }
    # HBM traffic: PMC counters need rocprofv3 (separate --pmc passes), so the figure comes from this round's committed
    # profile of the config's kernel; left null when that profile is missing
    # (NOT measured in this run: the key says so.  The newest round's record of this config wins.)
    for cand in (os.path.join(ROOT, "profiles", "r4", f"pmc_hbm_traffic_{wl.name}.json"), os.path.join(ROOT, "profiles", "r3", f"pmc_hbm_traffic_{wl.name}.json")):
        if wl.world == 1 and os.path.exists(cand) and roof["traffic"] is None:
            rec = json.load(open(cand))
            kk, k = next(((kk, v) for kk, v in rec.items() if wl.pmc_kernel in kk and isinstance(v, dict)), (None, None))
            if k:
                roof["traffic"] = k["hbm_bytes_corrected"]
                roof["traffic_from_committed_profile"] = ("%s (build %s): kernel %s, %s; FETCH_SIZE %.0f B raw (x2: gfx950 correction) + WRITE_SIZE %.0f B per launch, "
                                          "%d launches; against %.0f algorithmic bytes per launch (%.2f passes)"
                                          % (os.path.relpath(cand, ROOT), rec.get("_build", "unknown"), kk.replace("void icp::", ""), rec.get("_what", "one launch"),
                                             k["fetch_bytes_raw"], k["write_bytes"], k["launches"], alg_bytes * share * ppl, ppl))
    if wl.dtype == "f32" and wl.name in ("hall", "hall_plane", "bunny"):
        # the stand-alone matching kernel by the reference's method: min (and mean) of 10 launches after 2 warm-ups
        seeded = ctx.nn_match_bench_launches(10, 2, 0)
        roof["matching_only"] = {"what": "stand-alone seeded launches of the same search without its fused front end and tail (no transform, no moment "
                                         "rows), events around every launch, 2 warm-ups: the reference's method (src/CUDA/Matching_opt.cu:213-226)",
                                 "min_launch_us": 1e3 * float(seeded.min()), "avg_launch_us": 1e3 * float(seeded.mean()), "launches": 10}
        dense = ctx.nn_match_bench_launches(10, 2, 2)
        dinfo = ctx.nn_launch_info_ex(dense=True)
        dense_flop = float(FLOP_PAIR) * dinfo["n_pad"] * dinfo["m_pad"]
        t_dense = 1e-3 * float(dense.mean())
        roof["dense_kernel"] = {"what": "the LDS-tiled packed kernel that EXECUTES every pair (no boxes, no early-out) on the same resident clouds; "
                                        "executed flop = 8 x n_pad x m_pad",
                                "min_launch_us": 1e3 * float(dense.min()), "avg_launch_us": 1e3 * float(dense.mean()), "launches": 10,
                                "flop_per_launch": dense_flop, "achieved": dense_flop / t_dense / 1e12,
                                "frac": dense_flop / t_dense / 1e12 / FP32_PEAK_TFLOPS, "launch": dinfo}
        roof["time_to_solution"]["speedup_vs_dense_kernel_per_pass"] = t_dense / (t_launch / ppl)
    return roof


def attach_route(pkg, ctx, ranks, route):
    """the iteration's one sum over the ranks: 'local' = shared host memory (icp_comm_init_local), 'rccl' = the library-issued
    ncclAllReduce (icp_comm_init).  Returns (route in force, reason it is not the one asked for or "")."""
    use_dist = ranks.dist is not None or os.environ.get("ICP_BENCH_FORCE_DIST") == "1"
    if not use_dist or route == "none":
        return "none", ""
    if route == "rccl":
        ok, why = True, ""
        try:
            if ranks.dist:
                pkg.distributed.attach_native_comm(ctx, ranks.dist)
            else:
                ctx.comm_init(ctx.comm_unique_id(), 0, 1)
        except Exception as e:  # noqa: BLE001
            ok, why = False, f"{type(e).__name__}: {e}"
        if ranks.min_int(1 if ok else 0) == 1:
            return "rccl", ""
        if ok:
            ctx.comm_destroy()
        why = why if not ok else "another rank could not create its communicator"
        if not ranks.dist:
            return "none", why
        pkg.distributed.attach_local_comm(ctx, ranks.dist)
        return "local", why
    if ranks.dist:
        pkg.distributed.attach_local_comm(ctx, ranks.dist)
        return "local", ""
    return "none", ""


ROUTE_TEXT = {
    "local": ("sum of the loop's moment vector (32 doubles, the 19 / 28 the metric uses travel) per iteration over the node's ranks through "
              "shared host memory (icp_comm_init_local), rank order; every rank keeps its resident kernel"),
    "rccl": ("ONE ncclAllReduce(sum, 32 doubles, in place) per iteration, issued by libicp_mi355x right behind the finalize kernel on the "
             "loop's stream (icp_comm_init); one kernel launch per pass"),
    "none": "none",
}


def leg_main(args, rank, local_rank, world):
    """the measurement.  --brief: an auxiliary leg (another shard mode, another route, another config beside the line's own):
    the timed regions only -- no roofline, no CPU baseline, no fresh pair."""
    import torch  # noqa: F401  (device synchronisation in the timed region)
    from __graft_entry__ import load_package
    pkg = load_package()
    ranks = Ranks(rank, world)
    K, W = args.steps, args.warmup
    wl = WORKLOADS[args.config](pkg, args, rank, local_rank, world)
    ctx = wl.ctx
    brief = args.brief
    sharded = args.config in ("hall", "s5")            # the configs whose ranks work on ONE cloud; the others run replicas
    want = args.route or ("rccl" if (args.config == "s5" and world > 1) else "local")
    route, route_note = attach_route(pkg, ctx, ranks, want if sharded else "none")

    passes_full = wl.sanity()
    first_us = None
    if wl.name == "bunny" and not brief:    # (a context's very first registration ran above: what it cost per iteration is a figure of its own)
        t0 = time.perf_counter()
        k2 = wl.registration()
        first_us = 1e6 * (time.perf_counter() - t0) / max(1, k2)
    warm = None
    if W > 0:
        if wl.regime == "fixed":
            ctx.set_profiling(1)
        wl.run_steps(W)
        if wl.regime == "fixed":
            warm = ctx.loop_timing()
            ctx.set_profiling(0)
    # HIP events around the loop's kernel inside the timed regions: s5 -- every launch is a pass of milliseconds; a resident
    # kernel is a whole registration, so every 7th is timed when a region holds many of them (1-2 % of overhead) and a
    # region of a few registrations is not bracketed at all (the two event records and the wait for the kernel's end would
    # be a tenth of what is being measured)
    regs_expected = max(1, K // max(1, passes_full))
    if wl.regime == "fixed":
        stride = 0 if brief else 1
    else:
        stride = 7 if (regs_expected >= 70 and not brief) else 0
    dts, stats, sec1, cnt1, passes1 = timed_steps(wl, ranks, K, stride, args.repeats)
    st = ctx.loop_state()
    out = None
    if rank == 0:
        rs = region_stats(dts, K)
        replicas = world if not sharded else 1
        out = {
            "metric": wl.metric_id, "value": rs["value"], "unit": "iterations/s",
            "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": rs["ms_per_step"],
            "ms_per_step_min": rs["ms_per_step_min"], "ms_per_step_max": rs["ms_per_step_max"], "repeats": rs["repeats"],
            "higher_is_better": True, "scaling": wl.scaling, "vs_baseline": None, "dtype": wl.dtype, "data": wl.data,
            "config": {"workload": wl.workload, "moving_points_per_gpu": wl.n, "model_points": wl.m,
                       "global_moving_points": wl.n_global,
                       **({"shards": wl.shard_form} if getattr(wl, "shard_form", None) else {}),
                       "ranks_compute": ("ONE registration: the moving cloud is sharded over the ranks (%s), the model replicated" %
                                         ("--shard strong: the config's own cloud split" if wl.scaling == "strong" else
                                          "--shard weak: a whole config-sized shard per rank")) if (sharded and world > 1) else
                                        ("%d independent replicas of the registration, nothing crosses ranks" % world if world > 1 else "one registration"),
                       "what_value_is": "K / (median over `repeats` regions of: max over ranks of the time between barrier + synchronize on both sides "
                                        "of EXACTLY K steps); an iteration rate for any N, never multiplied by the number of ranks",
                       "regime": ("one registration of `steps` fixed iterations from the initial pose (cold first pass)" if wl.regime == "fixed" else
                                  "back-to-back full registrations from the initial pose (cold first pass, tol %g, stop rule on); "
                                  "a step = one iteration of such a registration" % wl.tol),
                       "registrations_timed": stats["registrations"],
                       "registrations_per_region": stats["registrations"] / rs["repeats"],
                       "iterations_per_registration": stats["iterations"] / max(1, stats["registrations"]),
                       "passes_of_a_full_registration": passes_full,
                       "collective_route": route, "collective": ROUTE_TEXT[route]},
            "final_rms_error": float(st["err"][-1]) if wl.regime == "fixed" else wl.final_err,
        }
        if route_note:
            out["config"]["collective_route_note"] = "asked for the %s route: %s" % (want, route_note)
        if replicas > 1:
            out["aggregate_iterations_per_s"] = replicas * rs["value"]      # (the replicas' rates added up: whole-job throughput, not an iteration rate)
        if sharded and world > 1:
            out["aggregate_moving_point_iterations_per_s"] = float(wl.n_global) * rs["value"]
        if wl.name == "s5":
            out["config"]["set_up_ms_rank0"] = 1e3 * wl.setup_s
            out["pairs_per_s_algorithmic"] = float(wl.n_global) * float(wl.m) * rs["value"]
            out["rms_error_series_head"] = [float(e) for e in st["err"][:6]]
        if first_us is not None:
            out["config"]["second_registration_of_the_context_us_per_iteration"] = first_us
        if wl.name == "hall_plane":
            out["config"]["normals_on_device_ms"] = 1e3 * wl.normals_s
        if getattr(wl, "exclusive", False):
            out["config"]["exclusive_device"] = "icp_set_exclusive(ctx, 1): this process owns its GPU (16-wave blocks, one row of 64 points per CU)"
        if brief:
            out = {k: out[k] for k in ("value", "unit", "ms_per_step", "ms_per_step_min", "ms_per_step_max", "repeats", "scaling", "final_rms_error")
                   if k in out} | {"ranks": world, "us_per_iteration": 1e3 * rs["ms_per_step"], "route": route, "registrations_timed": stats["registrations"],
                                   "moving_points_per_gpu": wl.n, "global_moving_points": wl.n_global, "model_points": wl.m, "steps": K,
                                   **({"route_note": out["config"]["collective_route_note"]} if route_note else {}),
                                   **({"aggregate_moving_point_iterations_per_s": out["aggregate_moving_point_iterations_per_s"]} if "aggregate_moving_point_iterations_per_s" in out else {}),
                                   **({"set_up_ms_rank0": 1e3 * wl.setup_s, "rms_error_series_head": out["rms_error_series_head"]} if wl.name == "s5" else {})}
    if route != "none":
        ctx.comm_destroy()     # rank 0 measures its roofline leg alone: no communicator may wait for the others
    if rank == 0 and not brief:
        region = None
        if cnt1 > 0:
            region = {"launches_timed": cnt1, "avg_launch_us": 1e6 * sec1 / cnt1, "passes_per_launch": passes1 / cnt1}
        out["roofline"] = roofline_leg(wl, K, passes_full, region, stride)
        if warm is not None and region is not None and warm[1] > 0:
            # (the figure `rocprofv3 --kernel-trace --stats` of this process shows for the kernel: its warm-up launches are the
            # heavy first passes of a registration, once more)
            out["roofline"]["all_launches_incl_warmup"] = {"launches": warm[1] + cnt1, "avg_launch_us": 1e6 * (warm[0] + sec1) / (warm[1] + cnt1)}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = wl.cpu_baseline()
        if world == 1 and not args.no_fresh_pair:
            out.update(wl.fresh_pair())
    # the auxiliary legs run in processes of their own (the supervisor starts them): agree on their ports here
    ports = []
    if not brief:
        for _ in range(4):
            ports.append(ranks.max_int(free_port() if rank == 0 else 0))
    ranks.barrier()
    ctx.close()
    ranks.close()
    return out, {"ports": ports}


# ======================================================================================================================
# the supervisor: no GPU call in this process
# ======================================================================================================================
def spawn_ranks(n):
    """`python bench.py --gpus N` with no WORLD_SIZE: N supervisors, one rank each; rank 0's line is relayed"""
    port = free_port()
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


def run_leg(leg, argv, env, timeout, extra=()):
    """one leg in a fresh child process; (rc or None when it had to be killed, its stdout, the side record it left).
    `extra`: options of an auxiliary leg (they follow the caller's, so they win)"""
    fd, side = tempfile.mkstemp(prefix="icp_bench_", suffix=".json")
    os.close(fd)
    cmd = [sys.executable, os.path.abspath(__file__)] + argv + list(extra) + ["--leg", leg, "--side-file", side]
    p = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE)
    try:
        out, _ = p.communicate(timeout=timeout)
        rc = p.returncode
    except subprocess.TimeoutExpired:
        p.kill()                      # (the exact process this supervisor started)
        out, _ = p.communicate()
        rc = None
    rec = {}
    try:
        txt = open(side).read()
        rec = json.loads(txt) if txt.strip() else {}
    except (OSError, ValueError):
        pass
    try:
        os.unlink(side)
    except OSError:
        pass
    return rc, out.decode(errors="replace"), rec


def last_json_line(text):
    for ln in reversed(text.splitlines()):
        ln = ln.strip()
        if ln.startswith("{"):
            try:
                return json.loads(ln)
            except ValueError:
                continue
    return None


def supervise(args, argv):
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return spawn_ranks(args.gpus)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} does not match WORLD_SIZE {world}")
    env = dict(os.environ)
    t_main = float(os.environ.get("ICP_BENCH_TIMEOUT", "2400"))
    t_rccl = float(os.environ.get("ICP_BENCH_RCCL_TIMEOUT", "240"))
    t_aux = float(os.environ.get("ICP_BENCH_AUX_TIMEOUT", "600"))
    rc, text, side = run_leg("main", argv, env, t_main)
    line = last_json_line(text) if rank == 0 else None
    if rc is None:
        sys.stderr.write(f"[bench] rank {rank}: the measuring process gave no answer within {t_main:.0f} s and was killed\n")
        return 3
    if rc != 0:
        return rc
    exit_code = 0
    ports = [int(x) for x in (side.get("ports") or [])]
    dist = world > 1 or os.environ.get("ICP_BENCH_FORCE_DIST") == "1"
    # the auxiliary legs: (key of the line, options, time limit, does a failure fail the run?)
    aux = []
    if dist and args.config == "hall" and not args.no_rccl:
        aux.append(("rccl", ["--brief", "--route", "rccl"], t_rccl, True))
    if dist and args.config == "s5" and not args.no_rccl:
        # (the line's own value is on the RCCL route; world == 1 under ICP_BENCH_FORCE_DIST: the line ran without a communicator)
        aux.append(("local" if world > 1 else "rccl", ["--brief", "--route", "local" if world > 1 else "rccl"], t_rccl, True))
    if world > 1 and args.config == "hall" and args.shard == "strong":
        aux.append(("weak_shards", ["--brief", "--shard", "weak", "--route", "local"], t_aux, False))
    if args.config == "hall" and not args.no_s5 and not os.environ.get("ICP_BENCH_NO_S5"):
        aux.append(("s5", ["--brief", "--config", "s5", "--steps", "30", "--warmup", "5", "--repeats", "3"] + (["--route", "rccl"] if world > 1 else []), t_aux, False))
    for n_aux, (key, extra, limit, fatal) in enumerate(aux):
        if n_aux >= len(ports) or not ports[n_aux]:
            break
        # (a rendezvous of its own on the agreed port: under torch.distributed.run the ranks would otherwise look for the
        # launcher's store there, which lives on the launcher's port)
        env2 = dict(env, MASTER_PORT=str(ports[n_aux]), MASTER_ADDR=env.get("MASTER_ADDR", "127.0.0.1"), TORCHELASTIC_USE_AGENT_STORE="False")
        rc2, text2, _ = run_leg("main", argv, env2, limit, extra)
        rec = last_json_line(text2) if rank == 0 else None
        if rc2 is None:
            rec = {"ranks": world, "error": f"no answer within {limit:.0f} s: the process was killed (a stuck communicator?)", "hung": True}
            exit_code = 3
        elif rc2 != 0:
            rec = {"ranks": world, "error": f"the leg ended with exit code {rc2}"}
            exit_code = 4 if fatal else exit_code
        elif rank == 0 and rec is None:
            rec = {"ranks": world, "error": "the leg printed no record"}
            exit_code = 4 if fatal else exit_code
        elif rank == 0 and key == "rccl" and rec.get("route") != "rccl":
            # (a communicator that REFUSES -- RCCL does for two ranks rehearsed on one device -- says so in the record: the line's
            # `value` does not depend on that leg, and nothing is left hanging, so the run itself has not failed)
            rec = {"ranks": world, "error": rec.get("route_note", "the RCCL communicator could not be created")}
        if line is not None:
            line[key] = rec
    if rank == 0:
        if line is None:
            sys.stderr.write("[bench] the measuring process printed no line\n")
            return 5
        print(json.dumps(line), flush=True)
    return exit_code


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--config", choices=CONFIGS, default="hall")
    ap.add_argument("--points", type=int, default=10_000_000, help="s5: size of the synthetic cloud")
    ap.add_argument("--width", type=int, default=32, help="cpu_f64: WIDTH of the synthetic grid (src/ICP_CPU.c ships 100)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-rccl", action="store_true", help="N > 1: skip the leg that measures the other route")
    ap.add_argument("--shard", choices=("strong", "weak"), default="strong",
                    help="hall, N > 1: strong = the 16 384 moving points split over the ranks (north_star); weak = a hall-sized shard per rank")
    ap.add_argument("--s5-shard", dest="s5_shard", choices=("blocks", "contiguous"), default="blocks",
                    help="s5, N > 1: blocks = the cloud dealt to the ranks in blocks of 16 384 points along a Hilbert curve (even loads); "
                         "contiguous = icp_shard_range's ranges (round 3: the eighths take 16 to 30 ms per registration)")
    ap.add_argument("--repeats", type=int, default=None, help="repeats of the K-step region (default: >= 25 for short regions, see timed_steps)")
    ap.add_argument("--no-s5", action="store_true", help="hall: skip the brief configs[4] leg beside the line")
    ap.add_argument("--no-fresh-pair", action="store_true")
    ap.add_argument("--route", choices=("local", "rccl", "none"), default=None, help="(internal) the iteration's sum over the ranks")
    ap.add_argument("--brief", action="store_true", help="(internal) an auxiliary leg: timed regions only")
    ap.add_argument("--leg", choices=("main",), default=None, help="(internal) run the measurement in this process")
    ap.add_argument("--side-file", default=None, help="(internal)")
    args = ap.parse_args()
    if args.steps is None:
        args.steps = DEFAULT_STEPS[args.config][0]
    if args.warmup is None:
        args.warmup = DEFAULT_STEPS[args.config][1]
    if args.gpus < 1 or args.steps < 1 or args.warmup < 0:
        raise SystemExit("--gpus >= 1, --steps >= 1, --warmup >= 0")
    if args.leg is None:
        argv = [a for a in sys.argv[1:]]
        sys.exit(supervise(args, argv))

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
    out, side = leg_main(args, rank, local_rank, world)
    sys.stdout.flush()
    os.dup2(saved_stdout, 1)
    os.close(saved_stdout)
    if args.side_file:
        with open(args.side_file, "w") as f:
            json.dump(side, f)
    if rank == 0 and out is not None:
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
