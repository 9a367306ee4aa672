"""GPU: behaviour of the library around the hot path -- the caller's CPU affinity, the mailbox time budgets (a host thread
that comes back late), the executed-work counters, and bench.py under the driver's own command lines."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _child(code, env=None, timeout=300):
    pre = ("import sys, os, json, numpy as np\n"
           f"sys.path.insert(0, {ROOT!r}); sys.path.insert(0, os.path.join({ROOT!r}, 'tests'))\n"
           "from __graft_entry__ import load_package\n"
           "pkg = load_package()\n")
    out = subprocess.run([sys.executable, "-c", pre + code], env=dict(os.environ, **(env or {})), capture_output=True, text=True,
                         timeout=timeout)
    assert out.returncode == 0, out.stderr[-3000:]
    return json.loads(out.stdout.strip().splitlines()[-1])


# ---------------------------------------------------------------------------------------------------
# a drop-in library leaves its caller's affinity as it found it
# ---------------------------------------------------------------------------------------------------
AFFINITY_CODE = """
before = sorted(os.sched_getaffinity(0))
D = pkg.datasets.synthetic_grid(48, np.float32)
M = pkg.datasets.make_model_gpu(D, *pkg.datasets.P2P_GPU)
seen = {}
with pkg.Context(0) as ctx:
    seen["create"] = sorted(os.sched_getaffinity(0))
    r = ctx.point_to_point(D, M, max_iter=20, tol=1e-6)
    seen["point_to_point"] = sorted(os.sched_getaffinity(0))
    ctx.set_model(M); ctx.set_moving(D)
    ctx.loop_begin(pkg.ICP_POINT_TO_POINT, max_iter=20, tol=1e-6)
    ctx.loop_run(3)
    seen["loop_run"] = sorted(os.sched_getaffinity(0))
    ctx.loop_enqueue(); ctx.loop_complete()
    seen["complete"] = sorted(os.sched_getaffinity(0))
seen["destroy"] = sorted(os.sched_getaffinity(0))
print(json.dumps(dict(before=before, seen=seen, T=r.T.tolist(), it=r.iterations)))
"""


@pytest.mark.parametrize("pin", [None, "0", "1"])
def test_callers_affinity_is_left_as_found(pin):
    """icp_create narrowed the caller's mask and never restored it.  Now the narrowing is scoped to the entry
    points that talk to the GPU in a loop (ICP_PIN=1, the default) or absent (ICP_PIN=0)."""
    env = {} if pin is None else {"ICP_PIN": pin}
    got = _child(AFFINITY_CODE, env)
    for where, mask in got["seen"].items():
        assert mask == got["before"], f"ICP_PIN={pin}: affinity changed after {where}"


def test_pin_settings_do_not_change_results():
    a = _child(AFFINITY_CODE, {"ICP_PIN": "0"})
    b = _child(AFFINITY_CODE, {"ICP_PIN": "1"})
    c = _child(AFFINITY_CODE.replace('for where', '#'), {"ICP_PIN": "2"})      # sticky narrowing (opt-in): still the same bits
    assert a["T"] == b["T"] == c["T"] and a["it"] == b["it"] == c["it"]
    cpus = os.cpu_count() or 1
    if len(c["before"]) == cpus and any(len(m) < len(c["before"]) for m in c["seen"].values()):
        assert set(c["seen"]["destroy"]) <= set(c["before"])                    # ICP_PIN=2 narrows and keeps it


# ---------------------------------------------------------------------------------------------------
# a host that comes back late: the waiting kernel is withdrawn and the registration resumes -- same bits
# ---------------------------------------------------------------------------------------------------
STALL_CODE = """
import oracle_lib
orc = oracle_lib.Oracle()
P, Q = orc.hall_clouds(os.path.join(%r, 'tests', 'golden'))
with pkg.Context(0) as ctx:
    r = ctx.point_to_point(P, Q, max_iter=100, tol=1e-6)
    moved = ctx.get_moving()
print(json.dumps(dict(it=r.iterations, T=r.T.tolist(), err=r.err.tolist(), sec=r.seconds_total,
                      idx=int(np.bitwise_xor.reduce(r.idx * np.arange(1, r.idx.size + 1, dtype=np.int64))),
                      moved=float(np.abs(moved).sum()))))
""" % ROOT


@pytest.mark.parametrize("form", ["resident", "armed"])
@pytest.mark.parametrize("stall", ["3:1.7", "5:4.6"])
def test_late_host_withdraws_and_resumes(form, stall):
    """ICP_DEBUG=stall=pass:seconds makes the host sleep before it publishes one pass's message: 1.7 s is past the host's lease (it
    withdraws the kernel itself), 4.6 s is past the kernel's own wall-clock budget (every block has given up).  Either
    way no block may act on a message another block missed: the run must end with exactly the bits of an undisturbed one."""
    env = {"ICP_RESIDENT": "0"} if form == "armed" else {}
    ref = _child(STALL_CODE, env)
    got = _child(STALL_CODE, dict(env, ICP_DEBUG="stall=" + stall))
    assert got["sec"] > float(stall.split(":")[1]) - 0.2            # (the stall really happened)
    for k in ("it", "T", "err", "idx", "moved"):
        assert got[k] == ref[k], k


LOST_CODE = """
import oracle_lib
orc = oracle_lib.Oracle()
P, Q = orc.hall_clouds(os.path.join(%r, 'tests', 'golden'))
with pkg.Context(0) as ctx:
    r = ctx.point_to_point(P, Q, max_iter=100, tol=1e-6)
    moved = ctx.get_moving()
    rec = ctx.recoveries()
    r2 = ctx.point_to_point(P, Q, max_iter=100, tol=1e-6)       # the context goes on as before
    rec2 = ctx.recoveries()
print(json.dumps(dict(it=r.iterations, T=r.T.tolist(), err=r.err.tolist(), sec=r.seconds_total, rec=rec, rec2=rec2, sec2=r2.seconds_total, T2=r2.T.tolist(),
                      idx=int(np.bitwise_xor.reduce(r.idx * np.arange(1, r.idx.size + 1, dtype=np.int64))),
                      moved=float(np.abs(moved).sum()))))
""" % ROOT


@pytest.mark.parametrize("form", ["resident", "armed"])
def test_a_pass_that_never_delivers_is_finished_step_wise(form):
    """ICP_DEBUG=lose=pass: the message of one pass is never posted -- what a block that never sees its message, or blocks
    kept off the machine, look like from the host.  Round 2 returned ICP_ERR_HIP there.  Now the waiting kernel is withdrawn
    (its own wall-clock budget ends it) and the registration is run again from the uploaded cloud with plain launches: the
    caller gets the bits of an undisturbed run, icp_recoveries counts it, and the context keeps working at full speed."""
    env = {"ICP_RESIDENT": "0"} if form == "armed" else {}
    ref = _child(LOST_CODE, env)
    got = _child(LOST_CODE, dict(env, ICP_DEBUG="lose=3"))
    assert ref["rec"] == 0 and got["rec"] == 1 and got["rec2"] == 1
    assert got["sec"] > 1.5                                          # (the pass really went missing: the row poll's 2 s)
    assert got["sec2"] < 0.1 and got["T2"] == ref["T"]
    for k in ("it", "T", "err", "idx", "moved"):
        assert got[k] == ref[k], k


def test_exclusive_device_gives_the_same_bits(pkg, orc, golden):
    """icp_set_exclusive(ctx, 1): the caller owns the device, the hall pair's rows of 64 points run as 16-wave blocks (one to a CU,
    no room left for a second resident context).  Same search, same arithmetic: the same bits, point-to-point and point-to-plane"""
    P, Q = orc.hall_clouds(golden)
    with pkg.Context(0) as ctx:
        a = ctx.point_to_point(P, Q, max_iter=100, tol=1e-6)
        ctx.set_exclusive(True)
        b = ctx.point_to_point(P, Q, max_iter=100, tol=1e-6)
        ctx.set_model(Q); ctx.set_moving(P); ctx.estimate_normals()
        pb = ctx.point_to_plane(P, Q, max_iter=50, tol=1e-6)
        ctx.set_exclusive(False)
        c = ctx.point_to_point(P, Q, max_iter=100, tol=1e-6)
        pc = ctx.point_to_plane(P, Q, max_iter=50, tol=1e-6)
    for r in (b, c):
        assert r.iterations == a.iterations and np.array_equal(r.T, a.T) and np.array_equal(r.err, a.err) and np.array_equal(r.idx, a.idx)
    assert pb.iterations == pc.iterations and np.array_equal(pb.T, pc.T) and np.array_equal(pb.idx, pc.idx)


# ---------------------------------------------------------------------------------------------------
# executed-work counters (the roofline of the pruned search is about EXECUTED arithmetic)
# ---------------------------------------------------------------------------------------------------
def test_work_counters(ctx, pkg, orc, golden):
    P, Q = orc.hall_clouds(golden)
    ctx.set_model(Q)
    ctx.set_moving(P)
    plain = ctx.nn_match_resident()
    idx_plain = ctx.get_indices()
    ctx.set_work_counting(True)
    try:
        ctx.nn_match_resident()
        w = ctx.get_work_counters()
        assert np.array_equal(ctx.get_indices(), idx_plain)            # the instrumented instantiation is the same search
        info = ctx.nn_launch_info()
        blocks = info["blocks"]
        assert w["block_passes"] == blocks and w["block_transforms"] == 0
        assert w["find_boxes"] == blocks * (info["m_pad"] // 8) // info["splits"]   # every chunk box is tested once per block
        assert w["hits_box"] >= w["hits_xy"] >= w["hits_full"] > 0
        # a cold pass evaluates a few per cent of the pairs in full -- and never more than all of them
        pts_per_hit = 64 if info["threads"] == 512 else 128               # a wave's moving points: rows of 64 / of 128
        frac = w["hits_full"] * pts_per_hit * 8 / (P.shape[0] * Q.shape[0])
        assert 1e-4 < frac < 0.2, frac
        assert ctx.get_work_counters()["block_passes"] == 0            # reading resets
    finally:
        ctx.set_work_counting(False)
    ctx.nn_match_resident()
    assert np.array_equal(ctx.get_indices(), idx_plain)


def test_per_launch_timing_modes(ctx, pkg, orc, golden):
    P, Q = orc.hall_clouds(golden)
    ctx.set_model(Q)
    ctx.set_moving(P)
    ctx.nn_match_resident()
    seeded = ctx.nn_match_bench_launches(5, 2, 0)
    cold = ctx.nn_match_bench_launches(5, 2, 1)
    dense = ctx.nn_match_bench_launches(5, 2, 2)
    assert seeded.shape == (5,) and (seeded > 0).all() and (cold > 0).all() and (dense > 0).all()
    # the dense kernel executes all 2.7e8 pairs: at the fp32 peak that alone takes 13.7 us
    assert dense.min() * 1e-3 > 13.7e-6
    assert dense.min() > seeded.min()
    assert ctx.nn_launch_info_ex(dense=True)["n_pad"] == 16384


# ---------------------------------------------------------------------------------------------------
# bench.py under the driver's command lines
# ---------------------------------------------------------------------------------------------------
def _bench(args, env=None, timeout=420):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=dict(os.environ, **(env or {})),
                         capture_output=True, text=True, timeout=timeout)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, out.stdout[-2000:]                          # ONE JSON line on stdout
    return json.loads(lines[0])


def test_bench_with_the_drivers_arguments():
    """round 1's bench died with ZeroDivisionError on exactly this command (BENCH_r01.json)"""
    d = _bench(["--gpus", "1", "--steps", "20", "--warmup", "5"])
    assert d["n_gpus"] == 1 and d["steps"] == 20 and d["warmup"] == 5 and d["value"] > 0 and d["unit"] == "iterations/s"
    assert abs(d["value"] - 20 / (d["ms_per_step"] * 20e-3)) < 1e-6 * d["value"]
    # round 4: the 0.2 ms region is repeated (each repeat between its own barriers), the median is the figure, the spread beside it
    assert d["repeats"] >= 25 and d["ms_per_step_min"] <= d["ms_per_step"] <= d["ms_per_step_max"]
    assert d["config"]["registrations_per_region"] == 2 and d["config"]["registrations_timed"] == 2 * d["repeats"]
    assert "aggregate_iterations_per_s" not in d
    # ... what a pair the context has not seen costs, and the host's model string beside the CPU figure
    assert 0 < d["setup_ms"] < d["fresh_pair_ms"] and len(d["fresh_pair"]["iterations"]) == 10
    assert d["cpu_baseline"]["cpu_model"] and d["cpu_baseline"]["cpu_model"] != "unknown"
    assert "traffic_from_committed_profile" in d["roofline"] and "traffic_source" not in d["roofline"]
    # ... and the config north_star shards, in brief, beside the hall line (the anchor of its scaling curve)
    assert d["s5"]["value"] > 0 and d["s5"]["global_moving_points"] == 10_000_000 and d["s5"]["repeats"] == 3
    r = d["roofline"]
    assert r["bound"] == "valu" and r["unit"] == "TFLOP/s" and r["peak"] == 157.3
    assert r["launches_timed"] >= 1 and r["avg_launch_us"] > 0 and r["timed_in"]
    assert 0 < r["frac"] <= 1.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert 0 < r["dense_kernel"]["frac"] <= 0.5 + 1e-9                  # no FMA: half the FMA-counted roof at most
    assert 0 < r["hbm"]["frac"] <= 1.0
    assert r["matching_only"]["min_launch_us"] <= r["matching_only"]["avg_launch_us"]
    assert r["executed"]["pairs_evaluated_in_full_fraction"] < 0.2
    c = d["cpu_baseline"]
    assert c["value"] > 0 and c["cores"] == 1 and c["kind"] == "port" and c["sample"]
    assert d["final_rms_error"] < 1e-3


@pytest.mark.parametrize("cfg,extra", [("hall_plane", ["--steps", "40", "--warmup", "10"]), ("bunny", ["--steps", "44", "--warmup", "0"]),
                                       ("s5", ["--points", "600000", "--steps", "8", "--warmup", "2"]), ("cpu_f64", ["--steps", "120", "--warmup", "10"])])
def test_bench_every_config_carries_roofline_and_cpu_baseline(cfg, extra):
    """BASELINE configs[0], [1], [3], [4] through `bench.py --config ...`: a line with an EXECUTED-work roofline of the config's own
    kernel (frac <= 1, recomputable from its parts) and a CPU baseline timed beside it"""
    d = _bench(["--config", cfg] + extra)
    assert d["fresh_pair_ms"] > d["setup_ms"] > 0 and d["repeats"] >= 3
    assert d["value"] > 0 and d["unit"] == "iterations/s" and d["dtype"] == ("f64" if cfg == "cpu_f64" else "f32")
    r = d["roofline"]
    assert r["peak"] == (78.6 if cfg == "cpu_f64" else 157.3) and r["unit"] == "TFLOP/s" and r["bound"] == "valu"
    assert 0 < r["frac"] <= 1.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    ex = r["executed"]
    assert abs(r["achieved"] - ex["flop_per_launch"] / (r["avg_launch_us"] * 1e-6) / 1e12) < 1e-9 * max(1.0, r["achieved"])
    assert abs(ex["flop_per_launch"] - ex["flop_per_pass"] * r["passes_per_launch"]) < 1e-6 * ex["flop_per_launch"]
    assert abs(sum(ex["flop_by_part_one_registration"].values()) - ex["flop_per_pass"] * ex["passes_counted"]) < 1e-6 * ex["flop_per_pass"] * ex["passes_counted"]
    assert ex["work_counters_one_registration"]["hits_full"] > 0 and 0 < ex["pairs_evaluated_in_full_fraction"] <= 1.0
    assert r["launches_timed"] >= 1 and r["avg_launch_us"] > 0
    c = d["cpu_baseline"]
    assert c["value"] > 0 and c["cores"] == 1 and c["kind"] == "port" and c["sample"]
    if cfg == "s5":
        assert c["extrapolated"] is True and "EXTRAPOLATED" in c["sample"]


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher: the parent spawns the ranks (rehearsed with both on the one GPU)"""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["ICP_BENCH_ONE_DEVICE"] = "1"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "120", "--warmup", "20", "--no-s5"], env=env,
                         capture_output=True, text=True, timeout=420)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    # the default for N > 1 is north_star's split of the metric's own cloud: 16 384 moving points over the ranks, ONE registration
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["scaling"] == "strong"
    assert d["config"]["global_moving_points"] == 16384 and d["config"]["moving_points_per_gpu"] == 8192
    assert "shared host memory" in d["config"]["collective"] and d["config"]["collective_route"] == "local"
    # `value` is an iteration rate: K over the median region, never multiplied by the ranks
    assert abs(d["value"] - 120 / (d["ms_per_step"] * 120e-3)) < 1e-6 * d["value"] and "aggregate_iterations_per_s" not in d
    # beside it: a hall-sized shard per rank (one registration of a 2 x 16 384-point cloud), also as K / dt
    w = d["weak_shards"]
    assert w["global_moving_points"] == 2 * 16384 and w["scaling"] == "weak" and w["value"] > 0 and w["route"] == "local"
    assert abs(w["aggregate_moving_point_iterations_per_s"] - 2 * 16384 * w["value"]) < 1e-6 * w["aggregate_moving_point_iterations_per_s"]
    # RCCL refuses two ranks on one device: the leg must say so instead of hanging or killing the line
    assert d["rccl"]["ranks"] == 2 and ("error" in d["rccl"] or d["rccl"]["value"] > 0)
    assert 0 < d["roofline"]["frac"] <= 1.0


def test_bench_sharded_cloud_two_ranks_on_one_device():
    """`bench.py --config s5 --gpus 2` (the moving cloud sharded over the ranks, large-model kernels, rows taken heaviest first),
    rehearsed with both ranks on the one GPU: the library sees that two ranks of its node communicator share a device and arms
    no pass ahead of its transform -- the waiting blocks of one rank would keep the other's pass off the CUs, and with the
    exchange between the ranks that is a circular wait (a pass missing its last rows, found with exactly this command).
    Run twice: with the rows added up inside the matching launch (round 4's default from 1 025 rows up: 4 688 here) and
    (ICP_HOST_ROWS_MAX=16384) by the host as they arrive -- the same registration either way."""
    errs = []
    for rows_max in (None, "16384"):
        env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
        env["ICP_BENCH_ONE_DEVICE"] = "1"
        if rows_max:
            env["ICP_HOST_ROWS_MAX"] = rows_max
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", "s5", "--points", "600000", "--gpus", "2", "--steps", "12", "--warmup", "2",
                              "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=420)
        assert out.returncode == 0, out.stderr[-3000:]
        d = json.loads([ln for ln in out.stdout.splitlines() if ln.strip().startswith("{")][0])
        assert d["n_gpus"] == 2 and d["value"] > 0 and d["config"]["global_moving_points"] == 600000
        assert d["rms_error_series_head"][1] > d["final_rms_error"] > 0
        # north_star's route for the config it shards is RCCL: asked for, refused for two ranks on ONE device, said so, and the line
        # falls back to the node-local route (on two devices `value` is the RCCL figure and `local` the one beside it)
        assert d["config"]["collective_route"] in ("rccl", "local")
        if d["config"]["collective_route"] == "local":
            assert "rccl" in d["config"]["collective_route_note"]
        assert d["local"]["value"] > 0 and d["local"]["route"] == "local"
        assert abs(d["local"]["final_rms_error"] - d["final_rms_error"]) < 1e-9
        errs.append(d["final_rms_error"])
    assert abs(errs[0] - errs[1]) < 1e-12 * errs[0], errs


def test_bench_under_torch_distributed_run():
    """the driver's launch line for N > 1 (`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`),
    rehearsed with both ranks on the one GPU"""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["ICP_BENCH_ONE_DEVICE"] = "1"
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", "29637", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "120", "--warmup", "20", "--no-s5"],
                         env=env, capture_output=True, text=True, timeout=420)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["scaling"] == "strong"
    assert d["rccl"]["ranks"] == 2 and d["weak_shards"]["value"] > 0


def test_bench_two_gpus_rccl_leg():
    """on a box with two or more GPUs: two ranks on two devices, the RCCL leg must report a rate for both ranks"""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (the driver's scaling run exercises this path)")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "ICP_BENCH_ONE_DEVICE")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "200", "--warmup", "20", "--no-s5"], env=env,
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    d = json.loads([ln for ln in out.stdout.splitlines() if ln.strip().startswith("{")][0])
    assert d["n_gpus"] == 2 and d["value"] > 0
    assert d["rccl"]["ranks"] == 2 and d["rccl"].get("value", 0) > 0, d["rccl"]


def test_bench_rccl_leg_single_rank():
    """the library-issued ncclAllReduce route, exercised with the one rank this box has"""
    d = _bench(["--steps", "120", "--warmup", "20", "--no-cpu-baseline", "--no-s5", "--no-fresh-pair"], {"ICP_BENCH_FORCE_DIST": "1"})
    assert d["rccl"]["ranks"] == 1 and d["rccl"]["value"] > 0 and d["rccl"]["us_per_iteration"] > 0


def test_bench_rccl_leg_on_the_sharded_config_single_rank():
    """configs[4] is the config north_star shards: its line carries the library-issued ncclAllReduce route too"""
    d = _bench(["--config", "s5", "--points", "600000", "--steps", "8", "--warmup", "2", "--no-cpu-baseline", "--no-fresh-pair"], {"ICP_BENCH_FORCE_DIST": "1"})
    assert d["rccl"]["ranks"] == 1 and d["rccl"]["value"] > 0 and d["rccl"]["us_per_iteration"] > 0
    assert abs(d["rccl"]["final_rms_error"] - d["final_rms_error"]) < 1e-9      # the same registration through either route


def test_bench_a_stuck_rccl_leg_is_visible():
    """a communicator attempt that never answers: the process is killed at its time limit, the line is still printed (with the
    reason) and the run ends NON-zero -- never rc 0 (round 2 left through os._exit(0))"""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "40", "--warmup", "5", "--no-cpu-baseline", "--no-s5", "--no-fresh-pair"],
                         env=dict(os.environ, ICP_BENCH_FORCE_DIST="1", ICP_BENCH_RCCL_TIMEOUT="0.2"), capture_output=True, text=True, timeout=420)
    assert out.returncode == 3, (out.returncode, out.stderr[-2000:])
    d = json.loads([ln for ln in out.stdout.splitlines() if ln.strip().startswith("{")][0])
    assert d["value"] > 0 and d["rccl"]["hung"] is True and "killed" in d["rccl"]["error"]


# ---------------------------------------------------------------------------------------------------
# round 4: the advisor's findings on round 3
# ---------------------------------------------------------------------------------------------------
def test_node_communicator_and_external_moments_buffer_refuse_each_other(pkg):
    """an external moments buffer is reduced by its owner between enqueue and complete; with the node communicator on top the
    ranks' sums would be added twice (silently: a wrong transform).  Either order of the two calls is ICP_ERR_STATE."""
    import torch
    buf = torch.zeros(32, dtype=torch.float64, device="cuda:0")
    D = pkg.datasets.synthetic_grid(32, np.float32)
    M = pkg.datasets.make_model_gpu(D, *pkg.datasets.P2P_GPU)
    with pkg.Context(0) as c:
        c.set_model(M); c.set_moving(D)
        c.comm_init_local(pkg.Context.comm_random_id(), 0, 1)
        with pytest.raises(pkg.IcpError) as e:
            c.loop_set_moments_dev(buf.data_ptr())
        assert e.value.code == pkg.capi.ICP_ERR_STATE
        c.comm_destroy()
        c.loop_set_moments_dev(buf.data_ptr())
        with pytest.raises(pkg.IcpError) as e:
            c.comm_init_local(pkg.Context.comm_random_id(), 0, 1)
        assert e.value.code == pkg.capi.ICP_ERR_STATE
        # the buffer alone works as before: the caller sees the sums between enqueue and complete
        c.loop_begin(pkg.ICP_POINT_TO_POINT, max_iter=5, tol=1e-6)
        c.loop_enqueue()
        torch.cuda.synchronize()
        done = c.loop_complete()
        assert not done and float(buf[1].item()) == D.shape[0]
        c.loop_set_moments_dev(0)
        c.comm_init_local(pkg.Context.comm_random_id(), 0, 1)   # and with the library's own vector the communicator attaches again


@pytest.mark.parametrize("form", ["run", "stepwise"])
def test_a_numeric_failure_keeps_the_loop_readable(pkg, form):
    """the minimisation refusing a pass's sums (degenerate correspondences) ends the loop, but what its completed passes produced
    stays readable; only a DEVICE failure discards the loop (include/icp_mi355x.h, icp_loop_state)"""
    g = np.stack(np.meshgrid(np.arange(8.0), np.arange(8.0), indexing="ij"), -1).reshape(-1, 2)
    P = np.concatenate([g, np.zeros((64, 1))], 1).astype(np.float32)
    N = np.tile(np.array([[0, 0, 1]], dtype=np.float32), (64, 1))
    with pkg.Context(0) as c:
        c.set_model(P); c.set_model_normals(N); c.set_moving(P)
        c.loop_begin(pkg.ICP_POINT_TO_PLANE, max_iter=3, tol=1e-6)
        with pytest.raises(pkg.IcpError) as e:
            if form == "run":
                c.loop_run(3)
            else:
                c.loop_enqueue(); c.loop_complete()
        assert e.value.code == pkg.capi.ICP_ERR_SINGULAR
        st = c.loop_state()                                  # still answers
        assert st["passes"] == 0 and st["iterations"] == 0 and np.array_equal(st["T"], np.eye(4))
        assert np.array_equal(c.loop_indices(), np.arange(64, dtype=np.int32))   # the pass itself completed: a point is its own match
        assert np.array_equal(c.get_moving(), P)


@pytest.mark.parametrize("rows", [9768, 16384, 1500, 257])
def test_row_roles_of_the_single_workgroup_launch(ctx, pkg, rows):
    """ordered + split rows: the roles of a launch's blocks, by the single-workgroup control kernel (round 4) and by the
    keys + radix sort + roles chain it replaces -- every row exactly once (a split row: parts = 2^k blocks, parts 0..2^k-1), the
    heaviest rows first, spare blocks -1, the counters zeroed.  Any assignment is exact; this is what keeps it a complete one."""
    rng = np.random.default_rng(rows)
    hits = (rng.pareto(1.5, rows) * 150).astype(np.uint32)
    hits[rng.integers(0, rows, 7)] = rng.integers(100_000, 900_000, 7)       # a few very heavy rows: they get split
    hits[rng.integers(0, rows, 50)] = 0
    ROW, PART, LG = (1 << 21) - 1, 63, 7
    seen = {}
    for control in (True, False):
        roles, left = ctx.diag_row_roles(hits, min_part=2048, total_div=4096, control=control)
        assert not left.any()                                                # read AND zeroed
        live = roles[roles >= 0]
        row, part, parts = live & ROW, (live >> 21) & PART, 1 << ((live >> 27) & LG)
        assert roles.size == rows + 4096 and (roles[live.size:] == -1).all() # the roles come first, the spare blocks after them
        count = np.bincount(row, minlength=rows)
        first = {}
        for r, p, ps in zip(row.tolist(), part.tolist(), parts.tolist()):
            first.setdefault(r, (ps, set()))[1].add(p)
            assert first[r][0] == ps
        assert len(first) == rows                                            # every row has a role
        for r, (ps, got) in first.items():
            assert got == set(range(ps)) and count[r] == ps                  # ... and all of its parts, once each
        split = {r: ps for r, (ps, _) in first.items() if ps > 1}
        assert split and all(hits[r] >= 2048 for r in split)                 # only heavy rows are split
        assert max(split, key=lambda r: split[r]) in np.argsort(hits)[-8:]   # the most parts go to one of the heaviest
        # heaviest first, by weight classes (a count's leading one and one bit behind it: NN_ORDER_CLASS_BITS); the rows of a class
        # keep their order on the curve -- neighbours run side by side
        order = [r for r in row.tolist() if first[r][0] == 1]
        h = hits[order].astype(np.float64)
        assert (h[1:] < np.maximum(h[:-1] * 1.5, h[:-1] + 1)).all()
        cls = [int(v) if v < 8 else (int(v).bit_length(), (int(v) >> (int(v).bit_length() - 2)) & 1) for v in hits[order]]
        for k in range(1, len(order)):
            if cls[k] == cls[k - 1]: assert order[k] > order[k - 1]
        seen[control] = (split, roles.copy())
    assert set(seen[True][0]) == set(seen[False][0])                         # both forms split the same rows (exact counts decide)
    assert np.array_equal(seen[True][1], seen[False][1])                     # ... and deal the same roles: same classes, both sorts stable
