"""CPU: the C ABI library loads and exports every symbol include/icp_mi355x.h declares, the host-only
entry points agree with numpy / the oracle, the generators and readers reproduce the fixtures, and the
device entry points fail loudly (no CPU fallback) when no GPU is present."""
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADERS = [os.path.join(ROOT, "include", h) for h in ("icp_mi355x.h", "icp_mi355x_diag.h")]


def declared_symbols():
    src = "".join(open(h).read() for h in HEADERS)
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(icp_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol(pkg):
    lib = pkg.load()
    names = declared_symbols()
    assert len(names) >= 40
    out = subprocess.run(["nm", "-D", "--defined-only", pkg.capi.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = set(re.findall(r" T (icp_[a-z0-9_]+)", out))
    for n in names:
        assert n in exported, f"{n} declared in the header but not exported"
        assert hasattr(lib, n)
    # and the Python stub binds exactly the header's surface
    assert sorted(pkg.capi.SIGNATURES) == names
    assert lib.icp_abi_version() == 2


def test_matching_isa_has_no_fused_multiply_add():
    """bit-exact correspondences need separately rounded sub/mul/add: the ISA of every nn_match / knn
    kernel must not contain fp fma/mad/fmac (integer mad for addressing is fine)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "fast-point-cloud-registration-with-gpus_amd", "csrc", "build", "icp_k_*.s")))
    assert len(files) == 6, "run `python __graft_entry__.py build` first"   # one translation unit per kernel family (csrc/Makefile)
    text = "".join(open(f).read() for f in files)
    kernels = re.findall(r"^(_ZN3icp\w*(?:nn_match|knn4)\w*):[^\n]*\n(.*?)\.Lfunc_end", text, flags=re.S | re.M)
    assert len(kernels) >= 6
    bad = re.compile(r"\bv_(?:pk_)?(?:fma|fmac|mad|mac)_(?:f32|f64|legacy_f32)")
    for name, body in kernels:
        hits = bad.findall(body)
        assert not hits, f"{name}: {hits[:3]}"
        if "merge" not in name:   # the merge kernels only compare
            assert re.search(r"v_(?:pk_)?mul_f(32|64)", body) and re.search(r"v_(?:pk_)?add_f(32|64)", body)


def test_no_device_is_loud(pkg):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(pkg.IcpError) as e:
        pkg.Context(0)
    assert e.value.code == pkg.capi.ICP_ERR_NO_DEVICE
    exe = os.path.join(ROOT, "fast-point-cloud-registration-with-gpus_amd", "bin", "icp_standard")
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode != 0 and "no usable gfx950" in r.stderr


def _moments(P, Q, idx):
    P = np.asarray(P, dtype=np.float64)
    Qi = np.asarray(Q, dtype=np.float64)[idx]
    mom = np.zeros(32)
    mom[1] = P.shape[0]
    mom[2:5] = P.sum(0)
    mom[5:8] = Qi.sum(0)
    mom[8:17] = (Qi.T @ P).reshape(9)
    mom[17] = (P * P).sum()
    mom[18] = (Qi * Qi).sum()
    return mom


def test_solve_point_to_point_matches_oracle_and_lapack(pkg, orc):
    rng = np.random.default_rng(11)
    P = rng.standard_normal((400, 3)) + 3.0
    Q = rng.standard_normal((350, 3)) + 2.5
    idx = orc.nn(P, Q)
    R, t = pkg.solve_point_to_point(_moments(P, Q, idx))
    Ro, to, _ = orc.p2p_minimize(P, Q, idx)
    assert np.abs(R - Ro).max() < 1e-11 and np.abs(t - to).max() < 1e-11
    # rank-deficient / reflected inputs must not crash and stay orthogonal
    for N in (np.diag([1.0, 2.0, 0.0]), -np.eye(3), np.zeros((3, 3)), np.diag([3.0, -2.0, 1.0])):
        mom = np.zeros(32)
        mom[1] = 1
        mom[8:17] = N.reshape(9)
        R, t = pkg.solve_point_to_point(mom)
        assert np.abs(R @ R.T - np.eye(3)).max() < 1e-12
    R, _ = pkg.solve_point_to_point(np.concatenate([[0, 1, 0, 0, 0, 0, 0, 0], np.diag([3.0, -2.0, 1.0]).reshape(9), np.zeros(15)]))
    assert np.linalg.det(R) < 0  # no reflection fix, like src/ICP_CPU.c:246


def test_solve_point_to_plane_matches_numpy(pkg):
    rng = np.random.default_rng(5)
    A = rng.standard_normal((40, 6))
    Cm = A.T @ A
    x_true = np.array([0.02, -0.01, 0.03, 0.1, -0.2, 0.05])
    b = Cm @ x_true
    mom = np.zeros(32)
    mom[2:23] = Cm[np.triu_indices(6)]
    mom[23:29] = b
    R, t, x = pkg.solve_point_to_plane(mom)
    assert np.abs(x - x_true).max() < 1e-10
    a, be, g = x_true[:3]
    Rx = np.array([[1, 0, 0], [0, np.cos(a), -np.sin(a)], [0, np.sin(a), np.cos(a)]])
    Ry = np.array([[np.cos(be), 0, np.sin(be)], [0, 1, 0], [-np.sin(be), 0, np.cos(be)]])
    Rz = np.array([[np.cos(g), -np.sin(g), 0], [np.sin(g), np.cos(g), 0], [0, 0, 1]])
    assert np.abs(R - Rz @ Ry @ Rx).max() < 1e-12 and np.abs(t - x_true[3:]).max() < 1e-10
    mom[2:23] = 0.0
    with pytest.raises(pkg.IcpError) as e:
        pkg.solve_point_to_plane(mom)
    assert e.value.code == pkg.capi.ICP_ERR_SINGULAR


def test_eigh3_matches_numpy(pkg):
    rng = np.random.default_rng(2)
    for _ in range(20):
        B = rng.standard_normal((3, 3))
        A = B @ B.T
        w, Z = pkg.eigh3(np.triu(A))
        w2, _ = np.linalg.eigh(A)
        assert np.abs(w - w2).max() < 1e-12 * max(1.0, abs(w2).max())
        assert np.abs(A @ Z - Z * w).max() < 1e-12 * max(1.0, abs(w2).max())


def test_shard_range_partitions(pkg):
    for n in (0, 1, 7, 16384, 10_000_000):
        for world in (1, 2, 3, 8):
            spans = [pkg.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and sum(c for _, c in spans) == n
            for (b0, c0), (b1, _) in zip(spans, spans[1:]):
                assert b0 + c0 == b1
            assert max(c for _, c in spans) - min(c for _, c in spans) <= 1


def test_generators_match_oracle_bitwise(pkg, orc):
    ds = pkg.datasets
    D = ds.synthetic_grid(32, np.float32)
    assert np.array_equal(D, orc.synth_grid_f32(32))
    assert np.array_equal(ds.make_model_gpu(D, *ds.P2P_GPU), orc.gpu_model_f32(D, *ds.P2P_GPU))
    Do, Mo = orc.synth_icp_standard(32)
    assert np.array_equal(D, Do) and np.array_equal(ds.make_model_standard(D), Mo)
    D64 = ds.synthetic_grid(20, np.float64)
    Dc, Mc = orc.synth_icp_cpu(20)
    assert np.array_equal(D64, Dc) and np.array_equal(ds.make_model_cpu(D64), Mc)


def test_readers_match_fixtures(pkg, golden):
    ds = pkg.datasets
    for head, full in (("bunny_res_head.csv", "bunny_res_xyz_f32.bin"), ("bunny_head.csv", "bunny_xyz_f32.bin")):
        pts = ds.read_xyz_text(os.path.join(golden, head))
        want = np.fromfile(os.path.join(golden, full), dtype=np.float32).reshape(-1, 3)
        assert np.array_equal(pts, want[:64])
    r, enc = ds.read_os1_ranges(os.path.join(golden, "os1_two_packets.csv"))
    want = np.fromfile(os.path.join(golden, "hall_ranges_u32.bin"), dtype=np.uint32)
    assert enc == 33616 and r.size == 512 and np.array_equal(r, want[:512])
    alt, az = ds.read_os1_intrinsics(os.path.join(golden, "beam_intrinsics.csv"))
    assert alt.shape == az.shape == (16,) and abs(alt[0] - 15.379) < 1e-6 and abs(az[15] + 0.857) < 1e-6
    # binary packet form of the same dump
    vals = np.array([int(x) for x in open(os.path.join(golden, "os1_two_packets.csv")).read().split()], dtype=np.uint8)
    path = os.path.join(golden, "_tmp_packets.bin")
    vals.tofile(path)
    try:
        r2, enc2 = ds.read_os1_ranges(path)
    finally:
        os.remove(path)
    assert enc2 == enc and np.array_equal(r2, r)
    with pytest.raises(pkg.IcpError) as e:
        ds.read_xyz_text(os.path.join(golden, "does_not_exist.csv"))
    assert e.value.code == pkg.capi.ICP_ERR_IO


def test_library_embeds_gfx950_code_object(pkg):
    """the fat binary must carry a gfx950 image (a wrong/default --offload-arch shows up here, on CPU,
    instead of as 'No compatible code objects' on the GPU box)"""
    blob = open(pkg.capi.LIB_PATH, "rb").read()
    assert b"hipv4-amdgcn-amd-amdhsa--gfx950" in blob
    assert b"amdhsa--gfx906" not in blob


def test_bench_executed_flop_arithmetic():
    """bench.py's roofline counts EXECUTED flop from the kernel's work counters: the conversion, on known tallies"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    work = dict(find_boxes=2048, upper_boxes=0, hits_box=10, hits_xy=6, hits_full=4, sample_groups=2, block_passes=1, block_transforms=1)
    total, parts = bench.executed_flop(work)
    assert parts["find (group box vs chunk boxes)"] == 2048 * 12
    assert parts["per-point box tests"] == 10 * 128 * 12
    assert parts["xy halves"] == 6 * 128 * 8 * 5 and parts["z halves"] == 4 * 128 * 8 * 3
    assert parts["cold-start samples"] == 2 * 128 * 8 * 8
    assert parts["seed distances"] == 16 * 128 * 8 and parts["transforms"] == 16 * 128 * 18
    assert total == sum(parts.values())
    # a hit evaluated in full costs exactly the reference's 8 flop per pair, plus its box test
    full = bench.executed_flop(dict(work, find_boxes=0, hits_box=1, hits_xy=1, hits_full=1, sample_groups=0, block_passes=0, block_transforms=0))[0]
    assert full == 128 * 8 * 8 + 128 * 12


def test_bench_refuses_a_rank_count_that_does_not_match_the_launcher():
    import subprocess, sys
    env = dict(os.environ, WORLD_SIZE="1", RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "does not match WORLD_SIZE" in r.stderr


def test_shared_rows_plan_always_fits_the_grid(pkg):
    """the role assignment of a launch with shared rows (the arithmetic nn_match_sparse runs in every block, mirrored on the host
    by icp_share_rows_plan): every row gets at least one block, never more than 32, the parts never outnumber the blocks --
    whatever the counters hold, including the sums past 2^32 that a soak once produced (rows missing a part never close) --
    a heavy row gets more parts than a light one, and the split is as fine as the grid allows"""
    rng = np.random.default_rng(11)
    cases = []
    for _ in range(300):
        rows = int(rng.integers(1, 513))
        blocks = int(rows + rng.integers(0, 513))
        kind = rng.integers(0, 5)
        if kind == 0:
            hits = rng.integers(0, 200, rows)
        elif kind == 1:
            hits = (rng.pareto(1.2, rows) * 80).astype(np.int64)
        elif kind == 2:
            hits = rng.integers(0, 2**32, rows)                    # counters that were never zeroed
        elif kind == 3:
            hits = np.zeros(rows, dtype=np.int64)
        else:
            hits = np.full(rows, 2**32 - 1)
        cases.append((np.minimum(hits, 2**32 - 1).astype(np.uint32), blocks, int(rng.choice([300, 4096, 35947, 65536]))))
    for hits, blocks, m in cases:
        parts, target = pkg.share_rows_plan(hits, blocks, m)
        assert parts.min() >= 1 and parts.max() <= 32 and parts.sum() <= blocks, (hits[:8], blocks, m)
        assert target >= 64
        o = np.argsort(np.minimum(hits, 1 << 20), kind="stable")
        assert (np.diff(parts[o]) >= 0).all()                       # monotone in the (clamped) count
    # Bunny.csv late in a registration: 288 rows, 512 blocks, a few rows far heavier than the rest
    hits = np.full(288, 70, dtype=np.uint32); hits[[175, 244, 90]] = (1147, 934, 538)
    parts, target = pkg.share_rows_plan(hits, 512, 35947)
    assert parts[175] >= 8 and parts[244] >= 6 and parts[90] >= 4 and target <= 128
    assert int(np.ceil(hits / parts).max()) <= target
    # no spare blocks, or nothing known: one block per row
    assert (pkg.share_rows_plan(hits, 288, 35947)[0] == 1).all()
    assert (pkg.share_rows_plan(np.zeros(288, dtype=np.uint32), 512, 35947)[0] == 1).all()
    with pytest.raises(pkg.IcpError):
        pkg.share_rows_plan(hits, 100, 35947)                      # fewer blocks than rows


def test_bench_region_statistics():
    """round 4: `value` = K over the MEDIAN of the repeated K-step regions, the spread beside it -- never a single sample"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod2", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    rs = bench.region_stats([0.30e-3, 0.22e-3, 0.25e-3, 0.21e-3, 0.90e-3], 20)
    assert rs["repeats"] == 5 and abs(rs["value"] - 20 / 0.25e-3) < 1e-6
    assert abs(rs["ms_per_step"] - 0.25 / 20) < 1e-12 and abs(rs["ms_per_step_min"] - 0.21 / 20) < 1e-12 and abs(rs["ms_per_step_max"] - 0.90 / 20) < 1e-12
    assert bench.cpu_model() != ""
