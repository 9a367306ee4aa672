"""GPU: parity of the HIP path (through the C ABI) against the CPU oracle.

Bar: correspondence indices bit-exact; composed transform within 1e-5 relative (north_star); error
series within 1e-5 absolute of the oracle's.  Trajectory parity is pinned PER ITERATION from captured
inputs (P_k downloaded from the device, idx_k compared with the oracle's matching of that same P_k), so a
last-bit difference in one transform cannot hide behind -- or be blamed on -- a later near-tie.
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

TOL_T = 1e-5      # relative, composed 4x4 transform (BASELINE.json north_star)
TOL_E = 1e-5      # absolute, RMS error series


def rel(a, b):
    return float(np.abs(np.asarray(a) - np.asarray(b)).max() / max(1e-300, np.abs(np.asarray(b)).max()))


TWIN_TOL = 2e-4   # the fp32 twin's own summation noise (naive float centroid sums over 16 384 points)


def assert_close_to_fp32_twin(res, twin):
    """loose: the letter-faithful fp32 twin (src/CUDA/CPU_ICP_point_to_point.cpp) carries ~1e-5..1e-4 of
    float summation noise in its centroids; the product (fp64 reductions) must stay inside that band."""
    n = min(len(res.err), len(twin["err"]))
    assert np.abs(res.err[:n] - twin["err"][:n]).max() < TWIN_TOL
    assert rel(res.T, twin["T"]) < TWIN_TOL


def assert_same_run(res_iterations, res_err, res_T, want, tol, fp32):
    """same trajectory as the oracle.  The fp64 path must stop at the same iteration.  The fp32 oracle
    (like the reference's snrm2) sums its error norm in float, ~1e-7 of noise on a 1e-6 stop threshold, so a
    run may legitimately stop one pass apart -- but only if the deciding |dE| really sits on the threshold."""
    n = min(len(res_err), len(want["err"]))
    assert np.abs(np.asarray(res_err)[:n] - want["err"][:n]).max() < TOL_E
    if res_iterations != want["iterations"]:
        assert fp32 and abs(res_iterations - want["iterations"]) == 1, (res_iterations, want["iterations"])
        k = min(res_iterations, want["iterations"]) + 1          # the error whose stop test disagreed
        dE = abs(want["err"][k] - want["err"][k - 1])
        assert abs(dE - tol) < 5e-7 or abs(want["err"][k] - tol) < 5e-7, f"stop rule disagreed away from the threshold: dE={dE}"
    assert rel(res_T, want["T"]) < TOL_T


# ---------------------------------------------------------------------------------------------------
# matching seam
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_matching_synthetic(ctx, pkg, orc, dtype):
    if dtype == np.float64:
        D, M = orc.synth_icp_cpu(32)          # configs[0]: the CPU program's own cloud
    else:
        D = pkg.datasets.synthetic_grid(32, np.float32)
        M = pkg.datasets.make_model_standard(D)
    idx = ctx.Matching(D, M)
    assert idx.dtype == np.int32 and np.array_equal(idx, orc.nn(D, M))


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("n,m", [(1, 1), (1, 5), (3, 17), (1000, 255), (1025, 257), (2049, 4097), (777, 16), (5000, 3000)])
def test_matching_ragged_sizes(ctx, orc, dtype, n, m):
    rng = np.random.default_rng(n * 7919 + m)
    P = rng.standard_normal((n, 3)).astype(dtype)
    Q = rng.standard_normal((m, 3)).astype(dtype)
    idx = ctx.Matching(P, Q)
    assert idx.min() >= 0 and idx.max() < m
    assert np.array_equal(idx, orc.nn(P, Q))


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_matching_ties_resolve_to_lowest_index(ctx, orc, dtype):
    # integer lattice + duplicated model points: equal distances everywhere
    g = np.stack(np.meshgrid(np.arange(12.0), np.arange(12.0), np.arange(5.0), indexing="ij"), -1).reshape(-1, 3)
    Q = np.concatenate([g, g[::3], g[5:40]]).astype(dtype)       # exact duplicates at higher indices
    P = np.concatenate([g[::2] + 0.5, g[1::5]]).astype(dtype)    # cell centres (8-way ties) and exact hits
    idx = ctx.Matching(P, Q)
    assert np.array_equal(idx, orc.nn(P, Q))
    # every model point is a copy of one point
    Q1 = np.tile(np.array([[0.25, -1.5, 3.0]], dtype=dtype), (1000, 1))
    assert (ctx.Matching(P[:300], Q1) == 0).all()


def test_matching_hall_and_bunny(ctx, pkg, orc, golden):
    P, Q = orc.hall_clouds(golden)     # configs[2]: 16384 pts, 4361 coincident points at the origin
    idx = ctx.Matching(P, Q)
    assert np.array_equal(idx, orc.nn(P, Q))
    B = np.fromfile(os.path.join(golden, "bunny_res_xyz_f32.bin"), dtype=np.float32).reshape(-1, 3)
    BM = pkg.datasets.make_model_gpu(B, *pkg.datasets.BUNNY)
    assert np.array_equal(ctx.Matching(B, BM), orc.nn(B, BM))


def test_matching_edge_cases(ctx, pkg):
    P = np.zeros((0, 3), dtype=np.float32)
    Q = np.ones((4, 3), dtype=np.float32)
    assert ctx.Matching(P, Q).shape == (0,)                    # empty moving cloud: nothing to do
    with pytest.raises(pkg.IcpError) as e:
        ctx.Matching(Q, P)                                     # empty model: loud
    assert e.value.code == pkg.capi.ICP_ERR_EMPTY
    with pytest.raises(pkg.IcpError):
        ctx.point_to_point(Q, P)
    with pytest.raises(pkg.IcpError):
        ctx.point_to_point(Q, Q, max_iter=0)


def test_matching_full_size_properties(ctx, pkg, golden):
    """BASELINE configs[1] size (Bunny.csv, 35 947 pts): size-independent properties, no oracle."""
    B = np.fromfile(os.path.join(golden, "bunny_xyz_f32.bin"), dtype=np.float32).reshape(-1, 3)
    M = pkg.datasets.make_model_gpu(B, *pkg.datasets.BUNNY)
    idx = ctx.Matching(B, M)
    assert idx.min() >= 0 and idx.max() < M.shape[0]

    def d2(P, Q, j):
        d = Q[j] - P
        d = d * d
        return (d[:, 0] + d[:, 1]) + d[:, 2]

    best = d2(B, M, idx)
    rng = np.random.default_rng(0)
    for _ in range(8):                                   # nothing sampled beats the reported minimum
        j = rng.integers(0, M.shape[0], size=B.shape[0])
        assert (best <= d2(B, M, j)).all()
    # self-match: every point finds itself (or an exact lower-index duplicate)
    s = ctx.Matching(M, M)
    assert (s <= np.arange(M.shape[0])).all() and np.array_equal(M[s], M)
    # sharding the moving axis does not change any answer (what the multi-GPU split relies on)
    a, b = 12345, 30001
    assert np.array_equal(ctx.Matching(B[a:b], M), idx[a:b])
    # permuting the model permutes the answers consistently (distances identical, ties aside)
    perm = rng.permutation(M.shape[0])
    idx2 = ctx.Matching(B, M[perm])
    assert np.array_equal(d2(B, M[perm], idx2), best)


@pytest.mark.parametrize("form", ["default", "armed_shared", "armed_then_resident", "resident_second_registration", "waves16"])
def test_bunny_registration_full_size_against_the_oracle(pkg, orc, golden, monkeypatch, form):
    """BASELINE configs[1] at full size (Bunny.csv against its moved copy, 35 947^2): nine fixed iterations through the loop the
    plan picks for this size -- rows of 128 as 8-wave blocks, one armed launch per pass, the spare blocks of every launch
    dealt to the heavy rows -- through the resident forms of the same (taking over after four passes; from the first pass of a
    second registration) and through the 16-wave blocks of rounds 1-2: correspondences bit-exact against the oracle's run
    (all threads of the host), transform within the tolerance"""
    # (default, round 4: a context's first registration is armed for its cold pass and one resident kernel from the second pass on)
    env = {"default": {}, "armed_shared": {"ICP_SHARE_AUTO": "0"}, "armed_then_resident": {"ICP_SHARE_RESIDENT_AFTER": "4"},
           "resident_second_registration": {"ICP_RESIDENT": "2"}, "waves16": {"ICP_NN_WAVES128": "16"}}[form]
    for k in ("ICP_NN_SHARE", "ICP_RESIDENT", "ICP_ARMED", "ICP_NN_WAVES128", "ICP_NN_SHARE_RESIDENT", "ICP_SHARE_RESIDENT_AFTER", "ICP_SHARE_AUTO"):
        monkeypatch.delenv(k, raising=False)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    B = np.fromfile(os.path.join(golden, "bunny_xyz_f32.bin"), dtype=np.float32).reshape(-1, 3)
    M = pkg.datasets.make_model_gpu(B, *pkg.datasets.BUNNY)
    orc.set_threads(os.cpu_count() or 1)
    try:
        want = orc.icp_p2p_f32x(B, M, 9, 0.0, fixed=True)
    finally:
        orc.set_threads(1)
    with pkg.Context(0) as c:
        for _ in range(2 if form == "resident_second_registration" else 1):
            res = c.point_to_point(B, M, max_iter=9, tol=0.0, fixed_iterations=True)
        assert c.nn_launch_info()["threads"] == (1024 if form == "waves16" else 512)
    assert res.iterations == want["iterations"] == 9
    assert np.array_equal(res.idx, want["idx"])
    assert rel(res.T, want["T"]) < TOL_T
    assert np.allclose(res.err[1:], want["err"][1:], rtol=1e-6)


def test_bunny_point_to_plane_forms_are_bit_identical(pkg, golden, monkeypatch):
    """configs[1]'s cloud through the point-to-plane loop (6 x 6 rows; 8-wave blocks with shared rows, 16-wave blocks, a resident
    kernel sharing its rows in a second registration, one launch per pass): the same iterations, transform, error series
    and correspondences"""
    B = np.fromfile(os.path.join(golden, "bunny_xyz_f32.bin"), dtype=np.float32).reshape(-1, 3)
    M = pkg.datasets.make_model_gpu(B, *pkg.datasets.BUNNY)
    res = {}
    for name, env in (("shared", {}), ("waves16", {"ICP_NN_WAVES128": "16"}), ("resident", {"ICP_RESIDENT": "2"}), ("stepwise", {"ICP_RESIDENT": "0", "ICP_ARMED": "0"})):
        for k in ("ICP_NN_SHARE", "ICP_RESIDENT", "ICP_ARMED", "ICP_NN_WAVES128", "ICP_NN_SHARE_RESIDENT", "ICP_SHARE_RESIDENT_AFTER"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        with pkg.Context(0) as c:
            for _ in range(2 if name == "resident" else 1):
                res[name] = c.point_to_plane(B, M, max_iter=30, tol=1e-6)
    ref = res["stepwise"]
    assert ref.iterations >= 3
    for name, r in res.items():
        assert r.iterations == ref.iterations and np.array_equal(r.T, ref.T) and np.array_equal(r.err, ref.err) and np.array_equal(r.idx, ref.idx), name


# ---------------------------------------------------------------------------------------------------
# the loop, pass by pass, from captured inputs
# ---------------------------------------------------------------------------------------------------
def _stepwise(ctx, pkg, orc, D, M, max_iter, tol, metric=None, fixed=False, check_every=1):
    ctx.set_model(M)
    ctx.set_moving(D)
    ctx.loop_begin(metric if metric is not None else pkg.ICP_POINT_TO_POINT, max_iter=max_iter, tol=tol, fixed_iterations=fixed)
    k = 0
    while True:
        ctx.loop_enqueue()          # [transform of pass k-1] + matching/moments of pass k
        if ctx.loop_complete():
            break
        if k % check_every == 0:
            Pk = ctx.get_moving()   # the cloud pass k was matched on, as the device holds it
            assert np.array_equal(ctx.get_indices(), orc.nn(Pk, M)), f"pass {k}: indices differ from the oracle on the same input"
        k += 1
    return ctx.loop_state()


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_loop_indices_bit_exact_every_pass(ctx, pkg, orc, dtype):
    if dtype == np.float64:
        D, M = orc.synth_icp_cpu(32)
        st = _stepwise(ctx, pkg, orc, D, M, 200, 1e-5)
        want = orc.icp_p2p(D, M, 200, 1e-5)
    else:
        D = pkg.datasets.synthetic_grid(32, np.float32)
        M = pkg.datasets.make_model_gpu(D, *pkg.datasets.P2P_GPU)
        st = _stepwise(ctx, pkg, orc, D, M, 40, 1e-6)
        want = orc.icp_p2p_f32x(D, M, 40, 1e-6)
    assert_same_run(st["iterations"], st["err"], st["T"], want, 1e-5 if dtype == np.float64 else 1e-6, fp32=(dtype == np.float32))


def test_single_pass_transform_and_error(ctx, pkg, orc):
    """one pass from identical inputs: R,t / moved cloud / E against the oracle's statements"""
    D, M = orc.synth_icp_cpu(24)
    res = ctx.point_to_point(D, M, max_iter=1, tol=1e-5)
    idx = orc.nn(D, M)
    R, t, _ = orc.p2p_minimize(D, M, idx)
    moved = orc.transform(D, R, t)
    assert res.passes == 1 and np.array_equal(res.idx, idx)
    assert rel(res.T[:3, :3], R) < 1e-12 and rel(res.T[:3, 3], t) < 1e-12
    assert rel(res.moved, moved) < 1e-11
    assert abs(res.err[1] - orc.rms_error(moved, M, idx)) < 1e-12


# ---------------------------------------------------------------------------------------------------
# full runs (the reference programs' configurations)
# ---------------------------------------------------------------------------------------------------
def test_icp_cpu_config_f64(ctx, orc):
    """configs[0]: src/ICP_CPU.c (fp64, tol 1e-5, MAX_ITER 200) at WIDTH 32 -- and the oracle itself
    reproduces the reference's recorded 56 iterations for this input (tests/test_oracle.py)."""
    D, M = orc.synth_icp_cpu(32)
    res = ctx.point_to_point(D, M, max_iter=200, tol=1e-5)
    want = orc.icp_p2p(D, M, 200, 1e-5)
    assert res.iterations == want["iterations"] == 56
    assert rel(res.T, want["T"]) < TOL_T
    assert np.abs(res.err - want["err"]).max() < TOL_E
    assert rel(res.moved, want["moved"]) < TOL_T


def test_icp_standard_config(ctx, pkg, orc):
    D = pkg.datasets.synthetic_grid(32, np.float32)
    M = pkg.datasets.make_model_standard(D)
    res = ctx.point_to_point(D, M, max_iter=40, tol=0.0, fixed_iterations=True)
    want = orc.icp_p2p_f32x(D, M, 40, 0.0, fixed=True)
    assert_close_to_fp32_twin(res, orc.icp_p2p(D, M, 40, 0.0, fixed=True))
    assert res.passes == want["passes"] == 40 and res.iterations == 40
    assert rel(res.T, want["T"]) < TOL_T
    assert np.abs(res.err - want["err"]).max() < TOL_E


def test_icp_point_to_point_128(ctx, pkg, orc):
    D = pkg.datasets.synthetic_grid(128, np.float32)          # src/ICP_point_to_point.cu as shipped
    M = pkg.datasets.make_model_gpu(D, *pkg.datasets.P2P_GPU)
    res = ctx.point_to_point(D, M, max_iter=40, tol=1e-6)
    want = orc.icp_p2p_f32x(D, M, 40, 1e-6)
    assert_close_to_fp32_twin(res, orc.icp_p2p(D, M, 40, 1e-6))
    assert_same_run(res.iterations, res.err, res.T, want, 1e-6, fp32=True)


def test_icp_hall(ctx, pkg, orc, golden):
    """configs[2]: hall LiDAR scan, point-to-point, fp32.  Clouds built by the product path (device
    polar->Cartesian) must agree with the oracle's; the run must recover the baked-in motion."""
    r = np.fromfile(os.path.join(golden, "hall_ranges_u32.bin"), dtype=np.uint32)
    alt, az = pkg.datasets.read_os1_intrinsics(os.path.join(golden, "beam_intrinsics.csv"))
    P, Q = pkg.datasets.hall_clouds(ctx, r, 33616, alt, az)
    Po, Qo = orc.hall_clouds(golden)
    assert np.abs(P - Po).max() < 2e-5 * np.abs(Po).max() and np.abs(Q - Qo).max() < 2e-5 * np.abs(Qo).max()
    # parity on the oracle-built clouds (identical inputs on both sides)
    res = ctx.point_to_point(Po, Qo, max_iter=100, tol=1e-6)
    want = orc.icp_p2p_f32x(Po, Qo, 100, 1e-6)
    assert_close_to_fp32_twin(res, orc.icp_p2p(Po, Qo, 100, 1e-6))
    assert_same_run(res.iterations, res.err, res.T, want, 1e-6, fp32=True)
    # ground truth baked into the dataset (mm translation scaled to metres): sanity bound
    ang, t_mm = pkg.datasets.HALL_MM
    assert np.abs(res.T[:3, 3] - np.array(t_mm) / 1000.0).max() < 5e-3
    assert abs(res.T[1, 0] - np.sin(ang[2])) < 5e-3


def test_icp_hall_against_the_fp64_cpu_path(ctx, pkg, orc, golden):
    """north_star's literal comparator on the headline cloud: "outputs must match the reference CPU path in
    src/ICP_CPU.c" (:217-271 -- fp64, tol 1e-5, MAX_ITER 200).  (a) the hall pair, widened, through ICP_F64: indices of
    every pass from the captured P_k, iteration count, error series, composed transform; (b) the fp32 run of the same
    pair -- the configuration the metric is quoted on -- against that same fp64 oracle run."""
    P, Q = orc.hall_clouds(golden)
    P64, Q64 = P.astype(np.float64), Q.astype(np.float64)
    want = orc.icp_p2p(P64, Q64, 200, 1e-5)
    st = _stepwise(ctx, pkg, orc, P64, Q64, 200, 1e-5)            # asserts idx_k == oracle(P_k) for every pass k
    assert st["iterations"] == want["iterations"]
    assert np.abs(st["err"] - want["err"]).max() < TOL_E and rel(st["T"], want["T"]) < TOL_T
    res = ctx.point_to_point(P64, Q64, max_iter=200, tol=1e-5)
    assert res.iterations == want["iterations"] and np.array_equal(res.idx, want["idx"])
    assert rel(res.T, want["T"]) < TOL_T and rel(res.moved, want["moved"]) < TOL_T
    # (b) fp32 arithmetic on the device, fp64 CPU path as the judge
    r32 = ctx.point_to_point(P, Q, max_iter=200, tol=1e-5)
    assert r32.iterations == want["iterations"]
    assert rel(r32.T, want["T"]) < TOL_T                          # measured: 7e-8
    assert np.abs(r32.err - want["err"]).max() < TOL_E            # measured: 8e-7
    assert np.array_equal(r32.idx, want["idx"])                   # the final correspondences agree point for point


@pytest.mark.parametrize("sparse", ["1", "0"])
def test_fp64_forms_sparse_and_dense(pkg, orc, golden, monkeypatch, sparse):
    """ICP_F64 runs on the sparse structure (nn_match_row64_f64: chunk boxes in double, one launch per pass) where the cloud
    fits, or (ICP_F64_SPARSE=0, larger clouds) on the dense thread-per-point kernel: indices bit-exact against the fp64
    oracle either way -- ragged sizes, lattice ties, duplicated points, the CPU program's own configuration"""
    monkeypatch.setenv("ICP_F64_SPARSE", sparse)
    rng = np.random.default_rng(99)
    with pkg.Context(0) as c:
        for n, m in [(1, 1), (3, 17), (1000, 255), (1025, 257), (2049, 4097), (777, 16), (5000, 3000)]:
            P = rng.standard_normal((n, 3)); Q = rng.standard_normal((m, 3))
            assert np.array_equal(c.Matching(P, Q), orc.nn(P, Q)), (n, m)
        g = np.stack(np.meshgrid(np.arange(12.0), np.arange(12.0), np.arange(5.0), indexing="ij"), -1).reshape(-1, 3)
        Q = np.concatenate([g, g[::3], g[5:40]]); P = np.concatenate([g[::2] + 0.5, g[1::5]])
        assert np.array_equal(c.Matching(P, Q), orc.nn(P, Q))               # 8-way ties, exact duplicates at higher indices
        Ph, Qh = orc.hall_clouds(golden)
        Ph, Qh = Ph.astype(np.float64), Qh.astype(np.float64)
        assert np.array_equal(c.Matching(Ph, Qh), orc.nn(Ph, Qh))            # 4361 coincident model points
        D, M = orc.synth_icp_cpu(32)
        res = c.point_to_point(D, M, max_iter=200, tol=1e-5)
        want = orc.icp_p2p(D, M, 200, 1e-5)
        assert res.iterations == want["iterations"] == 56 and np.array_equal(res.idx, want["idx"])
        assert rel(res.T, want["T"]) < 1e-9 and np.abs(res.err - want["err"]).max() < 1e-9
        info = c.nn_launch_info()
        assert info["threads"] == (512 if sparse == "1" else 256)


def test_fp64_loop_forms_are_bit_identical(pkg, orc, golden, monkeypatch):
    """ICP_F64: one resident kernel per registration (its (R, t) arrive as two-cache-line messages of twelve doubles), the
    step-wise loop (one launch per pass), each with its mailbox in BAR memory or in host memory, and the dense kernel's
    three-launch form: the same bits -- and the fp64 oracle's run"""
    D, M = orc.synth_icp_cpu(48)
    Ph, Qh = orc.hall_clouds(golden)
    Ph, Qh = Ph.astype(np.float64), Qh.astype(np.float64)
    forms = {"resident": {}, "resident_host_mailbox": {"ICP_MAILBOX": "host"}, "resident_plain_stores": {"ICP_MAILBOX": "plain"},
             "stepwise": {"ICP_RESIDENT": "0"}}
    def run(c):
        a = c.point_to_point(D, M, max_iter=200, tol=1e-5)
        b = c.point_to_point(Ph, Qh, max_iter=200, tol=1e-5)
        return a, b
    res = {name: _run_form(pkg, monkeypatch, env, run) for name, env in forms.items()}
    ref = res["stepwise"]
    for name, r in res.items():
        for x, y in zip(r, ref):
            assert x.iterations == y.iterations, name
            assert np.array_equal(x.T, y.T) and np.array_equal(x.err, y.err) and np.array_equal(x.idx, y.idx) and np.array_equal(x.moved, y.moved), name
    monkeypatch.setenv("ICP_F64_SPARSE", "0")
    dense = _run_form(pkg, monkeypatch, {}, run)
    monkeypatch.delenv("ICP_F64_SPARSE")
    for x, y, (A, B) in zip(dense, ref, ((D, M), (Ph, Qh))):
        want = orc.icp_p2p(A, B, 200, 1e-5)
        for r in (x, y):
            assert r.iterations == want["iterations"] and np.array_equal(r.idx, want["idx"])
            assert rel(r.T, want["T"]) < 1e-9 and np.abs(r.err - want["err"]).max() < 1e-9


def test_icp_bunny(ctx, pkg, orc, golden):
    B = np.fromfile(os.path.join(golden, "bunny_res_xyz_f32.bin"), dtype=np.float32).reshape(-1, 3)
    M = pkg.datasets.make_model_gpu(B, *pkg.datasets.BUNNY)
    res = ctx.point_to_point(B, M, max_iter=100, tol=1e-6)
    want = orc.icp_p2p_f32x(B, M, 100, 1e-6)
    assert_close_to_fp32_twin(res, orc.icp_p2p(B, M, 100, 1e-6))
    assert_same_run(res.iterations, res.err, res.T, want, 1e-6, fp32=True)


def test_loop_is_reentrant_and_deterministic(ctx, pkg):
    D = pkg.datasets.synthetic_grid(48, np.float32)
    M = pkg.datasets.make_model_gpu(D, *pkg.datasets.P2P_GPU)
    a = ctx.point_to_point(D, M, max_iter=15, tol=1e-6)
    b = ctx.point_to_point(D, M, max_iter=15, tol=1e-6)
    assert np.array_equal(a.T, b.T) and np.array_equal(a.idx, b.idx) and np.array_equal(a.err, b.err)
    with pkg.Context(0) as other:          # a second context is independent
        c = other.point_to_point(D, M, max_iter=15, tol=1e-6)
    assert np.array_equal(a.T, c.T)


def test_external_moments_buffer_and_stream(ctx, pkg):
    """the multi-GPU driver's plumbing on one device: the loop writes its 32-double vector into a torch
    tensor on torch's stream (device finalize path); results must agree with the single-GPU host-reduce path."""
    import torch
    D = pkg.datasets.synthetic_grid(64, np.float32)
    M = pkg.datasets.make_model_gpu(D, *pkg.datasets.P2P_GPU)
    want = ctx.point_to_point(D, M, max_iter=12, tol=1e-6)
    mom = torch.zeros(pkg.ICP_NMOM, dtype=torch.float64, device="cuda:0")
    torch.cuda.synchronize()
    ctx.set_model(M)
    ctx.set_moving(D)
    side = torch.cuda.Stream()
    ctx.set_stream(side.cuda_stream)
    ctx.loop_set_moments_dev(mom.data_ptr())
    try:
        with torch.cuda.stream(side):
            ctx.loop_begin(pkg.ICP_POINT_TO_POINT, max_iter=12, tol=1e-6)
            seen_counts = []
            while True:
                ctx.loop_enqueue()
                seen_counts.append(float(mom[1].item()))     # ordered behind the finalize kernel on torch's stream
                if ctx.loop_complete():
                    break
            st = ctx.loop_state()
    finally:
        ctx.loop_set_moments_dev(0)
        ctx.set_stream(0)
    assert st["iterations"] == want.iterations
    assert rel(st["T"], want.T) < 1e-12 and np.abs(st["err"] - want.err).max() < 1e-12
    assert seen_counts[0] == D.shape[0]


def test_run_sharded_single_rank_rccl(ctx, pkg, orc):
    """the multi-GPU driver end to end with world_size 1 over the nccl (= RCCL) backend: communicator set-up,
    all-reduce on the loop's stream, device finalize, D2H behind the collective"""
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29541")
    dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        D = pkg.datasets.synthetic_grid(64, np.float32)
        M = pkg.datasets.make_model_gpu(D, *pkg.datasets.P2P_GPU)
        Ps, begin = pkg.distributed.shard(D, 0, 1)
        st = pkg.distributed.run_sharded(ctx, Ps, M, dist, max_iter=40, tol=1e-6)
    finally:
        dist.destroy_process_group()
    want = orc.icp_p2p_f32x(D, M, 40, 1e-6)
    assert begin == 0 and st["iterations"] == want["iterations"]
    assert rel(st["T"], want["T"]) < TOL_T and np.abs(st["err"] - want["err"]).max() < TOL_E
    assert np.array_equal(st["idx"], want["idx"])


def test_run_sharded_native_comm_single_rank(ctx, pkg, orc):
    """library-issued RCCL all-reduce (icp_comm_*), world_size 1: id exchange, communicator, collective on the
    loop's own stream, device finalize"""
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ["MASTER_PORT"] = "29547"
    dist.init_process_group("gloo", rank=0, world_size=1)      # only carries the 128-byte id
    try:
        D = pkg.datasets.synthetic_grid(64, np.float32)
        M = pkg.datasets.make_model_gpu(D, *pkg.datasets.P2P_GPU)
        st = pkg.distributed.run_sharded_native(ctx, D, M, dist, max_iter=40, tol=1e-6)
    finally:
        dist.destroy_process_group()
    want = orc.icp_p2p_f32x(D, M, 40, 1e-6)
    assert st["iterations"] == want["iterations"] and rel(st["T"], want["T"]) < TOL_T
    assert np.array_equal(st["idx"], want["idx"])
    # and the context is back on its single-GPU fast path afterwards
    res = ctx.point_to_point(D, M, max_iter=40, tol=1e-6)
    assert res.iterations == want["iterations"] and rel(res.T, st["T"]) < 1e-12


def test_reset_moving_and_loop_run(ctx, pkg, orc):
    """icp_reset_moving restores the uploaded cloud on the device; icp_loop_run(k) == k x (enqueue + complete)"""
    D = pkg.datasets.synthetic_grid(48, np.float32)
    M = pkg.datasets.make_model_gpu(D, *pkg.datasets.P2P_GPU)
    ctx.set_model(M)
    ctx.set_moving(D)
    runs = []
    for _ in range(2):
        ctx.reset_moving()
        assert np.array_equal(ctx.get_moving(), D)
        ctx.loop_begin(pkg.ICP_POINT_TO_POINT, max_iter=40, tol=1e-6)
        k1, done = ctx.loop_run(3)
        assert k1 == 3 and not done
        k2, done = ctx.loop_run(1000)
        assert done
        st = ctx.loop_state()
        assert st["passes"] == k1 + k2 - 1            # the last step only evaluated the stop rule
        runs.append((st, ctx.get_moving(), ctx.loop_indices()))
    (a, pa, ia), (b, pb, ib) = runs
    assert np.array_equal(a["T"], b["T"]) and np.array_equal(pa, pb) and np.array_equal(ia, ib)
    want = orc.icp_p2p_f32x(D, M, 40, 1e-6)
    assert a["iterations"] == want["iterations"] and rel(a["T"], want["T"]) < TOL_T and np.array_equal(ia, want["idx"])


@pytest.mark.parametrize("tail", ["1", "0"])
def test_fused_and_two_kernel_forms_agree(pkg, orc, golden, tail):
    """ICP_FUSED_TAIL=0/1 (read at context creation): same indices, same transform to the last bits of the sums"""
    import subprocess, sys, json
    code = (
        "import sys, os, json, numpy as np\n"
        f"sys.path.insert(0, {ROOT!r}); sys.path.insert(0, os.path.join({ROOT!r}, 'tests'))\n"
        "from __graft_entry__ import load_package\n"
        "import oracle_lib\n"
        "pkg = load_package(); orc = oracle_lib.Oracle()\n"
        f"P, Q = orc.hall_clouds({golden!r})\n"
        "with pkg.Context(0) as ctx:\n"
        "    r = ctx.point_to_point(P, Q, max_iter=100, tol=1e-6)\n"
        "print(json.dumps(dict(it=r.iterations, T=r.T.tolist(), err=r.err.tolist(), idx=int(np.bitwise_xor.reduce(r.idx * np.arange(1, r.idx.size + 1, dtype=np.int64))))))\n")
    env = dict(os.environ, ICP_FUSED_TAIL=tail)
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    got = json.loads(out.stdout.strip().splitlines()[-1])
    P, Q = orc.hall_clouds(golden)
    want = orc.icp_p2p_f32x(P, Q, 100, 1e-6)
    assert got["it"] == want["iterations"] and rel(np.array(got["T"]), want["T"]) < TOL_T
    assert got["idx"] == int(np.bitwise_xor.reduce(want["idx"] * np.arange(1, want["idx"].size + 1, dtype=np.int64)))


# ---------------------------------------------------------------------------------------------------
# the three forms of the loop (resident kernel, armed launches, one launch per pass) are the same computation
# ---------------------------------------------------------------------------------------------------
def _set_row(monkeypatch, row):
    """ICP_NN_ROW = 64 / 128, and "128w8": rows of 128 as 8-wave blocks, two to a CU, whose launches (one per pass) share the
    rows -- the form clouds of 33-65 k points get by themselves, forced here onto small ones (few rows, hundreds of spare blocks)"""
    monkeypatch.setenv("ICP_NN_ROW", row[:3] if row.startswith("128") else row)
    if row in ("128w8", "128w4"):   # ("128w4": 4-wave blocks, the hierarchical search's form for clouds with rows for several rounds of blocks)
        monkeypatch.setenv("ICP_NN_WAVES128", row[-1])
    else:
        monkeypatch.delenv("ICP_NN_WAVES128", raising=False)


LOOP_FORMS = {
    "resident": {},
    "resident_forced": {"ICP_RESIDENT": "2"},           # (a plan with shared rows runs armed unless told otherwise)
    "resident_no_speculation": {"ICP_NN_SPECULATE": "0"},   # (cached per process: effective only in a run that starts with it)
    "resident_host_mailbox": {"ICP_MAILBOX": "host"},
    "resident_plain_stores": {"ICP_MAILBOX": "plain"},   # the mailbox line written word by word (a CPU without AVX)
    "armed": {"ICP_RESIDENT": "0"},
    "stepwise": {"ICP_RESIDENT": "0", "ICP_ARMED": "0"},
}


def _run_form(pkg, monkeypatch, env, fn):
    for k in ("ICP_MAILBOX", "ICP_RESIDENT", "ICP_ARMED", "ICP_NN_SPECULATE"):
        monkeypatch.delenv(k, raising=False)
    for k, v in env.items():
        monkeypatch.setenv(k, v)           # read by icp_create
    with pkg.Context(0) as c:
        return fn(c)


@pytest.mark.parametrize("row", ["64", "128", "128w8"])
@pytest.mark.parametrize("metric", ["point_to_point", "point_to_plane"])
def test_loop_forms_are_bit_identical(pkg, orc, golden, monkeypatch, metric, row):
    """icp_loop_run keeps one resident kernel for the registration (mailbox in BAR-visible device memory, or in
    pinned host memory relayed by block 0), or arms the next pass ahead of its (R, t), or launches pass by pass:
    same rows, same host half -> the same bits, and the oracle's run."""
    _set_row(monkeypatch, row)     # rows of 64 points (nn_match_row64, the default here) / of 128 (nn_match_sparse, 16 waves / 8 waves + shared rows)
    P, Q = orc.hall_clouds(golden)
    if metric == "point_to_point":
        fn = lambda c: c.point_to_point(P, Q, max_iter=100, tol=1e-6)
    else:
        fn = lambda c: c.point_to_plane(P, Q, max_iter=50, tol=1e-6)
    res = {name: _run_form(pkg, monkeypatch, env, fn) for name, env in LOOP_FORMS.items()}
    ref = res["stepwise"]
    for name, r in res.items():
        assert r.iterations == ref.iterations, name
        assert np.array_equal(r.T, ref.T) and np.array_equal(r.err, ref.err) and np.array_equal(r.idx, ref.idx), name
    if metric == "point_to_point":
        want = orc.icp_p2p_f32x(P, Q, 100, 1e-6)
        assert_same_run(ref.iterations, ref.err, ref.T, want, 1e-6, fp32=True)
        assert np.array_equal(ref.idx, want["idx"]) or ref.iterations != want["iterations"]


@pytest.mark.parametrize("metric", ["point_to_point", "point_to_plane"])
def test_ordered_rows_are_the_same_computation(pkg, orc, golden, monkeypatch, metric):
    """large models are searched through the box hierarchy, and when their rows outnumber the machine's blocks several times
    over, every launch takes the rows heaviest first (sorted by the hits of the launch before: launch_row_order).  Forced
    here onto the hall pair (128 rows): armed launches and launches pass by pass, rows ordered or in index order -- the same
    bits, and the oracle's run"""
    monkeypatch.setenv("ICP_NN_ROW", "128")
    monkeypatch.setenv("ICP_NN_HIER", "1")
    monkeypatch.delenv("ICP_NN_WAVES128", raising=False)
    P, Q = orc.hall_clouds(golden)
    fn = (lambda c: c.point_to_point(P, Q, max_iter=100, tol=1e-6)) if metric == "point_to_point" else (lambda c: c.point_to_plane(P, Q, max_iter=50, tol=1e-6))
    res = {}
    # (split rows: the heaviest rows of an ordered launch are searched by 2 .. 64 blocks each; a target of 16 hits splits most
    # of these, of 0 none)
    for name, env in (("index_order", {"ICP_NN_ORDER": "0", "ICP_RESIDENT": "0"}), ("ordered_armed", {"ICP_NN_ORDER": "2", "ICP_RESIDENT": "0"}),
                      ("ordered_stepwise", {"ICP_NN_ORDER": "2", "ICP_RESIDENT": "0", "ICP_ARMED": "0"}),
                      ("split_armed", {"ICP_NN_ORDER": "2", "ICP_RESIDENT": "0", "ICP_NN_SPLIT_MIN": "16"}),
                      ("split_stepwise", {"ICP_NN_ORDER": "2", "ICP_RESIDENT": "0", "ICP_ARMED": "0", "ICP_NN_SPLIT_MIN": "16"}),
                      ("unsplit_stepwise", {"ICP_NN_ORDER": "2", "ICP_RESIDENT": "0", "ICP_ARMED": "0", "ICP_NN_SPLIT_MIN": "0"}),
                      ("split_armed_8_waves", {"ICP_NN_ORDER": "2", "ICP_RESIDENT": "0", "ICP_NN_SPLIT_MIN": "16", "ICP_NN_WAVES128": "8"}),
                      ("split_stepwise_8_waves", {"ICP_NN_ORDER": "2", "ICP_RESIDENT": "0", "ICP_ARMED": "0", "ICP_NN_SPLIT_MIN": "16", "ICP_NN_WAVES128": "8"}),
                      ("resident_8_waves", {"ICP_NN_WAVES128": "8"}),
                      ("split_armed_4_waves", {"ICP_NN_ORDER": "2", "ICP_RESIDENT": "0", "ICP_NN_SPLIT_MIN": "16", "ICP_NN_WAVES128": "4"}),
                      ("split_stepwise_4_waves", {"ICP_NN_ORDER": "2", "ICP_RESIDENT": "0", "ICP_ARMED": "0", "ICP_NN_SPLIT_MIN": "16", "ICP_NN_WAVES128": "4"}),
                      ("stepwise_4_waves_throughout", {"ICP_NN_ORDER": "2", "ICP_RESIDENT": "0", "ICP_ARMED": "0", "ICP_NN_WAVES128": "4", "ICP_NN_COLD8": "0"}),
                      ("resident_4_waves", {"ICP_NN_WAVES128": "4", "ICP_NN_COLD8": "0"})):
        for k in ("ICP_NN_ORDER", "ICP_NN_SPLIT_MIN", "ICP_NN_WAVES128", "ICP_NN_COLD8"):
            monkeypatch.delenv(k, raising=False)
        res[name] = _run_form(pkg, monkeypatch, env, fn)
    for k in ("ICP_NN_ORDER", "ICP_NN_SPLIT_MIN", "ICP_NN_WAVES128", "ICP_NN_COLD8"):
        monkeypatch.delenv(k, raising=False)
    ref = res["index_order"]
    for name, r in res.items():
        assert r.iterations == ref.iterations and np.array_equal(r.T, ref.T) and np.array_equal(r.err, ref.err) and np.array_equal(r.idx, ref.idx), name
    if metric == "point_to_point":
        want = orc.icp_p2p_f32x(P, Q, 100, 1e-6)
        assert_same_run(ref.iterations, ref.err, ref.T, want, 1e-6, fp32=True)


def test_resident_kernel_resumes_and_fixed_iterations(pkg, orc, golden, monkeypatch):
    """a resident registration cut by max_steps leaves memory as the step-wise kernels do: the next icp_loop_run (a new
    resident kernel, seeded from the index buffer) continues it; fixed_iterations ends with the transform-only pass"""
    P, Q = orc.hall_clouds(golden)
    def cut(c):
        c.set_model(Q); c.set_moving(P)
        c.loop_begin(pkg.ICP_POINT_TO_POINT, max_iter=9, tol=1e-6, fixed_iterations=True)
        steps, done = 0, False
        for chunk in (1, 2, 3, 100):
            k, done = c.loop_run(chunk)
            steps += k
            if done:
                break
        assert done
        return c.loop_state(), c.get_moving(), c.loop_indices(), steps
    whole = _run_form(pkg, monkeypatch, {}, lambda c: c.point_to_point(P, Q, max_iter=9, tol=1e-6, fixed_iterations=True))
    st, moved, idx, steps = _run_form(pkg, monkeypatch, {}, cut)
    assert steps == 10 and st["iterations"] == whole.iterations == 9
    assert np.array_equal(st["T"], whole.T) and np.array_equal(st["err"], whole.err) and np.array_equal(idx, whole.idx)
    want = orc.icp_p2p_f32x(P, Q, 9, 1e-6, fixed=True)
    assert rel(st["T"], want["T"]) < TOL_T and np.array_equal(idx, want["idx"])


def test_moving_cloud_with_shared_rows(pkg, orc, monkeypatch):
    """36 864 moving points are 576 rows of 64 -- more than two blocks per CU hold -- so the plan is rows of 128 as 8-wave blocks
    (288 rows, two blocks to a CU) and one armed launch per pass whose spare blocks go to the heavy rows (shared rows); the same
    run with 16-wave blocks, with the rows unshared, launched pass by pass, and as a resident kernel (sharing its rows or not,
    from the first pass or taking over after a few armed ones): the same bits, and the oracle's run"""
    D = pkg.datasets.synthetic_grid(192, np.float32)
    M = pkg.datasets.make_model_gpu(D[:4096], *pkg.datasets.P2P_GPU)
    want = orc.icp_p2p_f32x(D, M, 6, 1e-6)
    forms = {"shared": {}, "unshared": {"ICP_NN_SHARE": "0"}, "stepwise": {"ICP_RESIDENT": "0", "ICP_ARMED": "0"}, "waves16": {"ICP_NN_WAVES128": "16"},
             # a resident kernel that shares its rows (roles fixed for the launch, the matches of a split row published by whichever
             # block closes it): from the first pass -- in a second registration, which has the first one's counts to go by --
             # without sharing, and taking over from the armed launches after two / four passes
             "resident": {"ICP_RESIDENT": "2"}, "resident_unshared": {"ICP_RESIDENT": "2", "ICP_NN_SHARE_RESIDENT": "0"},
             "armed_then_resident_2": {"ICP_SHARE_RESIDENT_AFTER": "2"}, "armed_then_resident_4": {"ICP_SHARE_RESIDENT_AFTER": "4"}}
    res = {}
    for name, env in forms.items():
        for k in ("ICP_NN_SHARE", "ICP_RESIDENT", "ICP_ARMED", "ICP_NN_WAVES128", "ICP_NN_SHARE_RESIDENT", "ICP_SHARE_RESIDENT_AFTER"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        with pkg.Context(0) as c:
            for _ in range(2 if name == "resident" else 1):
                res[name] = c.point_to_point(D, M, max_iter=6, tol=1e-6)
            assert c.nn_launch_info()["threads"] == (1024 if name == "waves16" else 512), name
    ref = res["shared"]
    assert ref.iterations == want["iterations"] and np.array_equal(ref.idx, want["idx"])
    assert rel(ref.T, want["T"]) < TOL_T
    for name, r in res.items():
        assert r.iterations == ref.iterations and np.array_equal(r.T, ref.T) and np.array_equal(r.idx, ref.idx), name
        assert np.array_equal(r.moved, ref.moved), name   # (the moved cloud as the device holds it after the loop)
        # (the loop ends at max_iter: the last entry is the error-only pass -- a TRANSFORM_ONLY message to the resident kernel, whose
        # compact rows carry their tag in the low 16 mantissa bits of the error share, or the stand-alone transform kernel)
        assert np.array_equal(r.err[:-1], ref.err[:-1]) and abs(r.err[-1] - ref.err[-1]) <= 1e-10 * ref.err[-1], name


@pytest.mark.parametrize("width", [176, 181])
def test_largest_clouds_that_stay_resident(ctx, pkg, orc, width):
    """30 976 and 32 761 moving points: 484 and 512 rows of 64, two blocks on (almost) every CU for the whole registration"""
    D = pkg.datasets.synthetic_grid(width, np.float32)
    M = pkg.datasets.make_model_gpu(D[:4096], *pkg.datasets.P2P_GPU)
    res = ctx.point_to_point(D, M, max_iter=4, tol=1e-6)
    want = orc.icp_p2p_f32x(D, M, 4, 1e-6)
    assert res.iterations == want["iterations"] and np.array_equal(res.idx, want["idx"])
    assert rel(res.T, want["T"]) < TOL_T
    assert ctx.nn_launch_info()["threads"] == 512


def test_mid_size_model_is_searched_by_what_the_cloud_asks_for(pkg, orc):
    """a model of 2^16 points is searched flat by a small cloud and through the box hierarchy by one of more rows than the shared
    8-wave blocks serve (65 536 points: 512 rows) -- the model is set BEFORE the cloud is known and must carry the upper levels and
    the records either way (round 3 built them by a one-row plan: the second case failed at its first pass).  Both against the oracle."""
    D = pkg.datasets.synthetic_grid(256, np.float32)
    M = pkg.datasets.make_model_gpu(D, *pkg.datasets.P2P_GPU)
    orc.set_threads(os.cpu_count() or 1)
    try:
        want_big = orc.icp_p2p_f32x(D, M, 2, 0.0, fixed=True)
        small = np.ascontiguousarray(D[:: 16])
        want_small = orc.icp_p2p_f32x(small, M, 2, 0.0, fixed=True)
    finally:
        orc.set_threads(1)
    with pkg.Context(0) as c:
        c.set_model(M)
        for cloud, want, threads in ((small, want_small, None), (D, want_big, 256), (small, want_small, None)):   # (and back again)
            c.set_moving(cloud)
            c.loop_begin(pkg.ICP_POINT_TO_POINT, max_iter=2, tol=0.0, fixed_iterations=True)
            done = False
            while not done:
                _, done = c.loop_run(1 << 20)
            st, idx = c.loop_state(), c.loop_indices()
            assert st["iterations"] == want["iterations"] == 2
            assert np.array_equal(idx, want["idx"])
            assert rel(st["T"], want["T"]) < TOL_T
            if threads is not None:
                assert c.nn_launch_info()["threads"] == threads        # 4-wave blocks: the hierarchical, ordered form


@pytest.mark.parametrize("n,m", [(32768, 65536), (32769, 65536), (57344, 65536), (57345, 65536), (65536, 65535), (70000, 65537),
                                 (57345, 131071), (70000, 131072), (16384, 131073), (33000, 524288)])
def test_borders_of_the_plans_size_classes_against_the_dense_kernel(pkg, monkeypatch, n, m):
    """(cloud, model) sizes on both sides of the borders where the plan changes form -- rows of 64 / 128, 8-wave shared rows / 16-wave
    blocks, flat / hierarchical search by the cloud's rows AND by the model's size, 16-bit / 32-bit hit lists --: two fixed iterations
    through the form the library picks against the dense kernel of a context created under ICP_NN_SPARSE=0, which executes every pair:
    the same correspondences, the same transform (tools/size_sweep.py runs 102 such pairs; round 4 found one border unguarded)."""
    G = pkg.datasets.synthetic_grid(725, np.float32)
    M = pkg.datasets.make_model_gpu(np.ascontiguousarray(G[:m]), *pkg.datasets.P2P_GPU)
    P = np.ascontiguousarray(G[np.sort(np.random.default_rng(n + m).choice(len(G), n, replace=False))])
    got = {}
    for dense in (False, True):
        monkeypatch.delenv("ICP_NN_SPARSE", raising=False)
        if dense:
            monkeypatch.setenv("ICP_NN_SPARSE", "0")
        with pkg.Context(0) as c:
            c.set_model(M); c.set_moving(P)
            c.loop_begin(pkg.ICP_POINT_TO_POINT, max_iter=2, tol=0.0, fixed_iterations=True)
            done = False
            while not done:
                _, done = c.loop_run(1 << 20)
            got[dense] = (c.loop_state(), c.loop_indices())
    monkeypatch.delenv("ICP_NN_SPARSE", raising=False)
    (sa, ia), (sb, ib) = got[False], got[True]
    assert sa["iterations"] == sb["iterations"] == 2
    assert np.array_equal(ia, ib)
    assert np.allclose(sa["T"], sb["T"], rtol=0, atol=1e-5)


@pytest.mark.parametrize("form,n,m", [("plane", 33000, 65536), ("plane", 57345, 65537), ("plane", 70000, 131072),
                                      ("f64", 16384, 65536), ("f64", 16385, 131071), ("f64", 33000, 4096)])
def test_borders_of_the_size_classes_point_to_plane_and_fp64(pkg, monkeypatch, form, n, m):
    """the same for the point-to-plane loop (normals estimated on the device; the 28-sum row tail of every kernel family) and for the
    fp64 path (rows of 64 on the sparse structure up to 16 384 points and models below 2^17, the dense thread-per-point kernel beyond)"""
    dt = np.float64 if form == "f64" else np.float32
    G = pkg.datasets.synthetic_grid(725, np.float32)
    M = pkg.datasets.make_model_gpu(np.ascontiguousarray(G[:m]), *pkg.datasets.P2P_GPU).astype(dt)
    P = np.ascontiguousarray(G[np.sort(np.random.default_rng(n + m).choice(len(G), n, replace=False))]).astype(dt)
    metric = pkg.ICP_POINT_TO_PLANE if form == "plane" else pkg.ICP_POINT_TO_POINT
    got = {}
    for dense in (False, True):
        monkeypatch.delenv("ICP_NN_SPARSE", raising=False)
        if dense:
            monkeypatch.setenv("ICP_NN_SPARSE", "0")
        with pkg.Context(0) as c:
            c.set_model(M); c.set_moving(P)
            if form == "plane":
                c.estimate_normals()
            c.loop_begin(metric, max_iter=2, tol=0.0, fixed_iterations=True)
            done = False
            while not done:
                _, done = c.loop_run(1 << 20)
            got[dense] = (c.loop_state(), c.loop_indices())
    monkeypatch.delenv("ICP_NN_SPARSE", raising=False)
    (sa, ia), (sb, ib) = got[False], got[True]
    assert sa["iterations"] == sb["iterations"] == 2
    assert np.array_equal(ia, ib)
    assert np.allclose(sa["T"], sb["T"], rtol=0, atol=1e-5)


def _two_ranks_on_one_device(pkg, golden, metric, dtype, env=None, max_iter=100, tol=1e-6):
    """two processes on cuda:0, each with a shard of the hall scan, meeting once per iteration in shared host memory
    (icp_comm_init_local); returns what each rank ended with"""
    import subprocess, sys, json
    code = (
        "import sys, os, json, numpy as np\n"
        f"sys.path.insert(0, {ROOT!r}); sys.path.insert(0, os.path.join({ROOT!r}, 'tests'))\n"
        "from __graft_entry__ import load_package\n"
        "import oracle_lib\n"
        "pkg = load_package(); orc = oracle_lib.Oracle()\n"
        "rank, world, idh, metric, dt = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], int(sys.argv[4]), np.dtype(sys.argv[5])\n"
        f"P, Q = orc.hall_clouds({golden!r})\n"
        "P, Q = P.astype(dt), Q.astype(dt)\n"
        "Ps, begin = pkg.distributed.shard(P, rank, world)\n"
        "with pkg.Context(0) as ctx:\n"
        "    ctx.set_model(Q); ctx.set_moving(Ps)\n"
        "    nrm = ctx.estimate_normals() if metric == pkg.ICP_POINT_TO_PLANE else None\n"
        "    ctx.comm_init_local(bytes.fromhex(idh), rank, world)\n"
        f"    ctx.loop_begin(metric, max_iter={max_iter}, tol={tol})\n"
        "    done = False\n"
        "    while not done:\n"
        "        _, done = ctx.loop_run(1 << 20)\n"
        "    st = ctx.loop_state(); idx = ctx.loop_indices()\n"
        "    ctx.comm_destroy()\n"
        "print(json.dumps(dict(it=st['iterations'], T=st['T'].tolist(), err=st['err'].tolist(), begin=int(begin), idx=idx.tolist(),\n"
        "                      nrm=None if nrm is None else nrm.tolist())))\n")
    idh = pkg.Context.comm_random_id().hex()
    procs = [subprocess.Popen([sys.executable, "-c", code, str(r), "2", idh, str(metric), np.dtype(dtype).name], stdout=subprocess.PIPE,
                              stderr=subprocess.PIPE, text=True, env=dict(os.environ, **(env or {}))) for r in range(2)]
    outs = [p.communicate(timeout=300) for p in procs]
    for p, (o, e) in zip(procs, outs):
        assert p.returncode == 0, e[-2000:]
    got = [json.loads(o.strip().splitlines()[-1]) for o, _ in outs]
    assert got[0]["it"] == got[1]["it"] and got[0]["T"] == got[1]["T"] and got[0]["err"] == got[1]["err"]   # both ranks: the same bits
    return got


@pytest.mark.parametrize("resident", [True, False])
def test_two_ranks_one_node_local_communicator(pkg, orc, golden, resident):
    """two ranks with a shard of the hall scan each, both on cuda:0: both end with the same bits, and with the run of one rank
    holding the whole cloud up to the association of the fp64 sums.  Ranks that share a DEVICE run one launch per pass by
    default (two resident kernels need not fit the machine together: the circular wait of icp_api.cpp, can_reside);
    ICP_DEBUG=shared_resident -- two hall-sized shards do fit -- keeps each rank's resident kernel, the form two ranks on
    two devices run"""
    got = _two_ranks_on_one_device(pkg, golden, pkg.ICP_POINT_TO_POINT, np.float32, {"ICP_DEBUG": "shared_resident"} if resident else None)
    P, Q = orc.hall_clouds(golden)
    want = orc.icp_p2p_f32x(P, Q, 100, 1e-6)
    assert_same_run(got[0]["it"], np.array(got[0]["err"]), np.array(got[0]["T"]), want, 1e-6, fp32=True)
    if got[0]["it"] == want["iterations"]:
        assert np.array_equal(np.concatenate([got[0]["idx"], got[1]["idx"]]), want["idx"])


def test_two_ranks_fp64_against_the_cpu_path(pkg, orc, golden):
    """ICP_F64 sharded over two ranks (src/ICP_CPU.c:217-271 is the comparator: tol 1e-5, MAX_ITER 200).  Alone, each rank's
    fp64 kernel would be resident with one 16-wave block on every CU it uses -- two of those on one device is the case
    the advisor named: by default neither resides, and the registration completes"""
    got = _two_ranks_on_one_device(pkg, golden, pkg.ICP_POINT_TO_POINT, np.float64, max_iter=200, tol=1e-5)
    P, Q = orc.hall_clouds(golden)
    want = orc.icp_p2p(P.astype(np.float64), Q.astype(np.float64), 200, 1e-5)
    assert got[0]["it"] == want["iterations"]
    assert np.abs(np.array(got[0]["err"]) - want["err"]).max() < 1e-9
    assert rel(np.array(got[0]["T"]), want["T"]) < 1e-9
    assert np.array_equal(np.concatenate([got[0]["idx"], got[1]["idx"]]), want["idx"])


def test_two_ranks_point_to_plane(pkg, orc, golden):
    """the 28-sum point-to-plane vector through the node communicator (src/ICP_point_to_plane.cu:517-631), two ranks, against
    the oracle run on the whole cloud with the normals the device estimated"""
    got = _two_ranks_on_one_device(pkg, golden, pkg.ICP_POINT_TO_PLANE, np.float32, max_iter=50, tol=1e-6)
    P, Q = orc.hall_clouds(golden)
    assert got[0]["nrm"] == got[1]["nrm"]
    nrm = np.array(got[0]["nrm"], dtype=np.float32)
    want = orc.icp_p2plane_f32x(P, Q, nrm, 50, 1e-6)
    n = min(len(got[0]["err"]), len(want["err"]))
    assert np.abs(np.array(got[0]["err"])[:n] - want["err"][:n]).max() < TOL_E
    if got[0]["it"] == want["iterations"]:
        assert rel(np.array(got[0]["T"]), want["T"]) < TOL_T
        assert np.array_equal(np.concatenate([got[0]["idx"], got[1]["idx"]]), want["idx"])
    else:
        k = min(got[0]["it"], want["iterations"]) + 1
        dE = abs(want["err"][k] - want["err"][k - 1])
        assert abs(got[0]["it"] - want["iterations"]) == 1 and (abs(dE - 1e-6) < 5e-7 or abs(want["err"][k] - 1e-6) < 5e-7)


def test_rows_added_on_the_device_give_the_same_registration(pkg, orc, monkeypatch):
    """a cloud of more than ICP_HOST_ROWS_MAX rows (round 4: 1 024 by default, 131 072 points) has its moment rows added on the
    device, INSIDE the matching launch -- whoever closes the last row of a range adds the range in index order, whoever closes the
    last range adds the ranges, the vector and the pass's tag land in pinned memory -- instead of by the host as the rows arrive:
    another association of the same fp64 sums.  Forced here on a 36 864-point grid (288 rows, limit 64: armed launches with shared
    rows, every range closed by whichever block happens to be last)"""
    D = pkg.datasets.synthetic_grid(192, np.float32)
    M = pkg.datasets.make_model_gpu(D, *pkg.datasets.P2P_GPU)
    with pkg.Context(0) as c:
        ref = c.point_to_point(D, M, max_iter=12, tol=1e-6, fixed_iterations=True)
    monkeypatch.setenv("ICP_HOST_ROWS_MAX", "64")
    with pkg.Context(0) as c:
        got = c.point_to_point(D, M, max_iter=12, tol=1e-6, fixed_iterations=True)
    assert got.passes == ref.passes and np.array_equal(got.idx, ref.idx)
    # (the same fp64 terms in another association; the error comes out of the moment sums by a difference that cancels ~30x)
    assert rel(got.T, ref.T) < 1e-12 and np.abs(got.err - ref.err).max() < 1e-11
    want = orc.icp_p2p_f32x(D, M, 12, 1e-6, fixed=True)
    assert np.array_equal(got.idx, want["idx"]) and rel(got.T, want["T"]) < TOL_T


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_non_finite_input_is_refused(ctx, pkg, orc, dtype):
    """include/icp_mi355x.h, "non-finite input": a cloud with a NaN or an infinite coordinate is refused with ICP_ERR_INVALID at
    every entry point that takes a cloud, the context keeps no such cloud, and stays usable.  (The reference has no usable
    answer there: src/ICP_CPU.c:232 leaves a NaN point at index 0 and its centroid sums then poison the whole transform.)"""
    D = pkg.datasets.synthetic_grid(32, dtype)
    M = pkg.datasets.make_model_cpu(D) if dtype == np.float64 else pkg.datasets.make_model_standard(D)
    for bad in (np.nan, np.inf, -np.inf):
        Db, Mb = D.copy(), M.copy()
        Db[7, 1] = bad
        Mb[1000, 2] = bad
        for call in (lambda: ctx.Matching(Db, M), lambda: ctx.Matching(D, Mb), lambda: ctx.point_to_point(Db, M, max_iter=3),
                     lambda: ctx.point_to_point(D, Mb, max_iter=3), lambda: ctx.set_model(Mb), lambda: ctx.set_moving(Db)):
            with pytest.raises(pkg.IcpError) as e:
                call()
            assert e.value.code == -1 and "non-finite" in str(e.value)
    ctx.set_model(M)
    with pytest.raises(pkg.IcpError):
        ctx.set_moving(Db)
    with pytest.raises(pkg.IcpError) as e:      # the refused cloud is not resident
        ctx.nn_match_resident()
    assert e.value.code == -7
    with pytest.raises(pkg.IcpError) as e:
        ctx.set_model_normals(np.full_like(M, np.nan))
    assert e.value.code == -1
    assert np.array_equal(ctx.Matching(D, M), orc.nn(D, M))


@pytest.mark.parametrize("force", ["1", "0"])
def test_morton_views_do_not_change_results(pkg, orc, golden, monkeypatch, force):
    """ICP_SORT=1 forces the sparse kernel onto Morton-ordered views of both clouds (model: sorted scan copy + permutation,
    tie rule on model indices; moving: slot permutation), ICP_SORT=0 forbids them: indices stay bit-exact -- on the hall scan
    (4360 voided duplicates), on the tie-rich synthetic grid, and through a whole registration."""
    monkeypatch.setenv("ICP_SORT", force)
    P, Q = orc.hall_clouds(golden)
    D = pkg.datasets.synthetic_grid(64, np.float32)
    M = pkg.datasets.make_model_gpu(D, *pkg.datasets.P2P_GPU)
    with pkg.Context(0) as c:
        assert np.array_equal(c.Matching(P, Q), orc.nn(P, Q))
        assert np.array_equal(c.Matching(D, M), orc.nn(D, M))
        assert np.array_equal(c.Matching(D, D[::-1].copy()), orc.nn(D, D[::-1].copy()))     # every distance tied with itself
        res = c.point_to_point(P, Q, max_iter=100, tol=1e-6)
    want = orc.icp_p2p_f32x(P, Q, 100, 1e-6)
    assert_same_run(res.iterations, res.err, res.T, want, 1e-6, fp32=True)
    if res.iterations == want["iterations"]:
        assert np.array_equal(res.idx, want["idx"])


@pytest.mark.parametrize("n,m", [(16, 16), (9, 30), (200, 9), (1025, 17), (130, 4097)])
def test_resident_loop_small_and_ragged_clouds(ctx, pkg, orc, n, m):
    """the resident registration kernel on clouds far smaller than a block row / a find pass, ragged against every
    padding granule (blocks of pure padding, a model of two chunks): same run as the oracle"""
    rng = np.random.default_rng(n * 1000 + m)
    M = rng.standard_normal((m, 3)).astype(np.float32)
    pick = rng.integers(0, m, size=n)
    ang = np.array([0.05, -0.03, 0.04])
    cx, sx, cy, sy, cz, sz = np.cos(ang[0]), np.sin(ang[0]), np.cos(ang[1]), np.sin(ang[1]), np.cos(ang[2]), np.sin(ang[2])
    R = (np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]]) @ np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]]) @
         np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]]))
    D = ((M[pick].astype(np.float64) - np.array([0.02, -0.01, 0.03])) @ R).astype(np.float32) + (1e-3 * rng.standard_normal((n, 3))).astype(np.float32)
    res = ctx.point_to_point(D, M, max_iter=12, tol=1e-9)
    want = orc.icp_p2p_f32x(D, M, 12, 1e-9)
    assert_same_run(res.iterations, res.err, res.T, want, 1e-9, fp32=True)
    if res.iterations == want["iterations"]:
        assert np.array_equal(res.idx, want["idx"])


def _fuzz_cloud(rng, n, kind, scale):
    if kind == "uniform":
        X = rng.uniform(-1, 1, (n, 3))
    elif kind == "clustered":          # a few tight clusters with exact duplicates sprinkled in
        c = rng.uniform(-1, 1, (6, 3))
        X = c[rng.integers(0, 6, n)] + 0.01 * rng.standard_normal((n, 3))
        d = rng.integers(0, n, max(1, n // 10))
        X[d] = X[rng.integers(0, n, d.size)]
    elif kind == "lattice":            # integer lattice: exact ties everywhere
        X = rng.integers(-3, 4, (n, 3)).astype(np.float64)
    else:                              # "line": degenerate extent in two axes
        X = np.zeros((n, 3))
        X[:, 0] = rng.uniform(-1, 1, n)
    return (scale * X).astype(np.float32)


# (sort, hier, row): Morton views forbidden / forced; flat search with 64-point rows (nn_match_row64) and with 128-point rows
# (nn_match_sparse), and the box hierarchy (128-point rows only)
FUZZ_VARIANTS = [("0", "0", "64"), ("1", "0", "64"), ("0", "0", "128"), ("1", "0", "128"), ("0", "1", "128"), ("1", "1", "128"),
                 ("0", "0", "128w8"), ("1", "0", "128w8"), ("0", "1", "128w8"), ("1", "1", "128w8"), ("0", "1", "128w4"), ("1", "1", "128w4")]


@pytest.mark.parametrize("sort,hier,row", FUZZ_VARIANTS)
def test_matching_fuzz_against_the_oracle(pkg, orc, monkeypatch, sort, hier, row):
    """randomised clouds (uniform / clustered with duplicates / integer lattice / collinear, three scales, ragged sizes)
    through the sparse kernel with its Morton views forbidden and forced, flat and through the box hierarchy of the
    large models (forced onto models of one or two super boxes): indices bit-exact against the CPU oracle"""
    monkeypatch.setenv("ICP_SORT", sort)
    monkeypatch.setenv("ICP_NN_HIER", hier)
    _set_row(monkeypatch, row)
    rng = np.random.default_rng(20260210 + int(sort))
    with pkg.Context(0) as c:
        for case in range(int(os.environ.get("ICP_FUZZ_CASES", "40"))):     # (a longer soak: ICP_FUZZ_CASES=1000)
            n, m = int(rng.integers(1, 700)), int(rng.integers(1, 900))
            kp, km = rng.choice(["uniform", "clustered", "lattice", "line"], 2)
            scale = float(rng.choice([1e-3, 1.0, 1e3]))
            P, Q = _fuzz_cloud(rng, n, kp, scale), _fuzz_cloud(rng, m, km, scale)
            got, want = c.Matching(P, Q), orc.nn(P, Q)
            assert np.array_equal(got, want), (case, n, m, kp, km, scale, int(np.flatnonzero(got != want)[0]))


@pytest.mark.parametrize("sort,hier,row", FUZZ_VARIANTS)
def test_registration_fuzz_against_the_oracle(pkg, orc, monkeypatch, sort, hier, row):
    """randomised registrations (seeded passes, resident kernel, clustered / lattice models with duplicates and ties),
    flat search and box hierarchy: error series, transform and final correspondences against the oracle's run"""
    monkeypatch.setenv("ICP_SORT", sort)
    monkeypatch.setenv("ICP_NN_HIER", hier)
    _set_row(monkeypatch, row)
    rng = np.random.default_rng(7300 + int(sort))
    with pkg.Context(0) as c:
        for case in range(max(1, int(os.environ.get("ICP_FUZZ_CASES", "40")) // 5)):
            m = int(rng.integers(300, 2500))
            n = int(rng.integers(200, 2500))
            M = _fuzz_cloud(rng, m, str(rng.choice(["uniform", "clustered", "lattice"])), 1.0)
            a = rng.uniform(-0.05, 0.05, 3)
            Rz = np.array([[np.cos(a[2]), -np.sin(a[2]), 0], [np.sin(a[2]), np.cos(a[2]), 0], [0, 0, 1]])
            Ry = np.array([[np.cos(a[1]), 0, np.sin(a[1])], [0, 1, 0], [-np.sin(a[1]), 0, np.cos(a[1])]])
            Rx = np.array([[1, 0, 0], [0, np.cos(a[0]), -np.sin(a[0])], [0, np.sin(a[0]), np.cos(a[0])]])
            D = ((M[rng.integers(0, m, n)].astype(np.float64) + rng.uniform(-0.02, 0.02, 3)) @ (Rz @ Ry @ Rx)).astype(np.float32)
            D += (2e-3 * rng.standard_normal((n, 3))).astype(np.float32)
            res = c.point_to_point(D, M, max_iter=15, tol=1e-7)
            want = orc.icp_p2p_f32x(D, M, 15, 1e-7)
            assert_same_run(res.iterations, res.err, res.T, want, 1e-7, fp32=True)
            if res.iterations == want["iterations"]:
                assert np.array_equal(res.idx, want["idx"]), case


# ---------------------------------------------------------------------------------------------------
# large models: the search goes through the box hierarchy (chunks < 512-point boxes < 32 768-point boxes)
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("kind", ["grid", "uniform", "clustered"])
@pytest.mark.parametrize("n", [300, 3000])
def test_large_model_matching_against_the_oracle(ctx, pkg, orc, kind, n):
    """a model above the size where the hierarchical search is the default (ragged against every level: 140 037 points
    = 17 504.6 chunks, 273.5 super boxes, 4.3 level-3 boxes), a moving cloud of a few rows (the model is then cut into
    segments on super-box boundaries) and of enough rows to go unsegmented; cold and seeded pass: bit-exact"""
    rng = np.random.default_rng(4242 + n)
    m = 140_037
    if kind == "grid":
        M = pkg.datasets.synthetic_grid(375, np.float32)[:m]          # row-major: thin 512-point boxes, own order kept or not
        P = pkg.datasets.make_model_gpu(M[rng.integers(0, m, n)], (0.02, -0.01, 0.015), (0.01, -0.02, 0.005))
    else:
        M = _fuzz_cloud(rng, m, kind, 1.0)
        P = (M[rng.integers(0, m, n)] + (5e-3 * rng.standard_normal((n, 3))).astype(np.float32)).astype(np.float32)
    want = orc.nn(P, M)
    ctx.set_model(M); ctx.set_moving(P)
    ctx.nn_match_resident()
    cold = ctx.get_indices()
    assert np.array_equal(cold, want), int(np.flatnonzero(cold != want)[0])
    ctx.nn_match_bench(1, seeded=True)
    assert np.array_equal(ctx.get_indices(), want)


def test_large_model_registration_against_the_oracle(ctx, pkg, orc):
    """a few iterations on a 2^17-point model (hierarchical search, armed launches: 40 rows + S = 4 segments would not
    be a resident grid): same run as the oracle, correspondences of the last pass bit-exact"""
    W = 363                                                   # 131 769 points
    M = pkg.datasets.synthetic_grid(W, np.float32)
    rng = np.random.default_rng(99)
    D = pkg.datasets.make_model_gpu(M[rng.integers(0, W * W, 5000)], (0.03, -0.02, 0.01), (0.02, -0.01, 0.015))
    res = ctx.point_to_point(D, M, max_iter=4, tol=1e-9, fixed_iterations=True)
    want = orc.icp_p2p_f32x(D, M, 4, 1e-9, fixed=True)
    assert_same_run(res.iterations, res.err, res.T, want, 1e-9, fp32=True)
    if res.iterations == want["iterations"]:
        assert np.array_equal(res.idx, want["idx"])


def test_million_point_self_match_is_the_identity(ctx, pkg):
    """size-independent property at a size the CPU cannot check pair by pair: a 1024 x 1024 grid matched against
    itself returns every point's own index (1.1e12 pairs, brute-force semantics), cold and seeded; against its moved
    copy every index is valid and 64 sampled points agree with numpy's brute force"""
    W = 1024
    D = pkg.datasets.synthetic_grid(W, np.float32)
    ctx.set_model(D); ctx.set_moving(D)
    ctx.nn_match_resident()
    assert np.array_equal(ctx.get_indices(), np.arange(W * W, dtype=np.int32))
    ctx.nn_match_bench(1, seeded=True)
    assert np.array_equal(ctx.get_indices(), np.arange(W * W, dtype=np.int32))
    M = pkg.datasets.make_model_gpu(D, *pkg.datasets.P2P_GPU)
    ctx.set_model(M)
    ctx.nn_match_resident()
    idx = ctx.get_indices()
    assert int(idx.min()) >= 0 and int(idx.max()) < W * W
    for i in np.random.default_rng(5).integers(0, W * W, 64):
        d = (D[i][None, :] - M) ** 2
        assert int(((d[:, 0] + d[:, 1]) + d[:, 2]).argmin()) == int(idx[i])


@pytest.mark.parametrize("deal", [False, True])
def test_two_ranks_large_model_sharded_like_configs4(pkg, orc, deal):
    """BASELINE configs[4] in small: a synthetic grid cloud large enough for the hierarchical search (131 769 model points,
    replicated), the MOVING cloud split over two ranks (both on cuda:0 here; too many rows for a resident kernel each is
    not required; here 2 x 96 rows so that both resident kernels fit the one GPU they share), one sum per iteration through the node-local communicator:
    both ranks end with the same bits, the run equals the oracle's on the whole cloud, indices bit-exact"""
    import subprocess, sys, json
    code = (
        "import sys, os, json, numpy as np\n"
        f"sys.path.insert(0, {ROOT!r})\n"
        "from __graft_entry__ import load_package\n"
        "pkg = load_package()\n"
        "rank, world, idh, deal = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], int(sys.argv[4])\n"
        "M = pkg.datasets.synthetic_grid(363, np.float32)\n"
        "D = pkg.datasets.make_model_gpu(M[np.random.default_rng(7).integers(0, 363 * 363, 24000)], (0.03, -0.02, 0.01), (0.02, -0.01, 0.015))\n"
        "Ds, begin = pkg.distributed.shard(D, rank, world)\n"
        "if deal:   # blocks of 1024 points along a Hilbert curve, dealt to the ranks (what bench.py --config s5 --gpus N does)\n"
        "    sel = pkg.distributed.shard_cyclic_index(len(D), rank, world, 1024, pkg.distributed.curve_order(D)); Ds = np.ascontiguousarray(D[sel])\n"
        "with pkg.Context(0) as ctx:\n"
        "    ctx.set_model(M); ctx.set_moving(Ds)\n"
        "    ctx.comm_init_local(bytes.fromhex(idh), rank, world)\n"
        "    ctx.loop_begin(pkg.ICP_POINT_TO_POINT, max_iter=4, tol=1e-9, fixed_iterations=True)\n"
        "    done = False\n"
        "    while not done:\n"
        "        _, done = ctx.loop_run(1 << 20)\n"
        "    st = ctx.loop_state(); idx = ctx.loop_indices()\n"
        "    ctx.comm_destroy()\n"
        "print(json.dumps(dict(it=st['iterations'], T=st['T'].tolist(), err=st['err'].tolist(), begin=int(begin), idx=idx.tolist())))\n")
    idh = pkg.Context.comm_random_id().hex()
    procs = [subprocess.Popen([sys.executable, "-c", code, str(r), "2", idh, str(int(deal))], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
             for r in range(2)]
    outs = [p.communicate(timeout=300) for p in procs]
    for p, (o, e) in zip(procs, outs):
        assert p.returncode == 0, e[-2000:]
    got = [json.loads(o.strip().splitlines()[-1]) for o, _ in outs]
    assert got[0]["it"] == got[1]["it"] and got[0]["T"] == got[1]["T"] and got[0]["err"] == got[1]["err"]
    M = pkg.datasets.synthetic_grid(363, np.float32)
    D = pkg.datasets.make_model_gpu(M[np.random.default_rng(7).integers(0, 363 * 363, 24000)], (0.03, -0.02, 0.01), (0.02, -0.01, 0.015))
    want = orc.icp_p2p_f32x(D, M, 4, 1e-9, fixed=True)
    assert_same_run(got[0]["it"], np.array(got[0]["err"]), np.array(got[0]["T"]), want, 1e-9, fp32=True)
    if got[0]["it"] == want["iterations"]:
        if deal:   # (a dealt shard's correspondences back in the cloud's order: the two shares are a partition of it)
            order = pkg.distributed.curve_order(D)
            sel = [pkg.distributed.shard_cyclic_index(len(D), r, 2, 1024, order) for r in range(2)]
            assert np.array_equal(np.sort(np.concatenate(sel)), np.arange(len(D)))
            full = np.empty(len(D), dtype=np.int64)
            for r in range(2): full[sel[r]] = got[r]["idx"]
            assert np.array_equal(full, want["idx"])
        else:
            assert np.array_equal(np.concatenate([got[0]["idx"], got[1]["idx"]]), want["idx"])


def test_sticky_pin_lands_on_the_device_numa_node(pkg):
    """ICP_PIN=2 (opt-in, a thread dedicated to the context): icp_create narrows the calling thread to the CPUs sysfs lists
    as local to the device and leaves it there; ICP_PIN=0 and the default (scoped narrowing) hand the caller's mask back
    (tests/test_gpu_runtime.py checks that after every entry point)"""
    import subprocess, sys, json
    code = (
        "import os, sys, json, ctypes\n"
        f"sys.path.insert(0, {ROOT!r})\n"
        "from __graft_entry__ import load_package\n"
        "pkg = load_package()\n"
        "before = sorted(os.sched_getaffinity(0))\n"
        "with pkg.Context(0) as ctx:\n"
        "    after = sorted(os.sched_getaffinity(0))\n"
        "hip = ctypes.CDLL('libamdhip64.so'); buf = ctypes.create_string_buffer(64); hip.hipDeviceGetPCIBusId(buf, 64, 0)\n"
        "p = '/sys/bus/pci/devices/' + buf.value.decode().lower() + '/local_cpulist'\n"
        "local = open(p).read().strip() if os.path.exists(p) else ''\n"
        "print(json.dumps(dict(before=before, after=after, local=local)))\n")
    def run(env):
        out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **env), capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stderr[-2000:]
        return json.loads(out.stdout.strip().splitlines()[-1])
    def parse(s):
        cpus = set()
        for part in filter(None, s.split(",")):
            a, _, b = part.partition("-")
            cpus.update(range(int(a), int(b or a) + 1))
        return cpus
    for pin in ("0", "1"):
        r = run({"ICP_PIN": pin})
        assert r["after"] == r["before"], pin
    r = run({"ICP_PIN": "2"})
    local = parse(r["local"])
    want = sorted(set(r["before"]) & local)
    if local and want and len(want) < len(r["before"]):
        assert r["after"] == want
    else:
        assert r["after"] == r["before"]      # nothing to narrow on this machine


def test_large_model_and_too_many_rows_for_a_resident_kernel(ctx, pkg, orc):
    """hierarchical search on the one-launch-per-pass path: 33 000 moving points are 264 rows, more blocks than the machine
    holds at once, so every pass is its own (armed) launch of the hierarchical kernel: same run as the oracle"""
    M = pkg.datasets.synthetic_grid(363, np.float32)
    D = pkg.datasets.make_model_gpu(M[np.random.default_rng(11).integers(0, 363 * 363, 33000)], (0.03, -0.02, 0.01), (0.02, -0.01, 0.015))
    res = ctx.point_to_point(D, M, max_iter=2, tol=1e-9, fixed_iterations=True)
    want = orc.icp_p2p_f32x(D, M, 2, 1e-9, fixed=True)
    assert_same_run(res.iterations, res.err, res.T, want, 1e-9, fp32=True)
    if res.iterations == want["iterations"]:
        assert np.array_equal(res.idx, want["idx"])


def test_configs4_full_size_properties(ctx, pkg):
    """BASELINE configs[4] at its full size, the share of one rank of 8: 1 250 000 moving points of the 10 M-point
    synthetic cloud against the whole moved 10 M-point model (1.25e13 pairs -- far beyond the CPU oracle), through
    size-independent properties: every index valid, cold and seeded passes agree, 48 sampled points agree with numpy's
    brute force over all 10 M model points, and matching the shard a second time after a no-op is idempotent"""
    N = 10_000_000
    W = int(np.ceil(np.sqrt(N)))
    D = pkg.datasets.synthetic_grid(W, np.float32)[:N]
    M = pkg.datasets.make_model_gpu(D, *pkg.datasets.P2P_GPU)
    lo, cnt = pkg.shard_range(N, 3, 8)
    P = np.ascontiguousarray(D[lo:lo + cnt])
    del D
    ctx.set_model(M); ctx.set_moving(P)
    ctx.nn_match_resident()
    cold = ctx.get_indices()
    assert cold.shape == (cnt,) and int(cold.min()) >= 0 and int(cold.max()) < N
    ctx.nn_match_bench(1, seeded=True)
    seeded = ctx.get_indices()
    assert np.array_equal(cold, seeded)
    for i in np.random.default_rng(17).integers(0, cnt, 48):
        d = (P[i][None, :] - M) ** 2
        assert int(((d[:, 0] + d[:, 1]) + d[:, 2]).argmin()) == int(cold[i])
    ctx.nn_match_bench(2, seeded=True)
    assert np.array_equal(ctx.get_indices(), cold)


def test_configs4_share_against_the_kernel_that_executes_every_pair(pkg, monkeypatch):
    """BASELINE configs[4] at its full model size, ALL correspondences of a share (rank 5 of 32: 312 500 moving points against the
    whole 10 M-point model, 3.1e12 pairs): the hierarchical search -- cold, then seeded by its own matches -- against the dense packed
    kernel of a context created under ICP_NN_SPARSE=0, which executes every pair (no boxes, no bounds, no order, no views): the two
    share the arithmetic of a pair and the tie rule and nothing of the search.  Beyond the CPU oracle's reach (hours); the oracle
    pins both kernels on the smaller configs."""
    N = 10_000_000
    W = int(np.ceil(np.sqrt(N)))
    D = pkg.datasets.synthetic_grid(W, np.float32)[:N]
    M = pkg.datasets.make_model_gpu(D, *pkg.datasets.P2P_GPU)
    lo, cnt = pkg.shard_range(N, 5, 32)
    P = np.ascontiguousarray(D[lo:lo + cnt])
    del D
    monkeypatch.delenv("ICP_NN_SPARSE", raising=False)
    with pkg.Context(0) as c:
        c.set_model(M); c.set_moving(P)
        assert c.nn_launch_info()["threads"] == 256                     # the hierarchical form: 4-wave blocks
        c.nn_match_resident(); cold = c.get_indices()
        c.nn_match_bench(1, seeded=True); seeded = c.get_indices()
    monkeypatch.setenv("ICP_NN_SPARSE", "0")
    with pkg.Context(0) as c:
        c.set_model(M); c.set_moving(P)
        c.nn_match_resident(); dense = c.get_indices()
    monkeypatch.delenv("ICP_NN_SPARSE", raising=False)
    assert np.array_equal(cold, dense) and np.array_equal(seeded, dense)


def test_configs4_share_loop_ordered_rows_and_padding(pkg, monkeypatch):
    """the same share (1.25 M points: 9766 rows, the last one partly filled) through five iterations of the loop: the launches
    take the rows heaviest first (sorted by the hits of the launch before) or in index order -- the same bits; the error
    falls; and the correspondences of sampled points -- the last, padded row's among them -- are numpy's brute-force
    matches of the cloud as it stood one transform earlier"""
    N = 10_000_000
    W = int(np.ceil(np.sqrt(N)))
    D = pkg.datasets.synthetic_grid(W, np.float32)[:N]
    M = pkg.datasets.make_model_gpu(D, *pkg.datasets.P2P_GPU)
    lo, cnt = pkg.shard_range(N, 3, 8)
    P = np.ascontiguousarray(D[lo:lo + cnt])
    del D
    res = {}
    for name, env in (("ordered", {}), ("index_order", {"ICP_NN_ORDER": "0"}), ("split_small", {"ICP_NN_SPLIT_MIN": "256"})):
        monkeypatch.delenv("ICP_NN_ORDER", raising=False)
        monkeypatch.delenv("ICP_NN_SPLIT_MIN", raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        with pkg.Context(0) as c:
            res[name] = c.point_to_point(P, M, max_iter=5, tol=0.0, fixed_iterations=True)
            if name == "ordered":
                # one more matching pass on the moved cloud, stand-alone: what the NEXT pass of the loop would have matched
                c.set_model(M); c.set_moving(res[name].moved)
                c.nn_match_resident()
                after = c.get_indices()
    monkeypatch.delenv("ICP_NN_ORDER", raising=False)
    monkeypatch.delenv("ICP_NN_SPLIT_MIN", raising=False)
    a = res["ordered"]
    assert a.iterations == 5
    for name in ("index_order", "split_small"):   # (split rows: the heaviest rows are searched by several blocks each)
        b = res[name]
        assert b.iterations == 5, name
        assert np.array_equal(a.T, b.T) and np.array_equal(a.err[:-1], b.err[:-1]) and np.array_equal(a.idx, b.idx) and np.array_equal(a.moved, b.moved), name
    assert (np.diff(a.err[1:]) < 0).all()
    sample = np.concatenate([np.random.default_rng(5).integers(0, cnt, 24), np.arange(cnt - 16, cnt)])   # (+ the padded last row)
    for i in sample:
        d = (a.moved[i][None, :] - M) ** 2
        assert int(((d[:, 0] + d[:, 1]) + d[:, 2]).argmin()) == int(after[i])


def test_armed_loop_in_single_steps_equals_one_run(pkg, orc, golden, monkeypatch):
    """one launch per pass (ICP_RESIDENT=0: armed launches that start from the previous pass's slot-ordered points and matches):
    driving the loop one step per call -- the pass armed ahead is withdrawn and armed again every time -- gives the bits of one
    uninterrupted run, and both equal the oracle's run"""
    P, Q = orc.hall_clouds(golden)
    def whole(c):
        c.set_model(Q); c.set_moving(P)
        c.loop_begin(pkg.ICP_POINT_TO_POINT, max_iter=100, tol=1e-6)
        done = False
        while not done:
            _, done = c.loop_run(1 << 20)
        return c.loop_state(), c.loop_indices()
    def stepped(c):
        c.set_model(Q); c.set_moving(P)
        c.loop_begin(pkg.ICP_POINT_TO_POINT, max_iter=100, tol=1e-6)
        done, calls = False, 0
        while not done:
            _, done = c.loop_run(1 if calls % 3 else 2)
            calls += 1
        return c.loop_state(), c.loop_indices()
    a, ia = _run_form(pkg, monkeypatch, {"ICP_RESIDENT": "0"}, whole)
    b, ib = _run_form(pkg, monkeypatch, {"ICP_RESIDENT": "0"}, stepped)
    assert a["iterations"] == b["iterations"] and np.array_equal(a["T"], b["T"]) and np.array_equal(a["err"], b["err"])
    assert np.array_equal(ia, ib)
    want = orc.icp_p2p_f32x(P, Q, 100, 1e-6)
    assert_same_run(a["iterations"], a["err"], a["T"], want, 1e-6, fp32=True)
    if a["iterations"] == want["iterations"]:
        assert np.array_equal(ia, want["idx"])
