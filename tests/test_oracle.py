"""CPU: pin the oracle.  (1) the run records of the unmodified reference kept in BASELINE.md section 2,
(2) an independent numpy/LAPACK restatement, (3) the committed dataset fixtures."""
import json
import os

import numpy as np
import pytest

import ref_numpy


def test_reference_run_record_width32(orc):
    # BASELINE.md section 2 / SURVEY.md 8c: src/ICP_CPU.c with WIDTH 32 -> 56 iterations (57 matching passes)
    D, M = orc.synth_icp_cpu(32)
    r = orc.icp_p2p(D, M, 200, 1e-5)
    assert r["iterations"] == 56 and r["passes"] == 57


def test_reference_run_record_width100(orc):
    # src/ICP_CPU.c as shipped (10 000 pts) -> 61 iterations, E = 0.82815
    D, M = orc.synth_icp_cpu(100)
    r = orc.icp_p2p(D, M, 200, 1e-5)
    assert r["iterations"] == 61
    assert f"{r['err'][61]:.5f}" == "0.82815"   # the last value the reference prints (E[num_iterations])


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_nn_against_numpy(orc, dtype):
    rng = np.random.default_rng(7)
    P = rng.standard_normal((257, 3)).astype(dtype)
    Q = rng.standard_normal((300, 3)).astype(dtype)
    Q[40] = Q[7]          # exact duplicates: the lower index must win
    Q[299] = Q[7]
    P[0] = Q[7]
    idx = orc.nn(P, Q)
    assert np.array_equal(idx, ref_numpy.nn(P, Q))
    assert idx[0] == 7


def test_nn_grid_ties(orc):
    # integer lattice: many exactly equal distances -> pure tie-break test
    g = np.stack(np.meshgrid(np.arange(6.0), np.arange(6.0), np.arange(3.0), indexing="ij"), -1).reshape(-1, 3)
    P = (g[::2] + 0.5).astype(np.float32)
    idx = orc.nn(P, g.astype(np.float32))
    assert np.array_equal(idx, ref_numpy.nn(P, g.astype(np.float32)))


def test_minimize_against_lapack(orc):
    rng = np.random.default_rng(3)
    P = rng.standard_normal((500, 3))
    Q = rng.standard_normal((500, 3))
    idx = orc.nn(P, Q)
    R, t, N = orc.p2p_minimize(P, Q, idx)
    R2, t2 = ref_numpy.minimize(P, Q, idx)
    assert np.abs(R - R2).max() < 1e-12 and np.abs(t - t2).max() < 1e-12
    assert abs(abs(np.linalg.det(R)) - 1) < 1e-12


def test_full_loop_against_numpy(orc):
    D, M = orc.synth_icp_cpu(16)
    a = orc.icp_p2p(D, M, 200, 1e-5)
    b = ref_numpy.icp(D, M, 200, 1e-5)
    assert a["iterations"] == b["iterations"]
    assert np.array_equal(a["idx"], b["idx"])
    assert np.abs(a["T"] - b["T"]).max() < 1e-9
    assert np.abs(a["err"] - b["err"][: len(a["err"])]).max() < 1e-12


def test_f32_twin_recovers_ground_truth(orc):
    # src/ICP_point_to_point.cu inputs (t = (0.8,-0.3,0.2), angles (0.2,-0.2,0.05)): small motion, ICP
    # must pull the moving cloud close to the model (sanity bound, not a parity target)
    D = orc.synth_grid_f32(24)
    M = orc.gpu_model_f32(D, (0.2, -0.2, 0.05), (0.8, -0.3, 0.2))
    r = orc.icp_p2p(D, M, 40, 1e-6)
    assert r["err"][-1] < r["err"][1]
    assert r["err"][-1] < 0.2


def test_hall_state_machine_against_fixture(orc, golden):
    lines = np.array([int(x) for x in open(os.path.join(golden, "os1_two_packets.csv")).read().split()], dtype=np.int32)
    assert lines.size == 2 * 12608
    r, enc, n = orc.os1_ranges_from_lines(lines)
    want = np.fromfile(os.path.join(golden, "hall_ranges_u32.bin"), dtype=np.uint32)
    meta = json.load(open(os.path.join(golden, "hall_meta.json")))
    assert enc == meta["encoder_count0"]
    # the reference's line-number state machine and the byte-offset parser agree on every range it emits
    assert n >= 2 * 256 - 1
    k = min(n, 512)
    assert np.array_equal(r[:k].astype(np.uint32), want[:k])
    assert meta["n_zero_ranges"] == int((want == 0).sum()) == 4361


def test_bunny_reader_against_fixture(orc, golden):
    for head, full in (("bunny_res_head.csv", "bunny_res_xyz_f32.bin"), ("bunny_head.csv", "bunny_xyz_f32.bin")):
        pts = orc.read_xyz_text(os.path.join(golden, head))
        want = np.fromfile(os.path.join(golden, full), dtype=np.float32).reshape(-1, 3)
        assert pts.shape == (64, 3)
        assert np.array_equal(pts, want[:64])


def test_hall_cloud_shape(orc, golden):
    P, Q = orc.hall_clouds(golden)
    assert P.shape == Q.shape == (16384, 3)
    assert int((np.abs(P).sum(1) == 0).sum()) == 4361      # no-return beams collapse onto the origin
    assert np.abs(P).max() < 200.0                          # metres


# ---------------------------------------------------------------------------------------------------
# (4) known-answer vectors from Intel MKL itself (tests/golden/mkl_vectors.npz, written in the build container by
#     tests/golden/make_mkl_vectors.py: ctypes on the MKL 2021.4 runtime the reference links -- MKL does not travel).
#     The reference's third-party statements, src/ICP_CPU.c:227-232 (vdSub, vdSqr, vdAdd x 2, cblas_idamin), :239-248
#     (cblas_dgemm, LAPACKE_dgesvd, cblas_dgemm x 2, vdSub), :251-253 (cblas_dgemm, vdAdd), :266 (cblas_dnrm2), were
#     restated from MKL's documentation; these replays check the restatement against the library.
#     They pin the oracle's model of MKL, not a run of the reference program: "parity unpinned"  stands.
# ---------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def mkl(golden):
    return np.load(os.path.join(golden, "mkl_vectors.npz"))


def test_mkl_amin_is_the_first_minimum_of_the_absolute_values(mkl):
    # cblas_i?amin (src/ICP_CPU.c:232): the oracle's `c == 0 || d < best` scan == first index of min |x|; distances are >= 0
    assert "2021.4" in str(mkl["mkl_version"])
    for k in range(int(mkl["amin_count"])):
        v = mkl[f"amin_vec_{k}"]
        first = int(np.argmin(np.abs(v)))                       # numpy: first occurrence
        first32 = int(np.argmin(np.abs(v.astype(np.float32))))  # (1e-300 is zero in float)
        assert int(mkl[f"amin_d_{k}"]) == first and int(mkl[f"amin_s_{k}"]) == first32, k
        best, besti = 0.0, 0                                    # the oracle's loop, on a vector of distances (all >= 0)
        for c, d in enumerate(np.abs(v)):
            if c == 0 or d < best:
                best, besti = d, c
        assert besti == first


@pytest.mark.parametrize("tag,dtype", [("f64", np.float64), ("f32", np.float32)])
def test_mkl_matching_chain_on_lattice_ties(orc, mkl, tag, dtype):
    # vdSub / vdSqr / vdAdd / vdAdd / idamin on clouds whose distances are exact and tie everywhere: the oracle's indices
    P, Q = mkl[f"lattice_P_{tag}"], mkl[f"lattice_Q_{tag}"]
    assert P.dtype == dtype
    want = mkl[f"lattice_idx_{tag}"]
    assert np.array_equal(orc.nn(P, Q), want)
    d = ((Q[want] - P) ** 2)
    assert np.array_equal(((d[:, 0] + d[:, 1]) + d[:, 2]).astype(dtype), mkl[f"lattice_dmin_{tag}"])
    # the sample really is tie-laden: most points have several model points at the minimum
    dist = (((Q[None, :, :] - P[:, None, :]) ** 2).sum(-1))
    assert ((dist == dist.min(1, keepdims=True)).sum(1) > 1).mean() > 0.5


@pytest.mark.parametrize("tag,dtype", [("f32", np.float32), ("f64", np.float64)])
def test_mkl_matching_chain_on_hall_points(orc, mkl, golden, tag, dtype):
    # the same chain on real coordinates: a spread sample of the hall scan (its coincident no-return points among them)
    # against the whole 16 384-point model; VML rounds every sub / square / add on its own, as the oracle assumes
    P = mkl["hall_P_f32"].astype(dtype)
    Q = mkl["hall_Q_f32"].astype(dtype)
    D, M = orc.hall_clouds(golden)                               # the fixture's model IS the oracle's hall model
    assert np.array_equal(M, mkl["hall_Q_f32"]) and np.array_equal(D[mkl["hall_rows"]], mkl["hall_P_f32"])
    want = mkl[f"hall_idx_{tag}"]
    assert np.array_equal(orc.nn(P, Q), want)
    dx, dy, dz = ((Q[want] - P) ** 2).T
    assert np.array_equal(((dx + dy) + dz).astype(dtype), mkl[f"hall_dmin_{tag}"])   # bit for bit the minimum MKL found


def _moments_of(P, Q, idx):
    P = np.asarray(P, dtype=np.float64)
    Qi = np.asarray(Q, dtype=np.float64)[idx]
    mom = np.zeros(32)
    mom[1] = P.shape[0]
    mom[2:5] = P.sum(0)
    mom[5:8] = Qi.sum(0)
    mom[8:17] = (Qi.T @ P).reshape(9)
    return mom


def test_mkl_whole_passes_of_the_cpu_program(orc, mkl, pkg):
    # src/ICP_CPU.c's own clouds (WIDTH 32), passes 0 / 1 / 5 / 30: matching, dgemm + dgesvd + dgemm (R = U Vt), t, the
    # transformed cloud (dgemm + vdAdd) and the error (dnrm2) as MKL computes them, against the oracle's statements and the
    # product's host solve
    D, M = orc.synth_icp_cpu(int(mkl["synth_W"]))
    for k in mkl["synth_passes"]:
        pt = mkl[f"synth_pt_{k}"]
        idx = mkl[f"synth_idx_{k}"]
        assert np.array_equal(orc.nn(pt, M), idx), k
        R, t, N = orc.p2p_minimize(pt, M, idx)
        scale = np.abs(mkl[f"synth_N_{k}"]).max()
        assert np.abs(N.reshape(9) - mkl[f"synth_N_{k}"]).max() < 1e-12 * scale           # dgemm's order of summation: rounding noise
        assert np.abs(R.reshape(9) - mkl[f"synth_R_{k}"]).max() < 1e-12                    # R = U * Vt, no reflection fix
        assert np.abs(t - mkl[f"synth_t_{k}"]).max() < 1e-12
        U, Vt = mkl[f"synth_U_{k}"].reshape(3, 3), mkl[f"synth_Vt_{k}"].reshape(3, 3)
        assert np.abs(U @ Vt - mkl[f"synth_R_{k}"].reshape(3, 3)).max() < 1e-14
        # the product's host half from raw moments (icp_solve_point_to_point; csrc/icp_host_math.cpp)
        R2, t2 = pkg.solve_point_to_point(_moments_of(pt, M, idx))
        assert np.abs(R2.reshape(9) - mkl[f"synth_R_{k}"]).max() < 1e-10 and np.abs(t2 - mkl[f"synth_t_{k}"]).max() < 1e-10
        # transformation: MKL's dgemm may fuse the three products of a row (fma) where the oracle and the kernels round each
        # operation -- the clouds agree to an ulp or two of their coordinates, which is what the parity tests' 1e-11 allows
        new = orc.transform(pt, mkl[f"synth_R_{k}"].reshape(3, 3), mkl[f"synth_t_{k}"])
        assert np.abs(new - mkl[f"synth_new_{k}"]).max() < 4e-15 * max(1.0, np.abs(new).max())
        E = orc.rms_error(mkl[f"synth_new_{k}"], M, idx)
        assert abs(E - float(mkl[f"synth_E_{k}"])) < 1e-13 * max(1.0, E)


def test_mkl_minimisation_on_the_hall_pair(orc, mkl, pkg):
    # the hall pair widened to double, pass 0, all 16 384 x 16 384 pairs through MKL's chain: indices, R, t, error
    P = mkl["hall_P_f32"]
    D64 = None
    Q64 = mkl["hall_Q_f32"].astype(np.float64)
    rows = mkl["hall_rows"]
    idx = mkl["hallmin_idx"]
    assert np.array_equal(idx[rows], mkl["hall_idx_f64"])        # (the sampled test above is a subset of this run)
    import oracle_lib  # noqa: F401
    golden = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    D, M = orc.hall_clouds(golden)
    D64 = D.astype(np.float64)
    assert np.array_equal(M.astype(np.float64), Q64) and np.array_equal(D[rows], P)
    R, t, N = orc.p2p_minimize(D64, Q64, idx)
    assert np.abs(N.reshape(9) - mkl["hallmin_N"]).max() < 1e-11 * np.abs(mkl["hallmin_N"]).max()
    assert np.abs(R.reshape(9) - mkl["hallmin_R"]).max() < 1e-11 and np.abs(t - mkl["hallmin_t"]).max() < 1e-11
    R2, t2 = pkg.solve_point_to_point(_moments_of(D64, Q64, idx))
    assert np.abs(R2.reshape(9) - mkl["hallmin_R"]).max() < 1e-9 and np.abs(t2 - mkl["hallmin_t"]).max() < 1e-9
    E = orc.rms_error(orc.transform(D64, mkl["hallmin_R"].reshape(3, 3), mkl["hallmin_t"]), Q64, idx)
    assert abs(E - float(mkl["hallmin_E"])) < 1e-12 * max(1.0, E)
