#!/usr/bin/env python3
"""Known-answer vectors from Intel MKL ITSELF for the third-party statements of the reference's CPU path.

    python tests/golden/make_mkl_vectors.py          (build container only: MKL does not travel to the GPU box)

The reference (src/ICP_CPU.c, src/CUDA/CPU_ICP_point_to_point.cpp) delegates its arithmetic to MKL 2021.4 -- vdSub / vdSqr /
vdAdd and cblas_idamin in the matching loop (ICP_CPU.c:227-232), cblas_dgemm + LAPACKE_dgesvd + cblas_dgemm in the minimisation
(:239-248), cblas_dgemm + vdAdd in the transformation (:251-253), cblas_dnrm2 in the error (:266).  MKL is not part of
/root/reference; the oracle (oracle/icp_oracle*.c/h) restates its documented semantics.  This script replaces "restated from the
documentation" by "checked against the library the reference links": it calls those very routines of the MKL runtime that
happens to be installed here (/opt/conda/lib/libmkl_rt.so, 2021.4.0 -- the version SURVEY.md 8c probed) through ctypes -- no
headers are needed for that, and none are written -- on tie-laden and real inputs, and stores inputs and answers as
tests/golden/mkl_vectors.npz.  tests/test_oracle.py (CPU suite) replays them against orc_nn_*, orc_p2p_minimize_*,
orc_transform_*, orc_rms_error_* and against the product's host solve.

What this does and does not pin: it pins the ORACLE's model of MKL (first minimum of cblas_i?amin, separately rounded VML
operations, R = U * Vt of LAPACKE_dgesvd, the summation of dgemm / dnrm2 to rounding noise).  It is not a run of the reference
program -- that would need the headers this image lacks -- so DESIGN.md's "parity unpinned" stands.

Nothing of the reference's source text is stored: inputs are lattice clouds made here, the synthetic surface of the generators
(through the oracle) and points of the committed hall fixture.
"""
import ctypes as C
import os
import sys

import numpy as np

os.environ.setdefault("MKL_THREADING_LAYER", "SEQUENTIAL")   # (no libiomp5 needed; the reference's sequential numbers are its own baseline)
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
MKL_PATH = os.environ.get("MKL_RT", "/opt/conda/lib/libmkl_rt.so")

CblasRowMajor, CblasNoTrans, CblasTrans, LAPACK_ROW_MAJOR = 101, 111, 112, 101


def load_mkl():
    mkl = C.CDLL(MKL_PATH, mode=C.RTLD_GLOBAL)
    i, d, f, vp, sz = C.c_int, C.c_double, C.c_float, C.c_void_p, C.c_size_t
    mkl.cblas_idamin.restype = sz; mkl.cblas_idamin.argtypes = [i, vp, i]
    mkl.cblas_isamin.restype = sz; mkl.cblas_isamin.argtypes = [i, vp, i]
    for name in ("vdSub", "vdAdd", "vsSub", "vsAdd"):
        getattr(mkl, name).restype = None; getattr(mkl, name).argtypes = [i, vp, vp, vp]
    for name in ("vdSqr", "vsSqr"):
        getattr(mkl, name).restype = None; getattr(mkl, name).argtypes = [i, vp, vp]
    mkl.cblas_dgemm.restype = None
    mkl.cblas_dgemm.argtypes = [i, i, i, i, i, i, d, vp, i, vp, i, d, vp, i]
    mkl.cblas_sgemm.restype = None
    mkl.cblas_sgemm.argtypes = [i, i, i, i, i, i, f, vp, i, vp, i, f, vp, i]
    mkl.LAPACKE_dgesvd.restype = i
    mkl.LAPACKE_dgesvd.argtypes = [i, C.c_char, C.c_char, i, i, vp, i, vp, vp, i, vp, i, vp]
    mkl.cblas_dnrm2.restype = d; mkl.cblas_dnrm2.argtypes = [i, vp, i]
    mkl.cblas_snrm2.restype = f; mkl.cblas_snrm2.argtypes = [i, vp, i]
    buf = C.create_string_buffer(256)
    mkl.mkl_get_version_string.restype = None
    mkl.mkl_get_version_string(buf, 256)
    return mkl, buf.value.decode(errors="replace").strip()


def p(a):
    return a.ctypes.data_as(C.c_void_p)


def mkl_matching(mkl, pt_soa, q_soa, n, m, rows, dtype):
    """the statement group src/ICP_CPU.c:222-233 (fp64) / src/CUDA/CPU_ICP_point_to_point.cpp:188-203 (fp32) for moving points
    `rows`: copy the point over a 3 x m scratch, ?Sub, ?Sqr, ?Add, ?Add, i?amin -- the MKL calls, in that order"""
    sub, sqr, add, amin = ((mkl.vdSub, mkl.vdSqr, mkl.vdAdd, mkl.cblas_idamin) if dtype == np.float64 else
                           (mkl.vsSub, mkl.vsSqr, mkl.vsAdd, mkl.cblas_isamin))
    p1 = np.zeros(3 * m, dtype=dtype)
    dist = np.zeros(m, dtype=dtype)
    es = np.dtype(dtype).itemsize
    idx = np.zeros(len(rows), dtype=np.int32)
    dmin = np.zeros(len(rows), dtype=dtype)
    for k, j in enumerate(rows):
        for c in range(3):
            p1[c * m:(c + 1) * m] = pt_soa[j + n * c]        # cblas_?copy with incx = 0: a broadcast
        sub(3 * m, p(q_soa), p(p1), p(p1))
        sqr(3 * m, p(p1), p(p1))
        add(m, p(p1), C.c_void_p(p1.ctypes.data + m * es), p(dist))
        add(m, p(dist), C.c_void_p(p1.ctypes.data + 2 * m * es), p(dist))
        idx[k] = int(amin(m, p(dist), 1))
        dmin[k] = dist[idx[k]]
    return idx, dmin


def centroid_deviation(cloud_soa, size, index):
    """centroid_deviation of src/ICP_CPU.c:342-366 -- the reference's own C, not MKL: sequential sums, x * (1/n)"""
    g = cloud_soa.reshape(3, size)[:, index]
    bar = np.array([np.cumsum(g[c])[-1] * (1.0 / float(size)) for c in range(3)])   # (np.cumsum adds left to right)
    return bar, (g - bar[:, None]).copy()


def mkl_minimise(mkl, pt_soa, q_soa, n, idx):
    """src/ICP_CPU.c:237-248 with the MKL calls as written there"""
    q_bar, q_mark = centroid_deviation(q_soa, n, idx)
    p_bar, p_mark = centroid_deviation(pt_soa, n, np.arange(n))
    q_mark = np.ascontiguousarray(q_mark); p_mark = np.ascontiguousarray(p_mark)
    N = np.zeros(9); S = np.zeros(3); U = np.zeros(9); Vt = np.zeros(9); superb = np.zeros(4)
    mkl.cblas_dgemm(CblasRowMajor, CblasNoTrans, CblasTrans, 3, 3, n, 1.0, p(q_mark), n, p(p_mark), n, 0.0, p(N), 3)
    N_in = N.copy()
    info = mkl.LAPACKE_dgesvd(LAPACK_ROW_MAJOR, b"A", b"A", 3, 3, p(N), 3, p(S), p(U), 3, p(Vt), 3, p(superb))
    assert info == 0
    R = np.zeros(9); G = np.zeros(3); t = np.zeros(3)
    mkl.cblas_dgemm(CblasRowMajor, CblasNoTrans, CblasNoTrans, 3, 3, 3, 1.0, p(U), 3, p(Vt), 3, 0.0, p(R), 3)
    pb = np.ascontiguousarray(p_bar)
    mkl.cblas_dgemm(CblasRowMajor, CblasNoTrans, CblasTrans, 3, 1, 3, 1.0, p(R), 3, p(pb), 3, 0.0, p(G), 1)
    qb = np.ascontiguousarray(q_bar)
    mkl.vdSub(3, p(qb), p(G), p(t))
    return dict(N=N_in, S=S, U=U, Vt=Vt, R=R, t=t, q_bar=q_bar, p_bar=p_bar)


def mkl_transform_error(mkl, pt_soa, q_soa, n, idx, R, t):
    """src/ICP_CPU.c:251-253 and :257-266"""
    Cm = np.zeros(3 * n); F = np.repeat(t, n)
    pt = np.ascontiguousarray(pt_soa); Rm = np.ascontiguousarray(R)
    mkl.cblas_dgemm(CblasRowMajor, CblasNoTrans, CblasNoTrans, 3, n, 3, 1.0, p(Rm), 3, p(pt), n, 0.0, p(Cm), n)
    new = np.zeros(3 * n)
    mkl.vdAdd(3 * n, p(Cm), p(F), p(new))
    gath = np.ascontiguousarray(q_soa.reshape(3, n)[:, idx]).reshape(-1)
    diff = np.zeros(3 * n)
    mkl.vdSub(3 * n, p(gath), p(new), p(diff))
    E = mkl.cblas_dnrm2(3 * n, p(diff), 1) / (float(n) ** 0.5)
    return new, float(E)


def main():
    import oracle_lib
    orc = oracle_lib.Oracle()
    mkl, version = load_mkl()
    print("MKL:", version)
    out = {"mkl_version": np.array(version)}
    rng = np.random.default_rng(20261005)

    # ---- A. cblas_i?amin on vectors with ties, signs and zeros: "first index of the minimum absolute value" -------------
    vecs = [np.array([3, 1, 2, 1, 1, 5.0]), np.array([2.0, 2, 2, 2]), np.array([5, 4, 3, 2, 1, 1.0]), np.array([0.0, -0.0, 0, 1]),
            np.array([-1, 1, -1, 1.0]), np.array([7.0]), np.array([4, -3, 3, 9, -3.0]), np.array([1e-300, 0, 1e-300, 0.0]),
            rng.integers(0, 4, 257).astype(np.float64), rng.integers(-3, 4, 1000).astype(np.float64)]
    for k, v in enumerate(vecs):
        v64 = np.ascontiguousarray(v, dtype=np.float64); v32 = np.ascontiguousarray(v, dtype=np.float32)
        out[f"amin_vec_{k}"] = v64
        out[f"amin_d_{k}"] = np.array(int(mkl.cblas_idamin(v64.size, p(v64), 1)))
        out[f"amin_s_{k}"] = np.array(int(mkl.cblas_isamin(v32.size, p(v32), 1)))
    out["amin_count"] = np.array(len(vecs))

    # ---- B. the matching chain on lattice clouds: exact distances, ties everywhere -----------------------------------------
    for tag, dtype in (("f64", np.float64), ("f32", np.float32)):
        Q = rng.integers(-3, 4, size=(512, 3)).astype(dtype) * dtype(0.5)        # a 7^3 lattice, many exact duplicates
        P = rng.integers(-7, 8, size=(96, 3)).astype(dtype) * dtype(0.25)         # half-way points: ties between lattice neighbours
        ps, qs = oracle_lib.soa(P), oracle_lib.soa(Q)
        idx, dmin = mkl_matching(mkl, ps, qs, P.shape[0], Q.shape[0], range(P.shape[0]), dtype)
        out[f"lattice_P_{tag}"] = P; out[f"lattice_Q_{tag}"] = Q; out[f"lattice_idx_{tag}"] = idx; out[f"lattice_dmin_{tag}"] = dmin

    # ---- C. the chain on the hall pair (committed fixture -> oracle's ingest), a spread sample of moving points ----------------
    D32, M32 = orc.hall_clouds(HERE)
    n = D32.shape[0]
    rows = np.unique(np.concatenate([np.arange(0, n, 97), np.flatnonzero((D32 == 0).all(1))[:16], [n - 1]])).astype(np.int64)
    out["hall_rows"] = rows
    out["hall_P_f32"] = D32[rows].copy()
    out["hall_Q_f32"] = M32.copy()
    idx32, dmin32 = mkl_matching(mkl, oracle_lib.soa(D32), oracle_lib.soa(M32), n, n, rows, np.float32)
    D64, M64 = D32.astype(np.float64), M32.astype(np.float64)
    idx64, dmin64 = mkl_matching(mkl, oracle_lib.soa(D64), oracle_lib.soa(M64), n, n, rows, np.float64)
    out["hall_idx_f32"] = idx32; out["hall_dmin_f32"] = dmin32; out["hall_idx_f64"] = idx64; out["hall_dmin_f64"] = dmin64

    # ---- D. ICP_CPU.c's own run (synthetic surface, WIDTH 32): whole passes through MKL -- matching, minimisation,
    #         transformation, error -- at pass 0 and from the oracle's clouds after 1, 5 and 30 iterations -------------------
    W = 32
    D, M = orc.synth_icp_cpu(W)
    n = W * W
    qs = oracle_lib.soa(M)
    passes = [0, 1, 5, 30]
    out["synth_W"] = np.array(W); out["synth_passes"] = np.array(passes)
    for k in passes:
        pt = D if k == 0 else orc.icp_p2p(D, M, k, 1e-5, fixed=True)["moved"]
        ps = oracle_lib.soa(pt)
        idx, _ = mkl_matching(mkl, ps, qs, n, n, range(n), np.float64)
        mn = mkl_minimise(mkl, ps, qs, n, idx)
        new, E = mkl_transform_error(mkl, ps, qs, n, idx, mn["R"], mn["t"])
        out[f"synth_pt_{k}"] = np.ascontiguousarray(pt); out[f"synth_idx_{k}"] = idx
        for key in ("N", "S", "U", "Vt", "R", "t", "q_bar", "p_bar"):
            out[f"synth_{key}_{k}"] = mn[key]
        out[f"synth_new_{k}"] = oracle_lib.aos(new, n); out[f"synth_E_{k}"] = np.array(E)

    # ---- E. the minimisation on the hall pair, widened to double (what src/ICP_CPU.c would do with that cloud), pass 0 ------
    n = D64.shape[0]
    ps, qs = oracle_lib.soa(D64), oracle_lib.soa(M64)
    idx, _ = mkl_matching(mkl, ps, qs, n, n, range(n), np.float64)
    mn = mkl_minimise(mkl, ps, qs, n, idx)
    out["hallmin_idx"] = idx
    for key in ("N", "S", "R", "t"):
        out[f"hallmin_{key}"] = mn[key]
    _, E = mkl_transform_error(mkl, ps, qs, n, idx, mn["R"], mn["t"])
    out["hallmin_E"] = np.array(E)

    dst = os.path.join(HERE, "mkl_vectors.npz")
    np.savez_compressed(dst, **out)
    print(f"wrote {dst}: {os.path.getsize(dst)} bytes, {len(out)} arrays")


if __name__ == "__main__":
    main()
