"""ctypes wrapper of oracle/liboracle.so -- the CPU restatement of the reference (TEST INFRASTRUCTURE:
only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this)."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB = os.path.join(ORACLE_DIR, "liboracle.so")


def build():
    subprocess.run(["make", "-C", ORACLE_DIR], check=True, stdout=subprocess.DEVNULL)


def soa(a):
    """(N,3) AoS -> the oracle's SoA (3N,) layout"""
    a = np.asarray(a)
    return np.ascontiguousarray(a.T).reshape(-1)


def aos(s, n):
    return np.ascontiguousarray(np.asarray(s).reshape(3, n).T)


class Oracle:
    def __init__(self):
        if not os.path.exists(LIB):
            build()
        self.lib = C.CDLL(LIB)
        self.lib.orc_icp_p2p_f64.restype = C.c_int
        self.lib.orc_icp_p2p_f32.restype = C.c_int
        self.lib.orc_icp_p2plane_f32.restype = C.c_int
        self.lib.orc_rms_error_f64.restype = C.c_double
        self.lib.orc_rms_error_f32.restype = C.c_float
        self.lib.orc_p2plane_minimize_f32.restype = C.c_int

    @staticmethod
    def _s(dtype):
        return "f64" if np.dtype(dtype) == np.float64 else "f32"

    # ---- matching -------------------------------------------------------------------------------
    def set_threads(self, k):
        """threads of the matching loops (default 1: the reference's scalar loop); results do not depend on it"""
        self.lib.orc_set_threads(int(k))

    def nn(self, P, Q):
        """P, Q (N,3)/(M,3) AoS of one dtype -> idx (N,) int32; first minimum of (dx^2+dy^2)+dz^2"""
        P = np.ascontiguousarray(P)
        Q = np.ascontiguousarray(Q, dtype=P.dtype)
        ps, qs = soa(P), soa(Q)
        idx = np.zeros(P.shape[0], dtype=np.int32)
        getattr(self.lib, "orc_nn_" + self._s(P.dtype))(ps.ctypes.data_as(C.c_void_p), P.shape[0], qs.ctypes.data_as(C.c_void_p),
                                                       Q.shape[0], idx.ctypes.data_as(C.c_void_p))
        return idx

    def p2p_minimize(self, P, Q, idx):
        P = np.ascontiguousarray(P)
        Q = np.ascontiguousarray(Q, dtype=P.dtype)
        ps, qs = soa(P), soa(Q)
        idx = np.ascontiguousarray(idx, dtype=np.int32)
        R = np.zeros(9, dtype=P.dtype)
        t = np.zeros(3, dtype=P.dtype)
        N = np.zeros(9, dtype=P.dtype)
        rc = getattr(self.lib, "orc_p2p_minimize_" + self._s(P.dtype))(
            ps.ctypes.data_as(C.c_void_p), P.shape[0], qs.ctypes.data_as(C.c_void_p), Q.shape[0],
            idx.ctypes.data_as(C.c_void_p), R.ctypes.data_as(C.c_void_p), t.ctypes.data_as(C.c_void_p),
            N.ctypes.data_as(C.c_void_p))
        assert rc == 0
        return R.reshape(3, 3), t, N.reshape(3, 3)

    def transform(self, P, R, t):
        P = np.ascontiguousarray(P)
        ps = soa(P).copy()
        R = np.ascontiguousarray(R, dtype=P.dtype).reshape(9)
        t = np.ascontiguousarray(t, dtype=P.dtype)
        getattr(self.lib, "orc_transform_" + self._s(P.dtype))(ps.ctypes.data_as(C.c_void_p), P.shape[0],
                                                              R.ctypes.data_as(C.c_void_p), t.ctypes.data_as(C.c_void_p))
        return aos(ps, P.shape[0])

    def rms_error(self, P, Q, idx):
        P = np.ascontiguousarray(P)
        Q = np.ascontiguousarray(Q, dtype=P.dtype)
        ps, qs = soa(P), soa(Q)
        idx = np.ascontiguousarray(idx, dtype=np.int32)
        return float(getattr(self.lib, "orc_rms_error_" + self._s(P.dtype))(
            ps.ctypes.data_as(C.c_void_p), P.shape[0], qs.ctypes.data_as(C.c_void_p), Q.shape[0], idx.ctypes.data_as(C.c_void_p)))

    def icp_p2p(self, D, M, max_iter, tol, fixed=False):
        D = np.ascontiguousarray(D)
        M = np.ascontiguousarray(M, dtype=D.dtype)
        n, m = D.shape[0], M.shape[0]
        ds, ms = soa(D), soa(M)
        E = np.zeros(max_iter + 1, dtype=D.dtype)
        T = np.zeros(16)
        idx = np.zeros(n, dtype=np.int32)
        pt = np.zeros(3 * n, dtype=D.dtype)
        passes = C.c_int(0)
        it = getattr(self.lib, "orc_icp_p2p_" + self._s(D.dtype))(
            ds.ctypes.data_as(C.c_void_p), ms.ctypes.data_as(C.c_void_p), n, m, int(max_iter), C.c_double(tol), 1 if fixed else 0,
            E.ctypes.data_as(C.c_void_p), T.ctypes.data_as(C.c_void_p), idx.ctypes.data_as(C.c_void_p),
            pt.ctypes.data_as(C.c_void_p), C.byref(passes))
        assert it >= 0
        return dict(iterations=it, passes=passes.value, err=E[: passes.value + 1].astype(np.float64), T=T.reshape(4, 4),
                    idx=idx, moved=aos(pt, n))

    def icp_p2p_f32x(self, D, M, max_iter, tol, fixed=False):
        """fp32 matching + fp64 minimisation (see oracle/icp_oracle.c, orc_icp_p2p_f32x)"""
        D = np.ascontiguousarray(D, dtype=np.float32)
        M = np.ascontiguousarray(M, dtype=np.float32)
        n, m = D.shape[0], M.shape[0]
        ds, ms = soa(D), soa(M)
        E = np.zeros(max_iter + 1, dtype=np.float64)
        T = np.zeros(16)
        idx = np.zeros(n, dtype=np.int32)
        pt = np.zeros(3 * n, dtype=np.float32)
        passes = C.c_int(0)
        self.lib.orc_icp_p2p_f32x.restype = C.c_int
        it = self.lib.orc_icp_p2p_f32x(ds.ctypes.data_as(C.c_void_p), ms.ctypes.data_as(C.c_void_p), n, m, int(max_iter), C.c_double(tol),
                                       1 if fixed else 0, E.ctypes.data_as(C.c_void_p), T.ctypes.data_as(C.c_void_p),
                                       idx.ctypes.data_as(C.c_void_p), pt.ctypes.data_as(C.c_void_p), C.byref(passes))
        assert it >= 0
        return dict(iterations=it, passes=passes.value, err=E[: passes.value + 1].copy(), T=T.reshape(4, 4), idx=idx, moved=aos(pt, n))

    # ---- synthetic clouds -----------------------------------------------------------------------
    def synth_icp_cpu(self, W):
        n = W * W
        D = np.zeros(3 * n)
        M = np.zeros(3 * n)
        self.lib.orc_synth_icp_cpu_f64(W, C.c_double(-2.0), C.c_double(2.0), D.ctypes.data_as(C.c_void_p), M.ctypes.data_as(C.c_void_p))
        return aos(D, n), aos(M, n)

    def synth_grid_f32(self, W):
        D = np.zeros((W * W, 3), dtype=np.float32)
        self.lib.orc_synth_grid_f32(W, C.c_float(-2.0), C.c_float(2.0), D.ctypes.data_as(C.c_void_p))
        return D

    def gpu_model_f32(self, D, angles, t):
        D = np.ascontiguousarray(D, dtype=np.float32)
        r = np.zeros(9, dtype=np.float32)
        self.lib.orc_rotation_gpu_f32(C.c_float(angles[0]), C.c_float(angles[1]), C.c_float(angles[2]), r.ctypes.data_as(C.c_void_p))
        tt = np.asarray(t, dtype=np.float32)
        M = np.zeros_like(D)
        self.lib.orc_apply_gpu_model_f32(r.ctypes.data_as(C.c_void_p), tt.ctypes.data_as(C.c_void_p), D.ctypes.data_as(C.c_void_p),
                                         D.shape[0], M.ctypes.data_as(C.c_void_p))
        return M

    def synth_icp_standard(self, W=32):
        D = np.zeros((W * W, 3), dtype=np.float32)
        M = np.zeros((W * W, 3), dtype=np.float32)
        self.lib.orc_synth_icp_standard_f32(W, D.ctypes.data_as(C.c_void_p), M.ctypes.data_as(C.c_void_p))
        return D, M

    # ---- point-to-plane -------------------------------------------------------------------------
    def knn4(self, Q):
        Q = np.ascontiguousarray(Q, dtype=np.float32)
        qs = soa(Q)
        nbr = np.zeros((Q.shape[0], 4), dtype=np.int32)
        self.lib.orc_knn4_f32(qs.ctypes.data_as(C.c_void_p), Q.shape[0], nbr.ctypes.data_as(C.c_void_p))
        return nbr

    def normals(self, Q, nbr):
        Q = np.ascontiguousarray(Q, dtype=np.float32)
        qs = soa(Q)
        nbr = np.ascontiguousarray(nbr, dtype=np.int32)
        nr = np.zeros(3 * Q.shape[0], dtype=np.float32)
        A = np.zeros((Q.shape[0], 9), dtype=np.float32)
        self.lib.orc_normals_f32(qs.ctypes.data_as(C.c_void_p), Q.shape[0], nbr.ctypes.data_as(C.c_void_p),
                                 nr.ctypes.data_as(C.c_void_p), A.ctypes.data_as(C.c_void_p))
        return aos(nr, Q.shape[0]), A

    def p2plane_minimize(self, P, Q, idx, normals, accumulate_f64=False):
        P = np.ascontiguousarray(P, dtype=np.float32)
        Q = np.ascontiguousarray(Q, dtype=np.float32)
        ps, qs, ns = soa(P), soa(Q), soa(np.asarray(normals, dtype=np.float32))
        idx = np.ascontiguousarray(idx, dtype=np.int32)
        R = np.zeros(9, dtype=np.float32)
        t = np.zeros(3, dtype=np.float32)
        Cm = np.zeros(36)
        b = np.zeros(6)
        rc = self.lib.orc_p2plane_minimize_f32(ps.ctypes.data_as(C.c_void_p), P.shape[0], qs.ctypes.data_as(C.c_void_p), Q.shape[0],
                                               idx.ctypes.data_as(C.c_void_p), ns.ctypes.data_as(C.c_void_p), 1 if accumulate_f64 else 0,
                                               R.ctypes.data_as(C.c_void_p), t.ctypes.data_as(C.c_void_p),
                                               Cm.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p))
        return rc, R.reshape(3, 3), t, Cm.reshape(6, 6), b

    def icp_p2plane(self, D, M, normals, max_iter, tol, fixed=False, accumulate_f64=False):
        D = np.ascontiguousarray(D, dtype=np.float32)
        M = np.ascontiguousarray(M, dtype=np.float32)
        n, m = D.shape[0], M.shape[0]
        ds, ms, ns = soa(D), soa(M), soa(np.asarray(normals, dtype=np.float32))
        E = np.zeros(max_iter + 1, dtype=np.float32)
        T = np.zeros(16)
        idx = np.zeros(n, dtype=np.int32)
        pt = np.zeros(3 * n, dtype=np.float32)
        passes = C.c_int(0)
        it = self.lib.orc_icp_p2plane_f32(ds.ctypes.data_as(C.c_void_p), ms.ctypes.data_as(C.c_void_p), n, m,
                                          ns.ctypes.data_as(C.c_void_p), int(max_iter), C.c_double(tol), 1 if fixed else 0,
                                          1 if accumulate_f64 else 0, E.ctypes.data_as(C.c_void_p), T.ctypes.data_as(C.c_void_p),
                                          idx.ctypes.data_as(C.c_void_p), pt.ctypes.data_as(C.c_void_p), C.byref(passes))
        assert it >= 0, it
        return dict(iterations=it, passes=passes.value, err=E[: passes.value + 1].astype(np.float64), T=T.reshape(4, 4),
                    idx=idx, moved=aos(pt, n))

    def icp_p2plane_f32x(self, D, M, normals, max_iter, tol, fixed=False):
        """fp32 matching + fp64 minimisation, point-to-plane (see oracle/icp_oracle.c, orc_icp_p2plane_f32x)"""
        D = np.ascontiguousarray(D, dtype=np.float32)
        M = np.ascontiguousarray(M, dtype=np.float32)
        n, m = D.shape[0], M.shape[0]
        ds, ms, ns = soa(D), soa(M), soa(np.asarray(normals, dtype=np.float32))
        E = np.zeros(max_iter + 1, dtype=np.float64)
        T = np.zeros(16)
        idx = np.zeros(n, dtype=np.int32)
        pt = np.zeros(3 * n, dtype=np.float32)
        passes = C.c_int(0)
        self.lib.orc_icp_p2plane_f32x.restype = C.c_int
        it = self.lib.orc_icp_p2plane_f32x(ds.ctypes.data_as(C.c_void_p), ms.ctypes.data_as(C.c_void_p), n, m,
                                           ns.ctypes.data_as(C.c_void_p), int(max_iter), C.c_double(tol), 1 if fixed else 0,
                                           E.ctypes.data_as(C.c_void_p), T.ctypes.data_as(C.c_void_p),
                                           idx.ctypes.data_as(C.c_void_p), pt.ctypes.data_as(C.c_void_p), C.byref(passes))
        assert it >= 0, it
        return dict(iterations=it, passes=passes.value, err=E[: passes.value + 1].copy(), T=T.reshape(4, 4), idx=idx, moved=aos(pt, n))

    def p2plane_minimize_f32x(self, P, Q, idx, normals):
        P = np.ascontiguousarray(P, dtype=np.float32)
        Q = np.ascontiguousarray(Q, dtype=np.float32)
        ps, qs, ns = soa(P), soa(Q), soa(np.asarray(normals, dtype=np.float32))
        idx = np.ascontiguousarray(idx, dtype=np.int32)
        R, t, Cm, b = np.zeros(9), np.zeros(3), np.zeros(36), np.zeros(6)
        self.lib.orc_p2plane_minimize_f32x.restype = C.c_int
        rc = self.lib.orc_p2plane_minimize_f32x(ps.ctypes.data_as(C.c_void_p), P.shape[0], qs.ctypes.data_as(C.c_void_p), Q.shape[0],
                                                idx.ctypes.data_as(C.c_void_p), ns.ctypes.data_as(C.c_void_p),
                                                R.ctypes.data_as(C.c_void_p), t.ctypes.data_as(C.c_void_p),
                                                Cm.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p))
        return rc, R.reshape(3, 3), t, Cm.reshape(6, 6), b

    # ---- hall ingest ----------------------------------------------------------------------------
    def os1_ranges_from_lines(self, lines, cap=16384):
        lines = np.ascontiguousarray(lines, dtype=np.int32)
        r = np.zeros(cap, dtype=np.float32)
        enc = C.c_ulong(0)
        self.lib.orc_os1_ranges_from_lines.restype = C.c_int
        n = self.lib.orc_os1_ranges_from_lines(lines.ctypes.data_as(C.c_void_p), lines.size, r.ctypes.data_as(C.c_void_p), cap, C.byref(enc))
        return r[: min(n, cap)], int(enc.value), n

    def os1_select_beams(self, alt64, az64):
        alt64 = np.ascontiguousarray(alt64, dtype=np.float64)
        az64 = np.ascontiguousarray(az64, dtype=np.float64)
        a, z = np.zeros(16, dtype=np.float32), np.zeros(16, dtype=np.float32)
        self.lib.orc_os1_select_beams(alt64.ctypes.data_as(C.c_void_p), az64.ctypes.data_as(C.c_void_p), a.ctypes.data_as(C.c_void_p), z.ctypes.data_as(C.c_void_p))
        return a, z

    def os1_conversion(self, r, enc, alt16, az16):
        r = np.ascontiguousarray(r, dtype=np.float32)
        out = np.zeros((r.size, 3), dtype=np.float32)
        a = np.ascontiguousarray(alt16, dtype=np.float32)
        z = np.ascontiguousarray(az16, dtype=np.float32)
        self.lib.orc_os1_conversion_f32(r.ctypes.data_as(C.c_void_p), r.size, C.c_ulong(enc), a.ctypes.data_as(C.c_void_p),
                                        z.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p))
        return out

    def hall_clouds(self, golden_dir):
        """the hall pair built entirely by the oracle from the committed fixtures (metres, fp32 AoS)"""
        r = np.fromfile(os.path.join(golden_dir, "hall_ranges_u32.bin"), dtype=np.uint32).astype(np.float32)
        import json
        enc = json.load(open(os.path.join(golden_dir, "hall_meta.json")))["encoder_count0"]
        vals = [float(x) for x in open(os.path.join(golden_dir, "beam_intrinsics.csv")).read().split() if x[0] in "-0123456789."]
        alt, az = self.os1_select_beams(vals[:64], vals[64:128])
        P_mm = self.os1_conversion(r, enc, alt, az)
        rmat = np.zeros(9, dtype=np.float32)
        self.lib.orc_rotation_gpu_f32(C.c_float(0.01), C.c_float(-0.003), C.c_float(0.05), rmat.ctypes.data_as(C.c_void_p))
        T = np.array([0.001, -0.0202, 0.02], dtype=np.float32)
        Q_mm = np.zeros_like(P_mm)
        self.lib.orc_ryt_f32(rmat.ctypes.data_as(C.c_void_p), T.ctypes.data_as(C.c_void_p), P_mm.ctypes.data_as(C.c_void_p), P_mm.shape[0],
                             Q_mm.ctypes.data_as(C.c_void_p))
        s = np.float32(1.0 / 1000.0)
        return (P_mm * s).astype(np.float32), (Q_mm * s).astype(np.float32)

    def read_xyz_text(self, path, cap=1 << 22):
        out = np.zeros(cap, dtype=np.float32)
        self.lib.orc_read_xyz_text.restype = C.c_int
        n = self.lib.orc_read_xyz_text(os.fsencode(path), out.ctypes.data_as(C.c_void_p), cap)
        assert n >= 0 and n % 3 == 0
        return out[:n].reshape(-1, 3).copy()
