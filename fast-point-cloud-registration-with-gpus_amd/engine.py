"""Host-side mirror of the reference's ICP interface on top of the C ABI (numpy in / numpy out).

The reference exposes its path only through `main()` programs; the seams are its kernel
signatures and loops.  `Context` wraps one `icp_ctx` (one HIP device) and offers those seams with
the reference's names and argument meaning:

    Matching(P, Q) -> idx                 src/CUDA/GPU_point_to_point_real.cu:38-79 / src/ICP_CPU.c:220-234
    point_to_point(D, M, ...) -> Result   src/ICP_point_to_point.cu:295-423 / src/ICP_CPU.c:217-271
    point_to_plane(D, M, ...) -> Result   src/ICP_point_to_plane.cu:517-631

Clouds are (N, 3) arrays, float32 or float64 (the dtype selects the device arithmetic), i.e. the
reference's AoS "xyzxyz" GPU layout.  No arithmetic happens here.
"""
import ctypes as C
from dataclasses import dataclass, field

import numpy as np

from . import _capi as capi


@dataclass
class Result:
    T: np.ndarray            # (4, 4) composed transform, moving -> model
    iterations: int          # the reference's loop counter at exit
    passes: int              # matching passes that contributed to T
    err: np.ndarray          # err[0] = 0, err[k] = RMS error after pass k-1 (length passes + 1)
    idx: np.ndarray          # (N,) int32 correspondences of the last contributing pass
    moved: np.ndarray        # (N, 3) final moving cloud
    seconds_total: float = 0.0
    seconds_nn: float = 0.0
    seconds_host: float = 0.0    # host half of the iterations (error, stop rule, solve), summed; 0 unless profiling is on
    seconds_setup: float = 0.0   # icp_set_model (+ normals) + icp_set_moving inside the call
    extra: dict = field(default_factory=dict)


def _as_cloud(a, dtype=None):
    a = np.asarray(a)
    if dtype is None:
        dtype = a.dtype if a.dtype in (np.float32, np.float64) else np.float32
    a = np.ascontiguousarray(a, dtype=dtype)
    if a.ndim != 2 or a.shape[1] != 3:
        raise ValueError("a cloud is an (N, 3) array")
    return a


def _prec(dtype):
    return capi.ICP_F64 if np.dtype(dtype) == np.float64 else capi.ICP_F32


class Context:
    """One `icp_ctx`: a HIP device, its stream and the HBM-resident clouds."""

    def __init__(self, device=0):
        self._lib = capi.load()
        h = C.c_void_p()
        capi.check(self._lib.icp_create(int(device), C.byref(h)), "icp_create")
        self._h = h
        self._dtype = None
        self._n = 0
        self._m = 0

    def close(self):
        if getattr(self, "_h", None):
            self._lib.icp_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # ---- plumbing ------------------------------------------------------------------------------
    def set_stream(self, hip_stream):
        capi.check(self._lib.icp_set_stream(self._h, C.c_void_p(hip_stream or 0)), "icp_set_stream")

    def set_exclusive(self, on=True):
        """the caller owns the device: hall-sized clouds run 16-wave blocks, one to a CU (same bits, ~4 % faster)"""
        capi.check(self._lib.icp_set_exclusive(self._h, 1 if on else 0), "icp_set_exclusive")

    def set_profiling(self, every_nth=1):
        """time every n-th matching launch of the loop with HIP events (0 / False = off)"""
        capi.check(self._lib.icp_set_profiling(self._h, int(every_nth)), "icp_set_profiling")

    # ---- multi-GPU: library-issued RCCL all-reduce ----------------------------------------------------
    @staticmethod
    def comm_unique_id():
        buf = (C.c_ubyte * 128)()
        capi.check(capi.load().icp_comm_unique_id(buf), "icp_comm_unique_id")
        return bytes(buf)

    def comm_init(self, id_bytes, rank, world):
        buf = (C.c_ubyte * 128).from_buffer_copy(id_bytes)
        capi.check(self._lib.icp_comm_init(self._h, buf, int(rank), int(world)), "icp_comm_init")

    def comm_destroy(self):
        capi.check(self._lib.icp_comm_destroy(self._h), "icp_comm_destroy")

    @staticmethod
    def comm_random_id():
        buf = (C.c_ubyte * 128)()
        capi.check(capi.load().icp_comm_random_id(buf), "icp_comm_random_id")
        return bytes(buf)

    def comm_init_local(self, id_bytes, rank, world):
        """ranks of ONE node: the loop's vector is exchanged through shared host memory (the resident loop stays on)"""
        buf = (C.c_ubyte * 128).from_buffer_copy(id_bytes)
        capi.check(self._lib.icp_comm_init_local(self._h, buf, int(rank), int(world)), "icp_comm_init_local")

    # ---- matching seam -------------------------------------------------------------------------
    def Matching(self, P, Q):
        """idx[i] = argmin_j |P_i - Q_j|^2, lowest j on ties (reference `Matching` kernel)."""
        P = _as_cloud(P)
        Q = _as_cloud(Q, P.dtype)
        idx = np.empty(P.shape[0], dtype=np.int32)
        fn = self._lib.icp_nn_match_f64 if P.dtype == np.float64 else self._lib.icp_nn_match_f32
        capi.check(fn(self._h, P.ctypes.data, P.shape[0], Q.ctypes.data, Q.shape[0], idx.ctypes.data), "icp_nn_match")
        self._dtype, self._n, self._m = P.dtype, P.shape[0], Q.shape[0]
        return idx

    nn_match = Matching

    # ---- resident clouds -----------------------------------------------------------------------
    def set_model(self, Q):
        Q = _as_cloud(Q)
        capi.check(self._lib.icp_set_model(self._h, Q.ctypes.data, Q.shape[0], _prec(Q.dtype)), "icp_set_model")
        self._dtype, self._m = Q.dtype, Q.shape[0]

    def set_moving(self, P):
        P = _as_cloud(P, self._dtype)
        capi.check(self._lib.icp_set_moving(self._h, P.ctypes.data, P.shape[0], _prec(P.dtype)), "icp_set_moving")
        self._dtype, self._n = P.dtype, P.shape[0]

    def set_model_normals(self, Nrm):
        Nrm = _as_cloud(Nrm, self._dtype)
        capi.check(self._lib.icp_set_model_normals(self._h, Nrm.ctypes.data, Nrm.shape[0]), "icp_set_model_normals")

    def reset_moving(self):
        capi.check(self._lib.icp_reset_moving(self._h), "icp_reset_moving")

    def get_moving(self):
        out = np.empty((self._n, 3), dtype=self._dtype)
        capi.check(self._lib.icp_get_moving(self._h, out.ctypes.data), "icp_get_moving")
        return out

    def get_indices(self):
        out = np.empty(self._n, dtype=np.int32)
        capi.check(self._lib.icp_get_indices(self._h, out.ctypes.data), "icp_get_indices")
        return out

    def nn_match_resident(self, timed=False):
        ms = C.c_float(0)
        capi.check(self._lib.icp_nn_match_resident(self._h, C.byref(ms) if timed else None), "icp_nn_match_resident")
        return ms.value

    def nn_match_bench(self, reps, seeded=True):
        ms = C.c_float(0)
        capi.check(self._lib.icp_nn_match_bench_ex(self._h, int(reps), 1 if seeded else 0, C.byref(ms)), "icp_nn_match_bench_ex")
        return ms.value

    def nn_match_bench_launches(self, reps=10, warmups=2, mode=0):
        """per-launch hipEvent durations [ms] of the matching kernel (reference method: Matching_opt.cu:213-226);
        mode 0 seeded, 1 cold, 2 the dense packed kernel that executes every pair"""
        out = np.zeros(int(reps), dtype=np.float32)
        capi.check(self._lib.icp_nn_match_bench_launches(self._h, int(reps), int(warmups), int(mode),
                                                         out.ctypes.data_as(C.POINTER(C.c_float))), "icp_nn_match_bench_launches")
        return out

    def nn_launch_info_ex(self, dense=False):
        v = [C.c_int(0) for _ in range(5)]
        capi.check(self._lib.icp_nn_launch_info_ex(self._h, 1 if dense else 0, *[C.byref(x) for x in v]), "icp_nn_launch_info_ex")
        return dict(zip(("splits", "blocks", "threads", "n_pad", "m_pad"), (x.value for x in v)))

    WORK_SLOTS = ("find_boxes", "upper_boxes", "hits_box", "hits_xy", "hits_full", "sample_groups", "block_passes", "block_transforms",
                  "spec_lists", "spec_covered", "spec_hits", "list_hits")

    def set_work_counting(self, enable=True):
        capi.check(self._lib.icp_set_work_counting(self._h, 1 if enable else 0), "icp_set_work_counting")

    def get_work_counters(self, reset=True):
        out = np.zeros(12, dtype=np.uint64)
        capi.check(self._lib.icp_get_work_counters(self._h, out.ctypes.data_as(C.POINTER(C.c_uint64)), 1 if reset else 0),
                   "icp_get_work_counters")
        return dict(zip(self.WORK_SLOTS, (int(x) for x in out)))

    def diag_row_roles(self, hits, min_part=2048, total_div=4096, control=True):
        """the roles of an ordered launch's blocks from its rows' hit counters, by the device kernels the loop uses (icp_diag_row_roles)"""
        h = np.ascontiguousarray(hits, dtype=np.uint32).copy()
        roles = np.zeros(h.size + 4096, dtype=np.int32)
        capi.check(self._lib.icp_diag_row_roles(self._h, h.ctypes.data_as(C.POINTER(C.c_uint32)), int(h.size), int(min_part), int(total_div), 1 if control else 0,
                                                roles.ctypes.data_as(C.POINTER(C.c_int32))), "icp_diag_row_roles")
        return roles, h

    def nn_launch_info(self):
        v = [C.c_int(0) for _ in range(5)]
        capi.check(self._lib.icp_nn_launch_info(self._h, *[C.byref(x) for x in v]), "icp_nn_launch_info")
        return dict(zip(("splits", "blocks", "threads", "n_pad", "m_pad"), (x.value for x in v)))

    def estimate_normals(self, want_neighbours=False):
        nrm = np.empty((self._m, 3), dtype=self._dtype)
        nbr = np.empty((self._m, 4), dtype=np.int32) if want_neighbours else None
        capi.check(self._lib.icp_estimate_normals(self._h, nrm.ctypes.data, nbr.ctypes.data if want_neighbours else None),
                   "icp_estimate_normals")
        return (nrm, nbr) if want_neighbours else nrm

    # ---- full loops ----------------------------------------------------------------------------
    def _run(self, metric, D, M, normals, max_iter, tol, fixed_iterations):
        D = _as_cloud(D)
        M = _as_cloud(M, D.dtype)
        n, m = D.shape[0], M.shape[0]
        prm = capi.icp_params(int(max_iter), float(tol), 1 if fixed_iterations else 0, _prec(D.dtype), metric)
        err = np.zeros(int(max_iter) + 1, dtype=np.float64)
        idx = np.zeros(n, dtype=np.int32)
        moved = np.zeros((n, 3), dtype=D.dtype)
        res = capi.icp_result()
        res.err = err.ctypes.data_as(C.POINTER(C.c_double))
        res.idx = idx.ctypes.data_as(C.POINTER(C.c_int32))
        res.moved = moved.ctypes.data
        if metric == capi.ICP_POINT_TO_PLANE:
            nptr = None
            if normals is not None:
                normals = _as_cloud(normals, D.dtype)
                nptr = normals.ctypes.data
            rc = self._lib.icp_point_to_plane(self._h, D.ctypes.data, n, M.ctypes.data, m, nptr, C.byref(prm), C.byref(res))
            capi.check(rc, "icp_point_to_plane")
        else:
            rc = self._lib.icp_point_to_point(self._h, D.ctypes.data, n, M.ctypes.data, m, C.byref(prm), C.byref(res))
            capi.check(rc, "icp_point_to_point")
        self._dtype, self._n, self._m = D.dtype, n, m
        return Result(T=np.array(res.T[:], dtype=np.float64).reshape(4, 4), iterations=res.iterations,
                      passes=res.passes, err=err[: res.passes + 1].copy(), idx=idx, moved=moved,
                      seconds_total=res.seconds_total, seconds_nn=res.seconds_nn,
                      seconds_host=res.seconds_host, seconds_setup=res.seconds_setup)

    def point_to_point(self, D, M, max_iter=40, tol=1e-6, fixed_iterations=False):
        return self._run(capi.ICP_POINT_TO_POINT, D, M, None, max_iter, tol, fixed_iterations)

    def point_to_plane(self, D, M, normals=None, max_iter=50, tol=1e-6, fixed_iterations=False):
        return self._run(capi.ICP_POINT_TO_PLANE, D, M, normals, max_iter, tol, fixed_iterations)

    # ---- step-wise loop (multi-GPU driver, per-iteration parity tests) -------------------------
    def loop_begin(self, metric=capi.ICP_POINT_TO_POINT, max_iter=40, tol=1e-6, fixed_iterations=False):
        prm = capi.icp_params(int(max_iter), float(tol), 1 if fixed_iterations else 0, _prec(self._dtype), metric)
        capi.check(self._lib.icp_loop_begin(self._h, C.byref(prm)), "icp_loop_begin")
        self._max_iter = int(max_iter)

    def loop_enqueue(self):
        capi.check(self._lib.icp_loop_enqueue(self._h), "icp_loop_enqueue")

    def loop_moments_dev(self):
        return self._lib.icp_loop_moments_dev(self._h)

    def loop_set_moments_dev(self, dev_ptr):
        capi.check(self._lib.icp_loop_set_moments_dev(self._h, C.c_void_p(dev_ptr or 0)), "icp_loop_set_moments_dev")

    def loop_complete(self):
        done = C.c_int(0)
        capi.check(self._lib.icp_loop_complete(self._h, C.byref(done)), "icp_loop_complete")
        return bool(done.value)

    def loop_run(self, max_steps):
        """up to max_steps iterations inside the library; returns (steps_done, done)"""
        k, d = C.c_int(0), C.c_int(0)
        capi.check(self._lib.icp_loop_run(self._h, int(max_steps), C.byref(k), C.byref(d)), "icp_loop_run")
        return k.value, bool(d.value)

    def recoveries(self):
        """registrations this context finished step-wise after a resident / armed pass never delivered its rows"""
        return int(self._lib.icp_recoveries(self._h))

    def loop_state(self):
        it, ps = C.c_int(0), C.c_int(0)
        err = np.zeros(self._max_iter + 1, dtype=np.float64)
        T = np.zeros(16, dtype=np.float64)
        capi.check(self._lib.icp_loop_state(self._h, C.byref(it), C.byref(ps), err.ctypes.data_as(C.POINTER(C.c_double)),
                                            err.size, T.ctypes.data_as(C.POINTER(C.c_double))), "icp_loop_state")
        return dict(iterations=it.value, passes=ps.value, err=err[: ps.value + 1].copy(), T=T.reshape(4, 4))

    def loop_timing(self):
        sec, cnt = C.c_double(0), C.c_int(0)
        capi.check(self._lib.icp_loop_timing(self._h, C.byref(sec), C.byref(cnt)), "icp_loop_timing")
        return sec.value, cnt.value

    def loop_phase_seconds(self):
        """(matching-kernel seconds, host-solve seconds) of the current / last loop; both 0 unless profiling is on"""
        a, b = C.c_double(0), C.c_double(0)
        capi.check(self._lib.icp_loop_phase_seconds(self._h, C.byref(a), C.byref(b)), "icp_loop_phase_seconds")
        return a.value, b.value

    def loop_timing_passes(self):
        n = C.c_longlong(0)
        capi.check(self._lib.icp_loop_timing_passes(self._h, C.byref(n)), "icp_loop_timing_passes")
        return n.value

    def loop_indices(self):
        out = np.empty(self._n, dtype=np.int32)
        capi.check(self._lib.icp_loop_indices(self._h, out.ctypes.data), "icp_loop_indices")
        return out

    # ---- hall ingest ---------------------------------------------------------------------------
    def os1_to_cartesian(self, ranges, encoder_count0, altitude16, azimuth16):
        r = np.ascontiguousarray(ranges, dtype=np.uint32)
        alt = np.ascontiguousarray(altitude16, dtype=np.float32)
        az = np.ascontiguousarray(azimuth16, dtype=np.float32)
        out = np.empty((r.size, 3), dtype=np.float32)
        pf = C.POINTER(C.c_float)
        capi.check(self._lib.icp_os1_to_cartesian(self._h, r.ctypes.data, r.size, int(encoder_count0),
                                                  alt.ctypes.data_as(pf), az.ctypes.data_as(pf), out.ctypes.data),
                   "icp_os1_to_cartesian")
        return out


    def os1_packets_to_cartesian(self, packets, altitude16, azimuth16):
        """raw packet bytes (uint8, n_packets * 12608) -> (xyz [mm] (N,3) float32, ranges (N,) uint32), decoded on the device"""
        pk = np.ascontiguousarray(packets, dtype=np.uint8).reshape(-1)
        assert pk.size % 12608 == 0
        npk = pk.size // 12608
        alt = np.ascontiguousarray(altitude16, dtype=np.float32)
        az = np.ascontiguousarray(azimuth16, dtype=np.float32)
        xyz = np.empty((npk * 256, 3), dtype=np.float32)
        rng = np.empty(npk * 256, dtype=np.uint32)
        pf = C.POINTER(C.c_float)
        capi.check(self._lib.icp_os1_packets_to_cartesian(self._h, pk.ctypes.data, npk, alt.ctypes.data_as(pf), az.ctypes.data_as(pf),
                                                          xyz.ctypes.data, rng.ctypes.data), "icp_os1_packets_to_cartesian")
        return xyz, rng


# ---- host-only helpers (no device) -------------------------------------------------------------
def solve_point_to_point(mom):
    lib = capi.load()
    mom = np.ascontiguousarray(mom, dtype=np.float64)
    R, t = np.zeros(9), np.zeros(3)
    pd = C.POINTER(C.c_double)
    capi.check(lib.icp_solve_point_to_point(mom.ctypes.data_as(pd), R.ctypes.data_as(pd), t.ctypes.data_as(pd)),
               "icp_solve_point_to_point")
    return R.reshape(3, 3), t


def solve_point_to_plane(mom):
    lib = capi.load()
    mom = np.ascontiguousarray(mom, dtype=np.float64)
    R, t, x = np.zeros(9), np.zeros(3), np.zeros(6)
    pd = C.POINTER(C.c_double)
    capi.check(lib.icp_solve_point_to_plane(mom.ctypes.data_as(pd), R.ctypes.data_as(pd), t.ctypes.data_as(pd),
                                            x.ctypes.data_as(pd)), "icp_solve_point_to_plane")
    return R.reshape(3, 3), t, x


def shard_range(n, rank, world):
    lib = capi.load()
    b, c = C.c_int64(0), C.c_int64(0)
    capi.check(lib.icp_shard_range(int(n), int(rank), int(world), C.byref(b), C.byref(c)), "icp_shard_range")
    return b.value, c.value


def share_rows_plan(hits, blocks, model_points, min_hits=64):
    """how a launch of `blocks` matching blocks deals itself to the rows whose hit counts of the previous launch are `hits`
    (icp_share_rows_plan): returns (parts per row, target hits per block)"""
    lib = capi.load()
    hits = np.ascontiguousarray(hits, dtype=np.uint32)
    parts = np.zeros(hits.shape[0], dtype=np.int32)
    target = C.c_uint32(0)
    capi.check(lib.icp_share_rows_plan(hits.ctypes.data_as(C.POINTER(C.c_uint32)), int(hits.shape[0]), int(blocks), int(model_points), int(min_hits),
                                       parts.ctypes.data_as(C.POINTER(C.c_int32)), C.byref(target)), "icp_share_rows_plan")
    return parts, int(target.value)


def eigh3(A):
    lib = capi.load()
    A = np.ascontiguousarray(A, dtype=np.float64).reshape(9)
    w, Z = np.zeros(3), np.zeros(9)
    pd = C.POINTER(C.c_double)
    capi.check(lib.icp_eigh3(A.ctypes.data_as(pd), w.ctypes.data_as(pd), Z.ctypes.data_as(pd)), "icp_eigh3")
    return w, Z.reshape(3, 3)


class LocalComm:
    """icp_lcomm_*: the host-memory communicator on its own (no device): sum of <= 32 doubles over the ranks of a node,
    added in rank order on every rank."""

    def __init__(self, id_bytes, rank, world):
        self._lib = capi.load()
        self._h = C.c_void_p()
        buf = (C.c_ubyte * 128).from_buffer_copy(id_bytes)
        capi.check(self._lib.icp_lcomm_create(buf, int(rank), int(world), C.byref(self._h)), "icp_lcomm_create")

    def allreduce(self, v):
        v = np.ascontiguousarray(v, dtype=np.float64).copy()
        capi.check(self._lib.icp_lcomm_allreduce(self._h, v.ctypes.data_as(C.POINTER(C.c_double)), int(v.size)), "icp_lcomm_allreduce")
        return v

    def close(self):
        if self._h:
            self._lib.icp_lcomm_destroy(self._h)
            self._h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()
