"""ctypes binding of libicp_mi355x.so -- exactly the symbols include/icp_mi355x.h and include/icp_mi355x_diag.h declare.

This is the stub a maintainer of a Python host would write; it contains no
arithmetic.  Loading fails loudly when the shared library has not been built: there is no
fallback implementation of any entry point.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ICP_LIB_PATH") or os.path.join(_HERE, "libicp_mi355x.so")   # ICP_LIB_PATH: A/B runs of two builds

ICP_OK = 0
ICP_ERR_INVALID = -1
ICP_ERR_NO_DEVICE = -2
ICP_ERR_HIP = -3
ICP_ERR_EMPTY = -4
ICP_ERR_SINGULAR = -5
ICP_ERR_IO = -6
ICP_ERR_STATE = -7
ICP_ERR_NOMEM = -8

ICP_F32, ICP_F64 = 0, 1
ICP_POINT_TO_POINT, ICP_POINT_TO_PLANE = 0, 1
ICP_NMOM = 32
MOM_ERR, MOM_CNT, MOM_SP, MOM_SQ, MOM_SQP, MOM_SPP, MOM_SQQ, MOM_C, MOM_B = 0, 1, 2, 5, 8, 17, 18, 2, 23


class icp_params(C.Structure):
    _fields_ = [("max_iter", C.c_int), ("tol", C.c_double), ("fixed_iterations", C.c_int),
                ("precision", C.c_int), ("metric", C.c_int)]


class icp_result(C.Structure):
    _fields_ = [("T", C.c_double * 16), ("iterations", C.c_int), ("passes", C.c_int),
                ("err", C.POINTER(C.c_double)), ("idx", C.POINTER(C.c_int32)), ("moved", C.c_void_p),
                ("seconds_total", C.c_double), ("seconds_nn", C.c_double),
                ("seconds_host", C.c_double), ("seconds_setup", C.c_double)]


_vp, _i, _pi = C.c_void_p, C.c_int, C.POINTER(C.c_int)
_pd, _pf, _pi32, _pu32 = C.POINTER(C.c_double), C.POINTER(C.c_float), C.POINTER(C.c_int32), C.POINTER(C.c_uint32)

# name -> (restype, argtypes); every symbol of include/icp_mi355x.h and include/icp_mi355x_diag.h
SIGNATURES = {
    "icp_abi_version": (_i, []),
    "icp_strerror": (C.c_char_p, [_i]),
    "icp_last_error": (C.c_char_p, []),
    "icp_device_count": (_i, []),
    "icp_create": (_i, [_i, C.POINTER(_vp)]),
    "icp_destroy": (None, [_vp]),
    "icp_set_stream": (_i, [_vp, _vp]),
    "icp_set_profiling": (_i, [_vp, _i]),
    "icp_nn_match_f32": (_i, [_vp, _vp, _i, _vp, _i, _vp]),
    "icp_nn_match_f64": (_i, [_vp, _vp, _i, _vp, _i, _vp]),
    "icp_set_model": (_i, [_vp, _vp, _i, _i]),
    "icp_set_moving": (_i, [_vp, _vp, _i, _i]),
    "icp_set_model_normals": (_i, [_vp, _vp, _i]),
    "icp_reset_moving": (_i, [_vp]),
    "icp_get_moving": (_i, [_vp, _vp]),
    "icp_get_indices": (_i, [_vp, _vp]),
    "icp_nn_match_resident": (_i, [_vp, _pf]),
    "icp_nn_match_bench": (_i, [_vp, _i, _pf]),
    "icp_nn_match_bench_ex": (_i, [_vp, _i, _i, _pf]),
    "icp_nn_match_bench_launches": (_i, [_vp, _i, _i, _i, _pf]),
    "icp_nn_launch_info": (_i, [_vp, _pi, _pi, _pi, _pi, _pi]),
    "icp_nn_launch_info_ex": (_i, [_vp, _i, _pi, _pi, _pi, _pi, _pi]),
    "icp_set_work_counting": (_i, [_vp, _i]),
    "icp_recoveries": (_i, [_vp]),
    "icp_set_exclusive": (_i, [_vp, _i]),
    "icp_get_work_counters": (_i, [_vp, C.POINTER(C.c_uint64), _i]),
    "icp_estimate_normals": (_i, [_vp, _vp, _vp]),
    "icp_point_to_point": (_i, [_vp, _vp, _i, _vp, _i, C.POINTER(icp_params), C.POINTER(icp_result)]),
    "icp_point_to_plane": (_i, [_vp, _vp, _i, _vp, _i, _vp, C.POINTER(icp_params), C.POINTER(icp_result)]),
    "icp_loop_begin": (_i, [_vp, C.POINTER(icp_params)]),
    "icp_loop_enqueue": (_i, [_vp]),
    "icp_loop_moments_dev": (_vp, [_vp]),
    "icp_loop_set_moments_dev": (_i, [_vp, _vp]),
    "icp_loop_complete": (_i, [_vp, _pi]),
    "icp_loop_run": (_i, [_vp, _i, _pi, _pi]),
    "icp_loop_state": (_i, [_vp, _pi, _pi, _pd, _i, _pd]),
    "icp_loop_timing": (_i, [_vp, _pd, _pi]),
    "icp_loop_timing_passes": (_i, [_vp, C.POINTER(C.c_longlong)]),
    "icp_loop_phase_seconds": (_i, [_vp, _pd, _pd]),
    "icp_loop_indices": (_i, [_vp, _vp]),
    "icp_comm_unique_id": (_i, [_vp]),
    "icp_comm_init": (_i, [_vp, _vp, _i, _i]),
    "icp_comm_destroy": (_i, [_vp]),
    "icp_comm_random_id": (_i, [_vp]),
    "icp_comm_init_local": (_i, [_vp, _vp, _i, _i]),
    "icp_lcomm_create": (_i, [_vp, _i, _i, C.POINTER(_vp)]),
    "icp_lcomm_allreduce": (_i, [_vp, _pd, _i]),
    "icp_lcomm_destroy": (None, [_vp]),
    "icp_solve_point_to_point": (_i, [_pd, _pd, _pd]),
    "icp_solve_point_to_plane": (_i, [_pd, _pd, _pd, _pd]),
    "icp_host_loop_create": (_i, [C.POINTER(icp_params), C.POINTER(_vp)]),
    "icp_host_loop_destroy": (None, [_vp]),
    "icp_host_loop_advance": (_i, [_vp, _pd, _pi, _pd, _pd]),
    "icp_host_loop_note_applied": (_i, [_vp]),
    "icp_host_loop_state": (_i, [_vp, _pi, _pi, _pd, _i, _pd]),
    "icp_shard_range": (_i, [C.c_int64, _i, _i, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "icp_share_rows_plan": (_i, [_pu32, _i, _i, _i, _i, _pi32, _pu32]),
    "icp_diag_row_roles": (_i, [_vp, _pu32, _i, _i, _i, _i, _pi32]),
    "icp_eigh3": (_i, [_pd, _pd, _pd]),
    "icp_synthetic_grid_f32": (_i, [_i, C.c_float, C.c_float, _vp]),
    "icp_synthetic_grid_f64": (_i, [_i, C.c_double, C.c_double, _vp]),
    "icp_make_model_f32": (_i, [_vp, _i, _pf, _pf, _vp]),
    "icp_make_model_cpu_f64": (_i, [_vp, _i, _pd, _pd, _vp]),
    "icp_make_model_standard_f32": (_i, [_vp, _i, _vp]),
    "icp_read_xyz_text": (_i, [C.c_char_p, _vp, _i]),
    "icp_read_os1_ranges": (_i, [C.c_char_p, _vp, _i, _pu32]),
    "icp_read_os1_intrinsics": (_i, [C.c_char_p, _pf, _pf]),
    "icp_os1_to_cartesian": (_i, [_vp, _vp, _i, C.c_uint32, _pf, _pf, _vp]),
    "icp_os1_packets_to_cartesian": (_i, [_vp, _vp, _i, _pf, _pf, _vp, _vp]),
}

_lib = None


class IcpError(RuntimeError):
    def __init__(self, code, where, detail=""):
        self.code = code
        super().__init__(f"{where}: error {code}" + (f" ({detail})" if detail else ""))


def load():
    """dlopen libicp_mi355x.so and attach the signatures.  Raises if the library is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: build it with `python __graft_entry__.py build` "
                          "(hipcc --offload-arch=gfx950); there is no fallback implementation")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        if os.environ.get("ICP_LIB_PATH") and not hasattr(lib, name):
            continue             # an older build loaded for an A/B run: symbols added since are simply absent
        fn = getattr(lib, name)  # AttributeError here == header/library mismatch
        fn.restype = res
        fn.argtypes = args
    ABI = 2
    if lib.icp_abi_version() != ABI and not os.environ.get("ICP_LIB_PATH"):   # (an older build loaded for an A/B run has the shorter icp_result: nothing reads the new tail)
        raise ImportError(f"libicp_mi355x.so ABI version {lib.icp_abi_version()}, this stub binds version {ABI}")
    _lib = lib
    return lib


def check(code, where):
    if code != ICP_OK:
        lib = load()
        detail = lib.icp_strerror(code).decode()
        last = lib.icp_last_error().decode()
        raise IcpError(code, where, f"{detail}: {last}" if last else detail)
