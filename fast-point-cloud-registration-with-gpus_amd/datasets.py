"""The reference's inputs, produced through the C ABI: synthetic z = x^2 - y^2 grids and
their rigidly moved models, the Bunny text formats and the OS1-16 "hall" packet dump.  Constants are
the ones hard-coded in the reference programs (cited per function)."""
import ctypes as C
import os

import numpy as np

from . import _capi as capi

# (angles xyz [rad], translation) baked into the reference programs
P2P_GPU = ((0.2, -0.2, 0.05), (0.8, -0.3, 0.2))            # src/ICP_point_to_point.cu:157-165 (also ..._plane.cu)
CPU_F64 = ((1.0, -0.5, 0.05), (1.0, -0.3, 0.2))            # src/ICP_CPU.c:100-108
BUNNY = ((0.15, -0.1, 0.05), (0.01, -0.04, 0.02))          # src/CUDA/GPU_point_to_point_bunny.cu:136-144
HALL_MM = ((0.01, -0.003, 0.05), (0.001, -0.0202, 0.02))   # src/CUDA/GPU_point_to_point_real.cu:585-593 (applied in mm)


def synthetic_grid(W, dtype=np.float32, xy_min=-2.0, xy_max=2.0):
    """W*W points, point k*W+j = (lin[k], lin[j], x^2-y^2)  (src/ICP_point_to_point.cu:103-152, src/ICP_CPU.c:51-95)"""
    lib = capi.load()
    D = np.empty((W * W, 3), dtype=dtype)
    if np.dtype(dtype) == np.float64:
        capi.check(lib.icp_synthetic_grid_f64(W, xy_min, xy_max, D.ctypes.data), "icp_synthetic_grid_f64")
    else:
        capi.check(lib.icp_synthetic_grid_f32(W, xy_min, xy_max, D.ctypes.data), "icp_synthetic_grid_f32")
    return D


def make_model_gpu(D, angles, t):
    """M = R*D + t with the closed-form column-major R of the GPU programs, fp32."""
    lib = capi.load()
    D = np.ascontiguousarray(D, dtype=np.float32)
    M = np.empty_like(D)
    a = (C.c_float * 3)(*angles)
    tt = (C.c_float * 3)(*t)
    capi.check(lib.icp_make_model_f32(D.ctypes.data, D.shape[0], a, tt, M.ctypes.data), "icp_make_model_f32")
    return M


def make_model_cpu(D, angles=CPU_F64[0], t=CPU_F64[1]):
    """M = (rx*ry*rz)*D + t of src/ICP_CPU.c:100-149, fp64."""
    lib = capi.load()
    D = np.ascontiguousarray(D, dtype=np.float64)
    M = np.empty_like(D)
    a = (C.c_double * 3)(*angles)
    tt = (C.c_double * 3)(*t)
    capi.check(lib.icp_make_model_cpu_f64(D.ctypes.data, D.shape[0], a, tt, M.ctypes.data), "icp_make_model_cpu_f64")
    return M


def make_model_standard(D):
    """the hard-coded rotation of src/ICP_standard.cu:247-249, t = (1,-0.3,0.2)"""
    lib = capi.load()
    D = np.ascontiguousarray(D, dtype=np.float32)
    M = np.empty_like(D)
    capi.check(lib.icp_make_model_standard_f32(D.ctypes.data, D.shape[0], M.ctypes.data), "icp_make_model_standard_f32")
    return M


def read_xyz_text(path, cap_points=1 << 22):
    """Bunny_res.csv ("x y z") / Bunny.csv ("x;y;z"), CRLF tolerated."""
    lib = capi.load()
    n = lib.icp_read_xyz_text(os.fsencode(path), None, 0)
    if n < 0:
        capi.check(n, f"icp_read_xyz_text({path})")
    n = min(n, cap_points)
    out = np.empty((n, 3), dtype=np.float32)
    got = lib.icp_read_xyz_text(os.fsencode(path), out.ctypes.data, n)
    if got < 0:
        capi.check(got, f"icp_read_xyz_text({path})")
    return out


def read_os1_ranges(path):
    """ranges [mm] of the 16 lasers, scan order, + encoder count of the first azimuth block"""
    lib = capi.load()
    enc = C.c_uint32(0)
    n = lib.icp_read_os1_ranges(os.fsencode(path), None, 0, C.byref(enc))
    if n < 0:
        capi.check(n, f"icp_read_os1_ranges({path})")
    out = np.empty(n, dtype=np.uint32)
    got = lib.icp_read_os1_ranges(os.fsencode(path), out.ctypes.data, n, C.byref(enc))
    if got < 0:
        capi.check(got, f"icp_read_os1_ranges({path})")
    return out, int(enc.value)


def read_os1_intrinsics(path):
    lib = capi.load()
    alt = np.zeros(16, dtype=np.float32)
    az = np.zeros(16, dtype=np.float32)
    pf = C.POINTER(C.c_float)
    capi.check(lib.icp_read_os1_intrinsics(os.fsencode(path), alt.ctypes.data_as(pf), az.ctypes.data_as(pf)),
               f"icp_read_os1_intrinsics({path})")
    return alt, az


def hall_clouds(ctx, ranges, encoder_count0, altitude16, azimuth16):
    """The "hall" pair as src/CUDA/GPU_point_to_point_real.cu builds it: polar->Cartesian on the device
    (mm), model = R*P + T in mm (:585-606), then both scaled by 1/1000 (:169-171)."""
    P_mm = ctx.os1_to_cartesian(ranges, encoder_count0, altitude16, azimuth16)
    Q_mm = make_model_gpu(P_mm, *HALL_MM)
    s = np.float32(1.0 / 1000.0)
    return (P_mm * s).astype(np.float32), (Q_mm * s).astype(np.float32)
