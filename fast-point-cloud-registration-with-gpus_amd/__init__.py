"""MI355X-native ICP registration hot path (drop-in for the reference's icp_standard /
ICP_point_to_point / ICP_point_to_plane path).  The product is csrc/ -> libicp_mi355x.so (hand-written
gfx950 HIP kernels behind the C ABI of include/icp_mi355x.h); this package is the thin host mirror."""
from . import _capi as capi  # noqa: F401
from . import datasets  # noqa: F401
from . import distributed  # noqa: F401
from ._capi import ICP_F32, ICP_F64, ICP_NMOM, ICP_POINT_TO_PLANE, ICP_POINT_TO_POINT, IcpError, load  # noqa: F401
from .engine import Context, LocalComm, Result, eigh3, shard_range, share_rows_plan, solve_point_to_plane, solve_point_to_point  # noqa: F401
