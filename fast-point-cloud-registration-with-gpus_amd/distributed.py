"""Multi-GPU driver of the ICP loop: one process per GPU, the MOVING cloud sharded into contiguous ranges
(`icp_shard_range`), the model replicated, and exactly one collective per iteration -- an in-place SUM
all-reduce of the ICP_NMOM-double moment vector (RCCL when the tensors live on MI355X, gloo in the CPU
tests).  Every rank then solves the same 3x3 / 6x6 problem from bit-identical inputs, so R, t and the stop
decision agree without a broadcast.  The reference has no multi-GPU path: this is new design.

`run_sharded` is the device driver (used by bench.py for N > 1).  `HostLoop` + `drive` are the same loop over
abstract shard operations; the CPU test-suite runs them with oracle-backed shards over gloo to check the
partitioning, the vector layout and the stop logic by construction.
"""
import ctypes as C

import numpy as np

from . import _capi as capi
from .engine import shard_range


class HostLoop:
    """`icp_host_loop_*`: the host half of the loop (error series, stop rule, R/t solve, composed transform)."""

    def __init__(self, metric=capi.ICP_POINT_TO_POINT, max_iter=40, tol=1e-6, fixed_iterations=False, precision=capi.ICP_F32):
        self._lib = capi.load()
        self._max_iter = int(max_iter)
        prm = capi.icp_params(int(max_iter), float(tol), 1 if fixed_iterations else 0, int(precision), int(metric))
        h = C.c_void_p()
        capi.check(self._lib.icp_host_loop_create(C.byref(prm), C.byref(h)), "icp_host_loop_create")
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            self._lib.icp_host_loop_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()

    def advance(self, mom):
        mom = np.ascontiguousarray(mom, dtype=np.float64)
        assert mom.size == capi.ICP_NMOM
        done = C.c_int(0)
        R, t = np.zeros(9), np.zeros(3)
        pd = C.POINTER(C.c_double)
        capi.check(self._lib.icp_host_loop_advance(self._h, mom.ctypes.data_as(pd), C.byref(done), R.ctypes.data_as(pd),
                                                   t.ctypes.data_as(pd)), "icp_host_loop_advance")
        return bool(done.value), R.reshape(3, 3), t

    def note_applied(self):
        capi.check(self._lib.icp_host_loop_note_applied(self._h), "icp_host_loop_note_applied")

    def state(self):
        it, ps = C.c_int(0), C.c_int(0)
        err = np.zeros(self._max_iter + 1)
        T = np.zeros(16)
        pd = C.POINTER(C.c_double)
        capi.check(self._lib.icp_host_loop_state(self._h, C.byref(it), C.byref(ps), err.ctypes.data_as(pd), err.size,
                                                 T.ctypes.data_as(pd)), "icp_host_loop_state")
        return dict(iterations=it.value, passes=ps.value, err=err[: ps.value + 1].copy(), T=T.reshape(4, 4))


def shard(P, rank, world):
    """this rank's contiguous slice of the moving cloud"""
    b, c = shard_range(len(P), rank, world)
    return P[b:b + c], b


def shard_cyclic_index(n, rank, world, block=65536, order=None):
    """the indices of this rank's share when the cloud is dealt in blocks of `block` points, round-robin: rank r takes blocks r,
    r + world, ... of `order` (a permutation of the points; None: the cloud's own order).  A large cloud's work per point varies
    over the cloud (configs[4]: the contiguous eighths take 16 to 30 ms per registration); dealt in blocks the ranks' loads even out --
    provided a block is a COMPACT piece of the cloud (see curve_order: stripes of a row-major grid are not, and cost more than they
    balance).  Any partition of the moving points gives the same registration (the moment sums are sums over points)."""
    nb = (n + block - 1) // block
    # block b goes to rank (sum of b's digits in base `world`) mod world: a plain b mod world would hand a rank the SAME corner of every
    # cell of a space-filling order at every scale (the curve visits a cell's eight sub-cells in turn)
    bb = np.arange(nb, dtype=np.int64); owner = np.zeros(nb, dtype=np.int64); v = bb.copy()
    while world > 1 and (v > 0).any():
        owner += v % world; v //= world
    owner %= max(world, 1)
    parts = [np.arange(b * block, min((b + 1) * block, n), dtype=np.int64) for b in bb[owner == rank]]
    idx = np.concatenate(parts) if parts else np.zeros(0, dtype=np.int64)
    return idx if order is None else np.asarray(order)[idx]


def curve_order(P, bits=10, hilbert=True):
    """the points of a cloud along a space-filling curve over its bounding cube, `bits` (<= 10) per axis (Hilbert: consecutive points
    are neighbours, consecutive stretches compact pieces of the cloud; hilbert=False: Z-order, which jumps).  Host-side numpy, ~5 s for
    10 M points; the same on every rank (no randomness, stable sort)."""
    P = np.asarray(P, dtype=np.float32)
    bits = int(min(max(bits, 1), 10))
    lo = P.min(axis=0); ext = float((P.max(axis=0) - lo).max()) or 1.0
    q = np.clip(((P - lo) * np.float32((1 << bits) / ext)).astype(np.int32), 0, (1 << bits) - 1).astype(np.uint32)
    X = [np.ascontiguousarray(q[:, 0]), np.ascontiguousarray(q[:, 1]), np.ascontiguousarray(q[:, 2])]
    del q
    if hilbert:
        # Skilling's transform of the three coordinates into the "transpose" of the Hilbert index (AIP Conf. Proc. 707, 2004), vectorised
        Q = 1 << (bits - 1)
        while Q > 1:
            Pm = np.uint32(Q - 1)
            for i in range(3):
                hit = (X[i] & np.uint32(Q)) != 0
                if i == 0:
                    X[0] ^= hit.astype(np.uint32) * Pm                 # (not hit: X[0] is swapped with itself)
                else:
                    t = (X[0] ^ X[i]) & Pm
                    t *= (~hit).astype(np.uint32)                        # hit: invert X[0]'s low bits; else swap them with X[i]'s
                    X[0] ^= t; X[i] ^= t
                    X[0] ^= hit.astype(np.uint32) * Pm
            Q >>= 1
        X[1] ^= X[0]; X[2] ^= X[1]
        t = np.zeros_like(X[0]); Q = 1 << (bits - 1)
        while Q > 1:
            t ^= ((X[2] & np.uint32(Q)) != 0).astype(np.uint32) * np.uint32(Q - 1)
            Q >>= 1
        X = [X[2] ^ t, X[1] ^ t, X[0] ^ t]   # (X[0] carries the most significant bit of every triple: it goes to the top position)
    def spread(v):   # the low 10 bits of v to every third position
        v = v & np.uint32(0x3ff)
        v = (v | (v << np.uint32(16))) & np.uint32(0x030000ff)
        v = (v | (v << np.uint32(8))) & np.uint32(0x0300f00f)
        v = (v | (v << np.uint32(4))) & np.uint32(0x030c30c3)
        v = (v | (v << np.uint32(2))) & np.uint32(0x09249249)
        return v
    key = spread(X[0]) | (spread(X[1]) << np.uint32(1)) | (spread(X[2]) << np.uint32(2))
    return np.argsort(key, kind="stable")


def drive(shard_ops, host_loop, allreduce):
    """The loop over abstract shard operations.
    shard_ops.moments() -> local ICP_NMOM vector of the current pass (slot 0 = squared error of the motion
    applied last); shard_ops.apply(R, t) moves the local shard.  allreduce(vec) sums in place over ranks."""
    while True:
        mom = np.ascontiguousarray(shard_ops.moments(), dtype=np.float64)
        allreduce(mom)
        done, R, t = host_loop.advance(mom)
        if done:
            return host_loop.state()
        shard_ops.apply(R, t)
        host_loop.note_applied()


def attach_native_comm(ctx, dist):
    """Give `ctx` its own RCCL communicator (icp_comm_init): rank 0 draws the id, torch.distributed only carries
    those 128 bytes.  Afterwards icp_loop_enqueue issues the all-reduce itself."""
    rank, world = dist.get_rank(), dist.get_world_size()
    box = [ctx.comm_unique_id() if rank == 0 else None]
    dist.broadcast_object_list(box, src=0)
    ctx.comm_init(box[0], rank, world)


def attach_local_comm(ctx, dist):
    """Ranks of ONE node: give `ctx` the host-memory communicator (icp_comm_init_local).  The loop's 32-double vector is
    summed over the ranks through shared memory (~1 us) and icp_loop_run keeps its resident kernel."""
    rank, world = dist.get_rank(), dist.get_world_size()
    box = [ctx.comm_random_id() if rank == 0 else None]
    # the 128 bytes travel over a host-side (gloo) group where one can be made: nothing about this communicator
    # needs the device collective library to be up
    group = None
    if dist.get_backend() != "gloo":
        try:
            group = dist.new_group(backend="gloo")
        except Exception:  # noqa: BLE001
            group = None
    dist.broadcast_object_list(box, src=0, group=group)
    ctx.comm_init_local(box[0], rank, world)


def run_sharded_local(ctx, P_shard, Q, dist, metric=capi.ICP_POINT_TO_POINT, max_iter=40, tol=1e-6,
                      fixed_iterations=False, normals=None):
    """Single-node driver: every rank runs the library's own loop (resident kernel) on its shard, the ranks meet in
    shared host memory once per iteration."""
    ctx.set_model(Q)
    if metric == capi.ICP_POINT_TO_PLANE:
        if normals is not None:
            ctx.set_model_normals(normals)
        else:
            ctx.estimate_normals()
    ctx.set_moving(P_shard)
    attach_local_comm(ctx, dist)
    try:
        ctx.loop_begin(metric, max_iter=max_iter, tol=tol, fixed_iterations=fixed_iterations)
        done = False
        while not done:
            _, done = ctx.loop_run(1 << 20)
        st = ctx.loop_state()
        st["idx"] = ctx.loop_indices()
        st["moved"] = ctx.get_moving()
        return st
    finally:
        ctx.comm_destroy()


def run_sharded_native(ctx, P_shard, Q, dist, metric=capi.ICP_POINT_TO_POINT, max_iter=40, tol=1e-6,
                       fixed_iterations=False, normals=None):
    """Device driver with the collective issued by the library (no Python between the kernels and RCCL)."""
    ctx.set_model(Q)
    if metric == capi.ICP_POINT_TO_PLANE:
        if normals is not None:
            ctx.set_model_normals(normals)
        else:
            ctx.estimate_normals()
    ctx.set_moving(P_shard)
    attach_native_comm(ctx, dist)
    try:
        ctx.loop_begin(metric, max_iter=max_iter, tol=tol, fixed_iterations=fixed_iterations)
        while True:
            ctx.loop_enqueue()
            if ctx.loop_complete():
                break
        st = ctx.loop_state()
        st["idx"] = ctx.loop_indices()
        st["moved"] = ctx.get_moving()
        return st
    finally:
        ctx.comm_destroy()


def run_sharded(ctx, P_shard, Q, dist, metric=capi.ICP_POINT_TO_POINT, max_iter=40, tol=1e-6, fixed_iterations=False,
                normals=None):
    """Device driver.  `ctx` is this rank's Context, `dist` an initialised torch.distributed (backend nccl ==
    RCCL).  The loop's 32-double vector lives in a torch tensor and the context runs on torch's current stream,
    so the all-reduce is stream-ordered behind the finalize kernel; the host waits once per iteration."""
    import torch

    dev = torch.device("cuda", torch.cuda.current_device())
    mom = torch.zeros(capi.ICP_NMOM, dtype=torch.float64, device=dev)
    ctx.set_model(Q)
    if metric == capi.ICP_POINT_TO_PLANE:
        if normals is not None:
            ctx.set_model_normals(normals)
        else:
            ctx.estimate_normals()
    ctx.set_moving(P_shard)
    torch.cuda.synchronize()
    # a torch-owned NON-default stream, made torch's current stream for the duration of the loop: the context
    # launches on it and torch orders the collective against it.  (Handle 0 = the legacy default stream would be
    # taken by icp_set_stream as "use your own stream", and torch would then order the all-reduce against the
    # wrong stream.)
    side = torch.cuda.Stream(device=dev)
    ctx.set_stream(side.cuda_stream)
    ctx.loop_set_moments_dev(mom.data_ptr())
    try:
        with torch.cuda.stream(side):
            ctx.loop_begin(metric, max_iter=max_iter, tol=tol, fixed_iterations=fixed_iterations)
            while True:
                ctx.loop_enqueue()
                dist.all_reduce(mom)
                if ctx.loop_complete():
                    break
            st = ctx.loop_state()
            st["idx"] = ctx.loop_indices()
            st["moved"] = ctx.get_moving()
        return st
    finally:
        ctx.loop_set_moments_dev(0)
        ctx.set_stream(0)
