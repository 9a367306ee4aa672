"""Bunny.csv: ONE fresh context in a warm process, three registrations, wall time of each (run under rocprofv3 --kernel-trace and
read the tail of the trace with tools/trace_seq.py <dir> 120 tail: what a context's first registration is made of)."""
import os, sys, time, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package
pkg = load_package()
g = os.path.join(ROOT, "tests", "golden")
B = np.fromfile(os.path.join(g, "bunny_xyz_f32.bin"), dtype=np.float32).reshape(-1, 3)
BM = pkg.datasets.make_model_gpu(B, *pkg.datasets.BUNNY)
def reg(ctx):
    ctx.reset_moving(); ctx.loop_begin(pkg.ICP_POINT_TO_POINT, max_iter=100, tol=1e-6)
    t0 = time.perf_counter(); it, d = ctx.loop_run(1 << 20); dt = time.perf_counter() - t0
    return it, 1e6 * dt
with pkg.Context(0) as warm:
    warm.set_model(BM); warm.set_moving(B)
    for _ in range(3): reg(warm)
with pkg.Context(0) as ctx:
    ctx.set_model(BM); ctx.set_moving(B)
    for k in range(3):
        it, us = reg(ctx)
        print(f"registration {k + 1}: {it} iterations, {us:.1f} us = {us / it:.2f} us per iteration", flush=True)
