"""Bunny.csv: what the FIRST registration of a fresh context costs per iteration (no hit counts to share the rows by yet),
in a process whose code objects are loaded (another context ran before), against the second, third and the steady ones."""
import os, sys, time, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package
pkg = load_package()
g = os.path.join(ROOT, "tests", "golden")
B = np.fromfile(os.path.join(g, "bunny_xyz_f32.bin"), dtype=np.float32).reshape(-1, 3)
BM = pkg.datasets.make_model_gpu(B, *pkg.datasets.BUNNY)
def regs(ctx, k):
    out = []
    for _ in range(k):
        ctx.reset_moving(); ctx.loop_begin(pkg.ICP_POINT_TO_POINT, max_iter=100, tol=1e-6)
        t0 = time.perf_counter(); it, d = ctx.loop_run(1 << 20); dt = time.perf_counter() - t0
        out.append(1e6 * dt / it)
    return out
with pkg.Context(0) as warm:
    warm.set_model(BM); warm.set_moving(B); regs(warm, 3)
for trial in range(3):
    with pkg.Context(0) as ctx:
        ctx.set_model(BM); ctx.set_moving(B)
        r = regs(ctx, 12)
        print(f"fresh context {trial}: us per iteration of registrations 1..12: " + " ".join(f"{v:.1f}" for v in r), flush=True)
