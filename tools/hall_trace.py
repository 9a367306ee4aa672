"""hall pair: a few registrations with the device to itself; run with ICP_TRACE=2 for the per-pass lines (message -> first row -> all rows, host turnaround)
usage: ICP_TRACE=2 python tools/hall_trace.py [registrations=3]"""
import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package
pkg = load_package()
g = os.path.join(ROOT, "tests", "golden")
with pkg.Context(0) as ctx:
    r = np.fromfile(os.path.join(g, "hall_ranges_u32.bin"), dtype=np.uint32)
    alt, az = pkg.datasets.read_os1_intrinsics(os.path.join(g, "beam_intrinsics.csv"))
    P, Q = pkg.datasets.hall_clouds(ctx, r, 33616, alt, az)
    ctx.set_exclusive(True)
    ctx.set_model(Q); ctx.set_moving(P)
    for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
        ctx.reset_moving(); ctx.loop_begin(pkg.ICP_POINT_TO_POINT, max_iter=100, tol=1e-6)
        ctx.loop_run(1 << 20)
