#!/usr/bin/env python3
"""What a FRESH pair costs: icp_set_model, icp_set_moving and a whole icp_point_to_point call on host buffers,
for a pair the context has not seen (the moving cloud against a differently moved copy, another copy every repeat).
    python3 tools/setup_time.py [hall|bunny|bunny_res|grid128] [repeats = 20]
Prints min / median of each, the reference's method (src/CUDA/Matching_opt.cu:213-226: after 2 warm-ups)."""
import os, sys, time, statistics, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package
pkg = load_package()
which = sys.argv[1] if len(sys.argv) > 1 else "hall"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
g = os.path.join(ROOT, "tests", "golden")
with pkg.Context(0) as ctx:
    if which == "hall":
        r = np.fromfile(os.path.join(g, "hall_ranges_u32.bin"), dtype=np.uint32)
        alt, az = pkg.datasets.read_os1_intrinsics(os.path.join(g, "beam_intrinsics.csv"))
        P, _ = pkg.datasets.hall_clouds(ctx, r, 33616, alt, az)
        ang, t = pkg.datasets.HALL_MM; t = tuple(x / 1000.0 for x in t); max_iter = 100
    elif which in ("bunny", "bunny_res"):
        P = np.fromfile(os.path.join(g, "bunny_xyz_f32.bin" if which == "bunny" else "bunny_res_xyz_f32.bin"), dtype=np.float32).reshape(-1, 3)
        ang, t = pkg.datasets.BUNNY; max_iter = 100
    else:
        P = pkg.datasets.synthetic_grid(128, np.float32); ang, t = pkg.datasets.P2P_GPU; max_iter = 40
    models = [pkg.datasets.make_model_gpu(P, tuple(a * (1 + 0.06 * (k + 1)) for a in ang), tuple(x * (1 - 0.05 * (k + 1)) for x in t)) for k in range(3)]
    tm, tp, tw, its = [], [], [], []
    for rr in range(-2, reps):
        Q = models[(rr + 2) % 3]
        t0 = time.perf_counter(); ctx.set_model(Q); t1 = time.perf_counter(); ctx.set_moving(P); t2 = time.perf_counter()
        t3 = time.perf_counter(); res = ctx.point_to_point(P, Q, max_iter=max_iter, tol=1e-6); t4 = time.perf_counter()
        if rr >= 0:
            tm.append(t1 - t0); tp.append(t2 - t1); tw.append(t4 - t3); its.append(res.passes)
    f = lambda v: f"min {1e3 * min(v):.3f} ms, median {1e3 * statistics.median(v):.3f} ms"
    print(f"{which}: {len(P)} points, {reps} fresh pairs after 2 warm-ups")
    print(f"  icp_set_model   {f(tm)}")
    print(f"  icp_set_moving  {f(tp)}")
    print(f"  icp_point_to_point on host buffers (set-up + registration + indices and cloud back), {statistics.median(its):.0f} passes: {f(tw)}")
    print(f"  ... of which set-up (icp_result.seconds_setup of the last call) {1e3 * res.seconds_setup:.3f} ms, loop {1e3 * res.seconds_total:.3f} ms")
