#!/usr/bin/env python3
"""Launch only the seeded matching kernel a few times on the hall cloud (target for rocprofv3 --pmc runs)."""
import os, sys, json, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package
pkg = load_package()
g = os.path.join(ROOT, "tests", "golden")
ctx = pkg.Context(0)
r = np.fromfile(os.path.join(g, "hall_ranges_u32.bin"), dtype=np.uint32)
alt, az = pkg.datasets.read_os1_intrinsics(os.path.join(g, "beam_intrinsics.csv"))
P, Q = pkg.datasets.hall_clouds(ctx, r, 33616, alt, az)
ctx.set_model(Q); ctx.set_moving(P)
ctx.nn_match_resident()
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
print("seeded us/launch:", 1e3 * ctx.nn_match_bench(reps, seeded=True) / reps)
print("cold   us/launch:", 1e3 * ctx.nn_match_bench(reps, seeded=False) / reps)
