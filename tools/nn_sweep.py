#!/usr/bin/env python3
"""Time the matching kernel alone under different launch knobs (one subprocess per setting: the knobs are
read once per process).  usage: python tools/nn_sweep.py [hall|bunny|grid128|big] ...   (GPU box only)"""
import itertools
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import os, sys, json, zlib, numpy as np
SEED = bool(int(os.environ.get('SWEEP_SEEDED', '1')))
sys.path.insert(0, %(root)r)
from __graft_entry__ import load_package
pkg = load_package()
name = %(name)r
g = os.path.join(%(root)r, "tests", "golden")
ctx = pkg.Context(0)
if name == "hall":
    r = np.fromfile(os.path.join(g, "hall_ranges_u32.bin"), dtype=np.uint32)
    alt, az = pkg.datasets.read_os1_intrinsics(os.path.join(g, "beam_intrinsics.csv"))
    P, Q = pkg.datasets.hall_clouds(ctx, r, 33616, alt, az)
elif name == "bunny":
    P = np.fromfile(os.path.join(g, "bunny_xyz_f32.bin"), dtype=np.float32).reshape(-1, 3)
    Q = pkg.datasets.make_model_gpu(P, *pkg.datasets.BUNNY)
elif name == "bunny_res":
    P = np.fromfile(os.path.join(g, "bunny_res_xyz_f32.bin"), dtype=np.float32).reshape(-1, 3)
    Q = pkg.datasets.make_model_gpu(P, *pkg.datasets.BUNNY)
elif name == "random":
    rng = np.random.default_rng(5)
    Q = rng.uniform(-1, 1, (16384, 3)).astype(np.float32)
    P = (Q[rng.permutation(16384)] + 0.01 * rng.standard_normal((16384, 3))).astype(np.float32)
elif name == "grid128":
    P = pkg.datasets.synthetic_grid(128, np.float32)
    Q = pkg.datasets.make_model_gpu(P, *pkg.datasets.P2P_GPU)
else:
    P = pkg.datasets.synthetic_grid(512, np.float32)   # 262144 points
    Q = pkg.datasets.make_model_gpu(P, *pkg.datasets.P2P_GPU)
if %(f64)d:
    P, Q = P.astype(np.float64), Q.astype(np.float64)
ctx.set_model(Q); ctx.set_moving(P)
ctx.nn_match_resident()
idx = ctx.get_indices()
reps = 30 if P.shape[0] < 100000 else 3
ctx.nn_match_bench(3, seeded=SEED)
ms = min(ctx.nn_match_bench(reps, seeded=SEED) / reps for _ in range(3))
info = ctx.nn_launch_info()
print(json.dumps(dict(us=1e3 * ms, crc=zlib.crc32(idx.tobytes()), n=int(P.shape[0]), m=int(Q.shape[0]), **info)))
'''


def run(name, env, f64=0):
    e = dict(os.environ)
    e.update({k: str(v) for k, v in env.items()})
    out = subprocess.run([sys.executable, "-c", CHILD % dict(root=ROOT, name=name, f64=f64)], env=e, capture_output=True, text=True)
    if out.returncode != 0:
        return dict(error=out.stderr[-300:])
    return json.loads(out.stdout.strip().splitlines()[-1])


def main():
    names = sys.argv[1:] or ["hall"]
    for name in names:
        base = None
        settings = [dict(ICP_NN_V1=1), dict(ICP_NN_CULL=0)]
        for C, bpc, seeded in itertools.product((8, 16), (6, 8), (0, 1)):
            settings.append(dict(ICP_NN_CULL=1, ICP_NN_CHUNK=C, ICP_NN_BLOCKS_PER_CU=bpc, SWEEP_SEEDED=seeded))
        for env in settings:
            r = run(name, env)
            if "error" in r:
                print(name, env, "ERROR", r["error"], flush=True)
                continue
            if base is None:
                base = r["crc"]
            pairs = r["n"] * r["m"]
            print(f"{name:8s} {str(env):70s} {r['us']:9.1f} us  {pairs / r['us'] / 1e6:6.2f} Tpairs/s  splits={r['splits']:3d} blocks={r['blocks']:5d} "
                  f"{'OK' if r['crc'] == base else 'MISMATCH'}", flush=True)


if __name__ == "__main__":
    main()
