#!/usr/bin/env python3
"""The dense packed kernel (executes every pair) on the hall pair under its launch knobs: min / mean of 10 launches after 2 warm-ups
(bench.py's `dense_kernel` figure), TFLOP/s at 8 flop per pair, index CRC against the first setting.
usage: python tools/dense_sweep.py [hall|bunny]   (GPU box only; one subprocess per setting: the knobs are read once per process)"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, json, zlib, numpy as np
sys.path.insert(0, %(root)r)
from __graft_entry__ import load_package
pkg = load_package()
g = os.path.join(%(root)r, "tests", "golden")
ctx = pkg.Context(0)
if %(name)r == "hall":
    r = np.fromfile(os.path.join(g, "hall_ranges_u32.bin"), dtype=np.uint32)
    alt, az = pkg.datasets.read_os1_intrinsics(os.path.join(g, "beam_intrinsics.csv"))
    P, Q = pkg.datasets.hall_clouds(ctx, r, 33616, alt, az)
else:
    P = np.fromfile(os.path.join(g, "bunny_xyz_f32.bin"), dtype=np.float32).reshape(-1, 3)
    Q = pkg.datasets.make_model_gpu(P, *pkg.datasets.BUNNY)
ctx.set_model(Q); ctx.set_moving(P)
ctx.nn_match_resident()            # (ICP_NN_SPARSE=0 in the environment: this IS the dense kernel + merge)
crc = zlib.crc32(ctx.get_indices().tobytes())
d = ctx.nn_match_bench_launches(10, 2, 2)
info = ctx.nn_launch_info_ex(dense=True)
print(json.dumps(dict(min_us=1e3 * float(d.min()), avg_us=1e3 * float(d.mean()), crc=crc, **info)))
'''
def run(name, env):
    e = dict(os.environ, ICP_NN_SPARSE="0")
    e.update({k: str(v) for k, v in env.items()})
    out = subprocess.run([sys.executable, "-c", CHILD % dict(root=ROOT, name=name)], env=e, capture_output=True, text=True)
    if out.returncode != 0:
        return dict(error=out.stderr[-400:])
    return json.loads(out.stdout.strip().splitlines()[-1])
def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "hall"
    settings = [dict()]
    extra = os.environ.get("DENSE_SWEEP")
    if extra:
        settings = [dict(kv.split("=") for kv in s.split(",") if kv) for s in extra.split(";")]
    else:
        for T in (2, 4):
            for S in (8, 16, 32):
                for C in (8, 16):
                    settings.append(dict(ICP_NN_T=T, ICP_NN_SPLITS=S, ICP_NN_CHUNK=C))
    base = None
    for env in settings:
        r = run(name, env)
        if "error" in r:
            print(env, "ERROR", r["error"], flush=True); continue
        base = base if base is not None else r["crc"]
        flop = 8.0 * r["n_pad"] * r["m_pad"]
        print(f"{name} {str(env):64s} min {r['min_us']:7.2f} avg {r['avg_us']:7.2f} us  {flop / r['avg_us'] / 1e6:6.2f} TFLOP/s  frac {flop / r['avg_us'] / 1e6 / 157.3:.3f}  "
              f"splits={r['splits']:3d} blocks={r['blocks']:5d} {'OK' if r['crc'] == base else 'MISMATCH'}", flush=True)
if __name__ == "__main__":
    main()
