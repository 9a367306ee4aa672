#!/usr/bin/env python3
"""Stand-alone matching kernel, sparse vs dense, seeded vs cold, on the four clouds; the index CRCs must agree.
usage: python tools/nn_compare.py [hall bunny grid128 big]   (GPU box only; one subprocess per setting)"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import nn_sweep

names = sys.argv[1:] or ["hall", "bunny", "grid128", "big"]
print(f"{'cloud':8s} {'n':>7s} {'m':>7s}  {'kernel':7s} {'seeded us':>10s} {'cold us':>10s}  blocks x threads   crc")
for name in names:
    ref = None
    for label, env in (("sparse", {}), ("dense", {"ICP_NN_SPARSE": "0"})):
        rs = nn_sweep.run(name, dict(env, SWEEP_SEEDED=1))
        rc = nn_sweep.run(name, dict(env, SWEEP_SEEDED=0))
        if "error" in rs or "error" in rc:
            print(name, label, "failed", rs.get("error", rc.get("error"))); continue
        ref = ref or rs["crc"]
        ok = "ok" if rs["crc"] == ref and rc["crc"] == ref else "MISMATCH"
        print(f"{name:8s} {rs['n']:7d} {rs['m']:7d}  {label:7s} {rs['us']:10.2f} {rc['us']:10.2f}  {rs['blocks']} x {rs['threads']}   {rs['crc']:08x} {ok}")
