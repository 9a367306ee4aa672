#!/usr/bin/env python3
"""one line per bench.py record: python tools/bench_brief.py file.json [...]"""
import json, sys
for f in sys.argv[1:]:
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        r = d.get("roofline", {}); w = r.get("executed", {}).get("work_counters_one_registration", {})
        print(f"{f}: value {d['value']:.5g} {d['unit']}  ms/step {d['ms_per_step']:.5g}  err {d.get('final_rms_error', 0):.6g}  kernel avg {r.get('avg_launch_us', 0):.2f} us x {r.get('passes_per_launch', 0):.2f} passes  "
              f"frac {r.get('frac', 0):.4f}  hits box/xy/full {w.get('hits_box')}/{w.get('hits_xy')}/{w.get('hits_full')} samples {w.get('sample_groups')}")
    except Exception as e:  # noqa: BLE001
        print(f, "no record:", e)
