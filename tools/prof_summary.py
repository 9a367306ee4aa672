#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace --stats run: per-kernel stats + median gaps between consecutive kernels."""
import csv, glob, statistics as st, sys
d = sys.argv[1]
ks = glob.glob(d + "/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(ks)):
    print(r["Name"].split("(")[0][-40:].ljust(41), r["Calls"].rjust(5), f"{float(r['AverageNs'])/1e3:9.2f} us", r["Percentage"].rjust(7), f"min {int(r['MinNs'])/1e3:7.2f} max {int(r['MaxNs'])/1e3:7.2f}")
kt = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(kt)), key=lambda r: int(r["Start_Timestamp"]))
ev = [(r["Kernel_Name"].split("(")[0].split("::")[-1][:22], int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows]
ev = ev[len(ev) // 3:-30]
gaps = {}
for (n0, s0, e0), (n1, s1, e1) in zip(ev, ev[1:]):
    gaps.setdefault((n0, n1), []).append(s1 - e0)
for k, v in gaps.items():
    if len(v) > 5:
        print("gap", k, "n=", len(v), f"median {st.median(v)/1e3:.2f} us min {min(v)/1e3:.2f}")
