#!/usr/bin/env python3
"""Per-dispatch durations of the matching kernel in a rocprofv3 --kernel-trace run (shows how the kernel time
moves over the iterations of one registration): trace_seq.py <dir> [count] [tail: the LAST count dispatches instead of the middle ones]"""
import csv, glob, sys
d = sys.argv[1]
cnt = int(sys.argv[2]) if len(sys.argv) > 2 else 60
kt = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(kt)), key=lambda r: int(r["Start_Timestamp"]))
ev = [(r["Kernel_Name"].split("(")[0].split("::")[-1][:24], int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows]
start = max(0, len(ev) - cnt) if len(sys.argv) > 3 and sys.argv[3] == "tail" else len(ev) // 2
prev_end = None
for n, s, e in ev[start:start + cnt]:
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    print(f"{n:26s} dur {(e - s)/1e3:7.2f} us   gap {gap:7.2f} us")
    prev_end = e
