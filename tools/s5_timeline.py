#!/usr/bin/env python3
"""Timeline of the launches of a registration from a rocprofv3 kernel trace: per matching pass its duration, the gap to the next
matching launch and what ran in the gap -- the chain between two passes of configs[4].

    rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 tools/s5_time.py 0 8 30
    python3 tools/s5_timeline.py DIR [match-kernel-substring = nn_match_sparse]"""
import csv, glob, os, sys, collections
d = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else "nn_match_sparse"
files = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
rows = []
for f in files:
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
short = lambda n: n.replace("void icp::", "").replace("rocprim::detail::", "rp::").split("(")[0][:70]
match = [i for i, r in enumerate(rows) if pat in r[2]]
print(f"{len(rows)} launches, {len(match)} of them {pat}")
gaps, chain = [], collections.Counter()
for a, b in zip(match[:-1], match[1:]):
    s0, e0, _ = rows[a]
    s1, _, _ = rows[b]
    gap = (s1 - e0) / 1e3
    inside = rows[a + 1:b]
    busy = sum((e - s) for s, e, _ in inside) / 1e3
    gaps.append((gap, busy, (e0 - s0) / 1e3, len(inside)))
    for s, e, n in inside:
        chain[short(n)] += (e - s) / 1e3
import statistics
if gaps:
    late = gaps[len(gaps) // 2:]
    print("pass  match_us  gap_to_next_us  kernels_in_gap  busy_in_gap_us")
    for i, (g, bz, dur, k) in enumerate(gaps):
        print(f"{i:4d}  {dur:9.1f}  {g:9.1f}  {k:3d}  {bz:9.1f}")
    print(f"median gap {statistics.median(g for g, *_ in gaps):.1f} us (late passes {statistics.median(g for g, *_ in late):.1f}); "
          f"median busy in gap {statistics.median(b for _, b, *_ in gaps):.1f} us; sum match {sum(d for _, _, d, _ in gaps) / 1e3:.2f} ms, sum gaps {sum(g for g, *_ in gaps) / 1e3:.2f} ms")
    print("kernels between two matching launches, us per pass:")
    for n, t in chain.most_common(12):
        print(f"  {t / len(gaps):8.2f}  {n}")
