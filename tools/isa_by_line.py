#!/usr/bin/env python3
"""Static instruction mix of ONE kernel of a device-ISA listing, attributed to source lines (.loc directives; build the listing
with -gline-tables-only): where a kernel's instructions come from, by source region.

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -gline-tables-only --cuda-device-only -S -o /tmp/k.s csrc/icp_k_sparse.hip
    python3 tools/isa_by_line.py /tmp/k.s 'nn_match_sparseILi1ELb0ELb1ELb1ELi4' [bucket-size-in-lines = 25]

Counts are STATIC (one per instruction in the listing, not per execution): read them together with how often a region runs."""
import re, sys, collections
path, pat = sys.argv[1], sys.argv[2]
bucket = int(sys.argv[3]) if len(sys.argv) > 3 else 25
files = {}
kind = lambda op: ("valu" if op.startswith("v_") else "salu" if op.startswith("s_") and not op.startswith(("s_waitcnt", "s_nop", "s_barrier", "s_cbranch", "s_branch", "s_load", "s_buffer")) else
                   "lds" if op.startswith("ds_") else "vmem" if op.startswith(("global_", "buffer_", "flat_", "scratch_")) else
                   "smem" if op.startswith(("s_load", "s_buffer")) else "branch" if op.startswith(("s_cbranch", "s_branch")) else "other")
inside = False
cur = (0, 0)
tally = collections.defaultdict(collections.Counter)
total = collections.Counter()
for ln in open(path):
    m = re.match(r"\s*\.file\s+(\d+)\s+\"[^\"]*\"\s+\"([^\"]+)\"", ln)
    if m:
        files[int(m.group(1))] = m.group(2)
        continue
    if not inside:
        if re.match(r"^_ZN\S*%s\S*:" % re.escape(pat), ln):
            inside = True
        continue
    if ln.startswith(".Lfunc_end"):
        break
    m = re.match(r"\s*\.loc\s+(\d+)\s+(\d+)", ln)
    if m:
        cur = (int(m.group(1)), int(m.group(2)))
        continue
    m = re.match(r"\s+([a-z][a-z0-9_]+)\s", ln)
    if m and not ln.lstrip().startswith((".", ";")):
        k = kind(m.group(1))
        tally[(cur[0], cur[1] // bucket * bucket)][k] += 1
        total[k] += 1
print("total:", dict(total))
for (f, l0), c in sorted(tally.items()):
    if sum(c.values()) >= 8:
        print(f"{files.get(f, f):24s} {l0:5d}-{l0 + bucket - 1:5d}  valu {c['valu']:5d} salu {c['salu']:4d} lds {c['lds']:4d} vmem {c['vmem']:3d} smem {c['smem']:3d} branch {c['branch']:3d}")
