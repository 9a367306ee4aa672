#!/usr/bin/env python3
"""register / scratch / LDS use of every kernel in build/icp_k_*.s (optionally filtered by a substring)"""
import re
import subprocess
import sys

import glob
paths = sorted(glob.glob("fast-point-cloud-registration-with-gpus_amd/csrc/build/icp_k_*.s"))
flt = sys.argv[1] if len(sys.argv) > 1 else ""
s = "".join(open(p).read() for p in paths)
rows = []
for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", s, re.S):
    name, body = m.group(1), m.group(2)
    g = lambda k: re.search(r"\.amdhsa_%s (\d+)" % k, body).group(1)
    rows.append((name, g("next_free_vgpr"), g("next_free_sgpr"), g("private_segment_fixed_size"), g("group_segment_fixed_size")))
names = subprocess.run(["c++filt"], input="\n".join(r[0] for r in rows), capture_output=True, text=True).stdout.split("\n")
for (name, v, sg, sc, lds), dn in zip(rows, names):
    dn = re.sub(r"\(.*", "", dn)
    if flt in dn:
        print(f"{dn[:70]:70s} vgpr {v:>4s} sgpr {sg:>4s} scratch {sc:>5s} lds {lds:>6s}")
