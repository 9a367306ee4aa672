#!/usr/bin/env python3
"""Offline (CPU, numpy) statistics of wave-uniform early-out tests for the matching kernel: for every wave (128
consecutive moving points) x chunk (8 consecutive model points), could the chunk be skipped given each lane's FINAL
minimum as its bound, using only dx^2, dy^2, dz^2, dx^2+dy^2 ... as the lower bound?"""
import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib
orc = oracle_lib.Oracle()
g = os.path.join(ROOT, "tests", "golden")
name = sys.argv[1] if len(sys.argv) > 1 else "hall"
if name == "hall":
    P, Q = orc.hall_clouds(g)
elif name == "bunny":
    P = np.fromfile(os.path.join(g, "bunny_res_xyz_f32.bin"), dtype=np.float32).reshape(-1, 3)
    Q = orc.gpu_model_f32(P, (0.15, -0.1, 0.05), (0.01, -0.04, 0.02))
else:
    P = orc.synth_grid_f32(128); Q = orc.gpu_model_f32(P, (0.2, -0.2, 0.05), (0.8, -0.3, 0.2))
# after convergence the clouds nearly coincide: use the registered pose (P moved onto Q) as the steady state
if "--initial" not in sys.argv:
    r = orc.icp_p2p_f32x(P, Q, 100, 1e-6)
    P = r["moved"]
n, m = len(P), len(Q)
# void exact duplicates like the product does
_, first = np.unique(Q, axis=0, return_index=True)
keep = np.zeros(m, bool); keep[first] = True
# keep lowest index of each duplicate group
order = np.lexsort((np.arange(m), Q[:, 2], Q[:, 1], Q[:, 0]))
Qs = Q.copy(); Qs[~keep] = np.inf
idx = orc.nn(P, Q)
best = ((Q[idx] - P) ** 2).sum(1).astype(np.float32) * (1 + 1e-6)
W, C = 128, 8
nw, nc = n // W, m // C
tests = {"x": (1, 0, 0), "y": (0, 1, 0), "z": (0, 0, 1), "xy": (1, 1, 0), "xz": (1, 0, 1), "yz": (0, 1, 1), "xyz": (1, 1, 1)}
skip = {k: 0 for k in tests}
for wv in range(nw):
    p = P[wv * W:(wv + 1) * W]                       # (128,3)
    b = best[wv * W:(wv + 1) * W][:, None]
    d = (Qs[None, :, :] - p[:, None, :]) ** 2        # (128, m, 3)
    for k, (a, bb, c) in tests.items():
        part = a * d[:, :, 0] + bb * d[:, :, 1] + c * d[:, :, 2]   # inf*0 -> nan for voided; treat as skip-friendly
        part = np.where(np.isnan(part), np.inf, part)
        cmin = part.reshape(W, nc, C).min(2)         # (128, nc)
        need = (cmin < b).any(0)                     # any lane needs the chunk
        skip[k] += int((~need).sum())
tot = nw * nc
print(name, "waves", nw, "chunks", nc)
for k in tests:
    print(f"  bound {k:3s}: {100.0 * skip[k] / tot:6.2f}% of (wave, chunk) pairs skippable")

# ---- interval (per-chunk bounding box) lower bounds: gap = max(lo - p, p - hi, 0) per axis --------------------
Qc = np.where(np.isinf(Qs), np.nan, Qs).reshape(nc, C, 3)
lo = np.nanmin(Qc, axis=1); hi = np.nanmax(Qc, axis=1)          # (nc, 3); all-void chunks -> nan -> always skippable
lo = np.where(np.isnan(lo), np.inf, lo); hi = np.where(np.isnan(hi), -np.inf, hi)
skipb = {k: 0 for k in tests}
for wv in range(nw):
    p = P[wv * W:(wv + 1) * W]
    b = best[wv * W:(wv + 1) * W][:, None]
    gap = np.maximum(np.maximum(lo[None, :, :] - p[:, None, :], p[:, None, :] - hi[None, :, :]), 0.0)   # (128, nc, 3)
    g2 = gap * gap
    for k, (a, bb, c) in tests.items():
        L = a * g2[:, :, 0] + bb * g2[:, :, 1] + c * g2[:, :, 2]
        L = np.where(np.isnan(L), np.inf, L)
        need = (L < b).any(0)
        skipb[k] += int((~need).sum())
print("  interval (chunk AABB) bounds:")
for k in tests:
    print(f"  box   {k:3s}: {100.0 * skipb[k] / tot:6.2f}% skippable")
