#!/usr/bin/env python3
"""Offline model (numpy + scipy, no GPU) of the hit lists of the hierarchical search on configs[4] (10 M x 10 M):
how many chunks a row of 128 points lists today, how many of those survive the per-point box test, and what
(a) sub-group boxes in the list building, (b) a near-first first sub-round with the group bound, (c) both, would list.
usage: python tools/s5_hits_model.py [points=10000000] [rows=24]"""
import sys, time
import numpy as np
from scipy.spatial import cKDTree

N = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
ROWS = int(sys.argv[2]) if len(sys.argv) > 2 else 24
W = int(np.ceil(np.sqrt(N)))
lin = np.linspace(-2.0, 2.0, W).astype(np.float32)
X, Y = np.meshgrid(lin, lin, indexing="ij")
D = np.stack([X.ravel(), Y.ravel(), (X * X - Y * Y).ravel()], axis=1)[:N].astype(np.float32)
def rot(rx, ry, rz):
    cx, sx, cy, sy, cz, sz = np.cos(rx), np.sin(rx), np.cos(ry), np.sin(ry), np.cos(rz), np.sin(rz)
    Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]]); Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]]); Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    return Rz @ Ry @ Rx
R0 = rot(0.2, -0.2, 0.05); t0 = np.array([0.8, -0.3, 0.2])
M = (D.astype(np.float64) @ R0.T + t0).astype(np.float32)

def morton(P):
    lo, hi = P.min(0), P.max(0)
    q = np.clip(((P - lo) / (hi - lo).max() * 1023.0), 0, 1023).astype(np.uint64)
    def spread(v):
        v = (v | (v << 16)) & 0x030000FF; v = (v | (v << 8)) & 0x0300F00F; v = (v | (v << 4)) & 0x030C30C3; v = (v | (v << 2)) & 0x09249249
        return v
    return spread(q[:, 0]) | (spread(q[:, 1]) << 1) | (spread(q[:, 2]) << 2)
t = time.time()
Ms = M[np.argsort(morton(M), kind="stable")]
Ds = D[np.argsort(morton(D), kind="stable")]
mc = (Ms.shape[0] // 8) * 8
blo = Ms[:mc].reshape(-1, 8, 3).min(1); bhi = Ms[:mc].reshape(-1, 8, 3).max(1)
print(f"{N} points, {blo.shape[0]} chunks; order + boxes {time.time() - t:.1f} s", flush=True)
t = time.time()
tree = cKDTree(Ms)
print(f"kd-tree {time.time() - t:.1f} s", flush=True)

# the registration's transforms, from a subsample of the moving cloud against the whole model
sub = Ds[:: max(1, N // 100_000)].astype(np.float64)
Ts = [(np.eye(3), np.zeros(3))]
P = sub.copy()
for k in range(30):
    _, j = tree.query(P, workers=-1)
    Q = Ms[j].astype(np.float64)
    pc, qc = P.mean(0), Q.mean(0)
    U, S, Vt = np.linalg.svd((Q - qc).T @ (P - pc))
    R = U @ Vt; tt = qc - R @ pc
    P = P @ R.T + tt
    Rk, tk = Ts[-1]
    Ts.append((R @ Rk, R @ tk + tt))
    if k < 6 or k % 5 == 4: print(f"pass {k}: rms residual {np.sqrt(((Q - P) ** 2).sum(1).mean()):.4f}", flush=True)

def boxL(glo, ghi, lo, hi):
    g = np.maximum(np.maximum(lo - ghi, glo - hi), 0.0)
    return (g * g).sum(-1)

ccen = 0.5 * (blo + bhi)
ctree = cKDTree(ccen)
Mpts = Ms[:mc].reshape(-1, 8, 3)
rows = np.linspace(0, Ds.shape[0] // 128 - 1, ROWS).astype(int)
for k in (1, 2, 4, 8, 12, 16, 22, 29):
    Rk, tk = Ts[k]; Rp, tp = Ts[k - 1]
    tot = dict(now=0, now_pp=0, a=0, b=0, c=0, ideal=0, s1=0, d=0, d_pp=0, d_sg=0, ratio=0.0)
    for r in rows:
        p0 = Ds[r * 128:(r + 1) * 128].astype(np.float64)
        pk = (p0 @ Rk.T + tk).astype(np.float32); pprev = (p0 @ Rp.T + tp).astype(np.float32)
        _, js = tree.query(pprev)
        seed = Ms[js]
        bound = ((pk - seed) ** 2).sum(1)
        dtrue, _ = tree.query(pk); dtrue = dtrue ** 2
        glo, ghi = pk.min(0), pk.max(0); B = bound.max()
        L = boxL(glo, ghi, blo, bhi)
        cand = np.nonzero(L < B)[0]
        tot["now"] += cand.size
        # per-point box test with the starting bounds (an upper bound of what survives: the bounds shrink while the list is worked through)
        Lp = np.stack([boxL(pk[i], pk[i], blo[cand], bhi[cand]) for i in range(128)], 1)   # (cand, 128)
        tot["now_pp"] += int((Lp <= bound[None, :]).any(1).sum())
        tot["ideal"] += int((Lp <= dtrue[None, :] * 1.000001).any(1).sum())
        # (a) sub-groups of 16 points, each with its own box and bound
        sg = [(pk[g * 16:(g + 1) * 16].min(0), pk[g * 16:(g + 1) * 16].max(0), bound[g * 16:(g + 1) * 16].max()) for g in range(8)]
        pa = np.zeros(cand.size, bool)
        for lo_, hi_, b_ in sg: pa |= boxL(lo_, hi_, blo[cand], bhi[cand]) < b_
        tot["a"] += int(pa.sum())
        # (b) near-first: the chunks nearest to the group box first, then the rest against the new bounds
        dG = np.sqrt(((ghi - glo) ** 2).sum())
        Lc = L[cand]
        thr = (np.sqrt(Lc.min()) + 0.25 * dG) ** 2
        s1 = Lc <= thr
        pts = Ms[:mc].reshape(-1, 8, 3)[cand[s1]].reshape(-1, 3)
        d1 = ((pk[:, None, :] - pts[None, :, :]) ** 2).sum(2).min(1)
        b2 = np.minimum(bound, d1)
        tot["s1"] += int(s1.sum())
        tot["b"] += int(s1.sum() + ((Lc < b2.max()) & ~s1).sum())
        pc_ = np.zeros(cand.size, bool)
        for g in range(8):
            sl = slice(g * 16, (g + 1) * 16)
            pc_ |= boxL(pk[sl].min(0), pk[sl].max(0), blo[cand], bhi[cand]) < b2[sl].max()
        tot["c"] += int(s1.sum() + (pc_ & ~s1).sum())
        # (d) descent: for each of 16 sub-groups of 8 slots, the chunk whose box centre is nearest to the sub-group's middle point;
        # all 128 points are measured against those 16 chunks first, the list is built with the new bounds
        reps = pk[4::8]
        _, cj = ctree.query(reps)
        ptsd = Mpts[np.unique(cj)].reshape(-1, 3)
        dd = ((pk[:, None, :] - ptsd[None, :, :]) ** 2).sum(2).min(1)
        b3 = np.minimum(bound, dd * (1 + 1e-6))
        B3 = b3.max()
        candd = np.nonzero(L < B3)[0]
        tot["d"] += candd.size + 16
        tot["ratio"] += B3 / B
        Lpd = np.stack([boxL(pk[i], pk[i], blo[candd], bhi[candd]) for i in range(128)], 1)
        tot["d_pp"] += int((Lpd <= b3[None, :]).any(1).sum())
        pd_ = np.zeros(candd.size, bool)
        for g in range(8):
            sl = slice(g * 16, (g + 1) * 16)
            pd_ |= boxL(pk[sl].min(0), pk[sl].max(0), blo[candd], bhi[candd]) < b3[sl].max()
        tot["d_sg"] += int(pd_.sum()) + 16
    n = len(rows)
    print(f"pass {k:2d}: listed now {tot['now']/n:8.0f} per row (survive the per-point box {tot['now_pp']/n:7.0f}; with the final bounds {tot['ideal']/n:6.0f}) | "
          f"(a) sub-group boxes {tot['a']/n:8.0f} | (b) near-first {tot['b']/n:8.0f} (first sub-round {tot['s1']/n:5.0f}) | (c) both {tot['c']/n:8.0f} | "
          f"(d) descent first: listed {tot['d']/n:8.0f} (survive per-point {tot['d_pp']/n:6.0f}; with sub-group boxes {tot['d_sg']/n:7.0f}; B'/B {tot['ratio']/n:.3f})", flush=True)
