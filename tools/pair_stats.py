"""Offline (CPU, numpy + oracle): how much of the brute-force work survives the sparse kernel's tests on the hall pair --
per block of 128 moving points: chunks that pass the group-box test (hits), hits whose box passes for at least one point (those are
evaluated in full: 128 points x 8 model points), and (point, hit) pairs that pass the per-point box test.
usage: python tools/pair_stats.py [steady|initial]"""
import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib
orc = oracle_lib.Oracle(); orc.set_threads(8)
g = os.path.join(ROOT, "tests", "golden")
P, Q = orc.hall_clouds(g)
which = sys.argv[1] if len(sys.argv) > 1 else "steady"
if which == "steady":
    r = orc.icp_p2p_f32x(P, Q, 6, 1e-12, fixed=True); P = r["moved"]
n, m = len(P), len(Q)
_, first = np.unique(Q, axis=0, return_index=True)
keep = np.zeros(m, bool); keep[first] = True
Qs = Q.copy(); Qs[~keep] = np.inf
idx = orc.nn(P, Q)
d = ((P - Q[idx])**2); d = (d[:,0]+d[:,1])+d[:,2]
Qc = Qs.reshape(-1, 8, 3)
fin = np.isfinite(Qc[:,:,0])
clo = np.where(fin[:,:,None], Qc, np.inf).min(1); chi = np.where(fin[:,:,None], Qc, -np.inf).max(1)
H=[]; PP=[]; FULL=[]
for b in range(n//128):
    pts = P[b*128:(b+1)*128]; bd = np.nextafter(d[b*128:(b+1)*128], np.float32(np.inf)); B = bd.max()
    glo=pts.min(0); ghi=pts.max(0)
    gg = np.maximum(np.maximum(clo-ghi, glo-chi),0); L=(gg**2).sum(1)
    hit = np.nonzero(L < B)[0]
    g2 = np.maximum(np.maximum(clo[hit][None]-pts[:,None], pts[:,None]-chi[hit][None]),0); Lp=(g2**2).sum(2)   # [128, H]
    passing = Lp <= bd[:,None]
    H.append(len(hit)); PP.append(passing.sum()); FULL.append((passing.any(0)).sum())
H=np.array(H); PP=np.array(PP); FULL=np.array(FULL)
print(which, "blocks", len(H), "hits/block median %.0f mean %.1f max %d" % (np.median(H), H.mean(), H.max()))
print("  hits whose box passes for >= 1 point: mean %.1f (%.0f%% of hits)" % (FULL.mean(), 100*FULL.sum()/H.sum()))
print("  (point, hit) pairs passing the per-point box test: mean %.1f per block = %.1f%% of 128 x (hits that pass for any point)" % (PP.mean(), 100*PP.sum()/(128*FULL.sum())))
