"""One-off robustness checks on a GPU box: 150 x create/registration/destroy (host and device memory stay put), degenerate shapes
(1 x 1 M, 1 M x 1, identical clouds, an infinite model point), two contexts alive at once."""
import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
pkg = load_package()
import psutil, torch
rng = np.random.default_rng(0)
proc = psutil.Process()
Q = rng.standard_normal((5000,3)).astype(np.float32); P = (Q[rng.integers(0,5000,3000)] + 0.01).astype(np.float32)
free0 = None
for k in range(150):
    with pkg.Context(0) as ctx:
        r = ctx.point_to_point(P, Q, max_iter=10, tol=1e-7)
    if k == 10:
        rss0 = proc.memory_info().rss; free0 = torch.cuda.mem_get_info()[0] if torch.cuda.is_available() else 0
rss1 = proc.memory_info().rss; free1 = torch.cuda.mem_get_info()[0] if torch.cuda.is_available() else 0
print("create/destroy x150: host rss growth %.1f MB, device free change %.1f MB" % ((rss1-rss0)/1e6, (free0-free1)/1e6))
with pkg.Context(0) as ctx:
    # degenerate shapes
    one = np.array([[0.1,0.2,0.3]], np.float32)
    big = rng.standard_normal((1_000_000,3)).astype(np.float32)
    idx = ctx.Matching(big, one); print("n=1M, m=1:", bool((idx==0).all()))
    idx = ctx.Matching(one, big); d=((one-big)**2); dd=(d[:,0]+d[:,1])+d[:,2]; print("n=1, m=1M:", int(idx[0])==int(dd.argmin()))
    # all points identical
    same = np.tile(one, (5000,1)); idx = ctx.Matching(same, same); print("identical clouds -> index 0:", bool((idx==0).all()))
    # NaN / inf in the model are never matched when a finite point exists? (reference has no rule; just must not hang)
    Qn = Q.copy(); Qn[7] = np.inf; idx = ctx.Matching(P, Qn); print("inf in model: no index 7:", bool((idx!=7).all()))
    # two contexts alive at once
    with pkg.Context(0) as c2:
        a = ctx.point_to_point(P, Q, max_iter=10, tol=1e-7); b = c2.point_to_point(P, Q, max_iter=10, tol=1e-7)
        print("two contexts, same result:", bool(np.array_equal(a.T, b.T)) and bool(np.array_equal(a.idx, b.idx)))
