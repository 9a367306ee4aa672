"""Independent numpy/LAPACK restatement of the reference's point-to-point iteration, used only to
cross-check the C oracle (a second opinion written against the same reference lines,
src/ICP_CPU.c:217-271, with numpy's LAPACK gesdd standing in for MKL's gesvd)."""
import numpy as np


def nn(P, Q):
    P = np.asarray(P)
    Q = np.asarray(Q, dtype=P.dtype)
    idx = np.empty(P.shape[0], dtype=np.int32)
    for i in range(P.shape[0]):
        d = Q - P[i]              # vdSub
        d = d * d                 # vdSqr
        s = d[:, 0] + d[:, 1]     # vdAdd
        s = s + d[:, 2]           # vdAdd
        idx[i] = int(np.argmin(s))  # first minimum
    return idx


def minimize(P, Q, idx):
    P = np.asarray(P, dtype=np.float64)
    Qi = np.asarray(Q, dtype=np.float64)[idx]
    pb, qb = P.mean(0), Qi.mean(0)
    N = (Qi - qb).T @ (P - pb)
    U, _, Vt = np.linalg.svd(N)
    R = U @ Vt
    return R, qb - R @ pb


def icp(D, M, max_iter, tol, fixed=False):
    P = np.array(D, dtype=np.float64)
    Q = np.asarray(M, dtype=np.float64)
    E = [0.0]
    T = np.eye(4)
    i = 0
    while True:
        idx = nn(P, Q)
        R, t = minimize(P, Q, idx)
        P = P @ R.T + t
        Tk = np.eye(4)
        Tk[:3, :3], Tk[:3, 3] = R, t
        T = Tk @ T
        E.append(float(np.sqrt(((Q[idx] - P) ** 2).sum() / P.shape[0])))
        if not fixed and (E[-1] < tol or abs(E[-1] - E[-2]) < tol):
            break
        i += 1
        if i > max_iter - 1:
            break
    return dict(iterations=i, err=np.array(E), T=T, idx=idx, moved=P)
