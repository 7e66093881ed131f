"""Numerics of dELBO/dZ at the conditioning of BASELINE configs[4] (M = 2048 grid inducing points, K_uu cond ~1e11), N reduced so
that an 80-bit reference is affordable.  Compares, against extended-precision linear algebra on the SAME fp64 Gram matrices:
  E   explicit inverse factors W = L^-1 applied as matrix products (round-1 device formulation)
  S   triangular solves everywhere (reference formulation)
  H   triangular solves with L_uu (cond ~1e11), explicit inverse of the well-conditioned L_q = chol(Q/s2 + I): what titsias.hip does
each with a LAPACK Cholesky and with the device's blocked Cholesky (panels formed with explicit 128 x 128 tile inverses).
usage: python tools/titsias_numerics.py [N]      (build container only; ~5 min)"""
import sys, time
import numpy as np
from scipy.linalg import solve_triangular
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from mogptk_amd import gpr, synth
from oracle.table_model import gram_from_table, _jr_block

LD = np.longdouble
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
C, Q, M = 4, 3, 2048
X, y = synth.make_data(N, C)
h = synth.mosm_hypers(C, Q)
k = gpr.MultiOutputSpectralMixtureKernel(Q=Q, output_dims=C)
for name in ("weight", "mean", "variance", "delay", "phase"):
    getattr(k, name).assign(h[name])
table = k._spectral_terms(1)
from mogptk_amd.gpr.model import init_inducing_points
Z = init_inducing_points([M // C] * C, X, "grid", C)
sigma, jitter = float(__import__('os').environ.get('TITSIAS_SIGMA', '0.25')), 1e-8
s2 = sigma * sigma
y = y.reshape(-1, 1)
Kuu = gram_from_table(table, Z)
A = Kuu + jitter * np.mean(np.diagonal(Kuu)) * np.eye(M)
B = gram_from_table(table, Z, X)
print("N", N, "M", M, "cond(Kuu+jit) ~ %.2e" % np.linalg.cond(A))


def chol_device(A, nb=128):
    """right-looking blocked Cholesky, panels = A_ik inv(L_kk)^T with the explicit tile inverse (mogp_api.hip:spd_potrf)"""
    A = A.copy(); n = A.shape[0]
    for k0 in range(0, n, nb):
        k1 = min(k0 + nb, n)
        Lkk = np.linalg.cholesky(A[k0:k1, k0:k1])
        Wkk = solve_triangular(Lkk, np.eye(k1 - k0), lower=True)
        A[k0:k1, k0:k1] = Lkk
        if k1 < n:
            A[k1:, k0:k1] = A[k1:, k0:k1] @ Wkk.T
            A[k1:, k1:] -= A[k1:, k0:k1] @ A[k1:, k0:k1].T
    return np.tril(A)


def chol_ld(A, nb=256):
    A = A.astype(LD).copy(); n = A.shape[0]
    for k0 in range(0, n, nb):
        k1 = min(k0 + nb, n)
        blk = A[k0:k1, k0:k1]
        for j in range(k1 - k0):                       # unblocked on the diagonal block
            blk[j, j] = np.sqrt(blk[j, j] - blk[j, :j] @ blk[j, :j])
            if j + 1 < k1 - k0:
                blk[j + 1:, j] = (blk[j + 1:, j] - blk[j + 1:, :j] @ blk[j, :j]) / blk[j, j]
        if k1 < n:
            P = A[k1:, k0:k1]
            for j in range(k1 - k0):                   # panel by forward substitution
                P[:, j] = (P[:, j] - P[:, :j] @ blk[j, :j]) / blk[j, j]
            A[k1:, k1:] -= P @ P.T
    return np.tril(A)


def trsm_ld(L, Bm, trans=False, nb=256):
    """X = L^-1 B (or L^-T B) in extended precision, blocked substitution with unblocked diagonal blocks"""
    L = L.astype(LD); Xm = Bm.astype(LD).copy(); n = L.shape[0]
    blocks = list(range(0, n, nb))
    if not trans:
        for k0 in blocks:
            k1 = min(k0 + nb, n)
            if k0 > 0: Xm[k0:k1] -= L[k0:k1, :k0] @ Xm[:k0]
            for j in range(k0, k1):
                Xm[j] = (Xm[j] - L[j, k0:j] @ Xm[k0:j]) / L[j, j]
    else:
        for k0 in reversed(blocks):
            k1 = min(k0 + nb, n)
            if k1 < n: Xm[k0:k1] -= L[k1:, k0:k1].T @ Xm[k1:]
            for j in reversed(range(k0, k1)):
                Xm[j] = (Xm[j] - L[j + 1:k1, j] @ Xm[j + 1:k1]) / L[j, j]
    return Xm


def grads(L, mode, dt=np.float64):
    """GA, GB (adjoints of Kuu_jittered and Kuf) from a Cholesky factor L; mode 'E' explicit inverses, 'S' solves, 'T' extended"""
    I = np.eye(M, dtype=dt)
    if mode == "T":
        sol = lambda Lm, R, tr=False: trsm_ld(Lm, R, tr)
        chol = chol_ld
    else:
        sol = lambda Lm, R, tr=False: solve_triangular(Lm, R, lower=True, trans=1 if tr else 0)
        chol = np.linalg.cholesky
    Bd, yd = B.astype(dt), y.astype(dt)
    if mode == "E":
        W = sol(L, I)
        v = W @ Bd
    else:
        v = sol(L, Bd)
    Qm = v @ v.T
    Qs = Qm / s2 + I
    Lq = chol(Qs)
    vy = v @ yd
    if mode == "E":
        Lqi = sol(Lq, I); Pq = Lqi.T @ Lqi
        t1 = Pq @ vy
        beta = W.T @ t1
        Rm = Pq @ Qm / s2
        r = yd / s2 ** 2 - (Bd.T @ beta) / s2 ** 3
        GB = W.T @ (Rm @ v) / s2 + beta @ r.T
        GA = 0.5 * W.T @ (Rm - Qm / s2) @ W - 0.5 * (beta @ beta.T) / s2 ** 2
    elif mode.startswith("H"):
        Lqi = sol(Lq, I); Pqe = Lqi.T @ Lqi                   # explicit inverse of the inner system
        Pqs = sol(Lq, sol(Lq, I), True)                       # the same by solves
        which = mode[1:] or "tge"                             # which uses take the explicit one: t(1), g(B), e (E)
        if "n" in which:                                      # the explicit inverse after one Newton-Schulz step Pq <- Pq + Pq (I - Qs Pq) (two M^3 products)
            Pqe = Pqe + Pqe @ (I - Qs @ Pqe)
            Pqe = 0.5 * (Pqe + Pqe.T)
        if "r" in which:                                      # explicit inverse + one step of iterative refinement
            t1 = Pqe @ vy
            t1 = t1 + Pqe @ (vy - Qs @ t1)
        else:
            t1 = Pqe @ vy if "t" in which else sol(Lq, sol(Lq, vy), True)
        beta = sol(L, t1, True)
        if "x" in which:                                      # beta = L^-T t1 by an 80-bit substitution, rounded once (is it this vector's forward error?)
            beta = trsm_ld(L, t1, True).astype(dt)
        if "y" in which:                                      # ... and t1 itself from an 80-bit solve of the inner system
            Lq_ld = chol_ld(Qs)
            t1 = trsm_ld(Lq_ld, trsm_ld(Lq_ld, vy), True).astype(dt)
            beta = trsm_ld(L, t1, True).astype(dt)
        r = yd / s2 ** 2 - (Bd.T @ beta) / s2 ** 3
        if "b" in which:                                      # round 4's order: the M x M solve first, then ONE M x M x N product
            GB = sol(L, (I - Pqe) / s2, True) @ v + beta @ r.T
        else:
            Pv = Pqe @ v if "g" in which else sol(Lq, sol(Lq, v), True)
            GB = sol(L, (v - Pv) / s2, True) + beta @ r.T
        Pq = Pqe if "e" in which else Pqs
        Em = 2.0 * I - Pq - Qs
        T1 = sol(L, Em, True)
        GA = 0.5 * sol(L, T1.T, True).T - 0.5 * (beta @ beta.T) / s2 ** 2
    else:
        t1 = sol(Lq, sol(Lq, vy), True)                       # Pq vy
        beta = sol(L, t1, True)
        r = yd / s2 ** 2 - (Bd.T @ beta) / s2 ** 3
        # (I - Pq) v / s2 :  Pq v by two solves
        Pv = sol(Lq, sol(Lq, v), True)
        GB = sol(L, (v - Pv) / s2, True) + beta @ r.T
        # E = 2I - Pq - Qs ;  GA = 1/2 L^-T E L^-1 - 1/2 beta beta^T / s2^2
        Pq = sol(Lq, sol(Lq, I), True)
        Em = 2.0 * I - Pq - Qs
        T1 = sol(L, Em, True)                                 # L^-T E
        GA = 0.5 * sol(L, T1.T, True).T - 0.5 * (beta @ beta.T) / s2 ** 2
    GA = 0.5 * (GA + GA.T)
    return GA, GB


def gz_from(GA, GB):
    dt = GA.dtype
    cz = Z[:, 0].astype(np.int64); cx = X[:, 0].astype(np.int64)
    gZ = np.zeros((M, 1), dtype=dt)
    for i in range(C):
        ri = np.nonzero(cz == i)[0]
        for j in range(C):
            rj = np.nonzero(cx == j)[0]
            gZ[ri] += np.einsum("nm,nmd->nd", GB[np.ix_(ri, rj)], _jr_block(table[i, j], Z[ri, 1:], X[rj, 1:]).astype(dt))
            zj = np.nonzero(cz == j)[0]
            gZ[ri] += 2.0 * np.einsum("nm,nmd->nd", GA[np.ix_(ri, zj)], _jr_block(table[i, j], Z[ri, 1:], Z[zj, 1:]).astype(dt))
    return gZ


import os
t0 = time.time()
cache = "/tmp/titsias_truth_%d_%g.npz" % (N, sigma)
if os.path.exists(cache):
    f = np.load(cache); gZt, GAt, GBt = f["gZ"], f["GA"], f["GB"]
else:
    Lt = chol_ld(A)
    GAt, GBt = grads(Lt, "T", LD)
    gZt = gz_from(GAt, GBt).astype(np.float64)
    GAt, GBt = GAt.astype(np.float64), GBt.astype(np.float64)
    np.savez(cache, gZ=gZt, GA=GAt, GB=GBt)
print("extended-precision reference: %.0f s; |gZ|max %.3e, |GA|max %.3e, |GB|max %.3e" % (time.time() - t0, np.abs(gZt).max(), float(np.abs(GAt).max()), float(np.abs(GBt).max())))
Ll = np.linalg.cholesky(A)
Ld = chol_device(A)
print("Cholesky residuals |LL^T - A|/|A|: lapack %.2e, device-style %.2e" % (np.abs(Ll @ Ll.T - A).max() / np.abs(A).max(), np.abs(Ld @ Ld.T - A).max() / np.abs(A).max()))
for lname, L in (("lapack chol", Ll), ("device chol", Ld)):
    for mode in (("S", "H", "Hge", "Hrge", "Hrgeb") if lname == "device chol" else ("S",)):
        GA, GB = grads(L, mode)
        gZ = gz_from(GA, GB)
        e = np.abs(gZ - gZt).max() / np.abs(gZt).max()
        cos = float(gZ[:, 0] @ gZt[:, 0] / np.linalg.norm(gZ) / np.linalg.norm(gZt))
        print("%-12s %s: gZ rel err %.3e  cos %.6f   GA rel err %.2e  GB rel err %.2e" % (lname, {"E": "explicit W", "S": "solves   ", "H": "solves L, explicit Lq", "Ht": "explicit Pq in t1 only", "Hg": "explicit Pq in GB only", "He": "explicit Pq in E only", "Hge": "explicit Pq in GB and E, t1 by solves", "Hrge": "explicit Pq everywhere, t1 refined once", "Hrgeb": "the same, M x M solve before the M x N product (titsias.hip)"}[mode], e, cos,
              float(np.abs(GA - GAt).max() / np.abs(GAt).max()), float(np.abs(GB - GBt).max() / np.abs(GBt).max())))
