"""numpy model of the exact path's two-level Cholesky (128-tile inverses inside a 512 block, W_KK by block recurrence, outer panel P = A W_KK^T) and of what would
repair its backward error when K_j is ill-conditioned: accurate / Newton-refined tile inverses, a Newton step on W_KK, refined outer / inner panels.
Residual |L L^T - A| / |A| (80-bit), log-determinant and y^T K_j^-1 y against LAPACK.  usage: python tools/potrf_two_level_model.py   (CPU, ~15 min)"""
import sys
import numpy as np
from scipy.linalg import solve_triangular
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from mogptk_amd import gpr, synth
from oracle.table_model import gram_from_table
N, C, Q = 2048, 2, 2
X, y = synth.make_data(N, C)
h = synth.mosm_hypers(C, Q)
k = gpr.MultiOutputSpectralMixtureKernel(Q=Q, output_dims=C)
for name in ("weight", "mean", "variance", "delay", "phase"):
    getattr(k, name).assign(h[name])
m = gpr.Exact(k, X, y, variance=1.0)
Xf = np.asarray(m.kernel._kernel_format(m.X), dtype=np.float64)
K = gram_from_table(np.asarray(k._spectral_terms(1)), Xf)
LD = np.longdouble

def tile_inv(L, mode):
    n = L.shape[0]
    if mode == "solve":
        return solve_triangular(L, np.eye(n), lower=True)
    b = 16; W = np.zeros_like(L)
    for i in range(0, n, b):
        W[i:i+b, i:i+b] = np.linalg.inv(L[i:i+b, i:i+b])
    for i in range(0, n, b):
        for j in range(i - b, -1, -b):
            W[i:i+b, j:j+b] = -W[i:i+b, i:i+b] @ (L[i:i+b, j:i] @ W[j:i, j:j+b])
    if mode == "newton":
        W = W + W @ (np.eye(n) - L @ W)
    return W

def potrf(A, tmode, wkk_newton=False, refine_outer=False, refine_inner=False):
    A = A.copy(); n = A.shape[0]; OB, T = 512, 128
    for k0 in range(0, n, OB):
        k1 = min(k0 + OB, n)
        Ws = {}
        for t0 in range(k0, k1, T):
            t1 = t0 + T
            Ltt = np.linalg.cholesky(A[t0:t1, t0:t1]); A[t0:t1, t0:t1] = Ltt
            W = tile_inv(Ltt, tmode); Ws[t0] = W
            if t1 < k1:                                     # panel inside the block + update inside the block
                A0 = A[t1:k1, t0:t1].copy()
                P = A0 @ W.T
                if refine_inner: P = P + (A0 - P @ Ltt.T) @ W.T
                A[t1:k1, t0:t1] = P
                A[t1:k1, t1:k1] -= P @ A[t1:k1, t0:t1].T
        if k1 < n:
            LKK = np.tril(A[k0:k1, k0:k1])
            # W_KK by block recurrence from the tile inverses
            nb = (k1 - k0) // T; WKK = np.zeros_like(LKK)
            for i in range(nb):
                WKK[i*T:(i+1)*T, i*T:(i+1)*T] = Ws[k0 + i*T]
            for i in range(nb):
                for j in range(i - 1, -1, -1):
                    WKK[i*T:(i+1)*T, j*T:(j+1)*T] = -WKK[i*T:(i+1)*T, i*T:(i+1)*T] @ (LKK[i*T:(i+1)*T, j*T:i*T] @ WKK[j*T:i*T, j*T:(j+1)*T])
            if wkk_newton: WKK = WKK + WKK @ (np.eye(k1 - k0) - LKK @ WKK)
            A0 = A[k1:, k0:k1].copy()
            P = A0 @ WKK.T
            if refine_outer: P = P + (A0 - P @ LKK.T) @ WKK.T
            A[k1:, k0:k1] = P
            A[k1:, k1:] -= P @ P.T
    return np.tril(A)

for sigma in (1e-2, 1e-3):
    A = K + (sigma ** 2 + 1e-8 * np.mean(np.diagonal(K))) * np.eye(N)
    Ll = np.linalg.cholesky(A)
    ref_ld = 2 * np.sum(np.log(np.diagonal(Ll))); ref_q = float(np.sum(solve_triangular(Ll, y.reshape(-1, 1)[np.argsort(X[:,0], kind='stable')] if False else np.asarray(m.y).reshape(-1,1), lower=True) ** 2))
    def report(nm, L):
        E = (L.astype(LD) @ L.astype(LD).T - A.astype(LD)).astype(np.float64)
        ld = 2 * np.sum(np.log(np.diagonal(L))); q = float(np.sum(solve_triangular(L, np.asarray(m.y).reshape(-1,1), lower=True) ** 2))
        print("  %-46s residual %.2e   logdet rel %.2e  quad rel %.2e" % (nm, np.abs(E).max() / np.abs(A).max(), abs(ld - ref_ld) / abs(ref_ld), abs(q - ref_q) / abs(ref_q)))
    print("sigma", sigma, "cond %.1e" % np.linalg.cond(A))
    report("LAPACK", Ll)
    report("device-style (blocked16 tiles, W_KK recurrence)", potrf(A, "blocked16"))
    report("accurate tile inverses (solve)", potrf(A, "solve"))
    report("Newton-refined tile inverses", potrf(A, "newton"))
    report("Newton tiles + Newton W_KK", potrf(A, "newton", wkk_newton=True))
    report("blocked16 + outer panel refined", potrf(A, "blocked16", refine_outer=True))
    report("blocked16 + inner and outer panels refined", potrf(A, "blocked16", refine_outer=True, refine_inner=True))
    report("Newton tiles + outer refined", potrf(A, "newton", refine_outer=True))
