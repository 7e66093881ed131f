"""Where the ~2e-3 of dELBO/dZ between any fp64 evaluation and the 80-bit truth comes from at the conditioning of BASELINE configs[4]
(M = 2048 grid inducing points): the rounding of the Gram ENTRIES or the fp64 linear algebra behind them?  Evaluates, against an 80-bit
pipeline on 80-bit Gram matrices (the function itself), fp64 pipelines (triangular solves everywhere) on
  a  fp64 Gram matrices as numpy computes them from the term table (exp / cos of fp64 arguments)
  b  the 80-bit Gram matrices rounded ONCE to fp64 (the best any fp64 Gram can be)
  c  K_uu rounded once, K_uf as numpy computes it        d  K_uf rounded once, K_uu as numpy computes it
and the 80-bit pipeline on the fp64 Gram matrices of (a) (the linear algebra exact, only the entries rounded).
usage: python tools/titsias_gram_rounding.py [N]      (build container only; ~10 min at N = 4000)"""
import sys, time
import numpy as np
sys.argv = sys.argv[:2]
exec(open(__import__("os").path.join(__import__("os").path.dirname(__import__("os").path.abspath(__file__)), "titsias_numerics.py")).read().split("import os\nt0 = time.time()")[0])

def gram_ld(table, X1, X2=None):
    """gram_from_table in numpy.longdouble from the same fp64 table and inputs"""
    from oracle.table_model import table_block
    C = table.shape[0]
    X2_ = X1 if X2 is None else X2
    c1, c2 = X1[:, 0].astype(np.int64), X2_[:, 0].astype(np.int64)
    K = np.zeros((X1.shape[0], X2_.shape[0]), dtype=LD)
    for i in range(C):
        r1 = np.nonzero(c1 == i)[0]
        for j in range(C):
            r2 = np.nonzero(c2 == j)[0]
            if X2 is None and i < j:
                Ec, _, _ = table_block(table[j, i].astype(LD), X1[r2, 1:].astype(LD), X1[r1, 1:].astype(LD))
                K[np.ix_(r1, r2)] = np.einsum("t,tnm->nm", table[j, i][:, 0].astype(LD), Ec).T
            else:
                Ec, _, _ = table_block(table[i, j].astype(LD), X1[r1, 1:].astype(LD), X2_[r2, 1:].astype(LD))
                K[np.ix_(r1, r2)] = np.einsum("t,tnm->nm", table[i, j][:, 0].astype(LD), Ec)
    return K

t0 = time.time()
Kuu_t, B_t = gram_ld(table, Z), gram_ld(table, Z, X)
jit_t = LD(jitter) * np.mean(np.diagonal(Kuu_t))
print("80-bit Gram matrices %.0f s; fp64 entries differ from them by at most %.2e (K_uu), %.2e (K_uf) of the largest entry"
      % (time.time() - t0, float(np.abs(Kuu - Kuu_t).max() / np.abs(Kuu_t).max()), float(np.abs(B - B_t).max() / np.abs(B_t).max())), flush=True)

def run(Kuu_in, B_in, mode, dt):
    global A, B
    A_keep, B_keep = A, B
    A = (Kuu_in.astype(dt) + (jit_t if dt is LD else float(jit_t)) * np.eye(M, dtype=dt))
    B = B_in.astype(dt)
    L = chol_ld(A) if dt is LD else np.linalg.cholesky(A)
    GA, GB = grads(L, mode, dt)
    g = gz_from(GA, GB).astype(np.float64)
    A, B = A_keep, B_keep
    return g

t0 = time.time()
truth = run(Kuu_t, B_t, "T", LD)
print("truth (80-bit pipeline, 80-bit Gram) %.0f s, |gZ|max %.3e" % (time.time() - t0, np.abs(truth).max()), flush=True)
sc = np.abs(truth).max()
Kuu_r, B_r = Kuu_t.astype(np.float64), B_t.astype(np.float64)
for name, ku, kb in (("a  numpy fp64 Gram", Kuu, B), ("b  Gram rounded once", Kuu_r, B_r), ("c  K_uu rounded once", Kuu_r, B), ("d  K_uf rounded once", Kuu, B_r)):
    g = run(ku, kb, "S", np.float64)
    print("%-24s fp64 solves: %.3e of the tensor from the truth" % (name, np.abs(g - truth).max() / sc), flush=True)
t0 = time.time()
g = run(Kuu, B, "T", LD)
print("numpy fp64 Gram, 80-bit pipeline: %.3e (%.0f s)" % (np.abs(g - truth).max() / sc, time.time() - t0))
