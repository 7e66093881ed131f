"""How accurate are the Gram ENTRIES the device computes, next to numpy's fp64 ones?  K_uu (2048 x 2048 inducing grid of configs[4]) and a slab of
K_uf against the same entries evaluated in 80-bit numpy.longdouble from the same fp64 term table and inputs.
usage: python tools/gram_accuracy.py [columns of the K_uf slab]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from mogptk_amd import gpr, synth, _lib
from oracle.table_model import gram_from_table, table_block

LD = np.longdouble
C, Q, N, M = 4, 3, 100000, 2048
S = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
X, y = synth.make_data(N, C)
h = synth.mosm_hypers(C, Q)
k = gpr.MultiOutputSpectralMixtureKernel(Q=Q, output_dims=C)
for name in ("weight", "mean", "variance", "delay", "phase"):
    getattr(k, name).assign(h[name])
m = gpr.Titsias(k, X, y, Z=[M // C] * C, variance=0.25 ** 2)
table = np.asarray(k._spectral_terms(1), dtype=np.float64)
Z = np.asarray(m.kernel._kernel_format(m.Z()), dtype=np.float64)
Xf = np.asarray(m.X, dtype=np.float64)
rng = np.random.default_rng(0)
Xs = Xf[np.sort(rng.choice(N, S, replace=False))]


def gram_ld(X1, X2=None):
    X2_ = X1 if X2 is None else X2
    c1, c2 = X1[:, 0].astype(np.int64), X2_[:, 0].astype(np.int64)
    K = np.zeros((X1.shape[0], X2_.shape[0]), dtype=LD)
    for i in range(C):
        r1 = np.nonzero(c1 == i)[0]
        for j in range(C):
            r2 = np.nonzero(c2 == j)[0]
            if len(r1) == 0 or len(r2) == 0:
                continue
            if X2 is None and i < j:
                Ec, _, _ = table_block(table[j, i].astype(LD), X1[r2, 1:].astype(LD), X1[r1, 1:].astype(LD))
                K[np.ix_(r1, r2)] = np.einsum("t,tnm->nm", table[j, i][:, 0].astype(LD), Ec).T
            else:
                Ec, _, _ = table_block(table[i, j].astype(LD), X1[r1, 1:].astype(LD), X2_[r2, 1:].astype(LD))
                K[np.ix_(r1, r2)] = np.einsum("t,tnm->nm", table[i, j][:, 0].astype(LD), Ec)
    return K


for name, a, b in (("K_uu", Z, None), ("K_uf slab", Z, Xs)):
    T = gram_ld(a, b)
    sc = float(np.abs(T).max())
    for who, K in (("device", _lib.gram(0, C, 1, table, a, b)), ("numpy fp64", gram_from_table(table, a, b))):
        e = (K.astype(LD) - T).astype(np.float64)
        big = np.abs(T.astype(np.float64)) > 1e-3 * sc
        print("%-10s %-11s max |err| %.2e, rms %.2e of the largest entry; relative to the entry itself (|K| > 1e-3 max): max %.2e, rms %.2e; mean err %+.2e"
              % (name, who, np.abs(e).max() / sc, np.sqrt(np.mean(e * e)) / sc, np.max(np.abs(e[big] / T.astype(np.float64)[big])),
                 np.sqrt(np.mean((e[big] / T.astype(np.float64)[big]) ** 2)), np.mean(e) / sc))
