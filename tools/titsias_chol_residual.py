"""Backward error of the device's Cholesky factor of K_uu + jitter at configs[4] (M = 2048 grid inducing points, cond ~1e11) next to LAPACK's,
per 128 x 128 tile: |L L^T - A| / max|A|.  usage: python tools/titsias_chol_residual.py   (GPU box)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from mogptk_amd import gpr, synth, _lib
from helpers import load, fixture_params
fx = load("titsias_dz_truth_cfg5.npz")
C, Q, D, Rq, N, M = [int(v) for v in fx["meta"]]
X, y = synth.make_data(N, C)
h = synth.mosm_hypers(C, Q)
k = gpr.MultiOutputSpectralMixtureKernel(Q=Q, output_dims=C)
for name in ("weight", "mean", "variance", "delay", "phase"):
    getattr(k, name).assign(h[name])
s = float(fx["scale"])
m = gpr.Titsias(k, X, y, Z=[M // C] * C, variance=s ** 2)
m.likelihood.scale.assign(s)
for p, f in zip(m.parameters(), fixture_params(fx)):
    p.data = np.array(f["raw"])
m.loss()
table = np.asarray(k._spectral_terms(1), dtype=np.float64)
Z = np.asarray(m.kernel._kernel_format(m.Z()), dtype=np.float64)
Kuu = _lib.gram(0, C, 1, table, Z)
A = Kuu + 1e-8 * np.mean(np.diagonal(Kuu)) * np.eye(M)
Ld = np.tril(m._handle.titsias_fetch(5, M))
Ll = np.linalg.cholesky(A)
LD = np.longdouble
for nm, L in (("device", Ld), ("LAPACK", Ll)):
    E = np.tril((L.astype(LD) @ L.astype(LD).T - A.astype(LD)).astype(np.float64))
    t = np.abs(E).reshape(16, 128, 16, 128).max(axis=(1, 3)) / np.abs(A).max()
    print("%s: max |L L^T - A| / max|A| = %.2e; diagonal tiles %.2e, below the diagonal %.2e; Frobenius %.2e" % (nm, t.max(), np.diagonal(t).max(), np.tril(t, -1).max(), np.linalg.norm(E) / np.linalg.norm(np.tril(A))))
    if nm == "device":
        np.set_printoptions(linewidth=250, precision=1)
        print((t / 1e-16).round(0).astype(int))
ev = np.linalg.eigvalsh(A)
print("eigenvalues of A: min %.3e max %.3e (cond %.2e); jitter %.3e" % (ev[0], ev[-1], ev[-1] / ev[0], 1e-8 * np.mean(np.diagonal(Kuu))))
