"""dELBO/dZ AT configs[4] (N = 100 000, M = 2048) by fp64 numpy / LAPACK pipelines, against the 80-bit truth of tests/golden/titsias_dz_truth_cfg5.npz:
which ingredient of the device's formulation costs accuracy at this size?  Same function as the fixture (the reference's raw parameter values ->
term table, grid inducing inputs, noise scale), Gram matrices by numpy from the term table.  Variants (tools/titsias_numerics.py:grads):
  S    triangular solves everywhere, LAPACK Cholesky (what the reference does)
  Sd   the same on the device-style blocked Cholesky (128 x 128 tile inverses)
  H    solves with L_uu, explicit inverse of the inner system (what titsias.hip does)
usage: python tools/titsias_n100k_numpy.py [variants]      (build container only; ~10 GB, a few minutes per variant)"""
import os, sys, time
import numpy as np
from scipy.linalg import solve_triangular
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from mogptk_amd import gpr, synth
from oracle.table_model import gram_from_table, _jr_block
from helpers import load, fixture_params

LD = np.longdouble
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
table = np.asarray(k._spectral_terms(1), dtype=np.float64)
Z = np.asarray(m.kernel._kernel_format(m.Z()), dtype=np.float64)
assert np.array_equal(Z, fx["Z"])
sigma = float(np.asarray(m.likelihood.scale()).reshape(-1)[0])
jitter = 1e-8
s2 = sigma * sigma
X = np.asarray(m.kernel._kernel_format(m.X) if X.shape[1] == 1 else m.X, dtype=np.float64) if False else np.asarray(m.X, dtype=np.float64)
y = np.asarray(m.y, dtype=np.float64).reshape(-1, 1)
t0 = time.time()
if os.environ.get("GRAM_FROM_DEVICE", "0") != "0":       # on a GPU box: the DEVICE's Gram matrices under numpy's linear algebra -- is it the entries?
    from mogptk_amd import _lib
    Kuu = _lib.gram(0, C, 1, table, Z)
    B = _lib.gram(0, C, 1, table, Z, X)
    print("Gram matrices from the device %.0f s" % (time.time() - t0), flush=True)
else:
    Kuu = gram_from_table(table, Z)
    B = gram_from_table(table, Z, X)
    print("Gram matrices %.0f s" % (time.time() - t0), flush=True)
A = Kuu + jitter * np.mean(np.diagonal(Kuu)) * np.eye(M)
noise = float(os.environ.get("GRAM_NOISE", "0"))          # relative rounding noise added to every Gram entry (how sensitive is the result to the Gram's accuracy?)
if noise > 0.0:
    rng = np.random.default_rng(5)
    which = os.environ.get("GRAM_NOISE_IN", "uu,uf").split(",")
    absolute = os.environ.get("GRAM_NOISE_ABS", "0") != "0"          # noise relative to the LARGEST entry (what a factorised evaluation leaves) instead of to each entry
    if "uf" in which:
        if absolute:
            B += noise * np.abs(B).max() * rng.standard_normal(B.shape)
        else:
            B *= 1.0 + noise * rng.standard_normal(B.shape)
    if "uu" in which:
        E_ = np.tril(noise * rng.standard_normal(Kuu.shape))
        E_ = E_ + np.tril(E_, -1).T
        A = (Kuu + np.abs(Kuu).max() * E_ if absolute else Kuu * (1.0 + E_)) + jitter * np.mean(np.diagonal(Kuu)) * np.eye(M)
    print("Gram noise %.1e in %s" % (noise, which), flush=True)
src = open(os.path.join(ROOT, "tools", "titsias_numerics.py")).read()
exec(src[src.index("def chol_device"):src.index("import os\nt0 = time.time()")])
truth = fx["gz_truth"]
sc, nt = np.max(np.abs(truth)), np.linalg.norm(truth)
for nm, g in (("reference, 8 threads", fx["gz_ref"]), ("reference, 3 threads", fx["gz_ref_alt"])):
    print("%-28s max-norm %.3e  2-norm %.3e" % (nm, np.max(np.abs(g - truth)) / sc, np.linalg.norm(g - truth) / nt))
want = sys.argv[1].split(",") if len(sys.argv) > 1 else ["S", "Sd", "H"]
Ll = np.linalg.cholesky(A)
Ldv = chol_device(A) if any(v.endswith("d") for v in want) else None
for v in want:
    t0 = time.time()
    L = Ldv if v.endswith("d") else Ll
    GA, GB = grads(L, v.rstrip("d"))
    g = gz_from(GA, GB)[:, 0]
    print("%-28s max-norm %.3e  2-norm %.3e   (%.0f s)" % ("numpy " + v, np.max(np.abs(g - truth)) / sc, np.linalg.norm(g - truth) / nt, time.time() - t0), flush=True)
