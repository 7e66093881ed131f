"""Which STAGE of the device's Titsias backward pass costs dELBO/dZ its accuracy at configs[4] (N = 100 000, M = 2048)?  On a GPU box: the
device's own Gram matrices go through a numpy / LAPACK restatement of titsias.hip's formulation (solves with L, explicit inverse of the inner
system with one refinement step for t1, the M x M solve before the M x N product); every intermediate the device keeps (mogp_titsias_fetch)
is compared with it, and the contraction with the kernel derivatives is done in numpy on BOTH sets of adjoints -- against the 80-bit truth
of tests/golden/titsias_dz_truth_cfg5.npz.  usage: python tools/titsias_stage_errors.py"""
import os, sys, time
import numpy as np
from scipy.linalg import solve_triangular
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from mogptk_amd import gpr, synth, _lib
from oracle.table_model import _jr_block
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
loss = float(m.loss())
zp = [p for p in m.parameters() if p._name.endswith("induction_points")][0]
gz_dev = -zp.grad[:, 1].copy()
truth = fx["gz_truth"]
sc, nt = np.max(np.abs(truth)), np.linalg.norm(truth)
rep = lambda nm, g: print("%-58s max-norm %.3e  2-norm %.3e" % (nm, np.max(np.abs(g - truth)) / sc, np.linalg.norm(g - truth) / nt), flush=True)
rep("device, end to end", gz_dev)
hd = m._handle
table = np.asarray(k._spectral_terms(1), dtype=np.float64)
Z = np.asarray(m.kernel._kernel_format(m.Z()), dtype=np.float64)
Xf = np.asarray(m.X, dtype=np.float64)
yv = np.asarray(m.y, dtype=np.float64).reshape(-1, 1)
assert np.all(np.diff(Z[:, 0]) >= 0) and np.all(np.diff(Xf[:, 0]) >= 0), "the comparison assumes inputs sorted by channel (device order = caller order)"
sigma = float(np.asarray(m.likelihood.scale()).reshape(-1)[0])
s2, jitter = sigma * sigma, 1e-8
dev = {nm: hd.titsias_fetch(i, M) for i, nm in ((0, "GA"), (2, "beta"), (3, "r"), (5, "L"), (6, "Qs"), (7, "Pq"), (8, "t1"))}
dev["L"] = np.tril(dev["L"])
gad = np.tril(dev["GA"]); dev["GA"] = gad + np.tril(gad, -1).T      # the buffer holds the lower triangle of the symmetrised adjoint, without -1/2 beta beta^T / s2^2
for nm in ("beta", "r", "t1"):
    dev[nm] = dev[nm].reshape(-1, 1)
Kuu = _lib.gram(0, C, 1, table, Z)
B = _lib.gram(0, C, 1, table, Z, Xf)
A = Kuu + jitter * np.mean(np.diagonal(Kuu)) * np.eye(M)
I = np.eye(M)
sol = lambda Lm, R, tr=False: solve_triangular(Lm, R, lower=True, trans=1 if tr else 0)
rel = lambda a, b: float(np.max(np.abs(a - b)) / np.max(np.abs(b)))


def contract(GAf, GBf):
    cz = Z[:, 0].astype(np.int64); cx = Xf[:, 0].astype(np.int64)
    g = np.zeros((M, 1))
    for i in range(C):
        ri = np.nonzero(cz == i)[0]
        for j in range(C):
            rj = np.nonzero(cx == j)[0]
            g[ri] += np.einsum("nm,nmd->nd", GBf[np.ix_(ri, rj)], _jr_block(table[i, j], Z[ri, 1:], Xf[rj, 1:]))
            zj = np.nonzero(cz == j)[0]
            g[ri] += 2.0 * np.einsum("nm,nmd->nd", GAf[np.ix_(ri, zj)], _jr_block(table[i, j], Z[ri, 1:], Z[zj, 1:]))
    return g[:, 0]


def pipeline(given, label):
    """titsias.hip's formulation in numpy / LAPACK; every stage named in `given` is taken from the DEVICE instead of being computed here"""
    g = lambda nm, f: dev[nm] if nm in given else f()
    L = g("L", lambda: np.linalg.cholesky(A))
    v = hd.titsias_fetch(4, M) if "v" in given else sol(L, B)
    Qs = g("Qs", lambda: (v @ v.T) / s2 + I)
    vy = v @ yv

    def pq():
        Lq = np.linalg.cholesky(Qs)
        Lqi = sol(Lq, I)
        return Lqi.T @ Lqi
    Pq = g("Pq", pq)

    def t1f():
        t = Pq @ vy
        return t + Pq @ (vy - Qs @ t)
    t1 = g("t1", t1f)
    beta = g("beta", lambda: sol(L, t1, True))
    r = g("r", lambda: yv / s2 ** 2 - (B.T @ beta) / s2 ** 3)
    GB = hd.titsias_fetch(1, M) if "GB" in given else sol(L, (I - Pq) / s2, True) @ v

    def gaf():
        T1 = sol(L, 2.0 * I - Pq - Qs, True)
        X_ = 0.5 * sol(L, T1.T, True).T
        return 0.5 * (X_ + X_.T)
    GA = g("GA", gaf) - 0.5 * (beta @ beta.T) / s2 ** 2
    rep(label, contract(GA, GB + beta @ r.T))
    return dict(L=L, v=v, Qs=Qs, Pq=Pq, t1=t1, beta=beta, r=r, GB=GB, GA=GA)


ref = pipeline((), "numpy everything (on the device's Gram matrices)")
stages = ["L", "v", "Qs", "Pq", "t1", "beta", "r", "GB", "GA"]
for n in range(1, len(stages) + 1):
    out = pipeline(tuple(stages[:n]), "device: " + ", ".join(stages[:n]) + "; numpy behind")
    if n < len(stages):
        nxt = stages[n]
        # the NEXT stage: the device's against numpy's computed from the device's inputs so far
        a_, b_ = (hd.titsias_fetch(4, M) if nxt == "v" else hd.titsias_fetch(1, M) if nxt == "GB" else dev[nxt]), out[nxt]
        if nxt == "GA":
            b_ = b_ + 0.5 * (out["beta"] @ out["beta"].T) / s2 ** 2
        print("    %-5s device vs numpy from the same inputs: %.2e of the largest entry" % (nxt, rel(a_, b_)), flush=True)
    del out
# single substitutions: everything by numpy from the device's L (so that all stages share one factor) except ONE stage taken from the device
if os.environ.get("STAGE_SINGLES", "0") != "0":
    for one in ("v", "Pq", "t1", "beta", "GB", "GA"):
        pipeline(("L", one), "device L and %s only" % one)
