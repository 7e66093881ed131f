"""How does the exact path (panels formed with explicit tile / block inverses: chain.hip, flow.hip) behave when K + sigma^2 I is ill-conditioned?
LML and gradient of the device against the numpy twin (LAPACK Cholesky, same term table) for decreasing noise.  usage: python tools/exact_illcond.py [N]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from mogptk_amd import gpr, synth, _lib
from oracle.table_model import TableDevice, gram_from_table

N = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
C, Q = 2, 2
X, y = synth.make_data(N, C)
h = synth.mosm_hypers(C, Q)
for sigma in (0.2, 1e-2, 1e-3, 1e-4):
    res = {}
    for who in ("device", "twin"):
        k = gpr.MultiOutputSpectralMixtureKernel(Q=Q, output_dims=C)
        for name in ("weight", "mean", "variance", "delay", "phase"):
            getattr(k, name).assign(h[name])
        m = gpr.Exact(k, X, y, variance=sigma ** 2)
        m.likelihood.scale.assign(sigma)
        if who == "twin":
            m._handle = TableDevice(0, m.kernel._kernel_format(m.X), m.y, C)
        loss = float(m.loss())
        res[who] = (loss, [p.grad.copy() for p in m.parameters()])
        if who == "twin":
            K = gram_from_table(np.asarray(k._spectral_terms(1)), np.asarray(m.kernel._kernel_format(m.X), dtype=np.float64))
            ev = np.linalg.eigvalsh(K + (sigma ** 2 + 1e-8 * np.mean(np.diagonal(K))) * np.eye(N))
    ld, lt = res["device"][0], res["twin"][0]
    ge = max(float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300)) for a, b in zip(res["device"][1], res["twin"][1]))
    print("sigma %.0e: cond(Kj) %.1e   loss device %.10e twin %.10e  rel %.2e   worst gradient tensor %.2e" % (sigma, ev[-1] / ev[0], ld, lt, abs(ld - lt) / abs(lt), ge), flush=True)
