"""Time and accuracy of the backward-stable form of the exact evaluation (mogp_model_set_accurate) next to the fast schedules, on an ill-conditioned
model (noise 1e-3 of unit-amplitude data).  usage: python tools/exact_accurate_time.py [N]"""
import os, sys, time, warnings
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from mogptk_amd import gpr, synth
from oracle.table_model import TableDevice
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
C, Q = 2, 2
X, y = synth.make_data(N, C)
h = synth.mosm_hypers(C, Q)
def model(sigma):
    k = gpr.MultiOutputSpectralMixtureKernel(Q=Q, output_dims=C)
    for name in ("weight", "mean", "variance", "delay", "phase"):
        getattr(k, name).assign(h[name])
    m = gpr.Exact(k, X, y, variance=sigma ** 2)
    m.likelihood.scale.assign(sigma)
    return m
warnings.simplefilter("ignore")
for sigma in (1e-3,):
    mt = model(sigma); mt._handle = TableDevice(0, mt.kernel._kernel_format(mt.X), mt.y, C)
    lt, gt = float(mt.loss()), [p.grad.copy() for p in mt.parameters()]
    for fallback in (False, True):
        gpr.config.accurate_fallback = fallback
        m = model(sigma)
        for _ in range(3): l = float(m.loss())
        t0 = time.perf_counter()
        for _ in range(5): l = float(m.loss())
        dt = (time.perf_counter() - t0) / 5
        eg = max(float(np.max(np.abs(p.grad - b)) / np.max(np.abs(b))) for p, b in zip(m.parameters(), gt))
        print("N %d sigma %.0e %-22s %.1f ms per LML + gradient; against the LAPACK twin: LML %.2e, worst gradient tensor %.2e" % (N, sigma, "backward-stable form:" if fallback else "fast schedule:", 1e3 * dt, abs(l - lt) / abs(lt), eg), flush=True)
