"""config 5 timing: Titsias ELBO+gradient evaluations, MOSM C=4 Q=3 N=100000 M=2048 (reference CPU: 53.8-58.6 s/eval)."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
from mogptk_amd import gpr, synth
C, Q, N, M = 4, 3, 100000, 2048
X, y = synth.make_data(N, C)
h = synth.mosm_hypers(C, Q)
k = gpr.MultiOutputSpectralMixtureKernel(Q=Q, output_dims=C)
for name in ("weight", "mean", "variance", "delay", "phase"):
    getattr(k, name).assign(h[name])
s = float(np.mean(h["scale"]))
m = gpr.Titsias(k, X, y, Z=[M // C] * C, variance=s ** 2)
m.likelihood.scale.assign(s)
for _ in range(2): m.loss()
t = time.perf_counter()
for _ in range(5): m.loss()
dt = (time.perf_counter() - t) / 5
print("titsias cfg5: %.1f ms / ELBO+grad eval  (%.2f evals/s)" % (dt * 1e3, 1 / dt))
