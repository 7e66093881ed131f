import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, 'tests')
import numpy as np
from mogptk_amd import gpr, synth
from helpers import load, fixture_params
fx = load("cfg5.npz")
C, Q, D, Rq, N, M = [int(v) for v in fx["meta"]]
X, y = synth.make_data(N, C)
h = synth.mosm_hypers(C, Q)
k = gpr.MultiOutputSpectralMixtureKernel(Q=Q, output_dims=C)
for name in ("weight", "mean", "variance", "delay", "phase"):
    getattr(k, name).assign(h[name])
s = float(fx["scale"])
m = gpr.Titsias(k, X, y, Z=[M // C] * C, variance=s ** 2)
m.likelihood.scale.assign(s)
fp = fixture_params(fx)
for p, f in zip(m.parameters(), fp): p.data = np.array(f["raw"])
loss = float(m.loss())
print("loss", loss, float(fx["loss"]))
for p, f in zip(m.parameters(), fp):
    err = np.max(np.abs(p.grad - f["grad"])) / np.max(np.abs(f["grad"]))
    print("%-45s err %.3e  max|ref| %.3e" % (p._name, err, np.max(np.abs(f["grad"]))))
gz, rz = m.Z.grad[:,1], fp[0]["grad"][:,1]
i = np.argsort(-np.abs(gz-rz))[:8]
print(np.c_[i, gz[i], rz[i]])
print("corr", np.corrcoef(gz, rz)[0,1], "norms", np.linalg.norm(gz), np.linalg.norm(rz))
# repeatability
l2 = float(m.loss()); gz2 = m.Z.grad[:,1]
print("repeat max diff", np.max(np.abs(gz2-gz)), l2-loss)
