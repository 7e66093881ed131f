"""configs[4] on the device against the reference's golden vector: relative error of every gradient tensor, and whether a second evaluation
repeats the first bit for bit.  usage: python tools/cfg5_err.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mogptk_amd import gpr, synth
from tests.helpers import load, fixture_params
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
for p, f in zip(m.parameters(), fp):
    p.data = np.array(f["raw"])
loss = float(m.loss())
print("loss rel err %.2e" % (abs(loss - float(fx["loss"])) / abs(float(fx["loss"]))))
g1 = [p.grad.copy() for p in m.parameters()]
for p, f in zip(m.parameters(), fp):
    print("%-60s rel err %.3e" % (p._name, np.max(np.abs(p.grad - f["grad"])) / np.max(np.abs(f["grad"]))))
import hashlib
print("checksum", hashlib.sha1(b"".join(np.ascontiguousarray(g).tobytes() for g in g1) + np.float64(loss).tobytes()).hexdigest())
loss2 = float(m.loss())
print("bitwise repeat:", loss2 == loss and all(np.array_equal(p.grad, g) for p, g in zip(m.parameters(), g1)))
