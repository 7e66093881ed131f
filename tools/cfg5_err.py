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
try:
    tr = load("titsias_dz_truth_cfg5.npz")
    zp = [p for p in m.parameters() if p._name.endswith("induction_points")][0]
    gz, truth = -zp.grad[:, 1], tr["gz_truth"]
    for nm, g in (("device", gz), ("reference, 8 threads", tr["gz_ref"]), ("reference, 3 threads", tr["gz_ref_alt"])):
        print("dELBO/dZ against the 80-bit truth, %-22s max-norm %.3e  2-norm %.3e  1 - cos %.2e" % (nm, np.max(np.abs(g - truth)) / np.max(np.abs(truth)),
              np.linalg.norm(g - truth) / np.linalg.norm(truth), 1.0 - np.dot(g, truth) / np.linalg.norm(g) / np.linalg.norm(truth)))
    d = gz - truth
    blk = np.abs(d).reshape(C, -1)
    print("  device error by channel (max): " + ", ".join("%.2e" % v for v in blk.max(axis=1) / np.max(np.abs(truth))))
except FileNotFoundError:
    pass
import hashlib
print("checksum", hashlib.sha1(b"".join(np.ascontiguousarray(g).tobytes() for g in g1) + np.float64(loss).tobytes()).hexdigest())
loss2 = float(m.loss())
print("bitwise repeat:", loss2 == loss and all(np.array_equal(p.grad, g) for p, g in zip(m.parameters(), g1)))
