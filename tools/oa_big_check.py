import sys, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
from mogptk_amd import gpr, synth, _lib
from oracle.table_model import TableDevice
N, C, Q = 10368, 3, 2          # 81 tile rows: the POTRF / TRTRI / LAUUM schedule
rng = np.random.default_rng(1)
X, _ = synth.make_data(N, C)
X = X[rng.permutation(N)]
y = rng.standard_normal(N)
h = synth.mosm_hypers(C, Q)
k = gpr.MultiOutputSpectralMixtureKernel(Q=Q, output_dims=C)
for name in ("weight", "mean", "variance", "delay", "phase"):
    getattr(k, name).assign(h[name])
table = k._spectral_terms(1)
nu, lam = rng.normal(0, 0.5, N), rng.uniform(0.5, 3.0, N)
dev = _lib.ExactHandle(0, X, y, C); dev.set_terms(table)
t = time.time(); a = dev.oa_forward(nu, lam); e, f = rng.standard_normal(N), -rng.uniform(0.5, 2.0, N); ga = dev.oa_backward(e, f); print("device s", time.time() - t)
ref = TableDevice(0, X, y, C); ref.set_terms(table)
t = time.time(); b = ref.oa_forward(nu, lam); gb = ref.oa_backward(e, f); print("twin s", time.time() - t)
rel = lambda u, v: float(np.max(np.abs(u - v)) / np.max(np.abs(v)))
print("mu", rel(a["mu"], b["mu"]), "var", rel(a["var"], b["var"]), "kl", abs(a["kl"] - b["kl"]) / abs(b["kl"]))
for key in ("mom", "g_nu", "g_lambda"):
    print(key, rel(ga[key], gb[key]))
