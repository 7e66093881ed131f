"""creates / evaluates / drops many device handles in a row, printing progress (run under a hard `timeout`)"""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mogptk_amd import gpr, _lib
for it, (N, C, Q, D) in enumerate([(300, 3, 2, 1), (517, 4, 3, 1), (260, 2, 9, 1), (200, 3, 2, 2), (129, 1, 2, 1)] * 6):
    rng = np.random.default_rng(N)
    X = np.concatenate([rng.integers(0, C, (N, 1)).astype(float), rng.uniform(0, 30, (N, D))], axis=1)
    y = rng.standard_normal(N)
    k = gpr.MultiOutputSpectralMixtureKernel(Q=Q, output_dims=C, input_dims=D)
    k.weight.assign(rng.uniform(0.5, 1.5, (C, Q))); k.mean.assign(rng.uniform(0.02, 0.4, (C, Q, D)))
    k.variance.assign(rng.uniform(0.005, 0.05, (C, Q, D)))
    t = time.time()
    dev = _lib.ExactHandle(0, X, y, C)
    dev.set_terms(k._spectral_terms(D))
    a = dev.eval(rng.uniform(0.05, 0.2, C), 1e-8, grad=True)
    b = dev.eval(rng.uniform(0.05, 0.2, C), 1e-8, grad=False)
    dev.fetch(1)
    print(it, N, a["lml"], b["lml"], "%.3fs" % (time.time() - t), flush=True)
    del dev
print("done", flush=True)
