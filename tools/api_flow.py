"""End-to-end run of the model API on the device: MOSM on three channels of 1500 irregularly sampled points, every inference this package has,
30 Adam iterations each, prediction error against the noiseless signal.  (The Opper-Archambeau loss is not monotone under Adam at this
learning rate -- in the reference neither: on a 240-point version its first steps read 277.74, 3272.10, 242.29, 381.43 in both.)  usage: python tools/api_flow.py"""
import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
import mogptk_amd as mogptk
from mogptk_amd import gpr
rng = np.random.default_rng(0)
t = np.sort(rng.uniform(0, 50, 1500))
ys = [np.sin(0.5 * t) + 0.1 * rng.standard_normal(t.size), np.cos(0.5 * t + 0.3) + 0.1 * rng.standard_normal(t.size), np.sin(0.25 * t) + 0.1 * rng.standard_normal(t.size)]
ds = mogptk.DataSet(t, ys)
for name, inf in (("Exact", mogptk.Exact()), ("Titsias", mogptk.Titsias(inducing_points=64)), ("Snelson", mogptk.Snelson(inducing_points=64)),
                  ("Hensman", mogptk.Hensman(inducing_points=64)), ("Hensman+StudentT", mogptk.Hensman(inducing_points=64, likelihood=gpr.StudentTLikelihood(dof=4, scale=0.2))),
                  ("OpperArchambeau", mogptk.OpperArchambeau())):
    m = mogptk.MOSM(ds, Q=2, inference=inf)
    m.init_parameters("LS")            # (BNSE degenerates on irregularly sampled inputs -- in the reference too, to the same digits)
    t0 = time.time()
    losses, _ = m.train("Adam", iters=30, lr=0.05)
    dt = time.time() - t0
    _, mu, lo, hi = m.predict(transformed=False)
    rmse = float(np.sqrt(np.mean((np.asarray(mu[0]) - np.sin(0.5 * t)) ** 2)))
    print("%-18s loss %10.2f -> %10.2f  %6.1f ms/iter  rmse ch0 %.3f" % (name, losses[0], losses[-1], 1e3 * dt / 30, rmse))
