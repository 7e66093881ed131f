"""Time of one gpr loss() (LML + gradient on the device + host chain rule) and of one Model.train iteration (plus Adam and bookkeeping) at
three sizes of a 3-channel MOSM.  MI355X: N = 1500: 1.70 / 1.87 ms, N = 4500: 5.43 / 5.75 ms, N = 8190: 13.06 / 13.76 ms.
usage: python tools/train_step_time.py"""
import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
import mogptk_amd as mogptk
rng = np.random.default_rng(0)
for n in (500, 1500, 2730):
    t = np.sort(rng.uniform(0, 50, n))
    ys = [np.sin(0.5 * t + c) + 0.1 * rng.standard_normal(n) for c in range(3)]
    m = mogptk.MOSM(mogptk.DataSet(t, ys), Q=2)
    m.init_parameters("LS")
    m.gpr.loss(); m.gpr.loss()
    t0 = time.perf_counter()
    for _ in range(20): m.gpr.loss()
    tl = (time.perf_counter() - t0) / 20
    t0 = time.perf_counter()
    m.train("Adam", iters=20, lr=0.01)
    tt = (time.perf_counter() - t0) / 20
    print("N=%5d  loss() %.2f ms   train iteration %.2f ms" % (3 * n, 1e3 * tl, 1e3 * tt))
