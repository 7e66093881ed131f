"""LML + gradient evaluation time of a 3-channel MOSM at several sizes, for the schedule chosen by MOGP_GRAD_PATH (fused | phases | unset).
usage: MOGP_GRAD_PATH=phases python tools/grad_path_sizes.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mogptk_amd as mogptk
rng = np.random.default_rng(0)
out = []
for n in [int(v) for v in os.environ.get("SIZES", "256,512,1024,1536,2048,2560,3072,4096").split(",")]:
    t = np.sort(rng.uniform(0, 50, n))
    ys = [np.sin(0.5 * t + c) + 0.1 * rng.standard_normal(n) for c in range(3)]
    m = mogptk.MOSM(mogptk.DataSet(t, ys), Q=2)
    m.init_parameters("LS")
    for _ in range(60): m.gpr.loss()                 # short runs on an idle GPU are bimodal (clock ramp): warm up well
    t0 = time.perf_counter()
    for _ in range(150): m.gpr.loss()
    sch = m.gpr._handle.schedule()
    out.append("%d:%.2f%s" % (3 * n, 1e3 * (time.perf_counter() - t0) / 150, ("[flow]" if sch["dataflow"] else "") + ("" if m.gpr._handle.inverse_fraction() >= 0.999 else "[inv %.2f]" % m.gpr._handle.inverse_fraction())))
print(os.environ.get("MOGP_GRAD_PATH", "default"), " ".join(out))
