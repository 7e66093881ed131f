"""Timings of every BASELINE.json config on one MI355X (wall clock per call, after warm-up; host chain rule and transfers included).
cfg2 is bench.py's workload; this script covers the others.  usage: python tools/bench_configs.py [cfg1 cfg3 cfg4 cfg5]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mogptk_amd import gpr, synth

def timeit(f, warm, reps):
    for _ in range(warm): f()
    t = time.perf_counter()
    for _ in range(reps): f()
    return (time.perf_counter() - t) / reps

def mosm(N, C, Q):
    X, y = synth.make_data(N, C); h = synth.mosm_hypers(C, Q)
    k = gpr.MultiOutputSpectralMixtureKernel(Q=Q, output_dims=C)
    for n in ("weight", "mean", "variance", "delay", "phase"): getattr(k, n).assign(h[n])
    return k, X, y, h

want = sys.argv[1:] or ["cfg1", "cfg3", "cfg4", "cfg5"]
out = {}
if "cfg1" in want:      # single-output SM Q=3, N=144 (airline-passenger sized): exact LML + gradient
    X, y = synth.make_data(144, 1); h = synth.sm_hypers(1, 3)
    k = gpr.SpectralMixtureKernel(Q=3, input_dims=1)
    k.magnitude.assign(h["magnitude"][0]); k.mean.assign(h["mean"][0]); k.variance.assign(h["variance"][0])
    m = gpr.Exact(gpr.IndependentMultiOutputKernel([k], output_dims=1), X, y, variance=h["scale"] ** 2)
    m.likelihood.scale.assign(h["scale"])
    out["cfg1_sm_q3_n144_lml_grad_ms"] = 1e3 * timeit(lambda: m.loss(), 5, 50)
if "cfg3" in want:
    k, X, y, h = mosm(32768, 8, 5)
    m = gpr.Exact(k, X, y, variance=h["scale"] ** 2); m.likelihood.scale.assign(h["scale"])
    out["cfg3_mosm_c8_q5_n32768_lml_grad_ms"] = 1e3 * timeit(lambda: m.loss(), 1, 3)
    del m
if "cfg4" in want:
    C, Q, N, S = 4, 3, 16384, 4096
    X, y = synth.make_data(N, C); h = synth.csm_hypers(C, Q)
    k = gpr.MixtureKernel(gpr.CrossSpectralKernel(output_dims=C, input_dims=1, Rq=1), Q)
    for q in range(Q):
        k[q].amplitude.assign(h["amplitude"][q]); k[q].mean.assign(h["mean"][q])
        k[q].variance.assign(h["variance"][q]); k[q].shift.assign(h["shift"][q])
    m = gpr.Exact(k, X, y, variance=h["scale"] ** 2); m.likelihood.scale.assign(h["scale"])
    Xs = synth.test_inputs(S, C)
    out["cfg4_csm_n16384_predict_s4096_ms"] = 1e3 * timeit(lambda: m.predict_f(Xs), 1, 3)
    del m
if "cfg5" in want:
    C, Q, N, M = 4, 3, 100000, 2048
    k, X, y, h = mosm(N, C, Q)
    s = float(np.mean(h["scale"]))
    m = gpr.Titsias(k, X, y, Z=[M // C] * C, variance=s ** 2); m.likelihood.scale.assign(s)
    out["cfg5_titsias_n100000_m2048_elbo_grad_ms"] = 1e3 * timeit(lambda: m.loss(), 2, 5)
print(json.dumps(out))
