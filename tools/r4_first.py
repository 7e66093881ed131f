"""wall clock of the FIRST gradient evaluation of a fresh model (allocation, first touch of the work matrices) against the following ones"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mogptk_amd import gpr, synth
def model(N, C=4, Q=3):
    X, y = synth.make_data(N, C); h = synth.mosm_hypers(C, Q)
    k = gpr.MultiOutputSpectralMixtureKernel(Q=Q, output_dims=C)
    for name in ("weight", "mean", "variance", "delay", "phase"): getattr(k, name).assign(h[name])
    m = gpr.Exact(k, X, y, variance=h["scale"] ** 2); m.likelihood.scale.assign(h["scale"]); return m
for N in [int(v) for v in sys.argv[1:]]:
    m = model(N)
    ts = []
    for i in range(4):
        t0 = time.perf_counter(); m.loss(); ts.append(1e3 * (time.perf_counter() - t0))
    print("N=%d MOGP_FLOW=%s: %s ms  %s" % (N, os.environ.get("MOGP_FLOW", "1"), " ".join("%.1f" % t for t in ts), m._handle.schedule()), flush=True)
