#!/bin/bash
# the fused schedule as dataflow against POTRF / TRTRI / LAUUM around the default threshold, on configs[1]'s kernel (full inverse) and on the LS-initialised MOSM of grad_path_sizes.py
cd $GRAFT_REPO_ROOT
python - <<'PY'
import os, sys, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
from mogptk_amd import gpr, synth
def model(N, C=4, Q=3):
    X, y = synth.make_data(N, C); h = synth.mosm_hypers(C, Q)
    k = gpr.MultiOutputSpectralMixtureKernel(Q=Q, output_dims=C)
    for name in ("weight", "mean", "variance", "delay", "phase"): getattr(k, name).assign(h[name])
    m = gpr.Exact(k, X, y, variance=h["scale"] ** 2); m.likelihood.scale.assign(h["scale"]); return m
for N in [int(v) for v in os.environ.get("SIZES", "9216,10240,11264,12288,13312,14336").split(",")]:
    m = model(N)
    line = "N=%d" % N
    for flow in ("1", "0"):
        os.environ["MOGP_FLOW"] = flow
        for _ in range(12): m.loss()
        t0 = time.perf_counter()
        for _ in range(30): m.loss()
        dt = 1e3 * (time.perf_counter() - t0) / 30
        s = m._handle.schedule()
        line += "  MOGP_FLOW=%s %.2f ms %s inv %.2f" % (flow, dt, "flow" if s["dataflow"] else "no-flow", m._handle.inverse_fraction())
    print(line, flush=True)
PY
