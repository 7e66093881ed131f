"""The host side of one training step of the headline configuration, by the wall clock (no profiler in the loop): Python in front of the native evaluation (raw ->
constrained parameters -> term table -> mogp_model_set_terms), the native call, Python behind it (moments -> chain rule -> raw gradients), the Adam update.
usage: python tools/host_profile.py [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
m, run_step, _ = bench.build_model("cfg2", 0)
step = bench.training_step(m, run_step, "exact")
for _ in range(10):
    step()
h = m._handle
T = dict(set_terms0=[], set_terms1=[], eval0=[], eval1=[])
o_eval, o_set = h.eval, h.set_terms
def eval_(*a, **k):
    T["eval0"].append(time.perf_counter()); r = o_eval(*a, **k); T["eval1"].append(time.perf_counter()); return r
def set_(*a, **k):
    T["set_terms0"].append(time.perf_counter()); r = o_set(*a, **k); T["set_terms1"].append(time.perf_counter()); return r
h.eval, h.set_terms = eval_, set_
from mogptk_amd.model import _Adam
opt = _Adam(list(m.parameters()), lr=1e-6)
marks, A = [], []
for _ in range(steps):                      # every evaluation sees new parameters, as in training (an unchanged parameter set hits the term-table memo)
    t0 = time.perf_counter(); run_step(); t1 = time.perf_counter(); opt.step(); t2 = time.perf_counter()
    marks.append((t0, t1)); A.append(t2 - t0)
n = steps
us = lambda v: 1e6 * float(np.median(v))
pre = [T["set_terms0"][i] - marks[i][0] for i in range(n)]
st = [T["set_terms1"][i] - T["set_terms0"][i] for i in range(n)]
mid = [T["eval0"][i] - T["set_terms1"][i] for i in range(n)]
nat = [T["eval1"][i] - T["eval0"][i] for i in range(n)]
post = [marks[i][1] - T["eval1"][i] for i in range(n)]
loss_only = [b - a for a, b in marks]
print("loss() alone: %.1f us = python before set_terms %.1f + set_terms %.1f + python up to the native call %.1f + native evaluation %.1f + python behind it %.1f" %
      (us(loss_only), us(pre), us(st), us(mid), us(nat), us(post)))
print("loss() + Adam step: %.1f us  (the optimiser's share: %.1f us)" % (us(A), us(A) - us(loss_only)))
