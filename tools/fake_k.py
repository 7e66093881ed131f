"""Timing experiment (WRONG numbers on purpose): MOGP_FAKE_K=d makes every 128x128-tile GEMM launch contract 1/d of its k range.  If an
evaluation is bound by what its GEMM streams deliver its time falls with d; if by its dependency chain it does not.
usage: MOGP_FAKE_K=2 python tools/fake_k.py"""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
warnings.filterwarnings("ignore")
np.seterr(all="ignore")
import bench
m, run_step, _ = bench.build_model("cfg2", 0)
for _ in range(5): run_step()
t0 = time.perf_counter()
for _ in range(20): run_step()
print("MOGP_FAKE_K=%s: %.3f ms per evaluation" % (os.environ.get("MOGP_FAKE_K", "0"), 1e3 * (time.perf_counter() - t0) / 20))
