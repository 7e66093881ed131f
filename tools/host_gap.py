"""Where a bench step's wall clock goes on this box: Python before / after the native call, the native call itself, and the device's own
span of the evaluation (HIP events).  usage: python tools/host_gap.py [steps]"""
import sys, time
import numpy as np
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import bench
from mogptk_amd import _lib

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
m, step, _ = bench.build_model("cfg2", 0)
for _ in range(5):
    step()
h = m._handle
orig = h.eval
acc = {"native": 0.0, "n": 0}
def timed(*a, **k):
    t = time.perf_counter(); r = orig(*a, **k); acc["native"] += time.perf_counter() - t; acc["n"] += 1
    return r
h.eval = timed
t0 = time.perf_counter()
for _ in range(steps):
    step()
wall = (time.perf_counter() - t0) / steps
native = acc["native"] / max(acc["n"], 1)
h.eval = orig
_lib.check(_lib.lib().mogp_set_profiling(h._h, 1))
step()
ms = h.stage_ms() if hasattr(h, "stage_ms") else None
print("per step: wall %.3f ms = native call %.3f ms + python %.3f ms; device span of one evaluation: %s" % (wall * 1e3, native * 1e3, (wall - native) * 1e3, ms))
import os
print("cpu count", os.cpu_count(), "loadavg", os.getloadavg())
