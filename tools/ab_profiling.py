import sys, time, os
sys.path.insert(0, os.getcwd())
import numpy as np
from bench import build_model
m, X, y = build_model(8192, 4, 3, 0)
for _ in range(3): m.loss()
h = m._handle
for prof in (False, True, False, True):
    h.set_profiling(prof)
    m.loss()
    t=time.perf_counter()
    for _ in range(10): m.loss()
    dt=(time.perf_counter()-t)/10
    print("profiling", prof, "ms/eval %.3f" % (dt*1e3))
# host-side overhead: time of python parts
import cProfile, pstats
h.set_profiling(False)
pr = cProfile.Profile(); pr.enable()
for _ in range(10): m.loss()
pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(12)
