import os, sys, time, cProfile, pstats, io
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import bench
m, run_step, _ = bench.build_model("cfg5", 0)
step = bench.training_step(m, run_step, "titsias")
for _ in range(5): step()
t=time.perf_counter()
for _ in range(30): step()
print("step %.3f ms" % (1e3*(time.perf_counter()-t)/30))
pr = cProfile.Profile(); pr.enable()
for _ in range(30): step()
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(14); print(s.getvalue()[:3500])
