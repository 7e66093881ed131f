"""The two tile kernels of the exact path by themselves: Gram build (k_gram_strip) and gradient-moment pass (k_moments_x), HIP events around the kernel inside
evaluations on the phases schedule (one Gram launch, one moment launch per evaluation), median of `reps`; bytes = the lower triangle 4 N (N + 1).
usage: python tools/tile_kernels_time.py [N] [C] [Q] [reps]   (environment switches of the kernels apply: MOGP_STRIP_RUN, MOGP_GRAM_NC, ...)"""
import json, os, sys
import numpy as np
os.environ.setdefault("MOGP_GRAD_PATH", "phases")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from mogptk_amd import _lib

N = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
C = int(sys.argv[2]) if len(sys.argv) > 2 else 4
Q = int(sys.argv[3]) if len(sys.argv) > 3 else 3
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 30
m = bench.build_mosm(N, C, Q, 0)
m.loss()
h = m._handle
h.set_profiling(True)
g, mo = [], []
for _ in range(reps):
    m.loss()
    ms, _, _ = h.stage_ms()
    g.append(ms[_lib.ST_GRAM_KERNEL]); mo.append(ms[_lib.ST_MOMENT_KERNEL])
b = 4.0 * N * (N + 1)
gm, mm = float(np.median(g)), float(np.median(mo))
print(json.dumps(dict(N=N, C=C, Q=Q, gram_us=1e3 * gm, gram_min_us=1e3 * min(g), gram_frac_hbm=b / (gm * 1e-3) / 8e12, moments_us=1e3 * mm, moments_min_us=1e3 * min(mo),
                      moments_frac_hbm=b / (mm * 1e-3) / 8e12, env={k: v for k, v in os.environ.items() if k.startswith("MOGP_")})))
