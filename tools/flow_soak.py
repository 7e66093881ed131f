"""Soak test of the dataflow schedule: the SAME evaluation (configs[1], fixed parameters) n times -- every loss and every gradient must come back with
identical bits, and the schedule must not have fallen back.  Reports the slowest evaluations (a hand-off that stalls shows up as one).
usage: python tools/flow_soak.py [n] [config] [replay|-] [N]"""
import os, sys, time, hashlib
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

n = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
cfg = sys.argv[2] if len(sys.argv) > 2 else "cfg2"
replay = len(sys.argv) > 3 and sys.argv[3] == "replay"      # the dataflow kernel ALONE (mogp_model_flow_replay): no chain kernels, no private-stream launches
n_override = int(sys.argv[4]) if len(sys.argv) > 4 else None      # another N for the same kernel and channel count (cfg2 only)
m, step, _ = bench.build_model(cfg, 0, n_override)
if replay:
    step()
    m._handle.flow_replay(True)
seen, times = {}, []
t_all = time.perf_counter()
for i in range(n):
    t0 = time.perf_counter()
    out = step()
    dt = time.perf_counter() - t0
    if cfg == "cfg4":
        key = hashlib.sha1(np.ascontiguousarray(out[0]).tobytes() + np.ascontiguousarray(out[1]).tobytes()).hexdigest()
    else:
        key = hashlib.sha1(np.float64(out).tobytes() + b"".join(np.ascontiguousarray(p.grad).tobytes() for p in m.parameters())).hexdigest()
    seen.setdefault(key, []).append(i)
    times.append(dt)
    h = m._handle
    s = h.schedule() if hasattr(h, "schedule") else {}
    if s.get("dataflow_fell_back") and "fell_at" not in seen:
        seen["fell_at"] = [i]
        print("evaluation %d: the dataflow schedule fell back (%.1f ms)" % (i, 1e3 * dt), flush=True)
fell = seen.pop("fell_at", None)
t = np.array(times)
print("%s: %d evaluations in %.1f s; distinct results: %d %s" % (cfg, n, time.perf_counter() - t_all, len(seen), {k[:8]: (len(v), v[:3]) for k, v in seen.items()}))
print("time per evaluation: median %.2f ms, max %.1f ms at #%d; the five slowest: %s" % (1e3 * np.median(t), 1e3 * t.max(), int(t.argmax()), ", ".join("%d: %.1f" % (j, 1e3 * t[j]) for j in np.argsort(-t)[:5])))
print("schedule at the end:", m._handle.schedule() if hasattr(m._handle, "schedule") else None, "fell back at", fell)
if hasattr(m._handle, "flow_diag"):
    print("deep looks:", m._handle.flow_diag())
