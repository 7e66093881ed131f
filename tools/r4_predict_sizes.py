"""predict_f wall clock (CSM C=4 Q=3, S = N / 4 test points) at several sizes; run once with MOGP_FLOW_PREDICT=0 for the stream form"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
out = []
for N in [int(v) for v in os.environ.get("SIZES", "1024,2048,4096,8192,12288,16384,20480").split(",")]:
    bench.CONFIGS["cfg4"] = ("predict", 4, 3, N, max(128, N // 4), "")
    m, step, _ = bench.build_model("cfg4", 0)
    for _ in range(6): step()
    n = 20 if N <= 8192 else 8
    t0 = time.perf_counter()
    for _ in range(n): step()
    out.append("%d:%.2f%s" % (N, 1e3 * (time.perf_counter() - t0) / n, "[flow]" if m._handle.schedule()["dataflow"] else ""))
print("MOGP_FLOW_PREDICT=%s" % os.environ.get("MOGP_FLOW_PREDICT", "default"), " ".join(out))
