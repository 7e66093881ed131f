"""Timeline of ONE prediction run as tile dataflow (BASELINE.json configs[3]; MOGP_FLOW_TRACE=1): chain kernels, queues, busy fraction.
usage: python tools/flow_trace_predict.py      (on the GPU box)"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["MOGP_FLOW_TRACE"] = "1"
import numpy as np
import bench
from mogptk_amd import _lib
m, step, flops = bench.build_model("cfg4", 0)
for _ in range(4):
    step()
l = _lib.lib()
hd = m._handle
print("schedule:", hd.schedule())
cnt = ctypes.c_int64(0)
_lib.check(l.mogp_flow_trace(hd._h, None, 0, ctypes.byref(cnt)))
if cnt.value == 0:
    print("no trace: the prediction did not run as dataflow"); sys.exit(1)
tr = np.zeros(cnt.value, dtype=np.int64)
_lib.check(l.mogp_flow_trace(hd._h, tr.ctypes.data_as(_lib.c_i64p), tr.size, ctypes.byref(cnt)))
nb, rhs = (16384 + 127) // 128, (4096 + 128) // 128
pc = ctypes.c_int64(0)
l.mogp_flow_plan_rhs(nb, rhs, None, 0, ctypes.byref(pc))
rows = np.zeros((pc.value, 24), dtype=np.int64)
l.mogp_flow_plan_rhs(nb, rhs, rows.ctypes.data_as(_lib.c_i64p), rows.size, ctypes.byref(pc))
tasks = rows[rows[:, 0] >= 0]
nt, no = len(tasks), int((rows[:, 0] == -1).sum())
t = tr[:6 * nt].reshape(nt, 6)
ch = tr[6 * nt:6 * nt + 4 * no].reshape(no, 4)
t00 = min(int(ch[0, 0]), int(t[:, 1][t[:, 1] > 0].min()))
us = lambda v: (np.asarray(v, dtype=np.float64) - t00) / 100.0
st, en, wg = us(t[:, 1]), us(t[:, 4]), t[:, 5] & 0xffff
dur = en - st
nq = int(tasks[:, 0].max()) + 1
print("%d tile tasks in %d queues, %d workgroups; last task ends %.0f us, chain kernels end %.0f us" % (nt, nq, len(np.unique(wg)), en.max(), us(ch[-1, 2])))
per = [us(ch[b, 0]) - (us(ch[b - 1, 2]) if b else 0.0) for b in range(no)]
print("chain kernels: run mean %.0f us; idle between two of them: mean %.0f, max %.0f us" % (np.mean(us(ch[:, 2]) - us(ch[:, 1])), np.mean(per[1:]), np.max(per[1:])))
names = {0: "look2", 1: "semi", 2: "rhs-cycle", nq - 1: "rhs-rest"}
for q in range(nq):
    s = tasks[:, 0] == q
    if q in names or q in (3, nq - 2):
        print("  queue %2d %-9s %6d tasks, mean %.1f us, busy %.1f ms, runs %.0f .. %.0f us" % (q, names.get(q, "trail"), s.sum(), dur[s].mean(), dur[s].sum() / 1e3, st[s].min(), en[s].max()))
nwg = len(np.unique(wg))
flop = 2.0 * 128 * 128 * 16 * tasks[:, 12]
T = en.max()
for w0 in np.arange(0.0, T, 2000.0):
    w1 = w0 + 2000.0
    ov = np.clip(np.minimum(en, w1) - np.maximum(st, w0), 0.0, None)
    print("  %6.0f..%-6.0f  busy %.3f   %.1f TFLOP/s" % (w0, w1, ov.sum() / (nwg * 2000.0), ((ov / np.maximum(dur, 1e-9)) * flop).sum() / 2000e-6 / 1e12))
