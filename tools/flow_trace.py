"""Timeline of ONE dataflow evaluation (csrc/flow.hip) from the kernel's own time stamps (MOGP_FLOW_TRACE=1): per outer block the chain kernel's
wait / run, per queue when its tasks of that block ran, the workgroup slots' occupancy, idle time of the private queue.
usage: python tools/flow_trace.py [N=8192] [C=4] [Q=3]     (run on the GPU box; prints a text report -> profiles/r4_cfg2_timeline.txt)"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["MOGP_FLOW_TRACE"] = "1"
os.environ.setdefault("MOGP_GRAD_PATH", "fused")
import numpy as np
from mogptk_amd import gpr, synth, _lib

QN = ["look2", "semi", "invcrit", "vec"]


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
    C = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    Q = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    X, y = synth.make_data(N, C)
    h = synth.mosm_hypers(C, Q)
    k = gpr.MultiOutputSpectralMixtureKernel(Q=Q, output_dims=C)
    for name in ("weight", "mean", "variance", "delay", "phase"):
        getattr(k, name).assign(h[name])
    m = gpr.Exact(k, X, y, variance=h["scale"] ** 2)
    m.likelihood.scale.assign(h["scale"])
    for _ in range(6):
        m.loss()
    l = _lib.lib()
    hd = m._handle
    cnt = ctypes.c_int64(0)
    _lib.check(l.mogp_flow_trace(hd._h, None, 0, ctypes.byref(cnt)))
    if cnt.value == 0:
        print("no trace: the evaluation did not run as dataflow")
        return 1
    tr = np.zeros(cnt.value, dtype=np.int64)
    _lib.check(l.mogp_flow_trace(hd._h, tr.ctypes.data_as(_lib.c_i64p), tr.size, ctypes.byref(cnt)))
    nb = (X.shape[0] + 127) // 128
    pc = ctypes.c_int64(0)
    l.mogp_flow_plan(nb, None, 0, ctypes.byref(pc))
    rows = np.zeros((pc.value, 24), dtype=np.int64)
    l.mogp_flow_plan(nb, rows.ctypes.data_as(_lib.c_i64p), rows.size, ctypes.byref(pc))
    tasks = rows[rows[:, 0] >= 0]
    nt, no = len(tasks), int((rows[:, 0] == -1).sum())
    if os.environ.get("FLOW_TRACE_SAVE"):
        np.savez_compressed(os.environ["FLOW_TRACE_SAVE"], trace=tr, rows=rows)
    t = tr[:6 * nt].reshape(nt, 6)
    ch = tr[6 * nt:6 * nt + 4 * no].reshape(no, 4)
    t00 = min(int(ch[0, 0]), int(t[:, 1][t[:, 1] > 0].min()))
    us = lambda v: (np.asarray(v, dtype=np.float64) - t00) / 100.0
    look, st, k0, k1, en, wg = us(t[:, 0]), us(t[:, 1]), us(t[:, 2]), us(t[:, 3]), us(t[:, 4]), t[:, 5] & 0xffff
    naps = t[:, 5] >> 32
    gem = (tasks[:, 11] & 32) == 0
    print('looks that found nothing at first: %.1f %% of the tasks; their wait: mean %.1f us; direct looks: mean %.1f us, median %.1f us'
          % (100.0 * (naps > 0).mean(), (st - look)[naps > 0].mean() if (naps > 0).any() else 0.0, (st - look)[naps == 0].mean(), np.median((st - look)[naps == 0])))
    dur = en - st
    print("per task (mean us): looking for work %.1f | taken -> k loop (acquire, C tile, first operands) %.1f | k loop %.1f | stores + drain + counters %.1f"
          % ((st - look).mean(), (k0 - st)[gem].mean(), (k1 - k0)[gem].mean(), (en - k1)[gem].mean()))
    full = tasks[:, 12] == 32
    print("full-K tiles only:  looking %.1f | prologue %.1f | k loop %.1f | epilogue %.1f"
          % ((st - look)[full].mean(), (k0 - st)[full].mean(), (k1 - k0)[full].mean(), (en - k1)[full].mean()))
    blk = tasks[:, 1] // 1024
    # queues 4 .. : INTO[d] / TRAIL[d - 3] by deadline, the last one the accumulations: fold them into three report classes
    nq = int(tasks[:, 0].max()) + 1
    cls = tasks[:, 0].copy()
    rest = tasks[:, 0] >= 4
    cls[rest & (tasks[:, 8] == 2)] = 4
    cls[rest & (tasks[:, 8] == 0)] = 5
    cls[tasks[:, 0] == nq - 1] = 6
    tasks = tasks.copy(); tasks[:, 0] = cls
    QN.extend(["into", "trail", "acc"])
    print("dataflow evaluation N=%d (nb=%d tiles, %d outer blocks): %d tile tasks, %d workgroups seen; times in us from the first chain launch"
          % (X.shape[0], nb, no, nt, len(np.unique(wg))))
    print("end of the last task %.0f us; chain kernels end %.0f us" % (en.max(), us(ch[-1, 2])))
    print("\nchain kernels (private stream): launch, wait over, end | waited, ran | idle between end of previous and launch of this one")
    for b in range(no):
        a = us(ch[b])
        print("  block %2d  %8.0f %8.0f %8.0f | %6.0f %6.0f | %6.0f" % (b, a[0], a[1], a[2], a[1] - a[0], a[2] - a[1], a[0] - (us(ch[b - 1, 2]) if b else 0.0)))
    print("  sum of waits %.0f us, of runs %.0f us" % (sum(us(ch[b, 1]) - us(ch[b, 0]) for b in range(no)), sum(us(ch[b, 2]) - us(ch[b, 1]) for b in range(no))))
    print("\nqueues: tasks, mean / p95 duration (us), busy workgroup-time (ms)")
    for q in range(len(QN)):
        s = tasks[:, 0] == q
        if s.any():
            print("  %-8s %6d  %7.1f %7.1f  %8.2f" % (QN[q], s.sum(), dur[s].mean(), np.percentile(dur[s], 95), dur[s].sum() / 1e3))
    print("  per full-K tile (kt = 32): mean %.1f us" % dur[tasks[:, 12] == 32].mean())
    print("\nper block: first start .. last end of each queue's tasks of that block (us)")
    for b in range(no):
        line = "  block %2d " % b
        for q in range(len(QN)):
            s = (tasks[:, 0] == q) & (blk == b)
            line += " %s %5.0f..%-5.0f" % (QN[q][:5], st[s].min(), en[s].max()) if s.any() else " %s     -      " % QN[q][:5]
        print(line)
    # occupancy of the workgroup slots per 250 us window
    T = en.max()
    nwg = len(np.unique(wg))
    print("\nbusy fraction of the %d workgroup slots and TFLOP/s per 500 us window" % nwg)
    flop = 2.0 * 128 * 128 * 16 * tasks[:, 12]
    for w0 in np.arange(0.0, T, 500.0):
        w1 = w0 + 500.0
        ov = np.clip(np.minimum(en, w1) - np.maximum(st, w0), 0.0, None)
        fr = ov / np.maximum(dur, 1e-9)
        print("  %6.0f..%-6.0f  busy %.3f   %.1f TFLOP/s" % (w0, w1, ov.sum() / (nwg * 500.0), (fr * flop).sum() / 500e-6 / 1e12))
    gaps = []
    for g in np.unique(wg):
        s = wg == g
        o = np.argsort(st[s])
        gaps.append(st[s][o][1:] - en[s][o][:-1])
    gaps = np.concatenate(gaps)
    print("\ngap between consecutive tasks of a workgroup: median %.1f us, mean %.1f us, p95 %.1f us, sum / workgroup %.2f ms"
          % (np.median(gaps), gaps.mean(), np.percentile(gaps, 95), gaps.sum() / nwg / 1e3))
    # the number that does not depend on where the time stamps of a task begin and end (round 5's kernel signals a tile from inside the NEXT
    # task's prologue, so "claimed .. signalled" intervals of one workgroup overlap): the time a workgroup spends BETWEEN two k loops
    kgaps, direct = [], []
    for g in np.unique(wg):
        s = (wg == g) & gem
        o = np.argsort(k0[s])
        d = k0[s][o][1:] - k1[s][o][:-1]
        kgaps.append(d)
        direct.append(d[naps[s][o][1:] == 0])
    kgaps, direct = np.concatenate(kgaps), np.concatenate(direct)
    print("between two k loops of a workgroup (stores, look, descriptor, C tile, first operands): median %.1f us, mean %.1f us, p95 %.1f us; "
          "when the look found something at once (%.1f %% of the tasks): median %.1f us, mean %.1f us"
          % (np.median(kgaps), kgaps.mean(), np.percentile(kgaps, 95), 100.0 * len(direct) / max(len(kgaps), 1), np.median(direct), direct.mean()))
    kin = np.clip(k1 - k0, 0.0, None)[gem].sum()
    print("workgroup slots inside a k loop: %.3f of %d slots x %.0f us" % (kin / (nwg * T), nwg, T))
    return 0


if __name__ == "__main__":
    sys.exit(main())
