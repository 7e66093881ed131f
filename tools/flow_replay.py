"""The dataflow kernel ALONE (mogp_model_flow_replay): one normal LML + gradient evaluation at BASELINE.json configs[1], then the same evaluation
replayed by the dataflow kernel without its chain kernels (their W_KK blocks left in place, their counters preset, the private stream's two products
as tile tasks).  Checks that the replay forms the SAME Kj^-1 and times the kernel; under `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (which
serialise dispatches: the normal evaluation then runs on the stream schedule, MOGP_FLOW=0) the replay's k_flow2 rows are the kernel's HBM traffic.
usage: python tools/flow_replay.py [N=8192] [reps=5]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch          # noqa: F401  (as bench.py does: with torch's copy of the HIP runtime in the process the profiler's exit hooks behave)
from mogptk_amd import gpr, synth, _lib


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    C, Q = 4, 3
    X, y = synth.make_data(N, C)
    h = synth.mosm_hypers(C, Q)
    k = gpr.MultiOutputSpectralMixtureKernel(Q=Q, output_dims=C)
    for name in ("weight", "mean", "variance", "delay", "phase"):
        getattr(k, name).assign(h[name])
    m = gpr.Exact(k, X, y, variance=h["scale"] ** 2)
    m.likelihood.scale.assign(h["scale"])
    serial = os.environ.get("FLOW_REPLAY_SERIAL") == "1"          # under a serialising profiler the normal evaluation must not wait for co-operating kernels
    if serial:
        os.environ["MOGP_FLOW"] = "0"
    loss0 = float(m.loss())
    g0 = [p.grad.copy() for p in m.parameters()]
    hd = m._handle
    print("normal evaluation: loss %.10f, schedule %s" % (loss0, hd.schedule()))
    Kinv0 = np.tril(hd.fetch(1))
    os.environ["MOGP_FLOW"] = "1"
    hd.flow_replay(True)
    m.loss()                                                        # (first replay: builds and uploads the replay plan)
    Kinv1 = np.tril(hd.fetch(1))
    g1 = [p.grad.copy() for p in m.parameters()]
    print("replay: schedule %s" % hd.schedule())
    print("Kj^-1 of the replay against the evaluation's: max |diff| %.3e (max |Kj^-1| %.3e), identical bits: %s"
          % (np.max(np.abs(Kinv1 - Kinv0)), np.max(np.abs(Kinv0)), bool(np.array_equal(Kinv1, Kinv0))))
    print("gradient of the replay against the evaluation's: %.3e relative" % max(np.max(np.abs(a - b)) / np.max(np.abs(b)) for a, b in zip(g1, g0)))
    hd.set_profiling(True)
    ts = []
    for _ in range(reps):
        m.loss()
        ms, nl, fl = hd.stage_ms()
        ts.append(ms[_lib.ST_GEMM_KERNEL])
    hd.set_profiling(False)
    print("dataflow kernel alone (no chain kernels to wait for): %s ms between HIP events; %.3e flop issued -> %.1f TFLOP/s"
          % (", ".join("%.3f" % t for t in ts), fl, fl / (min(ts) * 1e-3) / 1e12))
    hd.flow_replay(False)
    if serial:
        os.environ["MOGP_FLOW"] = "0"
    loss2 = float(m.loss())
    print("normal evaluation again: loss %.10f (same: %s)" % (loss2, loss2 == loss0))
    m._handle = None
    _lib.shutdown()                 # models and contexts go while the runtime is up (also registered with atexit: under rocprofv3 an implicit teardown crashed in the tool's exit hooks)
    return 0


if __name__ == "__main__":
    sys.exit(main())
