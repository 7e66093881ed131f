"""Gradient evaluation on a series ten times longer than BASELINE configs[1]'s (same N, C, Q and hyperparameters, inputs over [0, 1000]):
the kernel's support is then a narrow band of the matrix and the evaluation forms only the tiles of Kj^-1 the gradient reads
(mogp_api.hip:kinv_plan).  Times it against MOGP_FULL_INVERSE=1 in two subprocesses and compares loss and gradients.
usage: python tools/long_series.py [N] [stretch]        (run on the GPU box)"""
import os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np


def child(n, stretch, out):
    from mogptk_amd import gpr, synth
    C, Q = 4, 3
    X, y = synth.make_data(n, C)
    X = X.copy(); X[:, 1] *= stretch
    h = synth.mosm_hypers(C, Q)
    k = gpr.MultiOutputSpectralMixtureKernel(Q=Q, output_dims=C)
    for name in ("weight", "mean", "variance", "delay", "phase"):
        getattr(k, name).assign(h[name])
    m = gpr.Exact(k, X, y, variance=h["scale"] ** 2)
    m.likelihood.scale.assign(h["scale"])
    loss = float(m.loss())
    grads = np.concatenate([p.grad.reshape(-1) for p in m.parameters()])
    frac = m._handle.inverse_fraction()
    for _ in range(5): m.loss()
    t0 = time.perf_counter()
    for _ in range(20): m.loss()
    np.savez(out, loss=loss, grads=grads, ms=1e3 * (time.perf_counter() - t0) / 20, frac=frac)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        child(int(sys.argv[2]), float(sys.argv[3]), sys.argv[4]); sys.exit(0)
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
    stretch = float(sys.argv[2]) if len(sys.argv) > 2 else 10.0
    res = []
    with tempfile.TemporaryDirectory() as tmp:
        for full in ("1", "0"):
            out = os.path.join(tmp, "r%s.npz" % full)
            subprocess.run([sys.executable, os.path.abspath(__file__), "--child", str(n), str(stretch), out], check=True, env=dict(os.environ, MOGP_FULL_INVERSE=full))
            res.append(dict(np.load(out)))
    a, b = res
    print("N=%d stretch %.0f: every tile %.3f ms | tiles formed %.2f of all: %.3f ms (x%.2f) | loss rel %.1e grad rel %.1e" % (
        n, stretch, float(a["ms"]), float(b["frac"]), float(b["ms"]), float(a["ms"]) / float(b["ms"]),
        abs(float(a["loss"]) - float(b["loss"])) / abs(float(a["loss"])), np.max(np.abs(a["grads"] - b["grads"])) / np.max(np.abs(a["grads"]))))
