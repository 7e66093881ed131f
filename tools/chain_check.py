"""A/B of the persistent chain kernel (chain.hip, MOGP_CHAIN=1) against the launch-per-step chain (MOGP_CHAIN=0): the same LML + gradient
evaluation in two subprocesses per size, compared value by value; on a mismatch, the tile map of the error of W = L^-1 says which hand-off
broke.  Also times both forms.
usage: python tools/chain_check.py [sizes, total N over 3 channels]      (run on the GPU box; exit code 1 on a mismatch)"""
import os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np


def child(n, out, reps):
    import ctypes
    from mogptk_amd import gpr, synth, _lib
    C, Q = 3, 2
    X, y = synth.make_data(n, C)
    h = synth.mosm_hypers(C, Q)
    k = gpr.MultiOutputSpectralMixtureKernel(Q=Q, output_dims=C)
    for name in ("weight", "mean", "variance", "delay", "phase"):
        getattr(k, name).assign(h[name])
    m = gpr.Exact(k, X, y, variance=h["scale"] ** 2)
    m.likelihood.scale.assign(h["scale"])
    os.environ["MOGP_GRAD_PATH"] = "fused"
    loss = float(m.loss())
    grads = np.concatenate([p.grad.reshape(-1) for p in m.parameters()])
    hd = m._handle
    W = np.zeros((X.shape[0], X.shape[0]))
    _lib.check(_lib.lib().mogp_model_fetch(hd._h, 0, W.ctypes.data_as(_lib.c_dp)))
    for _ in range(10): m.loss()
    t0 = time.perf_counter()
    for _ in range(reps): m.loss()
    ms = 1e3 * (time.perf_counter() - t0) / reps
    np.savez(out, loss=loss, grads=grads, W=W[:1536, :1536], ms=ms)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        child(int(sys.argv[2]), sys.argv[3], int(sys.argv[4]))
        sys.exit(0)
    # which switch is A/B-ed: AB="MOGP_CHAIN:0,1" (default) or e.g. AB="MOGP_LOOKAHEAD:1,2"; the first value is the reference
    AB_VAR, vals = os.environ.get("AB", "MOGP_CHAIN:0,1").split(":")
    AB_VALUES = vals.split(",")
    sizes = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "100,128,300,512,600,900,1500,2048,4097,8192").split(",")]
    bad = 0
    for n in sizes:
        res = {}
        for mode in ("0", "1"):
            f = tempfile.mktemp(suffix=".npz")
            env = dict(os.environ, MOGP_GRAD_PATH="fused")
            env[AB_VAR] = AB_VALUES[int(mode)]
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", str(n), f, "30" if n <= 4097 else "20"], env=env,
                               capture_output=True, text=True, timeout=600)
            if r.returncode != 0:
                print("N=%d %s=%s FAILED rc=%d: %s" % (n, AB_VAR, AB_VALUES[int(mode)], r.returncode, (r.stderr or r.stdout)[-600:]))
                res[mode] = None
                continue
            res[mode] = dict(np.load(f))
            os.unlink(f)
        a, b = res["0"], res["1"]
        if a is None or b is None:
            bad += 1
            continue
        dl = abs(a["loss"] - b["loss"]) / max(1.0, abs(a["loss"]))
        dg = np.max(np.abs(a["grads"] - b["grads"])) / max(1e-300, np.max(np.abs(a["grads"])))
        dW = np.max(np.abs(a["W"] - b["W"])) / max(1e-300, np.max(np.abs(a["W"])))
        ok = dl < 1e-9 and dg < 1e-6
        print("N=%5d  loss %.10g vs %.10g  rel %.1e | grad rel %.1e | W rel %.1e | ms/eval %s=%s %.3f  %s=%s %.3f  %s"
              % (n, a["loss"], b["loss"], dl, dg, dW, AB_VAR, AB_VALUES[0], a["ms"], AB_VAR, AB_VALUES[1], b["ms"], "ok" if ok else "MISMATCH"))
        if not ok:
            bad += 1
            nb = (a["W"].shape[0] + 127) // 128
            scale = np.max(np.abs(a["W"]))
            print("   tile map of |W_persistent - W_launches| / max|W| (rows = tile row, first %d tile rows):" % min(nb, 12))
            for i in range(min(nb, 12)):
                print("   " + " ".join("%8.1e" % (np.max(np.abs(a["W"][128 * i:128 * i + 128, 128 * j:128 * j + 128] -
                                                            b["W"][128 * i:128 * i + 128, 128 * j:128 * j + 128])) / scale) for j in range(i + 1)))
    sys.exit(1 if bad else 0)
