"""
Golden vector for BASELINE.json configs[2]: MOSM C=8 Q=5 N=32768 (exact model).  Build container only.

The reference cannot back-propagate at this size (its autograd working set is far beyond the container's 62 GB), but its FORWARD
pass under torch.no_grad() fits (K, the dense eye it keeps, the noise product, the diagflat jitter, L: 45-50 GB).  So cfg3.npz pins

  lml_ref     Exact.log_marginal_likelihood() of the reference itself (/root/reference/mogptk/gpr/model.py:438-453) at the seeded
              inputs of mogptk_amd/synth.py, its raw parameter values, wall time and peak memory;
  fd_ref      the reference's own central difference of that LML along ONE seeded raw-space direction (two more forward runs):
              a derivative of the reference that no code of this repository took part in;
  lml_oracle, p*_grad   the numpy oracle (oracle/table_model.py:TableDeviceLean: the Gram from the term table, LAPACK potrf / potri in place
              through torch (MKL: the OpenBLAS build scipy bundles stops with info = 16545 on this matrix), moments of G = 1/2 (alpha alpha^T - Kj^-1), then the host chain rule) -- all 208 raw gradients.
              lml_oracle must agree with lml_ref (asserted here to 1e-11) and the gradient's projection on the direction with fd_ref.

Stages run as separate processes so that each one's memory is gone before the next starts:
    python tests/golden/gen_cfg3.py            # ref, fd, oracle, merge
    python tests/golden/gen_cfg3.py --stage oracle
"""
import os
import sys
import time
import argparse
import resource
import subprocess
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
TMP = os.environ.get("CFG3_TMP", "/tmp/cfg3_parts")
C, Q, N = 8, 5, 32768
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import cfg3_direction as direction, CFG3_FD_SEED as FD_SEED, CFG3_FD_EPS as FD_EPS     # noqa: E402  (shared with the GPU test)


def peak_gb():
    return resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1048576.0


def stage_ref(fd):
    sys.path.insert(0, HERE)
    import gen_golden as gg                     # imports the reference with the IPython stub
    import torch
    from mogptk_amd import synth
    X, y = synth.make_data(N, C)
    m = gg.ref_mosm(X, y, synth.mosm_hypers(C, Q), C, Q)
    params = list(m.parameters())
    out = {}
    if not fd:
        gg.dump_params("", params, out)
        t = time.time()
        with torch.no_grad():
            lml = float(m.log_marginal_likelihood())
        out.update(lml_ref=np.array(lml), ref_seconds=np.array(time.time() - t), ref_threads=np.array(torch.get_num_threads()),
                   ref_peak_gb=np.array(peak_gb()))
        print("reference LML %.12f  (%.0f s, peak %.1f GB)" % (lml, time.time() - t, peak_gb()), flush=True)
        np.savez(os.path.join(TMP, "ref.npz"), **out)
        return
    raw0 = [p.data.detach().clone() for p in params]
    d = direction([tuple(r.shape) for r in raw0])
    vals = []
    for sgn in (+1.0, -1.0):
        with torch.no_grad():
            for p, r, v in zip(params, raw0, d):
                p.data.copy_(r + sgn * FD_EPS * torch.tensor(v, dtype=r.dtype))
            vals.append(float(m.log_marginal_likelihood()))
        print("reference LML at %+g along the direction: %.12f" % (sgn * FD_EPS, vals[-1]), flush=True)
    np.savez(os.path.join(TMP, "fd.npz"), fd_ref=np.array((vals[0] - vals[1]) / (2.0 * FD_EPS)), fd_vals=np.array(vals),
             fd_eps=np.array(FD_EPS), fd_seed=np.array(FD_SEED))


def stage_oracle():
    sys.path.insert(0, ROOT)
    from mogptk_amd import gpr, synth, _lib
    from oracle.table_model import TableDeviceLean
    X, y = synth.make_data(N, C)
    h = synth.mosm_hypers(C, Q)
    k = gpr.MultiOutputSpectralMixtureKernel(Q=Q, output_dims=C)
    for name in ("weight", "mean", "variance", "delay", "phase"):
        getattr(k, name).assign(h[name])
    m = gpr.Exact(k, X, y, variance=h["scale"] ** 2)
    m.likelihood.scale.assign(h["scale"])
    _lib.ExactHandle = TableDeviceLean           # the host chain rule of the product over the numpy twin of the device
    t = time.time()
    loss = float(m.loss())
    out = dict(lml_oracle=np.array(-loss), oracle_seconds=np.array(time.time() - t), oracle_peak_gb=np.array(peak_gb()))
    params = list(m.parameters())
    for n, p in enumerate(params):
        out["p%d_grad" % n] = np.array(p.grad)
        out["p%d_oracle_raw" % n] = np.array(p.data)
    d = direction([p.data.shape for p in params])
    out["gd_oracle"] = np.array(sum(float(np.sum(p.grad * v)) for p, v in zip(params, d)))
    print("oracle LML %.12f  g.d %.12f (%.0f s, peak %.1f GB)" % (-loss, float(out["gd_oracle"]), time.time() - t, peak_gb()), flush=True)
    np.savez(os.path.join(TMP, "oracle.npz"), **out)


def merge():
    ref = dict(np.load(os.path.join(TMP, "ref.npz")))
    fd = dict(np.load(os.path.join(TMP, "fd.npz")))
    orc = dict(np.load(os.path.join(TMP, "oracle.npz")))
    out = {"meta": np.array([C, Q, 1, 1, N])}
    out.update(ref)
    out.update(fd)
    names = [str(s) for s in ref["names"]]
    for n in range(len(names)):
        raw_ref, raw_orc = ref["p%d_raw" % n], orc.pop("p%d_oracle_raw" % n)
        assert np.max(np.abs(raw_ref - raw_orc)) <= 1e-12 * max(1.0, np.max(np.abs(raw_ref))), names[n]
    out.update(orc)
    lr, lo = float(out["lml_ref"]), float(out["lml_oracle"])
    gd, fdr = float(out["gd_oracle"]), -float(out["fd_ref"])           # the gradients are of the LOSS = -LML
    print("LML reference %.12f oracle %.12f  rel %.2e" % (lr, lo, abs(lr - lo) / abs(lr)))
    print("d loss / d direction: oracle gradient %.10f, reference central difference %.10f  rel %.2e" % (gd, fdr, abs(gd - fdr) / abs(gd)))
    assert abs(lr - lo) <= 1e-11 * abs(lr)
    assert abs(gd - fdr) <= 1e-6 * abs(gd)
    np.savez_compressed(os.path.join(HERE, "cfg3.npz"), **out)
    print("wrote cfg3.npz")


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--stage", default="all", choices=["all", "ref", "fd", "oracle", "merge"])
    a = ap.parse_args()
    os.makedirs(TMP, exist_ok=True)
    if a.stage == "all":
        for s in ("ref", "fd", "oracle", "merge"):
            subprocess.run([sys.executable, os.path.abspath(__file__), "--stage", s], check=True)
    elif a.stage == "ref":
        stage_ref(False)
    elif a.stage == "fd":
        stage_ref(True)
    elif a.stage == "oracle":
        stage_oracle()
    else:
        merge()
