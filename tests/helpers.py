"""Shared test helpers: load golden fixtures, rebuild oracle specs and product models from them."""
import os
import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def bound(a):
    a = np.asarray(a)
    return None if (a.ndim == 0 and np.isnan(a)) else a


def fixture_params(fx, prefix=""):
    """-> list of dict(name, raw, lower, upper, cons, grad)"""
    names = [str(n) for n in fx[prefix + "names"]]
    out = []
    for n, name in enumerate(names):
        d = dict(name=name, raw=fx["%sp%d_raw" % (prefix, n)], lower=bound(fx["%sp%d_lower" % (prefix, n)]),
                 upper=bound(fx["%sp%d_upper" % (prefix, n)]), cons=fx["%sp%d_cons" % (prefix, n)])
        key = "%sp%d_grad" % (prefix, n)
        if key in fx:
            d["grad"] = bound(fx[key])
        out.append(d)
    return out


def oracle_pspec(kind, C, Q, D, Rq, params):
    """stack the reference's parameter list (registration order) into the oracle's pspec layout;
    returns (pspec, index map: pspec entry -> list of fixture param indices stacked on axis 0)"""
    ps = params[:-1]
    scale = params[-1]
    entries, imap = [], []

    def stack(idx):
        lo = [ps[i]["lower"] for i in idx]
        up = [ps[i]["upper"] for i in idx]
        raw = np.stack([ps[i]["raw"] for i in idx])

        def st(b):
            if all(v is None for v in b):
                return None
            return np.stack([np.broadcast_to(v, ps[idx[0]]["raw"].shape) for v in b])
        return raw, st(lo), st(up)

    if kind == "mosm":
        for n, name in enumerate(["weight", "mean", "variance", "delay", "phase"]):
            entries.append((name, ps[n]["raw"], ps[n]["lower"], ps[n]["upper"]))
            imap.append([n])
    elif kind == "sm":          # kernels[c].{magnitude, mean, variance}
        for k, name in enumerate(["magnitude", "mean", "variance"]):
            idx = [3 * c + k for c in range(C)]
            raw, lo, up = stack(idx)
            entries.append((name, raw, lo, up))
            imap.append(idx)
    elif kind == "csm":         # kernels[q].{amplitude, mean, variance, shift}
        for k, name in enumerate(["amplitude", "mean", "variance", "shift"]):
            idx = [4 * q + k for q in range(Q)]
            raw, lo, up = stack(idx)
            entries.append((name, raw, lo, up))
            imap.append(idx)
    entries.append(("scale", scale["raw"], scale["lower"], scale["upper"]))
    imap.append([len(params) - 1])
    pspec = dict(kind=kind, C=C, Q=Q, D=D, Rq=Rq, params=entries)
    return pspec, imap


def unstack_grads(grads, imap, nparams):
    """oracle grads (pspec layout) -> list in the reference's parameter order"""
    out = [None] * nparams
    for g, idx in zip(grads, imap):
        if len(idx) == 1:
            out[idx[0]] = g
        else:
            for k, i in enumerate(idx):
                out[i] = g[k]
    return out


def product_kernel(kind, C, Q, D, Rq):
    from mogptk_amd import gpr as g
    if kind == "mosm":
        return g.MultiOutputSpectralMixtureKernel(Q=Q, output_dims=C, input_dims=D)
    if kind == "sm":
        return g.IndependentMultiOutputKernel([g.SpectralMixtureKernel(Q=Q, input_dims=D) for _ in range(C)], output_dims=C)
    if kind == "csm":
        return g.MixtureKernel(g.CrossSpectralKernel(output_dims=C, input_dims=D, Rq=Rq), Q)
    if kind == "mosk":
        return g.MixtureKernel(g.MultiOutputSpectralKernel(output_dims=C, input_dims=D), Q)
    if kind == "lmc":
        return g.LinearModelOfCoregionalizationKernel(g.SpectralKernel(input_dims=D), output_dims=C, input_dims=D, Q=Q, Rq=Rq)
    if kind == "lmc_sm":
        return g.LinearModelOfCoregionalizationKernel(g.SpectralMixtureKernel(Q=3, input_dims=D), output_dims=C, input_dims=D, Q=Q, Rq=Rq)
    if kind == "conv":
        return g.MixtureKernel(g.GaussianConvolutionProcessKernel(output_dims=C, input_dims=D), Q)
    if kind == "mohsm":
        return g.MixtureKernel(g.MultiOutputHarmonizableSpectralKernel(output_dims=C, input_dims=D), Q)
    if kind == "umosm":
        return g.MixtureKernel(g.UncoupledMultiOutputSpectralKernel(output_dims=C, input_dims=D), Q)
    raise ValueError(kind)


def load_raw(params_iter, fxparams):
    """copy raw values + bounds from a fixture into product Parameters (same registration order)"""
    plist = list(params_iter)
    assert len(plist) == len(fxparams), (len(plist), len(fxparams))
    for p, f in zip(plist, fxparams):
        assert p.data.shape == f["raw"].shape, (p._name, f["name"], p.data.shape, f["raw"].shape)
        p.assign(f["cons"], lower=f["lower"], upper=f["upper"])
        p.data = np.array(f["raw"], dtype=p.data.dtype)    # exact raw values (assign() round-trips inexactly: quirk Q1); config.dtype
    return plist


def product_exact(fx, prefix=""):
    """gpr.Exact rebuilt from an lml/predict fixture"""
    from mogptk_amd import gpr as g
    C, Q, D, Rq = [int(v) for v in fx[prefix + "meta"][:4]]
    kind = str(fx[prefix + "kind"])
    fp = fixture_params(fx, prefix)
    k = product_kernel(kind, C, Q, D, Rq)
    scale = fp[-1]["cons"]
    m = g.Exact(k, fx[prefix + "X"], fx[prefix + "y"], variance=np.square(scale), jitter=float(fx[prefix + "jitter"]))
    load_raw(m.parameters(), fp)
    return m, fp


def relerr(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))


CFG3_FD_SEED, CFG3_FD_EPS = 3, 1e-4


def cfg3_direction(shapes):
    """the unit direction in raw-parameter space along which tests/golden/gen_cfg3.py takes the reference's central difference of its LML at
    BASELINE.json configs[2] (and tests/test_gpu_parity.py::test_cfg3_size_gradient_is_the_derivative_of_the_lml the device's)"""
    rng = np.random.default_rng(CFG3_FD_SEED)
    d = [rng.standard_normal(s) for s in shapes]
    nrm = np.sqrt(sum(float(np.sum(v * v)) for v in d))
    return [v / nrm for v in d]
