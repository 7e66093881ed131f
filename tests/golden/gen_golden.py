"""
Golden-vector generator.  Runs ONLY in the build container: imports the reference (GAMES-UChile/mogptk,
mounted read-only at /root/reference) with an IPython stub, evaluates its PyTorch-CPU path on seeded inputs
and writes small .npz fixtures next to this script.  The fixtures are data (inputs + expected outputs);
nothing of the reference travels.  Re-run:  python tests/golden/gen_golden.py [--full]

  kernels.npz   K(X), K(X,X2), K_diag for MOSM / SM-in-IMO / CSM-mixture (incl. shuffled row order, D=2)
  lml_*.npz     LML + d(loss)/d(raw) of every parameter (autograd) at N<=96, one at N=2048
  predict.npz   predict_f mean/var (+full covariance) and predict_y intervals
  adam_cfg1.npz airline-passengers SM(Q=3) Adam trajectory (BASELINE.json configs[0])
  kernels_8f2.npz / lml_mosk_* / lml_umosm_*  MultiOutputSpectralKernel and UncoupledMultiOutputSpectralKernel
  sm_lmc.npz     the SM_LMC wrapper: constructor state, loss + gradient, a short Adam trace
  init_ls.npz    Lomb-Scargle peak estimates and init_parameters('LS') of MOSM / SM / CSM / SM_LMC
  bnse.npz       BNSE spectra (init.py), BNSE peak estimates and MOSM.init_parameters('BNSE')
  transformers.npz Y transformers alone and chained; the raw airline series of configs[0]
  opt_traces.npz train('SGD' | 'AdaGrad') traces, the per-iteration error= path and a continued train() call on the cfg1 model
  peg.npz        loss + gradients of a model with pegged parameters (identity and 2x transforms)
  kernels_mohsm.npz / lml_mohsm_* / mohsm.npz  MultiOutputHarmonizableSpectralKernel and the MOHSM wrapper (SURVEY 8f-2 remainder)
  fp32.npz       config.use_single_precision(): float32 tensors, jitter floor 1e-6, loss / gradients / prediction of the reference's float32 run
  lbfgs_cfg1.npz the same model under train('LBFGS'): fixed-step and strong-Wolfe loss traces by function evaluation
  cfg2.npz      [--full] MOSM C=4 Q=3 N=8192 LML + gradient (BASELINE.json configs[1]; ~20 s, 10 GB)
  cfg4.npz      [--full] CSM C=4 Q=3 N=16384 predict_f at 64 probe rows of S=4096 (configs[3]; ~40 s, 16 GB)
  titsias.npz   small Titsias ELBO + gradients (kernel, scale, inducing points) + predict_f
  titsias_mohsm.npz  the same with an enveloped kernel (MOHSM): any kernel under any inference
  cfg5.npz      [--full] Titsias MOSM C=4 Q=3 N=100000 M=2048 ELBO + gradient (configs[4]; ~60 s, 37 GB)
                (reference self-consistency at cfg5, 8 vs 3 torch threads: kernel/noise gradients 4e-12..9e-12 relative,
                 inducing-point gradient 2.35e-3 relative / 2.5e-5 absolute -- that tensor is ill-conditioned)
"""
import os
import sys
import types
import argparse
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))

ip, disp = types.ModuleType("IPython"), types.ModuleType("IPython.display")
disp.display = lambda *a, **k: None
disp.HTML = lambda s: s
ip.display = disp
sys.modules["IPython"] = ip
sys.modules["IPython.display"] = disp
sys.path.insert(0, "/root/reference")
import torch          # noqa: E402
import mogptk         # noqa: E402

from mogptk_amd import synth   # noqa: E402  (seeded inputs shared with tests and bench)

g = mogptk.gpr
T = lambda a: torch.tensor(np.asarray(a), dtype=torch.float64)


def small_data(N, C, D, seed, shuffle=False):
    rng = np.random.default_rng(seed)
    n = [N // C + (1 if c < N % C else 0) for c in range(C)]
    X = np.concatenate([np.concatenate([np.full((n[c], 1), float(c)), rng.uniform(0, 10, (n[c], D))], axis=1)
                        for c in range(C)])
    y = np.sin(X[:, 1]) * (1 + 0.3 * X[:, 0]) + 0.1 * rng.standard_normal(N)
    if shuffle:
        p = rng.permutation(N)
        X, y = X[p], y[p]
    return X, y


def build_kernel(kind, C, Q, D, Rq, rng):
    if kind == "mosm":
        k = g.MultiOutputSpectralMixtureKernel(Q=Q, output_dims=C, input_dims=D)
        k.weight.assign(rng.uniform(0.5, 1.5, (C, Q)))
        k.mean.assign(rng.uniform(0.05, 0.5, (C, Q, D)))
        k.variance.assign(rng.uniform(0.05, 0.5, (C, Q, D)))
        k.delay.assign(rng.normal(0, 0.3, (C, Q, D)))
        k.phase.assign(rng.normal(0, 0.3, (C, Q)))
    elif kind == "sm":
        k = g.IndependentMultiOutputKernel([g.SpectralMixtureKernel(Q=Q, input_dims=D) for _ in range(C)], output_dims=C)
        for c in range(C):
            k[c].magnitude.assign(rng.uniform(0.5, 1.5, Q))
            k[c].mean.assign(rng.uniform(0.05, 0.5, (Q, D)))
            k[c].variance.assign(rng.uniform(0.01, 0.1, (Q, D)))
    elif kind == "csm":
        k = g.MixtureKernel(g.CrossSpectralKernel(output_dims=C, input_dims=D, Rq=Rq), Q)
        for q in range(Q):
            k[q].amplitude.assign(rng.uniform(0.5, 1.5, (C, Rq)))
            k[q].mean.assign(rng.uniform(0.05, 0.5, D))
            k[q].variance.assign(rng.uniform(0.05, 0.5, D))
            k[q].shift.assign(rng.normal(0, 0.3, (C, Rq)))
    elif kind == "mosk":       # MixtureKernel(MultiOutputSpectralKernel): MOSM components without the mixture axis
        k = g.MixtureKernel(g.MultiOutputSpectralKernel(output_dims=C, input_dims=D), Q)
        for q in range(Q):
            k[q].weight.assign(rng.uniform(0.5, 1.5, C))
            k[q].mean.assign(rng.uniform(0.05, 0.5, (C, D)))
            k[q].variance.assign(rng.uniform(0.05, 0.5, (C, D)))
            k[q].delay.assign(rng.normal(0, 0.3, (C, D)))
            k[q].phase.assign(rng.normal(0, 0.3, C))
    elif kind == "umosm":      # MixtureKernel(UncoupledMultiOutputSpectralKernel): lower-triangular (C,C) weight
        k = g.MixtureKernel(g.UncoupledMultiOutputSpectralKernel(output_dims=C, input_dims=D), Q)
        for q in range(Q):
            k[q].weight.assign(np.tril(rng.normal(0.8, 0.4, (C, C))))
            k[q].mean.assign(rng.uniform(0.05, 0.5, (C, D)))
            k[q].variance.assign(rng.uniform(0.05, 0.5, (C, D)))
            k[q].delay.assign(rng.normal(0, 0.3, (C, D)))
            k[q].phase.assign(rng.normal(0, 0.3, C))
    elif kind == "lmc":        # LMC over Q SpectralKernel base kernels, weight (C,Q,Rq)
        k = g.LinearModelOfCoregionalizationKernel(g.SpectralKernel(input_dims=D), output_dims=C, input_dims=D, Q=Q, Rq=Rq)
        k.weight.assign(rng.uniform(0.5, 1.5, (C, Q, Rq)))
        for q in range(Q):
            k[q].magnitude.assign(rng.uniform(0.5, 1.5))
            k[q].mean.assign(rng.uniform(0.05, 0.5, D))
            k[q].variance.assign(rng.uniform(0.01, 0.1, D))
    elif kind == "lmc_sm":     # LMC over spectral-mixture base kernels (3 components each)
        k = g.LinearModelOfCoregionalizationKernel(g.SpectralMixtureKernel(Q=3, input_dims=D), output_dims=C, input_dims=D, Q=Q, Rq=Rq)
        k.weight.assign(rng.uniform(0.5, 1.5, (C, Q, Rq)))
        for q in range(Q):
            k[q].magnitude.assign(rng.uniform(0.5, 1.5, 3))
            k[q].mean.assign(rng.uniform(0.05, 0.5, (3, D)))
            k[q].variance.assign(rng.uniform(0.01, 0.1, (3, D)))
    elif kind == "mohsm":      # MixtureKernel(MultiOutputHarmonizableSpectralKernel): Gaussian envelope on the input midpoint
        k = g.MixtureKernel(g.MultiOutputHarmonizableSpectralKernel(output_dims=C, input_dims=D), Q)
        for q in range(Q):
            k[q].weight.assign(rng.uniform(0.5, 1.5, C))
            k[q].mean.assign(rng.uniform(0.05, 0.5, (C, D)))
            k[q].variance.assign(rng.uniform(0.05, 0.5, (C, D)))
            k[q].lengthscale.assign(rng.uniform(0.1, 0.4, C))
            k[q].center.assign(rng.uniform(2.0, 8.0, D))
            k[q].delay.assign(rng.normal(0, 0.3, (C, D)))
            k[q].phase.assign(rng.normal(0, 0.3, C))
    elif kind == "conv":       # MixtureKernel(GaussianConvolutionProcessKernel)
        k = g.MixtureKernel(g.GaussianConvolutionProcessKernel(output_dims=C, input_dims=D), Q)
        for q in range(Q):
            k[q].weight.assign(rng.uniform(0.5, 1.5, C))
            k[q].variance.assign(rng.uniform(0.05, 1.0, (C, D)))
            k[q].base_variance.assign(rng.uniform(0.1, 1.0, D))
    return k


def dump_params(prefix, params, out, with_grad=False):
    for n, p in enumerate(params):
        out["%sp%d_raw" % (prefix, n)] = p.data.detach().numpy().copy()
        out["%sp%d_lower" % (prefix, n)] = np.array(np.nan) if p.lower is None else np.asarray(p.lower.detach().numpy() if torch.is_tensor(p.lower) else p.lower)
        out["%sp%d_upper" % (prefix, n)] = np.array(np.nan) if p.upper is None else np.asarray(p.upper.detach().numpy() if torch.is_tensor(p.upper) else p.upper)
        out["%sp%d_cons" % (prefix, n)] = p().detach().numpy().copy()
        if with_grad:
            out["%sp%d_grad" % (prefix, n)] = np.array(np.nan) if p.grad is None else p.grad.detach().numpy().copy()
    out[prefix + "names"] = np.array([p._name for p in params])


KERNEL_CASES = [  # kind, C, Q, D, Rq, N, N2, shuffle
    ("mosm", 2, 1, 1, 1, 40, 17, False),
    ("mosm", 3, 2, 1, 1, 61, 23, True),
    ("mosm", 3, 3, 2, 1, 48, 19, False),
    ("mosm", 1, 2, 1, 1, 30, 11, False),
    ("sm", 1, 3, 1, 1, 33, 12, False),
    ("sm", 2, 2, 2, 1, 40, 15, True),
    ("csm", 3, 2, 1, 1, 45, 16, False),
    ("csm", 2, 2, 1, 2, 38, 14, True),
    ("csm", 3, 1, 2, 2, 36, 13, False),
]


KERNEL_CASES_8F2 = [  # the multi-output kernels that share MOSM's term table
    ("mosk", 3, 2, 1, 1, 45, 16, False),
    ("mosk", 2, 1, 2, 1, 36, 13, True),
    ("umosm", 3, 2, 1, 1, 44, 15, False),
    ("umosm", 2, 2, 2, 1, 38, 12, True),
    ("umosm", 1, 1, 1, 1, 25, 9, False),
    ("lmc", 3, 2, 1, 2, 42, 14, False),
    ("lmc", 2, 3, 2, 1, 36, 12, True),
    ("lmc_sm", 2, 2, 1, 1, 30, 11, False),
    ("conv", 3, 2, 1, 1, 40, 13, False),
    ("conv", 2, 1, 2, 1, 34, 12, True),
]


def gen_kernels_8f2():
    gen_kernels(KERNEL_CASES_8F2, "kernels_8f2.npz", 1100)


KERNEL_CASES_MOHSM = [
    ("mohsm", 3, 2, 1, 1, 45, 16, False),
    ("mohsm", 2, 1, 2, 1, 36, 13, True),
    ("mohsm", 1, 2, 1, 1, 25, 9, False),
]
LML_CASES_MOHSM = [
    ("mohsm_c3q2", "mohsm", 3, 2, 1, 1, 84, False, False),
    ("mohsm_c2q1_d2", "mohsm", 2, 1, 2, 1, 60, True, False),
    ("mohsm_c1q2", "mohsm", 1, 2, 1, 1, 50, False, False),
]


def gen_mohsm():
    """SURVEY 8f-2 remainder: MultiOutputHarmonizableSpectralKernel (kernels, LML + autograd gradients, prediction) and the MOHSM wrapper
    (constructor state, init_parameters('LS'), a short Adam run)"""
    gen_kernels(KERNEL_CASES_MOHSM, "kernels_mohsm.npz", 1300)
    gen_lml(LML_CASES_MOHSM, 4300, synth_case=False)
    out = {}
    rng = np.random.default_rng(6300)
    C, Q, D, N, S = 2, 2, 1, 70, 27
    X, y = small_data(N, C, D, 7300)
    Xs, _ = small_data(S, C, D, 8300)
    Xs[:, 1:] = Xs[:, 1:] * 1.2
    m, lml, loss = lml_case("mohsm", C, Q, D, 1, X, y, rng)
    out["meta"] = np.array([C, Q, D, 1]); out["kind"] = np.array("mohsm")
    out["X"] = X; out["y"] = y; out["Xs"] = Xs; out["jitter"] = np.array(m.jitter)
    dump_params("", list(m.parameters()), out)
    mu, var = m.predict_f(T(Xs))
    out["mu"] = mu.numpy(); out["var"] = var.numpy()
    # the wrapper
    t = np.linspace(0, 10, 50)
    ds = mogptk.DataSet(t, [np.sin(0.5 * t), 2.0 * np.sin(0.2 * t) + 0.1 * np.cos(3 * t)])
    torch.manual_seed(5)
    mm = mogptk.MOHSM(ds, P=1, Q=2)
    out["w_t"] = t; out["w_Y"] = np.stack([ds[j].Y for j in range(2)])
    dump_params("w_ctor_", list(mm.gpr.parameters()), out)
    out["w_num_parameters"] = np.array(mm.num_parameters())
    mm.init_parameters("LS")
    dump_params("w_init_", list(mm.gpr.parameters()), out)
    out["w_lml"] = np.array(mm.log_marginal_likelihood())
    mm.gpr.zero_grad(); l0 = mm.gpr.loss()
    out["w_loss"] = np.array(float(l0))
    dump_params("w_", list(mm.gpr.parameters()), out, with_grad=True)
    losses, _ = mm.train("Adam", iters=8, lr=0.05, jit=False)
    out["w_adam_losses"] = np.array(losses)
    np.savez_compressed(os.path.join(HERE, "mohsm.npz"), **out)
    print("mohsm.npz lml=%.10f wrapper lml=%.10f params=%d" % (lml, out["w_lml"], out["w_num_parameters"]))


def gen_kernels(cases=None, fname="kernels.npz", seed0=1000):
    cases = KERNEL_CASES if cases is None else cases
    out = {"ncases": np.array(len(cases))}
    for n, (kind, C, Q, D, Rq, N, N2, shuffle) in enumerate(cases):
        rng = np.random.default_rng(seed0 + n)
        X, _ = small_data(N, C, D, 2000 + n, shuffle)
        X2, _ = small_data(N2, C, D, 3000 + n, shuffle)
        k = build_kernel(kind, C, Q, D, Rq, rng)
        pre = "c%d_" % n
        out[pre + "meta"] = np.array([C, Q, D, Rq])
        out[pre + "kind"] = np.array(kind)
        out[pre + "X"] = X
        out[pre + "X2"] = X2
        dump_params(pre, list(k.parameters()), out)
        with torch.no_grad():
            out[pre + "K"] = k.K(T(X)).numpy()
            out[pre + "K12"] = k.K(T(X), T(X2)).numpy()
            out[pre + "Kdiag"] = k.K_diag(T(X)).numpy()
    np.savez_compressed(os.path.join(HERE, fname), **out)
    print(fname, "written")


LML_CASES = [  # name, kind, C, Q, D, Rq, N, shuffle, scalar_variance
    ("mosm_c3q2", "mosm", 3, 2, 1, 1, 96, False, False),
    ("mosm_c2q3_shuf", "mosm", 2, 3, 1, 1, 70, True, False),
    ("mosm_c3q2_d2", "mosm", 3, 2, 2, 1, 60, False, False),
    ("mosm_c1q2", "mosm", 1, 2, 1, 1, 50, False, False),
    ("mosm_scalarvar", "mosm", 2, 2, 1, 1, 64, False, True),
    ("sm_c1q3", "sm", 1, 3, 1, 1, 80, False, False),
    ("sm_c2q2_d2", "sm", 2, 2, 2, 1, 72, False, False),
    ("csm_c3q2", "csm", 3, 2, 1, 1, 90, False, False),
    ("csm_c2q2r2", "csm", 2, 2, 1, 2, 66, True, False),
]


def lml_case(kind, C, Q, D, Rq, X, y, rng, scalar_variance=False, jitter=1e-8):
    k = build_kernel(kind, C, Q, D, Rq, rng)
    scale = rng.uniform(0.1, 0.4) if scalar_variance else rng.uniform(0.1, 0.4, C)
    m = g.Exact(k, T(X), T(y), variance=(scale ** 2 if scalar_variance else list(scale ** 2)), jitter=jitter)
    m.likelihood.scale.assign(scale)
    lml = float(m.log_marginal_likelihood())
    loss = float(m.loss())
    return m, lml, loss


LML_CASES_8F2 = [
    ("mosk_c3q2", "mosk", 3, 2, 1, 1, 84, False, False),
    ("mosk_c2q1_d2", "mosk", 2, 1, 2, 1, 60, True, False),
    ("umosm_c3q2", "umosm", 3, 2, 1, 1, 90, False, False),
    ("umosm_c2q2_d2", "umosm", 2, 2, 2, 1, 64, True, False),
    ("lmc_c3q2r2", "lmc", 3, 2, 1, 2, 84, False, False),
    ("lmc_c2q3_d2", "lmc", 2, 3, 2, 1, 60, True, False),
    ("lmcsm_c2q2", "lmc_sm", 2, 2, 1, 1, 66, False, False),
    ("conv_c3q2", "conv", 3, 2, 1, 1, 78, False, False),
    ("conv_c2q1_d2", "conv", 2, 1, 2, 1, 60, True, False),
]


def gen_lml_8f2():
    gen_lml(LML_CASES_8F2, 4100, synth_case=False)


def gen_lml(cases=None, seed0=4000, synth_case=True):
    cases = LML_CASES if cases is None else cases
    for n, (name, kind, C, Q, D, Rq, N, shuffle, sv) in enumerate(cases):
        rng = np.random.default_rng(seed0 + n)
        X, y = small_data(N, C, D, seed0 + 1000 + n, shuffle)
        m, lml, loss = lml_case(kind, C, Q, D, Rq, X, y, rng, sv)
        out = {"meta": np.array([C, Q, D, Rq]), "kind": np.array(kind), "X": X, "y": y, "jitter": np.array(m.jitter),
               "lml": np.array(lml), "loss": np.array(loss), "scalar_variance": np.array(sv)}
        dump_params("", list(m.parameters()), out, with_grad=True)
        np.savez_compressed(os.path.join(HERE, "lml_%s.npz" % name), **out)
        print("lml_%s.npz  lml=%.10f" % (name, lml))

    if not synth_case:
        return
    # one mid-size case on the shared synthetic generator: only outputs are stored
    C, Q, N = 4, 3, 2048
    X, y = synth.make_data(N, C)
    h = synth.mosm_hypers(C, Q)
    m = ref_mosm(X, y, h, C, Q)
    lml = float(m.log_marginal_likelihood())
    loss = float(m.loss())
    out = {"meta": np.array([C, Q, 1, 1, N]), "lml": np.array(lml), "loss": np.array(loss)}
    dump_params("", list(m.parameters()), out, with_grad=True)
    np.savez_compressed(os.path.join(HERE, "lml_synth2048.npz"), **out)
    print("lml_synth2048.npz lml=%.10f" % lml)


def ref_mosm(X, y, h, C, Q, D=1, jitter=1e-8):
    k = g.MultiOutputSpectralMixtureKernel(Q=Q, output_dims=C, input_dims=D)
    k.weight.assign(h["weight"]); k.mean.assign(h["mean"]); k.variance.assign(h["variance"])
    k.delay.assign(h["delay"]); k.phase.assign(h["phase"])
    m = g.Exact(k, T(X), T(y), variance=list(h["scale"] ** 2), jitter=jitter)
    m.likelihood.scale.assign(h["scale"])
    return m


def ref_csm(X, y, h, C, Q, Rq=1, D=1, jitter=1e-8):
    k = g.MixtureKernel(g.CrossSpectralKernel(output_dims=C, input_dims=D, Rq=Rq), Q)
    for q in range(Q):
        k[q].amplitude.assign(h["amplitude"][q]); k[q].mean.assign(h["mean"][q])
        k[q].variance.assign(h["variance"][q]); k[q].shift.assign(h["shift"][q])
    m = g.Exact(k, T(X), T(y), variance=list(h["scale"] ** 2), jitter=jitter)
    m.likelihood.scale.assign(h["scale"])
    return m


def gen_predict():
    out = {}
    cases = [("mosm", 3, 2, 1, 1, 80, 31, False), ("csm", 2, 2, 1, 1, 60, 25, True), ("sm", 1, 3, 1, 1, 70, 29, False)]
    out["ncases"] = np.array(len(cases))
    for n, (kind, C, Q, D, Rq, N, S, shuffle) in enumerate(cases):
        rng = np.random.default_rng(6000 + n)
        X, y = small_data(N, C, D, 7000 + n, shuffle)
        Xs, _ = small_data(S, C, D, 8000 + n, shuffle)
        Xs[:, 1:] = Xs[:, 1:] * 1.2
        m, lml, loss = lml_case(kind, C, Q, D, Rq, X, y, rng)
        pre = "c%d_" % n
        out[pre + "meta"] = np.array([C, Q, D, Rq]); out[pre + "kind"] = np.array(kind)
        out[pre + "X"] = X; out[pre + "y"] = y; out[pre + "Xs"] = Xs; out[pre + "jitter"] = np.array(m.jitter)
        dump_params(pre, list(m.parameters()), out)
        mu, var = m.predict_f(T(Xs))
        mu2, cov = m.predict_f(T(Xs), full=True)
        ymu, lo, up = m.predict_y(T(Xs), sigma=2.0)
        out[pre + "mu"] = mu.numpy(); out[pre + "var"] = var.numpy(); out[pre + "cov"] = cov.numpy()
        out[pre + "lower"] = lo.numpy(); out[pre + "upper"] = up.numpy()
    # single-output (no channel column) Exact with a scalar variance: the other CI branch (likelihood.py:370-378)
    rng = np.random.default_rng(6100)
    x = np.sort(rng.uniform(0, 10, 50)).reshape(-1, 1)
    y = np.sin(x[:, 0]) + 0.1 * rng.standard_normal(50)
    xs = np.linspace(-1, 12, 21).reshape(-1, 1)
    k = g.SpectralMixtureKernel(Q=2, input_dims=1)
    k.magnitude.assign([1.0, 0.5]); k.mean.assign([[0.15], [0.3]]); k.variance.assign([[0.02], [0.05]])
    m = g.Exact(k, T(x), T(y), variance=0.04)
    out["so_X"] = x; out["so_y"] = y; out["so_Xs"] = xs
    dump_params("so_", list(m.parameters()), out)
    mu, var = m.predict_f(T(xs))
    ymu, lo, up = m.predict_y(T(xs), sigma=2.0)
    out["so_mu"] = mu.numpy(); out["so_var"] = var.numpy(); out["so_lower"] = lo.numpy(); out["so_upper"] = up.numpy()
    out["so_lml"] = np.array(float(m.log_marginal_likelihood()))
    np.savez_compressed(os.path.join(HERE, "predict.npz"), **out)
    print("predict.npz written")


def gen_adam_cfg1():
    """BASELINE.json configs[0]: SM Q=3 on airline passengers, TransformDetrend(2)+TransformStandard,
    init_parameters('LS'), train('Adam', iters=100, lr=0.1).  Stores kernel-format inputs (after the
    reference's transforms), the initial raw parameters, the loss trace and the final raw parameters."""
    air = np.loadtxt("/root/reference/examples/data/Airline_passenger.csv")
    data = mogptk.Data(air[:, 0], air[:, 1], name="airline")
    data.transform(mogptk.TransformDetrend(degree=2))
    data.transform(mogptk.TransformStandard())
    torch.manual_seed(1)
    model = mogptk.SM(data, Q=3)
    model.init_parameters("LS")
    out = {"X": model.gpr.X.numpy().copy(), "y": model.gpr.y.numpy().copy(), "jitter": np.array(model.gpr.jitter),
           "lr": np.array(0.1), "iters": np.array(100)}
    dump_params("init_", list(model.gpr.parameters()), out)
    out["lml0"] = np.array(model.log_marginal_likelihood())
    losses, _ = model.train("Adam", iters=100, lr=0.1, jit=False)
    out["losses"] = np.array(losses)
    dump_params("final_", list(model.gpr.parameters()), out)
    xs = np.linspace(0, 160, 33)
    _, mu, lo, up = model.predict(xs, transformed=True)
    out["pred_X"] = xs; out["pred_mu"] = mu; out["pred_lower"] = lo; out["pred_upper"] = up
    np.savez_compressed(os.path.join(HERE, "adam_cfg1.npz"), **out)
    print("adam_cfg1.npz lml0=%.10f loss[100]=%.6f" % (out["lml0"], losses[-1]))


def gen_lbfgs_cfg1():
    """the reference's other documented optimiser on the cfg1 model: train('LBFGS') with torch's defaults (fixed step) and with lr / history_size
    overrides.  Stores the loss traces (indexed by function evaluation, model.py:546-552), model.iters and the
    final raw parameters; inputs and initial parameters are those of adam_cfg1.npz."""
    air = np.loadtxt("/root/reference/examples/data/Airline_passenger.csv")
    out = {}
    # (line_search_fn='strong_wolfe' cannot be recorded: with the installed torch the reference dies in LBFGS._clone_param, because its
    #  Parameter.clone() takes no memory_format -- the line search itself is pinned on torch.optim.LBFGS in tests/test_host_logic.py)
    for tag, kw in (("fixed", dict(iters=40)), ("fixed_lr", dict(iters=25, lr=0.5, history_size=5))):
        data = mogptk.Data(air[:, 0], air[:, 1], name="airline")
        data.transform(mogptk.TransformDetrend(degree=2))
        data.transform(mogptk.TransformStandard())
        torch.manual_seed(1)
        model = mogptk.SM(data, Q=3)
        model.init_parameters("LS")
        losses, _ = model.train("LBFGS", jit=False, **kw)
        out[tag + "_losses"] = np.array(model.losses)
        out[tag + "_iters"] = np.array(model.iters)
        out[tag + "_max_iter"] = np.array(kw["iters"])
        dump_params(tag + "_final_", list(model.gpr.parameters()), out)
        print("lbfgs %s: func evals %d, loss %.8f -> %.8f" % (tag, model.iters, model.losses[0], model.losses[-1]))
    np.savez_compressed(os.path.join(HERE, "lbfgs_cfg1.npz"), **out)



def _cfg1_model(remove=0):
    air = np.loadtxt("/root/reference/examples/data/Airline_passenger.csv")
    data = mogptk.Data(air[:, 0], air[:, 1], name="airline")
    if remove:
        data.remove_range(start=air[-remove, 0] - 1e-9)        # the last `remove` points become test points
    data.transform(mogptk.TransformDetrend(degree=2))
    data.transform(mogptk.TransformStandard())
    torch.manual_seed(1)
    model = mogptk.SM(data, Q=3)
    model.init_parameters("LS")
    return model


def gen_opt_traces():
    """SURVEY 8f-1 remainder: train('SGD') (plain and with momentum / nesterov / weight decay), train('AdaGrad') and the per-iteration
    `error=` path (model.py:531-532: a full predict per iteration), on the cfg1 model; a continued second train() call
    (iter_offset, model.py:501-509).  Inputs / initial parameters are those of adam_cfg1.npz."""
    out = {}
    runs = (("sgd", "SGD", dict(iters=12, lr=2e-4)),
            ("sgd_mom", "sgd", dict(iters=12, lr=1e-4, momentum=0.9, nesterov=True, weight_decay=1e-3)),
            ("adagrad", "AdaGrad", dict(iters=12, lr=0.05)),
            ("adagrad_decay", "adagrad", dict(iters=12, lr=0.05, lr_decay=0.1, initial_accumulator_value=0.5)))
    for tag, method, kw in runs:
        model = _cfg1_model()
        losses, _ = model.train(method, jit=False, **kw)
        out[tag + "_losses"] = np.array(model.losses)
        dump_params(tag + "_final_", list(model.gpr.parameters()), out)
        print("%s: loss %.8f -> %.8f" % (tag, model.losses[0], model.losses[-1]))
    # error= with held-out test points, then a continued call
    model = _cfg1_model(remove=24)
    out["err_X"] = model.gpr.X.numpy().copy(); out["err_y"] = model.gpr.y.numpy().copy()
    dump_params("err_init_", list(model.gpr.parameters()), out)
    losses, errors = model.train("Adam", iters=8, lr=0.05, error="MAE", jit=False)
    out["err_losses"] = np.array(model.losses); out["err_errors"] = np.array(model.errors)
    losses, errors = model.train("Adam", iters=5, lr=0.05, error="sMAPE", jit=False)
    out["err_losses2"] = np.array(model.losses); out["err_errors2"] = np.array(model.errors); out["err_iters2"] = np.array(model.iters)
    out["err_rmse_all"] = np.array(model.error("RMSE", use_all_data=True))
    # error= without test data (falls back to all data), custom callable
    model = _cfg1_model()
    losses, errors = model.train("Adam", iters=4, lr=0.05, error=lambda yt, yp: float(np.max(np.abs(yt - yp))), jit=False)
    out["errall_errors"] = np.array(model.errors)
    np.savez_compressed(os.path.join(HERE, "opt_traces.npz"), **out)
    print("opt_traces.npz written")


def gen_peg():
    """Parameter.peg (parameter.py:321-335): autograd sends a pegged parameter's gradient to the parameter it follows, through the
    peg transform; the pegged tensor itself keeps grad None.  MOSM with channel 1's noise pegged to channel 0's... is not expressible
    (one tensor); so: SM-in-IMO, kernel[1].mean pegged to 2 x kernel[0].mean, kernel[1].magnitude pegged to kernel[0].magnitude."""
    rng = np.random.default_rng(909)
    C, Q, D, N = 2, 2, 1, 64
    X, y = small_data(N, C, D, 910)
    k = build_kernel("sm", C, Q, D, 1, rng)
    k[1].mean.peg(k[0].mean, lambda x: 2.0 * x)
    k[1].magnitude.peg(k[0].magnitude)
    scale = rng.uniform(0.1, 0.4, C)
    m = g.Exact(k, T(X), T(y), variance=list(scale ** 2))
    m.likelihood.scale.assign(scale)
    out = {"meta": np.array([C, Q, D, 1]), "X": X, "y": y, "jitter": np.array(m.jitter)}
    out["lml"] = np.array(float(m.log_marginal_likelihood()))
    out["loss"] = np.array(float(m.loss()))
    dump_params("", list(m.parameters()), out, with_grad=True)
    np.savez_compressed(os.path.join(HERE, "peg.npz"), **out)
    print("peg.npz lml=%.10f" % out["lml"], [None if p.grad is None else p.grad.shape for p in m.parameters()])

def gen_fp32():
    """config.use_single_precision() (gpr/config.py:20-24): float32 host tensors and the 1e-6 jitter floor (gpr/model.py:106-110); the
    reference then ALSO factorises in float32, so its loss / gradients carry float32 rounding (recorded next to a float64 run of the same
    float32-rounded inputs for scale)"""
    out = {}
    rng = np.random.default_rng(7700)
    C, Q, D, N = 2, 2, 1, 64
    X, y = small_data(N, C, D, 7701)
    g.use_single_precision()
    try:
        T32 = lambda a: torch.tensor(np.asarray(a), dtype=torch.float32)
        k = build_kernel("mosm", C, Q, D, 1, rng)
        scale = rng.uniform(0.2, 0.4, C)
        m = g.Exact(k, T32(X), T32(y), variance=list(scale ** 2), jitter=1e-8)
        m.likelihood.scale.assign(scale)
        out["jitter"] = np.array(m.jitter)
        out["loss"] = np.array(float(m.loss()))
        out["loss_dtype"] = np.array(str(m.loss().dtype))
        dump_params("", list(m.parameters()), out, with_grad=True)
        Xs, _ = small_data(21, C, D, 7702)
        mu, var = m.predict_f(T32(Xs))
        out["Xs"] = Xs; out["mu"] = mu.numpy(); out["var"] = var.numpy()
    finally:
        g.use_double_precision()
    out["meta"] = np.array([C, Q, D, 1]); out["X"] = X; out["y"] = y
    np.savez_compressed(os.path.join(HERE, "fp32.npz"), **out)
    print("fp32.npz loss=%.6f jitter=%g dtype=%s" % (out["loss"], out["jitter"], out["loss_dtype"]))


def gen_quirks():
    """Q1/Q2 of SURVEY.md 8b as data."""
    out = {}
    p = g.Parameter(1.0, lower=1e-8)
    out["q1_readback"] = p().detach().numpy()
    out["q1_raw"] = p.data.detach().numpy()
    t = np.linspace(0, 10, 20)
    ds = mogptk.DataSet(t, [np.sin(t), np.cos(t)])
    torch.manual_seed(0)
    m = mogptk.MOSM(ds, Q=2)
    out["q2_mean"] = m.gpr.kernel.mean().detach().numpy()
    out["q2_mean_raw"] = m.gpr.kernel.mean.data.detach().numpy()
    out["q2_upper"] = m.gpr.kernel.mean.upper.detach().numpy()
    vals = np.array([0.3, 1e-3, 5.0, 250.0])
    p = g.Parameter(vals, lower=1e-8)
    out["sp_vals"] = vals; out["sp_raw"] = p.data.detach().numpy(); out["sp_cons"] = p().detach().numpy()
    p = g.Parameter(vals, lower=1e-8, upper=300.0)
    out["sg_raw"] = p.data.detach().numpy(); out["sg_cons"] = p().detach().numpy()
    np.savez_compressed(os.path.join(HERE, "quirks.npz"), **out)
    print("quirks.npz written")


def gen_smlmc():
    """the SM_LMC wrapper (models/sm_lmc.py): constructor state (bounds, train flags, Nyquist re-bounding incl. quirk Q2), then explicit
    values for everything it draws at random, the loss and its gradient, and a short Adam run"""
    t = np.linspace(0, 10, 50)
    ds = mogptk.DataSet(t, [np.sin(0.5 * t), 2.0 * np.sin(0.2 * t) + 0.1 * np.cos(3 * t), np.cos(0.7 * t)])
    torch.manual_seed(3)
    m = mogptk.SM_LMC(ds, Q=2, Rq=2)
    out = {"t": t, "Y": np.stack([ds[j].Y for j in range(3)]), "Q": np.array(2), "Rq": np.array(2)}
    dump_params("ctor_", list(m.gpr.parameters()), out)
    out["ctor_train"] = np.array([bool(p.train) for p in m.gpr.parameters()])
    out["num_parameters"] = np.array(m.num_parameters())
    rng = np.random.default_rng(77)
    m.gpr.kernel.weight.assign(rng.uniform(0.3, 1.2, (3, 2, 2)))
    for q in range(2):
        m.gpr.kernel[q].mean.assign(rng.uniform(0.05, 0.4, 1))
        m.gpr.kernel[q].variance.assign(rng.uniform(0.01, 0.1, 1))
    m.gpr.likelihood.scale.assign(rng.uniform(0.1, 0.3, 3))
    out["lml"] = np.array(m.log_marginal_likelihood())
    m.gpr.zero_grad(); loss = m.gpr.loss()
    out["loss"] = np.array(float(loss))
    dump_params("", list(m.gpr.parameters()), out, with_grad=True)
    losses, _ = m.train("Adam", iters=10, lr=0.05, jit=False)
    out["adam_losses"] = np.array(losses)
    np.savez_compressed(os.path.join(HERE, "sm_lmc.npz"), **out)
    print("sm_lmc.npz lml=%.10f params=%d" % (out["lml"], out["num_parameters"]))


def gen_init_ls():
    """init_parameters('LS') of the four model wrappers: the Lomb-Scargle peak estimates of a 3-channel, irregularly
    sampled data set and every kernel / noise parameter afterwards"""
    rng = np.random.default_rng(11)
    chans = []
    for j, (f1, f2) in enumerate(((0.11, 0.31), (0.07, 0.23), (0.19, 0.41))):
        x = np.sort(rng.uniform(0.0, 60.0, 90 + 10 * j))
        y = np.sin(2 * np.pi * f1 * x) + 0.5 * np.cos(2 * np.pi * f2 * x + 0.3 * j) + 0.05 * rng.standard_normal(x.size)
        chans.append((x, y))
    out = {"nchan": np.array(3)}
    for j, (x, y) in enumerate(chans):
        out["x%d" % j] = x; out["y%d" % j] = y
    ds = mogptk.DataSet(*[mogptk.Data(x, y) for x, y in chans])
    A, B, C = ds.get_ls_estimation(Q=3)
    out["ls_A"] = np.stack(A); out["ls_B"] = np.stack(B); out["ls_C"] = np.stack(C)
    out["nyquist"] = np.stack(ds.get_nyquist_estimation())
    for tag, make in (("mosm", lambda: mogptk.MOSM(ds, Q=2)), ("sm", lambda: mogptk.SM(ds, Q=3)),
                      ("csm", lambda: mogptk.CSM(ds, Q=2, Rq=2)), ("smlmc", lambda: mogptk.SM_LMC(ds, Q=2, Rq=2))):
        torch.manual_seed(5)
        m = make()
        m.init_parameters("LS")
        dump_params(tag + "_", list(m.gpr.parameters()), out)
        out[tag + "_lml"] = np.array(m.log_marginal_likelihood())
    # (two input dimensions cannot be recorded: the reference re-uses `n` for the number of peaks found, so the second dimension gets a
    #  one-point frequency grid and scipy's find_peaks raises -- the restatement keeps that behaviour, tests/test_host_logic.py)
    np.savez_compressed(os.path.join(HERE, "init_ls.npz"), **out)
    print("init_ls.npz written; mosm lml %.8f" % out["mosm_lml"])


def gen_bnse():
    """BNSE (init.py): the spectrum of a two-tone signal after a 60-step GP fit, with and without observation errors; the peak
    estimates of a 2-channel data set and MOSM.init_parameters('BNSE') on it"""
    rng = np.random.default_rng(21)
    x = np.sort(rng.uniform(0.0, 40.0, 110))
    y = np.sin(2 * np.pi * 0.15 * x) + 0.6 * np.cos(2 * np.pi * 0.37 * x) + 0.05 * rng.standard_normal(x.size)
    out = {"x": x.copy(), "y": y.copy()}
    w, mu, var = mogptk.BNSE(x.copy(), y, n=150, iters=60, jit=False)
    out["w"] = w; out["mu"] = mu; out["var"] = var
    yerr = rng.uniform(0.02, 0.1, x.size)
    w2, mu2, var2 = mogptk.BNSE(x.copy(), y, y_err=yerr, max_freq=0.9, n=120, iters=40, jit=False)
    out["yerr"] = yerr; out["w2"] = w2; out["mu2"] = mu2; out["var2"] = var2
    x1 = np.sort(rng.uniform(0.0, 50.0, 90))
    y1 = 1.5 * np.sin(2 * np.pi * 0.09 * x1 + 0.4) + 0.05 * rng.standard_normal(x1.size)
    ds = mogptk.DataSet(mogptk.Data(x, y), mogptk.Data(x1, y1))
    A, B, C = ds.get_bnse_estimation(Q=2, n=400, iters=50)
    out["x1"] = x1; out["y1"] = y1; out["est_A"] = np.stack(A); out["est_B"] = np.stack(B); out["est_C"] = np.stack(C)
    torch.manual_seed(2)
    m = mogptk.MOSM(ds, Q=2)
    m.init_parameters("BNSE", iters=50)
    dump_params("mosm_", list(m.gpr.parameters()), out)
    np.savez_compressed(os.path.join(HERE, "bnse.npz"), **out)
    print("bnse.npz written; PSD peak at %.4f" % w[np.argmax(mu)])


def gen_transformers():
    """Y transformers (transformer.py) one by one and chained, on seeded data and on the airline-passenger series of configs[0]
    (raw series stored: 144 points; its transformed version is what adam_cfg1.npz holds)"""
    rng = np.random.default_rng(9)
    x = np.sort(rng.uniform(0, 20, 60)).reshape(-1, 1)
    y = 5.0 + 0.8 * x[:, 0] + 0.05 * x[:, 0] ** 2 + np.sin(x[:, 0]) + 0.1 * rng.standard_normal(60)
    out = {"x": x, "y": y}
    cases = {"detrend2": [mogptk.TransformDetrend(degree=2)], "linear": [mogptk.TransformLinear(bias=1.5, slope=0.7)],
             "normalize": [mogptk.TransformNormalize], "log": [mogptk.TransformLog], "standard": [mogptk.TransformStandard],
             "chain": [mogptk.TransformDetrend(degree=1), mogptk.TransformLog, mogptk.TransformStandard()]}
    for name, ts in cases.items():
        d = mogptk.Data(x[:, 0].copy(), y.copy())
        for t in ts:
            d.transform(t)
        _, yt = d.get_data(transformed=True)
        out[name + "_fwd"] = yt
        out[name + "_bwd"] = d.Y_transformer.backward(yt + 0.25, d.X)
    air = np.loadtxt("/root/reference/examples/data/Airline_passenger.csv")
    out["air_x"] = air[:, 0]; out["air_y"] = air[:, 1]
    d = mogptk.Data(air[:, 0], air[:, 1], name="airline")
    d.transform(mogptk.TransformDetrend(degree=2)); d.transform(mogptk.TransformStandard())
    out["air_yt"] = d.get_data(transformed=True)[1]
    np.savez_compressed(os.path.join(HERE, "transformers.npz"), **out)
    print("transformers.npz written")


def gen_cfg2():
    import time
    C, Q, N = 4, 3, 8192
    X, y = synth.make_data(N, C)
    m = ref_mosm(X, y, synth.mosm_hypers(C, Q), C, Q)
    t = time.time()
    loss = float(m.loss())
    dt = time.time() - t
    out = {"meta": np.array([C, Q, 1, 1, N]), "loss": np.array(loss), "lml": np.array(-loss), "seconds": np.array(dt),
           "threads": np.array(torch.get_num_threads())}
    dump_params("", list(m.parameters()), out, with_grad=True)
    np.savez_compressed(os.path.join(HERE, "cfg2.npz"), **out)
    print("cfg2.npz loss=%.10f  (%.1f s/eval on %d threads)" % (loss, dt, torch.get_num_threads()))


def gen_cfg4():
    import time
    C, Q, N, S = 4, 3, 16384, 4096
    X, y = synth.make_data(N, C)
    Xs = synth.test_inputs(S, C)
    m = ref_csm(X, y, synth.csm_hypers(C, Q), C, Q)
    t = time.time()
    mu, var = m.predict_f(T(Xs))
    dt = time.time() - t
    probe = np.arange(0, S, S // 64)
    out = {"meta": np.array([C, Q, 1, 1, N, S]), "probe": probe, "mu": mu.numpy()[probe, 0], "var": var.numpy()[probe, 0],
           "seconds": np.array(dt)}
    np.savez_compressed(os.path.join(HERE, "cfg4.npz"), **out)
    print("cfg4.npz written (%.1f s)" % dt)



def ref_titsias_mosm(X, y, h, C, Q, Z, Z_init="grid", jitter=1e-8, scale=None):
    k = g.MultiOutputSpectralMixtureKernel(Q=Q, output_dims=C, input_dims=X.shape[1] - 1)
    k.weight.assign(h["weight"]); k.mean.assign(h["mean"]); k.variance.assign(h["variance"])
    k.delay.assign(h["delay"]); k.phase.assign(h["phase"])
    s = float(np.mean(h["scale"])) if scale is None else scale
    m = g.Titsias(k, T(X), T(y), Z=Z, Z_init=Z_init, variance=s ** 2, jitter=jitter)
    m.likelihood.scale.assign(s)
    return m


def gen_titsias():
    """small Titsias fixtures: ELBO, gradients of every parameter (kernel, scalar scale, inducing points Z incl. its
    gradient-free channel column), predict_f; quirks Q5 (int = per channel) and Q6 (float32-rounded grid)."""
    out = {}
    cases = [(3, 2, 90, [4, 5, 3], False), (2, 3, 120, 6, True), (1, 2, 60, [7], False)]
    out["ncases"] = np.array(len(cases))
    for n, (C, Q, N, Zspec, shuffle) in enumerate(cases):
        rng = np.random.default_rng(9000 + n)
        X, y = small_data(N, C, 1, 9100 + n, shuffle)
        h = dict(weight=rng.uniform(0.5, 1.5, (C, Q)), mean=rng.uniform(0.05, 0.5, (C, Q, 1)),
                 variance=rng.uniform(0.05, 0.5, (C, Q, 1)), delay=rng.normal(0, 0.3, (C, Q, 1)),
                 phase=rng.normal(0, 0.3, (C, Q)), scale=rng.uniform(0.1, 0.4, C))
        m = ref_titsias_mosm(X, y, h, C, Q, Zspec)
        pre = "c%d_" % n
        out[pre + "meta"] = np.array([C, Q, 1, 1]); out[pre + "X"] = X; out[pre + "y"] = y
        out[pre + "Zspec"] = np.atleast_1d(np.array(Zspec)); out[pre + "Zspec_is_int"] = np.array(isinstance(Zspec, int))
        out[pre + "jitter"] = np.array(m.jitter)
        out[pre + "elbo"] = np.array(float(m.log_marginal_likelihood()))
        out[pre + "loss"] = np.array(float(m.loss()))
        dump_params(pre, list(m.parameters()), out, with_grad=True)
        Xs, _ = small_data(23, C, 1, 9200 + n, shuffle)
        mu, var = m.predict_f(T(Xs))
        out[pre + "Xs"] = Xs; out[pre + "mu"] = mu.numpy(); out[pre + "var"] = var.numpy()
    np.savez_compressed(os.path.join(HERE, "titsias.npz"), **out)
    print("titsias.npz written")


def gen_titsias_mohsm():
    """Titsias bound with an ENVELOPED kernel (MixtureKernel of MultiOutputHarmonizableSpectralKernel): the reference runs any kernel under
    any inference (gpr/multioutput.py:340-395 under gpr/model.py:700-724).  ELBO, gradients of every parameter incl. lengthscale / center
    and the inducing inputs (through the envelope too), predict_f."""
    out = {}
    cases = [(3, 2, 1, 96, [5, 4, 6], False), (2, 1, 2, 80, [4, 9], True), (1, 2, 1, 60, [9], False)]
    out["ncases"] = np.array(len(cases))
    for n, (C, Q, D, N, Zspec, shuffle) in enumerate(cases):
        rng = np.random.default_rng(9500 + n)
        X, y = small_data(N, C, D, 9600 + n, shuffle)
        k = build_kernel("mohsm", C, Q, D, 1, rng)
        for q in range(Q):       # K_uu carries no noise: keep the cross-channel blocks a valid covariance (shared spectrum, no delay / phase,
            k[q].mean.assign(np.tile(rng.uniform(0.05, 0.5, (1, D)), (C, 1)))          # one lengthscale): a coregionalised harmonizable kernel
            k[q].variance.assign(np.tile(rng.uniform(0.05, 0.5, (1, D)), (C, 1)))
            k[q].lengthscale.assign(np.full(C, rng.uniform(0.1, 0.4)))
            k[q].delay.assign(np.zeros((C, D)))
            k[q].phase.assign(np.zeros(C))
        s = float(rng.uniform(0.15, 0.4))
        m = g.Titsias(k, T(X), T(y), Z=Zspec, Z_init="grid", variance=s ** 2, jitter=1e-6)
        m.likelihood.scale.assign(s)
        pre = "c%d_" % n
        out[pre + "meta"] = np.array([C, Q, D, 1]); out[pre + "X"] = X; out[pre + "y"] = y
        out[pre + "Zspec"] = np.atleast_1d(np.array(Zspec)); out[pre + "scale"] = np.array(s)
        out[pre + "jitter"] = np.array(m.jitter)
        out[pre + "elbo"] = np.array(float(m.log_marginal_likelihood()))
        out[pre + "loss"] = np.array(float(m.loss()))
        dump_params(pre, list(m.parameters()), out, with_grad=True)
        Xs, _ = small_data(21, C, D, 9700 + n, shuffle)
        mu, var = m.predict_f(T(Xs))
        out[pre + "Xs"] = Xs; out[pre + "mu"] = mu.numpy(); out[pre + "var"] = var.numpy()
    np.savez_compressed(os.path.join(HERE, "titsias_mohsm.npz"), **out)
    print("titsias_mohsm.npz written")


def gen_snelson_mohsm():
    """Snelson (FITC) with an ENVELOPED kernel (MixtureKernel of MultiOutputHarmonizableSpectralKernel; reference gpr/multioutput.py:340-395 under
    gpr/model.py:516-576): marginal likelihood, gradients of every parameter incl. lengthscale / center, the noise scale (scalar and per channel)
    and the inducing inputs (through the envelope and the point-dependent jitter too), predict_f."""
    out = {}
    cases = [(3, 2, 1, 96, [5, 4, 6], False, "vector"), (2, 1, 2, 80, [4, 9], True, "scalar"), (1, 2, 1, 60, [9], False, "scalar")]
    out["ncases"] = np.array(len(cases))
    for n, (C, Q, D, N, Zspec, shuffle, noise) in enumerate(cases):
        rng = np.random.default_rng(9900 + n)
        X, y = small_data(N, C, D, 9950 + n, shuffle)
        k = build_kernel("mohsm", C, Q, D, 1, rng)
        for q in range(Q):       # as in gen_titsias_mohsm: K_uu carries no noise, keep its cross-channel blocks a valid covariance
            k[q].mean.assign(np.tile(rng.uniform(0.05, 0.5, (1, D)), (C, 1)))
            k[q].variance.assign(np.tile(rng.uniform(0.05, 0.5, (1, D)), (C, 1)))
            k[q].lengthscale.assign(np.full(C, rng.uniform(0.1, 0.4)))
            k[q].delay.assign(np.zeros((C, D)))
            k[q].phase.assign(np.zeros(C))
        var = rng.uniform(0.02, 0.15, C) if noise == "vector" else float(rng.uniform(0.02, 0.15))
        m = g.Snelson(k, T(X), T(y), Z=Zspec, Z_init="grid", variance=(T(var) if noise == "vector" else var), jitter=1e-6)
        pre = "c%d_" % n
        out[pre + "meta"] = np.array([C, Q, D, 1]); out[pre + "X"] = X; out[pre + "y"] = y
        out[pre + "Zspec"] = np.atleast_1d(np.array(Zspec)); out[pre + "variance"] = np.asarray(var)
        out[pre + "jitter"] = np.array(m.jitter)
        out[pre + "lml"] = np.array(float(m.log_marginal_likelihood()))
        out[pre + "loss"] = np.array(float(m.loss()))
        dump_params(pre, list(m.parameters()), out, with_grad=True)
        Xs, _ = small_data(21, C, D, 9990 + n, shuffle)
        mu, var_p = m.predict_f(T(Xs))
        out[pre + "Xs"] = Xs; out[pre + "mu"] = mu.numpy(); out[pre + "var"] = var_p.numpy()
    np.savez_compressed(os.path.join(HERE, "snelson_mohsm.npz"), **out)
    print("snelson_mohsm.npz written")


def gen_cfg5():
    import time
    C, Q, N, M = 4, 3, 100000, 2048
    X, y = synth.make_data(N, C)
    h = synth.mosm_hypers(C, Q)
    m = ref_titsias_mosm(X, y, h, C, Q, [M // C] * C)
    t = time.time()
    loss = float(m.loss())
    dt = time.time() - t
    out = {"meta": np.array([C, Q, 1, 1, N, M]), "loss": np.array(loss), "elbo": np.array(-loss), "seconds": np.array(dt),
           "scale": np.array(float(np.mean(h["scale"])))}
    dump_params("", list(m.parameters()), out, with_grad=True)
    np.savez_compressed(os.path.join(HERE, "cfg5.npz"), **out)
    print("cfg5.npz loss=%.10f (%.1f s)" % (loss, dt))


def gen_snelson():
    """small Snelson (FITC) fixtures, reference gpr/model.py:485-576: marginal likelihood, gradients of every parameter (kernel, scalar and
    per-channel noise scale, inducing points incl. the gradient-free channel column), predict_f"""
    out = {}
    cases = [(3, 2, 90, [4, 5, 3], False, "vector"), (2, 3, 120, 6, True, "scalar"), (1, 2, 60, [7], False, "scalar"), (2, 2, 80, 5, False, "vector")]
    out["ncases"] = np.array(len(cases))
    for n, (C, Q, N, Zspec, shuffle, noise) in enumerate(cases):
        rng = np.random.default_rng(9500 + n)
        X, y = small_data(N, C, 1, 9600 + n, shuffle)
        k = g.MultiOutputSpectralMixtureKernel(Q=Q, output_dims=C, input_dims=1)
        k.weight.assign(rng.uniform(0.5, 1.5, (C, Q))); k.mean.assign(rng.uniform(0.05, 0.5, (C, Q, 1)))
        k.variance.assign(rng.uniform(0.05, 0.5, (C, Q, 1))); k.delay.assign(rng.normal(0, 0.3, (C, Q, 1))); k.phase.assign(rng.normal(0, 0.3, (C, Q)))
        var = rng.uniform(0.02, 0.15, C) if noise == "vector" else float(rng.uniform(0.02, 0.15))
        Z = Zspec if isinstance(Zspec, int) else None
        if Z is None:
            Z = np.concatenate([np.stack([np.full(z, float(c)), np.sort(rng.uniform(0, 10, z))], axis=1) for c, z in enumerate(Zspec)])
        m = g.Snelson(k, T(X), T(y), Z=(Z if isinstance(Z, int) else T(Z)), variance=(T(var) if noise == "vector" else var), jitter=1e-6)
        pre = "c%d_" % n
        out[pre + "meta"] = np.array([C, Q, 1, 1]); out[pre + "X"] = X; out[pre + "y"] = y
        out[pre + "Z"] = m.Z().detach().numpy(); out[pre + "variance"] = np.asarray(var); out[pre + "jitter"] = np.array(m.jitter)
        out[pre + "kparams"] = np.array(0)
        for name in ("weight", "mean", "variance", "delay", "phase"):
            out[pre + "k_" + name] = getattr(k, name)().detach().numpy()
        out[pre + "lml"] = np.array(float(m.log_marginal_likelihood()))
        out[pre + "loss"] = np.array(float(m.loss()))
        dump_params(pre, list(m.parameters()), out, with_grad=True)
        Xs, _ = small_data(23, C, 1, 9700 + n, shuffle)
        mu, var_p = m.predict_f(T(Xs))
        out[pre + "Xs"] = Xs; out[pre + "mu"] = mu.numpy(); out[pre + "var"] = var_p.numpy()
    np.savez_compressed(os.path.join(HERE, "snelson.npz"), **out)
    print("snelson.npz written")


def gen_hensman():
    """small SparseHensman / Hensman fixtures with the Gaussian likelihood (reference gpr/model.py:767-886): ELBO, gradients of every
    parameter (q_mu, q_sqrt, inducing points, kernel, scale), predict_f; q_mu / q_sqrt away from their initial values"""
    out = {}
    cases = [(3, 2, 90, [4, 5, 3]), (2, 3, 120, 6), (1, 2, 60, [7]), (2, 1, 40, None)]
    out["ncases"] = np.array(len(cases))
    for n, (C, Q, N, Zspec) in enumerate(cases):
        rng = np.random.default_rng(9800 + n)
        X, y = small_data(N, C, 1, 9900 + n, False)
        k = g.MultiOutputSpectralMixtureKernel(Q=Q, output_dims=C, input_dims=1)
        k.weight.assign(rng.uniform(0.5, 1.5, (C, Q))); k.mean.assign(rng.uniform(0.05, 0.5, (C, Q, 1)))
        k.variance.assign(rng.uniform(0.05, 0.5, (C, Q, 1))); k.delay.assign(rng.normal(0, 0.3, (C, Q, 1))); k.phase.assign(rng.normal(0, 0.3, (C, Q)))
        lik = g.GaussianLikelihood(float(rng.uniform(0.1, 0.4)))
        if Zspec is None:
            m = g.Hensman(k, T(X), T(y), likelihood=lik, jitter=1e-6)
        else:
            Z = Zspec if isinstance(Zspec, int) else T(np.concatenate([np.stack([np.full(z, float(c)), np.sort(rng.uniform(0, 10, z))], axis=1) for c, z in enumerate(Zspec)]))
            m = g.SparseHensman(k, T(X), T(y), Z=Z, likelihood=lik, jitter=1e-6)
        M = m.q_mu().shape[0]
        m.q_mu.assign(rng.normal(0, 0.5, (M, 1)))
        m.q_sqrt.assign(np.tril(rng.normal(0, 0.2, (M, M))) + np.diag(rng.uniform(0.5, 1.2, M)) + np.triu(rng.normal(0, 0.1, (M, M)), 1))
        pre = "c%d_" % n
        out[pre + "meta"] = np.array([C, Q, 1, 1]); out[pre + "X"] = X; out[pre + "y"] = y
        out[pre + "sparse"] = np.array(Zspec is not None)
        out[pre + "Z"] = m.Z().detach().numpy(); out[pre + "jitter"] = np.array(m.jitter)
        out[pre + "elbo"] = np.array(float(m.log_marginal_likelihood()))
        out[pre + "loss"] = np.array(float(m.loss()))
        dump_params(pre, list(m.parameters()), out, with_grad=True)
        Xs, _ = small_data(23, C, 1, 9950 + n, False)
        mu, var_p = m.predict_f(T(Xs))
        out[pre + "Xs"] = Xs; out[pre + "mu"] = mu.numpy(); out[pre + "var"] = var_p.numpy()
    np.savez_compressed(os.path.join(HERE, "hensman.npz"), **out)
    print("hensman.npz written")


def gen_hensman_mohsm():
    """SparseHensman / Hensman (Gaussian and Student-t likelihoods) with an ENVELOPED kernel (reference gpr/multioutput.py:340-395 under
    gpr/model.py:767-886): ELBO, gradients of every parameter (q_mu, q_sqrt, inducing inputs through the envelope and the point-dependent
    jitter, lengthscale / center, likelihood), predict_f."""
    out = {}
    cases = [(3, 2, 1, 96, [5, 4, 6], "gaussian"), (2, 1, 2, 80, [4, 9], "studentt"), (2, 2, 1, 40, None, "gaussian")]
    out["ncases"] = np.array(len(cases))
    for n, (C, Q, D, N, Zspec, lik_name) in enumerate(cases):
        rng = np.random.default_rng(9970 + n)
        X, y = small_data(N, C, D, 9975 + n, False)
        k = build_kernel("mohsm", C, Q, D, 1, rng)
        for q in range(Q):       # as in gen_titsias_mohsm: keep K_uu's cross-channel blocks a valid covariance
            k[q].mean.assign(np.tile(rng.uniform(0.05, 0.5, (1, D)), (C, 1)))
            k[q].variance.assign(np.tile(rng.uniform(0.05, 0.5, (1, D)), (C, 1)))
            k[q].lengthscale.assign(np.full(C, rng.uniform(0.1, 0.4)))
            k[q].delay.assign(np.zeros((C, D)))
            k[q].phase.assign(np.zeros(C))
        lik = g.GaussianLikelihood(float(rng.uniform(0.1, 0.4))) if lik_name == "gaussian" else g.StudentTLikelihood(dof=4, scale=float(rng.uniform(0.2, 0.4)))
        if Zspec is None:
            m = g.Hensman(k, T(X), T(y), likelihood=lik, jitter=1e-6)
        else:
            m = g.SparseHensman(k, T(X), T(y), Z=Zspec, Z_init="grid", likelihood=lik, jitter=1e-6)
        M = m.q_mu().shape[0]
        m.q_mu.assign(rng.normal(0, 0.5, (M, 1)))
        m.q_sqrt.assign(np.tril(rng.normal(0, 0.2, (M, M))) + np.diag(rng.uniform(0.5, 1.2, M)))
        pre = "c%d_" % n
        out[pre + "meta"] = np.array([C, Q, D, 1]); out[pre + "X"] = X; out[pre + "y"] = y
        out[pre + "sparse"] = np.array(Zspec is not None); out[pre + "lik"] = np.array(lik_name)
        out[pre + "Zspec"] = np.atleast_1d(np.array(Zspec if Zspec is not None else [0])); out[pre + "jitter"] = np.array(m.jitter)
        out[pre + "elbo"] = np.array(float(m.log_marginal_likelihood()))
        out[pre + "loss"] = np.array(float(m.loss()))
        dump_params(pre, list(m.parameters()), out, with_grad=True)
        Xs, _ = small_data(21, C, D, 9980 + n, False)
        mu, var_p = m.predict_f(T(Xs))
        out[pre + "Xs"] = Xs; out[pre + "mu"] = mu.numpy(); out[pre + "var"] = var_p.numpy()
    np.savez_compressed(os.path.join(HERE, "hensman_mohsm.npz"), **out)
    print("hensman_mohsm.npz written")


def gen_oa():
    """small OpperArchambeau fixtures with the Gaussian likelihood (reference gpr/model.py:578-668): ELBO, gradients of every parameter
    (q_nu, q_lambda, kernel, scale), predict_f (diagonal and full); q_nu / q_lambda away from their initial values, inputs NOT grouped by channel"""
    out = {}
    cases = [(3, 2, 90), (2, 3, 150), (1, 2, 60)]
    out["ncases"] = np.array(len(cases))
    for n, (C, Q, N) in enumerate(cases):
        rng = np.random.default_rng(9600 + n)
        X, y = small_data(N, C, 1, 9700 + n, n != 1)
        k = g.MultiOutputSpectralMixtureKernel(Q=Q, output_dims=C, input_dims=1)
        k.weight.assign(rng.uniform(0.5, 1.5, (C, Q))); k.mean.assign(rng.uniform(0.05, 0.5, (C, Q, 1)))
        k.variance.assign(rng.uniform(0.05, 0.5, (C, Q, 1))); k.delay.assign(rng.normal(0, 0.3, (C, Q, 1))); k.phase.assign(rng.normal(0, 0.3, (C, Q)))
        lik = g.GaussianLikelihood(float(rng.uniform(0.1, 0.4)))
        m = g.OpperArchambeau(k, T(X), T(y), likelihood=lik, jitter=1e-6)
        m.q_nu.assign(rng.normal(0, 0.5, (N, 1)))
        m.q_lambda.assign(rng.uniform(0.5, 3.0, (N, 1)))
        pre = "c%d_" % n
        out[pre + "meta"] = np.array([C, Q, 1, 1]); out[pre + "X"] = X; out[pre + "y"] = y
        out[pre + "elbo"] = np.array(float(m.log_marginal_likelihood()))
        out[pre + "loss"] = np.array(float(m.loss()))
        dump_params(pre, list(m.parameters()), out, with_grad=True)
        Xs, _ = small_data(23, C, 1, 9750 + n, True)
        mu, var_p = m.predict_f(T(Xs))
        _, cov = m.predict_f(T(Xs), full=True)
        out[pre + "Xs"] = Xs; out[pre + "mu"] = mu.numpy(); out[pre + "var"] = var_p.numpy(); out[pre + "cov"] = cov.numpy()
    np.savez_compressed(os.path.join(HERE, "oa.npz"), **out)
    print("oa.npz written")


def _likelihood_zoo(L):
    """(tag, constructor) for every likelihood of the reference, default links plus a few others"""
    return [
        ("gaussian", lambda: L.GaussianLikelihood(0.7)),
        ("studentt", lambda: L.StudentTLikelihood(dof=4, scale=0.6)),
        ("exponential", lambda: L.ExponentialLikelihood()),
        ("laplace", lambda: L.LaplaceLikelihood(scale=0.8)),
        ("bernoulli", lambda: L.BernoulliLikelihood()),
        ("bernoulli_sigmoid", lambda: L.BernoulliLikelihood(link=L.sigmoid)),
        ("beta", lambda: L.BetaLikelihood(scale=3.0)),
        ("beta_sigmoid", lambda: L.BetaLikelihood(scale=2.5, link=L.sigmoid)),
        ("gamma", lambda: L.GammaLikelihood(shape=1.7)),
        ("poisson", lambda: L.PoissonLikelihood()),
        ("weibull", lambda: L.WeibullLikelihood(shape=1.4)),
        ("weibull_square", lambda: L.WeibullLikelihood(shape=0.8, link=L.square)),
        ("loglogistic", lambda: L.LogLogisticLikelihood(shape=2.2)),
        ("loggaussian", lambda: L.LogGaussianLikelihood(scale=0.5)),
        ("chisquared", lambda: L.ChiSquaredLikelihood()),
    ]


def _likelihood_targets(tag, rng, n):
    if tag.startswith("bernoulli"):
        return (rng.uniform(size=n) < 0.5).astype(np.float64)
    if tag.startswith("beta"):
        return rng.uniform(0.05, 0.95, n)
    if tag == "poisson":
        return rng.poisson(2.0, n).astype(np.float64)
    if tag in ("gaussian", "studentt", "laplace"):
        return rng.normal(0, 1.0, n)
    return rng.uniform(0.2, 3.0, n)


def gen_likelihoods():
    """every likelihood of the reference (gpr/likelihood.py): log_prob on a grid of f, the variational expectation with the reference's autograd
    gradients with respect to mu, var and the likelihood's own parameters, conditional_mean, predict (mean; and quantiles under a fixed torch
    seed); a MultiOutputLikelihood of three different members; SparseHensman and OpperArchambeau models with non-Gaussian likelihoods
    (loss, every gradient, predict_f, predict_y)"""
    L = g
    out = {}
    n = 17
    tags = []
    for tag, make in _likelihood_zoo(L):
        rng = np.random.default_rng(sum(ord(c) for c in tag))
        lik = make()
        y = _likelihood_targets(tag, rng, n).reshape(-1, 1)
        mu = rng.normal(0.2, 0.6, (n, 1)); var = rng.uniform(0.05, 0.6, (n, 1))
        X = np.stack([np.zeros(n), np.linspace(0, 1, n)], axis=1)
        lik.validate_y(T(X), T(y))
        f = rng.normal(0.2, 0.8, (n, 5))
        tmu = torch.tensor(mu, dtype=torch.float64, requires_grad=True); tvar = torch.tensor(var, dtype=torch.float64, requires_grad=True)
        ve = lik.variational_expectation(T(X), T(y), tmu, tvar)
        ve.backward()
        out[tag + "_y"] = y; out[tag + "_mu"] = mu; out[tag + "_var"] = var; out[tag + "_f"] = f; out[tag + "_X"] = X
        out[tag + "_logp"] = lik.log_prob(T(X), T(y), T(f)).detach().numpy()
        out[tag + "_ve"] = np.array(float(ve))
        out[tag + "_dmu"] = tmu.grad.numpy().reshape(-1)
        out[tag + "_dvar"] = np.zeros(n) if tvar.grad is None else tvar.grad.numpy().reshape(-1)
        dump_params(tag + "_lik_", list(lik.parameters()), out, with_grad=True)
        with torch.no_grad():
            out[tag + "_cmean"] = lik.conditional_mean(T(X), T(f)).numpy()
            out[tag + "_pmean"] = np.asarray(lik.predict(T(X), T(mu), T(var))).reshape(-1)
            if tag not in ("gaussian",):
                torch.manual_seed(1234)
                try:
                    _, lo, hi = lik.predict(T(X), T(mu), T(var), ci=[0.1, 0.9], n=500)
                    out[tag + "_lo"] = np.asarray(lo).reshape(-1); out[tag + "_hi"] = np.asarray(hi).reshape(-1)
                except Exception as e:                      # a sampler the reference itself cannot run for this configuration
                    out[tag + "_ci_error"] = np.array(type(e).__name__)
        tags.append(tag)
    out["tags"] = np.array(tags)

    # MultiOutputLikelihood: three members, shuffled channels
    rng = np.random.default_rng(4242)
    n = 30
    X = np.stack([rng.integers(0, 3, n).astype(np.float64), rng.uniform(0, 1, n)], axis=1)
    y = np.where(X[:, 0] == 0, rng.normal(0, 1, n), np.where(X[:, 0] == 1, rng.poisson(2.0, n), rng.uniform(0.2, 3.0, n))).reshape(-1, 1)
    lik = L.MultiOutputLikelihood(L.StudentTLikelihood(dof=5, scale=0.5), L.PoissonLikelihood(), L.WeibullLikelihood(shape=1.3))
    lik.validate_y(T(X), T(y))
    mu = rng.normal(0.2, 0.6, (n, 1)); var = rng.uniform(0.05, 0.6, (n, 1))
    tmu = torch.tensor(mu, dtype=torch.float64, requires_grad=True); tvar = torch.tensor(var, dtype=torch.float64, requires_grad=True)
    ve = lik.variational_expectation(T(X), T(y), tmu, tvar)
    ve.backward()
    out["multi_X"] = X; out["multi_y"] = y; out["multi_mu"] = mu; out["multi_var"] = var
    out["multi_ve"] = np.array(float(ve)); out["multi_dmu"] = tmu.grad.numpy().reshape(-1); out["multi_dvar"] = tvar.grad.numpy().reshape(-1)
    dump_params("multi_lik_", list(lik.parameters()), out, with_grad=True)
    with torch.no_grad():
        out["multi_pmean"] = lik.predict(T(X), T(mu), T(var)).numpy().reshape(-1)
        out["multi_name"] = np.array(lik.name())

    # models: SparseHensman + StudentT, SparseHensman + multi-output [Poisson, Gaussian], OpperArchambeau + Bernoulli, Hensman + Laplace
    def kernel(C, Q, rng):
        k = g.MultiOutputSpectralMixtureKernel(Q=Q, output_dims=C, input_dims=1)
        k.weight.assign(rng.uniform(0.5, 1.5, (C, Q))); k.mean.assign(rng.uniform(0.05, 0.5, (C, Q, 1)))
        k.variance.assign(rng.uniform(0.05, 0.5, (C, Q, 1))); k.delay.assign(rng.normal(0, 0.3, (C, Q, 1))); k.phase.assign(rng.normal(0, 0.3, (C, Q)))
        return k

    specs = [("svgp_studentt", 2, "sparse"), ("svgp_multi", 2, "sparse"), ("oa_bernoulli", 2, "oa"), ("hensman_laplace", 1, "dense"), ("oa_gamma", 1, "oa")]
    out["model_tags"] = np.array([sp[0] for sp in specs])
    for tag, C, kind in specs:
        rng = np.random.default_rng(sum(ord(c) for c in tag))
        N = 60
        X, y = small_data(N, C, 1, 5100 + len(tag), False)
        if tag == "svgp_studentt":
            lik = L.StudentTLikelihood(dof=4, scale=0.4)
        elif tag == "svgp_multi":
            y = np.where(X[:, 0] == 0, rng.poisson(np.exp(0.5 * y)), y).astype(np.float64)
            lik = L.MultiOutputLikelihood(L.PoissonLikelihood(), L.GaussianLikelihood(0.3))
        elif tag == "oa_bernoulli":
            y = (y > 0).astype(np.float64)
            lik = L.BernoulliLikelihood()
        elif tag == "hensman_laplace":
            lik = L.LaplaceLikelihood(scale=0.3)
        else:
            y = np.exp(0.5 * y) * rng.gamma(2.0, 0.5, N)
            lik = L.GammaLikelihood(shape=2.0)
        k = kernel(C, 2, rng)
        if kind == "sparse":
            Z = T(np.concatenate([np.stack([np.full(5, float(c)), np.sort(rng.uniform(0, 10, 5))], axis=1) for c in range(C)]))
            m = g.SparseHensman(k, T(X), T(y), Z=Z, likelihood=lik, jitter=1e-6)
        elif kind == "dense":
            m = g.Hensman(k, T(X), T(y), likelihood=lik, jitter=1e-6)
        else:
            m = g.OpperArchambeau(k, T(X), T(y), likelihood=lik, jitter=1e-6)
        if kind == "oa":
            m.q_nu.assign(rng.normal(0, 0.3, (N, 1))); m.q_lambda.assign(rng.uniform(0.5, 2.0, (N, 1)))
        else:
            M = m.q_mu().shape[0]
            m.q_mu.assign(rng.normal(0, 0.5, (M, 1)))
            m.q_sqrt.assign(np.tril(rng.normal(0, 0.2, (M, M))) + np.diag(rng.uniform(0.5, 1.2, M)))
        pre = tag + "_"
        out[pre + "meta"] = np.array([C, 2, 1, 1]); out[pre + "X"] = X; out[pre + "y"] = y
        if kind == "sparse":
            out[pre + "Z"] = m.Z().detach().numpy()
        out[pre + "loss"] = np.array(float(m.loss()))
        dump_params(pre, list(m.parameters()), out, with_grad=True)
        Xs, _ = small_data(15, C, 1, 5200 + len(tag), False)
        mu_f, var_f = m.predict_f(T(Xs))
        out[pre + "Xs"] = Xs; out[pre + "mu_f"] = mu_f.numpy(); out[pre + "var_f"] = var_f.numpy()
        out[pre + "mu_y"] = np.asarray(m.predict_y(T(Xs))).reshape(-1)
    np.savez_compressed(os.path.join(HERE, "likelihoods.npz"), **out)
    print("likelihoods.npz written", len(tags), "likelihoods")


def gen_samples():
    """sample_f / sample_y / Model.sample under a fixed torch seed (reference gpr/model.py:346-401, model.py:692-734): a multi-output exact
    model's posterior samples of f, and a single-output model's samples of y through Model.sample"""
    out = {}
    rng = np.random.default_rng(31)
    X, y = small_data(40, 2, 1, 3100, True)
    k = build_kernel("mosm", 2, 2, 1, 1, rng)
    m = g.Exact(k, T(X), T(y), variance=[0.2, 0.3], jitter=1e-8)
    Z, _ = small_data(9, 2, 1, 3101, False)
    torch.manual_seed(99)
    out["f_single"] = m.sample_f(T(Z)).numpy()
    torch.manual_seed(99)
    out["f_many"] = m.sample_f(T(Z), n=4).numpy()
    out["X"] = X; out["y"] = y; out["Z"] = Z
    dump_params("exact_", list(m.parameters()), out)
    t = np.linspace(0, 10, 35)
    d = mogptk.Data(t, np.sin(t) + 0.05 * t + 0.1 * rng.standard_normal(35))
    d.transform(mogptk.TransformDetrend(degree=1))
    d.set_prediction_data(np.linspace(0, 11, 8))
    ms = mogptk.SM(mogptk.DataSet(d), Q=2)
    for p in ms.gpr.parameters():
        v = p.constrained.detach().numpy()
        p.assign(v * rng.uniform(0.8, 1.2, v.shape))
    torch.manual_seed(7)
    out["sm_sample"] = np.asarray(ms.sample())
    torch.manual_seed(7)
    out["sm_sample_transformed"] = np.asarray(ms.sample(transformed=True))
    out["sm_t"] = t; out["sm_y"] = d.Y.copy() if hasattr(d, "Y") else None
    out["sm_y_raw"] = np.sin(t) + 0.05 * t
    dump_params("sm_", list(ms.gpr.parameters()), out)
    out["sm_Y"] = np.asarray(ms.dataset[0].Y)
    np.savez_compressed(os.path.join(HERE, "samples.npz"), **{k2: v for k2, v in out.items() if v is not None})
    print("samples.npz written")


def gen_sparse_cov():
    """full predictive covariance of the sparse models (Titsias / SparseHensman / Hensman predict_f(full=True), reference
    gpr/model.py:758-760, 870-872) and posterior samples of f drawn from it under a fixed seed"""
    out = {}
    rng = np.random.default_rng(616)
    C, Q, N = 2, 2, 80
    X, y = small_data(N, C, 1, 6160, False)
    Xs, _ = small_data(11, C, 1, 6161, True)
    Z = np.concatenate([np.stack([np.full(6, float(c)), np.sort(rng.uniform(0, 10, 6))], axis=1) for c in range(C)])
    out["X"] = X; out["y"] = y; out["Xs"] = Xs; out["Z"] = Z
    for tag in ("titsias", "sparse_hensman", "hensman"):
        k = build_kernel("mosm", C, Q, 1, 1, rng)
        if tag == "titsias":
            m = g.Titsias(k, T(X), T(y), Z=T(Z), variance=0.09, jitter=1e-6)
        elif tag == "sparse_hensman":
            m = g.SparseHensman(k, T(X), T(y), Z=T(Z), likelihood=g.GaussianLikelihood(0.3), jitter=1e-6)
        else:
            m = g.Hensman(k, T(X), T(y), likelihood=g.GaussianLikelihood(0.3), jitter=1e-6)
        if tag != "titsias":
            M = m.q_mu().shape[0]
            m.q_mu.assign(rng.normal(0, 0.5, (M, 1)))
            m.q_sqrt.assign(np.tril(rng.normal(0, 0.1, (M, M))) + np.diag(rng.uniform(0.5, 1.0, M)))
        dump_params(tag + "_", list(m.parameters()), out)
        mu, cov = m.predict_f(T(Xs), full=True)
        out[tag + "_mu"] = mu.numpy(); out[tag + "_cov"] = cov.numpy()
        torch.manual_seed(5)
        out[tag + "_sample"] = m.sample_f(T(Xs)).numpy()
    np.savez_compressed(os.path.join(HERE, "sparse_cov.npz"), **out)
    print("sparse_cov.npz written")


def gen_checkpoints():
    """Files written by the reference's own Model.save() (model.py:320-336) -- the whole pickled model: MOSM with a fitted transformer
    chain, removed points and a pegged + a fixed parameter; the SM, CSM, SM-LMC and CONV wrappers; a Titsias MOSM -- stored as bytes next to what
    the reference computes on the loaded object: constrained parameter values, loss, predictions at the stored prediction inputs."""
    import tempfile
    rng = np.random.default_rng(77)
    out = {}

    def dataset(C, n, transform=False):
        ds = mogptk.DataSet()
        for c in range(C):
            x = np.sort(rng.uniform(0, 10, n))
            y = np.sin(x * (1 + 0.5 * c)) + 0.05 * x + 0.1 * rng.standard_normal(n)
            d = mogptk.Data(x, y, name="ch%d" % c)
            if transform:
                d.transform(mogptk.TransformDetrend(degree=1))
                d.transform(mogptk.TransformStandard())
            d.mask[rng.permutation(n)[:n // 5]] = False
            d.set_prediction_data(np.linspace(0, 11, 7))
            ds.append(d)
        return ds

    def randomise(model):
        for p in model.gpr.parameters():
            if p.pegged or (p._name or "").endswith("induction_points") or (p._name or "").split(".")[-1] in ("q_mu", "q_sqrt", "q_nu", "q_lambda"):      # the channel column of Z must stay integral; the variational parameters are moved by the training steps below
                continue
            v = p.constrained.detach().numpy()
            lo = None if p.lower is None else np.broadcast_to(p.lower.detach().numpy(), v.shape)
            hi = None if p.upper is None else np.broadcast_to(p.upper.detach().numpy(), v.shape)
            new = v * rng.uniform(0.8, 1.2, v.shape) + (rng.normal(0, 0.05, v.shape) if lo is None else 0.0)
            if hi is not None:
                new = np.minimum(new, lo + 0.9 * (hi - lo))
            if lo is not None:
                new = np.maximum(new, lo + 1e-3 * (1 + np.abs(lo)))
            p.assign(new)

    def record(tag, model):
        with tempfile.TemporaryDirectory() as d:
            model.save(os.path.join(d, "m"))
            raw = open(os.path.join(d, "m.npy"), "rb").read()
            loaded = mogptk.LoadModel(os.path.join(d, "m"))
        out[tag + "_file"] = np.frombuffer(raw, dtype=np.uint8)
        out[tag + "_names"] = np.array([p._name for p in loaded.gpr.parameters()])
        for i, p in enumerate(loaded.gpr.parameters()):
            out["%s_p%d" % (tag, i)] = p.constrained.detach().numpy()
            out["%s_train%d" % (tag, i)] = np.array(bool(p.train))
        out[tag + "_loss"] = np.array(float(loaded.loss()))
        for i, p in enumerate(loaded.gpr.parameters()):
            out["%s_g%d" % (tag, i)] = np.zeros(0) if p.grad is None else p.grad.detach().numpy()
        torch.manual_seed(4321)                               # the intervals of non-Gaussian likelihoods are sampled from torch's generator
        X, mu, lower, upper = loaded.predict(transformed=False)
        out[tag + "_mu"] = np.concatenate([np.asarray(m).reshape(-1) for m in mu])
        out[tag + "_lower"] = np.concatenate([np.asarray(m).reshape(-1) for m in lower])
        out[tag + "_upper"] = np.concatenate([np.asarray(m).reshape(-1) for m in upper])
        out[tag + "_history"] = np.array([loaded.iters, len(loaded.times), len(loaded.losses)])

    m = mogptk.MOSM(dataset(2, 40, transform=True), Q=2)
    randomise(m)
    m.gpr.kernel.phase.train = False
    m.gpr.likelihood.scale.assign([0.2, 0.3])
    m.train(method="Adam", lr=0.01, iters=3, verbose=False)
    record("mosm", m)
    m = mogptk.SM(dataset(2, 30), Q=2); randomise(m)
    import functools, operator
    m.gpr.kernel[1].mean.peg(m.gpr.kernel[0].mean, functools.partial(operator.mul, 2.0))      # a transform that pickles (a lambda does not)
    record("sm", m)
    m = mogptk.CSM(dataset(2, 30), Q=2, Rq=2); randomise(m); record("csm", m)
    m = mogptk.SM_LMC(dataset(3, 25), Q=2, Rq=1); randomise(m); record("smlmc", m)
    m = mogptk.CONV(dataset(2, 30), Q=2); randomise(m); record("conv", m)
    m = mogptk.MOSM(dataset(2, 60), Q=1, inference=mogptk.Titsias(inducing_points=8)); randomise(m); record("titsias", m)
    m = mogptk.MOSM(dataset(2, 60), Q=1, inference=mogptk.Snelson(inducing_points=7)); randomise(m); record("snelson", m)
    m = mogptk.MOSM(dataset(2, 60), Q=1, inference=mogptk.Hensman(inducing_points=6)); randomise(m)
    m.train(method="Adam", lr=0.02, iters=4, verbose=False); record("hensman", m)
    m = mogptk.MOSM(dataset(2, 40), Q=1, inference=mogptk.OpperArchambeau()); randomise(m)
    m.train(method="Adam", lr=0.02, iters=4, verbose=False); record("oa", m)
    ds = dataset(2, 40)
    ds[0].Y = np.round(np.exp(ds[0].Y)).astype(np.float64)           # counts for the Poisson channel
    lik = mogptk.gpr.MultiOutputLikelihood(mogptk.gpr.PoissonLikelihood(), mogptk.gpr.StudentTLikelihood(dof=5, scale=0.4, quadratures=12))
    m = mogptk.MOSM(ds, Q=1, inference=mogptk.Hensman(inducing_points=5, likelihood=lik)); randomise(m)
    m.train(method="Adam", lr=0.02, iters=3, verbose=False); record("hensman_lik", m)
    np.savez_compressed(os.path.join(HERE, "checkpoints.npz"), **out)
    print("checkpoints.npz", {k: v.shape for k, v in out.items() if k.endswith("_file")})


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--full", action="store_true")
    ap.add_argument("--only", default=None)
    a = ap.parse_args()
    steps = {"kernels": gen_kernels, "lml": gen_lml, "predict": gen_predict, "adam": gen_adam_cfg1, "lbfgs": gen_lbfgs_cfg1, "quirks": gen_quirks,
             "kernels_8f2": gen_kernels_8f2, "lml_8f2": gen_lml_8f2, "smlmc": gen_smlmc, "init_ls": gen_init_ls, "bnse": gen_bnse, "transformers": gen_transformers,
             "titsias": gen_titsias, "titsias_mohsm": gen_titsias_mohsm, "snelson_mohsm": gen_snelson_mohsm, "hensman_mohsm": gen_hensman_mohsm, "opt_traces": gen_opt_traces, "peg": gen_peg, "mohsm": gen_mohsm, "fp32": gen_fp32, "checkpoints": gen_checkpoints, "snelson": gen_snelson, "hensman": gen_hensman, "oa": gen_oa, "likelihoods": gen_likelihoods, "samples": gen_samples, "sparse_cov": gen_sparse_cov}
    full = {"cfg2": gen_cfg2, "cfg4": gen_cfg4, "cfg5": gen_cfg5}
    if a.only:
        {**steps, **full}[a.only]()
    else:
        for f in steps.values():
            f()
        if a.full:
            for f in full.values():
                f()
