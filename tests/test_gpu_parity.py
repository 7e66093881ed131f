"""
Parity tests proper: the HIP path, called through the C ABI, against the reference's golden vectors and the numpy
oracle on the same seeded inputs.  Tolerances: BASELINE.json's north_star asks for <= 1e-5 relative on the log
marginal likelihood and its gradient; most checks here are much tighter (fp64 end to end) and say so.
Run on the GPU box:  python -m pytest tests -m gpu
"""
import numpy as np
import pytest

import mogptk_amd
from mogptk_amd import gpr, synth, _lib
from helpers import load, fixture_params, product_exact, product_kernel, load_raw, relerr
from oracle.table_model import TableDevice, gram_from_table

pytestmark = pytest.mark.gpu


def grad_close(p, ref, tol):
    return np.max(np.abs(p - ref)) <= tol * max(np.max(np.abs(ref)), 1e-300)


def test_native_library_is_the_path():
    assert _lib.lib().mogp_device_count() >= 1
    assert "gfx950" in _lib.device_name(0)


@pytest.mark.parametrize("fixture", ["kernels.npz", "kernels_8f2.npz", "kernels_mohsm.npz"])
def test_gram_matches_reference_fixtures(fixture):
    fx = load(fixture)
    for n in range(int(fx["ncases"])):
        pre = "c%d_" % n
        C, Q, D, Rq = [int(v) for v in fx[pre + "meta"]]
        k = product_kernel(str(fx[pre + "kind"]), C, Q, D, Rq)
        load_raw(k.parameters(), fixture_params(fx, pre))
        X, X2 = fx[pre + "X"], fx[pre + "X2"]
        K = k.K(X)
        assert relerr(K, fx[pre + "K"]) < 1e-12, (n, relerr(K, fx[pre + "K"]))
        assert np.array_equal(K, K.T)                                    # mirrored, not recomputed
        assert relerr(k(X, X2), fx[pre + "K12"]) < 1e-12
        assert relerr(k.K_diag(X), fx[pre + "Kdiag"]) < 1e-14


LML = ["mosm_c3q2", "mosm_c2q3_shuf", "mosm_c3q2_d2", "mosm_c1q2", "mosm_scalarvar", "sm_c1q3", "sm_c2q2_d2",
       "csm_c3q2", "csm_c2q2r2",
       "mosk_c3q2", "mosk_c2q1_d2", "umosm_c3q2", "umosm_c2q2_d2", "lmc_c3q2r2", "lmc_c2q3_d2", "lmcsm_c2q2", "conv_c3q2", "conv_c2q1_d2",          # same term table, other parameter algebra
       "mohsm_c3q2", "mohsm_c2q1_d2", "mohsm_c1q2"]                                # ... and the enveloped (2 + 5 D) rows of MOHSM


@pytest.mark.parametrize("name", LML)
def test_lml_and_gradient_match_reference_autograd(name):
    fx = load("lml_%s.npz" % name)
    m, fp = product_exact(fx)
    assert abs(float(m.log_marginal_likelihood()) - float(fx["lml"])) < 1e-9 * abs(float(fx["lml"]))
    loss = m.loss()
    assert abs(float(loss) - float(fx["loss"])) < 1e-9 * abs(float(fx["loss"]))
    for p, f in zip(m.parameters(), fp):
        if f["grad"] is None:
            assert p.grad is None
        else:
            assert np.max(np.abs(p.grad - f["grad"])) <= 1e-7 * max(1.0, np.max(np.abs(f["grad"]))), (p._name, p.grad, f["grad"])


@pytest.mark.parametrize("N,C,Q,D,vmax", [(300, 3, 2, 1, 0.05), (517, 4, 3, 1, 0.05), (260, 2, 9, 1, 0.05), (200, 3, 2, 2, 0.05), (129, 1, 2, 1, 0.05),
                                          (1350, 3, 2, 1, 0.05), (2900, 4, 1, 1, 0.05), (1500, 2, 6, 1, 0.05), (1100, 2, 2, 1, 3.0)])
def test_device_raw_outputs_against_numpy_model(N, C, Q, D, vmax):
    """moments / diagG / trG / alpha / L^-1 / K^-1 of the device against the numpy restatement: ragged channel sizes,
    N not a multiple of the 128 tile, more terms than one LDS chunk (Q=9), D=2; runs of full interior tiles (the Gram strip kernel) with
    up to four terms, with six, and with spectral variances so large that tiles take the exp-per-entry path (vmax = 3)."""
    rng = np.random.default_rng(N)
    sizes = rng.multinomial(N - C, np.ones(C) / C) + 1
    X = np.concatenate([np.concatenate([np.full((s, 1), float(c)), rng.uniform(0, 30, (s, D))], axis=1) for c, s in enumerate(sizes)])
    X = X[rng.permutation(N)]
    y = rng.standard_normal(N)
    k = gpr.MultiOutputSpectralMixtureKernel(Q=Q, output_dims=C, input_dims=D)
    k.weight.assign(rng.uniform(0.5, 1.5, (C, Q))); k.mean.assign(rng.uniform(0.02, 0.4, (C, Q, D)))
    k.variance.assign(rng.uniform(0.1 * vmax, vmax, (C, Q, D))); k.delay.assign(rng.normal(0, 0.3, (C, Q, D)))
    k.phase.assign(rng.normal(0, 0.3, (C, Q)))
    table = k._spectral_terms(D)
    noise = rng.uniform(0.05, 0.2, C)
    dev = _lib.ExactHandle(0, X, y, C)
    ref = TableDevice(0, X, y, C)
    dev.set_terms(table); ref.set_terms(table)
    a = dev.eval(noise, 1e-8, grad=True)
    b = ref.eval(noise, 1e-8, grad=True)
    assert abs(a["lml"] - b["lml"]) < 1e-10 * abs(b["lml"])
    assert abs(a["jitter_abs"] - b["jitter_abs"]) < 1e-14 * b["jitter_abs"]
    scale = np.max(np.abs(b["moments"]))
    assert np.max(np.abs(a["moments"] - b["moments"])) < 1e-9 * scale, np.max(np.abs(a["moments"] - b["moments"])) / scale
    assert np.max(np.abs(a["diagG"] - b["diagG"])) < 1e-9 * np.max(np.abs(b["diagG"]))
    assert abs(a["trG"] - b["trG"]) < 1e-9 * abs(b["trG"])
    # matrices: K^-1 and alpha in the caller's row order
    Kj, _ = ref._Kj(noise, 1e-8, None)
    Kinv = np.linalg.inv(Kj)
    assert relerr(dev.fetch(1), Kinv) < 1e-9
    assert relerr(dev.fetch(2), Kinv @ y) < 1e-9
    # a data_variance vector goes through the same path
    dvar = rng.uniform(0.0, 0.1, N)
    a2 = dev.eval(noise, 1e-8, grad=False, data_var=dvar)
    b2 = ref.eval(noise, 1e-8, grad=False, data_var=dvar)
    assert abs(a2["lml"] - b2["lml"]) < 1e-10 * abs(b2["lml"])


def test_predict_matches_reference():
    fx = load("predict.npz")
    for n in range(int(fx["ncases"])):
        pre = "c%d_" % n
        m, fp = product_exact(fx, pre)
        Xs = fx[pre + "Xs"]
        mu, var = m.predict_f(Xs)
        assert relerr(mu, fx[pre + "mu"]) < 1e-9 and np.max(np.abs(var - fx[pre + "var"])) < 1e-9
        _, cov = m.predict_f(Xs, full=True)
        assert np.max(np.abs(cov - fx[pre + "cov"])) < 1e-9
        _, lo, up = m.predict_y(Xs, sigma=2.0)
        assert relerr(lo, fx[pre + "lower"]) < 1e-9 and relerr(up, fx[pre + "upper"]) < 1e-9
    k = gpr.SpectralMixtureKernel(Q=2, input_dims=1)
    m = gpr.Exact(k, fx["so_X"], fx["so_y"], variance=0.04)
    load_raw(m.parameters(), fixture_params(fx, "so_"))
    assert abs(float(m.log_marginal_likelihood()) - float(fx["so_lml"])) < 1e-9
    mu, var = m.predict_f(fx["so_Xs"])
    assert relerr(mu, fx["so_mu"]) < 1e-9 and np.max(np.abs(var - fx["so_var"])) < 1e-9


def test_adam_trajectory_cfg1_airline():
    """BASELINE.json configs[0]: all 100 iterations of the reference's Adam run, loss trace and final raw parameters."""
    fx = load("adam_cfg1.npz")
    data = mogptk_amd.Data(fx["X"][:, 1], fx["y"][:, 0], name="airline")
    model = mogptk_amd.SM(data, Q=3)
    load_raw(model.gpr.parameters(), fixture_params(fx, "init_"))
    assert abs(model.log_marginal_likelihood() - float(fx["lml0"])) < 1e-9
    losses, _ = model.train("Adam", iters=int(fx["iters"]), lr=float(fx["lr"]))
    assert relerr(losses, fx["losses"]) < 1e-7
    for p, f in zip(model.gpr.parameters(), fixture_params(fx, "final_")):
        assert np.max(np.abs(p.data - f["raw"])) < 1e-5 * max(1.0, np.max(np.abs(f["raw"]))), p._name
    X, mu, lo, up = model.predict(fx["pred_X"], transformed=True)
    assert relerr(mu, fx["pred_mu"]) < 1e-5 and relerr(lo, fx["pred_lower"]) < 1e-5 and relerr(up, fx["pred_upper"]) < 1e-5


def _synth_mosm(N, C, Q):
    X, y = synth.make_data(N, C)
    h = synth.mosm_hypers(C, Q)
    k = gpr.MultiOutputSpectralMixtureKernel(Q=Q, output_dims=C)
    for name in ("weight", "mean", "variance", "delay", "phase"):
        getattr(k, name).assign(h[name])
    m = gpr.Exact(k, X, y, variance=h["scale"] ** 2)
    m.likelihood.scale.assign(h["scale"])
    return m


@pytest.mark.parametrize("name", ["lml_synth2048.npz", "cfg2.npz"])
def test_full_size_goldens(name):
    """seed-generated inputs (mogptk_amd/synth.py), reference outputs stored: N=2048 and BASELINE.json configs[1]
    (MOSM C=4 Q=3 N=8192).  north_star tolerance: 1e-5 relative on LML and on each gradient tensor."""
    fx = load(name)
    C, Q, D, Rq, N = [int(v) for v in fx["meta"]]
    m = _synth_mosm(N, C, Q)
    fp = fixture_params(fx)
    for p, f in zip(m.parameters(), fp):                         # assign() round trip vs the reference's raw values
        assert np.max(np.abs(p.data - f["raw"])) < 1e-12 * max(1.0, np.max(np.abs(f["raw"])))
        p.data = np.array(f["raw"])
    loss = float(m.loss())
    assert abs(loss - float(fx["loss"])) < 1e-9 * abs(float(fx["loss"])), (loss, float(fx["loss"]))
    for p, f in zip(m.parameters(), fp):
        err = np.max(np.abs(p.grad - f["grad"])) / np.max(np.abs(f["grad"]))
        assert err < 1e-5, (p._name, err)


def test_full_size_properties_cfg2():
    """size-independent properties at N=8192: permutation invariance of LML/gradient, directional derivative by
    central differences, determinism across repeated evaluations."""
    C, Q, N = 4, 3, 8192
    m = _synth_mosm(N, C, Q)
    l0 = float(m.loss())
    g0 = [p.grad.copy() for p in m.parameters()]
    assert float(m.loss()) == l0                                   # bitwise repeatable
    for g, p in zip(g0, m.parameters()):
        assert np.array_equal(g, p.grad)
    # central difference along a random direction in raw space
    rng = np.random.default_rng(7)
    dirs = [rng.standard_normal(p.data.shape) for p in m.parameters()]
    eps = 1e-4
    base = [p.data.copy() for p in m.parameters()]
    vals = []
    for s in (+1, -1):
        for p, b, d in zip(m.parameters(), base, dirs):
            p.data = b + s * eps * d
        vals.append(-float(m.log_marginal_likelihood()))
    fd = (vals[0] - vals[1]) / (2 * eps)
    an = sum(float(np.sum(g * d)) for g, d in zip(g0, dirs))
    assert abs(fd - an) < 1e-5 * abs(an), (fd, an)
    # row permutation: same model, shuffled rows
    X, y = synth.make_data(N, C)
    perm = rng.permutation(N)
    h = synth.mosm_hypers(C, Q)
    k = gpr.MultiOutputSpectralMixtureKernel(Q=Q, output_dims=C)
    for name in ("weight", "mean", "variance", "delay", "phase"):
        getattr(k, name).assign(h[name])
    m2 = gpr.Exact(k, X[perm], y[perm], variance=h["scale"] ** 2)
    m2.likelihood.scale.assign(h["scale"])
    l2 = float(m2.loss())
    assert abs(l2 - l0) < 1e-10 * abs(l0)
    for g, p in zip(g0, m2.parameters()):
        assert np.max(np.abs(g - p.grad)) < 1e-8 * np.max(np.abs(g))


def test_train_lbfgs_on_device_matches_reference_trace():
    """Model.train('LBFGS') end to end on the device (reference model.py:541-553) against the trace recorded from the reference"""
    fx0, fx = load("adam_cfg1.npz"), load("lbfgs_cfg1.npz")
    data = mogptk_amd.Data(fx0["X"][:, 1], fx0["y"][:, 0], name="airline")
    model = mogptk_amd.SM(data, Q=3)
    load_raw(model.gpr.parameters(), fixture_params(fx0, "init_"))
    model.train("LBFGS", iters=int(fx["fixed_max_iter"]))
    ref = fx["fixed_losses"]
    assert model.iters == int(fx["fixed_iters"])
    assert relerr(model.losses[:15], ref[:15]) < 1e-7
    assert relerr(model.losses, ref) < 1e-3


def test_bnse_on_device_matches_reference():
    """BNSE end to end on the device: 60 Adam steps of an Exact + SpectralKernel fit, then W = L^-1 and alpha fetched
    from the library for the spectrum posterior; against the reference's spectra"""
    fx = load("bnse.npz")
    w, mu, var = mogptk_amd.BNSE(fx["x"].copy(), fx["y"], n=150, iters=60)
    assert relerr(mu, fx["mu"]) < 1e-5 and relerr(var, fx["var"]) < 1e-5
    w2, mu2, var2 = mogptk_amd.BNSE(fx["x"].copy(), fx["y"], y_err=fx["yerr"], max_freq=0.9, n=120, iters=40)
    assert relerr(mu2, fx["mu2"]) < 1e-5 and relerr(var2, fx["var2"]) < 1e-5


def test_cfg3_size_gradient_is_the_derivative_of_the_lml():
    """BASELINE.json configs[2] (MOSM C=8 Q=5 N=32768): next to the golden values of test_cfg3_golden, the size-independent property that the
    gradient the device returns is the derivative of the LML it returns -- central difference along a random direction in raw-parameter space
    (the reference cannot back-propagate at this size in the build container: its autograd working set exceeds the memory)."""
    C, Q, N = 8, 5, 32768
    X, y = synth.make_data(N, C)
    h = synth.mosm_hypers(C, Q)
    k = gpr.MultiOutputSpectralMixtureKernel(Q=Q, output_dims=C)
    for name in ("weight", "mean", "variance", "delay", "phase"):
        getattr(k, name).assign(h[name])
    m = gpr.Exact(k, X, y, variance=h["scale"] ** 2)
    m.likelihood.scale.assign(h["scale"])
    loss0 = float(m.loss())
    params = list(m.parameters())
    g = [p.grad.copy() for p in params]
    raw0 = [p.data.copy() for p in params]
    rng = np.random.default_rng(3)
    d = [rng.standard_normal(p.data.shape) for p in params]
    nrm = np.sqrt(sum(float(np.sum(v * v)) for v in d))
    d = [v / nrm for v in d]
    gd = sum(float(np.sum(a * b)) for a, b in zip(g, d))
    eps = 1e-4
    vals = []
    for sgn in (+1.0, -1.0):
        for p, r, v in zip(params, raw0, d):
            p.data = r + sgn * eps * v
        vals.append(-float(m.log_marginal_likelihood()))
    for p, r in zip(params, raw0):
        p.data = r
    fd = (vals[0] - vals[1]) / (2.0 * eps)
    assert np.isfinite(loss0) and abs(gd) > 1e-3 * np.sqrt(sum(float(np.sum(a * a)) for a in g)) * 1e-3
    assert abs(fd - gd) < 1e-5 * abs(gd), (fd, gd, loss0)


def test_cfg3_golden():
    """BASELINE.json configs[2] (MOSM C=8 Q=5 N=32768) against tests/golden/cfg3.npz (tests/golden/gen_cfg3.py): the REFERENCE's own forward LML at this
    size (its forward pass fits the build container under torch.no_grad(); 1e-9), all 208 raw gradients from the numpy oracle through Kj^-1
    (oracle/table_model.py:TableDeviceLean, whose LML agrees with the reference's to 3e-14 at this size; north_star tolerance 1e-5 per tensor), and the
    reference's own central difference of its LML along one raw-space direction -- a derivative no code of this repository took part in (1e-6)."""
    from helpers import cfg3_direction as direction
    fx = load("cfg3.npz")
    C, Q, D, Rq, N = [int(v) for v in fx["meta"]]
    m = _synth_mosm(N, C, Q)
    fp = fixture_params(fx)
    params = list(m.parameters())
    for p, f in zip(params, fp):
        assert np.max(np.abs(p.data - f["raw"])) < 1e-12 * max(1.0, np.max(np.abs(f["raw"])))
        p.data = np.array(f["raw"])
    loss = float(m.loss())
    lml_ref = float(fx["lml_ref"])
    assert abs(-loss - lml_ref) < 1e-9 * abs(lml_ref), (loss, lml_ref)
    assert abs(-loss - float(fx["lml_oracle"])) < 1e-9 * abs(lml_ref)
    for p, f in zip(params, fp):
        err = np.max(np.abs(p.grad - f["grad"])) / np.max(np.abs(f["grad"]))
        assert err < 1e-5, (p._name, err)
    d = direction([p.data.shape for p in params])
    gd = sum(float(np.sum(p.grad * v)) for p, v in zip(params, d))
    fd_ref = -float(fx["fd_ref"])                                  # the fixture differentiates the LML, the gradients are of the loss
    assert abs(gd - fd_ref) < 1e-6 * abs(fd_ref), (gd, fd_ref)


def test_cfg4_predict_golden():
    """BASELINE.json configs[3]: CSM C=4 Q=3 N=16384, predictive mean/variance at S=4096 (64 probe rows stored)."""
    fx = load("cfg4.npz")
    C, Q, D, Rq, N, S = [int(v) for v in fx["meta"]]
    X, y = synth.make_data(N, C)
    h = synth.csm_hypers(C, Q)
    k = gpr.MixtureKernel(gpr.CrossSpectralKernel(output_dims=C, input_dims=1, Rq=1), Q)
    for q in range(Q):
        k[q].amplitude.assign(h["amplitude"][q]); k[q].mean.assign(h["mean"][q])
        k[q].variance.assign(h["variance"][q]); k[q].shift.assign(h["shift"][q])
    m = gpr.Exact(k, X, y, variance=h["scale"] ** 2)
    m.likelihood.scale.assign(h["scale"])
    mu, var = m.predict_f(synth.test_inputs(S, C))
    probe = fx["probe"]
    assert np.max(np.abs(mu[probe, 0] - fx["mu"])) < 1e-6 * np.max(np.abs(fx["mu"]))
    assert np.max(np.abs(var[probe, 0] - fx["var"])) < 1e-6 * np.max(np.abs(fx["var"]))


def test_cholesky_failure_raises_reference_exception():
    X, y = synth.make_data(256, 2)
    k = gpr.MultiOutputSpectralMixtureKernel(Q=1, output_dims=2)
    k.weight.assign(np.full((2, 1), 50.0)); k.mean.assign(np.full((2, 1, 1), 1e-3)); k.variance.assign(np.full((2, 1, 1), 1e-6))
    m = gpr.Exact(k, X, y, variance=[1e-16, 1e-16], jitter=1e-15)
    m.likelihood.scale.assign([1e-8, 1e-8])
    with pytest.raises(mogptk_amd.CholeskyException):
        m.loss()
    # and the handle stays usable afterwards
    k.weight.assign(np.full((2, 1), 1.0)); k.variance.assign(np.full((2, 1, 1), 0.05)); m.likelihood.scale.assign([0.3, 0.3])
    assert np.isfinite(float(m.loss()))


def test_edge_cases():
    # one training point, one channel empty, N below / at / above a tile edge
    k = gpr.MultiOutputSpectralMixtureKernel(Q=1, output_dims=2)
    k.mean.assign(np.full((2, 1, 1), 0.1))
    for N in (1, 2, 127, 128, 129):
        X = np.stack([np.zeros(N), np.linspace(0, 5, N)], axis=1)        # channel 1 has no data
        y = np.sin(X[:, 1])
        dev = _lib.ExactHandle(0, X, y, 2)
        ref = TableDevice(0, X, y, 2)
        t = k._spectral_terms(1)
        dev.set_terms(t); ref.set_terms(t)
        a = dev.eval(np.array([0.1, 0.2]), 1e-8, grad=True)
        b = ref.eval(np.array([0.1, 0.2]), 1e-8, grad=True)
        assert abs(a["lml"] - b["lml"]) < 1e-10 * max(1.0, abs(b["lml"])), N
        assert np.max(np.abs(a["moments"] - b["moments"])) < 1e-9 * max(1.0, np.max(np.abs(b["moments"]))), N
    with pytest.raises(_lib.MogpError):
        _lib.ExactHandle(0, np.array([[2.0, 0.0]]), np.zeros(1), 2)     # channel id out of range


def _titsias_from_fixture(fx, pre):
    C, Q, D, Rq = [int(v) for v in fx[pre + "meta"]]
    fp = fixture_params(fx, pre)
    k = gpr.MultiOutputSpectralMixtureKernel(Q=Q, output_dims=C, input_dims=D)
    Zspec = fx[pre + "Zspec"]
    Zspec = int(Zspec[0]) if bool(fx[pre + "Zspec_is_int"]) else [int(z) for z in Zspec]
    m = gpr.Titsias(k, fx[pre + "X"], fx[pre + "y"], Z=Zspec, variance=float(fp[-1]["cons"]) ** 2, jitter=float(fx[pre + "jitter"]))
    load_raw(m.parameters(), fp)
    return m, fp


def test_titsias_matches_reference_fixtures():
    fx = load("titsias.npz")
    for n in range(int(fx["ncases"])):
        pre = "c%d_" % n
        m, fp = _titsias_from_fixture(fx, pre)
        assert abs(float(m.log_marginal_likelihood()) - float(fx[pre + "elbo"])) < 1e-9 * abs(float(fx[pre + "elbo"]))
        assert abs(float(m.loss()) - float(fx[pre + "loss"])) < 1e-9 * abs(float(fx[pre + "loss"]))
        for p, f in zip(m.parameters(), fp):
            if f["grad"] is None:
                assert p.grad is None
            else:
                assert np.max(np.abs(p.grad - f["grad"])) <= 1e-7 * max(1.0, np.max(np.abs(f["grad"]))), (n, p._name, p.grad, f["grad"])
        mu, var = m.predict_f(fx[pre + "Xs"])
        assert relerr(mu, fx[pre + "mu"]) < 1e-8 and np.max(np.abs(var - fx[pre + "var"])) < 1e-8


def test_titsias_device_raw_outputs_against_numpy_model():
    rng = np.random.default_rng(11)
    C, Q, N, M = 3, 2, 700, 150
    X, y = synth.make_data(N - N % C, C)
    X = X[rng.permutation(X.shape[0])]
    y = rng.standard_normal(X.shape[0])
    Z = np.concatenate([np.stack([np.full(M // C, float(c)), np.sort(rng.uniform(0, 100, M // C))], axis=1) for c in range(C)])
    Z = Z[rng.permutation(Z.shape[0])]
    h = synth.mosm_hypers(C, Q)
    k = gpr.MultiOutputSpectralMixtureKernel(Q=Q, output_dims=C)
    for name in ("weight", "mean", "variance", "delay", "phase"):
        getattr(k, name).assign(h[name])
    table = k._spectral_terms(1)
    kd = k._spectral_diag(1)
    dev = _lib.ExactHandle(0, X, y, C)
    ref = TableDevice(0, X, y, C)
    dev.set_terms(table); ref.set_terms(table)
    a = dev.titsias_eval(Z, 0.3, 1e-6, kd, grad=True)
    b = ref.titsias_eval(Z, 0.3, 1e-6, kd, grad=True)
    assert abs(a["elbo"] - b["elbo"]) < 1e-9 * abs(b["elbo"])
    for key in ("mom_uu", "mom_uf", "gZ"):
        assert np.max(np.abs(a[key] - b[key])) < 1e-7 * np.max(np.abs(b[key])), (key, np.max(np.abs(a[key] - b[key])) / np.max(np.abs(b[key])))
    assert abs(a["trGA"] - b["trGA"]) < 1e-7 * abs(b["trGA"]) and abs(a["dsigma"] - b["dsigma"]) < 1e-8 * abs(b["dsigma"])


def test_cfg5_titsias_golden():
    """BASELINE.json configs[4]: Titsias, MOSM C=4 Q=3, N=100000, M=2048 (= [512]*4, grid): ELBO and every gradient tensor
    against the reference (north_star tolerance 1e-5 relative)."""
    fx = load("cfg5.npz")
    C, Q, D, Rq, N, M = [int(v) for v in fx["meta"]]
    X, y = synth.make_data(N, C)
    h = synth.mosm_hypers(C, Q)
    k = gpr.MultiOutputSpectralMixtureKernel(Q=Q, output_dims=C)
    for name in ("weight", "mean", "variance", "delay", "phase"):
        getattr(k, name).assign(h[name])
    s = float(fx["scale"])
    m = gpr.Titsias(k, X, y, Z=[M // C] * C, variance=s ** 2)
    m.likelihood.scale.assign(s)
    fp = fixture_params(fx)
    for p, f in zip(m.parameters(), fp):
        assert np.max(np.abs(p.data - f["raw"])) < 1e-12 * max(1.0, np.max(np.abs(f["raw"]))), p._name
        p.data = np.array(f["raw"])
    loss = float(m.loss())
    print("ELBO against the reference's: %.2e relative" % (abs(loss - float(fx["loss"])) / abs(float(fx["loss"]))))
    assert abs(loss - float(fx["loss"])) < 1e-11 * abs(float(fx["loss"])), (loss, float(fx["loss"]))       # measured 1.3e-13 (2.9e-10 before the refined panels)
    for p, f in zip(m.parameters(), fp):
        err = np.max(np.abs(p.grad - f["grad"])) / np.max(np.abs(f["grad"]))
        if not p._name.endswith("induction_points"):
            print("    %-50s %.2e" % (p._name, err))
        if p._name.endswith("induction_points"):
            # dELBO/dZ is O(1e-2) here, the residue of O(1e4) terms cancelling through a K_uu with condition number ~1e11 (512 grid points
            # per channel, 0.2 apart): the reference's OWN value moves by 2.35e-3 of this tensor when only its thread count changes
            # (8 vs 3 torch threads, tests/golden/gen_golden.py docstring), so 1e-5 against one of its runs is below its noise floor.
            # With every K_uu^-1 a triangular solve (trsm.hip) the device sits at 3.6e-3 .. 4.8e-3 of the tensor, 1.5 - 2 x the
            # reference's own spread; an 80-bit evaluation at the same conditioning puts the solve formulation at 5e-5 of the truth and
            # the explicit-inverse one of round 1 at 12-19 % (tools/titsias_numerics.py).  The per-point accumulation has a fixed order
            # since round 3 (every tile's sums in a slot of their own, added by k_gz_reduce with compensated sums), so the value no
            # longer changes from run to run -- checked below -- and it is 6.23e-3 of the tensor against this one run of the reference;
            # compensating every sum of the accumulation (per thread, per tile, across tiles) does not move it in the fourth digit, so
            # what is left is the conditioning of the adjoints, common to both sides.  Asserted: within 3 x the reference's own
            # spread, and the direction to five nines.
            print("dELBO/dZ against the reference's recorded run: %.3e of the tensor" % err)
            assert err < 7e-3, (p._name, err)
            g, r = p.grad[:, 1], f["grad"][:, 1]
            assert np.dot(g, r) / (np.linalg.norm(g) * np.linalg.norm(r)) > 0.9999
            assert np.all(p.grad[:, 0] == 0.0)
        else:
            assert err < 1e-6, (p._name, err)          # measured <= 6.4e-8
    # bitwise repeatability of the sparse bound and of EVERY gradient, d/dZ included: no floating-point atomics on this path
    g1 = [p.grad.copy() for p in m.parameters()]
    loss2 = float(m.loss())
    assert loss2 == loss
    for p, g in zip(m.parameters(), g1):
        assert np.array_equal(p.grad, g), p._name


@pytest.mark.parametrize("path", ["sweep", "phases", "fused", "fused-streams"])
def test_every_gradient_schedule_matches_reference(path):
    """the schedules of the gradient evaluation (MOGP_GRAD_PATH: single-sweep inversion; POTRF -> TRTRI -> LAUUM, the default
    above 80 tile rows; inverse fused with the Cholesky chain, the default below -- as tile dataflow (forced on for every size here) and,
    "fused-streams", as streams of launches) against the same golden vectors, each in a fresh process because the schedule is chosen
    once per process"""
    import os, subprocess, sys, textwrap
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = textwrap.dedent('''
        import sys, numpy as np
        sys.path.insert(0, %r); sys.path.insert(0, %r)
        from helpers import load, product_exact, fixture_params
        from mogptk_amd import gpr, synth
        for name in ("mosm_c3q2", "mosm_c2q3_shuf", "csm_c3q2", "sm_c2q2_d2"):
            fx = load("lml_%%s.npz" %% name)
            m, fp = product_exact(fx)
            assert abs(float(m.loss()) - float(fx["loss"])) < 1e-9 * abs(float(fx["loss"])), name
            for p, f in zip(m.parameters(), fp):
                if f["grad"] is not None:
                    assert np.max(np.abs(p.grad - f["grad"])) <= 1e-7 * max(1.0, np.max(np.abs(f["grad"]))), (name, p._name)
        fx = load("cfg2.npz")
        C, Q, D, Rq, N = [int(v) for v in fx["meta"]]
        X, y = synth.make_data(N, C); h = synth.mosm_hypers(C, Q)
        k = gpr.MultiOutputSpectralMixtureKernel(Q=Q, output_dims=C)
        for n in ("weight", "mean", "variance", "delay", "phase"): getattr(k, n).assign(h[n])
        m = gpr.Exact(k, X, y, variance=h["scale"] ** 2); m.likelihood.scale.assign(h["scale"])
        fp = fixture_params(fx)
        for p, f in zip(m.parameters(), fp): p.data = np.array(f["raw"])
        assert abs(float(m.loss()) - float(fx["loss"])) < 1e-9 * abs(float(fx["loss"]))
        for p, f in zip(m.parameters(), fp):
            assert np.max(np.abs(p.grad - f["grad"])) / np.max(np.abs(f["grad"])) < 1e-5, p._name
        print("SWEEP_OK")
    ''') % (root, os.path.join(root, "tests"))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600,
                         env=dict(os.environ, MOGP_GRAD_PATH=path.split("-")[0], MOGP_FLOW="0" if path == "fused-streams" else "1", MOGP_FLOW_MIN="2"))
    assert "SWEEP_OK" in out.stdout, out.stdout[-2000:] + out.stderr[-3000:]


@pytest.mark.parametrize("path", ["phases", "fused"])
def test_gradient_with_only_the_needed_tiles_of_the_inverse(path):
    """a series much longer than the kernel's support (3000 points over [0, 900], spectral variances 0.5 .. 1.5: dK/dtheta is below e^-50
    of its peak beyond ~15): the evaluation forms only the tiles of Kj^-1 its gradient reads (mogp_model_inverse_fraction < 0.5), and
    moments, diagG, trG and the LML must equal -- bit for bit -- those of the same evaluation with every tile formed (MOGP_FULL_INVERSE=1),
    which in turn match the numpy restatement; mogp_model_fetch(1) afterwards still hands back the WHOLE inverse.  Both schedules that
    form W^T W (sweep inverts in place and has nothing to drop)."""
    import os, subprocess, sys, tempfile, textwrap
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = textwrap.dedent('''
        import sys, numpy as np
        sys.path.insert(0, %r); sys.path.insert(0, %r)
        from mogptk_amd import gpr, _lib
        from oracle.table_model import TableDevice
        N, C, Q = 3000, 2, 2
        rng = np.random.default_rng(5)
        sizes = [1400, 1600]
        X = np.concatenate([np.stack([np.full(s, float(c)), rng.uniform(0, 900, s)], axis=1) for c, s in enumerate(sizes)])
        X = X[np.argsort(X[:, 1])]                 # channels interleaved, each in the order of its series (the library sorts by channel, stably: a
        y = rng.standard_normal(N)                 # tile of 64 consecutive points of a channel is then local in x -- shuffled rows have no band)
        k = gpr.MultiOutputSpectralMixtureKernel(Q=Q, output_dims=C, input_dims=1)
        k.weight.assign(rng.uniform(0.5, 1.5, (C, Q))); k.mean.assign(rng.uniform(0.02, 0.4, (C, Q, 1)))
        k.variance.assign(rng.uniform(0.5, 1.5, (C, Q, 1))); k.delay.assign(rng.normal(0, 0.3, (C, Q, 1))); k.phase.assign(rng.normal(0, 0.3, (C, Q)))
        table = k._spectral_terms(1)
        noise = rng.uniform(0.05, 0.2, C)
        dev = _lib.ExactHandle(0, X, y, C)
        dev.set_terms(table)
        a = dev.eval(noise, 1e-8, grad=True)
        frac = dev.inverse_fraction()
        Kinv = dev.fetch(1)
        out = sys.argv[1]
        if out != "-":
            np.savez(out, lml=a["lml"], moments=a["moments"], diagG=a["diagG"], trG=a["trG"], frac=frac)
        else:
            ref = TableDevice(0, X, y, C); ref.set_terms(table)
            b = ref.eval(noise, 1e-8, grad=True)
            assert abs(a["lml"] - b["lml"]) < 1e-10 * abs(b["lml"])
            scale = np.max(np.abs(b["moments"]))
            assert np.max(np.abs(a["moments"] - b["moments"])) < 1e-9 * scale
            Kj, _ = ref._Kj(noise, 1e-8, None)
            Ki = np.linalg.inv(Kj)
            assert np.max(np.abs(Kinv - Ki)) < 1e-9 * np.max(np.abs(Ki))
            f = np.load(sys.argv[2])
            assert frac < 0.5 and float(f["frac"]) == 1.0, (frac, float(f["frac"]))
            assert a["lml"] == float(f["lml"]) and a["trG"] == float(f["trG"])
            assert np.array_equal(a["moments"], f["moments"]) and np.array_equal(a["diagG"], f["diagG"])
            print("PLAN_OK", frac)
    ''') % (root, os.path.join(root, "tests"))
    with tempfile.TemporaryDirectory() as tmp:
        full = os.path.join(tmp, "full.npz")
        r1 = subprocess.run([sys.executable, "-c", code, full], capture_output=True, text=True, timeout=600,
                            env=dict(os.environ, MOGP_GRAD_PATH=path, MOGP_FULL_INVERSE="1", MOGP_FLOW="0"))     # the same schedule for both runs: a
        # planned (partial) inverse takes the stream form of the fused schedule, and the dataflow form adds z^T z in a different order
        assert r1.returncode == 0, r1.stdout[-1500:] + r1.stderr[-3000:]
        r2 = subprocess.run([sys.executable, "-c", code, "-", full], capture_output=True, text=True, timeout=600,
                            env=dict({k_: v for k_, v in os.environ.items() if k_ != "MOGP_FULL_INVERSE"}, MOGP_GRAD_PATH=path))
        assert "PLAN_OK" in r2.stdout, r2.stdout[-1500:] + r2.stderr[-3000:]


def _check_owned_rows(o, world):
    """tools/shard_check.py's owned-rows case: a model whose first evaluation is sharded (physical memory under its own tile rows only), the sharded
    prediction on it, then a one-GPU evaluation of the same handle (the matrix made whole, the ownership masks gone)"""
    assert o["rel_loss"] < 1e-10 and o["rel_grad"] < 1e-7 and o["rel_predict"] < 1e-7, o
    assert o["rel_loss_one_gpu_after"] < 1e-10 and o["rel_grad_one_gpu_after"] < 1e-7, o
    assert o["backed_after_one_gpu_call"] == o["whole_bytes"], o
    assert o["backed_bytes"] <= o["whole_bytes"], o
    assert o["backed_after_sharded_predict"] == o["backed_bytes"], o          # the sharded prediction needs no more of the matrix than the evaluation
    if world >= 4:
        assert o["backed_bytes"] < o["whole_bytes"], o


@pytest.mark.parametrize("ranks", [2, 4, 8])
def test_sharded_eval_ranks_sharing_one_gpu(ranks):
    """the multi-GPU evaluation and prediction (mogp_exact_eval_sharded / mogp_exact_predict_sharded: owned Gram + moment tiles, the
    collectives issued inside the library): 2 / 4 / 8 ranks sharing this GPU -- RCCL refuses duplicate devices, so the library's
    communicator is the external one, its callbacks staging through the host over gloo -- must reproduce the single-process loss,
    raw-parameter gradients and predictive mean / variance; N = 3000 -> 24 tile rows, 6 pivot blocks, ragged last tile."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % ranks, "--master-addr", "127.0.0.1",
                          "--master-port", str(29630 + ranks), os.path.join(root, "tools", "shard_check.py"), "--points", "3000", "--backend", "gloo"],
                         capture_output=True, text=True, timeout=900, env=dict(os.environ, MASTER_ADDR="127.0.0.1"))
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-3000:]
    r = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert r["world"] == ranks
    assert r["rel_loss"] < 1e-10, r
    assert r["rel_grad"] < 1e-7, r          # tolerance: 1e-5 (north_star); measured ~1e-10
    assert r["rel_predict"] < 1e-7, r       # sharded prediction (every rank its own rows of Kj^-1: shares of the quadratic form, one all-reduce) against the one-GPU solve
    _check_owned_rows(r["owned_rows"], ranks)
    t = r["titsias"]                        # the sparse bound data-parallel (every rank holds every world-th point; sums over points all-reduced)
    assert t["rel_loss"] < 1e-10 and t["rel_grad"] < 1e-6 and t["rel_predict"] < 1e-7, t
    assert r["hensman"]["rel_loss"] < 1e-10 and r["hensman"]["rel_grad"] < 1e-6, r["hensman"]          # SparseHensman + Student-t, data-parallel
    assert r["snelson"]["rel_loss"] < 1e-10 and r["snelson"]["rel_grad"] < 1e-6 and r["snelson"]["rel_predict"] < 1e-7, r["snelson"]


@pytest.mark.parametrize("variant", ["one_message", "factor_once", "factor_once_one_message"])
def test_sharded_eval_exchange_variants(variant):
    """the switches of the sharded evaluation's exchange (round 5), four ranks sharing this GPU: MOGP_SHARD_SPLIT=0 (the whole panel of a pivot block in one
    all-gather on the critical stream, rounds 1-4; the default sends the pivot block's own rows first and the rest on a communication stream underneath the
    inversion) and MOGP_SHARD_FACTOR_ONCE=1 (the owner of the pivot block inverts it, an all-reduce carries the factor to the others) -- same results"""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    if "one_message" in variant:
        env["MOGP_SHARD_SPLIT"] = "0"
    if "factor_once" in variant:
        env["MOGP_SHARD_FACTOR_ONCE"] = "1"
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=4", "--master-addr", "127.0.0.1",
                          "--master-port", "29652", os.path.join(root, "tools", "shard_check.py"), "--points", "3000", "--backend", "gloo", "--exact-only"],
                         capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-3000:]
    r = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert r["world"] == 4 and r["rel_loss"] < 1e-10 and r["rel_grad"] < 1e-7 and r["rel_predict"] < 1e-7, r
    _check_owned_rows(r["owned_rows"], 4)


def test_sharded_owned_rows_allocation():
    """SURVEY.md 8e, block-cyclic ownership: a rank of a sharded evaluation holds ceil(T / P) tile rows of the work matrix and no second matrix.  N = 8192
    (64 tile rows of 8 MB, channels on tile boundaries), four ranks sharing this GPU: each has physical memory for exactly a quarter of the matrix
    (mogp_model_work_bytes; the rest of the address range is reserved but unmapped -- a kernel straying into another rank's rows would fault, so the run
    itself is the proof that nothing does), results as on one GPU; MOGP_SHARD_OWNED=0 keeps the replicated form (the whole matrix on every rank)."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for mode, port in (("1", 29661), ("0", 29662)):
        out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=4", "--master-addr", "127.0.0.1",
                              "--master-port", str(port), os.path.join(root, "tools", "shard_check.py"), "--points", "8192", "--backend", "gloo", "--exact-only",
                              "--reps", "1"], capture_output=True, text=True, timeout=900, env=dict(os.environ, MASTER_ADDR="127.0.0.1", MOGP_SHARD_OWNED=mode))
        assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-3000:]
        r = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
        assert r["world"] == 4 and r["rel_loss"] < 1e-10 and r["rel_grad"] < 1e-7 and r["rel_predict"] < 1e-7, r
        o = r["owned_rows"]
        _check_owned_rows(o, 4 if mode == "1" else 1)
        assert o["whole_bytes"] == 8 * 8192 * 8192, o
        assert o["backed_bytes"] == (o["whole_bytes"] // 4 if mode == "1" else o["whole_bytes"]), o


def test_sharded_stage_protocol_on_device():
    """the sharded evaluation stage by stage on the DEVICE (mogp_shard_config / begin / pack / unpack / block / alpha / finish, the collectives issued by the
    caller: mogptk_amd.dist.sharded_eval -- the sequence the numpy twin runs under gloo in tests/test_dist_cpu.py), two ranks sharing this GPU, in the owned-rows
    form: loss and gradients as on one GPU (the prediction of this mode is the one-GPU one)"""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                          "--master-port", "29664", os.path.join(root, "tools", "shard_check.py"), "--points", "3000", "--backend", "gloo", "--protocol"],
                         capture_output=True, text=True, timeout=900, env=dict(os.environ, MASTER_ADDR="127.0.0.1"))
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-3000:]
    r = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert r["world"] == 2 and r["rel_loss"] < 1e-10 and r["rel_grad"] < 1e-7, r
    _check_owned_rows(r["owned_rows"], 2)


def test_rccl_communicator_single_rank_owned_rows_form():
    """the owned-rows form of the sharded evaluation on the path only real multi-GPU ranks take -- the persistent chain kernel factoring the pivot block in the
    Schur workspace the exchange was unpacked into, RCCL's own all-gather delivering this rank's rows too -- forced onto a one-rank group (MOGP_SHARD_OWNED=2)"""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
                          "--master-port", "29663", os.path.join(root, "tools", "shard_check.py"), "--points", "3000", "--backend", "nccl", "--exact-only"],
                         capture_output=True, text=True, timeout=900, env=dict(os.environ, MASTER_ADDR="127.0.0.1", MOGP_SHARD_OWNED="2"))
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-3000:]
    r = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert r["transport"] == "rccl" and r["world"] == 1
    assert r["rel_loss"] < 1e-10 and r["rel_grad"] < 1e-7 and r["rel_predict"] < 1e-7, r
    _check_owned_rows(r["owned_rows"], 1)


def test_rccl_communicator_single_rank():
    """the library's own RCCL communicator (dlopen'ed librccl, unique id, ncclAllGather / ncclAllReduce on the library's streams) on a
    1-rank group: the sharded evaluation and prediction through it equal the one-GPU ones.  More ranks need more GPUs."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
                          "--master-port", "29641", os.path.join(root, "tools", "shard_check.py"), "--points", "2000", "--backend", "nccl"],
                         capture_output=True, text=True, timeout=900, env=dict(os.environ, MASTER_ADDR="127.0.0.1"))
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-3000:]
    r = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert r["transport"] == "rccl" and r["world"] == 1
    assert r["rel_loss"] < 1e-10 and r["rel_grad"] < 1e-7 and r["rel_predict"] < 1e-7, r
    assert r["titsias"]["rel_loss"] < 1e-10 and r["titsias"]["rel_grad"] < 1e-6 and r["titsias"]["rel_predict"] < 1e-7, r["titsias"]
    assert r["hensman"]["rel_loss"] < 1e-10 and r["hensman"]["rel_grad"] < 1e-6, r["hensman"]
    assert r["snelson"]["rel_loss"] < 1e-10 and r["snelson"]["rel_grad"] < 1e-6 and r["snelson"]["rel_predict"] < 1e-7, r["snelson"]


def test_dataflow_schedule_short_soak():
    """a short soak of the headline evaluation (tools/flow_soak.py in small: BASELINE.json configs[1], the SAME evaluation 400 times): every loss and every
    gradient with identical bits, every evaluation on the dataflow schedule, no time-out -- and the kernel on the 56 workgroups per XCD that round 6's 100 000-evaluation
    soak was run on (profiles/r6_flow_soak_100k.txt; with all 60 about one evaluation in 7000 stalls until a bounded wait gives up: profiles/r6_flow_stall.txt)"""
    import hashlib, os
    assert os.environ.get("MOGP_FLOW_DROP_CUS") in (None, "16"), "the soak describes the default grid"
    m = _synth_mosm(8192, 4, 3)
    seen = set()
    for _ in range(400):
        l = m.loss()
        seen.add(hashlib.sha1(np.float64(l).tobytes() + b"".join(np.ascontiguousarray(p.grad).tobytes() for p in m.parameters())).hexdigest())
    s = m._handle.schedule()
    assert len(seen) == 1, len(seen)
    assert s["dataflow"] and not s["dataflow_fell_back"] and s["dataflow_timeouts"] == 0, s


def test_rccl_two_ranks():
    """RCCL with REAL ranks, the moment the box has them: one process per GPU (up to 8), the library's own communicator over xGMI, the sharded
    evaluation + prediction (and the data-parallel sparse models) against each rank's own one-GPU result -- at N = 3000 with every model, and at
    configs[2] size (N = 32768, C = 8, Q = 5; the exact model only).  Skipped on a one-GPU box (the driver's GPU test box is one)."""
    import json, os, subprocess, sys
    from mogptk_amd import _lib
    ngpu = int(_lib.lib().mogp_device_count())
    if ngpu < 2:
        pytest.skip("needs two or more GPUs (this box has %d): RCCL between real ranks cannot run here" % ngpu)
    world = min(ngpu, 8)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

    def run(extra, port):
        out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % world, "--master-addr", "127.0.0.1",
                              "--master-port", str(port), os.path.join(root, "tools", "shard_check.py"), "--backend", "nccl"] + extra,
                             capture_output=True, text=True, timeout=1800, env=dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0"))
        assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-3000:]
        return json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])

    r = run(["--points", "3000"], 29643)
    assert r["transport"] == "rccl" and r["world"] == world and r["rccl_ranks"] == world, r
    assert r["rel_loss"] < 1e-10 and r["rel_grad"] < 1e-7 and r["rel_predict"] < 1e-7, r
    assert r["titsias"]["rel_loss"] < 1e-10 and r["titsias"]["rel_grad"] < 1e-6 and r["titsias"]["rel_predict"] < 1e-7, r["titsias"]
    assert r["hensman"]["rel_loss"] < 1e-10 and r["hensman"]["rel_grad"] < 1e-6, r["hensman"]
    assert r["snelson"]["rel_loss"] < 1e-10 and r["snelson"]["rel_grad"] < 1e-6 and r["snelson"]["rel_predict"] < 1e-7, r["snelson"]
    _check_owned_rows(r["owned_rows"], world)
    r = run(["--points", "32768", "--channels", "8", "--q", "5", "--exact-only", "--reps", "1"], 29644)
    assert r["transport"] == "rccl" and r["world"] == world and r["rccl_ranks"] == world, r
    assert r["rel_loss"] < 1e-10 and r["rel_grad"] < 1e-7 and r["rel_predict"] < 1e-7, r
    _check_owned_rows(r["owned_rows"], world)
    if 256 % world == 0:
        assert r["owned_rows"]["backed_bytes"] * world == r["owned_rows"]["whole_bytes"], r      # 256 tile rows of 32 MB: exactly 1 / world each


def test_sgd_adagrad_error_path_and_pegging_on_device():
    """SURVEY 8f-1 remainder on the device: SGD / AdaGrad traces, the per-iteration error= path (a device prediction per iteration)
    and pegged parameters, against the same reference recordings as the CPU suite"""
    from test_host_logic import check_opt_traces, check_error_path, check_pegged_parameters
    check_opt_traces(tol_loss=1e-8, tol_raw=1e-7)
    check_error_path(tol=1e-6)
    check_pegged_parameters()


def test_mohsm_predict_and_wrapper_on_device():
    """MOHSM end to end on the device: predict_f with the per-point diagonal, the wrapper's loss / gradient / Adam trace"""
    from test_host_logic import check_mohsm_predict_and_wrapper
    check_mohsm_predict_and_wrapper(tol_pred=1e-7, tol_loss=1e-9, tol_grad=1e-7, tol_trace=1e-7)


def test_titsias_with_enveloped_terms_on_device():
    """the sparse bound with MOHSM terms (rows of width 2 + 5 D) on the device against the reference's autograd: ELBO, every gradient incl.
    lengthscale / center and the inducing inputs (the envelope's share of d/dZ), predict_f with K_ss,diag per test point"""
    from test_host_logic import check_titsias_with_enveloped_terms
    check_titsias_with_enveloped_terms(tol_loss=1e-9, tol_grad=1e-6, tol_pred=1e-7)


def test_snelson_with_enveloped_terms_on_device():
    """the FITC model with MOHSM terms on the device against the reference's autograd (mogp_snelson_eval with rows of width 2 + 5 D: K_ff,diag
    per point in, dp/dK_ff,nn per point out)"""
    from test_host_logic import check_snelson_with_enveloped_terms
    check_snelson_with_enveloped_terms(tol_loss=1e-9, tol_grad=1e-6, tol_pred=1e-7)


def test_hensman_with_enveloped_terms_on_device():
    """SparseHensman / Hensman with MOHSM terms on the device against the reference's autograd (mogp_svgp_forward / _backward with rows of
    width 2 + 5 D)"""
    from test_host_logic import check_hensman_with_enveloped_terms
    check_hensman_with_enveloped_terms(tol_elbo=1e-9, tol_grad=1e-6, tol_pred=1e-7)


def test_single_precision_switch_on_device():
    from test_host_logic import check_single_precision_switch
    check_single_precision_switch()


def test_reference_checkpoints_on_device(tmp_path):
    """files written by the reference's Model.save() load without the reference and evaluate on the device like the
    reference evaluates them (loss, gradients, predictions incl. the transformer chain) -- Exact wrappers and a Titsias model"""
    pytest.importorskip("torch")
    from test_host_logic import check_reference_checkpoints
    check_reference_checkpoints(tmp_path, tol_loss=1e-8, tol_grad=1e-6, tol_pred=1e-6)


def test_snelson_on_device():
    """the Snelson (FITC) model on the device against the reference -- marginal likelihood, gradients of kernel / noise /
    inducing inputs (scalar and per-channel noise), predict_f"""
    from test_host_logic import check_snelson
    check_snelson(tol_lml=1e-9, tol_grad=1e-6, tol_pred=1e-7)


@pytest.mark.parametrize("N,M", [(699, 150), (1024, 300), (2048, 520)])
def test_snelson_device_raw_outputs_against_numpy_model(N, M):
    """the device against the numpy twin at sizes with several tile rows of inducing points, N both a multiple of 128 (beta rides in a panel
    of its own) and not (in the padding column), shuffled rows, per-channel noise; and the prediction"""
    rng = np.random.default_rng(N)
    C, Q = 3, 2
    X, _ = synth.make_data(N - N % C + (3 if N % C else 0), C)
    X = X[rng.permutation(X.shape[0])][:N]
    X = X[np.argsort(rng.random(N))]
    y = rng.standard_normal(N)
    Z = np.concatenate([np.stack([np.full(M // C, float(c)), np.sort(rng.uniform(0, 100, M // C))], axis=1) for c in range(C)])
    Z = Z[rng.permutation(Z.shape[0])]
    h = synth.mosm_hypers(C, Q)
    k = gpr.MultiOutputSpectralMixtureKernel(Q=Q, output_dims=C)
    for name in ("weight", "mean", "variance", "delay", "phase"):
        getattr(k, name).assign(h[name])
    table = k._spectral_terms(1)
    kd = k._spectral_diag(1)
    noise = rng.uniform(0.05, 0.3, C)
    dev = _lib.ExactHandle(0, X, y, C)
    ref = TableDevice(0, X, y, C)
    dev.set_terms(table); ref.set_terms(table)
    a = dev.snelson_eval(Z, noise, 1e-6, kd, grad=True)
    b = ref.snelson_eval(Z, noise, 1e-6, kd, grad=True)
    assert abs(a["lml"] - b["lml"]) < 1e-9 * abs(b["lml"])
    for key in ("mom_uu", "mom_uf", "gZ", "hsum"):
        assert np.max(np.abs(a[key] - b[key])) < 1e-6 * np.max(np.abs(b[key])), (key, np.max(np.abs(a[key] - b[key])) / np.max(np.abs(b[key])))
    assert abs(a["trGA"] - b["trGA"]) < 1e-6 * abs(b["trGA"])
    Xs = np.concatenate([np.stack([np.full(37, float(c)), np.linspace(0, 105, 37)], axis=1) for c in range(C)])
    mu, var = dev.snelson_predict(Z, noise, 1e-6, Xs, kd, kd)
    mu_r, var_r = ref.snelson_predict(Z, noise, 1e-6, Xs, kd, kd)
    assert relerr(mu, mu_r) < 1e-8 and np.max(np.abs(var - var_r)) < 1e-8


def test_hensman_on_device():
    """SparseHensman and the dense Hensman model (Gaussian likelihood) on the device against the reference -- ELBO, gradients of
    q_mu / q_sqrt / inducing inputs / kernel / noise scale, predict_f"""
    from test_host_logic import check_hensman
    check_hensman(tol_elbo=1e-9, tol_grad=1e-6, tol_pred=1e-7)


@pytest.mark.parametrize("N,M,dense", [(699, 150, False), (1024, 300, False), (2048, 520, False), (640, 640, True), (500, 500, True)])
def test_svgp_device_raw_outputs_against_numpy_model(N, M, dense):
    """the two device calls of the Hensman models against the numpy twin at sizes with several tile rows of inducing points, sparse and dense,
    with arbitrary dE/dmu, dE/dvar (the likelihood is the caller's); and the forward pass at test inputs"""
    rng = np.random.default_rng(N + M)
    C, Q = 3, 2
    X, _ = synth.make_data(N - N % C + (3 if N % C else 0), C)
    X = X[rng.permutation(X.shape[0])][:N]
    if dense:
        X = X[np.argsort(X[:, 0], kind="stable")]            # the whitened parameters depend on the order: inputs grouped by channel
    y = rng.standard_normal(N)
    if dense:
        Z = X.copy()
    else:
        Z = np.concatenate([np.stack([np.full(M // C, float(c)), rng.uniform(0, 100, M // C)], axis=1) for c in range(C)])
    M = Z.shape[0]
    h = synth.mosm_hypers(C, Q)
    k = gpr.MultiOutputSpectralMixtureKernel(Q=Q, output_dims=C)
    for name in ("weight", "mean", "variance", "delay", "phase"):
        getattr(k, name).assign(h[name])
    table = k._spectral_terms(1)
    kd = k._spectral_diag(1)
    q_mu = rng.normal(0, 0.5, M)
    q_sqrt = np.tril(rng.normal(0, 0.05, (M, M))) + np.diag(rng.uniform(0.5, 1.2, M)) + np.triu(rng.normal(0, 1.0, (M, M)), 1)
    jitter = 1e-4 if dense else 1e-6
    dev = _lib.ExactHandle(0, X, y, C)
    ref = TableDevice(0, X, y, C)
    dev.set_terms(table); ref.set_terms(table)
    a = dev.svgp_forward(Z, q_mu, q_sqrt, jitter, kd, dense=dense)
    b = ref.svgp_forward(Z, q_mu, q_sqrt, jitter, kd, dense=dense)
    assert relerr(a["mu"], b["mu"]) < 1e-8 and relerr(a["var"], b["var"]) < 1e-8
    e, f = rng.standard_normal(N), -rng.uniform(0.5, 2.0, N)
    ga = dev.svgp_backward(e, f)
    gb = ref.svgp_backward(e, f)
    for key in ("mom_uu", "mom_uf", "gZ", "g_qmu", "g_qsqrt"):
        scale = np.max(np.abs(gb[key]))
        if scale == 0.0:
            assert np.max(np.abs(ga[key])) == 0.0, key
        else:
            got = np.tril(ga[key]) if key == "g_qsqrt" else ga[key]
            assert np.max(np.abs(got - gb[key])) < 1e-6 * scale, (key, np.max(np.abs(got - gb[key])) / scale)
    assert abs(ga["trGA"] - gb["trGA"]) < 1e-6 * abs(gb["trGA"])
    Xs = np.concatenate([np.stack([np.full(37, float(c)), np.linspace(0, 105, 37)], axis=1) for c in range(C)])
    p1 = dev.svgp_forward(Z, q_mu, q_sqrt, jitter, kd, Xs=Xs, kss_diag=kd, dense=dense)
    p2 = ref.svgp_forward(Z, q_mu, q_sqrt, jitter, kd, Xs=Xs, kss_diag=kd, dense=dense)
    assert relerr(p1["mu"], p2["mu"]) < 1e-7 and np.max(np.abs(p1["var"] - p2["var"])) < 1e-7 * max(1.0, np.max(np.abs(p2["var"])))


def test_opper_archambeau_on_device():
    """the Opper-Archambeau model (Gaussian likelihood) on the device against the reference -- ELBO, gradients of q_nu / q_lambda /
    kernel / noise scale, predict_f (diagonal and full)"""
    from test_host_logic import check_oa
    check_oa(tol_elbo=1e-9, tol_grad=1e-6, tol_pred=1e-7)


@pytest.mark.parametrize("N", [700, 1500, 2304])
def test_oa_device_raw_outputs_against_numpy_model(N):
    """the device calls of the Opper-Archambeau model against the numpy twin at sizes with several tile rows and outer blocks, inputs NOT grouped
    by channel, with arbitrary dE/dmu, dE/dvar (the likelihood is the caller's); and the prediction at test inputs"""
    rng = np.random.default_rng(N)
    C, Q = 3, 2
    X, _ = synth.make_data(N - N % C + (3 if N % C else 0), C)
    X = X[rng.permutation(X.shape[0])][:N]
    y = rng.standard_normal(N)
    h = synth.mosm_hypers(C, Q)
    k = gpr.MultiOutputSpectralMixtureKernel(Q=Q, output_dims=C)
    for name in ("weight", "mean", "variance", "delay", "phase"):
        getattr(k, name).assign(h[name])
    table = k._spectral_terms(1)
    kd = k._spectral_diag(1)
    nu, lam = rng.normal(0, 0.5, N), rng.uniform(0.5, 3.0, N)
    dev = _lib.ExactHandle(0, X, y, C)
    ref = TableDevice(0, X, y, C)
    dev.set_terms(table); ref.set_terms(table)
    a = dev.oa_forward(nu, lam)
    b = ref.oa_forward(nu, lam)
    assert relerr(a["mu"], b["mu"]) < 1e-9 and relerr(a["var"], b["var"]) < 1e-8 and abs(a["kl"] - b["kl"]) < 1e-9 * abs(b["kl"])
    e, f = rng.standard_normal(N), -rng.uniform(0.5, 2.0, N)
    ga = dev.oa_backward(e, f)
    gb = ref.oa_backward(e, f)
    for key in ("mom", "g_nu", "g_lambda"):
        scale = np.max(np.abs(gb[key]))
        assert np.max(np.abs(ga[key] - gb[key])) < 1e-7 * scale, (key, np.max(np.abs(ga[key] - gb[key])) / scale)
    with pytest.raises(_lib.MogpError):                        # the forward pass's state is consumed
        dev.oa_backward(e, f)
    Xs = np.concatenate([np.stack([np.full(37, float(c)), np.linspace(0, 105, 37)], axis=1) for c in range(C)])[rng.permutation(37 * C)]
    mu1, v1 = dev.oa_predict(nu, lam, kd, Xs)
    mu2, v2 = ref.oa_predict(nu, lam, kd, Xs)
    assert relerr(mu1, mu2) < 1e-9 and np.max(np.abs(v1 - v2)) < 1e-8 * max(1.0, np.max(np.abs(v2)))
    mu1, c1 = dev.oa_predict(nu, lam, kd, Xs, full=True)
    _, c2 = ref.oa_predict(nu, lam, kd, Xs, full=True)
    assert relerr(mu1, mu2) < 1e-9 and np.max(np.abs(c1 - c2)) < 1e-8 * max(1.0, np.max(np.abs(c2)))
    ye = rng.standard_normal(N)                                 # the exact model's own predictions are untouched by the shared code path
    dev.set_y(ye); ref.set_y(ye)
    nv = np.full(C, 0.3)
    m1, s1 = dev.predict(nv, 1e-8, kd, Xs)
    m2, s2 = ref.predict(nv, 1e-8, kd, Xs)
    assert relerr(m1, m2) < 1e-8 and np.max(np.abs(s1 - s2)) < 1e-8


def test_variational_models_with_non_gaussian_likelihoods_on_device():
    """SparseHensman / Hensman / OpperArchambeau with Student-t, Poisson + Gaussian per channel, Bernoulli, Laplace, Gamma
    likelihoods -- the device algebra around the host's likelihood, against the reference's loss and autograd gradients"""
    from test_host_logic import check_likelihood_models
    check_likelihood_models(tol_loss=1e-9, tol_grad=1e-6, tol_pred=1e-7)


def test_sparse_and_variational_edge_cases():
    """the models added for SURVEY 8f-4 at the edges: one training point, an empty channel, N below / at / above a tile edge, more inducing
    points than data points -- device against the numpy twin"""
    rng = np.random.default_rng(3)
    k = gpr.MultiOutputSpectralMixtureKernel(Q=1, output_dims=2)
    k.mean.assign(np.full((2, 1, 1), 0.1))
    t = k._spectral_terms(1)
    kd = k._spectral_diag(1)
    Xs = np.stack([np.array([0.0, 1.0, 0.0]), np.array([0.5, 2.5, 4.0])], axis=1)       # test points also in the channel without data
    for N in (1, 2, 127, 128, 129):
        X = np.stack([np.zeros(N), np.linspace(0, 5, N)], axis=1)                        # channel 1 has no data
        y = np.sin(X[:, 1])
        dev = _lib.ExactHandle(0, X, y, 2)
        ref = TableDevice(0, X, y, 2)
        dev.set_terms(t); ref.set_terms(t)
        # Opper-Archambeau
        nu, lam = rng.normal(0, 0.5, N), rng.uniform(0.5, 2.0, N)
        a, b = dev.oa_forward(nu, lam), ref.oa_forward(nu, lam)
        assert relerr(a["mu"], b["mu"]) < 1e-10 and np.max(np.abs(a["var"] - b["var"])) < 1e-10 and abs(a["kl"] - b["kl"]) < 1e-9 * max(1.0, abs(b["kl"])), N
        e, f = rng.standard_normal(N), -rng.uniform(0.5, 2.0, N)
        ga, gb = dev.oa_backward(e, f), ref.oa_backward(e, f)
        for key in ("mom", "g_nu", "g_lambda"):
            assert np.max(np.abs(ga[key] - gb[key])) < 1e-8 * max(1.0, np.max(np.abs(gb[key]))), (N, key)
        m1, v1 = dev.oa_predict(nu, lam, kd, Xs)
        m2, v2 = ref.oa_predict(nu, lam, kd, Xs)
        assert np.max(np.abs(m1 - m2)) < 1e-10 and np.max(np.abs(v1 - v2)) < 1e-10, N
        # sparse models: 3 inducing points in channel 0, 2 in the channel without data
        Z = np.array([[0.0, 0.3], [0.0, 2.2], [0.0, 4.1], [1.0, 1.0], [1.0, 3.0]])
        a, b = dev.titsias_eval(Z, 0.3, 1e-6, kd), ref.titsias_eval(Z, 0.3, 1e-6, kd)
        assert abs(a["elbo"] - b["elbo"]) < 1e-9 * max(1.0, abs(b["elbo"])), N
        for key in ("mom_uu", "mom_uf", "gZ"):
            assert np.max(np.abs(a[key] - b[key])) < 1e-7 * max(1.0, np.max(np.abs(b[key]))), (N, key)
        a, b = dev.snelson_eval(Z, np.array([0.1, 0.2]), 1e-6, kd), ref.snelson_eval(Z, np.array([0.1, 0.2]), 1e-6, kd)
        assert abs(a["lml"] - b["lml"]) < 1e-9 * max(1.0, abs(b["lml"])), N
        for key in ("mom_uu", "mom_uf", "gZ"):
            assert np.max(np.abs(a[key] - b[key])) < 1e-7 * max(1.0, np.max(np.abs(b[key]))), (N, key)
        q_mu, q_sqrt = rng.normal(0, 0.5, 5), np.tril(rng.normal(0, 0.2, (5, 5))) + np.eye(5)
        a, b = dev.svgp_forward(Z, q_mu, q_sqrt, 1e-6, kd), ref.svgp_forward(Z, q_mu, q_sqrt, 1e-6, kd)
        assert np.max(np.abs(a["mu"] - b["mu"])) < 1e-9 and np.max(np.abs(a["var"] - b["var"])) < 1e-9, N
        ga, gb = dev.svgp_backward(e, f), ref.svgp_backward(e, f)
        for key in ("mom_uu", "mom_uf", "gZ", "g_qmu"):
            assert np.max(np.abs(ga[key] - gb[key])) < 1e-7 * max(1.0, np.max(np.abs(gb[key]))), (N, key)
        assert np.max(np.abs(np.tril(ga["g_qsqrt"]) - gb["g_qsqrt"])) < 1e-7 * max(1.0, np.max(np.abs(gb["g_qsqrt"]))), N


def test_posterior_samples_on_device():
    """sample_f / Model.sample: the posterior's full covariance from the device, the draws from torch's generator like the reference --
    the reference's own samples under the same seed"""
    from test_host_logic import check_samples
    check_samples(tol=1e-6)


def test_sparse_models_full_covariance_on_device():
    """predict_f(full=True) of Titsias / SparseHensman / Hensman (mogp_sparse_predict_cov) and posterior samples drawn from it"""
    from test_host_logic import check_sparse_cov
    check_sparse_cov(tol=1e-7, tol_sample=1e-5)


def test_side_stream_schedule_equals_the_serial_one(tmp_path):
    """the sparse models' M x M chains, K_uf and v y run on a side stream underneath the large products (titsias.hip:side_fork); with
    MOGP_SIDE_STREAM=0 everything is enqueued on one stream -- same kernels, same arithmetic: the results must agree to rounding of the
    atomically accumulated d/dZ (a missing dependency between the streams would show up here)"""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "side.py"
    script.write_text('''
import sys
sys.path.insert(0, %r)
import numpy as np
from mogptk_amd import gpr, synth, _lib
C, Q, N, M = 3, 2, 30000, 640
X, y = synth.make_data(N, C)
h = synth.mosm_hypers(C, Q)
k = gpr.MultiOutputSpectralMixtureKernel(Q=Q, output_dims=C)
for name in ("weight", "mean", "variance", "delay", "phase"):
    getattr(k, name).assign(h[name])
rng = np.random.default_rng(2)
Z = np.concatenate([np.stack([np.full(M // C, float(c)), np.sort(rng.uniform(0, 100, M // C))], axis=1) for c in range(C)])
dev = _lib.ExactHandle(0, X, y, C); dev.set_terms(k._spectral_terms(1))
kd = k._spectral_diag(1)
out = {}
for rep in range(3):
    a = dev.titsias_eval(Z, 0.3, 1e-6, kd)
    b = dev.snelson_eval(Z, np.full(C, 0.09), 1e-6, kd)
    q_mu, q_sqrt = rng.normal(0, 0.3, Z.shape[0]), np.eye(Z.shape[0]) * 0.8
    f = dev.svgp_forward(Z, q_mu, q_sqrt, 1e-6, kd)
    c = dev.svgp_backward(np.cos(np.arange(N)), -np.ones(N))
    rng = np.random.default_rng(2)
out = dict(t_elbo=a["elbo"], t_uu=a["mom_uu"], t_uf=a["mom_uf"], t_gz=a["gZ"], s_lml=b["lml"], s_uu=b["mom_uu"], s_uf=b["mom_uf"], s_gz=b["gZ"],
           h_mu=f["mu"], h_uu=c["mom_uu"], h_uf=c["mom_uf"], h_gz=c["gZ"], h_gs=c["g_qsqrt"])
np.savez(sys.argv[1], **out)
''' % root)
    res = []
    for mode in ("0", "1"):
        path = str(tmp_path / ("out%s.npz" % mode))
        out = subprocess.run([sys.executable, str(script), path], capture_output=True, text=True, timeout=600, env=dict(os.environ, MOGP_SIDE_STREAM=mode))
        assert out.returncode == 0, out.stdout[-1000:] + out.stderr[-2000:]
        res.append(np.load(path))
    for key in res[0].files:
        a, b = res[0][key], res[1][key]
        assert np.max(np.abs(a - b)) <= 1e-9 * max(1.0, np.max(np.abs(a))), (key, float(np.max(np.abs(a - b))))      # measured: <= 2e-11 (atomics)


def test_dataflow_schedule_ran_and_equals_the_stream_schedule():
    """BASELINE.json configs[1] in ONE process: the default gradient evaluation runs as tile dataflow (csrc/flow.hip) on the persistent chain
    kernels -- asserted through mogp_model_schedule, so that a silent fallback cannot pass -- and gives the SAME W = L^-1 and Kj^-1, bit for bit,
    as the stream schedule of the same tile products (MOGP_FLOW is read per evaluation); loss and gradient agree to rounding (the dataflow
    kernel adds z^T z per tile row)."""
    import os
    C, Q, N = 4, 3, 8192
    old = {k: os.environ.get(k) for k in ("MOGP_FLOW", "MOGP_FLOW_MIN")}
    try:
        os.environ.pop("MOGP_FLOW", None); os.environ.pop("MOGP_FLOW_MIN", None)
        m = _synth_mosm(N, C, Q)
        l1 = float(m.loss())
        g1 = [p.grad.copy() for p in m.parameters()]
        hd = m._handle
        s = hd.schedule()
        assert s["dataflow"] and s["chain_kernel"] and not s["dataflow_fell_back"] and not s["chain_fell_back"], s
        W1, K1 = hd.fetch(0), hd.fetch(1)
        d = hd.flow_diag()                                             # (round 6) the schedule's deep-look counters are there (a deep look needs a 2 ms wait: none in a healthy evaluation)
        assert set(d) >= {"deep_looks", "stale_heads", "stale_counters", "deep_polls", "stale_polls"} and d["deep_looks"] < 1000, d
        assert float(m.loss()) == l1                                   # bitwise repeatable although the tiles run in a different order every time
        for g, p in zip(g1, m.parameters()):
            assert np.array_equal(g, p.grad)
        os.environ["MOGP_FLOW"] = "0"
        l0 = float(m.loss())
        s0 = hd.schedule()
        assert not s0["dataflow"] and s0["chain_kernel"], s0
        W0, K0 = hd.fetch(0), hd.fetch(1)
        assert np.array_equal(W0, W1) and np.array_equal(K0, K1)
        assert abs(l0 - l1) <= 1e-13 * abs(l0)
        for g, p in zip(g1, m.parameters()):
            assert np.max(np.abs(g - p.grad)) <= 1e-11 * max(1.0, np.max(np.abs(g)))
        # a size whose last outer block is partial (14 tile rows: blocks of 4, 4, 4, 2), dataflow forced on
        os.environ["MOGP_FLOW"] = "1"; os.environ["MOGP_FLOW_MIN"] = "2"
        m2 = _synth_mosm(1700, 2, 2)
        la = float(m2.loss()); Wa = m2._handle.fetch(0)
        assert m2._handle.schedule()["dataflow"]
        os.environ["MOGP_FLOW"] = "0"
        lb = float(m2.loss()); Wb = m2._handle.fetch(0)
        assert not m2._handle.schedule()["dataflow"]
        assert np.array_equal(Wa, Wb) and abs(la - lb) <= 1e-13 * abs(la)
        # 112 tile rows, the largest size the dataflow schedule is the default for: more deadline queues than the kernel has (the farthest are folded)
        os.environ.pop("MOGP_FLOW", None); os.environ.pop("MOGP_FLOW_MIN", None)
        m3 = _synth_mosm(14336, 4, 3)                                  # configs[1]'s kernel: its support is the whole series, every tile of the inverse is formed
        lc = float(m3.loss()); Wc = m3._handle.fetch(0)
        s3 = m3._handle.schedule()
        assert s3["dataflow"] and not s3["dataflow_fell_back"], s3
        g3 = [p.grad.copy() for p in m3.parameters()]
        os.environ["MOGP_FLOW"] = "0"                                  # above 80 tile rows that means POTRF / TRTRI / LAUUM: other summation orders
        ld = float(m3.loss()); Wd = m3._handle.fetch(0)
        assert not m3._handle.schedule()["dataflow"]
        assert np.max(np.abs(Wc - Wd)) <= 1e-9 * np.max(np.abs(Wd)) and abs(lc - ld) <= 1e-11 * abs(lc)
        for g, p in zip(g3, m3.parameters()):
            assert np.max(np.abs(g - p.grad)) <= 1e-8 * max(1.0, np.max(np.abs(g)))
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def test_first_evaluation_after_allocation_in_concurrent_processes(tmp_path):
    """Eight fresh processes share the GPU, each allocates its workspaces and evaluates ONCE (exact model with the dataflow schedule, sparse bound):
    every result is finite and equal to the single-process value.  Two hazards live here: the zero fills behind a first allocation used to race
    with the first Gram kernel on a shared GPU (round 3: garbage in the first ELBO of 15-30 % of such runs), and the resident kernels of
    several processes compete for the same reserved CUs -- a hand-off that times out must end in the stream schedule with the same numbers."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "first.py"
    script.write_text('''
import sys
sys.path.insert(0, %r)
import numpy as np
from mogptk_amd import gpr, synth, _lib
C, Q, N, M = 3, 2, 3072, 384
X, y = synth.make_data(N, C)
h = synth.mosm_hypers(C, Q)
k = gpr.MultiOutputSpectralMixtureKernel(Q=Q, output_dims=C)
for name in ("weight", "mean", "variance", "delay", "phase"):
    getattr(k, name).assign(h[name])
dev = _lib.ExactHandle(0, X, y, C); dev.set_terms(k._spectral_terms(1))
a = dev.eval(h["scale"] ** 2, 1e-8)
rng = np.random.default_rng(2)
Z = np.concatenate([np.stack([np.full(M // C, float(c)), np.sort(rng.uniform(0, 100, M // C))], axis=1) for c in range(C)])
dev2 = _lib.ExactHandle(0, X, y, C); dev2.set_terms(k._spectral_terms(1))
b = dev2.titsias_eval(Z, 0.3, 1e-6, k._spectral_diag(1))
s = dev.schedule()
np.savez(sys.argv[1], lml=a["lml"], mom=a["moments"], elbo=b["elbo"], gz=b["gZ"], fell=int(s["dataflow_fell_back"]) + 2 * int(s["chain_fell_back"]))
''' % root)
    def launch(i):
        return subprocess.Popen([sys.executable, str(script), str(tmp_path / ("o%d.npz" % i))], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                                env=dict(os.environ, MOGP_FLOW_MIN="2"))          # 24 tile rows: below the default threshold of the dataflow form
    p = launch(0)
    out, err = p.communicate(timeout=600)
    assert p.returncode == 0, out[-1000:] + err[-2000:]
    ref = np.load(tmp_path / "o0.npz")
    assert int(ref["fell"]) == 0                                          # alone on the GPU nothing falls back
    procs = [launch(i) for i in range(1, 9)]
    for i, p in enumerate(procs, 1):
        out, err = p.communicate(timeout=900)
        assert p.returncode == 0, out[-1000:] + err[-2000:]
        r = np.load(tmp_path / ("o%d.npz" % i))
        assert np.isfinite(r["lml"]) and np.isfinite(r["elbo"])
        assert abs(float(r["lml"]) - float(ref["lml"])) <= 1e-12 * abs(float(ref["lml"])), (i, float(r["lml"]), float(ref["lml"]), int(r["fell"]))
        assert np.max(np.abs(r["mom"] - ref["mom"])) <= 1e-9 * np.max(np.abs(ref["mom"]))
        assert float(r["elbo"]) == float(ref["elbo"]) and np.array_equal(r["gz"], ref["gz"])


def test_prediction_as_dataflow_equals_the_stream_form():
    """predict_f with the factorisation AND the forward substitution as one tile-dataflow schedule (csrc/flow.hip, the prediction's plan) against the
    stream form of the same tile products, in one process (MOGP_FLOW_PREDICT is read per call), at a size below the default threshold too; the
    schedule is asserted through mogp_model_schedule.  BASELINE.json configs[3] itself is pinned on the reference's golden by test_cfg4_predict_golden."""
    import os
    old = os.environ.get("MOGP_FLOW_PREDICT")
    try:
        for N, S, C, Q in ((6400, 1500, 4, 3), (2304, 700, 3, 2)):              # 50 and 18 tile rows; S not a multiple of 128
            X, y = synth.make_data(N, C)
            h = synth.csm_hypers(C, Q)
            k = gpr.MixtureKernel(gpr.CrossSpectralKernel(output_dims=C, input_dims=1, Rq=1), Q)
            for q in range(Q):
                k[q].amplitude.assign(h["amplitude"][q]); k[q].mean.assign(h["mean"][q])
                k[q].variance.assign(h["variance"][q]); k[q].shift.assign(h["shift"][q])
            m = gpr.Exact(k, X, y, variance=h["scale"] ** 2)
            m.likelihood.scale.assign(h["scale"])
            Xs = synth.test_inputs(S, C)
            os.environ["MOGP_FLOW_PREDICT"] = "8:200"
            l_before = float(m.loss())                                         # a gradient evaluation (its own dataflow plan) before and after:
            g_before = [p.grad.copy() for p in m.parameters()]                 # the two plans share the handle's counters and buffers
            mu1, var1 = m.predict_f(Xs)
            s1 = m._handle.schedule()
            assert s1["dataflow"] and not s1["dataflow_fell_back"], s1
            assert float(m.loss()) == l_before
            for g, p in zip(g_before, m.parameters()):
                assert np.array_equal(g, p.grad)
            mu1b, var1b = m.predict_f(Xs)
            assert np.array_equal(mu1, mu1b) and np.array_equal(var1, var1b)     # the tiles run in a different order every time: same bits
            os.environ["MOGP_FLOW_PREDICT"] = "0"
            mu0, var0 = m.predict_f(Xs)
            assert not m._handle.schedule()["dataflow"]
            assert np.max(np.abs(mu1 - mu0)) <= 1e-9 * max(1.0, np.max(np.abs(mu0)))
            assert np.max(np.abs(var1 - var0)) <= 1e-9 * max(1.0, np.max(np.abs(var0)))
            cov1 = m.predict_f(Xs[:300], full=True)[1] if N < 3000 else None      # the full-covariance branch reads the same X
            if cov1 is not None:
                os.environ["MOGP_FLOW_PREDICT"] = "8:200"
                cov2 = m.predict_f(Xs[:300], full=True)[1]
                assert np.max(np.abs(cov1 - cov2)) <= 1e-9 * np.max(np.abs(cov1))
    finally:
        if old is None:
            os.environ.pop("MOGP_FLOW_PREDICT", None)
        else:
            os.environ["MOGP_FLOW_PREDICT"] = old


def test_first_evaluation_of_a_fresh_process_above_8192_points_runs_as_dataflow(tmp_path):
    """A fresh process, a model with more than 8192 points, ONE evaluation: it must run as tile dataflow and must not have fallen back.  Until round 4
    every such first evaluation timed out and the model stayed on the stream schedule for good (17.2 instead of 14.5 ms at N = 9216): the z^T z parts
    of N > 8192 points are more than 16 KB, and the process's first device-to-host copy of that size, enqueued while the co-operating kernels ran,
    stalled them (mogp_ctx_create now makes such a copy first).  Every other test and the benchmark sit at N <= 8192 or evaluate a smaller model first."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "fresh.py"
    script.write_text('''
import sys, json
sys.path.insert(0, %r)
from mogptk_amd import gpr, synth
C, Q, N = 4, 3, 9216
X, y = synth.make_data(N, C)
h = synth.mosm_hypers(C, Q)
k = gpr.MultiOutputSpectralMixtureKernel(Q=Q, output_dims=C)
for name in ("weight", "mean", "variance", "delay", "phase"):
    getattr(k, name).assign(h[name])
m = gpr.Exact(k, X, y, variance=h["scale"] ** 2)
m.likelihood.scale.assign(h["scale"])
l1 = float(m.loss())
s = m._handle.schedule()
l2 = float(m.loss())
print(json.dumps(dict(s, same=bool(l1 == l2))))
''' % root)
    env = {k: v for k, v in os.environ.items() if not k.startswith("MOGP_")}
    p = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0, p.stdout[-1000:] + p.stderr[-2000:]
    import json
    s = json.loads(p.stdout.strip().splitlines()[-1])
    assert s["dataflow"] and s["chain_kernel"] and not s["dataflow_fell_back"] and not s["chain_fell_back"] and s["same"], (s, p.stderr[-500:])


@pytest.mark.parametrize("fixture", ["titsias_dz_truth.npz", "titsias_dz_truth_cfg5.npz"])
def test_titsias_inducing_gradient_against_extended_precision_truth(fixture):
    """dELBO/dZ at the conditioning of BASELINE.json configs[4] (M = 2048 grid inducing points 0.2 apart, cond K_uu ~ 1e11) against the 80-bit
    evaluation of the same function (tests/golden/gen_titsias_truth.py: Gram matrices, Cholesky, solves, adjoints and kernel derivative all in
    numpy.longdouble on the fp64 inputs the device receives) -- at N = 20 000 (round 4) and AT configs[4] itself, N = 100 000 (round 5, the
    column-chunked generator).  One fp64 run of the reference is a few 1e-3 of the tensor away from that truth (its thread-count spread at
    configs[4] is 2.35e-3): the device must not be further away than the reference is -- the comparison with one noisy reference run that
    test_cfg5_titsias_golden has to make says nothing about who is right."""
    fx = load(fixture)
    C, Q, D, Rq, N, M = [int(v) for v in fx["meta"]]
    X, y = synth.make_data(N, C)
    h = synth.mosm_hypers(C, Q)
    k = gpr.MultiOutputSpectralMixtureKernel(Q=Q, output_dims=C)
    for name in ("weight", "mean", "variance", "delay", "phase"):
        getattr(k, name).assign(h[name])
    s = float(fx["scale"])
    m = gpr.Titsias(k, X, y, Z=[M // C] * C, variance=s ** 2)
    m.likelihood.scale.assign(s)
    for p, f in zip(m.parameters(), fixture_params(fx)):
        p.data = np.array(f["raw"])
    loss = float(m.loss())
    assert abs(-loss - float(fx["elbo_truth"])) < 1e-9 * abs(float(fx["elbo_truth"]))
    zp = [p for p in m.parameters() if p._name.endswith("induction_points")][0]
    gz, truth = -zp.grad[:, 1], fx["gz_truth"]
    scale = np.max(np.abs(truth))
    err_dev = float(np.max(np.abs(gz - truth)) / scale)
    err_ref = float(np.max(np.abs(fx["gz_ref"] - truth)) / scale)
    print("dELBO/dZ vs the extended-precision truth: device %.3e, reference fp64 %.3e of the tensor" % (err_dev, err_ref))
    assert abs(err_ref - float(fx["ref_err"])) < 1e-12
    if "gz_ref_alt" in fx:
        # configs[4] itself: where the reference lands depends on its summation order -- 1.442e-3 of the tensor from the truth on 8 torch
        # threads, 1.928e-3 on 3 (the two runs 2.35e-3 apart); the device, the same bits on every run, 1.591e-3 (round 6; 2.188e-3 before the Gram kernels' polynomial degrees changed, 2.598e-3 before the panels of
        # the K_uu factorisation were refined: DESIGN 4b)
        err_alt = float(np.max(np.abs(fx["gz_ref_alt"] - truth)) / scale)
        print("    the reference again on %d threads: %.3e" % (int(fx["ref_alt_threads"]), err_alt))
        assert err_dev <= 1.25 * max(err_ref, err_alt), (err_dev, err_ref, err_alt)
        assert err_dev <= 1.75 * min(err_ref, err_alt), (err_dev, err_ref, err_alt)
    else:
        assert err_dev <= 1.1 * err_ref, (err_dev, err_ref)      # N = 20 000, measured: 2.365e-3 against the reference's 2.399e-3 (bit-reproducible)
    assert np.dot(gz, truth) / (np.linalg.norm(gz) * np.linalg.norm(truth)) > 0.99999


def test_exact_path_conditioning_envelope():
    """the exact path forms its panels with explicit tile / block inverses, so its distance from a backward-stable (LAPACK) factorisation
    grows with cond(K_j).  Pinned here: at a noise scale of 1e-2 on data of unit amplitude (cond ~ 1e6) north_star's tolerances still hold -- LML
    1e-9, every gradient tensor 1e-5 -- against the numpy twin on the same term table (measured 7e-10 and 1.5e-7; tools/exact_illcond.py has the
    rows beyond, where they do not)."""
    from oracle.table_model import TableDevice
    C, Q, N = 2, 2, 2048
    X, y = synth.make_data(N, C)
    h = synth.mosm_hypers(C, Q)
    out = {}
    for who in ("device", "twin"):
        k = gpr.MultiOutputSpectralMixtureKernel(Q=Q, output_dims=C)
        for name in ("weight", "mean", "variance", "delay", "phase"):
            getattr(k, name).assign(h[name])
        m = gpr.Exact(k, X, y, variance=1e-4)
        m.likelihood.scale.assign(1e-2)
        if who == "twin":
            m._handle = TableDevice(0, m.kernel._kernel_format(m.X), m.y, C)
        out[who] = (float(m.loss()), [p.grad.copy() for p in m.parameters()])
    ld, lt = out["device"][0], out["twin"][0]
    assert abs(ld - lt) <= 1e-9 * abs(lt), (ld, lt)
    for a, b in zip(out["device"][1], out["twin"][1]):
        assert np.max(np.abs(a - b)) <= 1e-5 * np.max(np.abs(b))
    # ... and beyond it the model says so and switches to the backward-stable form: the factor's own diagonal gives a lower estimate of the condition number
    import warnings

    def model(sigma):
        k = gpr.MultiOutputSpectralMixtureKernel(Q=Q, output_dims=C)
        for name in ("weight", "mean", "variance", "delay", "phase"):
            getattr(k, name).assign(h[name])
        m = gpr.Exact(k, X, y, variance=sigma ** 2)
        m.likelihood.scale.assign(sigma)
        return m

    for sigma, tol_l, tol_g in ((1e-3, 2e-9, 1e-5), (1e-4, 1e-7, 1e-3)):      # cond(Kj) 7.4e7 and 4e9: the fast schedules are 3.7e-7 / 2e-4 and 4.6e-3 / 0.15 off
        mt = model(sigma)
        mt._handle = TableDevice(0, mt.kernel._kernel_format(mt.X), mt.y, C)
        lt, gt = float(mt.loss()), [p.grad.copy() for p in mt.parameters()]
        m = model(sigma)
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            ld = float(m.loss())
            ld2 = float(m.loss())
        assert len([x for x in w if "ill-conditioned" in str(x.message)]) == 1        # once per model
        assert m._handle.accurate_mode and ld == ld2
        est = m._handle.condition_estimate()
        assert 1e5 < est, est                            # a lower bound of cond(Kj) (8.8e5 where it is 7.4e7)
        el = abs(ld - lt) / abs(lt)
        eg = max(float(np.max(np.abs(p.grad - b)) / np.max(np.abs(b))) for p, b in zip(m.parameters(), gt))
        print("sigma %.0e: backward-stable form against the LAPACK twin: LML %.2e, worst gradient tensor %.2e (estimate %.1e)" % (sigma, el, eg, est))
        assert el <= tol_l and eg <= tol_g, (sigma, el, eg)
    # the prediction of an ill-conditioned model: fast form first, then -- the warning, once -- the refined one (panels and solved block columns refined against L)
    Xs = synth.test_inputs(256, C)
    mt = model(1e-3)
    mt._handle = TableDevice(0, mt.kernel._kernel_format(mt.X), mt.y, C)
    mu_t, var_t = mt.predict_f(Xs)
    gpr.config.accurate_fallback = False
    try:
        with warnings.catch_warnings(record=True):
            warnings.simplefilter("always")
            mu_f, var_f = model(1e-3).predict_f(Xs)
    finally:
        gpr.config.accurate_fallback = True
    mp = model(1e-3)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        mu_a, var_a = mp.predict_f(Xs)
    assert len([x for x in w if "ill-conditioned" in str(x.message)]) == 1 and mp._handle.accurate_mode
    e = lambda a, b: float(np.max(np.abs(a - b)) / np.max(np.abs(b)))
    print("prediction at sigma 1e-3 against the LAPACK twin: mean fast %.2e refined %.2e; variance fast %.2e refined %.2e" % (e(mu_f, mu_t), e(mu_a, mu_t), e(var_f, var_t), e(var_a, var_t)))
    assert e(mu_a, mu_t) <= 1e-6 and e(var_a, var_t) <= 1e-6
    assert e(mu_a, mu_t) <= e(mu_f, mu_t) and e(var_a, var_t) <= e(var_f, var_t)
    # a well-conditioned model is left alone, and a model whose noise comes back up returns to the fast form
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        m2 = model(0.2)
        m2.loss()
    assert not [x for x in w if "ill-conditioned" in str(x.message)] and m2._handle.condition_estimate() < 1e4 and not getattr(m2._handle, "accurate_mode", False)
    m.likelihood.scale.assign(0.2)
    m.loss()
    assert not m._handle.accurate_mode
    l_fast = float(m.loss())
    assert abs(l_fast - float(m2.loss())) <= 1e-12 * abs(l_fast)


def test_a_dataflow_time_out_is_a_detour_not_a_verdict(tmp_path):
    """A hand-off of the dataflow schedule that times out repeats the evaluation on streams (same result) and keeps the model there -- for 64
    evaluations, then the dataflow kernel gets another try (four times as many after every further time-out): a long training run meets one stall
    in a few thousand evaluations on an idle box (tools/flow_soak.py) and used to pay 20 % for the rest of its life.  Forced here by the library's test hook
    (MOGP_FLOW_FAULT=1: every dataflow evaluation reports a time-out; read once per process, hence a process of its own)."""
    import os, subprocess, sys, json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "detour.py"
    script.write_text('''
import sys, json
sys.path.insert(0, %r)
from mogptk_amd import gpr, synth
C, Q, N = 2, 2, 3072
X, y = synth.make_data(N, C)
h = synth.mosm_hypers(C, Q)
k = gpr.MultiOutputSpectralMixtureKernel(Q=Q, output_dims=C)
for name in ("weight", "mean", "variance", "delay", "phase"):
    getattr(k, name).assign(h[name])
m = gpr.Exact(k, X, y, variance=h["scale"] ** 2)
m.likelihood.scale.assign(h["scale"])
out = []
for i in range(75):
    l = float(m.loss())
    s = m._handle.schedule()
    out.append((l, s["dataflow"], s["dataflow_fell_back"], s["dataflow_timeouts"]))
print(json.dumps(out))
''' % root)
    env = {k: v for k, v in os.environ.items() if not k.startswith("MOGP_")}
    env["MOGP_FLOW_FAULT"] = "1"
    p = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0, p.stdout[-1000:] + p.stderr[-2000:]
    out = json.loads(p.stdout.strip().splitlines()[-1])
    losses = [o[0] for o in out]
    assert max(losses) - min(losses) <= 1e-10 * abs(losses[0])            # (the two schedules sum z^T z in different orders)
    assert out[0][1:] == [False, True, 1], out[0]                         # timed out, repeated on streams, on the stream schedule now
    assert all(o[1:] == [False, True, 1] for o in out[:60]), out[:60]     # ... and for the next 64 factorisations
    assert out[-1][3] == 2 and out[-1][2], out[-1]                        # then another try (which this process makes time out as well): 256 this time
    assert "said once" in p.stderr and p.stderr.count("timed out") == 1
