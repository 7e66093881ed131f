"""
The likelihoods of the variational models (reference gpr/likelihood.py), host-side O(N) work: every log density, its
Gauss-Hermite expectation and the derivatives this package writes out by hand (the reference takes them by autograd), the predictive mean
and the sampled quantiles -- against tests/golden/likelihoods.npz, recorded from the reference.  No device work in this file.
"""
import numpy as np
import pytest

from mogptk_amd import gpr
from helpers import load, fixture_params, load_raw, relerr


def zoo():
    L = gpr
    return {
        "gaussian": lambda: L.GaussianLikelihood(0.7),
        "studentt": lambda: L.StudentTLikelihood(dof=4, scale=0.6),
        "exponential": lambda: L.ExponentialLikelihood(),
        "laplace": lambda: L.LaplaceLikelihood(scale=0.8),
        "bernoulli": lambda: L.BernoulliLikelihood(),
        "bernoulli_sigmoid": lambda: L.BernoulliLikelihood(link=L.sigmoid),
        "beta": lambda: L.BetaLikelihood(scale=3.0),
        "beta_sigmoid": lambda: L.BetaLikelihood(scale=2.5, link=L.sigmoid),
        "gamma": lambda: L.GammaLikelihood(shape=1.7),
        "poisson": lambda: L.PoissonLikelihood(),
        "weibull": lambda: L.WeibullLikelihood(shape=1.4),
        "weibull_square": lambda: L.WeibullLikelihood(shape=0.8, link=L.square),
        "loglogistic": lambda: L.LogLogisticLikelihood(shape=2.2),
        "loggaussian": lambda: L.LogGaussianLikelihood(scale=0.5),
        "chisquared": lambda: L.ChiSquaredLikelihood(),
    }


def accumulate(pgrads):
    for p, g in pgrads:
        p.accumulate_grad(np.reshape(np.asarray(g, dtype=np.float64), p.data.shape))


@pytest.mark.parametrize("tag", sorted(zoo()))
def test_likelihood_matches_reference(tag):
    fx = load("likelihoods.npz")
    assert tag in [str(t) for t in fx["tags"]]
    lik = zoo()[tag]()
    fp = fixture_params(fx, tag + "_lik_")
    ps = load_raw(lik.parameters(), fp)
    assert [p._name for p in ps] == [f["name"] for f in fp]
    X, y, mu, var, f = fx[tag + "_X"], fx[tag + "_y"], fx[tag + "_mu"], fx[tag + "_var"], fx[tag + "_f"]
    lik.validate_y(X, y)
    assert relerr(lik.log_prob(X, y, f), fx[tag + "_logp"]) < 1e-12
    ve = lik.variational_expectation(X, y, mu, var)
    assert abs(ve - float(fx[tag + "_ve"])) < 1e-12 * max(1.0, abs(float(fx[tag + "_ve"])))
    ve2, e, g, pgrads = lik.variational_expectation(X, y, mu, var, grad=True)
    assert ve2 == ve
    assert np.max(np.abs(e - fx[tag + "_dmu"])) < 1e-10 * max(1.0, np.max(np.abs(fx[tag + "_dmu"])))
    assert np.max(np.abs(g - fx[tag + "_dvar"])) < 1e-10 * max(1.0, np.max(np.abs(fx[tag + "_dvar"])))
    for p in ps:
        p.grad = None
    accumulate(pgrads)
    for p, r in zip(ps, fp):
        assert p.grad is not None and np.max(np.abs(p.grad - r["grad"])) < 1e-10 * max(1.0, np.max(np.abs(r["grad"]))), p._name
    assert relerr(lik.conditional_mean(X, f), fx[tag + "_cmean"]) < 1e-12
    assert relerr(np.reshape(lik.predict(X, mu, var), -1), fx[tag + "_pmean"]) < 1e-12


@pytest.mark.parametrize("tag", [t for t in sorted(zoo()) if t != "gaussian"])
def test_likelihood_quantiles_draw_like_the_reference(tag):
    """predict(ci=...) samples from torch's global generator with the reference's own sequence of calls: same seed, same quantiles"""
    torch = pytest.importorskip("torch")
    fx = load("likelihoods.npz")
    lik = zoo()[tag]()
    load_raw(lik.parameters(), fixture_params(fx, tag + "_lik_"))
    X, mu, var = fx[tag + "_X"], fx[tag + "_mu"], fx[tag + "_var"]
    torch.manual_seed(1234)
    if tag + "_ci_error" in fx:                                  # a link the reference's sampler refuses
        with pytest.raises(ValueError):
            lik.predict(X, mu, var, ci=[0.1, 0.9], n=500)
        return
    m, lo, hi = lik.predict(X, mu, var, ci=[0.1, 0.9], n=500)
    assert relerr(np.reshape(m, -1), fx[tag + "_pmean"]) < 1e-12
    assert np.allclose(np.reshape(lo, -1), fx[tag + "_lo"], rtol=1e-9, atol=1e-9)          # (the log of a zero count is -inf on both sides)
    assert np.allclose(np.reshape(hi, -1), fx[tag + "_hi"], rtol=1e-9, atol=1e-9)


def test_multi_output_likelihood_matches_reference():
    fx = load("likelihoods.npz")
    L = gpr
    lik = L.MultiOutputLikelihood(L.StudentTLikelihood(dof=5, scale=0.5), L.PoissonLikelihood(), L.WeibullLikelihood(shape=1.3))
    fp = fixture_params(fx, "multi_lik_")
    ps = load_raw(lik.parameters(), fp)
    assert [p._name for p in ps] == [f["name"] for f in fp] and lik.name() == str(fx["multi_name"]) and lik.output_dims == 3
    X, y, mu, var = fx["multi_X"], fx["multi_y"], fx["multi_mu"], fx["multi_var"]
    lik.validate_y(X, y)
    ve, e, g, pgrads = lik.variational_expectation(X, y, mu, var, grad=True)
    assert abs(ve - float(fx["multi_ve"])) < 1e-12 * abs(float(fx["multi_ve"]))
    assert np.max(np.abs(e - fx["multi_dmu"])) < 1e-10 and np.max(np.abs(g - fx["multi_dvar"])) < 1e-10
    accumulate(pgrads)
    for p, r in zip(ps, fp):
        assert np.max(np.abs(p.grad - r["grad"])) < 1e-10 * max(1.0, np.max(np.abs(r["grad"]))), p._name
    assert relerr(np.reshape(lik.predict(X, mu, var), -1), fx["multi_pmean"]) < 1e-12
    with pytest.raises(ValueError):
        L.MultiOutputLikelihood(lik)
    with pytest.raises(ValueError):
        lik.validate_y(X, -np.abs(y) - 0.5)


def test_support_checks_and_links():
    L = gpr
    X = np.zeros((3, 2))
    for lik, bad in ((L.ExponentialLikelihood(), [-1.0, 1, 1]), (L.BernoulliLikelihood(), [0.0, 0.5, 1]), (L.BetaLikelihood(), [0.0, 0.5, 0.5]),
                     (L.GammaLikelihood(), [0.0, 1, 1]), (L.PoissonLikelihood(), [1.5, 1, 1]), (L.PoissonLikelihood(), [-1.0, 1, 1]),
                     (L.WeibullLikelihood(), [0.0, 1, 1]), (L.LogLogisticLikelihood(), [-0.1, 1, 1]), (L.LogGaussianLikelihood(), [0.0, 1, 1]),
                     (L.ChiSquaredLikelihood(), [0.0, 1, 1])):
        with pytest.raises(ValueError):
            lik.validate_y(X, np.array(bad).reshape(-1, 1))
    with pytest.raises(ValueError):
        L.PoissonLikelihood(link=np.exp)                              # a bare callable has no derivative
    x = np.linspace(-2, 2, 9)
    for link in (L.identity, L.square, L.exp, L.inv_probit, L.sigmoid):
        num = (link(x + 1e-6) - link(x - 1e-6)) / 2e-6
        assert np.max(np.abs(link.d(x) - num)) < 1e-8 * max(1.0, np.max(np.abs(num)))
    u = np.linspace(0.1, 0.9, 9)
    assert np.max(np.abs(L.probit.d(u) - (L.probit(u + 1e-6) - L.probit(u - 1e-6)) / 2e-6)) < 1e-7
    import pickle
    assert pickle.loads(pickle.dumps(L.PoissonLikelihood())).link is L.exp
