"""
Likelihoods -- host-side mirror of mogptk/gpr/likelihood.py.  Exact inference uses the Gaussian one only (its noise term sigma_c^2 is added
to the Gram diagonal on the device); the variational models (Hensman, OpperArchambeau) take any of them: the device returns the mean and
variance of q(f) per point, the likelihood here (O(N) host work) returns its expectation E = sum_n E_q[log p(y_n | f_n)] together with
dE/dmu, dE/dvar per point and dE/dparameter, and the device turns those into the gradients of kernel and variational parameters.
The reference differentiates the Gauss-Hermite sum by autograd (likelihood.py:65-81, 150-166); here the derivative of every log density
with respect to f and to its own parameters is written out and pinned on the reference's autograd (tests/golden/likelihoods.npz).
"""
import numpy as np
from math import erf, sqrt

from .config import config
from .parameter import Parameter, ParameterHolder


# ---- link functions (reference likelihood.py:5-52): callables that also know their derivative ---------------------------------------------
class _Link:
    def __init__(self, name, fn, dfn):
        self.__name__, self._fn, self._dfn = name, fn, dfn

    def __call__(self, x):
        return self._fn(np.asarray(x, dtype=np.float64))

    def d(self, x):
        return self._dfn(np.asarray(x, dtype=np.float64))

    def __repr__(self):
        return "<link %s>" % self.__name__

    def __reduce__(self):                    # pickles by name, so `link is exp` survives save / load
        return (_link_by_name, (self.__name__,))


def _link_by_name(name):
    return _LINKS[name]


def _erfinv(x):
    from scipy.special import erfinv
    return erfinv(x)


def _erf(x):
    from scipy.special import erf as _e
    return _e(x)


_PROBIT_JITTER = 1e-3
identity = _Link("identity", lambda x: x, lambda x: np.ones_like(x))
square = _Link("square", lambda x: x * x, lambda x: 2.0 * x)
exp = _Link("exp", np.exp, np.exp)
probit = _Link("probit", lambda x: np.sqrt(2.0) * _erfinv(2.0 * x - 1.0), lambda x: np.sqrt(2.0 * np.pi) * np.exp(_erfinv(2.0 * x - 1.0) ** 2))
inv_probit = _Link("inv_probit", lambda x: 0.5 * (1.0 + _erf(x / np.sqrt(2.0))) * (1.0 - 2.0 * _PROBIT_JITTER) + _PROBIT_JITTER,
                   lambda x: (1.0 - 2.0 * _PROBIT_JITTER) * np.exp(-0.5 * x * x) / np.sqrt(2.0 * np.pi))
sigmoid = _Link("sigmoid", lambda x: 1.0 / (1.0 + np.exp(-x)), lambda x: (1.0 / (1.0 + np.exp(-x))) * (1.0 - 1.0 / (1.0 + np.exp(-x))))
_LINKS = {l.__name__: l for l in (identity, square, exp, probit, inv_probit, sigmoid)}


def _check_link(link):
    if not isinstance(link, _Link):
        raise ValueError("link must be one of gpr.identity, square, exp, probit, inv_probit, sigmoid (their derivatives are needed: there is no autograd here)")
    return link


def _torch():
    try:
        import torch
    except ImportError as e:
        raise ImportError("sampling from a likelihood (confidence intervals of non-Gaussian likelihoods) draws from torch's generator, like "
                          "the reference: torch must be importable") from e
    return torch


class GaussHermiteQuadrature:
    """reference likelihood.py:65-81: nodes t (scaled) and weights w (scaled); __call__ integrates F(mu + sqrt(var) t) against w"""

    def __init__(self, deg=20, t_scale=None, w_scale=None):
        t, w = np.polynomial.hermite.hermgauss(deg)
        self.t = t * (1.0 if t_scale is None else t_scale)
        self.w = w * (1.0 if w_scale is None else w_scale)
        self.deg = deg

    def points(self, mu, var):
        return np.reshape(mu, (-1, 1)) + np.sqrt(np.reshape(var, (-1, 1))) * self.t[None, :]      # N x deg

    def __call__(self, mu, var, F):
        return (F(self.points(mu, var)) @ self.w).reshape(-1, 1)


class Likelihood(ParameterHolder):
    """Base likelihood (reference likelihood.py:83-212)."""

    def __init__(self, quadratures=20):
        self.quadrature = GaussHermiteQuadrature(deg=quadratures, t_scale=np.sqrt(2.0), w_scale=1.0 / np.sqrt(np.pi))
        self.output_dims = None

    def name(self):
        return self.__class__.__name__

    def _channel_indices(self, X):
        c = np.asarray(X)[:, 0].astype(np.int64)
        return [np.nonzero(c == i)[0] for i in range(self.output_dims)]

    def validate_y(self, X, y):
        pass

    def log_prob(self, X, y, f):
        """log p(y | f): y (N,1), f (N,Q) -> (N,Q)"""
        raise NotImplementedError()

    def _dlog_prob(self, X, y, f):
        """-> d log p / d f (N,Q) and a list of (parameter, d log p / d parameter (N,Q))"""
        raise NotImplementedError()

    def variational_expectation(self, X, y, mu, var, grad=False):
        """sum_n E_{N(mu_n, var_n)}[log p(y_n | f)] by Gauss-Hermite quadrature (reference likelihood.py:150-166).
        grad=True -> (E, dE/dmu (N,), dE/dvar (N,), [(parameter, dE/dparameter)]): the quadrature sum differentiated term by term."""
        y = np.reshape(y, (-1, 1))
        q = self.quadrature
        F = q.points(mu, var)
        ve = float(np.sum(self.log_prob(X, y, F) @ q.w))
        if not grad:
            return ve
        dF, dP = self._dlog_prob(X, y, F)
        e = dF @ q.w
        f = (dF * q.t[None, :]) @ q.w / (2.0 * np.sqrt(np.reshape(var, -1)))
        return ve, e, f, [(p, float(np.sum(g @ q.w))) for p, g in dP]

    def conditional_mean(self, X, f):
        raise NotImplementedError()

    def conditional_sample(self, X, f):
        """f: torch tensor of samples -> torch tensor of samples of y (or None), drawn from torch's global generator like the reference"""
        raise NotImplementedError()

    def predict(self, X, mu, var, ci=None, sigma=None, n=10000):
        """reference likelihood.py:196-212, as written there: the quantiles come from n samples f ~ N(mean of y, var) -- the quadrature mean
        of y in place of the mean of f and the variance in place of the standard deviation -- pushed through conditional_sample"""
        mu = self.quadrature(mu, var, lambda f: self.conditional_mean(X, f))
        if ci is None:
            return mu
        torch = _torch()
        tm, tv = torch.tensor(np.asarray(mu, dtype=np.float64)), torch.tensor(np.reshape(np.asarray(var, dtype=np.float64), (-1, 1)))
        samples_f = torch.distributions.normal.Normal(tm, tv).sample([n])
        samples_y = self.conditional_sample(X, samples_f)
        if samples_y is None:
            return mu, mu, mu
        samples_y, _ = samples_y.sort(dim=0)
        lower, upper = int(ci[0] * n + 0.5), int(ci[1] * n + 0.5)
        return mu, samples_y[lower, :].numpy(), samples_y[upper, :].numpy()


class MultiOutputLikelihood(Likelihood):
    """One likelihood per channel (reference likelihood.py:214-310)."""

    def __init__(self, *likelihoods):
        super().__init__()
        if len(likelihoods) == 1 and isinstance(likelihoods[0], list):
            likelihoods = likelihoods[0]
        likelihoods = list(likelihoods)
        if len(likelihoods) == 0:
            raise ValueError("must pass at least one likelihood")
        for likelihood in likelihoods:
            if not isinstance(likelihood, Likelihood):
                raise ValueError("must pass likelihoods")
            if isinstance(likelihood, MultiOutputLikelihood):
                raise ValueError("can not nest MultiOutputLikelihoods")
        self.output_dims = len(likelihoods)
        self.likelihoods = likelihoods

    def name(self):
        return "[%s]" % (",".join(l.name() for l in self.likelihoods),)

    def validate_y(self, X, y):
        if self.output_dims == 1:
            self.likelihoods[0].validate_y(X, y)
            return
        r = self._channel_indices(X)
        for i in range(self.output_dims):
            self.likelihoods[i].validate_y(X, np.reshape(y, (-1, 1))[r[i], :])

    def log_prob(self, X, y, f):
        r = self._channel_indices(X)
        res = np.empty(f.shape)
        for i in range(self.output_dims):
            res[r[i], :] = self.likelihoods[i].log_prob(X, y[r[i], :], f[r[i], :])
        return res

    def variational_expectation(self, X, y, mu, var, grad=False):
        y, mu, var = np.reshape(y, (-1, 1)), np.reshape(mu, (-1, 1)), np.reshape(var, (-1, 1))
        r = self._channel_indices(X)
        ve, e, f, pg = 0.0, np.zeros(y.shape[0]), np.zeros(y.shape[0]), []
        for i in range(self.output_dims):
            res = self.likelihoods[i].variational_expectation(X, y[r[i], :], mu[r[i], :], var[r[i], :], grad=grad)
            if not grad:
                ve += res
                continue
            ve += res[0]
            e[r[i]], f[r[i]] = res[1], res[2]
            pg += res[3]
        return (ve, e, f, pg) if grad else ve

    def conditional_mean(self, X, f):
        r = self._channel_indices(X)
        res = np.empty(f.shape)
        for i in range(self.output_dims):
            res[r[i], :] = self.likelihoods[i].conditional_mean(X, f[r[i], :])
        return res

    def conditional_sample(self, X, f):
        r = self._channel_indices(X)
        for i in range(self.output_dims):
            f[:, r[i]] = self.likelihoods[i].conditional_sample(X, f[:, r[i]])
        return f

    def predict(self, X, mu, var, ci=None, sigma=None, n=10000):
        mu, var = np.reshape(mu, (-1, 1)), np.reshape(var, (-1, 1))
        r = self._channel_indices(X)
        res = np.empty(mu.shape)
        if ci is None:
            for i in range(self.output_dims):
                res[r[i], :] = self.likelihoods[i].predict(X, mu[r[i], :], var[r[i], :], ci=ci, sigma=sigma, n=n)
            return res
        lower, upper = np.empty(mu.shape), np.empty(mu.shape)
        for i in range(self.output_dims):
            res[r[i], :], lower[r[i], :], upper[r[i], :] = self.likelihoods[i].predict(X, mu[r[i], :], var[r[i], :], ci=ci, sigma=sigma, n=n)
        return res, lower, upper


class GaussianLikelihood(Likelihood):
    """
    p(y|f) = N(f, scale^2); `scale` is a float or a (output_dims,) array (reference likelihood.py:325-329).
    """

    def __init__(self, scale=1.0):
        super().__init__()
        self.scale = Parameter(scale, lower=config.positive_minimum)
        if self.scale.ndim == 1:
            self.output_dims = self.scale.shape[0]

    def _channel_indices(self, X):
        c = X[:, 0].astype(np.int64)
        return [np.nonzero(c == i)[0] for i in range(self.output_dims)]

    def variational_expectation(self, X, y, mu, var, grad=False):
        """E_q[log p(y | f)] summed over the points, q(f_n) = N(mu_n, var_n) -- closed form for the Gaussian (reference likelihood.py:338-343).
        grad=True also returns dE/dmu, dE/dvar (per point) and dE/dscale.  Scalar scale only: the reference's formula broadcasts a
        per-channel scale against the points (N x C terms) instead of indexing it."""
        s = np.asarray(self.scale(), dtype=np.float64)
        if s.size != 1:
            raise NotImplementedError("variational_expectation with a per-channel Gaussian scale follows a defective reference formula "
                                      "(likelihood.py:338-343 broadcasts (N,1) against (channels,)); use a scalar scale")
        s = float(s.reshape(-1)[0])
        y, mu, var = np.reshape(y, -1), np.reshape(mu, -1), np.reshape(var, -1)
        r2 = (y - mu) ** 2 + var
        ve = 0.5 * np.sum(-r2 / s ** 2 - np.log(2.0 * np.pi) - 2.0 * np.log(s))
        if not grad:
            return ve
        return ve, (y - mu) / s ** 2, np.full(y.shape, -0.5 / s ** 2), [(self.scale, np.sum(r2) / s ** 3 - y.size / s)]

    def log_prob(self, X, y, f):
        s = np.asarray(self.scale(), dtype=np.float64)
        return -0.5 * (np.log(2.0 * np.pi) + 2.0 * np.log(s) + ((y - f) / s) ** 2)

    def conditional_mean(self, X, f):
        return f

    def conditional_sample(self, X, f):
        torch = _torch()
        return torch.distributions.normal.Normal(f, scale=torch.tensor(np.asarray(self.scale(), dtype=np.float64))).sample()

    def predict(self, X, mu, var, ci=None, sigma=None, n=10000):
        """reference likelihood.py:351-378, quirk Q4 included: with a per-channel scale the interval is
        mu -/+ sigma*scale_c and ignores the GP variance; the single-output branch adds scale^2 to var."""
        if ci is None and sigma is None:
            return mu
        scale = self.scale()
        if self.output_dims is not None:
            if sigma is None:
                raise NotImplementedError("ci= quantiles for the multi-output Gaussian likelihood follow a "
                                          "defective reference formula (likelihood.py:363-364); use sigma=")
            r = self._channel_indices(X)
            lower = np.empty(mu.shape)
            upper = np.empty(mu.shape)
            for i in range(self.output_dims):
                lower[r[i], :] = mu[r[i], :] - sigma * scale[i]
                upper[r[i], :] = mu[r[i], :] + sigma * scale[i]
            return mu, lower, upper
        var = var + scale ** 2
        if sigma is None:
            from scipy.special import erfinv
            lower = mu + np.sqrt(2.0 * var) * erfinv(2.0 * ci[0] - 1.0)
            upper = mu + np.sqrt(2.0 * var) * erfinv(2.0 * ci[1] - 1.0)
        else:
            lower = mu - sigma * np.sqrt(var)
            upper = mu + sigma * np.sqrt(var)
        return mu, lower, upper


def _gammaln(x):
    from scipy.special import gammaln
    return gammaln(x)


def _psi(x):
    from scipy.special import digamma
    return digamma(x)


def _t(x):
    return _torch().tensor(np.asarray(x, dtype=np.float64))


class StudentTLikelihood(Likelihood):
    """Student's t likelihood with fixed degrees of freedom and a trained scale (reference likelihood.py:380-420)."""

    def __init__(self, dof=3, scale=1.0, quadratures=20):
        super().__init__(quadratures)
        self.dof = float(dof)
        self.scale = Parameter(scale, lower=config.positive_minimum)

    def log_prob(self, X, y, f):
        nu, s = self.dof, np.asarray(self.scale(), dtype=np.float64)
        p = -0.5 * (nu + 1.0) * np.log1p(((y - f) / s) ** 2 / nu)
        p = p + _gammaln((nu + 1.0) / 2.0) - _gammaln(nu / 2.0)
        return p - 0.5 * (np.log(nu) + np.log(np.pi) + np.log(s ** 2))

    def _dlog_prob(self, X, y, f):
        nu, s = self.dof, float(np.asarray(self.scale()).reshape(-1)[0])
        r = (y - f) / s
        den = 1.0 + r * r / nu
        return (nu + 1.0) * r / (s * nu * den), [(self.scale, (nu + 1.0) * r * r / (nu * s * den) - 1.0 / s)]

    def conditional_mean(self, X, f):
        if self.dof <= 1.0:
            return np.full(np.shape(f), np.nan)
        return f

    def conditional_sample(self, X, f):
        return _torch().distributions.studentT.StudentT(_t(self.dof), f, _t(self.scale())).sample()


class _ExpClosedForm(Likelihood):
    """The three likelihoods whose expectation under the exponential link has a closed form.  The reference evaluates that closed form
    WHATEVER the link is (likelihood.py:449-454, 617-626, 673-678: the quadrature branch computes its value and drops it); so does this."""


class ExponentialLikelihood(_ExpClosedForm):
    """p(y|f) = exp(-y / h(f)) / h(f) (reference likelihood.py:422-470)."""

    def __init__(self, link=exp, quadratures=20):
        super().__init__(quadratures)
        self.link = _check_link(link)

    def validate_y(self, X, y):
        if np.any(np.asarray(y) < 0.0):
            raise ValueError("y must be positive")

    def log_prob(self, X, y, f):
        if self.link is exp:
            return -y / self.link(f) - f
        return -y / self.link(f) - np.log(self.link(f))

    def variational_expectation(self, X, y, mu, var, grad=False):
        y, mu, var = np.reshape(y, -1), np.reshape(mu, -1), np.reshape(var, -1)
        ex = y * np.exp(var / 2.0 - mu)
        ve = float(np.sum(-mu - ex))
        return (ve, -1.0 + ex, -0.5 * ex, []) if grad else ve

    def conditional_mean(self, X, f):
        return self.link(f)

    def conditional_sample(self, X, f):
        if self.link is not exp:
            raise ValueError("only exponential link function is supported")
        torch = _torch()
        return torch.distributions.exponential.Exponential(1.0 / torch.exp(f)).sample().log()


class LaplaceLikelihood(Likelihood):
    """p(y|f) = exp(-|y - f| / scale) / (2 scale) (reference likelihood.py:472-512)."""

    def __init__(self, scale=1.0, quadratures=20):
        super().__init__(quadratures)
        self.scale = Parameter(scale, lower=config.positive_minimum)

    def log_prob(self, X, y, f):
        s = np.asarray(self.scale(), dtype=np.float64)
        return -np.log(2.0 * s) - np.abs(y - f) / s

    def _dlog_prob(self, X, y, f):
        s = float(np.asarray(self.scale()).reshape(-1)[0])
        return np.sign(y - f) / s, [(self.scale, -1.0 / s + np.abs(y - f) / s ** 2)]

    def conditional_mean(self, X, f):
        return f

    def conditional_sample(self, X, f):
        return _torch().distributions.laplace.Laplace(f, _t(self.scale())).sample()


class BernoulliLikelihood(Likelihood):
    """p(y|f) = h(f) for y = 1, 1 - h(f) for y = 0 (reference likelihood.py:514-555)."""

    def __init__(self, link=inv_probit):
        super().__init__()
        self.link = _check_link(link)

    def validate_y(self, X, y):
        y = np.asarray(y)
        if np.any((y != 0.0) & (y != 1.0)):
            raise ValueError("y must have only 0.0 and 1.0 values")

    def log_prob(self, X, y, f):
        p = self.link(f)
        return np.log(np.where(0.5 <= y, p, 1.0 - p))

    def _dlog_prob(self, X, y, f):
        p, dp = self.link(f), self.link.d(f)
        return np.where(0.5 <= y, dp / p, -dp / (1.0 - p)), []

    def conditional_mean(self, X, f):
        return self.link(f)

    def conditional_sample(self, X, f):
        return None

    def predict(self, X, mu, var, ci=None, sigma=None, n=10000):
        if self.link is not inv_probit:
            return super().predict(X, mu, var, ci=ci, sigma=sigma, n=n)
        p = self.link(np.asarray(mu) / np.sqrt(1.0 + np.asarray(var)))
        if ci is None and sigma is None:
            return p
        return p, p, p


class BetaLikelihood(Likelihood):
    """Beta likelihood with mean h(f) and trained scale (reference likelihood.py:557-606)."""

    def __init__(self, scale=1.0, link=inv_probit, quadratures=20):
        super().__init__(quadratures)
        self.link = _check_link(link)
        self.scale = Parameter(scale, lower=config.positive_minimum)

    def validate_y(self, X, y):
        y = np.asarray(y)
        if np.any((y <= 0.0) | (1.0 <= y)):
            raise ValueError("y must be in the range (0.0,1.0)")

    def log_prob(self, X, y, f):
        s = np.asarray(self.scale(), dtype=np.float64)
        alpha = self.link(f) * s
        beta = s - alpha
        return (alpha - 1.0) * np.log(y) + (beta - 1.0) * np.log1p(-y) + _gammaln(alpha + beta) - _gammaln(alpha) - _gammaln(beta)

    def _dlog_prob(self, X, y, f):
        s = float(np.asarray(self.scale()).reshape(-1)[0])
        h = self.link(f)
        alpha, beta = h * s, s - h * s
        ly, l1y = np.log(y), np.log1p(-y)
        dh = s * (ly - l1y - _psi(alpha) + _psi(beta))
        ds = h * ly + (1.0 - h) * l1y + _psi(alpha + beta) - h * _psi(alpha) - (1.0 - h) * _psi(beta)
        return dh * self.link.d(f), [(self.scale, ds)]

    def conditional_mean(self, X, f):
        return self.link(f)

    def conditional_sample(self, X, f):
        if self.link is not inv_probit:
            raise ValueError("only inverse probit link function is supported")
        torch = _torch()
        jitter = _PROBIT_JITTER
        mixture = 0.5 * (1.0 + torch.erf(f / np.sqrt(2.0))) * (1.0 - 2.0 * jitter) + jitter
        alpha = mixture * _t(self.scale())
        beta = _t(self.scale()) - alpha
        return np.sqrt(2) * torch.erfinv(2.0 * torch.distributions.beta.Beta(alpha, beta).sample() - 1.0)


class GammaLikelihood(_ExpClosedForm):
    """Gamma likelihood with trained shape k and scale h(f) (reference likelihood.py:608-652)."""

    def __init__(self, shape=1.0, link=exp, quadratures=20):
        super().__init__(quadratures)
        self.link = _check_link(link)
        self.shape = Parameter(shape, lower=config.positive_minimum)

    def validate_y(self, X, y):
        if np.any(np.asarray(y) <= 0.0):
            raise ValueError("y must be in the range (0.0,inf)")

    def log_prob(self, X, y, f):
        k = np.asarray(self.shape(), dtype=np.float64)
        p = -y / self.link(f) + (k - 1.0) * np.log(y) - _gammaln(k)
        return p - k * (f if self.link is exp else np.log(self.link(f)))

    def variational_expectation(self, X, y, mu, var, grad=False):
        k = float(np.asarray(self.shape()).reshape(-1)[0])
        y, mu, var = np.reshape(y, -1), np.reshape(mu, -1), np.reshape(var, -1)
        ex = y * np.exp(var / 2.0 - mu)
        ve = float(np.sum(-k * mu - _gammaln(k) + (k - 1.0) * np.log(y) - ex))
        if not grad:
            return ve
        return ve, -k + ex, -0.5 * ex, [(self.shape, float(np.sum(-mu - _psi(k) + np.log(y))))]

    def conditional_mean(self, X, f):
        return np.asarray(self.shape(), dtype=np.float64) * self.link(f)

    def conditional_sample(self, X, f):
        if self.link is not exp:
            raise ValueError("only exponential link function is supported")
        torch = _torch()
        return torch.distributions.gamma.Gamma(_t(self.shape()), 1.0 / torch.exp(f)).sample().log()


class PoissonLikelihood(_ExpClosedForm):
    """Poisson likelihood with rate h(f) (reference likelihood.py:654-693)."""

    def __init__(self, link=exp, quadratures=20):
        super().__init__(quadratures)
        self.link = _check_link(link)

    def validate_y(self, X, y):
        y = np.asarray(y)
        if np.any(y < 0.0):
            raise ValueError("y must be in the range [0.0,inf)")
        if not np.all(y == np.trunc(y)):
            raise ValueError("y must have integer count values")

    def log_prob(self, X, y, f):
        p = y * f if self.link is exp else y * np.log(self.link(f))
        return p - _gammaln(y + 1.0) - self.link(f)

    def variational_expectation(self, X, y, mu, var, grad=False):
        y, mu, var = np.reshape(y, -1), np.reshape(mu, -1), np.reshape(var, -1)
        ex = np.exp(var / 2.0 + mu)
        ve = float(np.sum(y * mu - ex - _gammaln(y + 1.0)))
        return (ve, y - ex, -0.5 * ex, []) if grad else ve

    def conditional_mean(self, X, f):
        return self.link(f)

    def conditional_sample(self, X, f):
        if self.link is not exp:
            raise ValueError("only exponential link function is supported")
        torch = _torch()
        return torch.distributions.poisson.Poisson(torch.exp(f)).sample().log()


class WeibullLikelihood(Likelihood):
    """Weibull likelihood with trained shape k and scale h(f) (reference likelihood.py:695-738)."""

    def __init__(self, shape=1.0, link=exp, quadratures=20):
        super().__init__(quadratures)
        self.link = _check_link(link)
        self.shape = Parameter(shape, lower=config.positive_minimum)

    def validate_y(self, X, y):
        if np.any(np.asarray(y) <= 0.0):
            raise ValueError("y must be in the range (0.0,inf)")

    def log_prob(self, X, y, f):
        k = np.asarray(self.shape(), dtype=np.float64)
        p = -k * (f if self.link is exp else np.log(self.link(f)))
        return p + np.log(k) + (k - 1.0) * np.log(y) - (y / self.link(f)) ** k

    def _dlog_prob(self, X, y, f):
        k = float(np.asarray(self.shape()).reshape(-1)[0])
        h = self.link(f)
        z = (y / h) ** k
        return (k / h) * (z - 1.0) * self.link.d(f), [(self.shape, -np.log(h) + 1.0 / k + np.log(y) - z * np.log(y / h))]

    def conditional_mean(self, X, f):
        return self.link(f) * np.exp(_gammaln(1.0 + 1.0 / np.asarray(self.shape(), dtype=np.float64)))

    def conditional_sample(self, X, f):
        if self.link is not exp:
            raise ValueError("only exponential link function is supported")
        torch = _torch()
        return torch.distributions.weibull.Weibull(torch.exp(f), _t(self.shape())).sample().log()


class LogLogisticLikelihood(Likelihood):
    """Log-logistic likelihood with trained shape k and scale h(f) (reference likelihood.py:740-783)."""

    def __init__(self, shape=1.0, link=exp, quadratures=20):
        super().__init__(quadratures)
        self.link = _check_link(link)
        self.shape = Parameter(shape, lower=config.positive_minimum)

    def validate_y(self, X, y):
        if np.any(np.asarray(y) < 0.0):
            raise ValueError("y must be in the range [0.0,inf)")

    def log_prob(self, X, y, f):
        k = np.asarray(self.shape(), dtype=np.float64)
        p = -k * (f if self.link is exp else np.log(self.link(f)))
        return p - 2.0 * np.log1p((y / self.link(f)) ** k) + np.log(k) + (k - 1.0) * np.log(y)

    def _dlog_prob(self, X, y, f):
        k = float(np.asarray(self.shape()).reshape(-1)[0])
        h = self.link(f)
        z = (y / h) ** k
        return ((k / h) * (2.0 * z / (1.0 + z) - 1.0) * self.link.d(f),
                [(self.shape, -np.log(h) - 2.0 * z * np.log(y / h) / (1.0 + z) + 1.0 / k + np.log(y))])

    def conditional_mean(self, X, f):
        return self.link(f) / np.sinc(1.0 / np.asarray(self.shape(), dtype=np.float64))

    def conditional_sample(self, X, f):
        if self.link is not exp:
            raise ValueError("only exponential link function is supported")
        torch = _torch()
        td = torch.distributions
        dist = td.transformed_distribution.TransformedDistribution(
            base_distribution=td.uniform.Uniform(0.0, 1.0),
            transforms=[td.transforms.SigmoidTransform().inv, td.transforms.AffineTransform(loc=f, scale=1.0 / _t(self.shape())),
                        td.transforms.ExpTransform()])
        return dist.sample().log()


class LogGaussianLikelihood(Likelihood):
    """Log-Gaussian likelihood: log y ~ N(f, scale^2) (reference likelihood.py:785-825)."""

    def __init__(self, scale=1.0, quadratures=20):
        super().__init__(quadratures)
        self.scale = Parameter(scale, lower=config.positive_minimum)

    def validate_y(self, X, y):
        if np.any(np.asarray(y) <= 0.0):
            raise ValueError("y must be in the range (0.0,inf)")

    def log_prob(self, X, y, f):
        s = np.asarray(self.scale(), dtype=np.float64)
        logy = np.log(y)
        return -0.5 * (np.log(2.0 * np.pi) + 2.0 * np.log(s) + ((logy - f) / s) ** 2) - logy

    def _dlog_prob(self, X, y, f):
        s = float(np.asarray(self.scale()).reshape(-1)[0])
        d = np.log(y) - f
        return d / s ** 2, [(self.scale, -1.0 / s + d * d / s ** 3)]

    def conditional_mean(self, X, f):
        return np.exp(f + 0.5 * np.asarray(self.scale(), dtype=np.float64) ** 2)

    def conditional_sample(self, X, f):
        return _torch().distributions.log_normal.LogNormal(f, _t(self.scale())).sample().log()


class ChiSquaredLikelihood(Likelihood):
    """Chi-squared likelihood with h(f) degrees of freedom (reference likelihood.py:827-870)."""

    def __init__(self, link=exp, quadratures=20):
        super().__init__(quadratures)
        self.link = _check_link(link)

    def validate_y(self, X, y):
        if np.any(np.asarray(y) <= 0.0):
            raise ValueError("y must be in the range (0.0,inf)")

    def log_prob(self, X, y, f):
        g = self.link(f)
        return -0.5 * g * np.log(2.0) - _gammaln(g / 2.0) + (g / 2.0 - 1.0) * np.log(y) - 0.5 * y

    def _dlog_prob(self, X, y, f):
        g = self.link(f)
        return (-0.5 * np.log(2.0) - 0.5 * _psi(g / 2.0) + 0.5 * np.log(y)) * self.link.d(f), []

    def conditional_mean(self, X, f):
        return self.link(f)

    def conditional_sample(self, X, f):
        if self.link is not exp:
            raise ValueError("only exponential link function is supported")
        torch = _torch()
        return torch.distributions.chi2.Chi2(torch.exp(f)).sample().log()
