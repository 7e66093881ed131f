"""
Gaussian likelihood -- host-side mirror of mogptk/gpr/likelihood.py:312-378 (the only likelihood exact
inference uses).  The noise term sigma_c^2 is added to the Gram diagonal on the device.
"""
import numpy as np
from math import erf, sqrt

from .config import config
from .parameter import Parameter, ParameterHolder


class Likelihood(ParameterHolder):
    def __init__(self):
        self.output_dims = None

    def name(self):
        return self.__class__.__name__

    def validate_y(self, X, y):
        pass


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
        return ve, (y - mu) / s ** 2, np.full(y.shape, -0.5 / s ** 2), np.sum(r2) / s ** 3 - y.size / s

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
