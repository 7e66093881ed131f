"""
Single-output spectral kernels on the HIP path -- host-side mirror of mogptk/gpr/singleoutput.py for
SpectralKernel (:520-561) and SpectralMixtureKernel (:563-605).  The other stationary kernels of the
reference are not named by the hot path and are not provided.
"""
import numpy as np

from .config import config
from .parameter import Parameter
from .kernel import Kernel, term_width, cached_terms
from .multioutput import _accumulate

FOUR_PI2 = 4.0 * np.pi ** 2


class SpectralMixtureKernel(Kernel):
    """
    K = sum_q sum_d mag_q exp(-2 pi^2 tau_d^2 v_qd) cos(2 pi tau_d mu_qd)   (reference :594-600; the einsum
    SUMS over the input dimension d).  Each (q, d) is one spectral term that touches dimension d only.
    Parameters: magnitude (Q,), mean (Q,D), variance (Q,D).
    """

    def __init__(self, Q=1, input_dims=1, active_dims=None):
        super().__init__(input_dims, active_dims)
        self.magnitude = Parameter(np.ones(Q), lower=config.positive_minimum)
        self.mean = Parameter(np.zeros((Q, input_dims)), lower=config.positive_minimum)
        self.variance = Parameter(np.ones((Q, input_dims)), lower=config.positive_minimum)

    @cached_terms
    def _spectral_terms(self, D):
        if D != self.input_dims:
            raise ValueError("X must have %d input dimensions" % self.input_dims)
        mag, mu, var = self.magnitude(), self.mean(), self.variance()
        Q = mag.shape[0]
        table = np.zeros((1, 1, Q * D, term_width(D)))
        for q in range(Q):
            for d in range(D):
                t = q * D + d
                table[0, 0, t, 0] = mag[q]
                table[0, 0, t, 2 + d] = FOUR_PI2 * var[q, d]          # exp(-1/2 V u^2) with V = 4 pi^2 v
                table[0, 0, t, 2 + D + d] = mu[q, d]
        return table

    def _spectral_diag(self, D):
        return np.array([np.sum(self.magnitude())])                      # reference :602-605 (not x D)

    def _spectral_diag_backward(self, gc, D):
        _accumulate(self.magnitude, np.full(self.magnitude.shape, float(gc[0])))

    def _spectral_backward(self, gtable):
        D = self.input_dims
        Q = self.magnitude.shape[0]
        g = gtable[0, 0].reshape(Q, D, term_width(D))
        idx = np.arange(D)
        _accumulate(self.magnitude, np.sum(g[:, :, 0], axis=1))
        _accumulate(self.mean, g[:, idx, 2 + D + idx])
        _accumulate(self.variance, FOUR_PI2 * g[:, idx, 2 + idx])


class SpectralKernel(Kernel):
    """K = mag sum_d exp(-2 pi^2 tau_d^2 v_d) cos(2 pi tau_d mu_d)   (reference :555-561)."""

    def __init__(self, input_dims=1, active_dims=None):
        super().__init__(input_dims, active_dims)
        self.magnitude = Parameter(1.0, lower=config.positive_minimum)
        self.mean = Parameter(np.zeros(input_dims), lower=config.positive_minimum)
        self.variance = Parameter(np.ones(input_dims), lower=config.positive_minimum)

    @cached_terms
    def _spectral_terms(self, D):
        if D != self.input_dims:
            raise ValueError("X must have %d input dimensions" % self.input_dims)
        mag, mu, var = self.magnitude(), self.mean(), self.variance()
        table = np.zeros((1, 1, D, term_width(D)))
        for d in range(D):
            table[0, 0, d, 0] = mag
            table[0, 0, d, 2 + d] = FOUR_PI2 * var[d]
            table[0, 0, d, 2 + D + d] = mu[d]
        return table

    def _spectral_diag(self, D):
        return np.array([float(self.magnitude())])                       # reference :558-561

    def _spectral_diag_backward(self, gc, D):
        _accumulate(self.magnitude, np.reshape(float(gc[0]), self.magnitude.shape))

    def _spectral_backward(self, gtable):
        D = self.input_dims
        g = gtable[0, 0]
        idx = np.arange(D)
        _accumulate(self.magnitude, np.sum(g[:, 0]))
        _accumulate(self.mean, g[idx, 2 + D + idx])
        _accumulate(self.variance, FOUR_PI2 * g[idx, 2 + idx])
