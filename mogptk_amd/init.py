"""
BNSE -- Bayesian non-parametric spectral estimation (reference mogptk/init.py:5-122), the default `init_parameters` method.

The expensive part is a Gaussian-process fit: an `Exact` model with one `SpectralKernel` trained by `iters` Adam steps (lr = 2)
-- that runs on the device through the normal hot path.  The posterior of the spectrum needs the Cholesky factor of the trained
model against two closed-form cross-covariances (time-frequency, frequency-frequency) that are not spectral-kernel Grams; the
device hands back W = L^-1 and alpha = K^-1 y of its last factorisation (`mogp_model_fetch`) and the O(N^2 n) products with the
n-point frequency grid are formed on the host.
"""
import numpy as np

from . import gpr


def BNSE(x, y, y_err=None, max_freq=None, n=1000, iters=100, jit=True):
    """
    Power spectral density of a signal by BNSE [Tobar 2018]; reference init.py:5-122 (same arguments; `jit` is accepted and ignored).

    Returns:
        numpy.ndarray: frequencies (n,), PSD mean (n,), PSD variance (n,).
    """
    from .model import _Adam
    x -= np.median(x)                      # in place, like the reference (init.py:24)
    x_range = np.max(x) - np.min(x)
    x_dist = x_range / len(x)
    if max_freq is None:
        max_freq = 0.5 / x_dist
    x = np.asarray(x, dtype=np.float64)
    x = x.reshape(1, 1) if x.ndim == 0 else (x.reshape(-1, 1) if x.ndim == 1 else x)
    y = np.asarray(y, dtype=np.float64).reshape(-1, 1)

    kernel = gpr.SpectralKernel()
    model = gpr.Exact(kernel, x, y, data_variance=y_err ** 2 if y_err is not None else None)

    # initial values (init.py:41-48); torch's var()/std() are the unbiased estimators
    model.kernel.magnitude.assign(np.var(y, ddof=1))
    model.kernel.mean.assign(0.01, upper=max_freq)
    model.kernel.variance.assign(0.25 / np.pi ** 2 / x_dist ** 2)
    model.likelihood.scale.assign(np.std(y, ddof=1) / 10.0)

    optimizer = _Adam(list(model.parameters()), lr=2.0)
    for _ in range(iters):
        model.loss()
        optimizer.step()

    alpha_w = float(0.5 / x_range ** 2)
    w = np.linspace(0.0, max_freq, n).reshape(-1, 1)
    magnitude, mean, variance = model.kernel.magnitude(), model.kernel.mean().reshape(-1), model.kernel.variance().reshape(-1)

    def kernel_ff(f1, f2):
        gamma = 2.0 * np.pi ** 2 * variance.reshape(1, 1, -1)
        const = 0.5 * np.pi * magnitude / np.sqrt(alpha_w ** 2 + 2.0 * alpha_w * np.prod(gamma))
        d2 = (f1[:, None, :] - f2[None, :, :]) ** 2
        avg = 0.5 * (f1[:, None, :] + f2[None, :, :])
        e1 = -0.5 * np.pi ** 2 / alpha_w * d2
        e2a = -2.0 * np.pi ** 2 / (alpha_w + 2.0 * gamma) * (avg - mean.reshape(1, 1, -1)) ** 2
        e2b = -2.0 * np.pi ** 2 / (alpha_w + 2.0 * gamma) * (avg + mean.reshape(1, 1, -1)) ** 2
        return const * np.sum(np.exp(e1 + e2a) + np.exp(e1 + e2b), axis=2)

    def kernel_tf(t, f):
        mu = mean.reshape(1, -1)
        gamma = 2.0 * np.pi ** 2 * variance.reshape(1, -1)
        Lq_inv = 1.0 / (np.pi ** 2 * (1.0 / alpha_w + 1.0 / gamma))          # init.py:81-82
        const = np.sqrt(np.pi / (alpha_w + np.prod(gamma)))
        e1 = -np.pi ** 2 * (t ** 2) @ Lq_inv.T
        e2a = -(np.pi ** 2 / (alpha_w + gamma)) @ ((f - mu).T ** 2)
        e2b = -(np.pi ** 2 / (alpha_w + gamma)) @ ((f + mu).T ** 2)
        e3a = -2.0 * np.pi * (t @ Lq_inv) @ (np.pi ** 2 * (f / alpha_w + mu / gamma).T)
        e3b = -2.0 * np.pi * (t @ Lq_inv) @ (np.pi ** 2 * (f / alpha_w - mu / gamma).T)
        a = 0.5 * magnitude * const * np.exp(e1)
        real = np.exp(e2a) * np.cos(e3a) + np.exp(e2b) * np.cos(e3b)
        imag = np.exp(e2a) * np.sin(e3a) + np.exp(e2b) * np.sin(e3b)
        return a * real, a * imag

    # factor K_tt + sigma^2 I (+ relative jitter) at the trained parameters on the device; W = L^-1 and alpha = K^-1 y come back.
    # The observation variances took part in the fit only: the reference leaves them out of this matrix (init.py:96-98).
    model.data_variance = None
    model.log_marginal_likelihood()
    W = model._handle.fetch(0)
    a = model._handle.fetch(2).reshape(-1, 1)

    Kff, Pff = kernel_ff(w, w), kernel_ff(w, -w)
    Kff_real, Kff_imag = 0.5 * (Kff + Pff), 0.5 * (Kff - Pff)
    Ktf_real, Ktf_imag = kernel_tf(x, w)
    b, c = W @ Ktf_real, W @ Ktf_imag
    mu_real, mu_imag = Ktf_real.T @ a, Ktf_imag.T @ a
    var_real = (np.diagonal(Kff_real) - np.sum(b * b, axis=0)).reshape(-1, 1)
    var_imag = (np.diagonal(Kff_imag) - np.sum(c * c, axis=0)).reshape(-1, 1)
    # the PSD is N(mu_real, var_real)^2 + N(mu_imag, var_imag)^2: a generalised chi-squared distribution
    mu = mu_real ** 2 + mu_imag ** 2 + var_real + var_imag
    var = 2.0 * var_real ** 2 + 2.0 * var_imag ** 2 + 4.0 * var_real * mu_real ** 2 + 4.0 * var_imag * mu_imag ** 2
    return w.reshape(-1), mu.reshape(-1), var.reshape(-1)
