"""
GP inference models on the HIP path -- host-side mirror of mogptk/gpr/model.py:71-483 (CholeskyException,
Model, Exact).  The O(N^2)/O(N^3) work of every method below is one call into libmogp_hip.so; the host does
the O(C^2 Q) parameter algebra (term table, chain rule) in numpy.
"""
import sys
import numpy as np
from math import erf, sqrt

from .config import config
from .parameter import Parameter, ParameterHolder
from .kernel import Kernel
from .likelihood import Likelihood, GaussianLikelihood


class CholeskyException(Exception):
    """reference gpr/model.py:71-78"""

    def __init__(self, message, K, model):
        self.message = message
        self.K = K
        self.model = model

    def __str__(self):
        return self.message


def _to_array(X):
    if hasattr(X, "detach"):
        X = X.detach().cpu().numpy()
    return np.array(X, dtype=np.float64)


class Model(ParameterHolder):
    """Base model (reference gpr/model.py:80-401)."""

    def __init__(self, kernel, X, y, likelihood=None, jitter=1e-8, mean=None):
        if likelihood is None:
            likelihood = GaussianLikelihood(1.0)
        if not issubclass(type(kernel), Kernel):
            raise ValueError("kernel must derive from mogptk_amd.gpr.Kernel")
        X, y = self._check_input(X, y)
        if mean is not None:
            mu = np.asarray(mean(X)).reshape(-1, 1)
            if mu.shape != y.shape:
                raise ValueError("mean and y data must match shapes: %s != %s" % (mu.shape, y.shape))
            if any(True for _ in getattr(mean, "parameters", lambda: [])()):
                raise NotImplementedError("trainable mean functions are not on the HIP path")
        if likelihood.output_dims is not None and likelihood.output_dims != kernel.output_dims:
            raise ValueError("kernel and likelihood must have matching output dimensions")
        likelihood.validate_y(X, y)

        # limit to number of significant digits (reference gpr/model.py:106-110)
        jitter = max(jitter, 1e-15)

        self.kernel = kernel
        self.X = X
        self.y = y
        self.mean = mean
        self.likelihood = likelihood
        self.jitter = jitter
        self.input_dims = X.shape[1]
        self._handle = None

    def name(self):
        return self.__class__.__name__

    def _get_name(self):
        return self.__class__.__name__

    def __getstate__(self):
        """device handles are never pickled; they are rebuilt lazily (reference gpr/model.py:131-136 drops
        the traced forward the same way)"""
        state = self.__dict__.copy()
        state["_handle"] = None
        return state

    def _check_input(self, X, y=None):
        """reference gpr/model.py:149-181"""
        X = _to_array(X)
        if X.ndim == 0:
            X = X.reshape(1, 1)
        elif X.ndim == 1:
            X = X.reshape(-1, 1)
        elif X.ndim != 2:
            raise ValueError("X must have dimensions (data_points,input_dims) with input_dims optional")
        if X.shape[0] == 0 or X.shape[1] == 0:
            raise ValueError("X must not be empty")
        if y is not None:
            y = _to_array(y)
            if y.ndim == 0:
                y = y.reshape(1, 1)
            elif y.ndim == 1:
                y = y.reshape(-1, 1)
            elif y.ndim != 2 or y.shape[1] != 1:
                raise ValueError("y must have one dimension (data_points,)")
            if X.shape[0] != y.shape[0]:
                raise ValueError("number of data points for X and y must match")
            return X, y
        if X.shape[1] != self.input_dims:
            raise ValueError("X must have %s input dimensions" % self.input_dims)
        return X

    def print_parameters(self, file=None):
        """reference gpr/model.py:188-240 (plain-text branch)"""
        vals = [["Name", "Range", "Value"]]
        for p in self.parameters():
            vals.append([str(p._name), "", p.numpy().tolist()])
        nameWidth = max(len(val[0]) for val in vals)
        for val in vals:
            print("%-*s  %s" % (nameWidth, val[0], val[2]), file=file)

    def log_marginal_likelihood(self):
        raise NotImplementedError()

    def log_prior(self):
        """reference gpr/model.py:268-277"""
        return sum(p.log_prior() for p in self.parameters())

    def forward(self, x=None):
        """reference gpr/model.py:124-125"""
        return -self.log_marginal_likelihood() - self.log_prior()

    def compile(self):
        """reference gpr/model.py:127-129 traces the forward with torch.jit; the HIP path is already one
        native call per evaluation, so this is accepted and ignored."""
        pass

    def loss(self):
        raise NotImplementedError()

    def K(self, X1, X2=None):
        """reference gpr/model.py:294-306"""
        return self.kernel(X1, X2)

    def predict_f(self, X, full=False):
        raise NotImplementedError()

    def predict_y(self, X, ci=None, sigma=None, n=10000):
        """reference gpr/model.py:322-344"""
        X = self._check_input(X)
        mu, var = self.predict_f(X)
        if ci is None and sigma is not None:
            p = 0.5 * (1.0 + erf(sigma / sqrt(2.0)))
            ci = [1.0 - p, p]
        return self.likelihood.predict(self._likelihood_X(X), mu, var, ci, sigma=sigma, n=n)

    def _likelihood_X(self, X):
        return X


class Exact(Model):
    """
    Exact GP regression with a Gaussian likelihood (reference gpr/model.py:403-483):
        y ~ N(0, K + sigma^2 I)
    `variance` is a float (one trained scale) or a (channels,) array (one per channel).
    """

    def __init__(self, kernel, X, y, variance=1.0, data_variance=None, jitter=1e-8, mean=None):
        if data_variance is not None:
            data_variance = Parameter.to_tensor(data_variance)
            Xa = _to_array(X)
            if data_variance.ndim != 1 or Xa.ndim == 2 and data_variance.shape[0] != Xa.shape[0]:
                raise ValueError("data variance must have shape (data_points,)")
        self.data_variance = data_variance

        variance = Parameter.to_tensor(variance)
        channels = 1
        if kernel.output_dims is not None:
            channels = kernel.output_dims
        if 1 < variance.ndim or variance.ndim == 1 and variance.shape[0] != channels:
            raise ValueError("variance must be float or have shape (channels,)")

        super().__init__(kernel, X, y, GaussianLikelihood(np.sqrt(variance)), jitter, mean)
        self.log_marginal_likelihood_constant = 0.5 * self.X.shape[0] * np.log(2.0 * np.pi)

    # -- device plumbing ---------------------------------------------------------------------
    def _device_handle(self):
        if self._handle is None:
            from .._lib import ExactHandle
            y = self.y if self.mean is None else self.y - np.asarray(self.mean(self.X)).reshape(-1, 1)
            self._handle = ExactHandle(config.device, self.kernel._kernel_format(self.X), y, self.kernel._channels())
        return self._handle

    def _noise_var(self):
        """sigma_c^2 per channel: the vector `_index_channel` (gpr/model.py:183-186) would gather from"""
        s2 = np.square(self.likelihood.scale())
        C = self.kernel._channels()
        if s2.ndim == 1 and s2.shape[0] == C and self.kernel.output_dims is not None:
            return s2
        return np.repeat(np.asarray(s2).reshape(-1)[0], C)

    def _push_terms(self):
        h = self._device_handle()
        D = self.X.shape[1] - (0 if self.kernel.output_dims is None else 1)
        table = self.kernel._spectral_terms(D)
        h.set_terms(table)
        return h, table, D

    def _eval(self, grad):
        from .._lib import MogpError, MOGP_ENOTPD, MOGP_ENONFINITE
        h, table, D = self._push_terms()
        try:
            return h.eval(self._noise_var(), self.jitter, grad=grad, data_var=self.data_variance), table, D
        except MogpError as e:
            if e.code in (MOGP_ENOTPD, MOGP_ENONFINITE):
                # reference gpr/model.py:245-255: report, dump parameters, raise CholeskyException(msg, K, model)
                print("ERROR:", str(e), file=sys.__stdout__)
                self.print_parameters()
                raise CholeskyException(str(e), None, self)
            raise

    # -- reference surface -------------------------------------------------------------------
    def log_marginal_likelihood(self):
        """reference gpr/model.py:438-453 -- one forward-only device evaluation"""
        res, _, _ = self._eval(grad=False)
        return np.float64(res["lml"])

    def loss(self):
        """reference gpr/model.py:279-292: zero grads, loss = -LML - log prior, fresh `.grad` on every
        parameter in the graph.  Gradients come from the device's moment pass + the host chain rule."""
        self.zero_grad(set_to_none=True)
        res, table, D = self._eval(grad=True)
        h = self._handle
        C, T = h.C, h.T
        N = self.X.shape[0]
        W = 2 + 3 * D
        mom = res["moments"]
        counts = np.bincount(self.kernel._kernel_format(self.X)[:, 0].astype(np.int64), minlength=C).astype(np.float64)
        jit_rel = self.jitter * res["trG"] / N            # d LML / d (mean diag) through the jitter term (:244)

        # d LML / d table for the lower channel pairs (i >= j); zero elsewhere
        gt = np.zeros((C, C, T, W))
        for i in range(C):
            for j in range(i + 1):
                m = mom[i * (i + 1) // 2 + j]                   # (T, W): [m0, m4, m1_d, m2_d, m3_d]
                tb = table[i, j]
                A = tb[:, 0]
                V = tb[:, 2:2 + D]
                M = tb[:, 2 + D:2 + 2 * D]
                m0, m4 = m[:, 0], m[:, 1]
                m1, m2, m3 = m[:, 2:2 + D], m[:, 2 + D:2 + 2 * D], m[:, 2 + 2 * D:]
                gt[i, j, :, 0] = m0 + (jit_rel * counts[i] if i == j else 0.0)
                gt[i, j, :, 1] = -2.0 * np.pi * A * m4
                gt[i, j, :, 2:2 + D] = -0.5 * A[:, None] * m1
                gt[i, j, :, 2 + D:2 + 2 * D] = -2.0 * np.pi * A[:, None] * m3
                gt[i, j, :, 2 + 2 * D:] = -V * A[:, None] * m2 - 2.0 * np.pi * M * (A * m4)[:, None]
        self.kernel._spectral_backward(-gt)                    # loss = -LML

        # noise: d LML / d sigma_c = 2 sigma_c (sum_{k in c} G_kk + jitter n_c/N tr G)
        scale = self.likelihood.scale
        sc = scale()
        gnoise = res["diagG"] + jit_rel * counts
        if sc.ndim == 1 and sc.shape[0] == C and self.kernel.output_dims is not None:
            gsc = 2.0 * sc * gnoise
        else:
            gsc = np.reshape(2.0 * sc * np.sum(gnoise), sc.shape)
        scale.grad = -gsc * scale.dconstrained()
        return np.float64(-res["lml"] - self.log_prior())

    def predict_f(self, X, full=False):
        """reference gpr/model.py:455-483"""
        from .._lib import MogpError, MOGP_ENOTPD, MOGP_ENONFINITE
        X = self._check_input(X)
        h, _, D = self._push_terms()
        try:
            mu, var = h.predict(self._noise_var(), self.jitter, self.kernel._spectral_diag(D),
                                self.kernel._kernel_format(X), full=full,
                                data_var=self.data_variance)
        except MogpError as e:
            if e.code in (MOGP_ENOTPD, MOGP_ENONFINITE):
                print("ERROR:", str(e), file=sys.__stdout__)
                self.print_parameters()
                raise CholeskyException(str(e), None, self)
            raise
        if self.mean is not None:
            mu = mu + np.asarray(self.mean(X)).reshape(-1, 1)
        return mu, var
