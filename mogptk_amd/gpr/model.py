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
from .kernel import Kernel, terms_cache
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
    return np.array(X, dtype=config.dtype)


_PAIR_INDEX = {}


def _pair_index(C, lower):
    key = (C, lower)
    if key not in _PAIR_INDEX:
        _PAIR_INDEX[key] = np.tril_indices(C) if lower else tuple(np.divmod(np.arange(C * C), C))
    return _PAIR_INDEX[key]


def _gtable_from_moments(table, mom, D, lower=True):
    """d(objective)/d(term table) from the device's gradient moments [m0, m4, m1_d, m2_d, m3_d]:
    dA = m0, dPsi = -2 pi A m4, dV_d = -1/2 A m1_d, dM_d = -2 pi A m3_d, dDelta_d = -V_d A m2_d - 2 pi M_d A m4.
    lower=True: mom is indexed by lower channel pairs p = i(i+1)/2 + j (symmetric Gram, double count already
    included); lower=False: mom is indexed by all ordered pairs i*C + j (rectangular Gram)."""
    C, T = table.shape[0], table.shape[2]
    gt = np.zeros((C, C, T, table.shape[3]))
    ii, jj = _pair_index(C, lower)                      # lower: row-major lower pairs, exactly p = i(i+1)/2 + j
    tb = table[ii, jj]                                  # (P, T, W)
    A = tb[..., 0]
    V = tb[..., 2:2 + D]
    M = tb[..., 2 + D:2 + 2 * D]
    m0, m4 = mom[..., 0], mom[..., 1]
    m1, m2, m3 = mom[..., 2:2 + D], mom[..., 2 + D:2 + 2 * D], mom[..., 2 + 2 * D:2 + 3 * D]
    g = np.empty_like(tb)
    g[..., 0] = m0
    g[..., 1] = -2.0 * np.pi * A * m4
    g[..., 2:2 + D] = -0.5 * A[..., None] * m1
    g[..., 2 + D:2 + 2 * D] = -2.0 * np.pi * A[..., None] * m3
    g[..., 2 + 2 * D:2 + 3 * D] = -V * A[..., None] * m2 - 2.0 * np.pi * M * (A * m4)[..., None]
    if tb.shape[-1] > 2 + 3 * D:          # envelope exp(-1/2 L a^2), a = midpoint - c:  dL = -1/2 A m5,  dc = L A m6
        Lv = tb[..., 2 + 3 * D:2 + 4 * D]
        g[..., 2 + 3 * D:2 + 4 * D] = -0.5 * A[..., None] * mom[..., 2 + 3 * D:2 + 4 * D]
        g[..., 2 + 4 * D:2 + 5 * D] = Lv * A[..., None] * mom[..., 2 + 4 * D:2 + 5 * D]
    gt[ii, jj] = g
    return gt


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
        jitter = max(jitter, 1e-6 if config.dtype == np.float32 else 1e-15)

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
        total = 0.0
        for p in self._parameter_list():
            if p.prior is not None:
                total = total + p.log_prior()
        return total

    def forward(self, x=None):
        """reference gpr/model.py:124-125"""
        return -self.log_marginal_likelihood() - self.log_prior()

    def compile(self):
        """reference gpr/model.py:127-129 traces the forward with torch.jit; the HIP path is already one
        native call per evaluation, so this is accepted and ignored."""
        pass

    def loss(self):
        with terms_cache():                 # parameters are frozen between the table push and the chain rule
            return self._loss_impl()

    def _loss_impl(self):
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

    def sample_f(self, Z, n=None, prior=False):
        """Samples of f at Z (reference gpr/model.py:346-376): from the posterior (`predict_f(Z, full=True)`, the device path) or the prior,
        drawn by torch's MultivariateNormal from torch's global generator exactly as the reference does -- the same seed gives the same
        samples.  Shape as the reference returns it: (n, data_points), or (data_points,) without n."""
        from .likelihood import _torch
        torch = _torch()
        Z = self._check_input(Z)
        S = 1 if n is None else n
        if prior:
            if self.mean is None:
                raise TypeError("sampling from the prior needs a mean function (the reference calls self.mean(Z), gpr/model.py:364)")
            mu, var = np.asarray(self.mean(Z), dtype=np.float64), np.array(self.kernel(Z), dtype=np.float64)
        else:
            mu, var = self.predict_f(Z, full=True)
        var = np.array(var, dtype=np.float64)
        var += self.jitter * np.mean(np.diagonal(var)) * np.eye(var.shape[0])
        dist = torch.distributions.multivariate_normal.MultivariateNormal(torch.tensor(np.reshape(mu, -1), dtype=torch.float64), torch.tensor(var))
        samples = dist.sample([S])
        if n is None:
            samples = samples.squeeze()
        return samples.numpy()

    def sample_y(self, Z, n=None):
        """Samples of y at Z (reference gpr/model.py:378-401): samples of f pushed through the likelihood's sampler"""
        from .likelihood import _torch
        torch = _torch()
        Z = self._check_input(Z)
        S = 1 if n is None else n
        samples_f = torch.tensor(self.sample_f(Z, n=S))
        samples_y = self.likelihood.conditional_sample(self._likelihood_X(Z), samples_f)
        if n is None:
            samples_y = samples_y.squeeze()
        return samples_y.numpy()


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
        if table.shape[3] > 2 + 3 * D:        # envelope: the diagonal varies from point to point and enters the relative jitter (:244)
            h.set_point_diag(self.kernel._point_diag(table, self.kernel._kernel_format(self.X), D))
        return h, table, D

    def _eval(self, grad):
        from .._lib import MogpError, MOGP_ENOTPD, MOGP_ENONFINITE
        h, table, D = self._push_terms()
        try:
            run = lambda: h.eval(self._noise_var(), self.jitter, grad=grad, data_var=self.data_variance)
            res = run()
            again = self._check_conditioning(h, run) if grad else None          # (the LML alone: the fast factorisation's log-determinant and z are good to ~1e-7 even at 1e8)
            return (res if again is None else again), table, D
        except MogpError as e:
            if e.code in (MOGP_ENOTPD, MOGP_ENONFINITE):
                # reference gpr/model.py:245-255: report, dump parameters, raise CholeskyException(msg, K, model)
                print("ERROR:", str(e), file=sys.__stdout__)
                self.print_parameters()
                raise CholeskyException(str(e), None, self)
            raise

    CONDITION_WARN = 1e5       # on the pivot-spread estimate (max L_jj / min L_jj)^2, a LOWER bound of cond(Kj) -- 9e5 where cond is 7e7, 9e3 where it is 8e5;
                               # DESIGN 7 puts the envelope of the fast schedules at cond ~ 1e6 - 1e7

    def _check_conditioning(self, h, redo):
        """The reference's torch.linalg.cholesky is backward stable and silent; the fast schedules of this path form their panels with explicit block
        inverses and lose accuracy as K + noise becomes ill-conditioned.  The factor's own diagonal says when: the model then says so
        (once), repeats the evaluation in the backward-stable form (mogp_model_set_accurate; 25 ms against 10 at N = 8192) and stays there until the matrix is
        well-conditioned again.  `config.accurate_fallback = False` keeps the fast form and only warns."""
        if not hasattr(h, "condition_estimate"):
            return None
        est = h.condition_estimate()
        if est != est:
            return None
        accurate = getattr(h, "accurate_mode", False)       # the mode lives in the device handle, so the flag lives ON the handle: a model that is copied,
        if not accurate and est > self.CONDITION_WARN:      # reloaded or re-sharded gets a new handle in the fast mode and a flag that says so
            fallback = getattr(config, "accurate_fallback", True) and hasattr(h, "set_accurate")
            if not getattr(self, "_cond_warned", False):
                self._cond_warned = True
                import warnings
                warnings.warn("the kernel matrix plus noise is ill-conditioned (cond >= %.1e from the Cholesky factor's diagonal): beyond ~1e6 - 1e7 the "
                              "fast schedules' LML and gradients leave a backward-stable factorisation's by more than 1e-9 / 1e-5 (at 1e8: ~1e-7 / ~1e-4); %s"
                              % (est, "evaluating in the backward-stable form from here on (about two and a half times slower)" if fallback
                                 else "a larger noise variance or jitter brings it back"), RuntimeWarning, stacklevel=5)
            if fallback:
                h.set_accurate(True)
                h.accurate_mode = True
                return redo()
        elif accurate and est < 0.1 * self.CONDITION_WARN:
            h.set_accurate(False)                     # the next evaluation is a fast one again
            h.accurate_mode = False
        return None

    # -- reference surface -------------------------------------------------------------------
    def log_marginal_likelihood(self):
        """reference gpr/model.py:438-453 -- one forward-only device evaluation"""
        res, _, _ = self._eval(grad=False)
        return config.dtype(res["lml"])

    def _loss_impl(self):
        """reference gpr/model.py:279-292: zero grads, loss = -LML - log prior, fresh `.grad` on every
        parameter in the graph.  Gradients come from the device's moment pass + the host chain rule."""
        self.zero_grad(set_to_none=True)
        res, table, D = self._eval(grad=True)
        h = self._handle
        C, T = h.C, h.T
        N = self.X.shape[0]
        W = 2 + 3 * D
        mom = res["moments"]
        if getattr(self, "_counts", None) is None or len(self._counts) != C:     # X is fixed per model (reference gpr/model.py:113-118)
            self._counts = np.bincount(self.kernel._kernel_format(self.X)[:, 0].astype(np.int64), minlength=C).astype(np.float64)
        counts = self._counts
        jit_rel = self.jitter * res["trG"] / N            # d LML / d (mean diag) through the jitter term (:244)

        # d LML / d table for the lower channel pairs (i >= j); zero elsewhere
        gt = _gtable_from_moments(table, mom, D, lower=True)
        if table.shape[3] > 2 + 3 * D:
            gt += jit_rel * self.kernel._point_diag_table_grad(table, self.kernel._kernel_format(self.X), D)
        else:
            for i in range(C):
                gt[i, i, :, 0] += jit_rel * counts[i]
        self.kernel._spectral_backward(-gt)                    # loss = -LML

        # noise: d LML / d sigma_c = 2 sigma_c (sum_{k in c} G_kk + jitter n_c/N tr G)
        scale = self.likelihood.scale
        sc = scale()
        gnoise = res["diagG"] + jit_rel * counts
        if sc.ndim == 1 and sc.shape[0] == C and self.kernel.output_dims is not None:
            gsc = 2.0 * sc * gnoise
        else:
            gsc = np.reshape(2.0 * sc * np.sum(gnoise), sc.shape)
        scale.accumulate_grad(-gsc)
        return config.dtype(-res["lml"] - self.log_prior())

    def predict_f(self, X, full=False):
        """reference gpr/model.py:455-483"""
        from .._lib import MogpError, MOGP_ENOTPD, MOGP_ENONFINITE
        X = self._check_input(X)
        h, table, D = self._push_terms()
        Xk = self.kernel._kernel_format(X)
        kss = self.kernel._point_diag(table, Xk, D) if table.shape[3] > 2 + 3 * D else self.kernel._spectral_diag(D)
        try:
            run = lambda: h.predict(self._noise_var(), self.jitter, kss, Xk, full=full, data_var=self.data_variance)
            mu, var = run()
            again = self._check_conditioning(h, run)           # an ill-conditioned system: said once, predicted again in the refined form
            if again is not None:
                mu, var = again
        except MogpError as e:
            if e.code in (MOGP_ENOTPD, MOGP_ENONFINITE):
                print("ERROR:", str(e), file=sys.__stdout__)
                self.print_parameters()
                raise CholeskyException(str(e), None, self)
            raise
        if self.mean is not None:
            mu = mu + np.asarray(self.mean(X)).reshape(-1, 1)
        return mu.astype(config.dtype, copy=False), var.astype(config.dtype, copy=False)


# ---- inducing-point initialisation (reference gpr/model.py:11-69) ---------------------------------------------------
def _linspace_f32(lo, hi, n):
    """torch.linspace(lo, hi, n) in the reference runs in torch's default dtype float32 (quirk Q6, gpr/model.py:18):
    step = (end - start)/(n - 1) rounded to float32; first half start + i*step, second half end - (n-1-i)*step, each
    multiply-add rounded ONCE to float32 (torch's kernel fuses it; verified bit-exact against torch.linspace)."""
    start, end = np.float32(lo), np.float32(hi)
    if n == 1:
        return np.array([start], dtype=np.float64)
    step = np.float64(np.float32((end - start) / np.float32(n - 1)))
    i = np.arange(n)
    half = n // 2
    out = np.where(i < half, np.float64(start) + step * i, np.float64(end) - step * (n - i - 1))
    return out.astype(np.float32).astype(np.float64)


def _init_grid(N, X):
    n = np.power(N, 1.0 / X.shape[1])
    if not float(n).is_integer():
        raise ValueError("number of inducing points must equal N = n^%d" % X.shape[1])
    n = int(n)
    axes = [_linspace_f32(np.min(X[:, i]), np.max(X[:, i]), n) for i in range(X.shape[1])]
    grid = np.meshgrid(*axes, indexing="ij")
    return np.stack([g.reshape(-1) for g in grid], axis=1)


def _init_random(N, X):
    from scipy.stats import qmc
    samples = qmc.Halton(d=X.shape[1]).random(n=N)
    lo, hi = np.min(X, axis=0), np.max(X, axis=0)
    return lo + (hi - lo) * samples


def _init_density(N, X):
    from scipy.stats import gaussian_kde
    return gaussian_kde(X.T, bw_method="scott").resample(N).T


def init_inducing_points(Z, X, method="grid", output_dims=None):
    """reference gpr/model.py:36-69.  Z int / list of ints -> locations; an int means PER CHANNEL for multi-output
    kernels (quirk Q5).  Values pass through float32 like the reference's default-dtype tensor (quirk Q6)."""
    _init = {"grid": _init_grid, "random": _init_random, "density": _init_density}.get(method, _init_grid)
    if output_dims is not None:
        if isinstance(Z, (int, np.integer)) or (all(isinstance(z, (int, np.integer)) for z in Z) and len(Z) == output_dims):
            M = [int(Z)] * output_dims if isinstance(Z, (int, np.integer)) else [int(z) for z in Z]
            out = np.zeros((sum(M), X.shape[1]), dtype=np.float32)
            for j in range(len(M)):
                m0 = sum(M[:j])
                out[m0:m0 + M[j], 0] = j
                out[m0:m0 + M[j], 1:] = _init(M[j], X[X[:, 0] == j, 1:])
            return out.astype(np.float64)
    elif isinstance(Z, (int, np.integer)):
        return _init(int(Z), X)
    return Z


class _DataParallel:
    """Sparse models: the objective touches the training points only through sums over them.  Under `mogptk_amd.use_distributed()` every
    process builds its device model on every world-th training point (starting at its rank) and the library all-reduces those sums
    (mogp_titsias_eval_sharded, mogp_snelson_eval_sharded, mogp_svgp_backward_sharded): each rank gets the full model's value and gradient."""

    def _shardable(self):
        return True

    def _data_shard(self):
        """the communicator this process shards its training points over, or None"""
        comm = getattr(config, "comm", None)
        if self._shardable() and comm is not None and getattr(comm, "native", False) and (comm.world > 1 or comm.force):
            return comm
        return None

    def _local(self, a):
        """this rank's share of a per-point array (all of it without a communicator)"""
        comm = self._data_shard()
        return a if comm is None else a[comm.rank::comm.world]

    def _device_handle(self):
        comm = self._data_shard()
        key = None if comm is None else (comm.rank, comm.world)
        if self._handle is None or self.__dict__.get("_handle_key") != key:
            from .._lib import ExactHandle
            y = self.y if self.mean is None else self.y - np.asarray(self.mean(self.X)).reshape(-1, 1)
            self._handle = ExactHandle(config.device, self._local(self.kernel._kernel_format(self.X)), self._local(y), self.kernel._channels())
            self.__dict__["_handle_key"] = key
        return self._handle


class Titsias(_DataParallel, Model):
    """
    Sparse GP regression, Titsias 2009 (reference gpr/model.py:668-765): the bound
        ELBO = log N(y | 0, Kfu Kuu^-1 Kuf + s2 I) - tr(Kff - Kfu Kuu^-1 Kuf) / (2 s2)
    with trainable inducing inputs `Z` (the channel column of Z carries no gradient) and a SCALAR noise scale.
    """

    def __init__(self, kernel, X, y, Z, Z_init="grid", variance=1.0, jitter=1e-8, mean=None):
        variance = Parameter.to_tensor(variance)
        super().__init__(kernel, X, y, GaussianLikelihood(np.sqrt(variance)), jitter, mean)
        Z = init_inducing_points(Z, self.X, method=Z_init, output_dims=kernel.output_dims)
        Z = self._check_input(Z)
        self.log_marginal_likelihood_constant = 0.5 * self.X.shape[0] * np.log(2.0 * np.pi)
        self.Z = Parameter(Z, name="induction_points")
        if kernel.output_dims is not None:
            self.Z.num_parameters -= self.Z().shape[0]

    def _sigma(self):
        s = np.asarray(self.likelihood.scale())
        if s.ndim != 0 and s.size != 1:
            raise ValueError("Titsias takes a scalar noise variance (reference gpr/model.py:686-689)")
        return float(s.reshape(-1)[0])

    def _run(self, grad):
        from .._lib import MogpError, MOGP_ENOTPD, MOGP_ENONFINITE
        h = self._device_handle()
        D = self.X.shape[1] - (0 if self.kernel.output_dims is None else 1)
        table = self.kernel._spectral_terms(D)
        h.set_terms(table)
        Zk = self.kernel._kernel_format(self.Z())
        Xk = self._local(self.kernel._kernel_format(self.X))
        kff = self.kernel._point_diag(table, Xk, D) if table.shape[3] > 2 + 3 * D else self.kernel._spectral_diag(D)    # envelope: per point
        try:
            res = h.titsias_eval(Zk, self._sigma(), self.jitter, kff, grad=grad, sharded=self._data_shard() is not None)
        except MogpError as e:
            if e.code in (MOGP_ENOTPD, MOGP_ENONFINITE):
                print("ERROR:", str(e), file=sys.__stdout__)
                self.print_parameters()
                raise CholeskyException(str(e), None, self)
            raise
        return res, table, D, Zk

    def elbo(self):
        res, _, _, _ = self._run(grad=False)
        return config.dtype(res["elbo"])

    def log_marginal_likelihood(self):
        """maximise the lower bound (reference gpr/model.py:726-728)"""
        return self.elbo()

    def _loss_impl(self):
        self.zero_grad(set_to_none=True)
        res, table, D, Zk = self._run(grad=True)
        C = table.shape[0]
        s2 = self._sigma() ** 2
        M = Zk.shape[0]
        zc = np.bincount(Zk[:, 0].astype(np.int64), minlength=C).astype(np.float64)
        xc = self.__dict__.get("_xc_cache")                    # training points per channel: X does not change under a model (O(N) per evaluation otherwise)
        if xc is None or xc[0] is not self.X or xc[1].shape[0] != C:
            xc = (self.X, np.bincount(self.kernel._kernel_format(self.X)[:, 0].astype(np.int64), minlength=C).astype(np.float64))
            self.__dict__["_xc_cache"] = xc
        xc = xc[1]
        gt = _gtable_from_moments(table, res["mom_uu"], D, lower=True) + _gtable_from_moments(table, res["mom_uf"], D, lower=False)
        env = table.shape[3] > 2 + 3 * D
        gz_jit = 0.0
        if env:
            # with an envelope (MOHSM) the diagonal follows the points: jitter * mean(diag Kuu) depends on A, L, c AND on Z itself, and
            # sum_k Kff_diag[k] is a sum over the training points (reference gpr/model.py:244, :723 through autograd)
            gt += (self.jitter * res["trGA"] / M) * self.kernel._point_diag_table_grad(table, Zk, D)
            gz_jit = (self.jitter * res["trGA"] / M) * self.kernel._point_diag_input_grad(table, Zk, D)
            self.kernel._spectral_backward(-gt + (0.5 / s2) * self.kernel._point_diag_table_grad(table, self.kernel._kernel_format(self.X), D))
        else:
            for i in range(C):
                gt[i, i, :, 0] += self.jitter * res["trGA"] * zc[i] / M          # jitter * mean(diag Kuu), gpr/model.py:244
            # - 1/(2 s2) sum_k Kff_diag[k]  (gpr/model.py:723): K_diag is constant per channel.  Where K_diag[c] is the sum of the diagonal amplitudes
            # (the kernels that keep Kernel._spectral_diag_backward) that is a table gradient too: ONE pass through the parameter algebra instead of two
            from .kernel import Kernel
            if type(self.kernel)._spectral_diag_backward is Kernel._spectral_diag_backward:
                for i in range(C):
                    gt[i, i, :, 0] -= 0.5 * xc[i] / s2
                self.kernel._spectral_backward(-gt)
            else:
                self.kernel._spectral_backward(-gt)
                self.kernel._spectral_diag_backward(0.5 * xc / s2, D)
        scale = self.likelihood.scale
        scale.accumulate_grad(np.reshape(-res["dsigma"], scale.data.shape))
        gz = np.zeros(self.Z.data.shape)
        off = 0 if self.kernel.output_dims is None else 1
        gz[:, off:] = -(res["gZ"] + gz_jit)
        self.Z.accumulate_grad(gz)
        return config.dtype(-res["elbo"] - self.log_prior())

    def predict_f(self, X, full=False):
        """reference gpr/model.py:730-765"""
        X = self._check_input(X)
        h = self._device_handle()
        D = self.X.shape[1] - (0 if self.kernel.output_dims is None else 1)
        h.set_terms(self.kernel._spectral_terms(D))
        table = self.kernel._spectral_terms(D)
        Xsk = self.kernel._kernel_format(X)
        kss = self.kernel._point_diag(table, Xsk, D) if table.shape[3] > 2 + 3 * D else self.kernel._spectral_diag(D)
        mu, var = h.titsias_predict(self.kernel._kernel_format(self.Z()), self._sigma(), self.jitter,
                                    Xsk, kss, sharded=self._data_shard() is not None)
        if full:                                        # K_ss - a^T a + b^T b with the a, b this prediction left on the device (reference :758-760)
            var = h.sparse_predict_cov(X.shape[0])
        if self.mean is not None:
            mu = mu + np.asarray(self.mean(X)).reshape(-1, 1)
        return mu, var


class Snelson(_DataParallel, Model):
    """
    Sparse GP regression with pseudo-inputs, Snelson & Ghahramani 2005 (reference gpr/model.py:485-576): the FITC marginal likelihood
        p = log N(y | 0, Qff + diag(Kff - Qff) + sigma^2 I),   Qff = Kfu Kuu^-1 Kuf,
    with trainable inducing inputs `Z` (their channel column carries no gradient) and a scalar or per-channel noise variance.
    """

    def __init__(self, kernel, X, y, Z=10, Z_init="grid", variance=1.0, jitter=1e-8, mean=None):
        variance = np.squeeze(Parameter.to_tensor(variance))
        if 1 < variance.ndim or variance.ndim == 1 and variance.shape[0] != kernel.output_dims:
            raise ValueError("variance must be float or have shape (channels,)")
        super().__init__(kernel, X, y, GaussianLikelihood(np.sqrt(variance)), jitter, mean)
        Z = init_inducing_points(Z, self.X, method=Z_init, output_dims=kernel.output_dims)
        Z = self._check_input(Z)
        self.log_marginal_likelihood_constant = 0.5 * self.X.shape[0] * np.log(2.0 * np.pi)
        self.Z = Parameter(Z, name="induction_points")
        if kernel.output_dims is not None:
            self.Z.num_parameters -= self.Z().shape[0]

    def _noise_vector(self):
        """sigma_c^2 per channel (a scalar scale is shared by all channels: reference _index_channel, gpr/model.py:183-186)"""
        s = np.asarray(self.likelihood.scale(), dtype=np.float64)
        C = self.kernel._channels()
        if s.ndim == 1 and s.shape[0] == C and self.kernel.output_dims is not None:
            return s * s
        return np.repeat(s.reshape(-1)[0] ** 2, C)

    def _run(self, grad):
        from .._lib import MogpError, MOGP_ENOTPD, MOGP_ENONFINITE
        h = self._device_handle()
        D = self.X.shape[1] - (0 if self.kernel.output_dims is None else 1)
        table = self.kernel._spectral_terms(D)
        h.set_terms(table)
        Zk = self.kernel._kernel_format(self.Z())
        env = table.shape[3] > 2 + 3 * D                  # enveloped terms (MOHSM): the kernel diagonal follows the points
        if env and self._data_shard() is not None:
            raise NotImplementedError("Snelson with an enveloped kernel (MOHSM) is single-process only: its per-point diagonal gradient is not reduced "
                                      "over the ranks of the data-parallel form")
        kff = self.kernel._point_diag(table, self._local(self.kernel._kernel_format(self.X)), D) if env else self.kernel._spectral_diag(D)
        try:
            res = h.snelson_eval(Zk, self._noise_vector(), self.jitter, kff, grad=grad, sharded=self._data_shard() is not None)
        except MogpError as e:
            if e.code in (MOGP_ENOTPD, MOGP_ENONFINITE):
                print("ERROR:", str(e), file=sys.__stdout__)
                self.print_parameters()
                raise CholeskyException(str(e), None, self)
            raise
        return res, table, D, Zk

    def log_marginal_likelihood(self):
        """reference gpr/model.py:516-541"""
        res, _, _, _ = self._run(grad=False)
        return config.dtype(res["lml"])

    def _loss_impl(self):
        self.zero_grad(set_to_none=True)
        res, table, D, Zk = self._run(grad=True)
        C = table.shape[0]
        M = Zk.shape[0]
        zc = np.bincount(Zk[:, 0].astype(np.int64), minlength=C).astype(np.float64)
        gt = _gtable_from_moments(table, res["mom_uu"], D, lower=True) + _gtable_from_moments(table, res["mom_uf"], D, lower=False)
        gz_jit = 0.0
        if table.shape[3] > 2 + 3 * D:
            # enveloped terms: jitter * mean(diag Kuu) depends on A, L, c and on Z itself (gpr/model.py:244 through autograd); dp/dKff_nn comes
            # back per training point and goes through the per-point diagonal K_diag(x_n) = sum_t A_t env_t(x_n)
            Xk = self.kernel._kernel_format(self.X)
            h_pt = np.asarray(res["hsum"], dtype=np.float64)
            gt += (self.jitter * res["trGA"] / M) * self.kernel._point_diag_table_grad(table, Zk, D)
            gz_jit = (self.jitter * res["trGA"] / M) * self.kernel._point_diag_input_grad(table, Zk, D)
            self.kernel._spectral_backward(-gt - self.kernel._point_diag_table_grad(table, Xk, D, weights=h_pt))
            hsum = np.bincount(Xk[:, 0].astype(np.int64), weights=h_pt, minlength=C).astype(np.float64)
        else:
            for i in range(C):
                gt[i, i, :, 0] += self.jitter * res["trGA"] * zc[i] / M          # jitter * mean(diag Kuu), gpr/model.py:244
            self.kernel._spectral_backward(-gt)
            hsum = np.asarray(res["hsum"], dtype=np.float64)                     # d p / d Kff_diag = d p / d sigma^2, summed per channel
            self.kernel._spectral_diag_backward(-hsum, D)
        scale = self.likelihood.scale
        sc = np.asarray(scale(), dtype=np.float64)
        if sc.ndim == 1 and sc.shape[0] == C and self.kernel.output_dims is not None:
            gsc = 2.0 * sc * hsum
        else:
            gsc = np.reshape(2.0 * sc * np.sum(hsum), sc.shape)
        scale.accumulate_grad(-gsc)
        gz = np.zeros(self.Z.data.shape)
        off = 0 if self.kernel.output_dims is None else 1
        gz[:, off:] = -(res["gZ"] + gz_jit)
        self.Z.accumulate_grad(gz)
        return config.dtype(-res["lml"] - self.log_prior())

    def predict_f(self, X, full=False):
        """reference gpr/model.py:543-576 (its full=True branch uses undefined names; not on this path either)"""
        if full:
            raise NotImplementedError("full predictive covariance for Snelson is not on the HIP path")
        X = self._check_input(X)
        h = self._device_handle()
        D = self.X.shape[1] - (0 if self.kernel.output_dims is None else 1)
        table = self.kernel._spectral_terms(D)
        h.set_terms(table)
        Xsk = self.kernel._kernel_format(X)
        if table.shape[3] > 2 + 3 * D:                    # enveloped terms: K_diag per training / test point
            kd = self.kernel._point_diag(table, self._local(self.kernel._kernel_format(self.X)), D)
            ks = self.kernel._point_diag(table, Xsk, D)
        else:
            kd = ks = self.kernel._spectral_diag(D)
        mu, var = h.snelson_predict(self.kernel._kernel_format(self.Z()), self._noise_vector(), self.jitter,
                                    Xsk, kd, ks, sharded=self._data_shard() is not None)
        if self.mean is not None:
            mu = mu + np.asarray(self.mean(X)).reshape(-1, 1)
        return mu, var


class OpperArchambeau(Model):
    """
    Variational Gaussian approximation of Opper & Archambeau 2009 (reference gpr/model.py:578-668): q(f) = N(K nu, (K^-1 + diag(lambda^2))^-1)
    with one `q_nu` and one positive `q_lambda` per data point;  ELBO = E_q[log p(y | f)] - kl / 2.  The O(N^3) algebra runs on the device in
    two calls around the likelihood (like the Hensman models): forward -> per-point mu, var of q(f) and the kl term; the likelihood (host,
    O(N)) returns its expectation and dE/dmu, dE/dvar; backward -> the gradients of kernel, q_nu, q_lambda.  Any likelihood of gpr/likelihood.py (its expectation and derivatives are the host's; the device algebra never sees it).
    No jitter enters (the reference's Cholesky calls here pass add_jitter=False); `jitter` is kept for the signature.
    """

    def __init__(self, kernel, X, y, likelihood=None, jitter=1e-8, mean=None):
        if likelihood is None:
            likelihood = GaussianLikelihood(1.0)
        super().__init__(kernel, X, y, likelihood, jitter, mean)
        n = self.X.shape[0]
        self.q_nu = Parameter(np.zeros((n, 1)))
        self.q_lambda = Parameter(np.ones((n, 1)), lower=config.positive_minimum)

    def _device_handle(self):
        if self._handle is None:
            from .._lib import ExactHandle
            self._handle = ExactHandle(config.device, self.kernel._kernel_format(self.X), self.y, self.kernel._channels())
        return self._handle

    def _forward(self):
        from .._lib import MogpError, MOGP_ENOTPD, MOGP_ENONFINITE
        h = self._device_handle()
        D = self.X.shape[1] - (0 if self.kernel.output_dims is None else 1)
        table = self.kernel._spectral_terms(D)
        h.set_terms(table)
        try:
            res = h.oa_forward(self.q_nu(), self.q_lambda())
        except MogpError as e:
            if e.code in (MOGP_ENOTPD, MOGP_ENONFINITE):
                print("ERROR:", str(e), file=sys.__stdout__)
                self.print_parameters()
                raise CholeskyException(str(e), None, self)
            raise
        return h, res, table, D

    def _targets(self, mu):
        """reference :604-607, :625-626: with a mean function both the targets and the mean of q(f) have it subtracted"""
        if self.mean is None:
            return self.y, mu
        mean = np.asarray(self.mean(self.X)).reshape(-1)
        return self.y - mean.reshape(-1, 1), mu - mean

    def elbo(self):
        h, res, _, _ = self._forward()
        y, mu = self._targets(res["mu"])
        return config.dtype(self.likelihood.variational_expectation(self._likelihood_X(self.X), y, mu, res["var"]) - 0.5 * res["kl"])

    def log_marginal_likelihood(self):
        """maximise the lower bound (reference gpr/model.py:636-638)"""
        return self.elbo()

    def _loss_impl(self):
        self.zero_grad(set_to_none=True)
        h, res, table, D = self._forward()
        y, mu = self._targets(res["mu"])
        ve, e, f, pgrads = self.likelihood.variational_expectation(self._likelihood_X(self.X), y, mu, res["var"], grad=True)
        bw = h.oa_backward(e, f)
        self.kernel._spectral_backward(-_gtable_from_moments(table, bw["mom"], D, lower=True))
        for p, g in pgrads:
            p.accumulate_grad(np.reshape(-np.asarray(g, dtype=np.float64), p.data.shape))
        self.q_nu.accumulate_grad(-np.reshape(bw["g_nu"], self.q_nu.data.shape))
        self.q_lambda.accumulate_grad(-np.reshape(bw["g_lambda"], self.q_lambda.data.shape))
        return config.dtype(-(ve - 0.5 * res["kl"]) - self.log_prior())

    def predict_f(self, X, full=False):
        """reference gpr/model.py:640-668"""
        X = self._check_input(X)
        h = self._device_handle()
        D = self.X.shape[1] - (0 if self.kernel.output_dims is None else 1)
        h.set_terms(self.kernel._spectral_terms(D))
        mu, var = h.oa_predict(self.q_nu(), self.q_lambda(), self.kernel._spectral_diag(D), self.kernel._kernel_format(X), full=full)
        if self.mean is not None:
            mu = mu + np.asarray(self.mean(X)).reshape(-1, 1)
        return mu, var


class SparseHensman(_DataParallel, Model):
    """
    Sparse variational GP of Hensman et al. 2015, whitened (reference gpr/model.py:767-878): q(u) = N(L q_mu, L S S^T L^T), L L^T = Kuu,
    S = tril(q_sqrt);  ELBO = E_q[log p(y | f)] - KL(q || p).  The O(N M^2) algebra runs on the device in two calls around the
    likelihood: forward -> per-point mu, var of q(f) at the training inputs; the likelihood (host, O(N)) returns its expectation and
    dE/dmu, dE/dvar; backward -> the gradients of kernel, inducing inputs, q_mu, q_sqrt.  Any likelihood of gpr/likelihood.py (its expectation and derivatives are the host's; the device algebra never sees it).
    The KL term mirrors the reference's (:816-822), which counts only the DIAGONAL of q_sqrt in the trace and the determinant.
    """

    def __init__(self, kernel, X, y, Z=None, Z_init="grid", likelihood=None, jitter=1e-8, mean=None):
        if likelihood is None:
            likelihood = GaussianLikelihood(1.0)
        super().__init__(kernel, X, y, likelihood, jitter, mean)
        n = self.X.shape[0]
        self.is_sparse = Z is not None
        if self.is_sparse:
            Z = init_inducing_points(Z, self.X, method=Z_init, output_dims=kernel.output_dims)
            Z = self._check_input(Z)
            n = Z.shape[0]
        self.log_marginal_likelihood_constant = 0.5 * self.X.shape[0] * np.log(2.0 * np.pi)
        self.q_mu = Parameter(np.zeros((n, 1)))
        self.q_sqrt = Parameter(np.eye(n))
        self.q_sqrt.num_parameters = int((n * n + n) / 2)
        if self.is_sparse:
            self.Z = Parameter(Z, name="induction_points")
            if kernel.output_dims is not None:
                self.Z.num_parameters -= self.Z().shape[0]
        else:
            self.Z = Parameter(self.X, train=False)         # the data points themselves, not trained (reference :812)

    def _shardable(self):
        return self.is_sparse               # the non-sparse model lives on all data points

    def _reduce(self, value):
        """sum of a host scalar / small array over the ranks holding the other shards"""
        comm = self._data_shard()
        if comm is None:
            return value
        a = np.atleast_1d(np.array(value, dtype=np.float64))
        comm.all_reduce_host(a)
        return a if np.ndim(value) else float(a[0])

    def kl_gaussian(self, q_mu, q_sqrt):
        """reference gpr/model.py:816-822"""
        S_diag = np.diagonal(q_sqrt) ** 2
        return 0.5 * (float(np.sum(q_mu * q_mu)) - np.sum(np.log(S_diag)) + np.sum(S_diag) - q_mu.shape[0])

    def _forward(self):
        from .._lib import MogpError, MOGP_ENOTPD, MOGP_ENONFINITE
        h = self._device_handle()
        D = self.X.shape[1] - (0 if self.kernel.output_dims is None else 1)
        table = self.kernel._spectral_terms(D)
        h.set_terms(table)
        Zk = self.kernel._kernel_format(self.Z())
        env = table.shape[3] > 2 + 3 * D                  # enveloped terms (MOHSM): the kernel diagonal follows the points
        if env and self._data_shard() is not None:
            raise NotImplementedError("SparseHensman with an enveloped kernel (MOHSM) is single-process only: the per-point diagonal gradient is not "
                                      "reduced over the ranks of the data-parallel form")
        kff = self.kernel._point_diag(table, self._local(self.kernel._kernel_format(self.X)), D) if env else self.kernel._spectral_diag(D)
        try:
            res = h.svgp_forward(Zk, self.q_mu(), self.q_sqrt(), self.jitter, kff, dense=not self.is_sparse)
        except MogpError as e:
            if e.code in (MOGP_ENOTPD, MOGP_ENONFINITE):
                print("ERROR:", str(e), file=sys.__stdout__)
                self.print_parameters()
                raise CholeskyException(str(e), None, self)
            raise
        return h, res, table, D, Zk

    def _y(self):
        return self.y if self.mean is None else self.y - np.asarray(self.mean(self.X)).reshape(-1, 1)

    def _mu(self, res):
        """q(f)'s mean at the training inputs as the likelihood sees it.  The dense model (reference gpr/model.py:834-837) subtracts the mean
        function from BOTH y and q(f)'s mean, so that it cancels out of the bound -- reproduced; the sparse model (:861) shifts y only."""
        if self.is_sparse or self.mean is None:
            return res["mu"]
        return res["mu"] - self._local(np.asarray(self.mean(self.X)).reshape(-1))

    def elbo(self):
        h, res, _, _, _ = self._forward()
        ve = self.likelihood.variational_expectation(self._local(self._likelihood_X(self.X)), self._local(self._y()), self._mu(res), res["var"])
        return config.dtype(self._reduce(ve) - self.kl_gaussian(self.q_mu(), self.q_sqrt()))

    def log_marginal_likelihood(self):
        """maximise the lower bound (reference gpr/model.py:847-849)"""
        return self.elbo()

    def _loss_impl(self):
        self.zero_grad(set_to_none=True)
        h, res, table, D, Zk = self._forward()
        sharded = self._data_shard() is not None            # data-parallel: mu / var, e, f are those of this rank's points
        ve, e, f, pgrads = self.likelihood.variational_expectation(self._local(self._likelihood_X(self.X)), self._local(self._y()), self._mu(res), res["var"], grad=True)
        ve = self._reduce(ve)
        pgrads = [(p, self._reduce(g)) for p, g in pgrads]
        q_mu, q_sqrt = np.asarray(self.q_mu(), dtype=np.float64), np.asarray(self.q_sqrt(), dtype=np.float64)
        elbo = ve - self.kl_gaussian(q_mu, q_sqrt)
        bw = h.svgp_backward(e, f, sharded=sharded)
        C = table.shape[0]
        M = Zk.shape[0]
        zc = np.bincount(Zk[:, 0].astype(np.int64), minlength=C).astype(np.float64)
        xc = self._local(self.kernel._kernel_format(self.X))[:, 0].astype(np.int64)
        gt = _gtable_from_moments(table, bw["mom_uu"], D, lower=True) + _gtable_from_moments(table, bw["mom_uf"], D, lower=False)
        gz_jit = 0.0
        if table.shape[3] > 2 + 3 * D:
            # enveloped terms: jitter * mean(diag Kuu) depends on A, L, c and on the inducing inputs themselves (gpr/model.py:244 through autograd),
            # and var_n = K_diag(x_n) - ... goes back through the per-point diagonal with the likelihood's d/dvar_n as weights
            gt += (self.jitter * bw["trGA"] / M) * self.kernel._point_diag_table_grad(table, Zk, D)
            gz_jit = (self.jitter * bw["trGA"] / M) * self.kernel._point_diag_input_grad(table, Zk, D)
            if self.is_sparse:
                gt += self.kernel._point_diag_table_grad(table, self._local(self.kernel._kernel_format(self.X)), D, weights=f)
            self.kernel._spectral_backward(-gt)
        else:
            for i in range(C):
                gt[i, i, :, 0] += self.jitter * bw["trGA"] * zc[i] / M          # jitter * mean(diag Kuu), gpr/model.py:244
            self.kernel._spectral_backward(-gt)
            if self.is_sparse:                          # var_n = K_diag[c(n)] - ... (the dense model's variance at its own inputs has no such term)
                self.kernel._spectral_diag_backward(-self._reduce(np.bincount(xc, weights=f, minlength=C)), D)
        for p, g in pgrads:
            p.accumulate_grad(np.reshape(-np.asarray(g, dtype=np.float64), p.data.shape))
        self.q_mu.accumulate_grad(-(np.reshape(bw["g_qmu"], q_mu.shape) - q_mu))
        s = np.diagonal(q_sqrt)
        self.q_sqrt.accumulate_grad(-(np.tril(bw["g_qsqrt"]) - np.diag(s - 1.0 / s)))
        if self.is_sparse:
            gz = np.zeros(self.Z.data.shape)
            off = 0 if self.kernel.output_dims is None else 1
            gz[:, off:] = -(bw["gZ"] + gz_jit)
            self.Z.accumulate_grad(gz)
        return config.dtype(-elbo - self.log_prior())

    def predict_f(self, X, full=False):
        """reference gpr/model.py:851-878"""
        X = self._check_input(X)
        h = self._device_handle()
        D = self.X.shape[1] - (0 if self.kernel.output_dims is None else 1)
        table = self.kernel._spectral_terms(D)
        h.set_terms(table)
        Xsk = self.kernel._kernel_format(X)
        if table.shape[3] > 2 + 3 * D:                    # enveloped terms: K_diag per training / test point
            kd = self.kernel._point_diag(table, self._local(self.kernel._kernel_format(self.X)), D)
            ks = self.kernel._point_diag(table, Xsk, D)
        else:
            kd = ks = self.kernel._spectral_diag(D)
        res = h.svgp_forward(self.kernel._kernel_format(self.Z()), self.q_mu(), self.q_sqrt(), self.jitter, kd,
                             Xs=Xsk, kss_diag=ks)
        mu = np.reshape(res["mu"], (-1, 1))
        if self.mean is not None:
            mu = mu + np.asarray(self.mean(X)).reshape(-1, 1)
        if full:                                        # reference :870-872
            return mu, h.sparse_predict_cov(X.shape[0])
        return mu, np.reshape(res["var"], (-1, 1))


class Hensman(SparseHensman):
    """
    The non-sparse variational GP (reference gpr/model.py:880-886): the inducing inputs ARE the data points and are not trained; at the
    training inputs q(f) = N(L q_mu, L S S^T L^T) with L the factor of K_ff itself (reference :834-840) -- the same device algebra with
    a = L^T instead of L^-1 K_uf.  Predictions at new inputs take the sparse formula with Z = X, like the reference (:851-868).
    """

    def __init__(self, kernel, X, y, likelihood=None, jitter=1e-8, mean=None):
        super().__init__(kernel, X, y, None, "grid", likelihood, jitter, mean)
