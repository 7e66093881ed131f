"""
The six model wrappers -- MOSM, SM, CSM, SM_LMC, CONV, MOHSM -- as ONE skeleton plus a small table per model.

What is boundary here (and therefore reproduced, not invented): the constructor signatures and defaults of the reference's
mogptk/models/{mosm,sm,csm,sm_lmc,conv,mohsm}.py, the order and shapes of the random draws behind a fresh model (a seeded
run must give the parameters a seeded run of the reference's shapes would), the Nyquist re-bounding of the spectral means with
its quirk Q2 (`assign(upper=...)` without a value re-reads the RAW numbers as constrained ones, so every mean sits at
its lower bound until `init_parameters()` or an explicit assignment), and what `init_parameters(method)` puts where.

How it is written is this package's own: a wrapper declares
  * `_kernel(C, D)`   -> (kernel, parts): the gpr kernel and the sub-kernels that carry the randomised parameters,
  * `_draws(C, D)`    -> [(parameter name, shape)] drawn for every part, in order,
  * `_rebound(nyq)`   -> which means get the Nyquist upper bound (optional),
  * `_place(est)`     -> where the estimated (amplitudes, means, variances) go,
and `_Wrapper` does the rest: data-set coercion, the input-dimension check, the estimator call and the noise initialisation.
"""
import numpy as np

from .dataset import DataSet
from .model import Model, Exact, logger
from .gpr import (MultiOutputSpectralMixtureKernel, IndependentMultiOutputKernel, SpectralMixtureKernel,
                  CrossSpectralKernel, MixtureKernel, LinearModelOfCoregionalizationKernel, SpectralKernel,
                  GaussianConvolutionProcessKernel, MultiOutputHarmonizableSpectralKernel, GaussianLikelihood)

_rand = np.random.rand           # the reference draws torch.rand; numpy's global stream plays that part here


class _Wrapper(Model):
    _label = None                          # the name the "could not find peaks" warning uses
    _methods = ("BNSE", "LS", "SM")        # estimators init_parameters accepts, as the error message lists them
    _uniform_inputs = True                 # every channel must have the same number of input dimensions

    # ---- construction -------------------------------------------------------------------------------------------------------
    def _setup(self, dataset, inference, mean, name, **sizes):
        data = dataset if isinstance(dataset, DataSet) else DataSet(dataset)
        dims = data.get_input_dims()
        if self._uniform_inputs and len(set(dims)) > 1:
            raise ValueError("input dimensions for all channels must match")
        self.__dict__.update(sizes)                    # Q, Rq, P: needed by _kernel / _draws already
        C, D = data.get_output_dims(), dims[0]
        kernel, parts = self._kernel(C, D)
        for part in parts:
            for pname, shape in self._draws(C, D):
                getattr(part, pname).assign(_rand(*shape))
        Model.__init__(self, data, kernel, inference, mean, name)
        self.__dict__.update(sizes)
        if type(self)._rebound is not _Wrapper._rebound:
            self._rebound(np.asarray(self.dataset.get_nyquist_estimation()))

    def _rebound(self, nyquist):
        """which means get the Nyquist frequency as their upper bound (default: none)"""

    @staticmethod
    def _cap(parameter, upper):
        # quirk Q2 lives in Parameter.assign: no value is passed on purpose
        parameter.assign(upper=np.maximum(parameter.lower, upper))

    # ---- init_parameters ----------------------------------------------------------------------------------------------------
    def _check_method(self, method):
        if method.upper() not in self._methods:
            listed = ", ".join(self._methods[:-1]) + ", and " + self._methods[-1]
            raise ValueError("valid methods of estimation are " + listed)
        return method.upper()

    def _spectrum(self, method, iters):
        """(amplitudes, means, variances) per channel from the chosen estimator, or None (with the reference's warning) when it found no peak"""
        how = {"BNSE": lambda: self.dataset.get_bnse_estimation(self.Q, iters=iters),
               "LS": lambda: self.dataset.get_ls_estimation(self.Q),
               "SM": lambda: self.dataset.get_sm_estimation(self.Q, iters=iters)}[method]
        est = how()
        if len(est[0]) == 0:
            logger.warning("{} could not find peaks for {}".format(method, self._label))
            return None
        return est

    def _noise_from_data(self):
        """Gaussian likelihood: its scale starts at the spread of every channel's (transformed) targets"""
        lik = self.gpr.likelihood
        if not isinstance(lik, GaussianLikelihood):
            return
        _, Y = self.dataset.get_train_data(transformed=True)
        spread = [np.std(Y[j]) for j in range(self.dataset.get_output_dims())]
        lik.scale.assign(np.mean(spread) if np.ndim(lik.scale()) == 0 else spread)

    def init_parameters(self, method=None, iters=500):
        method = self._check_method(self._default_method if method is None else method)
        est = self._spectrum(method, iters)
        if est is not None:
            self._place(*est)
            self._noise_from_data()


class MOSM(_Wrapper):
    """Multi-output spectral mixture, Q components (reference models/mosm.py:10-113)."""
    _label, _default_method = "MOSM", "BNSE"

    def __init__(self, dataset, Q=1, inference=Exact(), mean=None, name="MOSM"):
        self._setup(dataset, inference, mean, name, Q=Q)

    def _kernel(self, C, D):
        k = MultiOutputSpectralMixtureKernel(Q=self.Q, output_dims=C, input_dims=D)
        return k, [k]

    def _draws(self, C, D):
        return [("weight", (C, self.Q)), ("mean", (C, self.Q, D)), ("variance", (C, self.Q, D))]

    def _rebound(self, nyquist):
        self._cap(self.gpr.kernel.mean, np.repeat(nyquist[:, None, :], self.Q, axis=1))

    def init_parameters(self, method="BNSE", iters=500):
        """spectrum peaks per channel (BNSE, Lomb-Scargle 'LS', or a fitted single-output mixture 'SM' trained on the device)"""
        super().init_parameters(method, iters)

    def _place(self, amplitudes, means, variances):
        k = self.gpr.kernel
        k.weight.assign(np.stack([10.0 * a[:self.Q].mean(axis=1) for a in amplitudes]))
        k.mean.assign(np.stack([m[:self.Q] for m in means]))
        k.variance.assign(np.stack([v[:self.Q] for v in variances]))


class SM(_Wrapper):
    """Independent spectral mixtures per channel (reference models/sm.py:9-121) -- under an IndependentMultiOutputKernel even for
    a single channel."""
    _label, _default_method, _methods, _uniform_inputs = "SM", "LS", ("IPS", "LS", "BNSE"), False

    def __init__(self, dataset, Q=1, inference=Exact(), mean=None, name="SM"):
        self._setup(dataset, inference, mean, name, Q=Q)

    def _kernel(self, C, D):
        subs = [SpectralMixtureKernel(Q=self.Q, input_dims=D) for _ in range(C)]
        return IndependentMultiOutputKernel(subs, output_dims=C), subs

    def _draws(self, C, D):
        return [("magnitude", (self.Q,)), ("mean", (self.Q, D)), ("variance", (self.Q, D))]

    def _rebound(self, nyquist):
        for j, row in enumerate(nyquist):
            self._cap(self.gpr.kernel[j].mean, np.repeat(row[None, :], self.Q, axis=0))

    def init_parameters(self, method="LS", iters=500):
        """'IPS' (independent parameter sampling), 'LS' (Lomb-Scargle peaks) or 'BNSE'"""
        method = self._check_method(method)
        if method == "IPS":
            return self._sample_independently()
        if method == "LS":
            est = self._spectrum("LS", iters)
        else:
            est = self.dataset.get_bnse_estimation(self.Q, iters=iters)
            if np.sum(est[0]) == 0.0:                  # (the reference tests the SUM here, sm.py:109)
                logger.warning("BNSE could not find peaks for SM")
                est = None
        if est is not None:
            self._place(*est)
            self._noise_from_data()

    def _sample_independently(self):
        for j, channel in enumerate(self.dataset):
            seen = channel.X[channel.mask, :]
            targets = channel.Y_transformer.forward(channel.Y[channel.mask], seen)
            D = seen.shape[1]
            k = self.gpr.kernel[j]
            k.magnitude.assign(np.full(self.Q, 2.0 * targets.std() / self.Q))
            k.mean.assign(channel.get_nyquist_estimation() * _rand(self.Q, D))
            k.variance.assign(1.0 / (np.abs(np.random.randn(self.Q, D)) * np.ptp(seen, axis=0)))

    def _place(self, amplitudes, means, variances):
        for k, a, m, v in zip(self.gpr.kernel.kernels, amplitudes, means, variances):
            k.magnitude.assign(a.mean(axis=1) ** 2)
            k.mean.assign(m)
            k.variance.assign(v)


class _SharedSpectrum(_Wrapper):
    """wrappers whose q-th component has ONE mean / variance for all channels: the estimates of the channels are stacked and row q is used"""

    def _rebound(self, nyquist):
        for q in range(self.Q):
            self._cap(self.gpr.kernel[q].mean, nyquist.min(axis=0))

    def _shared(self, amplitudes, means, variances, power):
        level = np.empty((len(amplitudes), self.Q, self.Rq), dtype=np.float32)        # (float32: the reference fills a default-dtype torch tensor)
        for j, a in enumerate(amplitudes):
            level[j] = (a[:self.Q].mean(axis=1) ** power / self.Rq)[:, None]
        return level, np.concatenate(means, axis=0), np.concatenate(variances, axis=0)


class CSM(_SharedSpectrum):
    """Cross spectral mixture, Q components of rank Rq (reference models/csm.py:9-111)."""
    _label, _default_method = "MOSM", "BNSE"               # (sic: the reference's warning names MOSM, csm.py:90)

    def __init__(self, dataset, Q=1, Rq=1, inference=Exact(), mean=None, name="CSM"):
        self._setup(dataset, inference, mean, name, Q=Q, Rq=Rq)

    def _kernel(self, C, D):
        k = MixtureKernel(CrossSpectralKernel(output_dims=C, input_dims=D, Rq=self.Rq), self.Q)
        return k, [k[q] for q in range(self.Q)]

    def _draws(self, C, D):
        return [("amplitude", (C, self.Rq)), ("mean", (D,)), ("variance", (D,))]

    def init_parameters(self, method="BNSE", iters=500):
        super().init_parameters(method, iters)

    def _place(self, amplitudes, means, variances):
        level, means, variances = self._shared(amplitudes, means, variances, power=2)
        for q in range(self.Q):
            k = self.gpr.kernel[q]
            k.amplitude.assign(level[:, q, :])
            k.mean.assign(means[q])
            k.variance.assign(variances[q])


class SM_LMC(_SharedSpectrum):
    """Linear model of coregionalization over Q SpectralKernel base kernels, Rq latent functions each (reference models/sm_lmc.py:8-121);
    the base kernels' magnitudes are fixed at 1 (train=False): the LMC weight carries the amplitude."""
    _label, _default_method = "SM-LMC", "BNSE"

    def __init__(self, dataset, Q=1, Rq=1, inference=Exact(), mean=None, name="SM-LMC"):
        self._setup(dataset, inference, mean, name, Q=Q, Rq=Rq)

    def _kernel(self, C, D):
        k = LinearModelOfCoregionalizationKernel([SpectralKernel(D) for _ in range(self.Q)], output_dims=C, input_dims=D, Q=self.Q, Rq=self.Rq)
        k.weight.assign(_rand(C, self.Q, self.Rq))         # drawn before the base kernels' parameters
        return k, [k[q] for q in range(self.Q)]

    def _draws(self, C, D):
        return [("magnitude", (1,)), ("mean", (D,)), ("variance", (D,))]

    def _rebound(self, nyquist):
        for q in range(self.Q):
            self.gpr.kernel[q].magnitude.assign(1.0, train=False)
        super()._rebound(nyquist)

    def init_parameters(self, method="BNSE", iters=500):
        super().init_parameters(method, iters)

    def _place(self, amplitudes, means, variances):
        level, means, variances = self._shared(amplitudes, means, variances, power=1)
        for q in range(self.Q):
            self.gpr.kernel[q].mean.assign(means[q])
            self.gpr.kernel[q].variance.assign(variances[q])
        self.gpr.kernel.weight.assign(level)


class CONV(_Wrapper):
    """Convolutional Gaussian process model, Q components (reference models/conv.py:8-97)."""
    _label, _default_method = "MOSM", "SM"                 # (sic, conv.py:82)

    def __init__(self, dataset, Q=1, inference=Exact(), mean=None, name="CONV"):
        self._setup(dataset, inference, mean, name, Q=Q)

    def _kernel(self, C, D):
        k = MixtureKernel(GaussianConvolutionProcessKernel(output_dims=C, input_dims=D), self.Q)
        return k, [k[q] for q in range(self.Q)]

    def _draws(self, C, D):
        return [("weight", (C,)), ("variance", (C, D)), ("base_variance", (D,))]

    def init_parameters(self, method="SM", iters=500):
        super().init_parameters(method, iters)

    def _place(self, amplitudes, means, variances):
        for q in range(self.Q):
            self.gpr.kernel[q].weight.assign([5.0 * a[q].mean() for a in amplitudes])
            self.gpr.kernel[q].variance.assign([10.0 * v[q] for v in variances])


class MOHSM(_Wrapper):
    """Multi-output harmonizable spectral mixture: P centres of Q components (reference models/mohsm.py:8-145) -- non-stationary
    (Gaussian envelopes on the input midpoint), on the device through the wide (2 + 5 D) term rows."""
    _label, _default_method = "MOHSM", "BNSE"

    def __init__(self, dataset, P=1, Q=1, inference=Exact(), mean=None, name="MOHSM"):
        self._setup(dataset, inference, mean, name, Q=Q, P=P)

    def _kernel(self, C, D):
        n = self.P * self.Q
        k = MixtureKernel(MultiOutputHarmonizableSpectralKernel(output_dims=C, input_dims=D), n)
        return k, [k[i] for i in range(n)]

    def _draws(self, C, D):
        return [("weight", (C,)), ("mean", (C, D)), ("variance", (C, D)), ("lengthscale", (C,))]

    def init_parameters(self, method="BNSE", iters=500):
        """centres and lengthscales spread over [0, 1000] when P > 1; per centre: the spectrum's peaks for mean and variance (variance scaled by
        4 + 20 (D - 1)), weights normalised to the channel variances and divided by sqrt(lengthscale); noise from the spread of every channel"""
        method = self._check_method(method)
        dims = self.dataset.get_input_dims()
        C, D = self.dataset.get_output_dims(), dims[0]
        widen = 4 + 20 * (max(dims) - 1)
        for p in range(self.P):
            group = [self.gpr.kernel[p * self.Q + q] for q in range(self.Q)]
            if self.P > 1:
                for k in group:
                    k.center.assign(np.full(D, 1000.0 * p / (self.P - 1)))
                    k.lengthscale.assign(np.full(C, (self.P + 1) / 1000.0))
            est = self._spectrum(method, iters)          # (the estimator runs once per centre, as in the reference)
            if est is None:
                return
            amplitudes, means, variances = est
            share = np.zeros((C, self.Q))
            for q, k in enumerate(group):
                mean, variance = np.zeros((C, D)), np.zeros((C, D))
                for j in range(C):
                    if q < len(amplitudes[j]):
                        share[j, q] = amplitudes[j][q].mean()
                        mean[j], variance[j] = means[j][q], variances[j][q] * widen
                k.mean.assign(mean)
                k.variance.assign(variance)
            for j, channel in enumerate(self.dataset):
                total = share[j].sum()
                if total > 0.0:
                    share[j] = 2.0 * np.sqrt(share[j] / total * channel.get_train_data(transformed=True)[1].var())
            for q, k in enumerate(group):
                k.weight.assign(share[:, q] / np.sqrt(k.lengthscale.numpy()))
        self._noise_from_data()
