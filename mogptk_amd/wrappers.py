"""
Model wrappers MOSM / SM / CSM / SM_LMC / CONV / MOHSM -- host-side mirror of mogptk/models/{mosm,sm,csm,sm_lmc,conv,mohsm}.py constructors.

Constructor semantics are part of the drop-in boundary and are reproduced including quirk Q2
(SURVEY.md 8b): the Nyquist re-bounding `mean.assign(upper=...)` at mosm.py:60 / sm.py:60 / csm.py:64
re-interprets the raw values as constrained ones and collapses every `mean` to its lower bound until the
user assigns values or calls init_parameters ('BNSE', 'LS', 'SM' and 'IPS').
"""
import numpy as np

from .dataset import DataSet
from .model import Model, Exact, logger
from .gpr import (MultiOutputSpectralMixtureKernel, IndependentMultiOutputKernel, SpectralMixtureKernel,
                  CrossSpectralKernel, MixtureKernel, LinearModelOfCoregionalizationKernel, SpectralKernel,
                  GaussianConvolutionProcessKernel, MultiOutputHarmonizableSpectralKernel)


def _rand(*shape):
    # the reference draws torch.rand (mosm.py:53-55); numpy's global stream plays that role here
    return np.random.rand(*shape)


def _estimate(model, method, iters, what):
    """shared front of init_parameters (reference mosm.py:80-91, csm.py:80-91, sm_lmc.py:94-105)"""
    if method.lower() not in ("bnse", "ls", "sm"):
        raise ValueError("valid methods of estimation are BNSE, LS, and SM")
    if method.lower() == "bnse":
        amplitudes, means, variances = model.dataset.get_bnse_estimation(model.Q, iters=iters)
    elif method.lower() == "ls":
        amplitudes, means, variances = model.dataset.get_ls_estimation(model.Q)
    else:
        amplitudes, means, variances = model.dataset.get_sm_estimation(model.Q, iters=iters)
    if len(amplitudes) == 0:
        logger.warning("{} could not find peaks for {}".format(method, what))
        return None
    return amplitudes, means, variances


def _init_noise(model):
    """the noise scale from the spread of every channel (reference mosm.py:106-113 and its siblings)"""
    from .gpr import GaussianLikelihood
    if isinstance(model.gpr.likelihood, GaussianLikelihood):
        _, Y = model.dataset.get_train_data(transformed=True)
        Y_std = [Y[j].std() for j in range(model.dataset.get_output_dims())]
        if np.ndim(model.gpr.likelihood.scale()) == 0:
            model.gpr.likelihood.scale.assign(np.mean(Y_std))
        else:
            model.gpr.likelihood.scale.assign(Y_std)


class MOSM(Model):
    """Multi-Output Spectral Mixture model with Q components (reference models/mosm.py:10-60)."""

    def __init__(self, dataset, Q=1, inference=Exact(), mean=None, name="MOSM"):
        if not isinstance(dataset, DataSet):
            dataset = DataSet(dataset)
        output_dims = dataset.get_output_dims()
        input_dims = dataset.get_input_dims()[0]
        for input_dim in dataset.get_input_dims()[1:]:
            if input_dim != input_dims:
                raise ValueError("input dimensions for all channels must match")

        kernel = MultiOutputSpectralMixtureKernel(Q=Q, output_dims=output_dims, input_dims=input_dims)
        kernel.weight.assign(_rand(output_dims, Q))
        kernel.mean.assign(_rand(output_dims, Q, input_dims))
        kernel.variance.assign(_rand(output_dims, Q, input_dims))

        super().__init__(dataset, kernel, inference, mean, name)
        self.Q = Q
        nyquist = np.array(self.dataset.get_nyquist_estimation())[:, None, :].repeat(Q, axis=1)
        self.gpr.kernel.mean.assign(upper=np.maximum(self.gpr.kernel.mean.lower, nyquist))

    def init_parameters(self, method="BNSE", iters=500):
        """Estimate kernel parameters from the data (reference models/mosm.py:62-113): spectrum peaks per channel by BNSE,
        Lomb-Scargle ('LS') or a fitted single-output spectral mixture ('SM', trained on the device); the noise scale from
        the spread of every channel."""
        input_dims = self.dataset.get_input_dims()
        output_dims = self.dataset.get_output_dims()
        est = _estimate(self, method, iters, "MOSM")
        if est is None:
            return
        amplitudes, means, variances = est
        weight = np.zeros((output_dims, self.Q))
        mean = np.zeros((output_dims, self.Q, input_dims[0]))
        variance = np.zeros((output_dims, self.Q, input_dims[0]))
        for q in range(self.Q):
            for j in range(output_dims):
                weight[j, q] = 10.0 * amplitudes[j][q, :].mean()
                mean[j, q, :] = means[j][q, :]
                variance[j, q, :] = variances[j][q, :]
        self.gpr.kernel.weight.assign(weight)
        self.gpr.kernel.mean.assign(mean)
        self.gpr.kernel.variance.assign(variance)
        _init_noise(self)


class SM(Model):
    """Independent Spectral Mixture kernels per channel (reference models/sm.py:9-60); wrapped in an
    IndependentMultiOutputKernel even for one channel."""

    def __init__(self, dataset, Q=1, inference=Exact(), mean=None, name="SM"):
        if not isinstance(dataset, DataSet):
            dataset = DataSet(dataset)
        output_dims = dataset.get_output_dims()
        input_dims = dataset.get_input_dims()[0]
        kernel = IndependentMultiOutputKernel(
            [SpectralMixtureKernel(Q=Q, input_dims=input_dims) for j in range(output_dims)],
            output_dims=output_dims)
        for j in range(output_dims):
            kernel[j].magnitude.assign(_rand(Q))
            kernel[j].mean.assign(_rand(Q, input_dims))
            kernel[j].variance.assign(_rand(Q, input_dims))

        super().__init__(dataset, kernel, inference, mean, name)
        self.Q = Q
        nyquist = np.array(self.dataset.get_nyquist_estimation())[:, None, :].repeat(Q, axis=1)
        for j in range(output_dims):
            self.gpr.kernel[j].mean.assign(upper=np.maximum(self.gpr.kernel[j].mean.lower, nyquist[j, :, :]))

    def init_parameters(self, method="LS", iters=500):
        """reference models/sm.py:62-121: 'IPS' (independent parameter sampling), 'LS' (Lomb-Scargle peaks) or 'BNSE'"""
        input_dims = self.dataset.get_input_dims()
        output_dims = self.dataset.get_output_dims()
        if method.lower() not in ("ips", "ls", "bnse"):
            raise ValueError("valid methods of estimation are IPS, LS, and BNSE")
        if method.lower() == "ips":
            for j in range(output_dims):
                nyquist = self.dataset[j].get_nyquist_estimation()
                x = self.dataset[j].X[self.dataset[j].mask, :]
                y = self.dataset[j].Y_transformer.forward(self.dataset[j].Y[self.dataset[j].mask], x)
                x_range = np.max(x, axis=0) - np.min(x, axis=0)
                self.gpr.kernel[j].magnitude.assign([2.0 * y.std() / self.Q] * self.Q)
                self.gpr.kernel[j].mean.assign(nyquist * _rand(self.Q, input_dims[j]))
                self.gpr.kernel[j].variance.assign(1.0 / (np.abs(np.random.randn(self.Q, input_dims[j])) * x_range))
            return
        elif method.lower() == "ls":
            amplitudes, means, variances = self.dataset.get_ls_estimation(self.Q)
            if len(amplitudes) == 0:
                logger.warning("LS could not find peaks for SM")
                return
        else:
            amplitudes, means, variances = self.dataset.get_bnse_estimation(self.Q, iters=iters)
            if np.sum(amplitudes) == 0.0:
                logger.warning("BNSE could not find peaks for SM")
                return
        for j in range(output_dims):
            self.gpr.kernel[j].magnitude.assign(amplitudes[j].mean(axis=1) ** 2)
            self.gpr.kernel[j].mean.assign(means[j])
            self.gpr.kernel[j].variance.assign(variances[j])
        _init_noise(self)


class CSM(Model):
    """Cross Spectral Mixture model with Q components of rank Rq (reference models/csm.py:9-64)."""

    def __init__(self, dataset, Q=1, Rq=1, inference=Exact(), mean=None, name="CSM"):
        if not isinstance(dataset, DataSet):
            dataset = DataSet(dataset)
        output_dims = dataset.get_output_dims()
        input_dims = dataset.get_input_dims()[0]
        for input_dim in dataset.get_input_dims()[1:]:
            if input_dim != input_dims:
                raise ValueError("input dimensions for all channels must match")

        spectral = CrossSpectralKernel(output_dims=output_dims, input_dims=input_dims, Rq=Rq)
        kernel = MixtureKernel(spectral, Q)
        for q in range(Q):
            kernel[q].amplitude.assign(_rand(output_dims, Rq))
            kernel[q].mean.assign(_rand(input_dims))
            kernel[q].variance.assign(_rand(input_dims))

        super().__init__(dataset, kernel, inference, mean, name)
        self.Q = Q
        self.Rq = Rq
        nyquist = np.amin(self.dataset.get_nyquist_estimation(), axis=0)
        for q in range(Q):
            self.gpr.kernel[q].mean.assign(upper=np.maximum(self.gpr.kernel[q].mean.lower, nyquist))

    def init_parameters(self, method="BNSE", iters=500):
        """reference models/csm.py:66-111"""
        est = _estimate(self, method, iters, "MOSM")           # (sic: the reference's message says MOSM here too, csm.py:90)
        if est is None:
            return
        amplitudes, means, variances = est
        output_dims = self.dataset.get_output_dims()
        means = np.concatenate(means, axis=0)
        variances = np.concatenate(variances, axis=0)
        constant = np.empty((output_dims, self.Q, self.Rq), dtype=np.float32)   # the reference fills a float32 torch.rand tensor (csm.py:96)
        for q in range(self.Q):
            for j in range(len(self.dataset)):
                constant[j, q, :] = amplitudes[j][q, :].mean() ** 2 / self.Rq
            self.gpr.kernel[q].amplitude.assign(constant[:, q, :])
            self.gpr.kernel[q].mean.assign(means[q, :])
            self.gpr.kernel[q].variance.assign(variances[q, :])
        _init_noise(self)


class SM_LMC(Model):
    """Spectral-mixture linear model of coregionalization with Q components of Rq latent functions (reference
    models/sm_lmc.py:8-67): LMC over Q SpectralKernel base kernels whose magnitudes are pegged to 1 (train=False; the LMC weight
    carries the amplitude), Nyquist upper bound on the means (with quirk Q2, as in the other wrappers)."""

    def __init__(self, dataset, Q=1, Rq=1, inference=Exact(), mean=None, name="SM-LMC"):
        if not isinstance(dataset, DataSet):
            dataset = DataSet(dataset)
        output_dims = dataset.get_output_dims()
        input_dims = dataset.get_input_dims()[0]
        for input_dim in dataset.get_input_dims()[1:]:
            if input_dim != input_dims:
                raise ValueError("input dimensions for all channels must match")

        spectral = [SpectralKernel(input_dims) for q in range(Q)]
        kernel = LinearModelOfCoregionalizationKernel(spectral, output_dims=output_dims, input_dims=input_dims, Q=Q, Rq=Rq)
        kernel.weight.assign(_rand(output_dims, Q, Rq))
        for q in range(Q):
            kernel[q].magnitude.assign(_rand(1))
            kernel[q].mean.assign(_rand(input_dims))
            kernel[q].variance.assign(_rand(input_dims))

        super().__init__(dataset, kernel, inference, mean, name)
        self.Q = Q
        self.Rq = Rq
        nyquist = np.amin(self.dataset.get_nyquist_estimation(), axis=0)
        for q in range(Q):
            self.gpr.kernel[q].magnitude.assign(1.0, train=False)      # handled by the LMC weight (sm_lmc.py:65)
            self.gpr.kernel[q].mean.assign(upper=np.maximum(self.gpr.kernel[q].mean.lower, nyquist))

    def init_parameters(self, method="BNSE", iters=500):
        """reference models/sm_lmc.py:69-121"""
        est = _estimate(self, method, iters, "SM-LMC")
        if est is None:
            return
        amplitudes, means, variances = est
        output_dims = self.dataset.get_output_dims()
        means = np.concatenate(means, axis=0)
        variances = np.concatenate(variances, axis=0)
        constant = np.empty((output_dims, self.Q, self.Rq), dtype=np.float32)   # float32 in the reference too (sm_lmc.py:110)
        for q in range(self.Q):
            for j in range(len(self.dataset)):
                constant[j, q, :] = amplitudes[j][q, :].mean() / self.Rq
            self.gpr.kernel[q].mean.assign(means[q, :])
            self.gpr.kernel[q].variance.assign(variances[q, :])
        self.gpr.kernel.weight.assign(constant)
        _init_noise(self)


class CONV(Model):
    """Convolutional Gaussian model with Q components (reference models/conv.py:8-52)."""

    def __init__(self, dataset, Q=1, inference=Exact(), mean=None, name="CONV"):
        if not isinstance(dataset, DataSet):
            dataset = DataSet(dataset)
        output_dims = dataset.get_output_dims()
        input_dims = dataset.get_input_dims()[0]
        for input_dim in dataset.get_input_dims()[1:]:
            if input_dim != input_dims:
                raise ValueError("input dimensions for all channels must match")
        conv = GaussianConvolutionProcessKernel(output_dims=output_dims, input_dims=input_dims)
        kernel = MixtureKernel(conv, Q)
        for q in range(Q):
            kernel[q].weight.assign(_rand(output_dims))
            kernel[q].variance.assign(_rand(output_dims, input_dims))
            kernel[q].base_variance.assign(_rand(input_dims))
        super().__init__(dataset, kernel, inference, mean, name)
        self.Q = Q

    def init_parameters(self, method="SM", iters=500):
        """reference models/conv.py:54-97"""
        est = _estimate(self, method, iters, "MOSM")           # (sic, conv.py:82)
        if est is None:
            return
        amplitudes, means, variances = est
        for q in range(self.Q):
            self.gpr.kernel[q].weight.assign([5.0 * amplitude[q, :].mean() for amplitude in amplitudes])
            self.gpr.kernel[q].variance.assign([10.0 * variance[q, :] for variance in variances])
        _init_noise(self)


class MOHSM(Model):
    """Multi-output harmonizable spectral mixture with P components of Q sub-components (reference models/mohsm.py:8-60): a mixture of
    P Q MultiOutputHarmonizableSpectralKernel terms (non-stationary: Gaussian envelopes on the input midpoint), exact inference on the
    device through the wide (2 + 5 D) term rows."""

    def __init__(self, dataset, P=1, Q=1, inference=Exact(), mean=None, name="MOHSM"):
        if not isinstance(dataset, DataSet):
            dataset = DataSet(dataset)
        output_dims = dataset.get_output_dims()
        input_dims = dataset.get_input_dims()[0]
        for input_dim in dataset.get_input_dims()[1:]:
            if input_dim != input_dims:
                raise ValueError("input dimensions for all channels must match")

        spectral = MultiOutputHarmonizableSpectralKernel(output_dims=output_dims, input_dims=input_dims)
        kernel = MixtureKernel(spectral, P * Q)
        for p in range(P):
            for q in range(Q):
                kernel[p * Q + q].weight.assign(_rand(output_dims))
                kernel[p * Q + q].mean.assign(_rand(output_dims, input_dims))
                kernel[p * Q + q].variance.assign(_rand(output_dims, input_dims))
                kernel[p * Q + q].lengthscale.assign(_rand(output_dims))

        super().__init__(dataset, kernel, inference, mean, name)
        self.Q = Q
        self.P = P

    def init_parameters(self, method="BNSE", iters=500):
        """reference models/mohsm.py:62-145: centres / lengthscales spread over [0, 1000] when P > 1, spectrum peaks per channel (BNSE,
        'LS' or 'SM') for mean and variance (variance x (4 + 20 (D - 1))), weights normalised to the channel variances and divided by
        sqrt(lengthscale), noise from the spread of every channel"""
        input_dims = self.dataset.get_input_dims()
        output_dims = self.dataset.get_output_dims()
        if method.lower() not in ("bnse", "ls", "sm"):
            raise ValueError("valid methods of estimation are BNSE, LS, and SM")
        for p in range(self.P):
            for q in range(self.Q):
                if self.P != 1:
                    self.gpr.kernel[p * self.Q + q].center.assign((1000 * p / (self.P - 1)) * np.ones(input_dims[0]))
                    self.gpr.kernel[p * self.Q + q].lengthscale.assign(((self.P + 1) / 1000) * np.ones(output_dims))
            est = _estimate(self, method, iters, "MOHSM")
            if est is None:
                return
            amplitudes, means, variances = est
            weight = np.zeros((output_dims, self.Q))
            for q in range(self.Q):
                mean = np.zeros((output_dims, input_dims[0]))
                variance = np.zeros((output_dims, input_dims[0]))
                for j in range(output_dims):
                    if q < amplitudes[j].shape[0]:
                        weight[j, q] = amplitudes[j][q, :].mean()
                        mean[j, :] = means[j][q, :]
                        variance[j, :] = variances[j][q, :] * (4 + 20 * (max(input_dims) - 1))
                self.gpr.kernel[p * self.Q + q].mean.assign(mean)
                self.gpr.kernel[p * self.Q + q].variance.assign(variance)
            for j, channel in enumerate(self.dataset):
                _, y = channel.get_train_data(transformed=True)
                if 0.0 < weight[j, :].sum():
                    weight[j, :] = (np.sqrt(weight[j, :] / weight[j, :].sum() * y.var())) * 2
            for q in range(self.Q):
                k = self.gpr.kernel[p * self.Q + q]
                k.weight.assign(weight[:, q] / np.sqrt(k.lengthscale.numpy()))
        _init_noise(self)
