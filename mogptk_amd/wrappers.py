"""
Model wrappers MOSM / SM / CSM / SM_LMC -- host-side mirror of mogptk/models/{mosm,sm,csm,sm_lmc}.py constructors.

Constructor semantics are part of the drop-in boundary and are reproduced including quirk Q2
(SURVEY.md 8b): the Nyquist re-bounding `mean.assign(upper=...)` at mosm.py:60 / sm.py:60 / csm.py:64
re-interprets the raw values as constrained ones and collapses every `mean` to its lower bound until the
user assigns values (or calls init_parameters, which is outside the hot path: SURVEY.md 8f-3).
"""
import numpy as np

from .dataset import DataSet
from .model import Model, Exact, logger
from .gpr import (MultiOutputSpectralMixtureKernel, IndependentMultiOutputKernel, SpectralMixtureKernel,
                  CrossSpectralKernel, MixtureKernel, LinearModelOfCoregionalizationKernel, SpectralKernel)


def _rand(*shape):
    # the reference draws torch.rand (mosm.py:53-55); numpy's global stream plays that role here
    return np.random.rand(*shape)


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
        raise NotImplementedError("init_parameters (BNSE / Lomb-Scargle / SM estimation, reference mosm.py:62-113) "
                                  "is a caller of the hot path and not built yet (SURVEY.md 8f-3); "
                                  "assign hyper-parameters with model.gpr.kernel.<param>.assign(...)")


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
        raise NotImplementedError("init_parameters is not built yet (SURVEY.md 8f-3)")


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
        raise NotImplementedError("init_parameters is not built yet (SURVEY.md 8f-3)")


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
        raise NotImplementedError("init_parameters is not built yet (SURVEY.md 8f-3)")
