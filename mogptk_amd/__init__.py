"""
mogptk_amd -- MI355X-native exact multi-output GP training/prediction path behind the mogptk API surface.

    import mogptk_amd as mogptk
    model = mogptk.MOSM(dataset, Q=3); model.train('Adam', iters=500, lr=0.1); model.predict()

Every O(N^2)/O(N^3) stage runs in hand-written HIP (gfx950) behind the C ABI of include/mogp_hip.h.
"""
from .gpr.config import *
from .gpr.model import CholeskyException
from .util import *
from .dataset import Data, DataSet
from .transformer import (Transformer, TransformBase, TransformDetrend, TransformLinear, TransformNormalize, TransformLog,
                          TransformStandard)
from .model import Model, Exact, Titsias, Snelson, OpperArchambeau, Hensman, LoadModel
from .wrappers import MOSM, SM, CSM, SM_LMC, CONV, MOHSM
from .init import BNSE
from . import gpr
from .dist import use_distributed, use_single_device, use_protocol, shutdown_distributed
