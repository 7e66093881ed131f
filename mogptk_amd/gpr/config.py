"""
Global configuration singleton, mirroring mogptk/gpr/config.py:3-73 for the HIP path.

`config.dtype` is the dtype of everything the host holds -- parameters (raw values, Adam state), X, y, losses, gradients, predictions --
float64 by default, float32 after use_single_precision(), with the reference's jitter floor per dtype (gpr/model.py:106-110).  The
device arithmetic itself is fp64 in both cases (the fp64 MFMA path is the product; there is no fp32 kernel set): single precision is
honoured at the boundary -- float32-rounded inputs and parameters go in, float32 results come out -- which is at least as accurate as
the reference's float32 run.  `config.device` is the HIP device ordinal the C-ABI contexts bind to.
"""
import numpy as np


class Config:
    dtype = np.float64
    device = 0
    positive_minimum = 1e-8
    comm = None          # mogptk_amd.dist.Comm when exact evaluations are sharded over several GPUs
    accurate_fallback = True      # gpr.Exact repeats a gradient evaluation in the backward-stable form when the factor's diagonal says K + noise is ill-conditioned


config = Config()


def use_double_precision():
    """mogptk/gpr/config.py:26-30."""
    config.dtype = np.float64


def use_single_precision():
    """mogptk/gpr/config.py:20-24: float32 for all host tensors created from now on (and the 1e-6 jitter floor); the device still
    factorises in fp64 (see the module docstring)."""
    config.dtype = np.float32


def use_half_precision():
    """mogptk/gpr/config.py:12-18.  float16 host tensors cannot carry the spectral hyper-parameters (variances of 1e-3 .. 1e-2 next to
    positive_minimum = 1e-8): refused rather than silently widened."""
    raise NotImplementedError("half precision is not on the MI355X exact-GP path")


def use_gpu(n=None):
    """mogptk/gpr/config.py:41-52: select the device new models bind to."""
    from .._lib import lib
    count = lib().mogp_device_count()
    if count <= 0:
        print("HIP device is not available")
    elif n is not None and (not isinstance(n, int) or n < 0 or count <= n):
        print("HIP GPU '%s' is not available" % (n,))
    else:
        config.device = 0 if n is None else n


def use_hip(n=None):
    """Backend selector in the style of use_gpu(); the HIP path is the only backend."""
    use_gpu(n)


def use_cpu(n=None):
    """mogptk/gpr/config.py:32-39.  There is no CPU fallback on this path by design."""
    raise NotImplementedError("mogptk_amd has no CPU path; use the reference mogptk on CPU")


def print_gpu_information():
    """mogptk/gpr/config.py:54-67."""
    from .._lib import lib, device_name
    count = lib().mogp_device_count()
    if count <= 0:
        print("HIP device is not available")
        return
    print("HIP is available:")
    for n in range(count):
        print("%2d  %s%s" % (n, device_name(n), " (selected)" if n == config.device else ""))


def set_positive_minimum(val):
    """mogptk/gpr/config.py:69-73."""
    config.positive_minimum = val
