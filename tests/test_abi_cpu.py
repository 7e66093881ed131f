"""CPU-side checks of the drop-in boundary: the shared library loads, exports exactly what include/mogp_hip.h declares,
and the product path fails loudly (no fallback) when no device is present.  No compute calls."""
import ctypes
import os
import re
import numpy as np
import pytest

import mogptk_amd
from mogptk_amd import _lib, gpr

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    src = open(os.path.join(ROOT, "include", "mogp_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(mogp_[a-z_0-9]+)\s*\(", src)))


def test_library_loads_and_exports_every_declared_symbol():
    l = _lib.lib()
    names = header_functions()
    assert len(names) >= 15
    for n in names:
        assert hasattr(l, n), "libmogp_hip.so does not export %s" % n
    assert sorted(_lib.SIGNATURES) == names          # the ctypes binding covers the header one to one
    assert b"gfx950" in l.mogp_version()


def test_no_torch_types_in_the_abi():
    src = open(os.path.join(ROOT, "include", "mogp_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)            # declarations only
    assert "torch" not in src.lower() and "at::" not in src and "Tensor" not in src


@pytest.mark.skipif(_lib.lib().mogp_device_count() > 0, reason="a GPU is present")
def test_product_path_fails_loudly_without_a_device():
    h = ctypes.c_void_p()
    rc = _lib.lib().mogp_ctx_create(0, ctypes.byref(h))
    assert rc == _lib.MOGP_ENODEVICE and b"no CPU path" in _lib.lib().mogp_last_error()
    k = gpr.MultiOutputSpectralMixtureKernel(Q=1, output_dims=2)
    X = np.array([[0.0, 0.0], [0.0, 1.0], [1.0, 2.0]])
    with pytest.raises(_lib.MogpError):
        k.K(X)
    m = gpr.Exact(k, X, np.zeros(3), variance=[1.0, 1.0])
    with pytest.raises(_lib.MogpError):
        m.loss()
