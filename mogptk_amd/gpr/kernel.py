"""
Kernel base classes for the HIP path -- host-side mirror of mogptk/gpr/kernel.py.

A kernel on this path does two things on the host (O(C^2 Q) scalars) and nothing else:
  * `_spectral_terms(D)`   : constrained parameters -> unified spectral term table
                              [C, C, T, 2+3D] = [A, Psi, V_d.., M_d.., Delta_d..]   
  * `_spectral_backward(g)`: d loss / d table  ->  `.grad` on every raw parameter (the chain rule the
                              reference gets from autograd through gpr/multioutput.py:182-199 etc.)
All O(N^2) / O(N^3) work (Gram build, Cholesky, solves, gradient moments) runs in the HIP library
behind the C ABI (include/mogp_hip.h).  There is no CPU fallback.
"""
import copy
import numpy as np

from .config import config
from .parameter import Parameter, ParameterHolder


def term_width(D):
    return 2 + 3 * D


def env_width(D):
    """rows with a Gaussian envelope on the input midpoint: [A, Psi, V_d, M_d, Delta_d, L_d, c_d]"""
    return 2 + 5 * D


def pad_width(table, width):
    """a table of the narrow row width as one of the wide width (no envelope: L = 0)"""
    if table.shape[3] == width:
        return table
    out = np.zeros(table.shape[:3] + (width,))
    out[..., :table.shape[3]] = table
    return out


# One evaluation asks every kernel of a composition for its term table several times (push to the device, then again at each level
# of the chain rule -- AddKernel even asks only to learn T) while no parameter can change.  A model evaluation opens this cache.
_TERMS_CACHE = None


class terms_cache:
    """context: memoise `_spectral_terms` per kernel object for the duration of ONE model evaluation (tables are shared: read-only)"""

    def __enter__(self):
        global _TERMS_CACHE
        self._prev = _TERMS_CACHE
        _TERMS_CACHE = {}
        return self

    def __exit__(self, *exc):
        global _TERMS_CACHE
        _TERMS_CACHE = self._prev
        return False


def cached_terms(fn):
    def wrapper(self, D):
        cache = _TERMS_CACHE
        if cache is None:
            return fn(self, D)
        key = (id(self), D)
        if key not in cache:
            cache[key] = fn(self, D)
        return cache[key]
    wrapper.__doc__ = fn.__doc__
    return wrapper


class Kernel(ParameterHolder):
    """Base kernel (reference gpr/kernel.py:5-191)."""

    def __init__(self, input_dims=None, active_dims=None):
        if active_dims is not None:
            raise NotImplementedError("active_dims is not on the HIP path")
        self.input_dims = input_dims
        self.active_dims = None
        self.output_dims = None

    def name(self):
        return self.__class__.__name__

    def __setattr__(self, name, val):
        if name == "train":
            for p in self.parameters():
                p.train = val
            return
        super().__setattr__(name, val)

    def __call__(self, X1, X2=None):
        """validate + K, reference gpr/kernel.py:23-35"""
        X1, X2 = self._check_input(X1, X2)
        return self.K(X1, X2)

    def _check_input(self, X1, X2=None):
        """reference gpr/kernel.py:60-80"""
        X1 = np.asarray(X1.detach().cpu().numpy() if hasattr(X1, "detach") else X1, dtype=np.float64)
        if X1.ndim != 2:
            raise ValueError("X should have two dimensions (data_points,input_dims)")
        if X1.shape[0] == 0 or X1.shape[1] == 0:
            raise ValueError("X must not be empty")
        if X2 is not None:
            X2 = np.asarray(X2.detach().cpu().numpy() if hasattr(X2, "detach") else X2, dtype=np.float64)
            if X2.ndim != 2:
                raise ValueError("X should have two dimensions (data_points,input_dims)")
            if X2.shape[0] == 0:
                raise ValueError("X must not be empty")
            if X1.shape[1] != X2.shape[1]:
                raise ValueError("input dimensions for X1 and X2 must match")
        return X1, X2

    def _check_kernels(self, kernels, length=None):
        """Normalise what a combinator was given -- one kernel, several, or one list (reference gpr/kernel.py:82-110, same messages) -- into a
        list of `length` kernels that agree on input and output dimensions; a single kernel is cloned up to `length`."""
        if isinstance(kernels, tuple):
            items = list(kernels[0]) if (len(kernels) == 1 and isinstance(kernels[0], list)) else list(kernels)
        else:
            items = kernels if isinstance(kernels, list) else [kernels]
        if not items:
            raise ValueError("must pass at least one kernel")
        if length is not None and len(items) != length:
            if len(items) > 1:
                raise ValueError("must pass %d kernels" % length)
            items = items + [items[0].clone() for _ in range(length - 1)]
        if not all(isinstance(k, Kernel) for k in items):
            raise ValueError("must pass kernels")
        if len({k.input_dims for k in items}) > 1:
            raise ValueError("kernels must have same input dimensions")
        if len({k.output_dims for k in items if k.output_dims is not None}) > 1:
            raise ValueError("multi-output kernels must have same output dimensions")
        return items

    def clone(self):
        return copy.deepcopy(self)

    def iterkernels(self):
        yield self

    # -- the HIP seam ------------------------------------------------------------------------
    def _channels(self):
        """number of channels the device sees (single-output kernels run as one implicit channel)"""
        return 1 if self.output_dims is None else self.output_dims

    def _kernel_format(self, X):
        """single-output kernels have no channel column: prepend channel 0"""
        if self.output_dims is None:
            return np.concatenate([np.zeros((X.shape[0], 1)), X], axis=1)
        return X

    def _spectral_terms(self, D):
        raise NotImplementedError("%s is not on the MI355X spectral path (MOSM / SM / CSM are)" % self.name())

    def _spectral_backward(self, gtable):
        raise NotImplementedError("%s is not on the MI355X spectral path" % self.name())

    def _spectral_diag(self, D):
        """K_diag value per channel AS THE REFERENCE RETURNS IT (constant per channel for every spectral kernel).
        Default: the true diagonal sum_t A_cct (Delta = Psi = 0 on i == j blocks); SM overrides (its K_diag
        differs from diag K when D > 1, reference singleoutput.py:602-605)."""
        table = self._spectral_terms(D)
        C = table.shape[0]
        return np.array([np.sum(table[c, c, :, 0]) for c in range(C)])

    def _spectral_diag_backward(self, gc, D):
        """accumulate d loss / d K_diag[c] = gc[c] (K_diag as `_spectral_diag` defines it) into the raw gradients.
        Default: K_diag[c] = sum_t A_cct, so it is a table gradient on the diagonal amplitudes."""
        table = self._spectral_terms(D)
        gt = np.zeros_like(table)
        for c in range(table.shape[0]):
            gt[c, c, :, 0] = gc[c]
        self._spectral_backward(gt)

    def K(self, X1, X2=None):
        """Kernel matrix, reference gpr/kernel.py:138-150 (MO: :446-481).  Runs the HIP Gram builder."""
        from .._lib import gram
        X1k = self._kernel_format(np.asarray(X1, dtype=np.float64))
        X2k = None if X2 is None else self._kernel_format(np.asarray(X2, dtype=np.float64))
        D = X1k.shape[1] - 1
        return gram(config.device, self._channels(), D, self._spectral_terms(D), X1k, X2k)

    def K_diag(self, X1):
        """reference gpr/kernel.py:152-163, MO :483-495.  Constant per channel for every stationary spectral kernel; with an
        envelope (MOHSM) it follows the points."""
        X1k = self._kernel_format(np.asarray(X1, dtype=np.float64))
        D = X1k.shape[1] - 1
        table = self._spectral_terms(D)
        if table.shape[3] > term_width(D):
            return self._point_diag(table, X1k, D)
        return self._spectral_diag(D)[X1k[:, 0].astype(np.int64)]

    @staticmethod
    def _point_env(table, Xk, D):
        """per point k (channel c) and term t: the envelope exp(-1/2 sum_d L_d (x_k,d - c_d)^2) of the diagonal pair (c, c), and x - c"""
        c = Xk[:, 0].astype(np.int64)
        rows = table[c, c]                                        # (N, T, W)
        Lv, cn = rows[..., 2 + 3 * D:2 + 4 * D], rows[..., 2 + 4 * D:2 + 5 * D]
        a = Xk[:, None, 1:] - cn                                  # (N, T, D)
        return np.exp(-0.5 * np.sum(Lv * a * a, axis=2)), a, rows

    def _point_diag(self, table, Xk, D):
        """K_diag per point from an enveloped term table: sum_t A_cct env_t(x)   (Delta = Psi = 0 on diagonal pairs)"""
        env, _, rows = self._point_env(table, Xk, D)
        return np.sum(rows[..., 0] * env, axis=1)

    def _point_diag_table_grad(self, table, Xk, D, weights=None):
        """d [ sum_k w_k K_diag(x_k) ] / d table (w = 1: what the relative jitter, gpr/model.py:244, contributes per unit of d/d mean(diag) * N)"""
        env, a, rows = self._point_env(table, Xk, D)
        if weights is not None:
            env = env * np.asarray(weights, dtype=np.float64).reshape(-1, 1)
        c = Xk[:, 0].astype(np.int64)
        gt = np.zeros_like(table)
        A, Lv = rows[..., 0], rows[..., 2 + 3 * D:2 + 4 * D]
        for ch in range(table.shape[0]):
            k = c == ch
            if not np.any(k):
                continue
            gt[ch, ch, :, 0] = np.sum(env[k], axis=0)
            gt[ch, ch, :, 2 + 3 * D:2 + 4 * D] = np.sum((A[k] * env[k])[..., None] * (-0.5 * a[k] * a[k]), axis=0)
            gt[ch, ch, :, 2 + 4 * D:2 + 5 * D] = np.sum((A[k] * env[k])[..., None] * (Lv[k] * a[k]), axis=0)
        return gt

    def _point_diag_input_grad(self, table, Xk, D):
        """d K_diag(x_k) / d x_k,d per point: sum_t A env_t(x) (-L_d (x_d - c_d))   (N, D)"""
        env, a, rows = self._point_env(table, Xk, D)
        A, Lv = rows[..., 0], rows[..., 2 + 3 * D:2 + 4 * D]
        return np.sum((A * env)[..., None] * (-Lv * a), axis=1)

    def __add__(self, other):
        return AddKernel(self, other)

    def __mul__(self, other):
        return MulKernel(self, other)


class Kernels(Kernel):
    """Base kernel for list of kernels (reference gpr/kernel.py:193-230)."""

    def __init__(self, *kernels):
        super().__init__()
        kernels = self._check_kernels(kernels)
        i = 0
        while i < len(kernels):
            if isinstance(kernels[i], self.__class__):
                subkernels = list(kernels[i].kernels)
                kernels = kernels[:i] + subkernels + kernels[i + 1:]
                i += len(subkernels) - 1
            i += 1
        self.kernels = list(kernels)
        self.input_dims = kernels[0].input_dims
        output_dims = [kernel.output_dims for kernel in kernels if kernel.output_dims is not None]
        self.output_dims = None if len(output_dims) == 0 else output_dims[0]
        if any(k.output_dims != self.output_dims for k in kernels):
            raise NotImplementedError("mixing single- and multi-output kernels is not on the HIP path")

    def name(self):
        return "[%s]" % (",".join(kernel.name() for kernel in self.kernels),)

    def __getitem__(self, key):
        return self.kernels[key]

    def iterkernels(self):
        yield self
        for kernel in self.kernels:
            yield kernel


class AddKernel(Kernels):
    """Sum of kernels (reference gpr/kernel.py:232-246).  On the spectral path a sum of kernels is the
    concatenation of their term tables along T -- one fused pass instead of Q stacked N x N Grams (:243)."""

    def _spectral_diag(self, D):
        return sum(k._spectral_diag(D) for k in self.kernels)          # :245-246

    def _spectral_diag_backward(self, gc, D):
        for k in self.kernels:
            k._spectral_diag_backward(gc, D)

    @cached_terms
    def _spectral_terms(self, D):
        tabs = [k._spectral_terms(D) for k in self.kernels]
        width = max(t.shape[3] for t in tabs)                     # a sum with an enveloped kernel: everything in the wide rows
        return np.concatenate([pad_width(t, width) for t in tabs], axis=2)

    def _spectral_backward(self, gtable):
        t0 = 0
        D = self.input_dims if self.input_dims is not None else (gtable.shape[3] - 2) // 3
        for k in self.kernels:
            tab = k._spectral_terms(D)
            T = tab.shape[2]
            k._spectral_backward(gtable[:, :, t0:t0 + T, :tab.shape[3]])
            t0 += T


class MulKernel(Kernels):
    """Product kernel (reference gpr/kernel.py:248-262): not a sum of spectral terms -> out of scope."""

    def _spectral_terms(self, D):
        raise NotImplementedError("MulKernel is not on the MI355X spectral path")


class MixtureKernel(AddKernel):
    """Sum of Q copies of a kernel (reference gpr/kernel.py:264-276)."""

    def __init__(self, kernel, Q):
        if not issubclass(type(kernel), Kernel):
            raise ValueError("must pass kernel")
        kernels = self._check_kernels(kernel, Q)
        super().__init__(*kernels)


class MultiOutputKernel(Kernel):
    """Base class of multi-output kernels (reference gpr/kernel.py:381-520): column 0 of X holds the
    channel id.  The channel split / pair loop / scatter of :446-481 happens inside the HIP library."""

    def __init__(self, output_dims, input_dims=None, active_dims=None):
        super().__init__(input_dims, active_dims)
        self.output_dims = output_dims

    def _check_input(self, X1, X2=None):
        """reference gpr/kernel.py:398-404 (including its slip of re-checking X1 for X2's range)"""
        X1, X2 = super()._check_input(X1, X2)
        if not np.all(X1[:, 0] == np.trunc(X1[:, 0])) or not np.all(X1[:, 0] < self.output_dims):
            raise ValueError("X must have integers for the channel IDs in the first input dimension")
        if X2 is not None and not np.all(X2[:, 0] == np.trunc(X2[:, 0])) or not np.all(X1[:, 0] < self.output_dims):
            raise ValueError("X must have integers for the channel IDs in the first input dimension")
        return X1, X2
