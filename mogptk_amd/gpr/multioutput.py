"""
Multi-output spectral kernels on the HIP path -- host-side mirror of mogptk/gpr/multioutput.py for the
kernels the hot path covers: IndependentMultiOutputKernel (:5-39), MultiOutputSpectralMixtureKernel (MOSM, :125-210),
CrossSpectralKernel (CSM, :397-454) and -- SURVEY 8f-2, same term table -- MultiOutputSpectralKernel (:41-123) and
UncoupledMultiOutputSpectralKernel (uMOSM, :212-293).

Each class maps its constrained parameters to the unified spectral term table and back-propagates the
table gradient to its raw parameters; see gpr/kernel.py (this package) for the protocol.
"""
import ctypes

import numpy as np

from .config import config
from .parameter import Parameter, _STRUCTURE_EPOCH
from .kernel import Kernel, MultiOutputKernel, term_width, cached_terms

PI = np.pi


def _accumulate(p, gconstrained):
    """constrained-space gradient -> raw-space `.grad` (Softplus/Sigmoid link, reference parameter.py:48-49,77-78); a pegged
    parameter hands it on to the parameter it follows"""
    p.accumulate_grad(gconstrained)


class IndependentMultiOutputKernel(MultiOutputKernel):
    """One sub-kernel per channel on the block diagonal, zeros elsewhere (reference gpr/multioutput.py:5-39)."""

    def __init__(self, *kernels, output_dims=None):
        if output_dims is None:
            output_dims = len(kernels)
        super().__init__(output_dims)
        self.kernels = self._check_kernels(kernels, output_dims)
        self.input_dims = self.kernels[0].input_dims

    def __getitem__(self, key):
        return self.kernels[key]

    def name(self):
        return "%s[%s]" % (self.__class__.__name__, ",".join(k.name() for k in self.kernels))

    @cached_terms
    def _spectral_terms(self, D):
        subs = [k._spectral_terms(D)[0, 0] for k in self.kernels]      # each (T_c, W)
        T = max(s.shape[0] for s in subs)
        C = self.output_dims
        table = np.zeros((C, C, T, term_width(D)))                       # A = 0 off the block diagonal (:34)
        for c, s in enumerate(subs):
            table[c, c, :s.shape[0]] = s
        return table

    def _spectral_diag(self, D):
        return np.array([k._spectral_diag(D)[0] for k in self.kernels])   # reference :36-39

    def _spectral_diag_backward(self, gc, D):
        for c, k in enumerate(self.kernels):
            k._spectral_diag_backward(np.array([gc[c]]), D)

    def _spectral_backward(self, gtable):
        D = (gtable.shape[3] - 2) // 3
        for c, k in enumerate(self.kernels):
            T = k._spectral_terms(D).shape[2]
            k._spectral_backward(gtable[c:c + 1, c:c + 1, :T])


class MultiOutputSpectralMixtureKernel(MultiOutputKernel):
    """
    MOSM (reference gpr/multioutput.py:125-210).  Parameters: weight (C,Q), mean/variance/delay (C,Q,D),
    phase (C,Q); delay/phase are flagged train=False for one channel (:172-174).
    """

    def __init__(self, Q, output_dims, input_dims=1, active_dims=None):
        super().__init__(output_dims, input_dims, active_dims)
        self.input_dims = input_dims
        self.weight = Parameter(np.ones((output_dims, Q)), lower=config.positive_minimum)
        self.mean = Parameter(np.zeros((output_dims, Q, input_dims)), lower=config.positive_minimum)
        self.variance = Parameter(np.ones((output_dims, Q, input_dims)), lower=config.positive_minimum)
        self.delay = Parameter(np.zeros((output_dims, Q, input_dims)))
        self.phase = Parameter(np.zeros((output_dims, Q)))
        if output_dims == 1:
            self.delay.train = False
            self.phase.train = False
        self.twopi = np.power(2.0 * np.pi, float(self.input_dims) / 2.0)

    # hooks shared with the sibling kernels below (same pair algebra, different parameter layout / magnitude)
    _phase_scale = 1.0                                   # Psi = _phase_scale * (phase_i - phase_j), in cycles

    def _values(self):
        """constrained (weight, mean, variance, delay, phase) with a mixture axis: (C,Q), (C,Q,D) x3, (C,Q)"""
        return self.weight(), self.mean(), self.variance(), self.delay(), self.phase()

    def _magnitude(self, w):
        """(C,C,Q) magnitude of every channel pair: w_i w_j (reference :184,:192)"""
        return w[:, None] * w[None, :]

    def _amp_scale(self):
        """extra factor of every pair amplitude that does not depend on (weight, mean, variance, delay, phase); (C,C,Q) or a scalar"""
        return 1.0

    def _magnitude_backward(self, w, gmag):
        """gmag: d loss / d magnitude for pairs i >= j (zero above the diagonal) -> gradient of the constrained weight"""
        return np.sum(gmag * w[None, :], axis=1) + np.sum(gmag * w[:, None], axis=0)

    def _store(self, gw, gmu, gv, gth, gph):
        _accumulate(self.weight, gw)
        _accumulate(self.mean, gmu)
        _accumulate(self.variance, gv)
        if self.output_dims > 1:                   # with one channel delay/phase never enter the graph (grad None)
            _accumulate(self.delay, gth)
            _accumulate(self.phase, gph)

    def _memo_key(self):
        parts = []
        for p in (self.weight, self.mean, self.variance, self.delay, self.phase):
            parts.append(p.data.tobytes())
            for b in (p.lower, p.upper):
                parts.append(b"-" if b is None else np.asarray(b, dtype=np.float64).tobytes())
            if p.pegged:
                parts.append(p.pegged_parameter.data.tobytes())
        return b"|".join(parts)

    def _pairs(self):
        # one evaluation asks for the pair algebra three times (terms for the device, terms and pairs again in the chain rule) at the
        # SAME raw values: keep the last result, keyed on the raw parameter bytes (~60 doubles)
        key = self._memo_key()
        memo = self.__dict__.get("_pairs_memo")
        if memo is not None and memo[0] == key:
            return memo[1]
        out = self._pairs_compute()
        self.__dict__["_pairs_memo"] = (key, out)
        return out

    def _pairs_compute(self):
        w, mu, v, th, ph = self._values()
        vi, vj = v[:, None], v[None, :]                 # (C,C,Q,D) broadcast
        mi, mj = mu[:, None], mu[None, :]
        s = vi + vj
        inv = 1.0 / s
        dmu = mi - mj
        return w, mu, v, th, ph, vi, vj, mi, mj, s, inv, dmu

    @cached_terms
    def _spectral_terms(self, D):
        """reference gpr/multioutput.py:182-199 (memoised like _pairs; the returned table is shared -- treat it as read-only)"""
        if D != self.input_dims:
            raise ValueError("X must have %d input dimensions" % self.input_dims)
        key = self._memo_key()
        memo = self.__dict__.get("_terms_memo")
        if memo is not None and memo[0] == key:
            return memo[1]
        table = self._spectral_terms_compute(D)
        self.__dict__["_terms_memo"] = (key, table)
        return table

    def _derived(self):
        """what the term table and its chain rule share beyond _pairs(): E = exp(-pi^2 sum dmu^2 / s), V, M, sqrt(prod V) and the masks of the
        off-diagonal channel pairs (memoised with the pairs: one evaluation builds them once)"""
        memo = self.__dict__.get("_derived_memo")
        pairs = self._pairs()
        if memo is not None and memo[0] is pairs:
            return memo[1]
        w, mu, v, th, ph, vi, vj, mi, mj, s, inv, dmu = pairs
        C = self.output_dims
        E = np.exp(-PI ** 2 * np.sum(dmu * inv * dmu, axis=3))                                  # :192
        M = inv * (vi * mj + vj * mi)                                                           # :194
        V = 2.0 * vi * inv * vj                                                                 # :195
        rootV = np.sqrt(np.prod(V, axis=3))
        off2 = (1.0 - np.eye(C))[:, :, None]
        out = (E, M, V, rootV, off2, off2[..., None])
        self.__dict__["_derived_memo"] = (pairs, out)
        return out

    def _native(self):
        """the library's native pair algebra (csrc/hostalg.hip) -- for the plain MOSM kernel only: its siblings override pieces of it"""
        if type(self) is not MultiOutputSpectralMixtureKernel or config.dtype != np.float64:
            return None
        from .. import _lib
        try:
            return _lib.lib()
        except OSError:
            return None

    @staticmethod
    def _c(a):
        a = np.ascontiguousarray(a, dtype=np.float64)
        return a, a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))

    def _native_values(self):
        """constrained values as contiguous arrays + pointers, once per parameter set (the table push and the chain rule share them)"""
        key = self._memo_key()
        memo = self.__dict__.get("_native_memo")
        if memo is None or memo[0] != key:
            memo = (key, [np.ascontiguousarray(x, dtype=np.float64) for x in self._values()])       # arrays only: the memo is pickled with the kernel
            self.__dict__["_native_memo"] = memo
        return [(a, a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))) for a in memo[1]]

    def _spectral_terms_compute(self, D):
        C = self.output_dims
        lib = self._native()
        if lib is not None:
            vals = self._native_values()
            Q = vals[1][0].shape[1]
            table = np.empty((C, C, Q, term_width(D)))
            rc = lib.mogp_mosm_terms(C, Q, D, *[p for _, p in vals], float(self.twopi), float(self._phase_scale),
                                     table.ctypes.data_as(ctypes.POINTER(ctypes.c_double)))
            if rc != 0:
                raise RuntimeError("mogp_mosm_terms failed (%d)" % rc)
            return table
        w, mu, v, th, ph, vi, vj, mi, mj, s, inv, dmu = self._pairs()
        E, M, V, rootV, off2, off3 = self._derived()
        Q = mu.shape[1]
        table = np.empty((C, C, Q, term_width(D)))
        magw = self._magnitude(w)
        amp = self._amp_scale()
        table[..., 0] = magw * E * (self.twopi * rootV) * amp                                   # :192, :199
        table[..., 1] = self._phase_scale * (ph[:, None] - ph[None, :])                         # :197
        table[..., 2:2 + D] = V
        table[..., 2 + D:2 + 2 * D] = M
        table[..., 2 + 2 * D:] = th[:, None] - th[None, :]                                      # :196
        ampd = (np.asarray(amp) * np.ones((C, C, Q)))
        for c in range(C):                                                                      # i == j branch :183-187
            table[c, c, :, 0] = magw[c, c] * self.twopi * np.sqrt(np.prod(v[c], axis=1)) * ampd[c, c]
            table[c, c, :, 1] = 0.0
            table[c, c, :, 2:2 + D] = v[c]
            table[c, c, :, 2 + D:2 + 2 * D] = mu[c]
            table[c, c, :, 2 + 2 * D:] = 0.0
        return table

    def _spectral_backward(self, gtable):
        """Chain rule table -> (weight, mean, variance, delay, phase); gtable is zero for i < j and already
        carries the symmetric double count of off-diagonal channel blocks."""
        C = self.output_dims
        D = self.input_dims
        lib = self._native()
        if lib is not None:
            vals = self._native_values()
            Q = vals[1][0].shape[1]
            gt, gtp = self._c(gtable)
            gw, gph = np.empty((C, Q)), np.empty((C, Q))
            gmu, gv, gth = np.empty((C, Q, D)), np.empty((C, Q, D)), np.empty((C, Q, D))
            ptr = lambda a: a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))
            rc = lib.mogp_mosm_terms_backward(C, Q, D, *[p for _, p in vals], float(self.twopi), float(self._phase_scale), gtp,
                                              ptr(gw), ptr(gmu), ptr(gv), ptr(gth), ptr(gph))
            if rc != 0:
                raise RuntimeError("mogp_mosm_terms_backward failed (%d)" % rc)
            self._store(gw, gmu, gv, gth, gph)
            return
        w, mu, v, th, ph, vi, vj, mi, mj, s, inv, dmu = self._pairs()
        A = self._spectral_terms(D)[..., 0]
        gA, gPsi = gtable[..., 0], gtable[..., 1]
        gV, gM, gDl = gtable[..., 2:2 + D], gtable[..., 2 + D:2 + 2 * D], gtable[..., 2 + 2 * D:2 + 3 * D]

        E, M, V, rootV, o2, o3 = self._derived()                            # o2 / o3: 1 on off-diagonal channel pairs, 0 on the diagonal
        gAA3 = (gA * A * o2)[..., None]                                     # (C,C,Q,1)
        # d A / d magnitude = everything but the magnitude (kept explicit: a magnitude may be zero or negative for the uncoupled kernel)
        gmag = gA * (E * (self.twopi * rootV) * self._amp_scale())          # gtable is zero above the diagonal already
        gw = self._magnitude_backward(w, gmag)
        gPo = gPsi * o2
        gph = self._phase_scale * (np.sum(gPo, axis=1) - np.sum(gPo, axis=0))
        gDlo = gDl * o3
        gth = np.sum(gDlo, axis=1) - np.sum(gDlo, axis=0)
        gMi = gM * (o3 * inv)                                               # gM / s on the off-diagonal pairs
        gVo = gV * o3
        inv2 = inv * inv
        dA_dmu_i = gAA3 * (-2.0 * PI ** 2) * (dmu * inv)
        gmu = np.sum(dA_dmu_i + gMi * vj, axis=1) + np.sum(gMi * vi - dA_dmu_i, axis=0)
        dVi = 2.0 * (vj * vj) * inv2
        dVj = 2.0 * (vi * vi) * inv2
        common = gAA3 * (PI ** 2 * (dmu * dmu) * inv2)
        half = 0.5 * gAA3 / V
        gv_i = common + (half + gVo) * dVi + gMi * (mj - M)
        gv_j = common + (half + gVo) * dVj + gMi * (mi - M)
        gv = np.sum(gv_i, axis=1) + np.sum(gv_j, axis=0)
        for c in range(C):                                                  # i == j blocks (their magnitude part is in gmag already)
            gv[c] += (gA[c, c] * A[c, c])[:, None] / (2.0 * v[c]) + gV[c, c]
            gmu[c] += gM[c, c]
        self._store(gw, gmu, gv, gth, gph)


class MultiOutputSpectralKernel(MultiOutputSpectralMixtureKernel):
    """
    One MOSM component without the mixture axis (reference gpr/multioutput.py:41-123); use `MixtureKernel(MultiOutputSpectralKernel(...), Q)`.
    Parameters: weight (C,), mean/variance/delay (C,D), phase (C,).
    """

    def __init__(self, output_dims, input_dims=1, active_dims=None):
        MultiOutputKernel.__init__(self, output_dims, input_dims, active_dims)
        self.input_dims = input_dims
        self.weight = Parameter(np.ones(output_dims), lower=config.positive_minimum)
        self.mean = Parameter(np.zeros((output_dims, input_dims)), lower=config.positive_minimum)
        self.variance = Parameter(np.ones((output_dims, input_dims)), lower=config.positive_minimum)
        self.delay = Parameter(np.zeros((output_dims, input_dims)))
        self.phase = Parameter(np.zeros(output_dims))
        if output_dims == 1:
            self.delay.train = False
            self.phase.train = False
        self.twopi = np.power(2.0 * np.pi, float(self.input_dims) / 2.0)

    def _values(self):
        return (self.weight()[:, None], self.mean()[:, None], self.variance()[:, None], self.delay()[:, None], self.phase()[:, None])

    def _store(self, gw, gmu, gv, gth, gph):
        super()._store(gw[:, 0], gmu[:, 0], gv[:, 0], gth[:, 0], gph[:, 0])


class UncoupledMultiOutputSpectralKernel(MultiOutputSpectralKernel):
    """
    uMOSM (reference gpr/multioutput.py:212-293): like MultiOutputSpectralKernel, but the pair magnitudes are the entries of
    tril(weight) tril(weight)^T with an unconstrained (C,C) weight, and the phase difference enters the cosine WITHOUT the 2 pi factor
    (reference :283 -- kept).  Use `MixtureKernel(UncoupledMultiOutputSpectralKernel(...), Q)`.
    """
    _phase_scale = 1.0 / (2.0 * np.pi)

    def __init__(self, output_dims, input_dims=1, active_dims=None):
        super().__init__(output_dims, input_dims, active_dims)
        del self.__dict__["weight"]
        self.__dict__["_order"].remove("weight")
        self.__dict__["_order"].insert(0, "weight")
        _STRUCTURE_EPOCH[0] += 1
        self.weight = Parameter(np.tril(np.ones((output_dims, output_dims))))
        self.weight.num_parameters = int((output_dims * output_dims + output_dims) / 2)

    def _values(self):
        return (self.weight(), self.mean()[:, None], self.variance()[:, None], self.delay()[:, None], self.phase()[:, None])

    def _magnitude(self, w):
        t = np.tril(w)
        return (t @ t.T)[:, :, None]

    def _magnitude_backward(self, w, gmag):
        g = gmag[:, :, 0]
        return np.tril((g + g.T) @ np.tril(w))

    def _store(self, gw, gmu, gv, gth, gph):
        MultiOutputSpectralMixtureKernel._store(self, gw, gmu[:, 0], gv[:, 0], gth[:, 0], gph[:, 0])


class CrossSpectralKernel(MultiOutputKernel):
    """
    CSM component (reference gpr/multioutput.py:397-454); use `MixtureKernel(CrossSpectralKernel(...), Q)`.
    Parameters: amplitude (C,Rq), mean (D,), variance (D,), shift (C,Rq).
    """

    def __init__(self, output_dims, input_dims=1, Rq=1, active_dims=None):
        super().__init__(output_dims, input_dims, active_dims)
        self.input_dims = input_dims
        self.Rq = Rq
        self.amplitude = Parameter(np.ones((output_dims, Rq)), lower=config.positive_minimum)
        self.mean = Parameter(np.zeros(input_dims), lower=config.positive_minimum)
        self.variance = Parameter(np.ones(input_dims), lower=config.positive_minimum)
        self.shift = Parameter(np.zeros((output_dims, Rq)))

    @cached_terms
    def _spectral_terms(self, D):
        """reference gpr/multioutput.py:432-449"""
        if D != self.input_dims:
            raise ValueError("X must have %d input dimensions" % self.input_dims)
        C, Rq = self.output_dims, self.Rq
        amp, mu, var, sh = self.amplitude(), self.mean(), self.variance(), self.shift()
        table = np.empty((C, C, Rq, term_width(D)))
        table[..., 0] = np.sqrt(amp[:, None] * amp[None, :])        # :444
        table[..., 1] = sh[:, None] - sh[None, :]                   # :441
        table[..., 2:2 + D] = var
        table[..., 2 + D:2 + 2 * D] = mu
        table[..., 2 + 2 * D:] = 0.0
        for c in range(C):                                          # i == j branch :433-439
            table[c, c, :, 0] = amp[c]
            table[c, c, :, 1] = 0.0
        return table

    def _spectral_backward(self, gtable):
        C, D = self.output_dims, self.input_dims
        amp = self.amplitude()
        table = self._spectral_terms(D)
        A = table[..., 0]
        gA, gPsi = gtable[..., 0], gtable[..., 1]
        off = ~np.eye(C, dtype=bool)[:, :, None]
        gAA = np.where(off, gA * A, 0.0)
        gamp = np.sum(gAA / (2.0 * amp[:, None]), axis=1) + np.sum(gAA / (2.0 * amp[None, :]), axis=0)
        gPo = np.where(off, gPsi, 0.0)
        gsh = np.sum(gPo, axis=1) - np.sum(gPo, axis=0)
        for c in range(C):
            gamp[c] += gA[c, c]
        _accumulate(self.amplitude, gamp)
        _accumulate(self.variance, np.sum(gtable[..., 2:2 + D], axis=(0, 1, 2)))
        _accumulate(self.mean, np.sum(gtable[..., 2 + D:2 + 2 * D], axis=(0, 1, 2)))
        if C > 1:
            _accumulate(self.shift, gsh)


class LinearModelOfCoregionalizationKernel(MultiOutputKernel):
    """
    LMC (reference gpr/multioutput.py:456-502): K_ij = sum_q (sum_r w_iqr w_jqr) k_q(x, x') over Q single-output base kernels.
    On the spectral path every base kernel must provide a term table (SpectralKernel / SpectralMixtureKernel): the LMC table is
    their concatenation along T with the amplitudes scaled by B_q[i, j] = sum_r w_iqr w_jqr -- still ONE fused Gram pass.
    Parameters: weight (C, Q, Rq) > 0, then the base kernels' own.
    """

    def __init__(self, *kernels, output_dims, input_dims=1, Q=None, Rq=1):
        super().__init__(output_dims, input_dims)
        if Q is None:
            Q = len(kernels)
        kernels = self._check_kernels(kernels, Q)
        if any(k.output_dims is not None for k in kernels):
            raise NotImplementedError("LMC over multi-output base kernels is not on the HIP path")
        self.input_dims = input_dims
        self.kernels = list(kernels)
        self.weight = Parameter(np.ones((output_dims, Q, Rq)), lower=config.positive_minimum)

    def __getitem__(self, key):
        return self.kernels[key]

    def name(self):
        return "%s[%s]" % (self.__class__.__name__, ",".join(k.name() for k in self.kernels))

    def iterkernels(self):
        yield self
        for kernel in self.kernels:
            yield kernel

    def _coreg(self):
        w = self.weight()                                                   # (C,Q,Rq)
        return np.einsum("iqr,jqr->ijq", w, w)                               # B_q[i,j]  (:493)

    @cached_terms
    def _spectral_terms(self, D):
        B = self._coreg()
        C = self.output_dims
        parts = []
        for q, k in enumerate(self.kernels):
            sub = k._spectral_terms(D)[0, 0]                                 # (T_q, W): Psi = Delta = 0 for single-output kernels
            part = np.broadcast_to(sub, (C, C) + sub.shape).copy()
            part[..., 0] = B[:, :, q, None] * sub[None, None, :, 0]
            parts.append(part)
        return np.concatenate(parts, axis=2)

    def _spectral_diag(self, D):
        """reference :497-502: sum_q (sum_r w_cqr^2) K_diag_q -- with the base kernel's own K_diag convention"""
        B = self._coreg()
        kd = np.array([k._spectral_diag(D)[0] for k in self.kernels])        # (Q,)
        return np.einsum("ccq,q->c", B, kd)

    def _weight_backward(self, gB):
        """gB[i,j,q]: d loss / d B_q[i,j] over the pairs the loss actually uses (any pattern) -> weight"""
        w = self.weight()
        _accumulate(self.weight, np.einsum("ijq,jqr->iqr", gB, w) + np.einsum("ijq,iqr->jqr", gB, w))

    def _spectral_diag_backward(self, gc, D):
        B = self._coreg()
        C = self.output_dims
        gB = np.zeros_like(B)
        for q, k in enumerate(self.kernels):
            kd = k._spectral_diag(D)[0]
            for c in range(C):
                gB[c, c, q] = gc[c] * kd
            k._spectral_diag_backward(np.array([np.sum(gc * B[np.arange(C), np.arange(C), q])]), D)
        self._weight_backward(gB)

    def _spectral_backward(self, gtable):
        D = (gtable.shape[3] - 2) // 3
        B = self._coreg()
        gB = np.zeros_like(B)
        t0 = 0
        for q, k in enumerate(self.kernels):
            sub = k._spectral_terms(D)[0, 0]
            T = sub.shape[0]
            g = gtable[:, :, t0:t0 + T, :]                                   # zero above the diagonal, double count included
            gB[:, :, q] = np.sum(g[..., 0] * sub[None, None, :, 0], axis=2)
            gsub = np.sum(g, axis=(0, 1))                                    # V, M columns: the same base value in every pair
            gsub[:, 0] = np.einsum("ijt,ij->t", g[..., 0], B[:, :, q])
            k._spectral_backward(gsub[None, None])
            t0 += T
        self._weight_backward(gB)


class GaussianConvolutionProcessKernel(MultiOutputKernel):
    """
    CONV (reference gpr/multioutput.py:504-553): K_ij = w_i w_j sqrt(prod b / prod s_ij) exp(-1/2 sum_d tau_d^2 / s_ij,d) with
    s_ij = v_i + v_j + b.  One Gaussian term per channel pair in the term table (V = 1 / s_ij, no cosine: M = Delta = Psi = 0); the
    reference's `X2 is None` branch (:536-540) is the i == j case of the same expression.  Use `MixtureKernel(..., Q)`.
    Parameters: weight (C,) > 0, variance (C, D) >= 0, base_variance (D,) > 0.
    """

    def __init__(self, output_dims, input_dims=1, active_dims=None):
        super().__init__(output_dims, input_dims, active_dims)
        self.input_dims = input_dims
        self.weight = Parameter(np.ones(output_dims), lower=config.positive_minimum)
        self.variance = Parameter(np.ones((output_dims, input_dims)), lower=0.0)
        self.base_variance = Parameter(np.ones(input_dims), lower=config.positive_minimum)

    def _parts(self):
        w, v, b = self.weight(), self.variance(), self.base_variance()
        s = v[:, None, :] + v[None, :, :] + b                               # (C,C,D)
        A = w[:, None] * w[None, :] * np.sqrt(np.prod(b) / np.prod(s, axis=2))
        return w, v, b, s, A

    @cached_terms
    def _spectral_terms(self, D):
        if D != self.input_dims:
            raise ValueError("X must have %d input dimensions" % self.input_dims)
        w, v, b, s, A = self._parts()
        C = self.output_dims
        table = np.zeros((C, C, 1, term_width(D)))
        table[:, :, 0, 0] = A
        table[:, :, 0, 2:2 + D] = 1.0 / s
        return table

    def _spectral_backward(self, gtable):
        D = self.input_dims
        w, v, b, s, A = self._parts()
        gA = gtable[:, :, 0, 0]                                               # pairs i >= j, double count included
        gV = gtable[:, :, 0, 2:2 + D]
        gAA = gA * A
        gs = -0.5 * gAA[:, :, None] / s - gV / (s * s)                       # d loss / d s_ij,d
        _accumulate(self.weight, (np.sum(gAA, axis=1) + np.sum(gAA, axis=0)) / w)
        _accumulate(self.variance, np.sum(gs, axis=1) + np.sum(gs, axis=0))   # s_ij depends on v_i and on v_j (twice on v_i when i == j)
        _accumulate(self.base_variance, np.sum(gs, axis=(0, 1)) + 0.5 * np.sum(gAA) / b)


class MultiOutputHarmonizableSpectralKernel(MultiOutputSpectralKernel):
    """
    MOHSM component (reference gpr/multioutput.py:295-395): a MultiOutputSpectralKernel pair term times a Gaussian envelope on the input
    MIDPOINT, exp(-1/2 l_ij sum_d ((x_d + x'_d)/2 - c_d)^2) -- non-stationary.  Use `MixtureKernel(MultiOutputHarmonizableSpectralKernel(...), Q)`.
    Parameters: weight (C,), mean / variance (C,D), lengthscale (C,), center (D,), delay (C,D), phase (C,).  As in the reference: the
    amplitude carries (2 pi)^D (not ^(D/2)) and l_ij^(D/2), the envelope precision is lengthscale^2 (harmonic mean over a channel pair), and
    the phase difference enters the cosine without the 2 pi factor.
    On the device this is a term row of width 2 + 5 D: [A, Psi, V_d, M_d, Delta_d, L_d = l_ij, c_d].
    """
    _phase_scale = 1.0 / (2.0 * np.pi)

    def __init__(self, output_dims, input_dims=1, active_dims=None):
        super().__init__(output_dims, input_dims, active_dims)
        # registration order of the reference: weight, mean, variance, lengthscale, center, delay, phase
        order = self.__dict__["_order"]
        delay, phase = self.__dict__.pop("delay"), self.__dict__.pop("phase")
        order.remove("delay"); order.remove("phase")
        self.lengthscale = Parameter(np.ones(output_dims), lower=config.positive_minimum)
        self.center = Parameter(np.zeros(input_dims))
        self.delay = delay
        self.phase = phase
        self.twopi = np.power(2.0 * np.pi, float(self.input_dims))

    def _memo_key(self):
        parts = [super()._memo_key()]
        for p in (self.lengthscale, self.center):
            parts.append(p.data.tobytes())
            for b in (p.lower, p.upper):
                parts.append(b"-" if b is None else np.asarray(b, dtype=np.float64).tobytes())
            if p.pegged:
                parts.append(p.pegged_parameter.data.tobytes())
        return b"|".join(parts)

    def _pair_precision(self):
        """l_ij (C,C): lengthscale_i^2 on the diagonal, 2 l_i l_j / (l_i + l_j) off it (reference :349, :357-366)"""
        l = np.square(self.lengthscale())
        li, lj = l[:, None], l[None, :]
        Lp = 2.0 * li * lj / (li + lj)
        Lp[np.diag_indices(self.output_dims)] = l
        return Lp, li, lj

    def _amp_scale(self):
        Lp, _, _ = self._pair_precision()
        return np.power(Lp, 0.5 * self.input_dims)[:, :, None]

    def _spectral_terms_compute(self, D):
        narrow = super()._spectral_terms_compute(D)
        C = self.output_dims
        Lp, _, _ = self._pair_precision()
        table = np.zeros((C, C, 1, 2 + 5 * D))
        table[..., :2 + 3 * D] = narrow
        table[..., 2 + 3 * D:2 + 4 * D] = Lp[:, :, None, None]
        table[..., 2 + 4 * D:2 + 5 * D] = self.center()[None, None, None, :]
        return table

    def _spectral_diag(self, D):
        raise NotImplementedError("the diagonal of a harmonizable kernel follows the points: use K_diag(X)")

    def _spectral_backward(self, gtable):
        C, D = self.output_dims, self.input_dims
        table = self._spectral_terms(D)
        super()._spectral_backward(gtable)                        # weight, mean, variance, delay, phase (the amplitude scale is in Fm / A)
        Lp, li, lj = self._pair_precision()
        l = self.lengthscale()
        gA, A = gtable[..., 0, 0], table[..., 0, 0]               # one term per component
        gLp = np.sum(gtable[..., 0, 2 + 3 * D:2 + 4 * D], axis=2) + gA * A * (0.5 * D) / Lp          # zero above the diagonal already
        off = ~np.eye(C, dtype=bool)
        dLi = 2.0 * lj * lj / np.square(li + lj)                  # d l_ij / d l_i  (l_i = lengthscale_i^2), i != j
        dLj = 2.0 * li * li / np.square(li + lj)
        g_l2 = np.sum(np.where(off, gLp * dLi, 0.0), axis=1) + np.sum(np.where(off, gLp * dLj, 0.0), axis=0) + np.diagonal(gLp)
        _accumulate(self.lengthscale, g_l2 * 2.0 * l)
        _accumulate(self.center, np.sum(gtable[..., 0, 2 + 4 * D:2 + 5 * D], axis=(0, 1)))
