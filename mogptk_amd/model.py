"""
Training / prediction driver -- host-side mirror of mogptk/model.py (Exact factory :76-100, Model :180-664).

`Model.train()` runs the reference's optimiser loop (mogptk/model.py:541-566) with every loss/gradient
evaluation being one native call into the HIP library; Adam/SGD/AdaGrad are restated in numpy over the
P <= a few hundred raw scalars exactly as torch.optim does (defaults, bias correction, per-parameter step
count, parameters with grad None skipped).
"""
import os
import time
import math
import pickle
import inspect
import logging
import numpy as np

from . import gpr
from .dataset import DataSet
from .util import (mean_absolute_error, mean_absolute_percentage_error, symmetric_mean_absolute_percentage_error,
                   mean_squared_error, root_mean_squared_error)

logger = logging.getLogger("mogptk")


def LoadModel(filename, allow=None, trusted=False):
    """reference mogptk/model.py:62-74.  Reads this package's own checkpoints and -- through mogptk_amd.compat, without the reference
    installed -- files written by the reference's `Model.save()`.

    The reference unpickles whatever the file names.  Here a checkpoint of this package may name classes defined in this package, numpy arrays
    and a few builtins; a model saved with the caller's own code in it -- a mean function, a Kernel / Likelihood / transformer subclass -- needs
    `allow=[MyMean, ...]` (exactly those objects are admitted as well), or `trusted=True` (plain pickle.load, the reference's behaviour: only
    for files you wrote).  The refusal message of a blocked load says which name was blocked."""
    filename += ".npy"
    with open(filename, "rb") as r:
        raw = r.read()
    from . import compat
    if compat.is_reference_checkpoint(raw):
        return compat.load_reference_model(raw)
    return compat.load_native_model(raw, allow=allow or (), trusted=trusted)


class Exact:
    """
    Exact inference (reference mogptk/model.py:76-100).

    Args:
        variance (float): Variance of the Gaussian likelihood (default: 1.0 per channel).
        data_variance: fixed per-point variances added to the diagonal.
        jitter (float): Relative jitter added before the Cholesky.
    """

    def __init__(self, variance=None, data_variance=None, jitter=1e-8):
        self.variance, self.data_variance, self.jitter = variance, data_variance, jitter

    def _build(self, kernel, x, y, y_err=None, mean=None):
        """the gpr model this choice of inference stands for.  Defaults as the reference's: unit noise variance -- one per channel under a
        multi-output kernel --, and the data's own error bars as fixed per-point variances when none were given"""
        channels = kernel.output_dims
        noise = self.variance if self.variance is not None else (1.0 if channels is None else [1.0] * channels)
        fixed = self.data_variance
        if fixed is None and y_err is not None:
            fixed = np.square(y_err)
        return gpr.Exact(kernel, x, y, variance=noise, data_variance=fixed, jitter=self.jitter, mean=mean)


class Titsias:
    """
    Sparse inference of Titsias 2009 (reference mogptk/model.py:140-157).

    Args:
        inducing_points (int, list): number of inducing points (PER CHANNEL for multi-output kernels) or locations.
        init_inducing_points (str): `grid`, `random`, or `density`.
        variance (float): variance of the Gaussian likelihood.
        jitter (float): relative jitter added before the Cholesky.
    """

    def __init__(self, inducing_points=10, init_inducing_points="grid", variance=1.0, jitter=1e-6):
        self.inducing_points = inducing_points
        self.init_inducing_points = init_inducing_points
        self.variance = variance
        self.jitter = jitter

    def _build(self, kernel, x, y, y_err=None, mean=None):
        return gpr.Titsias(kernel, x, y, Z=self.inducing_points, Z_init=self.init_inducing_points,
                           variance=self.variance, jitter=self.jitter, mean=mean)


class Snelson:
    """
    Sparse inference of Snelson & Ghahramani 2005 (reference mogptk/model.py:102-120).

    Args:
        inducing_points (int, list): number of inducing points (PER CHANNEL for multi-output kernels) or locations.
        init_inducing_points (str): `grid`, `random`, or `density`.
        variance (float, array): variance of the Gaussian likelihood (a float: one trained variance; (channels,): one per channel).
        jitter (float): relative jitter added before the Cholesky.
    """

    def __init__(self, inducing_points=10, init_inducing_points="grid", variance=None, jitter=1e-6):
        self.inducing_points = inducing_points
        self.init_inducing_points = init_inducing_points
        self.variance = variance
        self.jitter = jitter

    def _build(self, kernel, x, y, y_err=None, mean=None):
        variance = self.variance
        if variance is None:
            variance = [1.0] * kernel.output_dims if kernel.output_dims is not None else 1.0
        return gpr.Snelson(kernel, x, y, Z=self.inducing_points, Z_init=self.init_inducing_points, variance=variance, jitter=self.jitter, mean=mean)


class OpperArchambeau:
    """
    Variational inference of Opper & Archambeau 2009 (reference mogptk/model.py:125-138).  Any likelihood of gpr/likelihood.py (Gaussian by default; the non-Gaussian ones through Gauss-Hermite quadrature on the host).

    Args:
        likelihood (gpr.Likelihood): likelihood p(y|f) (default: Gaussian with unit scale).
        jitter (float): kept for the signature (the reference's model adds none).
    """

    def __init__(self, likelihood=None, jitter=1e-6):
        self.likelihood = likelihood
        self.jitter = jitter

    def _build(self, kernel, x, y, y_err=None, mean=None):
        likelihood = self.likelihood if self.likelihood is not None else gpr.GaussianLikelihood(1.0)
        return gpr.OpperArchambeau(kernel, x, y, likelihood=likelihood, jitter=self.jitter, mean=mean)


class Hensman:
    """
    Variational inference of Hensman et al. 2015 (reference mogptk/model.py:159-178).  Any likelihood of gpr/likelihood.py (Gaussian by default; the non-Gaussian ones through Gauss-Hermite quadrature on the host).

    Args:
        inducing_points (int, list): number of inducing points (PER CHANNEL for multi-output kernels) or locations; None (default): the
            non-sparse model, whose variational distribution lives on the data points.
        init_inducing_points (str): `grid`, `random`, or `density`.
        likelihood (gpr.Likelihood): likelihood p(y|f) (default: Gaussian with unit scale).
        jitter (float): relative jitter added before the Cholesky.
    """

    def __init__(self, inducing_points=None, init_inducing_points="grid", likelihood=None, jitter=1e-6):
        self.inducing_points = inducing_points
        self.init_inducing_points = init_inducing_points
        self.likelihood = likelihood
        self.jitter = jitter

    def _build(self, kernel, x, y, y_err=None, mean=None):
        likelihood = self.likelihood if self.likelihood is not None else gpr.GaussianLikelihood(1.0)
        if self.inducing_points is None:
            return gpr.Hensman(kernel, x, y, likelihood=likelihood, jitter=self.jitter, mean=mean)
        return gpr.SparseHensman(kernel, x, y, Z=self.inducing_points, Z_init=self.init_inducing_points, likelihood=likelihood, jitter=self.jitter, mean=mean)


# ---- optimisers over raw parameters (torch.optim semantics) -----------------------------------------
class _Adam:
    """torch.optim.Adam(params, lr=1e-3, betas=(0.9,0.999), eps=1e-8, weight_decay=0, amsgrad=False)"""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, amsgrad=False, **unused):
        self.params = list(params)
        self.lr, self.betas, self.eps, self.wd, self.amsgrad = lr, betas, eps, weight_decay, amsgrad
        self.state = {}

    def step(self):
        """One Adam update (torch.optim.Adam's arithmetic, element for element).  The raw parameters of a model are a handful of tiny tensors
        (64 numbers at configs[1]): updated tensor by tensor, the ~10 small numpy calls per tensor were a third of the host's share of a training
        step, so the float64 tensors that have a gradient are updated as ONE flat vector -- the same elementwise operations in the same order."""
        b1, b2 = self.betas
        flat = [p for p in self.params if p.grad is not None and p.data.dtype == np.float64 and p.grad.dtype == np.float64]
        if len(flat) > 1:
            key = tuple(id(p) for p in flat)
            st = self.state.get("flat")
            if st is None or st["key"] != key:
                # (re)build the flat state from the per-tensor states, so that a change of the set of tensors loses nothing: the old flat state goes
                # back into per-tensor states FIRST, whether or not the new set shares a tensor with it
                self._unflatten()
                parts = [self._tensor_state(p) for p in flat]
                st = dict(key=key, t=[q["t"] for q in parts], m=np.concatenate([q["m"].ravel() for q in parts]),
                          v=np.concatenate([q["v"].ravel() for q in parts]), vmax=np.concatenate([q["vmax"].ravel() for q in parts]),
                          ends=np.cumsum([p.data.size for p in flat]))
                self.state["flat"] = st
            g = np.concatenate([p.grad.ravel() for p in flat])
            x = np.concatenate([p.data.ravel() for p in flat])
            if self.wd != 0.0:
                g = g + self.wd * x
            st["t"] = [t + 1 for t in st["t"]]
            st["m"] = b1 * st["m"] + (1.0 - b1) * g
            st["v"] = b2 * st["v"] + (1.0 - b2) * g * g
            v = st["v"]
            if self.amsgrad:
                st["vmax"] = np.maximum(st["vmax"], v)
                v = st["vmax"]
            if len(set(st["t"])) == 1:
                bc1, bc2 = 1.0 - b1 ** st["t"][0], 1.0 - b2 ** st["t"][0]
                x = x - (self.lr / bc1) * st["m"] / (np.sqrt(v) / math.sqrt(bc2) + self.eps)
            else:                                           # tensors that joined later carry their own step counts
                sizes = np.diff(np.concatenate([[0], st["ends"]]))
                bc1 = np.repeat([1.0 - b1 ** t for t in st["t"]], sizes)
                bc2 = np.repeat([1.0 - b2 ** t for t in st["t"]], sizes)
                x = x - (self.lr / bc1) * st["m"] / (np.sqrt(v) / np.sqrt(bc2) + self.eps)
            lo = 0
            for p, hi in zip(flat, st["ends"]):
                p.data = x[lo:hi].reshape(p.data.shape)
                lo = hi
            done = set(key)
        else:
            self._unflatten()
            done = set()
        for p in self.params:
            if p.grad is None or id(p) in done:
                continue
            g = p.grad
            if self.wd != 0.0:
                g = g + self.wd * p.data
            st = self._tensor_state(p)
            st["t"] += 1
            st["m"] = b1 * st["m"] + (1.0 - b1) * g
            st["v"] = b2 * st["v"] + (1.0 - b2) * g * g
            bc1 = 1.0 - b1 ** st["t"]
            bc2 = 1.0 - b2 ** st["t"]
            v = st["v"]
            if self.amsgrad:
                st["vmax"] = np.maximum(st["vmax"], v)
                v = st["vmax"]
            denom = np.sqrt(v) / math.sqrt(bc2) + self.eps
            p.data = p.data - (self.lr / bc1) * st["m"] / denom

    def _tensor_state(self, p):
        """the per-tensor state (created on first use); a tensor that is part of the flat vector has its state THERE until _unflatten"""
        self._unflatten(only=id(p))
        return self.state.setdefault(id(p), dict(t=0, m=np.zeros_like(p.data), v=np.zeros_like(p.data), vmax=np.zeros_like(p.data)))

    def _unflatten(self, only=None):
        """write the flat state back into per-tensor states (the set of tensors with a gradient changed)"""
        st = self.state.get("flat")
        if st is None or (only is not None and only not in st["key"]):
            return
        del self.state["flat"]
        byid = {id(p): p for p in self.params}
        lo = 0
        for pid, hi, t in zip(st["key"], st["ends"], st["t"]):
            shape = byid[pid].data.shape
            self.state[pid] = dict(t=t, m=st["m"][lo:hi].reshape(shape).copy(), v=st["v"][lo:hi].reshape(shape).copy(),
                                   vmax=st["vmax"][lo:hi].reshape(shape).copy())
            lo = hi


class _LBFGS:
    """torch.optim.LBFGS(params, lr=1, max_iter=20, max_eval=None, tolerance_grad=1e-7, tolerance_change=1e-9, history_size=100,
    line_search_fn=None) over the raw parameters: ONE call of step(closure) runs up to max_iter iterations, as the reference uses it
    (mogptk/model.py:541-553).  Two-loop recursion with the y.s > 1e-10 curvature guard, first step min(1, 1/|g|_1) lr, optional
    'strong_wolfe' line search (bracketing by bounded cubic extrapolation, then zoom; c1 = 1e-4, c2 = 0.9)."""

    def __init__(self, params, lr=1.0, max_iter=20, max_eval=None, tolerance_grad=1e-7, tolerance_change=1e-9, history_size=100,
                 line_search_fn=None):
        if not 0.0 <= lr:
            raise ValueError("Invalid learning rate: %s" % lr)
        self.params = list(params)
        self.lr, self.max_iter = float(lr), int(max_iter)
        self.max_eval = int(max_eval) if max_eval is not None else self.max_iter * 5 // 4
        self.tol_grad, self.tol_change, self.history, self.line_search = tolerance_grad, tolerance_change, int(history_size), line_search_fn
        self.state = dict(func_evals=0, n_iter=0)

    # flat views over the raw tensors
    def _grad(self):
        return np.concatenate([(np.zeros(p.data.size) if p.grad is None else np.asarray(p.grad, dtype=np.float64).reshape(-1)) for p in self.params])

    def _move(self, t, d):
        o = 0
        for p in self.params:
            n = p.data.size
            p.data = p.data + t * d[o:o + n].reshape(p.data.shape)
            o += n

    def _snapshot(self):
        return [p.data.copy() for p in self.params]

    def _restore(self, x):
        for p, v in zip(self.params, x):
            p.data = v.copy()

    @staticmethod
    def _cubic(x1, f1, g1, x2, f2, g2, bounds=None):
        lo, hi = bounds if bounds is not None else ((x1, x2) if x1 <= x2 else (x2, x1))
        d1 = g1 + g2 - 3.0 * (f1 - f2) / (x1 - x2)
        disc = d1 * d1 - g1 * g2
        if disc >= 0.0:
            d2 = math.sqrt(disc)
            if x1 <= x2:
                pos = x2 - (x2 - x1) * ((g2 + d2 - d1) / (g2 - g1 + 2.0 * d2))
            else:
                pos = x1 - (x1 - x2) * ((g1 + d2 - d1) / (g1 - g2 + 2.0 * d2))
            return min(max(pos, lo), hi)
        return 0.5 * (lo + hi)

    def _strong_wolfe(self, closure, x0, t, d, f, g, gtd, max_ls, c1=1e-4, c2=0.9, tol=1e-9):
        def probe(step):
            self._move(step, d)
            fv = float(closure())
            gv = self._grad()
            self._restore(x0)
            return fv, gv

        d_norm = float(np.max(np.abs(d)))
        f_new, g_new = probe(t)
        evals, it = 1, 0
        gtd_new = float(g_new @ d)
        t_prev, f_prev, g_prev, gtd_prev = 0.0, f, g.copy(), gtd
        done, br = False, None
        while it < max_ls:
            if f_new > f + c1 * t * gtd or (it > 1 and f_new >= f_prev):       # sufficient decrease violated: the minimum is bracketed
                br = dict(t=[t_prev, t], f=[f_prev, f_new], g=[g_prev, g_new.copy()], gtd=[gtd_prev, gtd_new])
                break
            if abs(gtd_new) <= -c2 * gtd:                                          # strong Wolfe conditions hold
                br = dict(t=[t], f=[f_new], g=[g_new], gtd=[gtd_new])
                done = True
                break
            if gtd_new >= 0.0:                                                     # slope changed sign: bracketed
                br = dict(t=[t_prev, t], f=[f_prev, f_new], g=[g_prev, g_new.copy()], gtd=[gtd_prev, gtd_new])
                break
            lo, hi, keep = t + 0.01 * (t - t_prev), t * 10.0, t
            t = self._cubic(t_prev, f_prev, gtd_prev, t, f_new, gtd_new, bounds=(lo, hi))
            t_prev, f_prev, g_prev, gtd_prev = keep, f_new, g_new.copy(), gtd_new
            f_new, g_new = probe(t)
            evals += 1
            gtd_new = float(g_new @ d)
            it += 1
        if it == max_ls:
            br = dict(t=[0.0, t], f=[f, f_new], g=[g, g_new], gtd=[gtd, gtd_new])
        stalled = False
        low, high = (0, 1) if br["f"][0] <= br["f"][-1] else (1, 0)
        while not done and it < max_ls:
            if abs(br["t"][1] - br["t"][0]) * d_norm < tol:
                break
            t = self._cubic(br["t"][0], br["f"][0], br["gtd"][0], br["t"][1], br["f"][1], br["gtd"][1])
            tmax, tmin = max(br["t"]), min(br["t"])
            eps = 0.1 * (tmax - tmin)
            if min(tmax - t, t - tmin) < eps:
                if stalled or t >= tmax or t <= tmin:
                    t = tmax - eps if abs(t - tmax) < abs(t - tmin) else tmin + eps
                    stalled = False
                else:
                    stalled = True
            else:
                stalled = False
            f_new, g_new = probe(t)
            evals += 1
            gtd_new = float(g_new @ d)
            it += 1
            if f_new > f + c1 * t * gtd or f_new >= br["f"][low]:
                br["t"][high], br["f"][high], br["g"][high], br["gtd"][high] = t, f_new, g_new.copy(), gtd_new
                low, high = (0, 1) if br["f"][0] <= br["f"][1] else (1, 0)
            else:
                if abs(gtd_new) <= -c2 * gtd:
                    done = True
                elif gtd_new * (br["t"][high] - br["t"][low]) >= 0.0:
                    br["t"][high], br["f"][high], br["g"][high], br["gtd"][high] = br["t"][low], br["f"][low], br["g"][low], br["gtd"][low]
                br["t"][low], br["f"][low], br["g"][low], br["gtd"][low] = t, f_new, g_new.copy(), gtd_new
        return br["f"][low], br["g"][low], br["t"][low], evals

    def step(self, closure):
        st = self.state
        first = closure()
        loss = float(first)
        evals = 1
        st["func_evals"] += 1
        g = self._grad()
        if np.max(np.abs(g)) <= self.tol_grad:
            return first
        d, t = st.get("d"), st.get("t")
        ys_hist, s_hist, rho = st.get("old_dirs", []), st.get("old_stps", []), st.get("ro", [])
        h0, g_prev, loss_prev = st.get("H_diag", 1.0), st.get("prev_flat_grad"), st.get("prev_loss")
        n = 0
        while n < self.max_iter:
            n += 1
            st["n_iter"] += 1
            if st["n_iter"] == 1:
                d, ys_hist, s_hist, rho, h0 = -g, [], [], [], 1.0
            else:
                y, sv = g - g_prev, d * t
                ys = float(y @ sv)
                if ys > 1e-10:
                    if len(ys_hist) == self.history:
                        ys_hist.pop(0); s_hist.pop(0); rho.pop(0)
                    ys_hist.append(y); s_hist.append(sv); rho.append(1.0 / ys)
                    h0 = ys / float(y @ y)
                k = len(ys_hist)
                al = [0.0] * k
                q = -g
                for i in range(k - 1, -1, -1):
                    al[i] = float(s_hist[i] @ q) * rho[i]
                    q = q - al[i] * ys_hist[i]
                d = q * h0
                for i in range(k):
                    be = float(ys_hist[i] @ d) * rho[i]
                    d = d + (al[i] - be) * s_hist[i]
            g_prev, loss_prev = g.copy(), loss
            t = min(1.0, 1.0 / float(np.sum(np.abs(g)))) * self.lr if st["n_iter"] == 1 else self.lr
            gtd = float(g @ d)
            if gtd > -self.tol_change:
                break
            ls_evals = 0
            if self.line_search is not None:
                if self.line_search != "strong_wolfe":
                    raise RuntimeError("only 'strong_wolfe' is supported")
                x0 = self._snapshot()
                loss, g, t, ls_evals = self._strong_wolfe(closure, x0, t, d, loss, g, gtd, self.max_eval - evals)
                self._move(t, d)
                converged = np.max(np.abs(g)) <= self.tol_grad
            else:
                self._move(t, d)
                converged = False
                if n != self.max_iter:
                    loss = float(closure())
                    g = self._grad()
                    converged = np.max(np.abs(g)) <= self.tol_grad
                    ls_evals = 1
            evals += ls_evals
            st["func_evals"] += ls_evals
            if n == self.max_iter or evals >= self.max_eval or converged:
                break
            if np.max(np.abs(d * t)) <= self.tol_change or abs(loss - loss_prev) < self.tol_change:
                break
        st.update(d=d, t=t, old_dirs=ys_hist, old_stps=s_hist, ro=rho, H_diag=h0, prev_flat_grad=g_prev, prev_loss=loss_prev)
        return first


class _SGD:
    """torch.optim.SGD(params, lr=1e-3, momentum=0, dampening=0, weight_decay=0, nesterov=False)"""

    def __init__(self, params, lr=1e-3, momentum=0.0, dampening=0.0, weight_decay=0.0, nesterov=False, **unused):
        self.params = list(params)
        self.lr, self.mom, self.damp, self.wd, self.nesterov = lr, momentum, dampening, weight_decay, nesterov
        self.state = {}

    def step(self):
        for p in self.params:
            if p.grad is None:
                continue
            g = p.grad
            if self.wd != 0.0:
                g = g + self.wd * p.data
            if self.mom != 0.0:
                if id(p) not in self.state:
                    buf = self.state[id(p)] = np.array(g)
                else:
                    buf = self.state[id(p)] = self.mom * self.state[id(p)] + (1.0 - self.damp) * g
                g = g + self.mom * buf if self.nesterov else buf
            p.data = p.data - self.lr * g


class _Adagrad:
    """torch.optim.Adagrad(params, lr=1e-2, lr_decay=0, weight_decay=0, initial_accumulator_value=0, eps=1e-10)"""

    def __init__(self, params, lr=1e-2, lr_decay=0.0, weight_decay=0.0, initial_accumulator_value=0.0, eps=1e-10, **unused):
        self.params = list(params)
        self.lr, self.lr_decay, self.wd, self.eps = lr, lr_decay, weight_decay, eps
        self.state = {id(p): dict(t=0, s=np.full_like(p.data, initial_accumulator_value)) for p in self.params}

    def step(self):
        for p in self.params:
            if p.grad is None:
                continue
            st = self.state[id(p)]
            st["t"] += 1
            g = p.grad
            if self.wd != 0.0:
                g = g + self.wd * p.data
            clr = self.lr / (1.0 + (st["t"] - 1) * self.lr_decay)
            st["s"] = st["s"] + g * g
            p.data = p.data - clr * g / (np.sqrt(st["s"]) + self.eps)


def _format_time(t):
    hours = int(t / 3600)
    minutes = int((t % 3600) / 60)
    seconds = int(t % 60)
    return "%3d:%02d:%02d" % (hours, minutes, seconds)


def _format_duration(t):
    s = ""
    if 3600 <= t:
        s += "%d hours " % int(t / 3600)
    if 60 <= t:
        s += "%d minutes " % int((t % 3600) / 60)
    return s + "%.3f seconds" % (t % 60)


class Model:
    def __init__(self, dataset, kernel, inference=Exact(), mean=None, name=None):
        """
        Base class of multi-output GP models (reference mogptk/model.py:181-236).

        Args:
            dataset (DataSet, Data): data of all channels.
            kernel (mogptk_amd.gpr.Kernel): the kernel.
            inference: inference model factory, e.g. `mogptk_amd.Exact()`.
            mean: mean function hook (fixed functions only).
            name (str): name of the model.

        Attributes:
            dataset, gpr, times, losses, errors: as the reference.
        """
        if not isinstance(dataset, DataSet):
            dataset = DataSet(dataset)
        if dataset.get_output_dims() == 0:
            raise ValueError("dataset must have at least one channel")
        names = [n for n in dataset.get_names() if n is not None]
        if len(set(names)) != len(names):
            raise ValueError("all data channels must have unique names")

        self.name = name
        self.dataset = dataset
        self.is_multioutput = kernel.output_dims is not None

        X, Y = self.dataset.get_train_data()
        x, y = self._to_kernel_format(X, Y)

        y_err = None
        if all(channel.Y_err is not None for channel in self.dataset):
            Y_err = [channel.Y_err[channel.mask] for channel in self.dataset]
            lo = [self.dataset[j].Y_transformer.forward(Y[j] - Y_err[j], X[j]) for j in range(len(self.dataset))]
            up = [self.dataset[j].Y_transformer.forward(Y[j] + Y_err[j], X[j]) for j in range(len(self.dataset))]
            y_err = (np.concatenate(up, axis=0) - np.concatenate(lo, axis=0)) / 2.0
        self.gpr = inference._build(kernel, x, y, y_err, mean)

        self.iters = 0
        self.times = np.zeros(0)
        self.losses = np.zeros(0)
        self.errors = np.zeros(0)

    def __str__(self):
        s = "Model: %s\n" % self.gpr._get_name()
        s += "‣ Kernel: %s\n" % self.gpr.kernel.name()
        s += "‣ Likelihood: %s\n" % self.gpr.likelihood.name()
        s += "‣ Parameters: %d\n" % self.num_parameters()
        for p in self.gpr.parameters():
            s += "  - %s %s\n" % (p._name, p.shape)
        s += "‣ Channels: %d\n" % len(self.dataset)
        s += "‣ Training points: %d\n" % self.num_training_points()
        return s

    def print_parameters(self):
        self.gpr.print_parameters()

    def parameters(self):
        """reference mogptk/model.py:266-276"""
        return list(self.gpr.parameters())

    def num_parameters(self):
        """reference mogptk/model.py:296-306 (the only place `Parameter.train` matters: quirk Q3)"""
        return sum(p.num_parameters if p.train else 0 for p in self.gpr.parameters())

    def num_training_points(self):
        return sum(len(channel.get_train_data()[1]) for channel in self.dataset)

    def save(self, filename, reference=False):
        """reference mogptk/model.py:320-336 (pickles the whole model; device handles are dropped and rebuilt lazily).

        reference=True writes the file in the REFERENCE's format instead -- `mogptk.LoadModel` of the reference reads it (and so does
        LoadModel here): Exact / Titsias models of the six wrappers, see mogptk_amd.compat.dump_reference_model for the scope."""
        filename += ".npy"
        if reference:
            from . import compat
            raw = compat.dump_reference_model(self)           # before the old file goes: a model outside the writer's scope leaves it alone
        try:
            os.remove(filename)
        except OSError:
            pass
        with open(filename, "wb") as w:
            if reference:
                w.write(raw)
            else:
                pickle.dump(self, w)

    def log_marginal_likelihood(self):
        """reference mogptk/model.py:338-348"""
        return float(self.gpr.log_marginal_likelihood())

    def BIC(self):
        return self.num_parameters() * np.log(self.num_training_points()) - 2.0 * self.log_marginal_likelihood()

    def AIC(self):
        return 2.0 * self.num_parameters() - 2.0 * self.log_marginal_likelihood()

    def loss(self):
        """reference mogptk/model.py:374-384"""
        return float(self.gpr.loss())

    _ERROR_MEASURES = {"mae": mean_absolute_error, "mape": mean_absolute_percentage_error,
                       "smape": symmetric_mean_absolute_percentage_error, "mse": mean_squared_error,
                       "rmse": root_mean_squared_error}

    def error(self, method="MAE", use_all_data=False):
        """Prediction error on the held-out points (all points when none are held out or `use_all_data`), in the data's own units: the
        device predicts in transformed space, every channel's slice goes back through its Y transformer.  `method`: MAE, MAPE, sMAPE, MSE,
        RMSE, a function (y_true, y_pred), or a function of the model alone.  Same contract as reference mogptk/model.py:386-439."""
        takes_model = callable(method) and len(inspect.signature(method).parameters) == 1
        if takes_model:
            return method(self)
        if not callable(method):
            measure = self._ERROR_MEASURES.get(str(method).lower())
            if measure is None:
                raise ValueError("valid error calculation methods are MAE, MAPE, sMAPE, MSE, and RMSE")
        else:
            measure = method
        held_out = any(self.dataset.has_test_data()) and not use_all_data
        X, truth = self.dataset.get_test_data() if held_out else self.dataset.get_data()
        flat = np.reshape(self.gpr.predict_y(self._to_kernel_format(X)), -1)
        ends = np.cumsum([len(x) for x in X])
        predicted = [channel.Y_transformer.backward(part, x)
                     for channel, part, x in zip(self.dataset, np.split(flat, ends[:-1]), X)]
        return measure(np.concatenate(truth), np.concatenate(predicted))

    def train(self, method="Adam", iters=500, verbose=False, error=None, plot=False, jit=None, **kwargs):
        """
        Optimise the hyper-parameters (reference mogptk/model.py:441-579): `iters` optimiser steps and
        `iters+1` loss evaluations; `times/losses/errors` are continued across calls, the optimiser state is not.

        Args:
            method (str): LBFGS, Adam, SGD or AdaGrad.
            iters (int): number of iterations (the maximum for LBFGS).
            verbose (bool): print progress (at most every ~10 s).
            error (str, function): prediction error evaluated per iteration.
            plot (bool): accepted; plotting is out of scope.
            jit (bool): accepted and ignored (one native call per evaluation already).
            **kwargs: passed to the optimiser (lr=..., betas=..., ...).

        Returns:
            numpy.ndarray: losses, numpy.ndarray: errors.
        """
        error_use_all_data = False
        if error is not None and all(not channel.has_test_data() for channel in self.dataset):
            error_use_all_data = True
        if callable(error):
            if len(inspect.signature(error).parameters) == 1:
                e = error(self)
            else:
                e = error(np.zeros((1, 1)), np.zeros((1, 1)))
            if not isinstance(e, float) and (not isinstance(e, np.ndarray) or e.size != 1):
                raise ValueError("error function must return a float")

        if method.lower() in ("l-bfgs", "lbfgs", "l-bfgs-b", "lbfgsb"):
            method = "LBFGS"
        elif method.lower() == "adam":
            method = "Adam"
        elif method.lower() == "sgd":
            method = "SGD"
        elif method.lower() == "adagrad":
            method = "AdaGrad"
        else:
            raise ValueError("optimizer must be LBFGS, Adam, SGD, or AdaGrad")

        if verbose:
            print("Starting optimization using", method)
            print("‣ Model: %s" % self.gpr.name())
            print("  ‣ Kernel: %s" % self.gpr.kernel.name())
            print("  ‣ Likelihood: %s" % self.gpr.likelihood.name())
            print("‣ Channels: %d" % len(self.dataset))
            print("‣ Parameters: %d" % self.num_parameters())
            print("‣ Training points: %d" % self.num_training_points())
            print("‣ Iterations: %d" % iters)

        # the trace of this call behind the trace of the earlier ones: row 0 seconds, row 1 loss, row 2 error; the last entry of an earlier call
        # (its closing evaluation) is this call's first
        done = max(int(self.times.shape[0]) - 1, 0)
        trace = np.zeros((3, done + iters + 1))
        if done:
            trace[0, :done] = self.times[:done]
            trace[1, :done] = self.losses[:done]
            if self.errors.shape[0] == self.times.shape[0]:
                trace[2, :done] = self.errors[:done]
        t_start = time.time()
        quiet_until = 0.0                              # seconds into the call before which no progress line is due
        width = 1 if iters == 0 else int(math.log10(done + iters)) + 1

        def progress(i, loss, last=False):
            nonlocal quiet_until, trace
            at = done + i
            if at >= trace.shape[1]:                   # (LBFGS may evaluate more often than it was given iterations)
                trace = np.concatenate((trace, np.zeros((3, at + 1 - trace.shape[1]))), axis=1)
            now = time.time() - t_start
            trace[0, at], trace[1, at] = now, loss
            if error is not None:
                trace[2, at] = float(self.error(error, error_use_all_data))
            if verbose and (last or now >= quiet_until):
                line = "  %*d/%*d %s  loss=%12g" % (width, at, width, done + iters, _format_time(now), trace[1, at])
                if error is not None:
                    line += "  error=%12g" % trace[2, at]
                print(line)
                quiet_until += 10.0 * (1 + int((now - quiet_until) / 10.0))

        params = list(self.gpr.parameters())
        if method == "LBFGS":
            # one optimizer.step(closure) runs the whole optimisation; losses are indexed by the function-evaluation count and `iters`
            # becomes that count afterwards (reference model.py:541-553)
            if "max_iter" not in kwargs:
                kwargs["max_iter"] = iters
            else:
                iters = kwargs["max_iter"]
            optimizer = _LBFGS(params, **kwargs)

            def closure():
                i = int(optimizer.state["func_evals"])
                value = self.loss()
                progress(i, value)
                return value
            optimizer.step(closure)
            iters = int(optimizer.state["func_evals"])
        else:
            if method == "Adam":
                optimizer = _Adam(params, **kwargs)
            elif method == "SGD":
                optimizer = _SGD(params, **kwargs)
            else:
                optimizer = _Adagrad(params, **kwargs)
            for i in range(iters):
                progress(i, self.loss())
                optimizer.step()
        progress(iters, self.loss(), last=True)

        if verbose:
            print("Optimization finished in %s" % _format_duration(time.time() - t_start))

        kept = done + iters + 1
        self.iters = kept - 1
        self.times, self.losses = trace[0, :kept].copy(), trace[1, :kept].copy()
        if error is not None:
            self.errors = trace[2, :kept].copy()
        return trace[1], trace[2]

    # ---- predictions ---------------------------------------------------------------------
    def _to_kernel_format(self, X, Y=None):
        """reference mogptk/model.py:585-606: concatenate channels in order, prepend the channel id column,
        apply the per-channel Y transformers."""
        x = np.concatenate(X, axis=0)
        if self.is_multioutput:
            chan = np.concatenate([j * np.ones(len(X[j])) for j in range(len(X))]).reshape(-1, 1)
            x = np.concatenate([chan, x], axis=1)
        if Y is None:
            return x
        Y = list(Y)
        for j in range(len(Y)):
            Y[j] = self.dataset[j].Y_transformer.forward(Y[j], X[j])
        y = np.concatenate(Y, axis=0).reshape(-1, 1)
        return x, y

    def predict(self, X=None, ci=None, sigma=2, n=10000, transformed=False):
        """
        reference mogptk/model.py:608-664.  Returns (X, mu, lower, upper) as lists per channel, or bare arrays
        for a single channel.
        """
        X = self.dataset.get_prediction_data() if X is None else self.dataset._format_X(X)
        if isinstance(ci, float):                       # a coverage -> the two quantile limits around the median
            ci = [0.5 - 0.5 * ci, 0.5 + 0.5 * ci]
        if ci is not None:
            ci = [max(0.0, ci[0]), min(1.0, ci[1])]
        stacked = self.gpr.predict_y(self._to_kernel_format(X), ci, sigma=sigma, n=n)        # (mean, lower, upper), channels stacked in order

        edges = np.cumsum([0] + [Xj.shape[0] for Xj in X])
        per_channel = []                                 # [channel] -> [mean, lower, upper]
        for j, Xj in enumerate(X):
            undo = None if transformed else self.dataset[j].Y_transformer.backward
            parts = [np.squeeze(v[edges[j]:edges[j + 1]]) for v in stacked]
            per_channel.append(parts if undo is None else [undo(v, Xj) for v in parts])
        if len(self.dataset) == 1:
            return (X[0],) + tuple(per_channel[0])
        mu, lower, upper = ([c[k] for c in per_channel] for k in range(3))
        return X, mu, lower, upper

    def sample(self, X=None, n=None, prior=False, transformed=False):
        """Draws of y at X (the behaviour of reference mogptk/model.py:692-734; the posterior's full covariance comes from the device).  As in
        the reference `prior` is accepted and unused, and the stacked draws are cut into channels along their FIRST axis whatever n is."""
        X = self.dataset.get_prediction_data() if X is None else self.dataset._format_X(X)
        draws = np.asarray(self.gpr.sample_y(Z=self._to_kernel_format(X), n=n))
        edges = np.cumsum([0] + [Xj.shape[0] for Xj in X])              # rows of each channel inside the stacked prediction inputs
        per_channel = []
        for j, Xj in enumerate(X):
            block = draws[edges[j]:edges[j + 1]]
            undo = None if transformed else self.dataset[j].Y_transformer.backward
            if n is None:
                block = np.squeeze(block)
                if undo is not None:
                    block = undo(block, Xj)
            elif undo is not None:
                for k in range(n):                                       # one draw per column
                    block[:, k] = undo(block[:, k], Xj)
            per_channel.append(block)
        return per_channel[0] if len(per_channel) == 1 else per_channel

    def K(self, X1, X2=None):
        """reference mogptk/model.py:666-700"""
        X1 = self._to_kernel_format(self.dataset._format_X(X1))
        if X2 is not None:
            X2 = self._to_kernel_format(self.dataset._format_X(X2))
        return self.gpr.K(X1, X2)
