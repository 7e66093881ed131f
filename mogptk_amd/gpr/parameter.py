"""
Constrained parameters for the HIP path -- the host-side mirror of mogptk/gpr/parameter.py.

Same surface as the reference's `Parameter` (assign / __call__ / constrained / numpy / peg / lower /
upper / train / data / grad), with numpy arrays in place of torch tensors: the raw (unconstrained)
value lives in `.data`, the gradient of the loss with respect to the RAW value in `.grad`
(that is the coordinate system torch.optim.Adam sees in the reference, mogptk/model.py:557).

Behavioural quirks reproduced on purpose:
  Q1  Softplus.inverse misplaces `lower` (reference parameter.py:59) -> assign(1.0)() == 1.0000001
  Q2  assign(lower=/upper=) without a value re-interprets the RAW data as a constrained value
      (reference parameter.py:253-254)
  Q3  `train=False` only affects num_parameters(), the optimiser still updates the tensor.
"""
import copy
import sys
import numpy as np

from .config import config


def _asarray(value):
    """host tensor in config.dtype (float64, or float32 after use_single_precision(): reference parameter.py:206-218 `to_tensor`)"""
    if isinstance(value, Parameter):
        return np.array(value.constrained, dtype=config.dtype)
    if hasattr(value, "detach"):            # torch tensors are accepted at the boundary
        value = value.detach().cpu().numpy()
    return np.array(value, dtype=config.dtype)


class Transform:
    def forward(self, x):
        raise NotImplementedError()

    def inverse(self, y):
        raise NotImplementedError()

    def dforward(self, x):
        """d constrained / d raw (elementwise)."""
        raise NotImplementedError()


class Softplus(Transform):
    """y = lower + log(1 + exp(beta x))/beta, linear above beta*x > threshold  (reference parameter.py:30-59)."""

    def __init__(self, lower=0.0, beta=0.1, threshold=20.0):
        self.beta = beta
        self.lower = lower
        self.threshold = threshold

    def forward(self, x):
        z = self.beta * x
        big = z > self.threshold
        if not np.any(big):                # (every training step of every model passes here several times: the same values without two selects and a context manager)
            return self.lower + np.log1p(np.exp(z)) / self.beta
        with np.errstate(over="ignore"):
            sp = np.where(big, x, np.log1p(np.exp(np.where(big, 0.0, z))) / self.beta)
        return self.lower + sp

    def dforward(self, x):
        z = self.beta * x
        big = z > self.threshold
        if not np.any(big):
            return 1.0 / (1.0 + np.exp(-z))
        with np.errstate(over="ignore"):
            return np.where(big, 1.0, 1.0 / (1.0 + np.exp(-np.where(big, 0.0, z))))

    def inverse(self, y):
        if abs(self.beta) <= 1e-9 * max(abs(self.beta), 0.0):
            return 0.0
        elif self.beta < 0.0:
            if np.any(self.lower < y):
                raise ValueError("values must be smaller than %s" % self.lower)
        elif np.any(y < self.lower):
            raise ValueError("values must be greater than %s" % self.lower)
        # quirk Q1: `lower` sits outside the beta product, exactly as reference parameter.py:59
        with np.errstate(divide="ignore", invalid="ignore"):
            return (y - self.lower) + np.log(-np.expm1(-self.beta * y - self.lower)) / self.beta


class Sigmoid(Transform):
    """y = lower + (upper-lower) sigmoid(x)  (reference parameter.py:61-96)."""

    def __init__(self, lower=0.0, upper=1.0):
        self.lower = lower
        self.upper = upper

    def _sig(self, x):
        with np.errstate(over="ignore"):
            return 1.0 / (1.0 + np.exp(-x))

    def forward(self, x):
        return self.lower + (self.upper - self.lower) * self._sig(x)

    def dforward(self, x):
        s = self._sig(x)
        return (self.upper - self.lower) * s * (1.0 - s)

    def inverse(self, y):
        if np.any(y < self.lower) or np.any(self.upper < y):
            raise ValueError("values must be between %s and %s" % (self.lower, self.upper))
        y = (y - self.lower) / (self.upper - self.lower)
        close = np.isclose(self.lower * np.ones_like(y), self.upper * np.ones_like(y))
        y = np.where(close, sys.float_info.epsilon, y)
        with np.errstate(divide="ignore"):
            return np.log(y) - np.log(1 - y)


def _vjp(fn, x, g):
    """g^T d fn(x)/dx for a small numpy-analytic `fn` (a peg transform): complex-step Jacobian, column by column; central
    differences when `fn` does not accept complex input.  x has at most a few hundred entries."""
    x = np.asarray(x, dtype=np.float64)
    g = np.asarray(g, dtype=np.float64).reshape(-1)
    out = np.zeros(x.size)
    flat = x.reshape(-1)
    for k in range(x.size):
        try:
            z = flat.astype(np.complex128)
            z[k] += 1e-30j
            col = np.imag(np.asarray(fn(z.reshape(x.shape)))).reshape(-1) / 1e-30
            if not np.all(np.isfinite(col)):
                raise ValueError
        except (TypeError, ValueError):            # the transform does not take complex input (or is not analytic there); anything else is a bug in it
            h = 1e-6 * max(1.0, abs(flat[k]))
            a, b = flat.copy(), flat.copy()
            a[k] += h
            b[k] -= h
            col = (np.asarray(fn(a.reshape(x.shape)), dtype=np.float64) - np.asarray(fn(b.reshape(x.shape)), dtype=np.float64)).reshape(-1) / (2 * h)
        out[k] = np.dot(g, col)
    return out.reshape(x.shape)


class Parameter:
    """
    Parameter trained in an unconstrained space (reference parameter.py:99-346).

    Args:
        value: initial value in the CONSTRAINED space.
        name (str), lower, upper, prior, train: as the reference.
    """

    def __init__(self, value, name=None, lower=None, upper=None, prior=None, train=True):
        value = _asarray(value)
        self.data = value
        self.grad = None
        self._name = name
        self.lower = None
        self.upper = None
        self.prior = prior
        self.train = train
        self.transform = None
        self.pegged_parameter = None
        self.pegged_transform = None
        self.num_parameters = int(np.prod(value.shape))
        self.assign(value, lower=lower, upper=upper)

    # -- tensor-like surface -----------------------------------------------------------------
    @property
    def shape(self):
        return self.data.shape

    @property
    def ndim(self):
        return self.data.ndim

    def __repr__(self):
        name = self._name
        if self.pegged:
            name = self.pegged_parameter._name
        if name is None:
            return "{}".format(self.constrained.tolist())
        return "{}={}".format(self._name, self.constrained.tolist())

    def __call__(self):
        return self.constrained

    def clone(self):
        return copy.deepcopy(self)

    @property
    def pegged(self):
        return self.pegged_parameter is not None

    @property
    def constrained(self):
        """reference parameter.py:186-201"""
        if self.pegged:
            other = self.pegged_parameter.constrained
            if self.pegged_transform is not None:
                other = self.pegged_transform(other)
            return other
        if self.transform is not None:
            return self.transform.forward(self.data)
        return self.data

    def dconstrained(self):
        """d constrained / d raw, elementwise; the link the reference gets from autograd through
        Softplus/Sigmoid.forward (parameter.py:48-49,77-78)."""
        if self.transform is not None:
            return self.transform.dforward(self.data)
        return np.ones_like(self.data)

    def accumulate_grad(self, gconstrained):
        """add d loss / d constrained-value to the raw-space `.grad` of the parameter that OWNS the value.  A pegged parameter owns
        nothing: the reference's autograd routes its gradient through `pegged_transform` to the parameter it follows
        (parameter.py:186-201), and the pegged tensor itself keeps grad None (quirk Q3)."""
        g = np.asarray(gconstrained, dtype=np.float64)
        if self.pegged:
            other = self.pegged_parameter
            if self.pegged_transform is not None:
                g = _vjp(self.pegged_transform, other.constrained, g)
            other.accumulate_grad(np.reshape(g, other.data.shape))
            return
        g = (g * self.dconstrained()).astype(self.data.dtype, copy=False)       # .grad lives in the parameter's own dtype, like autograd's
        self.grad = g if self.grad is None else self.grad + g

    def numpy(self):
        return np.array(self.constrained)

    @staticmethod
    def to_tensor(value):
        return _asarray(value)

    @staticmethod
    def to_transform(lower, upper):
        """the transform the bounds call for (reference parameter.py:220-230): both -> sigmoid between them, one -> softplus away from it (beta
        0.1 above a lower bound, -0.1 below an upper one), none -> the raw value is the value"""
        bounded = (lower is not None, upper is not None)
        if bounded == (False, False):
            return None
        if bounded == (True, True):
            if np.any(upper < lower):
                raise ValueError("lower limit %s must be lower than upper limit %s" % (lower, upper))
            return Sigmoid(lower=lower, upper=upper)
        return Softplus(lower=lower) if bounded[0] else Softplus(lower=upper, beta=-0.1)

    def _fit_shape(self, arr, value, what):
        if arr.ndim != 0:
            while arr.ndim < value.ndim and value.shape[arr.ndim] == 1:
                arr = arr[..., None]
            while value.ndim < arr.ndim and arr.shape[-1] == 1:
                arr = arr[..., 0]
            if arr.shape != value.shape:
                raise ValueError("%s and value must match shapes: %s != %s" % (what, arr.shape, value.shape))
        return arr

    def assign(self, value=None, name=None, lower=None, upper=None, prior=None, train=None):
        """reference parameter.py:232-319 (same order of operations, same quirks)."""
        if value is not None:
            value = _asarray(value)
            origshape = value.shape
            while value.ndim < self.ndim and self.shape[value.ndim] == 1:
                value = value[..., None]
            while self.ndim < value.ndim and value.shape[-1] == 1:
                value = value[..., 0]
            if value.shape != self.shape:
                raise ValueError("parameter shape must match: %s != %s" % (origshape, self.shape))
        else:
            value = self.data          # quirk Q2: RAW data re-interpreted as a constrained value

        if lower is not None:
            lower = self._fit_shape(_asarray(lower), value, "lower")
        else:
            lower = self.lower
        if upper is not None:
            upper = self._fit_shape(_asarray(upper), value, "upper")
        else:
            upper = self.upper

        if name is None:
            name = self._name
        elif self._name is not None:
            idx = self._name.rfind(".")
            if idx != -1:
                name = self._name[: idx + 1] + name
        if prior is None:
            prior = self.prior
        if train is None:
            train = True if self.pegged else self.train

        transform = Parameter.to_transform(lower, upper)
        if transform is not None:
            if lower is not None:
                value = np.where(value < lower, lower * np.ones_like(value), value)
            if upper is not None:
                value = np.where(upper < value, upper * np.ones_like(value), value)
            value = transform.inverse(value)

        self._name = name
        self.data = np.array(value, dtype=config.dtype)
        self.lower = lower
        self.upper = upper
        self.prior = prior
        self.train = train
        self.transform = transform
        self.pegged_parameter = None
        self.pegged_transform = None

    def peg(self, other, transform=None):
        """follow `other` (optionally through `transform`) instead of being trained: reference parameter.py:321-335, same refusals.  Chains of
        pegged parameters are not allowed, so a gradient has exactly one hop to make (gpr/model.py: accumulate_grad through the peg)."""
        if not isinstance(other, Parameter):
            raise ValueError("parameter must be pegged to other parameter object")
        if other.pegged:
            raise ValueError("cannot peg parameter to another pegged parameter")
        self.pegged_parameter, self.pegged_transform, self.train = other, transform, False

    def log_prior(self):
        """reference parameter.py:337-346.  Priors are out of the hot-path scope (all None in the configs)."""
        if self.prior is None:
            return 0.0
        raise NotImplementedError("parameter priors are not on the HIP path")


_STRUCTURE_EPOCH = [0]


class ParameterHolder:
    """Minimal stand-in for torch.nn.Module's parameter registry: attributes that are Parameters (or
    holders, or lists of holders) are enumerated in registration order, like Module.parameters()."""

    def __setattr__(self, name, val):
        if name != "_order":
            cur = self.__dict__.get(name)
            if isinstance(cur, Parameter) and val is not cur:
                raise AttributeError("parameter is read-only, use Parameter.assign()")
            if isinstance(val, Parameter) and val._name is None:
                val._name = "%s.%s" % (self.__class__.__name__, name)
            elif isinstance(val, (list, tuple)) and len(val) and all(isinstance(v, ParameterHolder) for v in val):
                for i, item in enumerate(val):
                    for p in item.parameters():
                        p._name = "%s[%d].%s" % (self.__class__.__name__, i, p._name)
            if isinstance(val, (Parameter, ParameterHolder)) or (
                isinstance(val, (list, tuple)) and len(val) and all(isinstance(v, ParameterHolder) for v in val)
            ):
                order = self.__dict__.setdefault("_order", [])
                if name not in order:
                    order.append(name)
                _STRUCTURE_EPOCH[0] += 1            # some holder's registry changed: every cached parameter list is stale
        object.__setattr__(self, name, val)

    def __delattr__(self, name):
        order = self.__dict__.get("_order")
        if order is not None and name in order:
            order.remove(name)
            _STRUCTURE_EPOCH[0] += 1
        object.__delattr__(self, name)

    def _registry_shape(self):
        """lengths and member identities of the registered holder lists, recursively: what an in-place edit of such a list (append, del, item
        assignment -- none of which passes through __setattr__) changes.  Cheap next to parameters(): no Parameter is visited."""
        sig = []
        for name in self.__dict__.get("_order", ()):
            val = self.__dict__.get(name)
            if isinstance(val, (list, tuple)):
                sig.append(tuple(id(v) for v in val))
                sig.extend(v._registry_shape() for v in val if isinstance(v, ParameterHolder))
            elif isinstance(val, ParameterHolder):
                sig.append(val._registry_shape())
        return tuple(sig)

    def _parameter_list(self):
        """`list(self.parameters())`, kept until a registry changes anywhere: the training loop asks for it several times per evaluation
        (zero_grad, log_prior, the optimiser), and the recursive walk was 15 % of the host's share of a configs[1] step"""
        c = self.__dict__.get("_plist")
        shape = self._registry_shape()
        if c is None or c[0] != _STRUCTURE_EPOCH[0] or c[2] != shape:
            c = (_STRUCTURE_EPOCH[0], list(self.parameters()), shape)
            self.__dict__["_plist"] = c
        return c[1]

    def parameters(self):
        """All Parameters in torch.nn.Module.parameters() order: the holder's own Parameters first (registration
        order), then those of its sub-holders (registration order, recursively)."""
        seen = set()
        order = self.__dict__.get("_order", [])
        for name in order:
            val = self.__dict__[name]
            if isinstance(val, Parameter) and id(val) not in seen:
                seen.add(id(val))
                yield val
        for name in order:
            val = self.__dict__[name]
            if isinstance(val, Parameter):
                continue
            items = val if isinstance(val, (list, tuple)) else [val]
            for item in items:
                for p in item.parameters():
                    if id(p) not in seen:
                        seen.add(id(p))
                        yield p

    def zero_grad(self, set_to_none=True):
        for p in self._parameter_list():
            p.grad = None if set_to_none else np.zeros_like(p.data)
