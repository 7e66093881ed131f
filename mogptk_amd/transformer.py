"""
Output (Y) transformations of a channel -- the behaviour of mogptk/transformer.py:4-153 behind the same names.

A channel keeps a chain (`Transformer`): `Data.transform(t)` fits `t` on the data AS ALREADY TRANSFORMED by the chain so far and
appends it; `forward` runs the chain, `backward` unwinds it.  The model trains on the forward-transformed targets and
`Model.predict` maps mean and confidence bounds back.  Three of the five reference transformers are the same map
y -> (y - offset) / scale with differently fitted constants; they share one implementation here.
"""
import copy

import numpy as np


class TransformBase:
    """interface: set_data(y, x) fits, forward / backward map the targets (x: inputs of the same points, (n, input_dims))"""

    def set_data(self, y, x=None):
        pass

    def forward(self, y, x=None):
        raise NotImplementedError

    def backward(self, y, x=None):
        raise NotImplementedError


class Transformer:
    """the chain of a channel (reference transformer.py:4-31)"""

    def __init__(self, transformers=None):
        chain = [] if transformers is None else (transformers if isinstance(transformers, list) else [transformers])
        if any(not isinstance(t, TransformBase) for t in chain):
            raise ValueError("transformer must derive from TransformBase")
        self.transformers = chain

    def append(self, t, y, x=None):
        t = t() if isinstance(t, type) else copy.deepcopy(t)        # a class is instantiated, an instance is copied (not shared)
        t.set_data(self.forward(y, x), x)
        self.transformers.append(t)

    def forward(self, y, x=None):
        for t in self.transformers:
            y = t.forward(y, x)
        return y

    def backward(self, y, x=None):
        for t in reversed(self.transformers):
            y = t.backward(y, x)
        return y


class _Affine(TransformBase):
    """y -> (y - offset) / scale"""
    offset, scale = 0.0, 1.0

    def forward(self, y, x=None):
        return (y - self.offset) / self.scale

    def backward(self, y, x=None):
        return self.offset + self.scale * y


class TransformLinear(_Affine):
    """fixed bias and slope: y -> (y - bias) / slope"""

    def __init__(self, bias=0.0, slope=1.0):
        self.bias, self.slope = bias, slope
        self.offset, self.scale = bias, slope

    def __repr__(self):
        return "TransformLinear(bias=%g, slope=%g)" % (self.bias, self.slope)


class TransformStandard(_Affine):
    """zero mean, unit (population) standard deviation"""

    def set_data(self, y, x=None):
        self.mean, self.std = np.mean(y), np.std(y)
        self.offset, self.scale = self.mean, self.std

    def __repr__(self):
        return "TransformStandard(mean=%g, std=%g)" % (self.mean, self.std)


class TransformNormalize(_Affine):
    """onto [-1, 1]"""

    def set_data(self, y, x=None):
        self.ymin, self.ymax = np.amin(y), np.amax(y)
        self.scale = 0.5 * (self.ymax - self.ymin)
        self.offset = self.ymin + self.scale

    def forward(self, y, x=None):
        return -1.0 + 2.0 * (y - self.ymin) / (self.ymax - self.ymin)      # the reference's operation order (bit-for-bit fixtures)

    def backward(self, y, x=None):
        return (y + 1.0) / 2.0 * (self.ymax - self.ymin) + self.ymin

    def __repr__(self):
        return "TransformNormalize(min=%g, max=%g)" % (self.ymin, self.ymax)


class TransformLog(TransformBase):
    """log of the data shifted to >= 1, minus its mean"""

    def set_data(self, y, x=None):
        self.shift = 1 - np.min(y)
        self.mean = np.mean(np.log(y + self.shift))

    def forward(self, y, x=None):
        return np.log(y + self.shift) - self.mean

    def backward(self, y, x=None):
        return np.exp(y + self.mean) - self.shift

    def __repr__(self):
        return "TransformLog(shift=%g, mean=%g)" % (self.shift, self.mean)


class TransformDetrend(TransformBase):
    """removes a least-squares polynomial trend of the given degree along one input dimension (numpy.polyfit)"""

    def __init__(self, degree=1, input_dim=0):
        self.degree, self.dim = degree, input_dim

    def set_data(self, y, x=None):
        self.coef = np.polyfit(x[:, self.dim], y, self.degree)

    def _trend(self, x):
        if x is None:
            raise ValueError("must set X for transformation")
        return np.polyval(self.coef, x[:, self.dim])

    def forward(self, y, x):
        return y - self._trend(x)

    def backward(self, y, x):
        return y + self._trend(x)

    def __repr__(self):
        return "TransformDetrend(degree=%g)" % (self.degree,)
