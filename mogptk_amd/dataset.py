"""
Minimal Data / DataSet containers.  The reference's data layer (mogptk/data.py, dataset.py, transformer.py:
pandas loaders, masks, detrending, plotting) is host-side O(N) work outside the hot path (SURVEY.md section 2 rows
16-18); the HIP path consumes its output `(X, y)` unchanged.  These classes provide only the hooks
`mogptk.Model` calls on its dataset (mogptk/model.py:200-231, 585-664; models/mosm.py:59-60) so that
`MOSM(dataset, Q).train(); .predict()` runs end to end without the reference installed.
"""
import numpy as np


from .transformer import Transformer, TransformBase


class Data:
    """One channel: X (n,) or (n, input_dims), Y (n,).  Reference mogptk/data.py:197."""

    def __init__(self, X, Y, Y_err=None, name=None):
        X = np.asarray(X, dtype=np.float64)
        if X.ndim == 1:
            X = X.reshape(-1, 1)
        Y = np.asarray(Y, dtype=np.float64).reshape(-1)
        if X.ndim != 2 or X.shape[0] != Y.shape[0]:
            raise ValueError("X must be (data_points,) or (data_points,input_dims) and match Y")
        self.X = X
        self.Y = Y
        self.Y_err = None if Y_err is None else np.asarray(Y_err, dtype=np.float64).reshape(-1)
        self.name = name
        self.mask = np.ones(Y.shape[0], dtype=bool)
        self.Y_transformer = Transformer()
        self.X_pred = X

    def transform(self, transformer):
        """fit a Y transformer (class or instance) on the data as transformed so far and append it -- reference data.py:457-471"""
        self.Y_transformer.append(transformer, self.Y, self.X)

    def get_name(self):
        return self.name

    def get_input_dims(self):
        return self.X.shape[1]

    def has_test_data(self):
        return bool(np.any(~self.mask))

    def remove_randomly(self, n=None, pct=None, seed=None):
        """reference data.py:683-705 (uniform removal of training points -> test points): `n` points, or a fraction `pct`; nothing when
        neither is given.  `seed` (not in the reference, which draws from torch's global generator) makes the draw reproducible."""
        if n is None:
            n = 0 if pct is None else int(pct * len(self.Y))
        elif isinstance(n, bool) or not isinstance(n, (int, np.integer)):
            raise ValueError("n must be an integer")
        idx = np.random.default_rng(seed).permutation(len(self.Y))[:n]
        self.mask[idx] = False

    def remove_range(self, start=None, end=None, dim=None):
        """reference data.py:731-775: the observations with start <= x <= end (all input dimensions, or `dim` only) become test points"""
        D = self.get_input_dims()
        dims = range(D) if dim is None else [dim]
        lo = np.full(D, -np.inf) if start is None else np.broadcast_to(np.asarray(start, dtype=np.float64), (D,))
        hi = np.full(D, np.inf) if end is None else np.broadcast_to(np.asarray(end, dtype=np.float64), (D,))
        m = np.ones(len(self.Y), dtype=bool)
        for i in dims:
            m &= (self.X[:, i] >= lo[i]) & (self.X[:, i] <= hi[i])
        self.mask[m] = False

    def _observations(self, rows, transformed):
        """(X, Y) of the selected observations (None: all of them), Y through the fitted transformers on request -- reference data.py:585-640"""
        X, Y = (self.X, self.Y) if rows is None else (self.X[rows, :], self.Y[rows])
        return X, (self.Y_transformer.forward(Y, X) if transformed else Y)

    def get_data(self, transformed=False):
        return self._observations(None, transformed)

    def get_train_data(self, transformed=False):
        return self._observations(self.mask, transformed)

    def get_test_data(self, transformed=False):
        return self._observations(~self.mask, transformed)

    def set_prediction_data(self, X):
        self.X_pred, _ = self._format_X(X)

    def get_prediction_data(self):
        return self.X_pred

    def _format_X(self, X):
        """prediction inputs of this channel as an (n, input_dims) float64 array (a flat vector is one input dimension)"""
        X = np.atleast_1d(np.asarray(X, dtype=np.float64))
        X = X[:, None] if X.ndim == 1 else X
        if X.ndim != 2 or X.shape[1] != self.get_input_dims():
            raise ValueError("X must have %d input dimensions" % self.get_input_dims())
        return X, None

    # ---- spectrum-based estimates for init_parameters (reference data.py:924-1087: the numbers are boundary, golden-pinned by init_ls.npz / bnse.npz) ----
    @staticmethod
    def _get_psd_peaks(w, psd):
        """local maxima of a sampled spectrum, highest first -> (amplitude, position, variance): amplitude = sqrt(height), variance from the
        width at half height read as a Gaussian's FWHM (sigma^2 = FWHM^2 / (8 ln 2))"""
        from scipy.signal import find_peaks, peak_widths
        at = find_peaks(psd)[0]
        at = at[psd[at] > 0.0]
        if at.size == 0:
            return np.empty(0), np.empty(0), np.empty(0)
        at = at[np.argsort(-psd[at], kind="stable")]
        fwhm = peak_widths(psd, at, rel_height=0.5)[0] * (w[1] - w[0])
        return np.sqrt(psd[at]), w[at], fwhm * fwhm / (8.0 * np.log(2.0))

    def _peak_table(self, Q, spectrum_of):
        """(amplitudes, means, variances), each (Q, input_dims): column i holds the (at most) Q highest peaks of spectrum_of(i) -> (w, psd),
        zeros where a dimension has fewer.  spectrum_of also receives how many peaks the previous dimensions kept (see get_ls_estimation)."""
        table = np.zeros((3, Q, self.get_input_dims()))
        kept = None
        for i in range(table.shape[2]):
            found = self._get_psd_peaks(*spectrum_of(i, kept))
            if found[1].size:
                kept = min(Q, found[1].size)
                table[:, :kept, i] = [f[:kept] for f in found]
        return table[0], table[1], table[2]

    def get_ls_estimation(self, Q=1, n=10000):
        """peaks of the Lomb-Scargle periodogram on n frequencies up to the Nyquist estimate, per input dimension.  One oddity of the
        reference is part of the numbers (data.py:963-1002 reuses the name `n` for the count of peaks it kept): after a dimension with peaks,
        the NEXT dimension's frequency grid has only that many points."""
        from scipy.signal import lombscargle
        x, y = self._observations(self.mask, True)
        top = self.get_nyquist_estimation()

        def periodogram(i, kept):
            w = np.linspace(0.0, top[i], n if kept is None else kept)[1:]
            return w, lombscargle(2.0 * np.pi * x[:, i], y, w) * (4.0 / len(x))
        return self._peak_table(Q, periodogram)

    def get_bnse_estimation(self, Q=1, n=1000, iters=200):
        """peaks of the Bayesian nonparametric spectral estimate (init.BNSE: a GP fit on the device) per input dimension, scaled by
        pi / range^2 -- reference data.py:1004-1051; observation errors go through the Y transformer as half the transformed interval"""
        from .init import BNSE
        x, y = self._observations(self.mask, True)
        top = self.get_nyquist_estimation()
        half_width = None
        if self.Y_err is not None:
            e = self.Y_err[self.mask]
            half_width = 0.5 * (self.Y_transformer.forward(y + e, x) - self.Y_transformer.forward(y - e, x))

        def posterior_spectrum(i, _):
            w, psd, _var = BNSE(x[:, i], y, y_err=half_width, max_freq=top[i], n=n, iters=iters)
            return w, psd * (np.pi / np.ptp(x[:, i]) ** 2)
        return self._peak_table(Q, posterior_spectrum)

    def get_sm_estimation(self, Q=1, method="LS", optimizer="Adam", iters=200, params={}):
        """fit a single-output spectral mixture to this channel on the device; its (magnitude repeated per input dimension, mean, variance)
        -- reference data.py:1053-1087"""
        from .wrappers import SM
        fit = SM(self, Q)
        fit.init_parameters(method)
        fit.train(method=optimizer, iters=iters, **params)
        k = fit.gpr.kernel[0]
        return np.tile(k.magnitude.numpy().reshape(-1, 1), (1, self.get_input_dims())), k.mean.numpy(), k.variance.numpy()

    def get_nyquist_estimation(self):
        """half the inverse of the smallest non-zero gap between training inputs, per input dimension (0 with fewer than two points)
        -- reference data.py:924-944"""
        seen = np.sort(self.X[self.mask], axis=0)
        if len(seen) < 2:
            return np.zeros(self.get_input_dims())
        gaps = np.diff(seen, axis=0)
        return np.array([0.5 / g[g != 0.0].min() for g in gaps.T])


class DataSet:
    """Ordered list of channels.  Reference mogptk/dataset.py:130.
    DataSet(data0, data1, ...), DataSet([data0, ...]) or DataSet(x, [y0, y1, ...])."""

    def __init__(self, *args, names=None):
        self.channels = []
        if len(args) == 2 and not isinstance(args[0], (Data, DataSet)) and isinstance(args[1], (list, tuple)) \
                and not any(isinstance(a, Data) for a in args[1]):
            x, ys = args
            for j, y in enumerate(ys):
                self.channels.append(Data(x, y, name=None if names is None else names[j]))
        else:
            for arg in args:
                self.append(arg)

    def append(self, arg):
        if isinstance(arg, (Data, DataSet)):
            self.channels += [arg] if isinstance(arg, Data) else arg.channels
        elif isinstance(arg, (list, tuple)) and all(isinstance(a, Data) for a in arg):
            self.channels.extend(arg)
        elif isinstance(arg, dict) and all(isinstance(a, Data) for a in arg.values()):
            for k, v in arg.items():
                v.name = k
                self.channels.append(v)
        else:
            raise ValueError("must append Data, DataSet, or a list/dict of Data")

    def __iter__(self):
        return iter(self.channels)

    def __len__(self):
        return len(self.channels)

    def __getitem__(self, key):
        return self.channels[self.get_index(key) if isinstance(key, str) else key]

    def get_names(self):
        return [c.get_name() for c in self.channels]

    def get_index(self, name):
        return self.get_names().index(name) if isinstance(name, str) else name

    def get_output_dims(self):
        return len(self.channels)

    def get_input_dims(self):
        return [c.get_input_dims() for c in self.channels]

    def has_test_data(self):
        return [c.has_test_data() for c in self.channels]

    def get_data(self, transformed=False):
        return [c.get_data(transformed)[0] for c in self.channels], [c.get_data(transformed)[1] for c in self.channels]

    def get_train_data(self, transformed=False):
        """reference dataset.py:455-485"""
        return ([c.get_train_data(transformed)[0] for c in self.channels],
                [c.get_train_data(transformed)[1] for c in self.channels])

    def get_test_data(self, transformed=False):
        return ([c.get_test_data(transformed)[0] for c in self.channels],
                [c.get_test_data(transformed)[1] for c in self.channels])

    def get_prediction_data(self):
        return [c.get_prediction_data() for c in self.channels]

    def set_prediction_data(self, X):
        X = self._format_X(X)
        for j, c in enumerate(self.channels):
            c.X_pred = X[j]

    def transform(self, transformer):
        """the same Y transformer (fitted per channel) on every channel -- reference dataset.py:transform"""
        for c in self.channels:
            c.transform(transformer)

    def get_ls_estimation(self, Q=1, n=10000):
        """per channel -- reference dataset.py:579-603"""
        out = [channel.get_ls_estimation(Q, n) for channel in self.channels]
        return [o[0] for o in out], [o[1] for o in out], [o[2] for o in out]

    def get_sm_estimation(self, Q=1, method="BNSE", optimizer="Adam", iters=200, params={}):
        """per channel -- reference dataset.py:632-660 (its default initialisation of the fitted mixtures is BNSE)"""
        out = [channel.get_sm_estimation(Q, method, optimizer, iters, params) for channel in self.channels]
        return [o[0] for o in out], [o[1] for o in out], [o[2] for o in out]

    def get_bnse_estimation(self, Q=1, n=1000, iters=200):
        """per channel -- reference dataset.py:605-632"""
        out = [channel.get_bnse_estimation(Q, n, iters) for channel in self.channels]
        return [o[0] for o in out], [o[1] for o in out], [o[2] for o in out]

    def get_nyquist_estimation(self):
        return [c.get_nyquist_estimation() for c in self.channels]

    def _format_X(self, X):
        """prediction inputs per channel from what `predict(X)` accepts (reference dataset.py:199-221): a dict {channel name: inputs} on top of
        the current prediction inputs, one array for all channels, a (channels, n, input_dims) array, or a list with one entry per channel"""
        C = self.get_output_dims()
        if hasattr(X, "detach"):
            X = X.detach().cpu().numpy()
        if isinstance(X, dict):
            per_channel = self.get_prediction_data()
            for name, inputs in X.items():
                per_channel[self.get_index(name)] = inputs
        elif isinstance(X, np.ndarray):
            per_channel = list(X) if X.ndim == 3 and len(X) == C else [X] * C
        elif isinstance(X, list):
            nested = any(isinstance(entry, (list, np.ndarray)) for entry in X)
            per_channel = list(X) if nested else [X] * C
        else:
            raise ValueError("X must be a list, dict, or numpy.ndarray")
        if len(per_channel) != C:
            raise ValueError("X must be of shape (data_points,), (data_points,input_dims), or "
                             "[(data_points,)] * input_dims for each channel")
        return [channel._format_X(inputs)[0] for channel, inputs in zip(self.channels, per_channel)]
