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

    def get_data(self, transformed=False):
        if transformed:
            return self.X, self.Y_transformer.forward(self.Y, self.X)
        return self.X, self.Y

    def get_train_data(self, transformed=False):
        """reference data.py:602-619"""
        if transformed:
            return self.X[self.mask, :], self.Y_transformer.forward(self.Y[self.mask], self.X[self.mask, :])
        return self.X[self.mask, :], self.Y[self.mask]

    def get_test_data(self, transformed=False):
        X, Y = self.X[~self.mask, :], self.Y[~self.mask]
        if transformed:
            return X, self.Y_transformer.forward(Y, X)
        return X, Y

    def set_prediction_data(self, X):
        self.X_pred, _ = self._format_X(X)

    def get_prediction_data(self):
        return self.X_pred

    def _format_X(self, X):
        X = np.asarray(X, dtype=np.float64)
        if X.ndim == 1:
            X = X.reshape(-1, 1)
        if X.ndim != 2 or X.shape[1] != self.get_input_dims():
            raise ValueError("X must have %d input dimensions" % self.get_input_dims())
        return X, None

    def _get_psd_peaks(self, w, psd):
        """peaks of a spectrum as (amplitude, position, variance), biggest first -- reference data.py:946-961: scipy's find_peaks,
        half-maximum widths turned into Gaussian variances (FWHM^2 / 8 ln 2), amplitude = sqrt(peak height)"""
        from scipy import signal
        peaks, _ = signal.find_peaks(psd)
        if len(peaks) == 0:
            return np.array([]), np.array([]), np.array([])
        peaks = peaks[np.argsort(psd[peaks])[::-1]]
        peaks = peaks[0.0 < psd[peaks]]
        widths, _, _, _ = signal.peak_widths(psd, peaks, rel_height=0.5)
        widths = widths * (w[1] - w[0])
        return np.sqrt(psd[peaks]), w[peaks], widths ** 2 / (8.0 * np.log(2.0))

    def get_ls_estimation(self, Q=1, n=10000):
        """Q biggest peaks of the Lomb-Scargle periodogram per input dimension: (amplitudes, means, variances), each (Q, input_dims)
        -- reference data.py:963-1002, including its re-use of `n` for the number of peaks found (which shortens the frequency grid
        of the following input dimensions)"""
        from scipy import signal
        input_dims = self.get_input_dims()
        A, B, C = np.zeros((Q, input_dims)), np.zeros((Q, input_dims)), np.zeros((Q, input_dims))
        nyquist = self.get_nyquist_estimation()
        x, y = self.get_train_data(transformed=True)
        for i in range(input_dims):
            w = np.linspace(0.0, nyquist[i], n)[1:]
            psd = signal.lombscargle(x[:, i] * 2.0 * np.pi, y, w)
            psd /= x.shape[0] / 4.0
            amplitudes, positions, variances = self._get_psd_peaks(w, psd)
            if len(positions) == 0:
                continue
            if Q < len(amplitudes):
                amplitudes, positions, variances = amplitudes[:Q], positions[:Q], variances[:Q]
            n = len(amplitudes)
            A[:n, i] = amplitudes
            B[:n, i] = positions
            C[:n, i] = variances
        return A, B, C

    def get_sm_estimation(self, Q=1, method="LS", optimizer="Adam", iters=200, params={}):
        """fit a single-output spectral mixture to this channel on the device and return its (magnitude, mean, variance) --
        reference data.py:1053-1087"""
        from .wrappers import SM
        input_dims = self.get_input_dims()
        sm = SM(self, Q)
        sm.init_parameters(method)
        sm.train(method=optimizer, iters=iters, **params)
        A = sm.gpr.kernel[0].magnitude.numpy().reshape(-1, 1).repeat(input_dims, axis=1)
        return A, sm.gpr.kernel[0].mean.numpy(), sm.gpr.kernel[0].variance.numpy()

    def get_bnse_estimation(self, Q=1, n=1000, iters=200):
        """Q biggest peaks of the BNSE spectrum per input dimension -- reference data.py:1004-1051 (the GP fit runs on the device)"""
        from .init import BNSE
        input_dims = self.get_input_dims()
        A, B, C = np.zeros((Q, input_dims)), np.zeros((Q, input_dims)), np.zeros((Q, input_dims))
        nyquist = self.get_nyquist_estimation()
        x, y = self.get_train_data(transformed=True)
        y_err = None
        if self.Y_err is not None:
            y_err_lower = self.Y_transformer.forward(y - self.Y_err[self.mask], x)
            y_err_upper = self.Y_transformer.forward(y + self.Y_err[self.mask], x)
            y_err = (y_err_upper - y_err_lower) / 2.0
        for i in range(input_dims):
            w, psd, _ = BNSE(x[:, i], y, y_err=y_err, max_freq=nyquist[i], n=n, iters=iters)
            psd /= (np.max(x[:, i]) - np.min(x[:, i])) ** 2
            psd *= np.pi
            amplitudes, positions, variances = self._get_psd_peaks(w, psd)
            if len(positions) == 0:
                continue
            if Q < len(amplitudes):
                amplitudes, positions, variances = amplitudes[:Q], positions[:Q], variances[:Q]
            num = len(amplitudes)
            A[:num, i] = amplitudes
            B[:num, i] = positions
            C[:num, i] = variances
        return A, B, C

    def get_nyquist_estimation(self):
        """0.5 / (minimum distance between points) per input dimension -- reference data.py:924-944"""
        input_dims = self.get_input_dims()
        nyquist = np.empty((input_dims,))
        for i in range(input_dims):
            x = np.sort(self.X[self.mask, i])
            dist = np.abs(x[1:] - x[:-1])
            if len(dist) == 0:
                nyquist[i] = 0.0
            else:
                nyquist[i] = 0.5 / np.min(dist[np.nonzero(dist)])
        return nyquist


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
        if isinstance(arg, Data):
            self.channels.append(arg)
        elif isinstance(arg, DataSet):
            self.channels.extend(arg.channels)
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
        if isinstance(key, str):
            return self.channels[self.get_names().index(key)]
        return self.channels[key]

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
        for channel in self.channels:
            channel.transform(transformer)

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
        """reference dataset.py:199-221"""
        if isinstance(X, dict):
            x_dict = X
            X = self.get_prediction_data()
            for name, channel_x in x_dict.items():
                X[self.get_index(name)] = channel_x
        elif isinstance(X, np.ndarray) or hasattr(X, "detach"):
            if hasattr(X, "detach"):
                X = X.detach().cpu().numpy()
            if X.ndim == 3 and X.shape[0] == self.get_output_dims():
                X = [X[i, :, :] for i in range(self.get_output_dims())]
            else:
                X = [X] * self.get_output_dims()
        elif not isinstance(X, list):
            raise ValueError("X must be a list, dict, or numpy.ndarray")
        elif not any(isinstance(x, (list, np.ndarray)) for x in X):
            X = [X] * self.get_output_dims()
        if len(X) != self.get_output_dims():
            raise ValueError("X must be of shape (data_points,), (data_points,input_dims), or "
                             "[(data_points,)] * input_dims for each channel")
        X = list(X)
        for j, channel in enumerate(self.channels):
            X[j], _ = channel._format_X(X[j])
        return X
