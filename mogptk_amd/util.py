"""Error metrics behind `Model.error` / `train(error=...)` (the five names of reference mogptk/util.py:6-44, same values).

One residual helper parameterised by the metric instead of five bodies: every metric is a mean over a transformed residual, the two
percentage metrics over the points whose target exceeds 1e-6 only (the reference's rule -- strictly positive targets, so a zero or a
negative target never divides)."""
import numpy as np

_POSITIVE_FLOOR = 1e-6

# name -> (restrict to positive targets, residual transform (t, p) -> per-point value, scale, take the root of the mean)
_METRICS = {
    "mae": (False, lambda t, p: np.abs(t - p), 1.0, False),
    "mape": (True, lambda t, p: np.abs((t - p) / t), 100.0, False),
    "smape": (True, lambda t, p: np.abs((t - p) / (t + p)), 200.0, False),
    "mse": (False, lambda t, p: np.square(t - p), 1.0, False),
    "rmse": (False, lambda t, p: np.square(t - p), 1.0, True),
}


def _metric(name, y_true, y_pred):
    positive_only, per_point, scale, root = _METRICS[name]
    t, p = np.asarray(y_true, dtype=float), np.asarray(y_pred, dtype=float)
    if positive_only:
        keep = t > _POSITIVE_FLOOR
        t, p = t[keep], p[keep]
    m = np.mean(per_point(t, p))
    return (np.sqrt(m) if root else m) * scale


def mean_absolute_error(y_true, y_pred):
    return _metric("mae", y_true, y_pred)


def mean_absolute_percentage_error(y_true, y_pred):
    return _metric("mape", y_true, y_pred)


def symmetric_mean_absolute_percentage_error(y_true, y_pred):
    return _metric("smape", y_true, y_pred)


def mean_squared_error(y_true, y_pred):
    return _metric("mse", y_true, y_pred)


def root_mean_squared_error(y_true, y_pred):
    return _metric("rmse", y_true, y_pred)
