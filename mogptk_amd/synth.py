"""
Deterministic synthetic multi-channel time series + hyper-parameters: the inputs of
BASELINE.json's configs, shared by bench.py, the parity tests and the golden-vector generator so that only
outputs need to be stored for the full-size configs.  numpy PCG64 streams are stable across numpy versions.
"""
import numpy as np

DATA_SEED = 20250620
HYPER_SEED = 20250621


def make_data(N, C, D=1, seed=DATA_SEED):
    """-> X (N,1+D) channel-contiguous kernel format, y (N,)"""
    n = N // C
    rng = np.random.default_rng(seed)
    a = (1.0, 0.5, 0.25)
    f = (0.05, 0.13, 0.31)
    Xs, ys = [], []
    for c in range(C):
        x = np.sort(rng.uniform(0.0, 100.0, n))
        y = sum(a[k] * np.sin(2.0 * np.pi * f[k] * (1.0 + 0.1 * c) * x + 0.7 * c) for k in range(3))
        y = y + 0.1 * rng.standard_normal(n)
        cols = [np.full(n, float(c)), x]
        for d in range(1, D):
            cols.append(rng.uniform(0.0, 100.0, n))
        Xs.append(np.stack(cols, axis=1))
        ys.append(y)
    return np.concatenate(Xs, axis=0), np.concatenate(ys)


def mosm_hypers(C, Q, D=1, seed=HYPER_SEED):
    """constrained values, drawn in the order weight, mean, variance, delay, phase, scale"""
    rng = np.random.default_rng(seed)
    return dict(
        weight=rng.uniform(0.5, 1.5, (C, Q)),
        mean=rng.uniform(0.02, 0.4, (C, Q, D)),
        variance=rng.uniform(0.005, 0.05, (C, Q, D)),
        delay=rng.normal(0.0, 0.3, (C, Q, D)),
        phase=rng.normal(0.0, 0.3, (C, Q)),
        scale=rng.uniform(0.1, 0.4, C),
    )


def sm_hypers(C, Q, D=1, seed=HYPER_SEED):
    rng = np.random.default_rng(seed)
    return dict(
        magnitude=rng.uniform(0.5, 1.5, (C, Q)),
        mean=rng.uniform(0.02, 0.4, (C, Q, D)),
        variance=rng.uniform(0.005, 0.05, (C, Q, D)),
        scale=rng.uniform(0.1, 0.4, C),
    )


def csm_hypers(C, Q, Rq=1, D=1, seed=HYPER_SEED):
    rng = np.random.default_rng(seed)
    return dict(
        amplitude=rng.uniform(0.5, 1.5, (Q, C, Rq)),
        mean=rng.uniform(0.02, 0.4, (Q, D)),
        variance=rng.uniform(0.005, 0.05, (Q, D)),
        shift=rng.normal(0.0, 0.3, (Q, C, Rq)),
        scale=rng.uniform(0.1, 0.4, C),
    )


def test_inputs(S, C, hi=110.0):
    """cfg4 test inputs: linspace(0, hi, S//C) per channel, kernel format"""
    s = S // C
    return np.concatenate([np.stack([np.full(s, float(c)), np.linspace(0.0, hi, s)], axis=1) for c in range(C)], axis=0)
