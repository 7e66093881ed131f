"""
ORACLE -- TEST INFRASTRUCTURE / CPU BASELINE ONLY.

A from-scratch functional restatement, in torch-CPU fp64, of the reference's op SEQUENCE for one
`gpr.Model.loss()` evaluation of an exact MOSM model: softplus-constrained parameters
(gpr/parameter.py:48-49), per-channel-pair Q x n_i x n_j temporaries (gpr/multioutput.py:182-204), dense
scatter into K with index assignment and its transpose (gpr/kernel.py:457-467), `+ scale^2[chan] * eye`
(gpr/model.py:440), relative jitter via repeat().diagflat() (:244), torch.linalg.cholesky (:246),
cholesky_solve (:452) and reverse-mode autograd through all of it (:291).  It is what bench.py times as
`cpu_baseline` (kind "port") on the GPU box's host cores, because the reference itself cannot travel there.
Parity: pinned -- tests/test_oracle_golden.py checks loss and gradients against the reference's own output
(tests/golden/lml_synth2048.npz) and its wall time here matches the reference's.
"""
import numpy as np
import torch


def softplus_c(raw, lower=1e-8):
    return lower + torch.nn.functional.softplus(raw, beta=0.1, threshold=20.0)


def mosm_block(i, j, xi, xj, w, mu, v, th, ph):
    D = xi.shape[1]
    twopi = float(np.power(2.0 * np.pi, D / 2.0))
    tau = xi.unsqueeze(1) - xj
    if i == j:
        var = v[i]
        alpha = w[i] ** 2 * twopi * var.prod(dim=1).sqrt()
        ex = torch.exp(-0.5 * torch.einsum("nmd,qd->qnm", tau ** 2, var))
        co = torch.cos(2.0 * np.pi * torch.einsum("nmd,qd->qnm", tau, mu[i]))
        Kq = alpha[:, None, None] * ex * co
    else:
        inv = 1.0 / (v[i] + v[j])
        dmu = mu[i] - mu[j]
        mag = w[i] * w[j] * torch.exp(-np.pi ** 2 * torch.sum(dmu * inv * dmu, dim=1))
        mean = inv * (v[i] * mu[j] + v[j] * mu[i])
        var = 2.0 * v[i] * inv * v[j]
        dl = th[i] - th[j]
        p = ph[i] - ph[j]
        alpha = mag * twopi * var.prod(dim=1).sqrt()
        td = tau[None, :, :, :] + dl[:, None, None, :]
        ex = torch.exp(-0.5 * torch.einsum("qnmd,qd->qnm", td ** 2, var))
        co = torch.cos(2.0 * np.pi * (torch.einsum("qnmd,qd->qnm", td, mean) + p[:, None, None]))
        Kq = alpha[:, None, None] * ex * co
    return torch.sum(Kq, dim=0)


def mosm_loss_and_grad(X, y, raws, C, jitter=1e-8, eye=None):
    """raws: dict of numpy raw arrays weight, mean, variance, delay, phase, scale (softplus lower 1e-8 on the
    positive ones).  Returns (loss, dict of d loss / d raw)."""
    X = torch.as_tensor(X, dtype=torch.float64)
    y = torch.as_tensor(y, dtype=torch.float64).reshape(-1, 1)
    N = X.shape[0]
    P = {k: torch.tensor(np.asarray(v), dtype=torch.float64, requires_grad=True) for k, v in raws.items()}
    w, mu, v = softplus_c(P["weight"]), softplus_c(P["mean"]), softplus_c(P["variance"])
    th, ph, scale = P["delay"], P["phase"], softplus_c(P["scale"])
    c = X[:, 0].long()
    m = [c == i for i in range(C)]
    x = [X[m[i], 1:] for i in range(C)]
    r1 = [torch.nonzero(m[i], as_tuple=False) for i in range(C)]
    r2 = [r1[i].reshape(1, -1) for i in range(C)]
    K = torch.empty(N, N, dtype=torch.float64)
    for i in range(C):
        for j in range(i + 1):
            k = mosm_block(i, j, x[i], x[j], w, mu, v, th, ph)
            if i == j:
                K[r1[i], r2[i]] = k
            else:
                K[r1[i], r2[j]] = k
                K[r1[j], r2[i]] = k.T
    if eye is None:
        eye = torch.eye(N, dtype=torch.float64)
    K = K + torch.index_select(scale.square(), 0, c) * eye
    K = K + (max(jitter, 1e-15) * K.diagonal().mean()).repeat(N).diagflat()
    L = torch.linalg.cholesky(K)
    p = -0.5 * N * np.log(2.0 * np.pi)
    p = p - L.diagonal().log().sum()
    p = p - 0.5 * y.T.mm(torch.cholesky_solve(y, L)).squeeze()
    loss = -p
    loss.backward()
    return float(loss), {k: t.grad.numpy() for k, t in P.items()}
