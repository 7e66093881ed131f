"""
ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the product.

CPU restatement (numpy, fp64) of the reference's exact multi-output GP hot path
(GAMES-UChile/mogptk v0.5.1).  Only tests/, __graft_entry__.smoke() and the
`cpu_baseline` leg of bench.py may import this module, and only as the checker.
The product path (mogptk_amd/) never imports it and fails loudly when the HIP
extension is missing.

Parity status: PINNED.  Every function here is checked (tests/test_oracle_golden.py)
against golden vectors produced by importing the reference itself in the build
container (tests/golden/gen_golden.py, reference imported from /root/reference with
an IPython stub), and against the only result the reference's own unit test pins at
this boundary (tests/unit/test_kernels.py:43-57: K_diag == diag(K), bitwise).

Each function cites the reference file:line it follows (paths relative to the
reference root).  Gradients are obtained by complex-step differentiation of the
restated forward pass (no hand-derived chain rule here, so the oracle is independent
of the product's moment/chain-rule formulation); the reference uses torch autograd
(mogptk/gpr/model.py:291).
"""
import numpy as np

TWO_PI = 2.0 * np.pi


# --------------------------------------------------------------------------------------
# constrained parameters -- mogptk/gpr/parameter.py:30-96, 220-230
# --------------------------------------------------------------------------------------
def softplus_forward(raw, lower, beta=0.1, threshold=20.0):
    """parameter.py:48-49 (torch softplus: linear above beta*x > threshold)."""
    raw = np.asarray(raw)
    z = beta * raw
    big = np.real(z) > threshold
    zs = np.where(big, 0.0, z)
    sp = np.where(big, raw, np.log1p(np.exp(zs)) / beta)
    return lower + sp


def softplus_inverse(y, lower, beta=0.1):
    """parameter.py:59 -- NOTE the reference misplaces `lower` (quirk Q1): reproduced."""
    y = np.asarray(y, dtype=np.float64)
    return (y - lower) + np.log(-np.expm1(-beta * y - lower)) / beta


def sigmoid_forward(raw, lower, upper):
    """parameter.py:77-78."""
    return lower + (upper - lower) / (1.0 + np.exp(-np.asarray(raw)))


def sigmoid_inverse(y, lower, upper):
    """parameter.py:80-96 (degenerate lower==upper -> eps)."""
    y = (np.asarray(y, dtype=np.float64) - lower) / (upper - lower)
    y = np.where(np.isclose(lower * np.ones_like(y), upper * np.ones_like(y)), np.finfo(float).eps, y)
    return np.log(y) - np.log(1.0 - y)


def constrain(raw, lower=None, upper=None):
    """parameter.py:187-201 + to_transform :220-230 (upper-only Softplus(beta=-0.1) not used on the path)."""
    if lower is not None and upper is not None:
        return sigmoid_forward(raw, lower, upper)
    if lower is not None:
        return softplus_forward(raw, lower)
    return np.asarray(raw)


def unconstrain(value, lower=None, upper=None):
    """parameter.py:302-308: clamp into [lower, upper], then transform.inverse."""
    value = np.array(value, dtype=np.float64)
    if lower is not None:
        value = np.where(value < lower, lower * np.ones_like(value), value)
    if upper is not None:
        value = np.where(upper < value, upper * np.ones_like(value), value)
    if lower is not None and upper is not None:
        return sigmoid_inverse(value, lower, upper)
    if lower is not None:
        return softplus_inverse(value, lower)
    return value


# --------------------------------------------------------------------------------------
# spectral sub-kernels -- one channel pair at a time
# --------------------------------------------------------------------------------------
def _distance(x1, x2):
    """kernel.py:172-177: SIGNED tau = x1[:,None,:] - x2[None,:,:]  (n1,n2,D)."""
    return x1[:, None, :] - x2[None, :, :]


def mosm_ksub(i, j, x1, x2, weight, mean, variance, delay, phase):
    """multioutput.py:178-204.  weight (C,Q); mean/variance/delay (C,Q,D); phase (C,Q)."""
    D = x1.shape[1]
    twopi = np.power(TWO_PI, D / 2.0)
    tau = _distance(x1, x2)
    if i == j:
        var = variance[i]                                             # (Q,D)
        alpha = weight[i] ** 2 * twopi * np.sqrt(np.prod(var, axis=1))  # (Q,)
        ex = np.exp(-0.5 * np.einsum("nmd,qd->qnm", tau ** 2, var))
        co = np.cos(TWO_PI * np.einsum("nmd,qd->qnm", tau, mean[i]))
        Kq = alpha[:, None, None] * ex * co
    else:
        inv_var = 1.0 / (variance[i] + variance[j])
        dmean = mean[i] - mean[j]
        mag = weight[i] * weight[j] * np.exp(-np.pi ** 2 * np.sum(dmean * inv_var * dmean, axis=1))
        m = inv_var * (variance[i] * mean[j] + variance[j] * mean[i])
        var = 2.0 * variance[i] * inv_var * variance[j]
        dl = delay[i] - delay[j]
        ph = phase[i] - phase[j]
        alpha = mag * twopi * np.sqrt(np.prod(var, axis=1))
        td = tau[None, :, :, :] + dl[:, None, None, :]
        ex = np.exp(-0.5 * np.einsum("qnmd,qd->qnm", td ** 2, var))
        co = np.cos(TWO_PI * (np.einsum("qnmd,qd->qnm", td, m) + ph[:, None, None]))
        Kq = alpha[:, None, None] * ex * co
    return np.sum(Kq, axis=0)


def mosm_ksub_diag(i, n, weight, variance):
    """multioutput.py:206-210."""
    D = variance.shape[2]
    twopi = np.power(TWO_PI, D / 2.0)
    alpha = weight[i] ** 2 * twopi * np.sqrt(np.prod(variance[i], axis=1))
    return np.repeat(np.sum(alpha), n)


def sm_k(x1, x2, magnitude, mean, variance):
    """singleoutput.py:594-600: note the einsum SUMS over the input dimension d."""
    tau = _distance(x1, x2)[None]
    ex = -2.0 * np.pi ** 2 * tau ** 2 * variance[:, None, None, :]
    co = TWO_PI * tau * mean[:, None, None, :]
    return np.einsum("q,qnmd,qnmd->nm", magnitude, np.exp(ex), np.cos(co))


def csm_ksub(i, j, x1, x2, amplitude, mean, variance, shift):
    """multioutput.py:428-449.  amplitude/shift (C,Rq); mean/variance (D,)."""
    tau = _distance(x1, x2)
    ex = np.exp(-0.5 * np.tensordot(tau ** 2, variance, axes=1))[:, :, None]
    if i == j:
        amp = amplitude[i].reshape(1, 1, -1)
        co = np.cos(TWO_PI * np.tensordot(tau, mean, axes=1)[:, :, None])
        return np.sum(amp * ex * co, axis=2)
    sh = shift[i] - shift[j]
    amp = np.sqrt(amplitude[i] * amplitude[j]).reshape(1, 1, -1)
    co = np.cos(TWO_PI * (np.tensordot(tau, mean, axes=1)[:, :, None] + sh.reshape(1, 1, -1)))
    return np.sum(amp * ex * co, axis=2)


# --------------------------------------------------------------------------------------
# model "spec": a plain dict describing the kernel + its constrained parameter values
#   {'kind': 'mosm', 'C':, 'Q':, 'D':, 'weight','mean','variance','delay','phase'}
#   {'kind': 'sm',   'C':, 'Q':, 'D':, 'magnitude' (C,Q), 'mean' (C,Q,D), 'variance' (C,Q,D)}   (SM inside IMO)
#   {'kind': 'csm',  'C':, 'Q':, 'Rq':, 'D':, 'amplitude' (Q,C,Rq), 'mean' (Q,D), 'variance' (Q,D), 'shift' (Q,C,Rq)}
# --------------------------------------------------------------------------------------
def ksub(spec, i, j, x1, x2):
    kind = spec["kind"]
    if kind == "mosm":
        return mosm_ksub(i, j, x1, x2, spec["weight"], spec["mean"], spec["variance"], spec["delay"], spec["phase"])
    if kind == "sm":
        # IndependentMultiOutputKernel.Ksub, multioutput.py:26-34
        if i == j:
            return sm_k(x1, x2, spec["magnitude"][i], spec["mean"][i], spec["variance"][i])
        return np.zeros((x1.shape[0], x2.shape[0]), dtype=np.result_type(x1, spec["magnitude"]))
    if kind == "csm":
        # MixtureKernel == AddKernel over Q CrossSpectralKernels, kernel.py:242-243,264-276
        out = 0.0
        for q in range(spec["Q"]):
            out = out + csm_ksub(i, j, x1, x2, spec["amplitude"][q], spec["mean"][q], spec["variance"][q], spec["shift"][q])
        return out
    raise ValueError(kind)


def ksub_diag(spec, i, n):
    kind = spec["kind"]
    if kind == "mosm":
        return mosm_ksub_diag(i, n, spec["weight"], spec["variance"])
    if kind == "sm":
        return np.repeat(np.sum(spec["magnitude"][i]), n)         # singleoutput.py:602-605
    if kind == "csm":
        return np.repeat(np.sum(spec["amplitude"][:, i, :]), n)    # multioutput.py:451-454 summed by kernel.py:245-246
    raise ValueError(kind)


def mo_K(spec, X1, X2=None):
    """MultiOutputKernel.K, kernel.py:446-481: split rows by channel id (col 0), loop channel pairs
    (lower pairs + transpose when X2 is None, all C*C pairs otherwise), scatter into the dense result.
    Works for arbitrary row order."""
    C = spec["C"]
    c1 = np.real(X1[:, 0]).astype(np.int64)
    r1 = [np.nonzero(c1 == i)[0] for i in range(C)]
    x1 = [X1[r1[i], 1:] for i in range(C)]
    dt = np.result_type(X1.dtype, *[np.asarray(v).dtype for k, v in spec.items() if isinstance(v, np.ndarray)])
    if X2 is None:
        res = np.empty((X1.shape[0], X1.shape[0]), dtype=dt)
        for i in range(C):
            for j in range(i + 1):
                k = ksub(spec, i, j, x1[i], x1[j])
                res[np.ix_(r1[i], r1[j])] = k
                if i != j:
                    res[np.ix_(r1[j], r1[i])] = k.T
        return res
    c2 = np.real(X2[:, 0]).astype(np.int64)
    r2 = [np.nonzero(c2 == j)[0] for j in range(C)]
    x2 = [X2[r2[j], 1:] for j in range(C)]
    res = np.empty((X1.shape[0], X2.shape[0]), dtype=dt)
    for i in range(C):
        for j in range(C):
            res[np.ix_(r1[i], r2[j])] = ksub(spec, i, j, x1[i], x2[j])
    return res


def mo_K_diag(spec, X1):
    """MultiOutputKernel.K_diag, kernel.py:483-495."""
    C = spec["C"]
    c1 = np.real(X1[:, 0]).astype(np.int64)
    dt = np.result_type(*[np.asarray(v).dtype for k, v in spec.items() if isinstance(v, np.ndarray)])
    res = np.empty(X1.shape[0], dtype=dt)
    for i in range(C):
        r = np.nonzero(c1 == i)[0]
        res[r] = ksub_diag(spec, i, len(r))
    return res


# --------------------------------------------------------------------------------------
# Exact GP -- mogptk/gpr/model.py:418-483
# --------------------------------------------------------------------------------------
def effective_jitter(jitter, dtype=np.float64):
    """gpr/model.py:106-110."""
    return max(jitter, 1e-15) if dtype == np.float64 else max(jitter, 1e-6)


def exact_Kj(spec, scale, X, jitter, data_variance=None):
    """K + sigma_c(k)^2 I (+data_variance) + jitter*mean(diag)*I -- gpr/model.py:439-442, 242-244.
    scale: scalar or (C,) Gaussian likelihood scale (constrained)."""
    K = mo_K(spec, X)
    scale = np.asarray(scale)
    if scale.ndim == 1 and scale.shape[0] == spec["C"]:
        s2 = (scale ** 2)[np.real(X[:, 0]).astype(np.int64)]    # _index_channel, gpr/model.py:183-186
    else:
        s2 = scale ** 2 * np.ones(X.shape[0])
    idx = np.arange(X.shape[0])
    K[idx, idx] = K[idx, idx] + s2
    if data_variance is not None:
        K[idx, idx] = K[idx, idx] + data_variance
    K[idx, idx] = K[idx, idx] + effective_jitter(jitter) * np.mean(np.diagonal(K))
    return K


def exact_lml_from_Kj(Kj, y):
    """gpr/model.py:443-453.  Returns (lml, L, alpha). Raises LinAlgError like torch.linalg.cholesky."""
    N = Kj.shape[0]
    L = np.linalg.cholesky(Kj)
    from scipy.linalg import solve_triangular
    z = solve_triangular(L, y, lower=True)
    alpha = solve_triangular(L.T, z, lower=False)
    lml = -0.5 * N * np.log(TWO_PI) - np.sum(np.log(np.diagonal(L))) - 0.5 * (y.T @ alpha).item()
    return lml, L, alpha


def exact_lml(spec, scale, X, y, jitter=1e-8, data_variance=None):
    return exact_lml_from_Kj(exact_Kj(spec, scale, X, jitter, data_variance), y.reshape(-1, 1))[0]


def exact_predict_f(spec, scale, X, y, Xs, jitter=1e-8, full=False):
    """gpr/model.py:455-483 (no noise added to var)."""
    from scipy.linalg import solve_triangular
    Kj = exact_Kj(spec, scale, X, jitter)
    Kfs = mo_K(spec, X, Xs)
    _, L, alpha = exact_lml_from_Kj(Kj, y.reshape(-1, 1))
    v = solve_triangular(L, Kfs, lower=True)
    mu = Kfs.T @ alpha
    if full:
        var = mo_K(spec, Xs) - v.T @ v
    else:
        var = (mo_K_diag(spec, Xs) - np.sum(v.T ** 2, axis=1)).reshape(-1, 1)
    return mu, var


def gaussian_predict_ci(scale, C, Xs, mu, var, sigma=2.0):
    """GaussianLikelihood.predict sigma-path, likelihood.py:351-378 (quirk Q4: the multi-output CI ignores
    the GP variance).  scale 1-D (C,) -> multi-output branch; 0-d -> single-output branch."""
    scale = np.asarray(scale)
    if scale.ndim == 1:
        s = scale[Xs[:, 0].astype(np.int64)].reshape(-1, 1)
        return mu, mu - sigma * s, mu + sigma * s
    v = var + scale ** 2
    return mu, mu - sigma * np.sqrt(v), mu + sigma * np.sqrt(v)


# --------------------------------------------------------------------------------------
# raw-parameter view: a "pspec" holds raw values + bounds, in the reference's registration order
#   pspec = {'kind','C','Q','D',('Rq'), 'params': [ (name, raw ndarray, lower, upper), ... ] }
#   names: MOSM weight, mean, variance, delay, phase ; SM magnitude/mean/variance stacked on channel ;
#          CSM amplitude, mean, variance, shift stacked on q ; last entry always 'scale'.
# --------------------------------------------------------------------------------------
def spec_from_raw(pspec, raws=None):
    spec = {k: v for k, v in pspec.items() if k != "params"}
    for n, (name, raw, lo, up) in enumerate(pspec["params"]):
        r = raw if raws is None else raws[n]
        spec[name] = constrain(r, lo, up)
    return spec


def neg_loss_terms(pspec, X, y, jitter, raws=None):
    spec = spec_from_raw(pspec, raws)
    Kj = exact_Kj(spec, spec["scale"], X, jitter)
    return Kj


def exact_lml_grad(pspec, X, y, jitter=1e-8):
    """LML and d(LML)/d(raw) for every parameter.  dLML/dKj = G = 0.5*(alpha alpha^T - Kj^-1)
    (the adjoint torch autograd arrives at through CholeskySolveBackward/LinalgCholeskyExBackward,
    gpr/model.py:291), then dKj/d(raw_p) by complex-step differentiation of the restated forward."""
    y = y.reshape(-1, 1)
    raws = [np.array(p[1], dtype=np.float64) for p in pspec["params"]]
    Kj = neg_loss_terms(pspec, X, y, jitter, raws)
    lml, L, alpha = exact_lml_from_Kj(Kj, y)
    from scipy.linalg import solve_triangular
    Linv = solve_triangular(L, np.eye(L.shape[0]), lower=True)
    G = 0.5 * (alpha @ alpha.T - Linv.T @ Linv)
    h = 1e-30
    Xc = X.astype(np.complex128)
    grads = []
    for n, r in enumerate(raws):
        g = np.zeros_like(r)
        it = np.nditer(r, flags=["multi_index"])
        for _ in it:
            rc = [a.astype(np.complex128) for a in raws]
            rc[n][it.multi_index] += 1j * h
            dK = np.imag(neg_loss_terms(pspec, Xc, y, jitter, rc)) / h
            g[it.multi_index] = np.sum(G * dK)
        grads.append(g)
    return lml, grads


# --------------------------------------------------------------------------------------
# Adam -- torch.optim.Adam defaults as used by mogptk/model.py:557 (no weight decay / amsgrad)
# --------------------------------------------------------------------------------------
def adam_step(raws, grads, state, lr=1e-3, b1=0.9, b2=0.999, eps=1e-8):
    """One torch.optim.Adam step minimising `loss` where grads = d(loss)/d(raw)."""
    state["t"] = state.get("t", 0) + 1
    t = state["t"]
    out = []
    for n, (r, g) in enumerate(zip(raws, grads)):
        m = state.setdefault(("m", n), np.zeros_like(r))
        v = state.setdefault(("v", n), np.zeros_like(r))
        m[...] = b1 * m + (1 - b1) * g
        v[...] = b2 * v + (1 - b2) * g * g
        bc1 = 1 - b1 ** t
        bc2 = 1 - b2 ** t
        denom = np.sqrt(v) / np.sqrt(bc2) + eps
        out.append(r - (lr / bc1) * m / denom)
    return out


# --------------------------------------------------------------------------------------
# Titsias sparse variational bound -- gpr/model.py:700-724
# --------------------------------------------------------------------------------------
def titsias_elbo(spec, scale, X, y, Z, jitter=1e-8):
    """scale is a SCALAR (gpr/model.py:686-689).  Z (M,1+D) inducing inputs with channel column."""
    from scipy.linalg import solve_triangular
    y = y.reshape(-1, 1)
    N = X.shape[0]
    Kff_diag = mo_K_diag(spec, X)
    Kuf = mo_K(spec, Z, X)
    Kuu = mo_K(spec, Z)
    M = Kuu.shape[0]
    idx = np.arange(M)
    Kuu[idx, idx] = Kuu[idx, idx] + effective_jitter(jitter) * np.mean(np.diagonal(Kuu))
    Luu = np.linalg.cholesky(Kuu)
    v = solve_triangular(Luu, Kuf, lower=True)
    Q = v @ v.T
    s2 = scale ** 2
    L = np.linalg.cholesky(Q / s2 + np.eye(M))
    c = solve_triangular(L, v @ y, lower=True) / s2
    p = -0.5 * N * np.log(TWO_PI)
    p -= np.sum(np.log(np.diagonal(L)))
    p -= N * np.log(scale)
    p -= 0.5 * (y.T @ y).item() / s2
    p += 0.5 * (c.T @ c).item()
    p -= 0.5 * (np.sum(Kff_diag) - np.trace(Q)) / s2
    return p
