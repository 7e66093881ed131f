"""
ORACLE -- TEST INFRASTRUCTURE ONLY.  numpy model of what the device computes from the unified spectral term
table: Gram from the table, LML, G = 1/2(alpha alpha^T - Kj^-1), the gradient moments with
the symmetric double count, and the predictive equations.  Used by tests to check (a) the HIP kernels' raw
outputs (moments, diagG, trG) and (b) the host chain rule without a GPU.  Parity: pinned through
tests/test_host_logic.py, which drives the host chain rule with this model and compares against the
reference's autograd gradients in tests/golden/lml_*.npz.
"""
import numpy as np

TWO_PI = 2.0 * np.pi


def table_block(tab, x1, x2, sin=False, with_mid=False):
    """one channel-pair block from T term rows [A, Psi, V_d, M_d, Delta_d (, L_d, c_d)]; x1 (n1,D), x2 (n2,D).
    returns per-term arrays E*cos (or E*sin) WITHOUT the amplitude, and u (T,n1,n2,D); rows of width 2+5D carry a Gaussian envelope
    exp(-1/2 sum_d L_d a_d^2) on the midpoint offset a_d = (x1_d + x2_d)/2 - c_d (with_mid: also return a)."""
    D = x1.shape[1]
    A, Psi = tab[:, 0], tab[:, 1]
    V, M, Dl = tab[:, 2:2 + D], tab[:, 2 + D:2 + 2 * D], tab[:, 2 + 2 * D:2 + 3 * D]
    u = (x1[None, :, None, :] - x2[None, None, :, :]) + Dl[:, None, None, :]
    E = np.exp(-0.5 * np.einsum("tnmd,td->tnm", u * u, V))
    a = None
    if tab.shape[1] > 2 + 3 * D:
        Lv, cn = tab[:, 2 + 3 * D:2 + 4 * D], tab[:, 2 + 4 * D:2 + 5 * D]
        a = 0.5 * (x1[None, :, None, :] + x2[None, None, :, :]) - cn[:, None, None, :]
        E = E * np.exp(-0.5 * np.einsum("tnmd,td->tnm", a * a, Lv))
    ph = TWO_PI * (np.einsum("tnmd,td->tnm", u, M) + Psi[:, None, None])
    if with_mid:
        return E * np.cos(ph), E * np.sin(ph), u, a
    return E * np.cos(ph), E * np.sin(ph), u


def gram_from_table(table, X1, X2=None):
    C = table.shape[0]
    c1 = X1[:, 0].astype(np.int64)
    X2_ = X1 if X2 is None else X2
    c2 = X2_[:, 0].astype(np.int64)
    K = np.zeros((X1.shape[0], X2_.shape[0]))
    for i in range(C):
        r1 = np.nonzero(c1 == i)[0]
        for j in range(C):
            r2 = np.nonzero(c2 == j)[0]
            if len(r1) == 0 or len(r2) == 0:
                continue
            tab = table[i, j] if (X2 is not None or i >= j) else None
            if tab is None:       # symmetric case: mirror of the lower block (kernel.py:466-467)
                Ec, _, _ = table_block(table[j, i], X1[r2, 1:], X1[r1, 1:])
                K[np.ix_(r1, r2)] = np.einsum("t,tnm->nm", table[j, i][:, 0], Ec).T
            else:
                Ec, _, _ = table_block(tab, X1[r1, 1:], X2_[r2, 1:])
                K[np.ix_(r1, r2)] = np.einsum("t,tnm->nm", tab[:, 0], Ec)
    return K


class TableDevice:
    """numpy stand-in with the same methods as mogptk_amd._lib.ExactHandle"""

    def __init__(self, device, X, y, C):
        self.X = np.array(X, dtype=np.float64)
        self.y = np.array(y, dtype=np.float64).reshape(-1, 1)
        self.N, self.D, self.C = X.shape[0], X.shape[1] - 1, C

    def set_y(self, y):
        self.y = np.array(y, dtype=np.float64).reshape(-1, 1)

    def set_terms(self, table):
        self.table = np.array(table, dtype=np.float64)
        self.T = table.shape[2]

    def set_point_diag(self, kdiag):
        pass                         # the twin takes the diagonal from its own Gram matrix

    def _Kj(self, noise_var, jitter, data_var):
        K = gram_from_table(self.table, self.X)
        c = self.X[:, 0].astype(np.int64)
        d = np.diagonal(K) + np.asarray(noise_var)[c] + (0.0 if data_var is None else data_var)
        jit = jitter * np.mean(d)
        K[np.arange(self.N), np.arange(self.N)] = d + jit
        return K, jit

    def eval(self, noise_var, jitter, grad=True, data_var=None):
        from scipy.linalg import solve_triangular
        K, jit = self._Kj(noise_var, jitter, data_var)
        L = np.linalg.cholesky(K)
        z = solve_triangular(L, self.y, lower=True)
        alpha = solve_triangular(L.T, z, lower=False)
        lml = -0.5 * self.N * np.log(TWO_PI) - np.sum(np.log(np.diagonal(L))) - 0.5 * (self.y.T @ alpha).item()
        self._last = (K, alpha)
        if not grad:
            return dict(lml=lml, moments=None, diagG=None, trG=0.0, jitter_abs=jit)
        Li = solve_triangular(L, np.eye(self.N), lower=True)
        G = 0.5 * (alpha @ alpha.T - Li.T @ Li)
        C, T, D = self.C, self.T, self.D
        c = self.X[:, 0].astype(np.int64)
        W = self.table.shape[3]
        mom = np.zeros((C * (C + 1) // 2, T, W))
        for i in range(C):
            ri = np.nonzero(c == i)[0]
            for j in range(i + 1):
                rj = np.nonzero(c == j)[0]
                if len(ri) == 0 or len(rj) == 0:
                    continue
                Ec, Es, u, a = table_block(self.table[i, j], self.X[ri, 1:], self.X[rj, 1:], with_mid=True)
                g = G[np.ix_(ri, rj)] * (1.0 if i == j else 2.0)
                m = mom[i * (i + 1) // 2 + j]
                m[:, 0] = np.einsum("nm,tnm->t", g, Ec)
                m[:, 1] = np.einsum("nm,tnm->t", g, Es)
                m[:, 2:2 + D] = np.einsum("nm,tnm,tnmd->td", g, Ec, u * u)
                m[:, 2 + D:2 + 2 * D] = np.einsum("nm,tnm,tnmd->td", g, Ec, u)
                m[:, 2 + 2 * D:2 + 3 * D] = np.einsum("nm,tnm,tnmd->td", g, Es, u)
                if a is not None:
                    m[:, 2 + 3 * D:2 + 4 * D] = np.einsum("nm,tnm,tnmd->td", g, Ec, a * a)
                    m[:, 2 + 4 * D:2 + 5 * D] = np.einsum("nm,tnm,tnmd->td", g, Ec, a)
        dG = np.diagonal(G)
        diagG = np.array([np.sum(dG[c == k]) for k in range(C)])
        return dict(lml=lml, moments=mom, diagG=diagG, trG=float(np.sum(dG)), jitter_abs=jit)

    def fetch(self, which):
        """what mogp_model_fetch returns after the last evaluation: 0 = W = L^-1 in channel-sorted row order, 1 = Kj^-1, 2 = alpha"""
        from scipy.linalg import solve_triangular
        K, alpha = self._last
        if which == 2:
            return alpha.reshape(-1).copy()
        if which == 1:
            return np.linalg.inv(K)
        order = np.argsort(self.X[:, 0], kind="stable")
        L = np.linalg.cholesky(K[np.ix_(order, order)])
        return solve_triangular(L, np.eye(self.N), lower=True)

    def predict(self, noise_var, jitter, kss_diag, Xs, full=False, data_var=None):
        from scipy.linalg import solve_triangular
        K, _ = self._Kj(noise_var, jitter, data_var)
        L = np.linalg.cholesky(K)
        Kfs = gram_from_table(self.table, self.X, Xs)
        alpha = solve_triangular(L.T, solve_triangular(L, self.y, lower=True), lower=False)
        v = solve_triangular(L, Kfs, lower=True)
        mu = Kfs.T @ alpha
        if full:
            return mu, gram_from_table(self.table, Xs) - v.T @ v
        cs = Xs[:, 0].astype(np.int64)
        kdiag = np.asarray(kss_diag) if self.table.shape[3] > 2 + 3 * self.D else np.asarray(kss_diag)[cs]      # enveloped terms: per test point
        return mu, (kdiag - np.sum(v * v, axis=0)).reshape(-1, 1)


# ----------------------------------------------------------------------------------------------------------------
# Titsias sparse bound from the term table (what mogp_titsias_eval computes) -- reference gpr/model.py:700-724.
# Whitened forms:  A = Kuu + jitter*mean(diag Kuu) I = Luu Luu^T, W = Luu^-1, v = W Kuf, Q = v v^T, Qs = Q/s2 + I = Lq Lq^T,
# Pq = Qs^-1, R = I - Pq = Pq Q / s2, beta = W^T Pq v y.
#   dELBO/dKuf = W^T (R v)/s2 + beta (y/s2^2 - Kuf^T beta/s2^3)^T
#   dELBO/dA   = 1/2 W^T (R - Q/s2) W - 1/2 beta beta^T / s2^2
# ----------------------------------------------------------------------------------------------------------------
def _jr_block(tab, x1, x2):
    """J_ab,d = d K_ab / d x1_a,d = sum_t A E [ -V_d u_d cos - 2 pi M_d sin ]   (n1, n2, D)"""
    D = x1.shape[1]
    A, V, M = tab[:, 0], tab[:, 2:2 + D], tab[:, 2 + D:2 + 2 * D]
    Ec, Es, u, a = table_block(tab, x1, x2, with_mid=True)
    J = np.einsum("t,tnm,tnmd,td->nmd", A, Ec, u, -V) + np.einsum("t,tnm,td->nmd", A, Es, -TWO_PI * M)
    if a is not None:             # envelope on the midpoint: d/dx1 of exp(-1/2 L a^2) = -1/2 L a
        J = J + np.einsum("t,tnm,tnmd,td->nmd", A, Ec, a, -0.5 * tab[:, 2 + 3 * D:2 + 4 * D])
    return J


def _jc_block(tab, x1, x2):
    """d K_ab / d x2_b,d (n1, n2, D): minus the stationary part of _jr_block, plus the SAME envelope part (the midpoint moves with both inputs)"""
    D = x1.shape[1]
    A, V, M = tab[:, 0], tab[:, 2:2 + D], tab[:, 2 + D:2 + 2 * D]
    Ec, Es, u, a = table_block(tab, x1, x2, with_mid=True)
    J = -(np.einsum("t,tnm,tnmd,td->nmd", A, Ec, u, -V) + np.einsum("t,tnm,td->nmd", A, Es, -TWO_PI * M))
    if a is not None:
        J = J + np.einsum("t,tnm,tnmd,td->nmd", A, Ec, a, -0.5 * tab[:, 2 + 3 * D:2 + 4 * D])
    return J


def moments_dense(table, G, X1, X2, sym):
    """moments of a dense adjoint G (rows X1, cols X2).  sym: lower channel pairs with the symmetric double count
    (G symmetric, X2 is X1) -> (P, T, W); otherwise all ordered pairs -> (C*C, T, W)."""
    C, T = table.shape[0], table.shape[2]
    D = X1.shape[1] - 1
    c1, c2 = X1[:, 0].astype(np.int64), X2[:, 0].astype(np.int64)
    env = table.shape[3] > 2 + 3 * D
    out = np.zeros(((C * (C + 1) // 2) if sym else C * C, T, table.shape[3]))
    for i in range(C):
        ri = np.nonzero(c1 == i)[0]
        for j in range((i + 1) if sym else C):
            rj = np.nonzero(c2 == j)[0]
            if len(ri) == 0 or len(rj) == 0:
                continue
            Ec, Es, u, amid = table_block(table[i, j], X1[ri, 1:], X2[rj, 1:], with_mid=True)
            g = G[np.ix_(ri, rj)] * (2.0 if (sym and i != j) else 1.0)
            m = out[i * (i + 1) // 2 + j] if sym else out[i * C + j]
            m[:, 0] = np.einsum("nm,tnm->t", g, Ec)
            m[:, 1] = np.einsum("nm,tnm->t", g, Es)
            m[:, 2:2 + D] = np.einsum("nm,tnm,tnmd->td", g, Ec, u * u)
            m[:, 2 + D:2 + 2 * D] = np.einsum("nm,tnm,tnmd->td", g, Ec, u)
            m[:, 2 + 2 * D:2 + 3 * D] = np.einsum("nm,tnm,tnmd->td", g, Es, u)
            if env:
                m[:, 2 + 3 * D:2 + 4 * D] = np.einsum("nm,tnm,tnmd->td", g, Ec, amid * amid)
                m[:, 2 + 4 * D:2 + 5 * D] = np.einsum("nm,tnm,tnmd->td", g, Ec, amid)
            if sym and i == j:          # odd-in-tau moments cancel over the full symmetric block
                m[:, 1] = 0.0
                m[:, 2 + D:2 + 2 * D] = 0.0
    return out


def _all_reduce(self, sharded):
    """sum over the ranks holding the other shards of the data: `TableDevice.reduce` is set by the test to the group's all-reduce"""
    if not sharded:
        return lambda a: a
    red = getattr(self, "reduce", None)
    if red is None:
        raise RuntimeError("sharded evaluation of the numpy twin: set TableDevice.reduce to an all-reduce over the ranks")
    return lambda a: red(np.ascontiguousarray(a, dtype=np.float64))


def titsias_eval(self, Z, sigma, jitter, kff_diag, grad=True, sharded=False):
    """numpy twin of mogp_titsias_eval; sharded: of mogp_titsias_eval_sharded -- this object holds one shard of the data, the sums over data
    points (v v^T, v y, y^T y, N, sum K_ff,nn; then the (Z, X) moments and their share of d/dZ) are all-reduced at the same places"""
    from scipy.linalg import solve_triangular
    X, y, table, C = self.X, self.y, self.table, self.C
    M, D = Z.shape[0], self.D
    red = _all_reduce(self, sharded)
    s2 = sigma * sigma
    cz = Z[:, 0].astype(np.int64)
    cx = X[:, 0].astype(np.int64)
    Kuu = gram_from_table(table, Z)
    jit = jitter * np.mean(np.diagonal(Kuu))
    A = Kuu + jit * np.eye(M)
    B = gram_from_table(table, Z, X)
    Luu = np.linalg.cholesky(A)
    W = solve_triangular(Luu, np.eye(M), lower=True)
    v = W @ B
    Q = red(v @ v.T)
    vy = red(v @ y)
    env = table.shape[3] > 2 + 3 * D                       # enveloped terms: K_ff,diag per training point
    yy, N, kff = red(np.array([(y.T @ y).item(), float(X.shape[0]),
                               float(np.sum(np.asarray(kff_diag)) if env else np.sum(np.asarray(kff_diag)[cx]))]))
    Lq = np.linalg.cholesky(Q / s2 + np.eye(M))
    c = solve_triangular(Lq, vy, lower=True) / s2
    elbo = (-0.5 * N * np.log(TWO_PI) - np.sum(np.log(np.diagonal(Lq))) - N * np.log(sigma) - 0.5 * yy / s2
            + 0.5 * (c.T @ c).item() - 0.5 * (kff - np.trace(Q)) / s2)
    if not grad:
        return dict(elbo=elbo, jitter_abs=jit)
    Lqi = solve_triangular(Lq, np.eye(M), lower=True)
    Pq = Lqi.T @ Lqi
    R = Pq @ Q / s2
    beta = W.T @ (Pq @ vy)
    r = y / s2 ** 2 - (B.T @ beta) / s2 ** 3
    GB = W.T @ (R @ v) / s2 + beta @ r.T
    GA = 0.5 * W.T @ (R - Q / s2) @ W - 0.5 * (beta @ beta.T) / s2 ** 2
    GA = 0.5 * (GA + GA.T)
    ds2 = (-0.5 * N / s2 + 0.5 * np.trace(Pq @ Q) / s2 ** 2 + 0.5 * yy / s2 ** 2
           - (vy.T @ Pq @ vy).item() / s2 ** 3 + 0.5 * (vy.T @ Pq @ Q @ Pq @ vy).item() / s2 ** 4
           + 0.5 * (kff - np.trace(Q)) / s2 ** 2)
    mom_uu = moments_dense(table, GA, Z, Z, sym=True)
    mom_uf = red(moments_dense(table, GB, Z, X, sym=False))
    gZ_uf, gZ_uu = np.zeros((M, D)), np.zeros((M, D))
    for i in range(C):
        ri = np.nonzero(cz == i)[0]
        if len(ri) == 0:
            continue
        for j in range(C):
            rj = np.nonzero(cx == j)[0]
            if len(rj):
                gZ_uf[ri] += np.einsum("nm,nmd->nd", GB[np.ix_(ri, rj)], _jr_block(table[i, j], Z[ri, 1:], X[rj, 1:]))
            zj = np.nonzero(cz == j)[0]
            if len(zj):             # K_uu depends on z_a through row a AND column a: sum_b GA_ab dK_ab/dz_a + sum_b GA_ba dK_ba/dz_a
                tij = table[i, j] if i >= j else None
                if tij is not None:
                    Jr = _jr_block(tij, Z[ri, 1:], Z[zj, 1:])
                else:               # block (i, j) above the diagonal is the transpose of block (j, i): K_ab = K'_ba
                    Jr = np.transpose(_jc_block(table[j, i], Z[zj, 1:], Z[ri, 1:]), (1, 0, 2))
                gZ_uu[ri] += 2.0 * np.einsum("nm,nmd->nd", GA[np.ix_(ri, zj)], Jr)
    return dict(elbo=elbo, jitter_abs=jit, mom_uu=mom_uu, mom_uf=mom_uf, gZ=red(gZ_uf) + gZ_uu, trGA=float(np.trace(GA)),
                dsigma=2.0 * sigma * ds2)


def titsias_predict(self, Z, sigma, jitter, Xs, kss_diag, sharded=False):
    """reference gpr/model.py:730-765"""
    from scipy.linalg import solve_triangular
    X, y, table = self.X, self.y, self.table
    red = _all_reduce(self, sharded)
    s2 = sigma * sigma
    M = Z.shape[0]
    Kuu = gram_from_table(table, Z)
    A = Kuu + jitter * np.mean(np.diagonal(Kuu)) * np.eye(M)
    Luu = np.linalg.cholesky(A)
    v = solve_triangular(Luu, gram_from_table(table, Z, X), lower=True)
    Lq = np.linalg.cholesky(red(v @ v.T) / s2 + np.eye(M))
    a = solve_triangular(Luu, gram_from_table(table, Z, Xs), lower=True)
    b = solve_triangular(Lq, a, lower=True)
    c = solve_triangular(Lq, red(v @ y), lower=True) / s2
    mu = b.T @ c
    kd = np.asarray(kss_diag) if table.shape[3] > 2 + 3 * self.D else np.asarray(kss_diag)[Xs[:, 0].astype(np.int64)]      # enveloped terms: per test point
    var = kd - np.sum(a * a, axis=0) + np.sum(b * b, axis=0)
    self._sparse_pred = (np.array(Xs), a, b)
    return mu, var.reshape(-1, 1)


def sparse_predict_cov(self, S):
    """twin of mogp_sparse_predict_cov: K_ss - a^T a + b^T b of the last sparse prediction (reference gpr/model.py:758-760, 870-872)"""
    Xs, a, b = self._sparse_pred
    assert Xs.shape[0] == S
    return gram_from_table(self.table, Xs) - a.T @ a + b.T @ b


TableDevice.sparse_predict_cov = sparse_predict_cov
TableDevice.titsias_eval = titsias_eval
TableDevice.titsias_predict = titsias_predict


def snelson_eval(self, Z, noise_var, jitter, kff_diag, grad=True, sharded=False):
    """numpy twin of mogp_snelson_eval -- reference gpr/model.py:516-541 (Snelson & Ghahramani's pseudo-input GP, FITC):
        g_n = Kff_nn - Qff_nn + sigma_c(n)^2,   p = log N(y | 0, Qff + diag g),   Qff = Kfu Kuu^-1 Kuf,
    and the adjoints the device contracts with the kernel derivatives (A = Kuu + jitter, B = Kuf, v = L^-1 B, G = diag 1/g,
    Bq = I + v G v^T, Pq = Bq^-1, r = Pq v G y, alpha = G (y - v^T r), h_n = 1/2 (alpha_n^2 - [S^-1]_nn), beta = L^-T r):
        dp/dB = L^-T (r alpha^T - Pq v G - 2 v diag h),   dp/dA = 1/2 L^-T (I - Pq + 2 v diag(h) v^T) L^-1 - 1/2 beta beta^T,
        dp/dKff_nn = dp/dsigma_n^2 = h_n."""
    from scipy.linalg import solve_triangular
    X, y, table, C = self.X, self.y, self.table, self.C
    N, M, D = X.shape[0], Z.shape[0], self.D
    red = _all_reduce(self, sharded)          # sharded (mogp_snelson_eval_sharded): this object holds one shard; sums over points all-reduced
    cz = Z[:, 0].astype(np.int64)
    cx = X[:, 0].astype(np.int64)
    noise_var = np.asarray(noise_var, dtype=np.float64).reshape(-1)
    s2 = noise_var[cx] if noise_var.size > 1 else np.full(N, noise_var[0])
    Kuu = gram_from_table(table, Z)
    jit = jitter * np.mean(np.diagonal(Kuu))
    A = Kuu + jit * np.eye(M)
    B = gram_from_table(table, Z, X)
    Luu = np.linalg.cholesky(A)
    v = solve_triangular(Luu, B, lower=True)
    env = table.shape[3] > 2 + 3 * D                       # enveloped terms: K_ff,diag per training point
    g = (np.asarray(kff_diag) if env else np.asarray(kff_diag)[cx]) - np.sum(v * v, axis=0) + s2
    G = 1.0 / g
    yv = y.reshape(-1)
    Bq = red((v * G) @ v.T) + np.eye(M)
    Lq = np.linalg.cholesky(Bq)
    vGy = red(v @ (G * yv))
    c = solve_triangular(Lq, vGy, lower=True)
    slg, yGy, Nt = red(np.array([np.sum(np.log(g)), np.sum(yv * yv * G), float(N)]))
    p = -0.5 * Nt * np.log(TWO_PI) - np.sum(np.log(np.diagonal(Lq))) - 0.5 * slg - 0.5 * yGy + 0.5 * c @ c
    if not grad:
        return dict(lml=p, jitter_abs=jit)
    Lqi = solve_triangular(Lq, np.eye(M), lower=True)
    Pq = Lqi.T @ Lqi
    r = Pq @ vGy
    alpha = G * (yv - v.T @ r)
    R1 = Pq @ v                                              # M x N
    Sinv_diag = G - G * G * np.sum(v * R1, axis=0)
    h = 0.5 * (alpha * alpha - Sinv_diag)
    W = solve_triangular(Luu, np.eye(M), lower=True)
    beta = W.T @ r
    GB = W.T @ (np.outer(r, alpha) - R1 * G - 2.0 * v * h)
    GA = 0.5 * W.T @ (np.eye(M) - Pq + 2.0 * red((v * h) @ v.T)) @ W - 0.5 * np.outer(beta, beta)
    GA = 0.5 * (GA + GA.T)
    mom_uu = moments_dense(table, GA, Z, Z, sym=True)
    mom_uf = red(moments_dense(table, GB, Z, X, sym=False))
    gZ = np.zeros((M, D))
    gZ_uu = np.zeros((M, D))
    for i in range(C):
        ri = np.nonzero(cz == i)[0]
        if len(ri) == 0:
            continue
        for j in range(C):
            rj = np.nonzero(cx == j)[0]
            if len(rj):
                gZ[ri] += np.einsum("nm,nmd->nd", GB[np.ix_(ri, rj)], _jr_block(table[i, j], Z[ri, 1:], X[rj, 1:]))
            zj = np.nonzero(cz == j)[0]
            if len(zj):             # K_uu depends on z_a through row a AND column a: sum_b GA_ab dK_ab/dz_a + sum_b GA_ba dK_ba/dz_a
                tij = table[i, j] if i >= j else None
                if tij is not None:
                    Jr = _jr_block(tij, Z[ri, 1:], Z[zj, 1:])
                else:               # block (i, j) above the diagonal is the transpose of block (j, i): K_ab = K'_ba
                    Jr = np.transpose(_jc_block(table[j, i], Z[zj, 1:], Z[ri, 1:]), (1, 0, 2))
                gZ_uu[ri] += 2.0 * np.einsum("nm,nmd->nd", GA[np.ix_(ri, zj)], Jr)
    hsum = h if env else red(np.bincount(cx, weights=h, minlength=C).astype(np.float64))   # per channel (per point with an envelope): d p / d Kff_diag = d p / d sigma^2
    return dict(lml=p, jitter_abs=jit, mom_uu=mom_uu, mom_uf=mom_uf, gZ=red(gZ) + gZ_uu, trGA=float(np.trace(GA)), hsum=hsum)


def snelson_predict(self, Z, noise_var, jitter, Xs, kff_diag, kss_diag, sharded=False):
    """reference gpr/model.py:543-576"""
    from scipy.linalg import solve_triangular
    X, y, table = self.X, self.y, self.table
    M = Z.shape[0]
    cx = X[:, 0].astype(np.int64)
    noise_var = np.asarray(noise_var, dtype=np.float64).reshape(-1)
    s2 = noise_var[cx] if noise_var.size > 1 else np.full(X.shape[0], noise_var[0])
    Kuu = gram_from_table(table, Z)
    A = Kuu + jitter * np.mean(np.diagonal(Kuu)) * np.eye(M)
    Luu = np.linalg.cholesky(A)
    v = solve_triangular(Luu, gram_from_table(table, Z, X), lower=True)
    env = table.shape[3] > 2 + 3 * self.D
    G = 1.0 / ((np.asarray(kff_diag) if env else np.asarray(kff_diag)[cx]) - np.sum(v * v, axis=0) + s2)
    red = _all_reduce(self, sharded)
    Lq = np.linalg.cholesky(red((v * G) @ v.T) + np.eye(M))
    a = solve_triangular(Luu, gram_from_table(table, Z, Xs), lower=True)
    b = solve_triangular(Lq, a, lower=True)
    c = solve_triangular(Lq, red(v @ (G * y.reshape(-1))), lower=True)
    mu = b.T @ c
    var = (np.asarray(kss_diag) if env else np.asarray(kss_diag)[Xs[:, 0].astype(np.int64)]) - np.sum(a * a, axis=0) + np.sum(b * b, axis=0)
    return mu.reshape(-1, 1), var.reshape(-1, 1)


TableDevice.snelson_eval = snelson_eval
TableDevice.snelson_predict = snelson_predict


def svgp_forward(self, Z, q_mu, q_sqrt, jitter, kff_diag, Xs=None, kss_diag=None, dense=False):
    """numpy twin of mogp_svgp_forward -- reference gpr/model.py:851-868 (SparseHensman._predict_f, whitened q(u) = N(L q_mu, L S S^T L^T)):
        a = L^-1 K(Z, X*),  b = tril(q_sqrt)^T a,  mu = a^T q_mu,  var = K_diag - colsum(a^2) + colsum(b^2)
    at the training inputs (Xs None; the state for svgp_backward is kept) or at test inputs.  dense (the non-sparse model, :834-840, at its
    own training inputs): Z = X, a = L^T exactly and var = colsum(b^2)."""
    from scipy.linalg import solve_triangular
    table = self.table
    M = Z.shape[0]
    Kuu = gram_from_table(table, Z)
    jit = jitter * np.mean(np.diagonal(Kuu))
    Luu = np.linalg.cholesky(Kuu + jit * np.eye(M))
    Xq = self.X if Xs is None else Xs
    kd = np.asarray(kff_diag if Xs is None else kss_diag)
    if self.table.shape[3] <= 2 + 3 * self.D:               # per channel; enveloped terms: already per point
        kd = kd[Xq[:, 0].astype(np.int64)]
    S = np.tril(np.asarray(q_sqrt, dtype=np.float64))
    qm = np.asarray(q_mu, dtype=np.float64).reshape(-1)
    if dense and Xs is None:
        a = Luu.T.copy()
        b = S.T @ a
        mu, var = a.T @ qm, np.sum(b * b, axis=0)
    else:
        a = solve_triangular(Luu, gram_from_table(table, Z, Xq), lower=True)
        b = S.T @ a
        mu = a.T @ qm
        var = kd - np.sum(a * a, axis=0) + np.sum(b * b, axis=0)
    if Xs is None:
        self._svgp = dict(Z=np.array(Z), Luu=Luu, v=a, S=S, q_mu=qm, dense=bool(dense))
    else:
        self._sparse_pred = (np.array(Xs), a, b)
    return dict(mu=mu, var=var, jitter_abs=jit)


def svgp_backward(self, e, f, sharded=False):
    """adjoints of  E(mu, var)  with  e = dE/dmu, f = dE/dvar  (any likelihood: the caller differentiates its own expectation):
        Gv = q_mu e^T + 2 (S S^T - I) v diag f,   dE/dB = L^-T Gv,   dE/dA = -1/2 L^-T Psi(Gv v^T) L^-1   (Psi(Y) = tril(Y) mirrored),
        dE/dq_mu = v e,   dE/dS = 2 (v diag(f) v^T) S  (lower part);
    dense (v = L^T, a function of A alone):  Gv = q_mu e^T + 2 S S^T v diag f,   dE/dA = +1/2 L^-T Psi(v Gv^T) L^-1."""
    from scipy.linalg import solve_triangular
    st = self._svgp
    X, table, C, D = self.X, self.table, self.C, self.D
    Z, Luu, v, S, q_mu, dense = st["Z"], st["Luu"], st["v"], st["S"], st["q_mu"], st["dense"]
    M = Z.shape[0]
    e = np.asarray(e, dtype=np.float64).reshape(-1)
    f = np.asarray(f, dtype=np.float64).reshape(-1)
    cz = Z[:, 0].astype(np.int64)
    cx = X[:, 0].astype(np.int64)
    W = solve_triangular(Luu, np.eye(M), lower=True)
    psi = lambda Y: np.tril(Y) + np.tril(Y, -1).T
    red = _all_reduce(self, sharded)          # sharded: this object holds one shard of the data; sums over points are all-reduced
    if dense:
        Gv = np.outer(q_mu, e) + 2.0 * (S @ (S.T @ v)) * f
        GB = np.zeros((M, X.shape[0]))
        GA = 0.5 * W.T @ psi(v @ Gv.T) @ W
    else:
        Gv = np.outer(q_mu, e) + 2.0 * (S @ (S.T @ v) - v) * f
        GB = W.T @ Gv
        GA = -0.5 * W.T @ psi(red(np.tril(Gv @ v.T))) @ W
    GA = 0.5 * (GA + GA.T)
    mom_uu = moments_dense(table, GA, Z, Z, sym=True)
    mom_uf = red(moments_dense(table, GB, Z, X, sym=False))
    gZ = np.zeros((M, D))
    gZ_uu = np.zeros((M, D))
    if not dense:
        for i in range(C):
            ri = np.nonzero(cz == i)[0]
            if len(ri) == 0:
                continue
            for j in range(C):
                rj = np.nonzero(cx == j)[0]
                if len(rj):
                    gZ[ri] += np.einsum("nm,nmd->nd", GB[np.ix_(ri, rj)], _jr_block(table[i, j], Z[ri, 1:], X[rj, 1:]))
                zj = np.nonzero(cz == j)[0]
                if len(zj):
                    gZ_uu[ri] += 2.0 * np.einsum("nm,nmd->nd", GA[np.ix_(ri, zj)], _jr_block(table[i, j], Z[ri, 1:], Z[zj, 1:]))
    g_qmu = red(v @ e)
    g_S = np.tril(2.0 * red((v * f) @ v.T) @ S)
    return dict(mom_uu=mom_uu, mom_uf=mom_uf, gZ=red(gZ) + gZ_uu, trGA=float(np.trace(GA)), g_qmu=g_qmu, g_qsqrt=g_S)


TableDevice.svgp_forward = svgp_forward
TableDevice.svgp_backward = svgp_backward


def oa_forward(self, q_nu, q_lambda):
    """numpy twin of mogp_oa_forward -- reference gpr/model.py:613-634 (OpperArchambeau.elbo): q(f) = N(K nu, (K^-1 + diag(lambda^2))^-1),
        B = Lambda K Lambda + I = L L^T,   mu = K nu,   var = (1 - diag(B^-1)) / lambda^2,
        kl = nu^T K nu + log det B + tr(B^-1) - N        (the reference's `kl`: the ELBO is  E(mu, var) - kl / 2)
    No jitter anywhere (the reference's _cholesky is called with add_jitter False here)."""
    K = gram_from_table(self.table, self.X)
    nu = np.asarray(q_nu, dtype=np.float64).reshape(-1)
    lam = np.asarray(q_lambda, dtype=np.float64).reshape(-1)
    N = K.shape[0]
    B = lam[:, None] * lam[None, :] * K + np.eye(N)
    L = np.linalg.cholesky(B)
    Binv = np.linalg.inv(B)
    Binv = 0.5 * (Binv + Binv.T)
    mu = K @ nu
    var = (1.0 - np.diagonal(Binv)) / lam ** 2
    kl = float(nu @ mu + 2.0 * np.sum(np.log(np.diagonal(L))) + np.trace(Binv) - N)
    self._oa = dict(K=K, Binv=Binv, nu=nu, lam=lam)
    return dict(mu=mu, var=var, kl=kl)


def oa_backward(self, e, f):
    """gradient of  E(mu, var) - kl / 2  given  e = dE/dmu, f = dE/dvar  per point:
        dK  = 1/2 (e nu^T + nu e^T) - 1/2 nu nu^T + Lambda (B^-1 diag(w) B^-1 - 1/2 B^-1) Lambda,   w = f / lambda^2 + 1/2
        dnu = K (e - nu)
        dlambda_m = -2 f_m (1 - b_m) / lambda_m^3 + (2 / lambda_m) (f_m b_m / lambda_m^2 - r_m) - (1 - b_m) / lambda_m + (b_m - s_m) / lambda_m
                    with b = diag(B^-1), s = diag(B^-2), r = diag(B^-1 diag(f / lambda^2) B^-1)
    (only the diagonals of the matrix products enter the lambda gradient, through  K Lambda = Lambda^-1 (B - I))."""
    st = self._oa
    K, Binv, nu, lam = st["K"], st["Binv"], st["nu"], st["lam"]
    e = np.asarray(e, dtype=np.float64).reshape(-1)
    f = np.asarray(f, dtype=np.float64).reshape(-1)
    d = f / lam ** 2
    Y = (Binv * (d + 0.5)) @ Binv
    GK = 0.5 * (np.outer(e, nu) + np.outer(nu, e)) - 0.5 * np.outer(nu, nu) + lam[:, None] * (Y - 0.5 * Binv) * lam[None, :]
    GK = 0.5 * (GK + GK.T)
    b = np.diagonal(Binv)
    s = np.sum(Binv * Binv, axis=1)
    r = np.sum(Binv * Binv * d[None, :], axis=1)
    g_lam = -2.0 * f * (1.0 - b) / lam ** 3 + (2.0 / lam) * (d * b - r) - (1.0 - b) / lam + (b - s) / lam
    return dict(mom=moments_dense(self.table, GK, self.X, self.X, sym=True), g_nu=K @ (e - nu), g_lambda=g_lam)


def oa_predict(self, q_nu, q_lambda, kss_diag, Xs, full=False):
    """reference gpr/model.py:640-668:  mu = K_sf nu,  var = K_ss - K_sf (K + diag(1 / lambda^2))^-1 K_fs  (no jitter)"""
    from scipy.linalg import solve_triangular
    nu = np.asarray(q_nu, dtype=np.float64).reshape(-1)
    lam = np.asarray(q_lambda, dtype=np.float64).reshape(-1)
    Kfs = gram_from_table(self.table, self.X, Xs)
    L = np.linalg.cholesky(gram_from_table(self.table, self.X) + np.diag(1.0 / lam ** 2))
    a = solve_triangular(L, Kfs, lower=True)
    mu = (Kfs.T @ nu).reshape(-1, 1)
    if full:
        return mu, gram_from_table(self.table, Xs) - a.T @ a
    kd = np.asarray(kss_diag)[Xs[:, 0].astype(np.int64)]
    return mu, (kd - np.sum(a * a, axis=0)).reshape(-1, 1)


TableDevice.oa_forward = oa_forward
TableDevice.oa_backward = oa_backward
TableDevice.oa_predict = oa_predict


# ----------------------------------------------------------------------------------------------------------------
# numpy twin of the SHARDED evaluation stages (mogp_shard_*, mogptk_amd/csrc/sweep.hip): single-sweep blocked
# inversion with 128-row tiles owned cyclically (tile row i -> rank i % world), 512-wide pivot blocks, the panel of
# each pivot block assembled by ONE all-gather (pivot-block columns of every owned tile row + left part of the owned pivot rows), the serial chain repeated
# by every rank and the rank-512 update restricted to owned rows.  Drives the same mogptk_amd.dist.sharded_eval as the
# device handle, so tests/test_dist_cpu.py exercises the real orchestration under gloo with two CPU ranks.
# ----------------------------------------------------------------------------------------------------------------
TILE, OB = 128, 4


def _shard_begin(self, rank, world, noise_var, jitter, data_var=None):
    self.rank, self.world = rank, world
    X = self.X
    order = np.argsort(X[:, 0], kind="stable")            # the device works in channel-sorted order
    self._order = order
    Xs = X[order]
    K = gram_from_table(self.table, Xs)
    c = Xs[:, 0].astype(np.int64)
    d = np.diagonal(K) + np.asarray(noise_var)[c] + (0.0 if data_var is None else np.asarray(data_var)[order])
    jit = jitter * np.mean(d)
    N = self.N
    self.Npad = -(-N // TILE) * TILE
    self.nb = self.Npad // TILE
    A = np.eye(self.Npad)
    A[:N, :N] = K
    A[np.arange(N), np.arange(N)] = d + jit
    self.A = np.tril(A)                                      # lower storage, like the device
    self.ys = np.zeros(self.Npad); self.ys[:N] = self.y[order, 0]
    self.logdet = 0.0
    self._Xs = Xs
    return jit, -(-self.nb // OB)


def _geom(self, kb):
    k0 = kb * OB
    k1 = min(k0 + OB, self.nb)
    Kd = (k1 - k0) * TILE
    maxrows, maxpiv = -(-(self.nb - k0) // self.world), -(-(k1 - k0) // self.world)
    rowoff = maxrows * TILE * Kd
    return k0, k1, Kd, maxrows, rowoff, rowoff + maxpiv * TILE * k0 * TILE


def _owned_rows(self, k0, r):
    first = k0 + ((r - k0 % self.world) + self.world) % self.world
    return list(range(first, self.nb, self.world))


def _shard_pack(self, kb):
    k0, k1, Kd, maxrows, rowoff, chunk = _geom(self, kb)
    cols = k0 * TILE
    send = np.zeros(chunk)
    for idx, i in enumerate(_owned_rows(self, k0, self.rank)):
        send[idx * TILE * Kd:(idx + 1) * TILE * Kd] = self.A[i * TILE:(i + 1) * TILE, k0 * TILE:k0 * TILE + Kd].reshape(-1)
        if i < k1 and cols:          # pivot tile row: its part left of the block travels in the same chunk
            send[rowoff + idx * TILE * cols:rowoff + (idx + 1) * TILE * cols] = self.A[i * TILE:(i + 1) * TILE, :cols].reshape(-1)
    self._recv = np.zeros(chunk * self.world)
    return send, self._recv, chunk


def _shard_unpack(self, kb):
    k0, k1, Kd, maxrows, rowoff, chunk = _geom(self, kb)
    cols = k0 * TILE
    for r in range(self.world):
        for idx, i in enumerate(_owned_rows(self, k0, r)):
            self.A[i * TILE:(i + 1) * TILE, k0 * TILE:k0 * TILE + Kd] = self._recv[r * chunk + idx * TILE * Kd:r * chunk + (idx + 1) * TILE * Kd].reshape(TILE, Kd)
            if i < k1 and cols:
                self.A[i * TILE:(i + 1) * TILE, :cols] = self._recv[r * chunk + rowoff + idx * TILE * cols:r * chunk + rowoff + (idx + 1) * TILE * cols].reshape(TILE, cols)


def _shard_block(self, kb):
    k0, k1, Kd = _geom(self, kb)[:3]
    A, T_ = self.A, TILE
    a0, a1 = k0 * T_, k1 * T_
    S = A[a0:a1, a0:a1]
    S = np.tril(S) + np.tril(S, -1).T
    L = np.linalg.cholesky(S)
    self.logdet += np.sum(np.log(np.diagonal(L)))
    Li = np.linalg.inv(L)
    P = Li.T @ Li
    Uc = A[a1:, a0:a1].copy()                      # rows below
    Ur = A[a0:a1, :a0].copy()                      # row block (transposed part of the panel)
    A[a1:, a0:a1] = Uc @ P
    A[a0:a1, :a0] = P @ Ur
    A[a0:a1, a0:a1] = -P
    Xc, Xr = A[a1:, a0:a1], A[a0:a1, :a0]
    mine = lambda i: (i % self.world) == self.rank
    for i in range(self.nb):                       # rank-Kd update of the owned tile rows outside the pivot block
        if k0 <= i < k1 or not mine(i):
            continue
        rows = slice(i * T_, (i + 1) * T_)
        if i >= k1:
            xi = Xc[(i - k1) * T_:(i - k1 + 1) * T_]
            A[rows, a1:(i + 1) * T_] -= xi @ Uc[:(i - k1 + 1) * T_].T          # (a): columns below the block, j <= i
            A[rows, :a0] -= xi @ Ur                                             # (b)
        else:
            A[rows, :(i + 1) * T_] -= Xr[:, rows].T @ Ur[:, :(i + 1) * T_]      # (c)
    self.A = A


def _shard_alpha(self):
    A = np.tril(self.A)
    mask = np.array([(i // TILE) % self.world == self.rank for i in range(self.Npad)])
    Am = A * mask[:, None]                        # owned rows only
    z = Am @ self.ys + Am.T @ self.ys - np.where(mask, np.diagonal(A) * self.ys, 0.0)
    self._alpha = -z
    return self._alpha, self.Npad


def _shard_finish(self):
    N, C, T, D = self.N, self.C, self.T, self.D
    alpha = self._alpha[:N]
    lml = -0.5 * N * np.log(TWO_PI) - self.logdet - 0.5 * float(self.ys[:N] @ alpha)
    Kinv = -(np.tril(self.A) + np.tril(self.A, -1).T)[:N, :N]
    mask = np.array([(i // TILE) % self.world == self.rank for i in range(N)])
    G = 0.5 * (np.outer(alpha, alpha) - Kinv)
    hi_mine = np.maximum.outer(np.arange(N), np.arange(N))
    Gm = G * mask[hi_mine]                        # an entry belongs to the owner of row max(a, b)
    Xs = self._Xs
    c = Xs[:, 0].astype(np.int64)
    mom = np.zeros((C * (C + 1) // 2, T, 2 + 3 * D))
    for i in range(C):
        ri = np.nonzero(c == i)[0]
        for j in range(i + 1):
            rj = np.nonzero(c == j)[0]
            if len(ri) == 0 or len(rj) == 0:
                continue
            Ec, Es, u = table_block(self.table[i, j], Xs[ri, 1:], Xs[rj, 1:])
            g = Gm[np.ix_(ri, rj)]
            if i == j:                            # lower triangle counted twice, diagonal once (full symmetric sum)
                g = np.tril(g, -1) * 2.0 + np.diag(np.diagonal(g))
            else:
                g = g * 2.0
            m = mom[i * (i + 1) // 2 + j]
            m[:, 0] = np.einsum("nm,tnm->t", g, Ec)
            m[:, 1] = 0.0 if i == j else np.einsum("nm,tnm->t", g, Es)
            m[:, 2:2 + D] = np.einsum("nm,tnm,tnmd->td", g, Ec, u * u)
            m[:, 2 + D:2 + 2 * D] = 0.0 if i == j else np.einsum("nm,tnm,tnmd->td", g, Ec, u)
            m[:, 2 + 2 * D:2 + 3 * D] = np.einsum("nm,tnm,tnmd->td", g, Es, u)
    dG = np.where(mask, np.diagonal(G), 0.0)
    diagG = np.array([np.sum(dG[c == k]) for k in range(C)])
    return lml, mom, diagG


for _n, _f in (("shard_begin", _shard_begin), ("shard_pack", _shard_pack), ("shard_unpack", _shard_unpack),
               ("shard_block", _shard_block),
               ("shard_alpha", _shard_alpha), ("shard_finish", _shard_finish)):
    setattr(TableDevice, _n, _f)
TableDevice.mem_get = lambda self, buf, count: np.array(buf[:count], dtype=np.float64)
TableDevice.mem_put = lambda self, buf, arr: buf.__setitem__(slice(0, len(arr)), arr)


# ----------------------------------------------------------------------------------------------------------------
# The exact evaluation at sizes where TableDevice.eval's handful of dense N x N temporaries do not fit (BASELINE.json
# configs[2], N = 32768: 8.6 GB each).  Same formulas, ONE N x N array: the Gram goes into it block row by block row, LAPACK
# factors it in place (dpotrf), turns the factor into Kj^-1 in place (dpotri), and the moments are taken from row chunks of
# G = 1/2 (alpha alpha^T - Kj^-1) that are never held as a matrix.  Rows must be channel-contiguous (what the wrappers pass,
# reference model.py:594-598).  Pinned on TableDevice.eval itself at small N (tests/test_oracle_golden.py) and on the
# reference's own forward LML at N = 32768 (tests/golden/gen_cfg3.py).
# ----------------------------------------------------------------------------------------------------------------
class TableDeviceLean(TableDevice):
    chunk = 256

    def eval(self, noise_var, jitter, grad=True, data_var=None):
        N, C, D, T = self.N, self.C, self.D, self.T
        c = self.X[:, 0].astype(np.int64)
        if np.any(np.diff(c) < 0):
            raise ValueError("TableDeviceLean needs channel-contiguous rows")
        lo = np.searchsorted(c, np.arange(C), side="left")
        hi = np.searchsorted(c, np.arange(C), side="right")
        x = self.X[:, 1:]
        K = np.empty((N, N))
        for i in range(C):
            for j in range(i + 1):
                if hi[i] == lo[i] or hi[j] == lo[j]:
                    continue
                tab = self.table[i, j]
                for r0 in range(lo[i], hi[i], self.chunk):
                    r1 = min(r0 + self.chunk, hi[i])
                    Ec, _, _ = table_block(tab, x[r0:r1], x[lo[j]:hi[j]])
                    K[r0:r1, lo[j]:hi[j]] = np.einsum("t,tnm->nm", tab[:, 0], Ec)
        idx = np.arange(N)
        d = K[idx, idx] + np.asarray(noise_var)[c] + (0.0 if data_var is None else np.asarray(data_var))
        jit = jitter * np.mean(d)
        K[idx, idx] = d + jit
        # LAPACK through torch's CPU build (MKL: what the reference itself factors with, gpr/model.py:246) when torch is there, scipy's otherwise.
        # (scipy 1.15's bundled OpenBLAS 0.3.28 returns dpotrf info = 16545 for THIS matrix at N = 32768 -- a positive definite matrix that MKL, the
        # device and OpenBLAS itself at N = 4096 factor without complaint; a random diagonally dominant matrix of the same size passes.)
        try:
            import torch
        except ImportError:
            torch = None
        if torch is not None:
            Kt = torch.from_numpy(K)
            L, info = torch.linalg.cholesky_ex(Kt)                       # reads the lower triangle only
            if int(info) != 0:
                raise np.linalg.LinAlgError("cholesky info=%d" % int(info))
            del Kt
            logdet_half = float(torch.log(torch.diagonal(L)).sum())
            alpha = torch.cholesky_solve(torch.from_numpy(self.y), L).numpy()
            lml = -0.5 * N * np.log(TWO_PI) - logdet_half - 0.5 * (self.y.T @ alpha).item()
            if not grad:
                return dict(lml=lml, moments=None, diagG=None, trG=0.0, jitter_abs=jit)
            K = None
            Ki = torch.cholesky_inverse(L)                               # full symmetric Kj^-1
            del L
            K = Ki.numpy()
        else:
            from scipy.linalg import lapack
            # K is C-ordered with its lower triangle filled: K.T is the Fortran-ordered matrix with its UPPER triangle filled
            U, info = lapack.dpotrf(K.T, lower=0, clean=0, overwrite_a=1)
            if info != 0:
                raise np.linalg.LinAlgError("dpotrf info=%d" % info)
            assert np.shares_memory(U, K)
            logdet_half = float(np.sum(np.log(K[idx, idx])))
            alpha, info = lapack.dpotrs(U, self.y, lower=0)
            lml = -0.5 * N * np.log(TWO_PI) - logdet_half - 0.5 * (self.y.T @ alpha).item()
            if not grad:
                return dict(lml=lml, moments=None, diagG=None, trG=0.0, jitter_abs=jit)
            Ui, info = lapack.dpotri(U, lower=0, overwrite_c=1)              # lower triangle of K (C order) now holds Kj^-1
            if info != 0:
                raise np.linalg.LinAlgError("dpotri info=%d" % info)
            assert np.shares_memory(Ui, K)
        for i in range(C):                                               # the diagonal channel blocks are read as full symmetric blocks
            B = K[lo[i]:hi[i], lo[i]:hi[i]]
            B[:] = np.tril(B) + np.tril(B, -1).T
        a = alpha[:, 0]
        W = self.table.shape[3]
        if W > 2 + 3 * D:
            raise ValueError("TableDeviceLean: plain (2 + 3 D) term rows only")
        mom = np.zeros((C * (C + 1) // 2, T, W))
        for i in range(C):
            for j in range(i + 1):
                if hi[i] == lo[i] or hi[j] == lo[j]:
                    continue
                tab = self.table[i, j]
                m = mom[i * (i + 1) // 2 + j]
                for r0 in range(lo[i], hi[i], self.chunk):
                    r1 = min(r0 + self.chunk, hi[i])
                    Ec, Es, u = table_block(tab, x[r0:r1], x[lo[j]:hi[j]])
                    g = (0.5 if i == j else 1.0) * (np.outer(a[r0:r1], a[lo[j]:hi[j]]) - K[r0:r1, lo[j]:hi[j]])
                    gc = g[None] * Ec
                    gs = g[None] * Es
                    m[:, 0] += gc.sum(axis=(1, 2))
                    m[:, 1] += gs.sum(axis=(1, 2))
                    gcu = gc[..., None] * u
                    m[:, 2:2 + D] += (gcu * u).sum(axis=(1, 2))
                    m[:, 2 + D:2 + 2 * D] += gcu.sum(axis=(1, 2))
                    m[:, 2 + 2 * D:2 + 3 * D] += (gs[..., None] * u).sum(axis=(1, 2))
        dG = 0.5 * (a * a - K[idx, idx])
        diagG = np.array([np.sum(dG[lo[k]:hi[k]]) for k in range(C)])
        self._last = None
        return dict(lml=lml, moments=mom, diagG=diagG, trG=float(np.sum(dG)), jitter_abs=jit)
