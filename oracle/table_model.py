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


def table_block(tab, x1, x2, sin=False):
    """one channel-pair block from T term rows [A, Psi, V_d, M_d, Delta_d]; x1 (n1,D), x2 (n2,D).
    returns per-term arrays E*cos (or E*sin) WITHOUT the amplitude, and u (T,n1,n2,D)."""
    D = x1.shape[1]
    A, Psi = tab[:, 0], tab[:, 1]
    V, M, Dl = tab[:, 2:2 + D], tab[:, 2 + D:2 + 2 * D], tab[:, 2 + 2 * D:]
    u = (x1[None, :, None, :] - x2[None, None, :, :]) + Dl[:, None, None, :]
    E = np.exp(-0.5 * np.einsum("tnmd,td->tnm", u * u, V))
    ph = TWO_PI * (np.einsum("tnmd,td->tnm", u, M) + Psi[:, None, None])
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
        if not grad:
            return dict(lml=lml, moments=None, diagG=None, trG=0.0, jitter_abs=jit)
        Li = solve_triangular(L, np.eye(self.N), lower=True)
        G = 0.5 * (alpha @ alpha.T - Li.T @ Li)
        C, T, D = self.C, self.T, self.D
        c = self.X[:, 0].astype(np.int64)
        mom = np.zeros((C * (C + 1) // 2, T, 2 + 3 * D))
        for i in range(C):
            ri = np.nonzero(c == i)[0]
            for j in range(i + 1):
                rj = np.nonzero(c == j)[0]
                if len(ri) == 0 or len(rj) == 0:
                    continue
                Ec, Es, u = table_block(self.table[i, j], self.X[ri, 1:], self.X[rj, 1:])
                g = G[np.ix_(ri, rj)] * (1.0 if i == j else 2.0)
                m = mom[i * (i + 1) // 2 + j]
                m[:, 0] = np.einsum("nm,tnm->t", g, Ec)
                m[:, 1] = np.einsum("nm,tnm->t", g, Es)
                m[:, 2:2 + D] = np.einsum("nm,tnm,tnmd->td", g, Ec, u * u)
                m[:, 2 + D:2 + 2 * D] = np.einsum("nm,tnm,tnmd->td", g, Ec, u)
                m[:, 2 + 2 * D:] = np.einsum("nm,tnm,tnmd->td", g, Es, u)
        dG = np.diagonal(G)
        diagG = np.array([np.sum(dG[c == k]) for k in range(C)])
        return dict(lml=lml, moments=mom, diagG=diagG, trG=float(np.sum(dG)), jitter_abs=jit)

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
        kdiag = np.asarray(kss_diag)[cs]
        return mu, (kdiag - np.sum(v * v, axis=0)).reshape(-1, 1)
