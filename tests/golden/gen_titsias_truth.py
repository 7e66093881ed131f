"""Ground truth for dELBO/dZ of the Titsias bound at the conditioning of BASELINE.json configs[4] -- and the reference's own distance from it.

At configs[4] (M = 2048 inducing points on a grid 0.2 apart, K_uu of condition ~1e11) the gradient with respect to the inducing inputs is the
O(1e-2) residue of O(1e4) terms: the reference's fp64 value moves by 2.35e-3 of the tensor when only its thread count changes, so "1e-5 against
one reference run" cannot be asserted.  This script evaluates the SAME function -- the bound of reference gpr/model.py:700-724 as a function of Z,
with the term table, X, y, Z the fp64 numbers the device receives -- entirely in 80-bit extended precision (numpy.longdouble: Gram matrices with
expl / cosl, Cholesky, triangular solves, adjoints, the kernel's derivative; eps 1.1e-19, so ~1e-8 of the tensor survives the conditioning),
at N reduced to 20 000 points (same grid, same spacing, same K_uu), and runs the reference itself (torch fp64) on the same inputs.
Output: tests/golden/titsias_dz_truth.npz = {meta, scale, Z, gz_truth, gz_ref, elbo_truth, loss_ref, ref_err}.  The device test asserts that
its dELBO/dZ is no further from the truth than the reference's.
usage (build container, reference importable; ~35 min on one core):  python tests/golden/gen_titsias_truth.py [N=20000] [M=2048]
round 5, the truth AT configs[4] (column chunks over 8 processes):  python tests/golden/gen_titsias_truth.py 100000 2048 --workers 8 --out titsias_dz_truth_cfg5.npz"""
import os
import sys
import time
import types
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
LD = np.longdouble
PI = 4 * np.arctan(LD(1))
TWO_PI = 2 * PI


def kblock(tab, x1, x2, want_j=False):
    """K[a, b] = sum_t A exp(-V u^2 / 2) cos(2 pi (M u + Psi)),  u = x1_a - x2_b + Delta;  J = dK / dx1   (extended precision)"""
    K = np.zeros((x1.size, x2.size), dtype=LD)
    J = np.zeros_like(K) if want_j else None
    d = x1[:, None] - x2[None, :]
    for A, Psi, V, Mm, Dl in tab.astype(LD):
        u = d + Dl
        E = A * np.exp(-V * u * u / 2)
        ph = TWO_PI * (Mm * u + Psi)
        c = np.cos(ph)
        K += E * c
        if want_j:
            J += E * (-V * u * c - TWO_PI * Mm * np.sin(ph))
    return K, J


def chol_ld(A, nb=256):
    A = A.copy(); n = A.shape[0]
    for k0 in range(0, n, nb):
        k1 = min(k0 + nb, n)
        blk = A[k0:k1, k0:k1]
        for j in range(k1 - k0):
            blk[j, j] = np.sqrt(blk[j, j] - blk[j, :j] @ blk[j, :j])
            if j + 1 < k1 - k0:
                blk[j + 1:, j] = (blk[j + 1:, j] - blk[j + 1:, :j] @ blk[j, :j]) / blk[j, j]
        if k1 < n:
            P = A[k1:, k0:k1]
            for j in range(k1 - k0):
                P[:, j] = (P[:, j] - P[:, :j] @ blk[j, :j]) / blk[j, j]
            A[k1:, k1:] -= P @ P.T
    return np.tril(A)


def trsm_ld(L, Bm, trans=False, nb=256):
    Xm = Bm.copy(); n = L.shape[0]
    blocks = list(range(0, n, nb))
    if not trans:
        for k0 in blocks:
            k1 = min(k0 + nb, n)
            if k0 > 0:
                Xm[k0:k1] -= L[k0:k1, :k0] @ Xm[:k0]
            for j in range(k0, k1):
                Xm[j] = (Xm[j] - L[j, k0:j] @ Xm[k0:j]) / L[j, j]
    else:
        for k0 in reversed(blocks):
            k1 = min(k0 + nb, n)
            if k1 < n:
                Xm[k0:k1] -= L[k1:, k0:k1].T @ Xm[k1:]
            for j in reversed(range(k0, k1)):
                Xm[j] = (Xm[j] - L[j + 1:k1, j] @ Xm[j + 1:k1]) / L[j, j]
    return Xm


def truth(table, Z, X, y, sigma, jitter):
    """-> (elbo, dELBO/dZ[:, 1]) in extended precision.  Kuu as the reference builds it: lower channel-pair blocks and their mirror."""
    t0 = time.time()
    C = table.shape[0]
    cz, cx = Z[:, 0].astype(np.int64), X[:, 0].astype(np.int64)
    z, x = Z[:, 1].astype(LD), X[:, 1].astype(LD)
    M, N = z.size, x.size
    rz = [np.nonzero(cz == c)[0] for c in range(C)]
    rx = [np.nonzero(cx == c)[0] for c in range(C)]
    Kuu = np.zeros((M, M), dtype=LD)
    for i in range(C):
        for j in range(i + 1):
            Kb, _ = kblock(table[i, j], z[rz[i]], z[rz[j]])
            Kuu[np.ix_(rz[i], rz[j])] = Kb
            if j < i:
                Kuu[np.ix_(rz[j], rz[i])] = Kb.T
    B = np.zeros((M, N), dtype=LD)
    for i in range(C):
        for j in range(C):
            B[np.ix_(rz[i], rx[j])] = kblock(table[i, j], z[rz[i]], x[rx[j]])[0]
    print("  Gram matrices %.0f s" % (time.time() - t0), flush=True)
    s2 = LD(sigma) * LD(sigma)
    I = np.eye(M, dtype=LD)
    A = Kuu + LD(jitter) * np.mean(np.diagonal(Kuu)) * I
    L = chol_ld(A)
    v = trsm_ld(L, B)
    print("  v = L^-1 Kuf %.0f s" % (time.time() - t0), flush=True)
    yv = y.astype(LD).reshape(-1, 1)
    Qm = v @ v.T
    Qs = Qm / s2 + I
    Lq = chol_ld(Qs)
    vy = v @ yv
    t1 = trsm_ld(Lq, trsm_ld(Lq, vy), True)                      # Pq v y
    beta = trsm_ld(L, t1, True)
    r = yv / s2 ** 2 - (B.T @ beta) / s2 ** 3
    Pv = trsm_ld(Lq, trsm_ld(Lq, v), True)
    print("  Pq v %.0f s" % (time.time() - t0), flush=True)
    GB = trsm_ld(L, (v - Pv) / s2, True) + beta @ r.T
    Pq = trsm_ld(Lq, trsm_ld(Lq, I), True)
    Em = 2 * I - Pq - Qs
    T1 = trsm_ld(L, Em, True)
    GA = trsm_ld(L, T1.T, True).T / 2 - (beta @ beta.T) / (2 * s2 ** 2)
    GA = (GA + GA.T) / 2
    print("  adjoints %.0f s" % (time.time() - t0), flush=True)
    kff = sum(rx[c].size * kblock(table[c, c], np.zeros(1, dtype=LD), np.zeros(1, dtype=LD))[0][0, 0] for c in range(C))
    logdet_q = 2 * np.sum(np.log(np.diagonal(Lq)))
    elbo = (-LD(N) / 2 * np.log(TWO_PI) - logdet_q / 2 - LD(N) * np.log(LD(sigma)) - (yv.T @ yv)[0, 0] / (2 * s2)
            + (t1.T @ vy)[0, 0] / (2 * s2 ** 2) - (kff - np.trace(Qm)) / (2 * s2))
    gz = np.zeros(M, dtype=LD)
    for i in range(C):
        for j in range(C):
            _, J = kblock(table[i, j], z[rz[i]], x[rx[j]], True)
            gz[rz[i]] += np.sum(GB[np.ix_(rz[i], rx[j])] * J, axis=1)
            if i >= j:
                _, Ju = kblock(table[i, j], z[rz[i]], z[rz[j]], True)
            else:                                                 # the mirrored block: K_ab = K^(j,i)(z_b, z_a), stationary in z_b - z_a
                _, Jt = kblock(table[j, i], z[rz[j]], z[rz[i]], True)
                Ju = -Jt.T
            gz[rz[i]] += 2 * np.sum(GA[np.ix_(rz[i], rz[j])] * Ju, axis=1)
    print("  done %.0f s" % (time.time() - t0), flush=True)
    return float(elbo), gz.astype(np.float64)


# ---- the same evaluation in column chunks over worker processes (round 5: the truth AT configs[4], N = 100 000) --------------------------------
# numpy.longdouble products do not go through BLAS and run on one core: at N = 100 000 the seven M x M x N products of truth() would take ~3 h.  Every
# M x N quantity is a function of its own columns only (v = L^-1 Kuf, Pq v, the adjoint of Kuf) and enters the rest through sums over columns
# (v v^T, v y, the row sums of the adjoint times the kernel derivative), so the columns are dealt to worker processes in chunks: pass 1 returns the
# sums, the M x M algebra runs in the parent, pass 2 the share of dELBO/dZ.  Same formulas, same extended precision as truth().
_W = {}


def _w_init(table, Z, X, y, L):
    _W.update(table=table, Z=Z, X=X, y=y, L=L)
    C = table.shape[0]
    cz, cx = Z[:, 0].astype(np.int64), X[:, 0].astype(np.int64)
    _W["rz"] = [np.nonzero(cz == c)[0] for c in range(C)]
    _W["cx"] = cx


def _kuf(cols, want_j=False):
    table, Z, X, rz, cx = _W["table"], _W["Z"], _W["X"], _W["rz"], _W["cx"]
    C = table.shape[0]
    z, x = Z[:, 1].astype(LD), X[cols, 1].astype(LD)
    B = np.zeros((z.size, x.size), dtype=LD)
    J = np.zeros_like(B) if want_j else None
    for j in range(C):
        cj = np.nonzero(cx[cols] == j)[0]
        if cj.size == 0:
            continue
        for i in range(C):
            Kb, Jb = kblock(table[i, j], z[rz[i]], x[cj], want_j)
            B[np.ix_(rz[i], cj)] = Kb
            if want_j:
                J[np.ix_(rz[i], cj)] = Jb
    return B, J


def _pass1(cols):
    B, _ = _kuf(cols)
    v = trsm_ld(_W["L"], B)
    yv = _W["y"][cols].astype(LD).reshape(-1, 1)
    return v @ v.T, v @ yv, (yv.T @ yv)[0, 0]


def _pass2(args):
    cols, Lq, beta, s2 = args
    B, J = _kuf(cols, True)
    v = trsm_ld(_W["L"], B)
    yv = _W["y"][cols].astype(LD).reshape(-1, 1)
    r = yv / s2 ** 2 - (B.T @ beta) / s2 ** 3
    Pv = trsm_ld(Lq, trsm_ld(Lq, v), True)
    GB = trsm_ld(_W["L"], (v - Pv) / s2, True) + beta @ r.T
    return np.sum(GB * J, axis=1)


def truth_chunked(table, Z, X, y, sigma, jitter, workers, chunk=3125):
    import multiprocessing as mp
    t0 = time.time()
    C = table.shape[0]
    cz = Z[:, 0].astype(np.int64)
    z = Z[:, 1].astype(LD)
    M, N = z.size, X.shape[0]
    rz = [np.nonzero(cz == c)[0] for c in range(C)]
    Kuu = np.zeros((M, M), dtype=LD)
    for i in range(C):
        for j in range(i + 1):
            Kb, _ = kblock(table[i, j], z[rz[i]], z[rz[j]])
            Kuu[np.ix_(rz[i], rz[j])] = Kb
            if j < i:
                Kuu[np.ix_(rz[j], rz[i])] = Kb.T
    s2 = LD(sigma) * LD(sigma)
    I = np.eye(M, dtype=LD)
    A = Kuu + LD(jitter) * np.mean(np.diagonal(Kuu)) * I
    L = chol_ld(A)
    print("  Kuu factored %.0f s" % (time.time() - t0), flush=True)
    chunks = [np.arange(c0, min(c0 + chunk, N)) for c0 in range(0, N, chunk)]
    with mp.get_context("fork").Pool(workers, initializer=_w_init, initargs=(table, Z, X, y, L)) as pool:
        Qm, vy, yy = np.zeros((M, M), dtype=LD), np.zeros((M, 1), dtype=LD), LD(0)
        for q, w, t in pool.imap(_pass1, chunks):                 # (in chunk order: the sums are reproducible)
            Qm += q; vy += w; yy += t
        print("  pass 1 (v v^T, v y) %.0f s" % (time.time() - t0), flush=True)
        Qs = Qm / s2 + I
        Lq = chol_ld(Qs)
        t1 = trsm_ld(Lq, trsm_ld(Lq, vy), True)                  # Pq v y
        beta = trsm_ld(L, t1, True)
        Pq = trsm_ld(Lq, trsm_ld(Lq, I), True)
        Em = 2 * I - Pq - Qs
        T1 = trsm_ld(L, Em, True)
        GA = trsm_ld(L, T1.T, True).T / 2 - (beta @ beta.T) / (2 * s2 ** 2)
        GA = (GA + GA.T) / 2
        print("  M x M algebra %.0f s" % (time.time() - t0), flush=True)
        gz = np.zeros(M, dtype=LD)
        for part in pool.imap(_pass2, [(c, Lq, beta, s2) for c in chunks]):
            gz += part
        print("  pass 2 (adjoint of Kuf) %.0f s" % (time.time() - t0), flush=True)
    cx = X[:, 0].astype(np.int64)
    kff = sum(int(np.sum(cx == c)) * kblock(table[c, c], np.zeros(1, dtype=LD), np.zeros(1, dtype=LD))[0][0, 0] for c in range(C))
    logdet_q = 2 * np.sum(np.log(np.diagonal(Lq)))
    elbo = (-LD(N) / 2 * np.log(TWO_PI) - logdet_q / 2 - LD(N) * np.log(LD(sigma)) - yy / (2 * s2)
            + (t1.T @ vy)[0, 0] / (2 * s2 ** 2) - (kff - np.trace(Qm)) / (2 * s2))
    for i in range(C):
        for j in range(C):
            if i >= j:
                _, Ju = kblock(table[i, j], z[rz[i]], z[rz[j]], True)
            else:
                _, Jt = kblock(table[j, i], z[rz[j]], z[rz[i]], True)
                Ju = -Jt.T
            gz[rz[i]] += 2 * np.sum(GA[np.ix_(rz[i], rz[j])] * Ju, axis=1)
    print("  done %.0f s" % (time.time() - t0), flush=True)
    return float(elbo), gz.astype(np.float64)


def main():
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("N", nargs="?", type=int, default=20000)
    ap.add_argument("M", nargs="?", type=int, default=2048)
    ap.add_argument("--workers", type=int, default=0, help="> 0: the column-chunked evaluation on this many processes (the N = 100 000 fixture: 8)")
    ap.add_argument("--out", default="titsias_dz_truth.npz")
    ap.add_argument("--add-ref-run", type=int, default=0, metavar="THREADS",
                    help="add one more fp64 run of the reference, on this many torch threads, to an existing fixture (gz_ref_alt, ref_alt_threads, "
                         "ref_alt_err): how far from the truth the reference lands depends on its own summation order")
    a = ap.parse_args()
    N, M = a.N, a.M
    C, Q = 4, 3
    from mogptk_amd import synth, gpr as agpr
    X, y = synth.make_data(N, C)
    h = synth.mosm_hypers(C, Q)
    s = float(np.mean(h["scale"]))
    jitter = 1e-8
    # the reference on the same inputs (torch fp64)
    ip, disp = types.ModuleType("IPython"), types.ModuleType("IPython.display")
    disp.display = lambda *a, **k: None; disp.HTML = lambda v: v; ip.display = disp
    sys.modules["IPython"] = ip; sys.modules["IPython.display"] = disp
    sys.path.insert(0, "/root/reference")
    import torch
    import mogptk
    if a.add_ref_run > 0:
        out = dict(np.load(os.path.join(HERE, a.out)))
        assert [int(v) for v in out["meta"]] == [C, Q, 1, 1, N, M], "the fixture was made for another size"
        torch.set_num_threads(a.add_ref_run)
        g = mogptk.gpr
        T = lambda v: torch.tensor(np.asarray(v), dtype=torch.float64)
        k = g.MultiOutputSpectralMixtureKernel(Q=Q, output_dims=C, input_dims=1)
        k.weight.assign(h["weight"]); k.mean.assign(h["mean"]); k.variance.assign(h["variance"]); k.delay.assign(h["delay"]); k.phase.assign(h["phase"])
        m = g.Titsias(k, T(X), T(y), Z=[M // C] * C, Z_init="grid", variance=s ** 2, jitter=jitter)
        m.likelihood.scale.assign(s)
        t0 = time.time()
        loss = float(m.loss())
        gz = -m.Z.grad.detach().numpy()[:, 1].copy()
        truth = out["gz_truth"]
        err = float(np.max(np.abs(gz - truth)) / np.max(np.abs(truth)))
        print("reference on %d threads: loss %.10f in %.0f s; %.3e of the tensor from the truth (the fixture's first run: %.3e), %.3e from that run"
              % (a.add_ref_run, loss, time.time() - t0, err, float(out["ref_err"]), float(np.max(np.abs(gz - out["gz_ref"])) / np.max(np.abs(truth)))))
        out.update(gz_ref_alt=gz, ref_alt_threads=np.array(a.add_ref_run), ref_alt_err=np.array(err), loss_ref_alt=np.array(loss))
        np.savez_compressed(os.path.join(HERE, a.out), **out)
        return
    g = mogptk.gpr
    T = lambda a: torch.tensor(np.asarray(a), dtype=torch.float64)
    k = g.MultiOutputSpectralMixtureKernel(Q=Q, output_dims=C, input_dims=1)
    k.weight.assign(h["weight"]); k.mean.assign(h["mean"]); k.variance.assign(h["variance"]); k.delay.assign(h["delay"]); k.phase.assign(h["phase"])
    m = g.Titsias(k, T(X), T(y), Z=[M // C] * C, Z_init="grid", variance=s ** 2, jitter=jitter)
    m.likelihood.scale.assign(s)
    t0 = time.time()
    loss = float(m.loss())
    gz_ref = -m.Z.grad.detach().numpy()[:, 1].copy()         # loss = -elbo; m.Z: the inducing-point Parameter (reference gpr/model.py:696)
    print("reference: loss %.10f in %.0f s, |dELBO/dZ|max %.3e" % (loss, time.time() - t0, np.abs(gz_ref).max()), flush=True)
    out = {"meta": np.array([C, Q, 1, 1, N, M]), "loss_ref": np.array(loss), "gz_ref": gz_ref}
    params = list(m.parameters())
    for n, p in enumerate(params):                            # the layout tests/helpers.py:fixture_params reads (gen_golden.py:dump_params)
        out["p%d_raw" % n] = p.data.detach().numpy().copy()
        out["p%d_lower" % n] = np.array(np.nan) if p.lower is None else np.asarray(p.lower.detach().numpy() if torch.is_tensor(p.lower) else p.lower)
        out["p%d_upper" % n] = np.array(np.nan) if p.upper is None else np.asarray(p.upper.detach().numpy() if torch.is_tensor(p.upper) else p.upper)
        out["p%d_cons" % n] = p().detach().numpy().copy()
        out["p%d_grad" % n] = np.array(np.nan) if p.grad is None else p.grad.detach().numpy().copy()
    out["names"] = np.array([p._name for p in params])
    del m, k
    import gc
    gc.collect()
    # this package's model with the reference's RAW parameter values: its term table, inducing inputs and noise scale are the fp64 numbers the
    # device receives (host algebra pinned on the reference elsewhere) -- the function whose exact derivative is taken below
    ka = agpr.MultiOutputSpectralMixtureKernel(Q=Q, output_dims=C)
    for name in ("weight", "mean", "variance", "delay", "phase"):
        getattr(ka, name).assign(h[name])
    ma = agpr.Titsias(ka, X, y, Z=[M // C] * C, variance=s ** 2)
    ma.likelihood.scale.assign(s)
    for pa, n in zip(ma.parameters(), range(len(params))):
        assert pa.data.shape == tuple(out["p%d_raw" % n].shape), pa._name
        pa.data = out["p%d_raw" % n].copy()
    table = np.asarray(ka._spectral_terms(1), dtype=np.float64)
    Z = np.asarray(ma.kernel._kernel_format(ma.Z()), dtype=np.float64)
    sigma = float(np.asarray(ma.likelihood.scale()).reshape(-1)[0])
    if a.workers > 0:
        elbo, gz = truth_chunked(table, Z, X, y, sigma, jitter, a.workers)
    else:
        elbo, gz = truth(table, Z, X, y, sigma, jitter)
    ref_err = float(np.max(np.abs(gz_ref - gz)) / np.max(np.abs(gz)))
    print("extended precision: elbo %.10f (reference %.10f), |dELBO/dZ|max %.3e; the reference's fp64 run is %.3e of the tensor away"
          % (elbo, -loss, np.abs(gz).max(), ref_err))
    out.update(scale=np.array(s), Z=Z, gz_truth=gz, elbo_truth=np.array(elbo), ref_err=np.array(ref_err))
    np.savez_compressed(os.path.join(HERE, a.out), **out)


if __name__ == "__main__":
    main()
