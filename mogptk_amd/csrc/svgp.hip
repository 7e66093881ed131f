// svgp.hip -- the variational GP of Hensman et al. (whitened), the O(N M^2) algebra around ANY likelihood, on the device.
// Reference: gpr/model.py:767-886 (SparseHensman / Hensman): q(u) = N(L q_mu, L S S^T L^T), L L^T = Kuu + jitter, S = tril(q_sqrt).
//
// Two calls per evaluation, the likelihood in between (host, O(N): its expectation E(mu, var) and e = dE/dmu, f = dE/dvar per point):
//   forward   a = L^-1 K(Z, X),  b = S^T a,  mu = a^T q_mu,  var = K_diag - colsum(a^2) + colsum(b^2)          (reference :851-868)
//             dense (the non-sparse model at its own inputs, :834-840): Z = X, a = L^T exactly, var = colsum(b^2)
//   backward  Gv = q_mu e^T + 2 (S S^T - I) v diag f                   (dense: without the - I)
//             dE/dKuf = L^-T Gv,   dE/dKuu = -1/2 L^-T Psi(Gv v^T) L^-1     (Psi(Y) = tril(Y) mirrored; dense: +1/2 L^-T Psi(v Gv^T) L^-1, no Kuf)
//             dE/dq_mu = v e,   dE/dS = 2 (v diag(f) v^T) S
// contracted with the kernel derivatives by the dense-mode moment kernel (which also yields d/dZ), like titsias.hip / snelson.hip.
// Z, q_mu and the rows of q_sqrt come in the caller's order; the device works with Z sorted by channel (rows of S permuted, columns kept).
// Checked against the reference's autograd through the numpy twin (oracle/table_model.py:svgp_forward / svgp_backward) and on the device.
#include "mogp_model.h"

#include <cmath>
#include <cstring>
#include <limits>

using namespace mogp;

#define RC(x) do { int r__ = (x); if (r__) return r__; } while (0)

namespace {



// out[m][n] = in[m][n] * s[n]
__global__ void k_sv_scale_cols(const double* __restrict__ in, double* __restrict__ out, int64_t ld, int64_t n, const double* __restrict__ s) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const int64_t r = blockIdx.y;
    out[r * ld + j] = in[r * ld + j] * s[j];
}
// a[i][j] = 0 for i > j   (a = L^T is upper triangular; the tiles above the diagonal of L are never written by the factorisation)
__global__ void k_sv_mask_upper(double* __restrict__ a, int64_t ld, int64_t n) {
    const int64_t i = blockIdx.x;
    for (int64_t j = threadIdx.x; j < i && j < n; j += blockDim.x) a[i * ld + j] = 0.0;
}
// Gv[m][n] = q[m] e[n] + 2 (Sb[m][n] - keep * v[m][n]) f[n]      (in place over Sb)
__global__ void k_sv_adjoint(double* __restrict__ Gv, const double* __restrict__ v, int64_t ld, int64_t n, const double* __restrict__ q,
                             const double* __restrict__ e, const double* __restrict__ f, double keep) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const int64_t i = blockIdx.y;
    Gv[i * ld + j] = q[i] * e[j] + 2.0 * (Gv[i * ld + j] - keep * v[i * ld + j]) * f[j];
}


// Kuu -> L (t.a), q_mu and S = tril(q_sqrt) onto the device in the sorted order of Z (t.vec[0:Mpad], t.R with rows permuted)
int svgp_setup(mogp_model* m, int64_t M, const double* Z, const double* q_mu, const double* q_sqrt, double jitter, SortedX& sz,
               std::vector<GTile>& tuu, std::vector<int>& psuu, double& jit, int64_t* info) {
    const int C = m->C, D = m->D, W = m->Wt;                   // 2 + 3 D, or 2 + 5 D: terms with an envelope on the input midpoint (MOHSM)
    const int64_t Npad = m->Npad;
    if (m->T <= 0) return fail(MOGP_EINVAL, "mogp_model_set_terms must be called before an evaluation");
    RC(sort_inputs(Z, M, D, C, MOGP_TILE, sz));
    // The whitened parametrisation q(u) = N(L q_mu, ..) depends on the ORDER of the inducing inputs through the Cholesky factor; the device
    // factorises Kuu with the inputs grouped by channel, so that is the order they have to come in (what init_inducing_points and the
    // reference's own data formatting produce).
    if (!sz.identity) return fail(MOGP_EINVAL, "the Hensman models take inducing inputs grouped by channel, in ascending channel order "
                                               "(the whitened variational parameters depend on their order)");
    const int64_t Mpad = sz.Mpad;
    if (!m->tw) m->tw = new TitsiasWork();
    TitsiasWork& t = *m->tw;
    if (t.Mpad != Mpad) {
        t.Mpad = Mpad;
        RC(spd_alloc(t.a, Mpad)); RC(spd_alloc(t.q, Mpad));
        RC(t.zx.ensure((size_t)D * Mpad));
        RC(t.B.ensure((size_t)Mpad * Npad)); RC(t.v.ensure((size_t)Mpad * Npad));
        { int r__ = dev_fill_zero(t.v.p, (size_t)Mpad * Npad * sizeof(double)); if (r__) return r__; }      // its padding is zero from here on (titsias.hip relies on it)
        RC(t.Qs.ensure((size_t)Mpad * Mpad));
        RC(t.vec.ensure((size_t)8 * Mpad + 4 * Npad));
        RC(t.scratch.ensure((size_t)(Mpad / 256 + 2) * std::max(Npad, Mpad) + (size_t)(Mpad / 512 + 2) * Mpad));
        RC(t.zero_noise.ensure(C));
        { int r__ = dev_fill_zero(t.zero_noise.p, C * sizeof(double)); if (r__) return r__; }
    }
    RC(t.R.ensure((size_t)Mpad * Mpad)); RC(t.E.ensure((size_t)Mpad * Mpad)); RC(t.GA.ensure((size_t)Mpad * Mpad));
    RC(t.GB.ensure((size_t)Mpad * Npad));
    RC(t.nvec.ensure((size_t)8 * Npad + 2 * C));
    m->gemm_ev_used = 0; m->gemm_launches = 0; m->gemm_flops = 0.0;
    build_sym_tiles(sz.off, C, tuu, psuu);
    RC(t.tiles_uu.ensure(tuu.size()));
    t.tile_key.clear();                                     // the (Z, Z) list is rewritten here: a Titsias evaluation on this handle must not trust its cached lists
    HIP_TRY(hipMemcpyAsync(t.zx.p, sz.xs.data(), (size_t)D * Mpad * sizeof(double), hipMemcpyHostToDevice, m->st));
    HIP_TRY(hipMemcpyAsync(t.tiles_uu.p, tuu.data(), tuu.size() * sizeof(GTile), hipMemcpyHostToDevice, m->st));
    const unsigned long long big = std::numeric_limits<unsigned long long>::max();
    HIP_TRY(hipMemcpyAsync(m->d_info.p, &big, sizeof(big), hipMemcpyHostToDevice, m->st));
    jit = jitter * table_diag_points(m, sz) / (double)M;        // relative jitter on Kuu (reference gpr/model.py:855 -> :244); with an envelope the diagonal follows Z
    GramArgs ga{};
    ga.tiles = t.tiles_uu.p; ga.xr = t.zx.p; ga.xc = t.zx.p; ga.ldxr = ga.ldxc = Mpad; ga.nrows = ga.ncols = M;
    RC(t.ph_zz.prepare(sz.off, sz.off, C, m->T, Mpad, Mpad, m->st, ga.ph));
    ga.table = m->d_table.p; ga.T = m->T; ga.D = D; ga.C = C; ga.W = W; ga.out = t.a.A.p; ga.ldo = Mpad;
    ga.noise = t.zero_noise.p; ga.dvar = nullptr; ga.jitter_abs = jit; ga.mirror = 0;
    RC(launch_gram(ga, (int)tuu.size(), m->st));
    RC(launch_pad_identity(t.a.A.p, Mpad, M, Mpad, m->st));
    t.a.keep_L = true;
    t.a.refine_panels = !(std::getenv("MOGP_REFINE_PANELS") && std::atoi(std::getenv("MOGP_REFINE_PANELS")) == 0);   // K_uu + jitter is ill-conditioned: mogp_api.hip:spd_potrf
    RC(spd_potrf(m, t.a));
    RC(spd_check_info(m, "Kuu", info));
    // q_mu and S in the device's order of the inducing points: row pos of the device = row sz.perm[pos] of the caller
    std::vector<double> hq(Mpad, 0.0), hS((size_t)Mpad * Mpad, 0.0);
    for (int64_t pos = 0; pos < M; ++pos) {
        const int64_t src = sz.perm[pos];
        hq[pos] = q_mu[src];
        for (int64_t c = 0; c <= src; ++c) hS[(size_t)pos * Mpad + c] = q_sqrt[(size_t)src * M + c];     // tril: columns <= the CALLER's row index
    }
    HIP_TRY(dev_upload(t.vec.p, hq.data(), Mpad * sizeof(double)));
    HIP_TRY(dev_upload(t.R.p, hS.data(), (size_t)Mpad * Mpad * sizeof(double)));
    return 0;
}

}  // namespace

extern "C" {

int mogp_svgp_forward(mogp_model* m, int64_t M, const double* Z, const double* q_mu, const double* q_sqrt, double jitter,
                      const double* kff_diag, int dense, int64_t S, const double* Xs, const double* kss_diag,
                      double* mu, double* var, double* jitter_abs, int64_t* info) {
    if (!m || !Z || !q_mu || !q_sqrt || !kff_diag || !mu || !var || M <= 0) return fail(MOGP_EINVAL, "mogp_svgp_forward: bad argument");
    if (S > 0 && (!Xs || !kss_diag)) return fail(MOGP_EINVAL, "mogp_svgp_forward: test inputs without Xs / kss_diag");
    RC(use_device(m->ctx));
    if (info) *info = 0;
    const int C = m->C, D = m->D;
    const int64_t N = m->N, Npad = m->Npad;
    const bool train = S <= 0;
    if (dense && M != N) return fail(MOGP_EINVAL, "mogp_svgp_forward: the dense model has its inducing inputs at the data points (M = N)");
    if (dense && !m->sx.identity) return fail(MOGP_EINVAL, "the dense Hensman model takes data points grouped by channel, in ascending channel order");
    SortedX sz;
    std::vector<GTile> tuu;
    std::vector<int> psuu;
    double jit = 0.0;
    if (m->tw) m->tw->sv_valid = false;
    RC(svgp_setup(m, M, Z, q_mu, q_sqrt, jitter, sz, tuu, psuu, jit, info));
    TitsiasWork& t = *m->tw;
    const int64_t Mpad = t.Mpad;
    const int mt = (int)(Mpad / MOGP_TILE);
    if (jitter_abs) *jitter_abs = jit;
    double* q = t.vec.p;

    // the points mu / var are asked at: the training inputs (state kept for the backward call) or test inputs
    SortedX ss;
    const SortedX* sp = &m->sx;
    const double* xq = m->d_x.p;
    int64_t Qpad = Npad, Qn = N;
    double* a = t.v.p;
    double* b = t.GB.p;
    std::vector<GTile> tuf;
    std::vector<int> psuf;
    if (!train) {
        t.pred_valid = false;
        RC(sort_inputs(Xs, S, D, C, MOGP_TILE, ss));
        sp = &ss; Qpad = ss.Mpad; Qn = S;
        RC(t.Kus.ensure((size_t)Mpad * Qpad)); RC(t.Aus.ensure((size_t)Mpad * Qpad)); RC(t.Bus.ensure((size_t)Mpad * Qpad));
        RC(m->d_xs.ensure((size_t)D * Qpad));
        HIP_TRY(hipMemcpyAsync(m->d_xs.p, ss.xs.data(), (size_t)D * Qpad * sizeof(double), hipMemcpyHostToDevice, m->st));
        xq = m->d_xs.p; a = t.Aus.p; b = t.Bus.p;
    }
    if (dense && train) {
        RC(launch_transpose(a, t.a.A.p, Mpad, Mpad, m->st));                      // a = L^T  (Mpad = Npad)
        hipLaunchKernelGGL(k_sv_mask_upper, dim3((unsigned)Mpad), dim3(256), 0, m->st, a, Mpad, Mpad);
        HIP_TRY(hipGetLastError());
    } else {
        build_rect_tiles(sz.off, sp->off, C, tuf, &psuf);
        DevBuf<GTile>& dt = train ? t.tiles_uf : m->d_ptiles;
        RC(dt.ensure(tuf.size()));
        t.tile_key.clear();
        HIP_TRY(hipMemcpyAsync(dt.p, tuf.data(), tuf.size() * sizeof(GTile), hipMemcpyHostToDevice, m->st));
        double* Kq = train ? t.B.p : t.Kus.p;
        HIP_TRY(hipMemsetAsync(Kq, 0, (size_t)Mpad * Qpad * sizeof(double), m->st));
        GramArgs ga{};
        ga.tiles = dt.p; ga.xr = t.zx.p; ga.ldxr = Mpad; ga.xc = xq; ga.ldxc = Qpad; ga.nrows = M; ga.ncols = Qn;
        RC((train ? t.ph_zx : t.ph_zs).prepare(sz.off, sp->off, C, m->T, Mpad, Qpad, m->st, ga.ph));
        ga.table = m->d_table.p; ga.T = m->T; ga.D = D; ga.C = C; ga.W = m->Wt; ga.out = Kq; ga.ldo = Qpad; ga.mirror = 0;
        RC(launch_gram(ga, (int)tuf.size(), m->st));
        HIP_TRY(hipMemcpyAsync(a, Kq, (size_t)Mpad * Qpad * sizeof(double), hipMemcpyDeviceToDevice, m->st));
        RC(trsm_lower(m, t.a.A.p, Mpad, mt, a, Qpad, Qpad, false));               // a = L^-1 K(Z, .)
    }
    // b = S^T a;  mu = a^T q_mu;  column sums of squares
    GemmArgs g = make_gemm(t.R.p, Mpad, 1, a, Qpad, 1, b, Qpad, 1.0, GM_RECT, mt, (int)(Qpad / MOGP_TILE), Mpad);
    RC(gemm_call(m, g, gemm_flops(g, nullptr)));
    RC(m->d_mu.ensure(Qpad)); RC(m->d_var.ensure(2 * Qpad));
    RC(launch_gemv_cols(a, Qpad, Mpad, Qpad, q, m->d_mu.p, t.scratch.p, m->st));
    RC(launch_gemv_cols(a, Qpad, Mpad, Qpad, nullptr, m->d_var.p, t.scratch.p, m->st));
    RC(launch_gemv_cols(b, Qpad, Mpad, Qpad, nullptr, m->d_var.p + Qpad, t.scratch.p, m->st));
    std::vector<double> hmu(Qpad), hv(2 * Qpad);
    HIP_TRY(hipMemcpyAsync(hmu.data(), m->d_mu.p, Qpad * sizeof(double), hipMemcpyDeviceToHost, m->st));
    HIP_TRY(hipMemcpyAsync(hv.data(), m->d_var.p, 2 * Qpad * sizeof(double), hipMemcpyDeviceToHost, m->st));
    HIP_TRY(hipStreamSynchronize(m->st));
    RC(sparse_timeout_check(m));
    const double* kd = train ? kff_diag : kss_diag;
    const bool env = m->Wt > 2 + 3 * D;                          // enveloped terms: K_diag per point (caller's order) instead of per channel
    for (int c = 0; c < C; ++c)
        for (int pos = sp->off[c]; pos < sp->off[c + 1]; ++pos) {
            mu[sp->perm[pos]] = hmu[pos];
            var[sp->perm[pos]] = (dense && train) ? hv[Qpad + pos] : (env ? kd[sp->perm[pos]] : kd[c]) - hv[pos] + hv[Qpad + pos];
        }
    if (train) {
        t.sv_sz = sz; t.sv_tuu = tuu; t.sv_psuu = psuu; t.sv_tuf = tuf; t.sv_psuf = psuf;
        t.sv_M = M; t.sv_dense = dense != 0; t.sv_valid = true;
    } else {
        t.pred_ss = ss; t.pred_valid = true;             // a, b stay in t.Aus / t.Bus for mogp_sparse_predict_cov
    }
    return MOGP_OK;
}

}  // extern "C"

// sharded: this handle holds ONE SHARD of the training points (see mogp_titsias_eval_sharded): everything that sums over data points -- the two
// M x M products over N, v e, the (Z, X) moments and their share of d/dZ -- is all-reduced; e, f are those of the local points
static int svgp_backward_impl(mogp_model* m, const double* e, const double* f, double* mom_uu, double* mom_uf, double* gZ, double* trGA,
                              double* g_qmu, double* g_qsqrt, bool sharded) {
    if (!m || !e || !f || !mom_uu || !mom_uf || !gZ || !trGA || !g_qmu || !g_qsqrt) return fail(MOGP_EINVAL, "mogp_svgp_backward: bad argument");
    RC(use_device(m->ctx));
    if (!m->tw || !m->tw->sv_valid) return fail(MOGP_EINVAL, "mogp_svgp_backward: no forward pass at the training inputs precedes it");
    TitsiasWork& t = *m->tw;
    t.sv_valid = false;                                         // the buffers of the forward pass are consumed
    const int C = m->C, D = m->D, W = m->Wt, T = m->T, P = C * (C + 1) / 2;
    const int64_t N = m->N, Npad = m->Npad, Mpad = t.Mpad, M = t.sv_M;
    if (sharded && W > 2 + 3 * D) return fail(MOGP_EINVAL, "the data-parallel Hensman path does not take terms with an envelope (MOHSM)");
    const int mt = (int)(Mpad / MOGP_TILE), nt = (int)(Npad / MOGP_TILE);
    const bool dense = t.sv_dense;
    const SortedX& sz = t.sv_sz;
    RC(t.ps_uu.ensure(t.sv_psuu.size()));
    HIP_TRY(hipMemcpyAsync(t.ps_uu.p, t.sv_psuu.data(), t.sv_psuu.size() * sizeof(int), hipMemcpyHostToDevice, m->st));
    RC(t.partial_uu.ensure(t.sv_tuu.size() * (size_t)T * W)); RC(t.mom_uu.ensure((size_t)P * T * W));
    RC(t.mom_uf.ensure((size_t)C * C * T * W));
    RC(t.gz.ensure((size_t)D * Mpad));
    if (!dense) {
        RC(t.ps_uf.ensure(t.sv_psuf.size()));
        HIP_TRY(hipMemcpyAsync(t.ps_uf.p, t.sv_psuf.data(), t.sv_psuf.size() * sizeof(int), hipMemcpyHostToDevice, m->st));
        RC(t.partial_uf.ensure(t.sv_tuf.size() * (size_t)T * W));
    }
    // e, f in the device's (channel-sorted) order of the points
    std::vector<double> he(Npad, 0.0), hf(Npad, 0.0);
    for (int64_t pos = 0; pos < N; ++pos) { he[pos] = e[m->sx.perm[pos]]; hf[pos] = f[m->sx.perm[pos]]; }
    double* de = t.nvec.p;
    double* df = t.nvec.p + Npad;
    HIP_TRY(hipMemcpyAsync(de, he.data(), Npad * sizeof(double), hipMemcpyHostToDevice, m->st));
    HIP_TRY(hipMemcpyAsync(df, hf.data(), Npad * sizeof(double), hipMemcpyHostToDevice, m->st));
    double* q = t.vec.p;
    double* gq = t.vec.p + Mpad;
    double* dga = t.vec.p + 2 * Mpad;
    const dim3 gmn((unsigned)((Npad + 255) / 256), (unsigned)Mpad);
    // dE/dq_mu = v e
    RC(launch_gemv_rows(t.v.p, Npad, Mpad, Npad, de, gq, m->st));
    if (sharded) {
        if (dense) return fail(MOGP_EINVAL, "the dense Hensman model lives on ALL data points: it cannot be sharded over them");
        RC(comm_allreduce(m->ctx, gq, Mpad, m->st));
    }
    // Gv = q e^T + 2 (S b - v) diag f   (t.B; b = S^T v is in t.GB from the forward pass)
    GemmArgs g = make_gemm(t.R.p, Mpad, 0, t.GB.p, Npad, 1, t.B.p, Npad, 1.0, GM_RECT, mt, nt, Mpad);
    RC(gemm_call(m, g, gemm_flops(g, nullptr)));
    hipLaunchKernelGGL(k_sv_adjoint, gmn, dim3(256), 0, m->st, t.B.p, t.v.p, Npad, Npad, q, de, df, dense ? 0.0 : 1.0);
    HIP_TRY(hipGetLastError());
    // Psi: tril(Gv v^T) (dense: tril(v Gv^T)) mirrored;  GA = -/+ 1/2 L^-T Psi L^-1
    RC(mm_lower_splitk(m, t, dense ? t.v.p : t.B.p, dense ? t.B.p : t.v.p, t.E.p, mt, Mpad, Npad, Npad));
    if (sharded) RC(comm_allreduce(m->ctx, t.E.p, Mpad * Mpad, m->st));
    RC(launch_symmetrize(t.E.p, Mpad, Mpad, m->st));
    hipStream_t side;                              // the two M x M solves: on the side stream, underneath the M x N work below
    RC(side_fork(m, t, &side));
    RC(trsm_lower(m, t.a.A.p, Mpad, mt, t.E.p, Mpad, Mpad, true, side));
    RC(launch_transpose(t.GA.p, t.E.p, Mpad, Mpad, side));
    RC(trsm_lower(m, t.a.A.p, Mpad, mt, t.GA.p, Mpad, Mpad, true, side));
    RC(launch_sym_lower_avg(t.GA.p, Mpad, Mpad, dense ? 0.5 : -0.5, side));
    RC(launch_get_diag(t.GA.p, Mpad, Mpad, dga, side));
    // dE/dS = 2 (v diag(f) v^T) S: (v f) into t.GB (b is no longer needed), the M x M product into t.Qs, times S into t.q.A
    hipLaunchKernelGGL(k_sv_scale_cols, gmn, dim3(256), 0, m->st, t.v.p, t.GB.p, Npad, Npad, df);
    HIP_TRY(hipGetLastError());
    RC(mm_lower_splitk(m, t, t.GB.p, t.v.p, t.Qs.p, mt, Mpad, Npad, Npad));
    if (sharded) RC(comm_allreduce(m->ctx, t.Qs.p, Mpad * Mpad, m->st));
    RC(launch_symmetrize(t.Qs.p, Mpad, Mpad, m->st));
    g = make_gemm(t.Qs.p, Mpad, 0, t.R.p, Mpad, 1, t.q.A.p, Mpad, 2.0, GM_RECT, mt, mt, Mpad);
    RC(gemm_call(m, g, gemm_flops(g, nullptr)));
    HIP_TRY(hipMemsetAsync(t.gz.p, 0, (size_t)D * Mpad * sizeof(double), m->st));
    RC(gz_prepare(m, t, sz.off, D));

    MomentArgs ma{};
    ma.x = t.zx.p; ma.ldx = Mpad; ma.nrows = M;
    ma.table = m->d_table.p; ma.T = T; ma.D = D; ma.C = C; ma.W = W;
    ma.ru = q; ma.rw = q; ma.rcoef = 0.0; ma.ldgz = Mpad;
    if (!dense) {
        // dE/dKuf = L^-T Gv, in place
        RC(trsm_lower(m, t.a.A.p, Mpad, mt, t.B.p, Npad, Npad, true));
        ma.tiles = t.tiles_uf.p; ma.ntiles = (int)t.sv_tuf.size(); ma.xc = m->d_x.p; ma.ldxc = Npad; ma.ncols = N;
        RC(t.ph_zx.prepare(sz.off, m->sx.off, C, T, Mpad, Npad, m->st, ma.ph));
        ma.G = t.B.p; ma.ldg = Npad; ma.rw = de; ma.sym = 0;
        ma.gzr = t.gz.p; ma.gzc = nullptr; ma.partial = t.partial_uf.p;
        gz_attach(t, ma, true);
        RC(launch_moments(ma, m->st));
        RC(launch_moment_reduce(t.partial_uf.p, t.ps_uf.p, C * C, T, W, D, t.mom_uf.p, m->st, 0));
        if (sharded) {
            RC(comm_allreduce(m->ctx, t.mom_uf.p, (int64_t)C * C * T * W, m->st));
            RC(comm_allreduce(m->ctx, t.gz.p, (int64_t)D * Mpad, m->st));
        }
    } else {
        HIP_TRY(hipMemsetAsync(t.mom_uf.p, 0, (size_t)C * C * T * W * sizeof(double), m->st));
    }
    RC(side_join(m, t, side));
    ma.tiles = t.tiles_uu.p; ma.ntiles = (int)t.sv_tuu.size(); ma.xc = nullptr; ma.ldxc = 0; ma.ncols = M;
    RC(t.ph_zz.prepare(sz.off, sz.off, C, T, Mpad, Mpad, m->st, ma.ph));
    ma.G = t.GA.p; ma.ldg = Mpad; ma.ru = q; ma.rw = q; ma.rcoef = 0.0; ma.sym = 1;
    ma.gzr = dense ? nullptr : t.gz.p; ma.gzc = dense ? nullptr : t.gz.p; ma.partial = t.partial_uu.p;      // dense: the inputs are the data, not parameters
    gz_attach(t, ma, false);
    RC(launch_moments(ma, m->st));
    RC(launch_moment_reduce(t.partial_uu.p, t.ps_uu.p, P, T, W, D, t.mom_uu.p, m->st, 1));

    std::vector<double> hgz((size_t)D * Mpad), hq(Mpad), hd(Mpad), hS((size_t)Mpad * Mpad);
    HIP_TRY(hipMemcpyAsync(mom_uu, t.mom_uu.p, (size_t)P * T * W * sizeof(double), hipMemcpyDeviceToHost, m->st));
    HIP_TRY(hipMemcpyAsync(mom_uf, t.mom_uf.p, (size_t)C * C * T * W * sizeof(double), hipMemcpyDeviceToHost, m->st));
    HIP_TRY(hipMemcpyAsync(hgz.data(), t.gz.p, hgz.size() * sizeof(double), hipMemcpyDeviceToHost, m->st));
    HIP_TRY(hipMemcpyAsync(hq.data(), gq, Mpad * sizeof(double), hipMemcpyDeviceToHost, m->st));
    HIP_TRY(hipMemcpyAsync(hd.data(), dga, Mpad * sizeof(double), hipMemcpyDeviceToHost, m->st));
    HIP_TRY(hipMemcpyAsync(hS.data(), t.q.A.p, hS.size() * sizeof(double), hipMemcpyDeviceToHost, m->st));
    HIP_TRY(hipStreamSynchronize(m->st));
    RC(sparse_timeout_check(m));
    double tr = 0.0;
    for (int64_t pos = 0; pos < M; ++pos) {
        const int64_t dst = sz.perm[pos];
        for (int d = 0; d < D; ++d) gZ[dst * D + d] = hgz[(size_t)d * Mpad + pos];
        g_qmu[dst] = hq[pos];
        for (int64_t c = 0; c < M; ++c) g_qsqrt[(size_t)dst * M + c] = hS[(size_t)pos * Mpad + c];
        tr += hd[pos];
    }
    *trGA = tr;
    return MOGP_OK;
}

extern "C" {

int mogp_svgp_backward(mogp_model* m, const double* e, const double* f, double* mom_uu, double* mom_uf, double* gZ, double* trGA,
                       double* g_qmu, double* g_qsqrt) {
    return svgp_backward_impl(m, e, f, mom_uu, mom_uf, gZ, trGA, g_qmu, g_qsqrt, false);
}

int mogp_svgp_backward_sharded(mogp_model* m, const double* e, const double* f, double* mom_uu, double* mom_uf, double* gZ, double* trGA,
                               double* g_qmu, double* g_qsqrt) {
    return svgp_backward_impl(m, e, f, mom_uu, mom_uf, gZ, trGA, g_qmu, g_qsqrt, true);
}

}  // extern "C"
