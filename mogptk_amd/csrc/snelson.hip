// snelson.hip -- the pseudo-input (FITC) sparse GP of Snelson & Ghahramani and its gradient on the device.
// Reference: gpr/model.py:516-541 (log_marginal_likelihood), :543-576 (predict_f); the gradient replaces autograd through that code.
//
//   A = Kuu + jitter mean(diag Kuu) I = L L^T,  B = Kuf,  v = L^-1 B,  q_n = sum_m v_mn^2 (= Qff_nn),
//   g_n = Kff_nn - q_n + sigma_c(n)^2,  G = diag 1/g,  Bq = I + v G v^T = Lq Lq^T,  Pq = Bq^-1,  r = Pq (v G y)
//   p = -N/2 log 2pi - sum log Lq_kk - 1/2 sum log g_n - 1/2 sum y_n^2 / g_n + 1/2 r.(v G y)        (= log N(y | 0, Qff + diag g))
// Adjoints (S = Qff + diag g;  alpha = S^-1 y = G (y - v^T r);  [S^-1]_nn = G_n - G_n^2 sum_m v_mn (Pq v)_mn;  h_n = 1/2 (alpha_n^2 - [S^-1]_nn)):
//   dp/dB = L^-T (r alpha^T - Pq v G - 2 v diag h)                       (the diagonal of Qff enters g with a minus sign: the -2 v diag h)
//   dp/dA = 1/2 L^-T (I - Pq + 2 v diag(h) v^T) L^-1 - 1/2 beta beta^T,  beta = L^-T r
//   dp/dKff_nn = dp/dsigma_n^2 = h_n
// Same building blocks as the Titsias bound (titsias.hip): solves with L by blocked substitution (trsm.hip), the inner M x M system inverted
// explicitly, the two adjoints contracted with the kernel derivatives by the dense-mode moment kernel (which also yields d/dZ).
// Checked against the reference's autograd through the numpy twin (oracle/table_model.py:snelson_eval) and on the device (snelson.npz).
#include "mogp_model.h"

#include <cmath>
#include <cstring>
#include <limits>

using namespace mogp;

#define RC(x) do { int r__ = (x); if (r__) return r__; } while (0)

namespace {



// per point (channel-sorted order): g = Kff_diag[c] - q + s2[c], G = 1/g, Gy = G y, sg = sqrt(G); zero on the padding.  kd_point (terms with
// an envelope, MOHSM): the kernel diagonal follows the points -- Kff_diag per point instead of per channel
__global__ void k_sn_point(const double* __restrict__ q, const double* __restrict__ y, const int* __restrict__ off, int C,
                           const double* __restrict__ kd, const double* __restrict__ kd_point, const double* __restrict__ s2, int64_t N, int64_t Npad,
                           double* __restrict__ g, double* __restrict__ G, double* __restrict__ Gy, double* __restrict__ sg) {
    const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= Npad) return;
    if (n >= N) { g[n] = 1.0; G[n] = 0.0; Gy[n] = 0.0; sg[n] = 0.0; return; }
    int c = 0;
    while (c + 1 < C && n >= off[c + 1]) ++c;
    const double gv = (kd_point ? kd_point[n] : kd[c]) - q[n] + s2[c];
    g[n] = gv;
    const double Gv = 1.0 / gv;
    G[n] = Gv; Gy[n] = Gv * y[n]; sg[n] = sqrt(fabs(Gv));
}
// out[m][n] = in[m][n] * s[n]
__global__ void k_scale_cols(const double* __restrict__ in, double* __restrict__ out, int64_t ld, int64_t n, const double* __restrict__ s) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const int64_t r = blockIdx.y;
    out[r * ld + j] = in[r * ld + j] * s[j];
}
// w[n] = sum_m a[m][n] b[m][n] over `rows` rows (row chunks of 256 into part, then summed in order)
__global__ __launch_bounds__(256) void k_coldot_part(const double* __restrict__ a, const double* __restrict__ b, int64_t ld, int64_t rows, int64_t n,
                                                      double* __restrict__ part) {
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= n) return;
    const int64_t r0 = (int64_t)blockIdx.y * 256, r1 = min(rows, r0 + 256);
    double s = 0.0;
    for (int64_t i = r0; i < r1; ++i) s = fma(a[i * ld + j], b[i * ld + j], s);
    part[(int64_t)blockIdx.y * n + j] = s;
}
__global__ void k_sum_parts_n(const double* __restrict__ part, int64_t n, int nparts, double* __restrict__ out) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    double s = 0.0;
    for (int k = 0; k < nparts; ++k) s += part[(int64_t)k * n + j];
    out[j] = s;
}
// alpha = G (y - v^T r);  h = 1/2 (alpha^2 - G + G^2 w)   (zero on the padding: G = 0 there)
__global__ void k_sn_alpha(const double* __restrict__ G, const double* __restrict__ y, const double* __restrict__ vtr, int64_t N, int64_t Npad,
                           double* __restrict__ alpha) {
    const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= Npad) return;
    alpha[n] = n < N ? G[n] * (y[n] - vtr[n]) : 0.0;
}
__global__ void k_sn_h(const double* __restrict__ G, const double* __restrict__ alpha, const double* __restrict__ w, int64_t N, int64_t Npad,
                       double* __restrict__ h) {
    const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= Npad) return;
    h[n] = n < N ? 0.5 * (alpha[n] * alpha[n] - G[n] + G[n] * G[n] * w[n]) : 0.0;
}
// T[m][n] = r[m] alpha[n] - R1[m][n] G[n] - 2 v[m][n] h[n]   (in place over R1)
__global__ void k_sn_adjoint(double* __restrict__ T, const double* __restrict__ v, int64_t ld, int64_t n, const double* __restrict__ r,
                             const double* __restrict__ alpha, const double* __restrict__ G, const double* __restrict__ h) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const int64_t i = blockIdx.y;
    T[i * ld + j] = r[i] * alpha[j] - T[i * ld + j] * G[j] - 2.0 * v[i * ld + j] * h[j];
}

struct SnScalars { double logdet_q, sumlogg, yGy, rvGy, jit, ntot; };


// Front end shared by the evaluation and the prediction: everything up to r = Pq (v G y) and the scalars of p.
// On return: t.a = L (Kuu), t.v = v, t.Wq = Lq^-1, t.q.B = Pq (full), t.Qs = Bq (full), t.nvec = [g | G | Gy | sqrt G | ...],
// t.vec[0:Mpad] = v G y, t.vec[Mpad:2Mpad] = r.
// sharded: this handle holds ONE SHARD of the training points (cf. mogp_titsias_eval_sharded): v G v^T, v G y, sum log g, y^T G y and N are
// all-reduced (and the per-point positivity check with them, so that every rank takes the same exit).
int snelson_front(mogp_model* m, int64_t M, const double* Z, const double* noise_var, double jitter, const double* kff_diag, SortedX& sz,
                  std::vector<GTile>& tuu, std::vector<int>& psuu, std::vector<GTile>& tuf, std::vector<int>& psuf, SnScalars& sc,
                  int64_t* info, bool need_moment_tiles, bool sharded) {
    const int C = m->C, D = m->D, W = m->Wt;                   // 2 + 3 D, or 2 + 5 D: terms with an envelope on the input midpoint (MOHSM)
    const bool env = W > 2 + 3 * D;
    const int64_t N = m->N, Npad = m->Npad;
    if (m->T <= 0) return fail(MOGP_EINVAL, "mogp_model_set_terms must be called before an evaluation");
    if (env && sharded) return fail(MOGP_EINVAL, "the data-parallel Snelson path does not take terms with an envelope (MOHSM)");
    for (int c = 0; c < C; ++c) if (!(noise_var[c] > 0.0)) return fail(MOGP_EINVAL, "noise variances must be positive");
    RC(sort_inputs(Z, M, D, C, MOGP_TILE, sz));
    const int64_t Mpad = sz.Mpad;
    if (!m->tw) m->tw = new TitsiasWork();
    TitsiasWork& t = *m->tw;
    const int mt = (int)(Mpad / MOGP_TILE);
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
    RC(t.nvec.ensure((size_t)8 * Npad + 2 * C));
    // Kuf's padding (rows >= M, columns >= N) must be zero and this path overwrites t.B with scaled copies of v: cleared per call
    HIP_TRY(hipMemsetAsync(t.B.p, 0, (size_t)Mpad * Npad * sizeof(double), m->st));
    m->gemm_ev_used = 0; m->gemm_launches = 0; m->gemm_flops = 0.0;
    build_sym_tiles(sz.off, C, tuu, psuu);
    build_rect_tiles(sz.off, m->sx.off, C, tuf, &psuf);
    t.tile_key.clear();                                     // (the Titsias path keeps its lists in these buffers between evaluations)
    RC(t.tiles_uu.ensure(tuu.size())); RC(t.tiles_uf.ensure(tuf.size()));
    HIP_TRY(hipMemcpyAsync(t.zx.p, sz.xs.data(), (size_t)D * Mpad * sizeof(double), hipMemcpyHostToDevice, m->st));
    HIP_TRY(hipMemcpyAsync(t.tiles_uu.p, tuu.data(), tuu.size() * sizeof(GTile), hipMemcpyHostToDevice, m->st));
    HIP_TRY(hipMemcpyAsync(t.tiles_uf.p, tuf.data(), tuf.size() * sizeof(GTile), hipMemcpyHostToDevice, m->st));
    if (need_moment_tiles) {
        RC(t.ps_uu.ensure(psuu.size())); RC(t.ps_uf.ensure(psuf.size()));
        HIP_TRY(hipMemcpyAsync(t.ps_uu.p, psuu.data(), psuu.size() * sizeof(int), hipMemcpyHostToDevice, m->st));
        HIP_TRY(hipMemcpyAsync(t.ps_uf.p, psuf.data(), psuf.size() * sizeof(int), hipMemcpyHostToDevice, m->st));
        RC(t.partial_uu.ensure(tuu.size() * (size_t)m->T * W)); RC(t.partial_uf.ensure(tuf.size() * (size_t)m->T * W));
        RC(t.mom_uu.ensure((size_t)(C * (C + 1) / 2) * m->T * W)); RC(t.mom_uf.ensure((size_t)C * C * m->T * W));
    }
    const unsigned long long big = std::numeric_limits<unsigned long long>::max();
    HIP_TRY(hipMemcpyAsync(m->d_info.p, &big, sizeof(big), hipMemcpyHostToDevice, m->st));

    sc.jit = jitter * table_diag_points(m, sz) / (double)M;      // relative jitter on Kuu (reference gpr/model.py:524 -> :244); with an envelope the diagonal follows Z

    GramArgs ga{};
    ga.tiles = t.tiles_uu.p; ga.xr = t.zx.p; ga.xc = t.zx.p; ga.ldxr = ga.ldxc = Mpad; ga.nrows = ga.ncols = M;
    RC(t.ph_zz.prepare(sz.off, sz.off, C, m->T, Mpad, Mpad, m->st, ga.ph));
    ga.table = m->d_table.p; ga.T = m->T; ga.D = D; ga.C = C; ga.W = W; ga.out = t.a.A.p; ga.ldo = Mpad;
    ga.noise = t.zero_noise.p; ga.dvar = nullptr; ga.jitter_abs = sc.jit; ga.mirror = 0;
    RC(launch_gram(ga, (int)tuu.size(), m->st));
    RC(launch_pad_identity(t.a.A.p, Mpad, M, Mpad, m->st));
    ga.tiles = t.tiles_uf.p; ga.xc = m->d_x.p; ga.ldxc = Npad; ga.ncols = m->N; ga.out = t.B.p; ga.ldo = Npad; ga.noise = nullptr; ga.jitter_abs = 0.0;
    RC(t.ph_zx.prepare(sz.off, m->sx.off, C, m->T, Mpad, Npad, m->st, ga.ph));
    RC(launch_gram(ga, (int)tuf.size(), m->st));

    t.a.keep_L = true;
    t.a.refine_panels = !(std::getenv("MOGP_REFINE_PANELS") && std::atoi(std::getenv("MOGP_REFINE_PANELS")) == 0);   // K_uu + jitter is ill-conditioned: mogp_api.hip:spd_potrf
    RC(spd_potrf(m, t.a));
    RC(spd_check_info(m, "Kuu", info));
    HIP_TRY(hipMemcpyAsync(t.v.p, t.B.p, (size_t)Mpad * Npad * sizeof(double), hipMemcpyDeviceToDevice, m->st));
    RC(trsm_lower(m, t.a.A.p, Mpad, mt, t.v.p, Npad, Npad, false));              // v = L^-1 B   (reference gpr/model.py:525)

    // per point: g, G, G y, sqrt G
    double* g = t.nvec.p;
    double* G = g + Npad;
    double* Gy = G + Npad;
    double* sg = Gy + Npad;
    double* q = sg + Npad;                                                       // Qff_nn, then reused
    double* kd = t.nvec.p + 8 * Npad;
    double* s2 = kd + C;
    double* kd_point = nullptr;                                                  // envelope: Kff_diag per training point (caller order in, sorted order here)
    std::vector<double> hkd;
    if (env) {
        RC(t.kd_point.ensure((size_t)Npad));
        hkd.assign((size_t)Npad, 0.0);
        for (int64_t pos = 0; pos < N; ++pos) hkd[pos] = kff_diag[m->sx.perm[pos]];
        HIP_TRY(hipMemcpyAsync(t.kd_point.p, hkd.data(), Npad * sizeof(double), hipMemcpyHostToDevice, m->st));
        kd_point = t.kd_point.p;
    } else {
        HIP_TRY(hipMemcpyAsync(kd, kff_diag, C * sizeof(double), hipMemcpyHostToDevice, m->st));
    }
    HIP_TRY(hipMemcpyAsync(s2, noise_var, C * sizeof(double), hipMemcpyHostToDevice, m->st));
    RC(launch_gemv_cols(t.v.p, Npad, Mpad, Npad, nullptr, q, t.scratch.p, m->st));
    hipLaunchKernelGGL(k_sn_point, dim3((unsigned)((Npad + 255) / 256)), dim3(256), 0, m->st, q, m->d_y.p, m->d_chan_off.p, C, kd, kd_point, s2, N, Npad, g, G, Gy, sg);
    HIP_TRY(hipGetLastError());
    std::vector<double> hg(Npad);
    HIP_TRY(hipMemcpyAsync(hg.data(), g, Npad * sizeof(double), hipMemcpyDeviceToHost, m->st));
    // Bq = I + (v sqrt G)(v sqrt G)^T
    hipLaunchKernelGGL(k_scale_cols, dim3((unsigned)((Npad + 255) / 256), (unsigned)Mpad), dim3(256), 0, m->st, t.v.p, t.B.p, Npad, Npad, sg);
    HIP_TRY(hipGetLastError());
    RC(mm_lower_splitk(m, t, t.B.p, t.B.p, t.q.A.p, mt, Mpad, Npad, Npad));
    double* vGy = t.vec.p;
    RC(launch_gemv_rows(t.v.p, Npad, Mpad, Npad, Gy, vGy, m->st));
    HIP_TRY(hipStreamSynchronize(m->st));
    int64_t bad = -1;
    sc.sumlogg = sc.yGy = 0.0;
    for (int64_t n = 0; n < N; ++n) {
        if (!(hg[n] > 0.0)) { if (bad < 0) bad = n; continue; }
        sc.sumlogg += std::log(hg[n]); sc.yGy += m->hy[n] * m->hy[n] / hg[n];
    }
    sc.ntot = (double)N;
    double nbad = bad >= 0 ? 1.0 : 0.0;
    if (sharded) {
        RC(t.red.ensure((size_t)Mpad + 4));
        double hs[4] = {sc.sumlogg, sc.yGy, sc.ntot, nbad};
        HIP_TRY(hipMemcpyAsync(t.red.p, vGy, Mpad * sizeof(double), hipMemcpyDeviceToDevice, m->st));
        HIP_TRY(hipMemcpyAsync(t.red.p + Mpad, hs, sizeof(hs), hipMemcpyHostToDevice, m->st));
        HIP_TRY(hipStreamSynchronize(m->st));
        RC(comm_allreduce(m->ctx, t.red.p, Mpad + 4, m->st));
        HIP_TRY(hipMemcpyAsync(vGy, t.red.p, Mpad * sizeof(double), hipMemcpyDeviceToDevice, m->st));
        HIP_TRY(hipMemcpyAsync(hs, t.red.p + Mpad, sizeof(hs), hipMemcpyDeviceToHost, m->st));
        HIP_TRY(hipStreamSynchronize(m->st));
        sc.sumlogg = hs[0]; sc.yGy = hs[1]; sc.ntot = hs[2]; nbad = hs[3];
    }
    if (nbad > 0.0)
        return fail(MOGP_ENOTPD, bad >= 0 ? "Snelson: Kff - Qff + sigma^2 has a non-positive entry (point " + std::to_string(bad) + " in channel-sorted order)"
                                          : std::string("Snelson: Kff - Qff + sigma^2 has a non-positive entry (on another rank's shard)"));
    if (sharded) RC(comm_allreduce(m->ctx, t.q.A.p, Mpad * Mpad, m->st));
    RC(launch_add_diag(t.q.A.p, Mpad, Mpad, 1.0, m->st));
    HIP_TRY(hipMemcpyAsync(t.Qs.p, t.q.A.p, (size_t)Mpad * Mpad * sizeof(double), hipMemcpyDeviceToDevice, m->st));
    RC(launch_info_rearm(m->d_info.p, m->st));              // (not a plain overwrite: a time-out of the wide solve above must reach the host)
    RC(spd_invert(m, t.q, "v G v^T + I", info, &t.Wq));                         // t.Wq = Lq^-1, t.q.B = Pq (lower)
    RC(launch_symmetrize(t.q.B.p, Mpad, Mpad, m->st));
    RC(launch_symmetrize(t.Qs.p, Mpad, Mpad, m->st));
    double* r = t.vec.p + Mpad;
    {   // r = Pq (v G y): the explicit inverse plus one step of iterative refinement against Bq (see titsias.hip)
        double* tmp = t.vec.p + 5 * Mpad;
        double* res = t.vec.p + 6 * Mpad;
        RC(launch_gemv_rows(t.q.B.p, Mpad, Mpad, Mpad, vGy, r, m->st));
        RC(launch_gemv_rows(t.Qs.p, Mpad, Mpad, Mpad, r, tmp, m->st));
        RC(launch_axpby(Mpad, 1.0, vGy, -1.0, tmp, res, m->st));
        RC(launch_gemv_rows(t.q.B.p, Mpad, Mpad, Mpad, res, tmp, m->st));
        RC(launch_axpby(Mpad, 1.0, r, 1.0, tmp, r, m->st));
    }
    const int nbq = t.q.nb;
    std::vector<double> hv((size_t)2 * Mpad), hl(nbq);
    HIP_TRY(hipMemcpyAsync(hv.data(), t.vec.p, (size_t)2 * Mpad * sizeof(double), hipMemcpyDeviceToHost, m->st));
    HIP_TRY(hipMemcpyAsync(hl.data(), t.q.logdet.p, nbq * sizeof(double), hipMemcpyDeviceToHost, m->st));
    HIP_TRY(hipStreamSynchronize(m->st));
    sc.logdet_q = 0.0; for (double x : hl) sc.logdet_q += x;
    sc.rvGy = 0.0;
    for (int64_t i = 0; i < M; ++i) sc.rvGy += hv[Mpad + i] * hv[i];
    return 0;
}

int snelson_eval_impl(mogp_model* m, int64_t M, const double* Z, const double* noise_var, double jitter, const double* kff_diag, int flags,
                      double* lml, double* mom_uu, double* mom_uf, double* gZ, double* trGA, double* hsum, double* jitter_abs, int64_t* info,
                      bool sharded) {
    if (!m || !Z || !noise_var || !kff_diag || !lml || M <= 0) return fail(MOGP_EINVAL, "mogp_snelson_eval: bad argument");
    RC(use_device(m->ctx));
    if (info) *info = 0;
    const int C = m->C, D = m->D, W = m->Wt, T = m->T, P = C * (C + 1) / 2;
    const bool env = W > 2 + 3 * D;
    const int64_t N = m->N, Npad = m->Npad;
    const bool grad = (flags & MOGP_EVAL_GRAD) != 0;
    SortedX sz;
    std::vector<GTile> tuu, tuf;
    std::vector<int> psuu, psuf;
    SnScalars sc;
    RC(snelson_front(m, M, Z, noise_var, jitter, kff_diag, sz, tuu, psuu, tuf, psuf, sc, info, grad, sharded));
    TitsiasWork& t = *m->tw;
    const int64_t Mpad = t.Mpad;
    const int mt = (int)(Mpad / MOGP_TILE), nt = (int)(Npad / MOGP_TILE);
    *lml = -0.5 * sc.ntot * std::log(2.0 * M_PI) - sc.logdet_q - 0.5 * sc.sumlogg - 0.5 * sc.yGy + 0.5 * sc.rvGy;
    if (jitter_abs) *jitter_abs = sc.jit;
    if (!grad) return sparse_timeout_check(m);
    if (!mom_uu || !mom_uf || !gZ || !trGA || !hsum) return fail(MOGP_EINVAL, "mogp_snelson_eval: gradient outputs are null");

    RC(t.GB.ensure((size_t)Mpad * Npad)); RC(t.E.ensure((size_t)Mpad * Mpad)); RC(t.R.ensure((size_t)Mpad * Mpad));
    RC(t.GA.ensure((size_t)Mpad * Mpad));
    RC(t.gz.ensure((size_t)D * Mpad));
    if (t.zero_col.n < (size_t)Mpad) { RC(t.zero_col.ensure(Mpad)); HIP_TRY(hipMemsetAsync(t.zero_col.p, 0, Mpad * sizeof(double), m->st)); }
    double* r = t.vec.p + Mpad;
    double* beta = t.vec.p + 4 * Mpad;
    double* dga = t.vec.p + 2 * Mpad;
    double* G = t.nvec.p + Npad;
    double* vtr = t.nvec.p + 4 * Npad;            // v^T r, then w = coldot(v, Pq v)
    double* alpha = t.nvec.p + 5 * Npad;
    double* h = t.nvec.p + 6 * Npad;
    const dim3 gn((unsigned)((Npad + 255) / 256)), gmn((unsigned)((Npad + 255) / 256), (unsigned)Mpad);
    RC(launch_gemv_cols(t.v.p, Npad, Mpad, Npad, r, vtr, t.scratch.p, m->st));
    hipLaunchKernelGGL(k_sn_alpha, gn, dim3(256), 0, m->st, G, m->d_y.p, vtr, N, Npad, alpha);
    // R1 = Pq v
    GemmArgs g = make_gemm(t.q.B.p, Mpad, 0, t.v.p, Npad, 1, t.GB.p, Npad, 1.0, GM_RECT, mt, nt, Mpad);
    RC(gemm_call(m, g, gemm_flops(g, nullptr)));
    {
        const int nparts = (int)((Mpad + 255) / 256);
        hipLaunchKernelGGL(k_coldot_part, dim3((unsigned)((Npad + 255) / 256), (unsigned)nparts), dim3(256), 0, m->st, t.v.p, t.GB.p, Npad, Mpad, Npad, t.scratch.p);
        hipLaunchKernelGGL(k_sum_parts_n, gn, dim3(256), 0, m->st, t.scratch.p, Npad, nparts, vtr);
    }
    hipLaunchKernelGGL(k_sn_h, gn, dim3(256), 0, m->st, G, alpha, vtr, N, Npad, h);
    hipLaunchKernelGGL(k_sn_adjoint, gmn, dim3(256), 0, m->st, t.GB.p, t.v.p, Npad, Npad, r, alpha, G, h);
    HIP_TRY(hipGetLastError());
    // E = I - Pq + 2 (v diag h) v^T;  GA = 1/2 L^-T E L^-1 (the - 1/2 beta beta^T goes through the moment kernel's rank-one term).  The M x M x N
    // product first (it needs the whole chip); the two M x M solves behind it go to the side stream, underneath the M x N solve below
    hipLaunchKernelGGL(k_scale_cols, gmn, dim3(256), 0, m->st, t.v.p, t.B.p, Npad, Npad, h);
    HIP_TRY(hipGetLastError());
    RC(mm_lower_splitk(m, t, t.B.p, t.v.p, t.R.p, mt, Mpad, Npad, Npad));
    if (sharded) RC(comm_allreduce(m->ctx, t.R.p, Mpad * Mpad, m->st));
    RC(launch_symmetrize(t.R.p, Mpad, Mpad, m->st));
    hipStream_t side;
    RC(side_fork(m, t, &side));
    RC(launch_combine(t.E.p, t.q.B.p, t.R.p, Mpad, Mpad, 1.0, 1.0, -2.0, side));
    RC(trsm_lower(m, t.a.A.p, Mpad, mt, t.E.p, Mpad, Mpad, true, side));
    RC(launch_transpose(t.GA.p, t.E.p, Mpad, Mpad, side));
    RC(trsm_lower(m, t.a.A.p, Mpad, mt, t.GA.p, Mpad, Mpad, true, side));
    RC(launch_sym_lower_avg(t.GA.p, Mpad, Mpad, 0.5, side));
    RC(launch_get_diag(t.GA.p, Mpad, Mpad, dga, side));
    // GB = L^-T T, with beta = L^-T r riding along (padding column, or a panel of its own)
    const bool ride = Npad > N;
    if (ride) RC(launch_copy2d(t.GB.p + N, Npad, r, 1, Mpad, 1, 1.0, m->st));
    RC(trsm_lower(m, t.a.A.p, Mpad, mt, t.GB.p, Npad, Npad, true));
    if (ride) {
        RC(launch_copy2d(beta, 1, t.GB.p + N, Npad, Mpad, 1, 1.0, m->st));
        RC(launch_copy2d(t.GB.p + N, Npad, t.zero_col.p, 1, Mpad, 1, 1.0, m->st));
    } else {
        RC(t.Hm.ensure((size_t)Mpad * MOGP_TILE));
        HIP_TRY(hipMemsetAsync(t.Hm.p, 0, (size_t)Mpad * MOGP_TILE * sizeof(double), m->st));
        RC(launch_copy2d(t.Hm.p, MOGP_TILE, r, 1, Mpad, 1, 1.0, m->st));
        RC(trsm_lower(m, t.a.A.p, Mpad, mt, t.Hm.p, MOGP_TILE, MOGP_TILE, true));
        RC(launch_copy2d(beta, 1, t.Hm.p, MOGP_TILE, Mpad, 1, 1.0, m->st));
    }
    HIP_TRY(hipMemsetAsync(t.gz.p, 0, (size_t)D * Mpad * sizeof(double), m->st));
    RC(gz_prepare(m, t, sz.off, D));

    MomentArgs ma{};
    ma.tiles = t.tiles_uf.p; ma.ntiles = (int)tuf.size(); ma.x = t.zx.p; ma.ldx = Mpad; ma.xc = m->d_x.p; ma.ldxc = Npad;
    ma.nrows = M; ma.ncols = N;
    RC(t.ph_zx.prepare(sz.off, m->sx.off, C, T, Mpad, Npad, m->st, ma.ph));
    ma.table = m->d_table.p; ma.T = T; ma.D = D; ma.C = C; ma.W = W;
    ma.G = t.GB.p; ma.ldg = Npad; ma.ru = beta; ma.rw = alpha; ma.rcoef = 0.0; ma.sym = 0;
    ma.gzr = t.gz.p; ma.gzc = nullptr; ma.ldgz = Mpad; ma.partial = t.partial_uf.p;
    gz_attach(t, ma, true);
    RC(launch_moments(ma, m->st));
    RC(launch_moment_reduce(t.partial_uf.p, t.ps_uf.p, C * C, T, W, D, t.mom_uf.p, m->st, 0));
    if (sharded) {
        RC(comm_allreduce(m->ctx, t.mom_uf.p, (int64_t)C * C * T * W, m->st));
        RC(comm_allreduce(m->ctx, t.gz.p, (int64_t)D * Mpad, m->st));
    }
    RC(side_join(m, t, side));
    ma.tiles = t.tiles_uu.p; ma.ntiles = (int)tuu.size(); ma.xc = nullptr; ma.ldxc = 0; ma.ncols = M;
    RC(t.ph_zz.prepare(sz.off, sz.off, C, T, Mpad, Mpad, m->st, ma.ph));
    ma.G = t.GA.p; ma.ldg = Mpad; ma.ru = beta; ma.rw = beta; ma.rcoef = -0.5; ma.sym = 1;
    ma.gzr = t.gz.p; ma.gzc = t.gz.p; ma.partial = t.partial_uu.p;
    gz_attach(t, ma, false);
    RC(launch_moments(ma, m->st));
    RC(launch_moment_reduce(t.partial_uu.p, t.ps_uu.p, P, T, W, D, t.mom_uu.p, m->st, 1));

    std::vector<double> hgz((size_t)D * Mpad), hb(Mpad), hd(Mpad), hh(Npad);
    HIP_TRY(hipMemcpyAsync(mom_uu, t.mom_uu.p, (size_t)P * T * W * sizeof(double), hipMemcpyDeviceToHost, m->st));
    HIP_TRY(hipMemcpyAsync(mom_uf, t.mom_uf.p, (size_t)C * C * T * W * sizeof(double), hipMemcpyDeviceToHost, m->st));
    HIP_TRY(hipMemcpyAsync(hgz.data(), t.gz.p, hgz.size() * sizeof(double), hipMemcpyDeviceToHost, m->st));
    HIP_TRY(hipMemcpyAsync(hb.data(), beta, Mpad * sizeof(double), hipMemcpyDeviceToHost, m->st));
    HIP_TRY(hipMemcpyAsync(hd.data(), dga, Mpad * sizeof(double), hipMemcpyDeviceToHost, m->st));
    HIP_TRY(hipMemcpyAsync(hh.data(), h, Npad * sizeof(double), hipMemcpyDeviceToHost, m->st));
    HIP_TRY(hipStreamSynchronize(m->st));
    RC(sparse_timeout_check(m));
    for (int64_t pos = 0; pos < M; ++pos)
        for (int d = 0; d < D; ++d) gZ[sz.perm[pos] * D + d] = hgz[(size_t)d * Mpad + pos];
    double tr = 0.0;
    for (int64_t i = 0; i < M; ++i) tr += hd[i] - 0.5 * hb[i] * hb[i];
    *trGA = tr;
    if (env) {                                   // the diagonal follows the points: dp/dKff_nn per training point, caller order (N values)
        for (int64_t pos = 0; pos < N; ++pos) hsum[m->sx.perm[pos]] = hh[pos];
        return MOGP_OK;
    }
    for (int c = 0; c < C; ++c) {
        double s = 0.0;
        for (int pos = m->sx.off[c]; pos < m->sx.off[c + 1]; ++pos) s += hh[pos];
        hsum[c] = s;
    }
    if (sharded) {                               // sum of h over the points of a channel: one more small all-reduce
        RC(t.red.ensure((size_t)Mpad + 4 + C));
        HIP_TRY(hipMemcpyAsync(t.red.p, hsum, C * sizeof(double), hipMemcpyHostToDevice, m->st));
        HIP_TRY(hipStreamSynchronize(m->st));
        RC(comm_allreduce(m->ctx, t.red.p, C, m->st));
        HIP_TRY(hipMemcpyAsync(hsum, t.red.p, C * sizeof(double), hipMemcpyDeviceToHost, m->st));
        HIP_TRY(hipStreamSynchronize(m->st));
    }
    return MOGP_OK;
}

int snelson_predict_impl(mogp_model* m, int64_t M, const double* Z, const double* noise_var, double jitter, const double* kff_diag,
                         const double* kss_diag, int64_t S, const double* Xs, double* mu, double* var, int64_t* info, bool sharded);

}  // namespace

extern "C" {

int mogp_snelson_eval(mogp_model* m, int64_t M, const double* Z, const double* noise_var, double jitter, const double* kff_diag, int flags,
                      double* lml, double* mom_uu, double* mom_uf, double* gZ, double* trGA, double* hsum, double* jitter_abs, int64_t* info) {
    return snelson_eval_impl(m, M, Z, noise_var, jitter, kff_diag, flags, lml, mom_uu, mom_uf, gZ, trGA, hsum, jitter_abs, info, false);
}

int mogp_snelson_eval_sharded(mogp_model* m, int64_t M, const double* Z, const double* noise_var, double jitter, const double* kff_diag, int flags,
                              double* lml, double* mom_uu, double* mom_uf, double* gZ, double* trGA, double* hsum, double* jitter_abs,
                              int64_t* info) {
    return snelson_eval_impl(m, M, Z, noise_var, jitter, kff_diag, flags, lml, mom_uu, mom_uf, gZ, trGA, hsum, jitter_abs, info, true);
}

int mogp_snelson_predict(mogp_model* m, int64_t M, const double* Z, const double* noise_var, double jitter, const double* kff_diag,
                         const double* kss_diag, int64_t S, const double* Xs, double* mu, double* var, int64_t* info) {
    return snelson_predict_impl(m, M, Z, noise_var, jitter, kff_diag, kss_diag, S, Xs, mu, var, info, false);
}

int mogp_snelson_predict_sharded(mogp_model* m, int64_t M, const double* Z, const double* noise_var, double jitter, const double* kff_diag,
                                 const double* kss_diag, int64_t S, const double* Xs, double* mu, double* var, int64_t* info) {
    return snelson_predict_impl(m, M, Z, noise_var, jitter, kff_diag, kss_diag, S, Xs, mu, var, info, true);
}

}  // extern "C"

namespace {

int snelson_predict_impl(mogp_model* m, int64_t M, const double* Z, const double* noise_var, double jitter, const double* kff_diag,
                         const double* kss_diag, int64_t S, const double* Xs, double* mu, double* var, int64_t* info, bool sharded) {
    if (!m || !Z || !noise_var || !kff_diag || !kss_diag || !Xs || !mu || !var || M <= 0 || S <= 0)
        return fail(MOGP_EINVAL, "mogp_snelson_predict: bad argument");
    RC(use_device(m->ctx));
    if (info) *info = 0;
    const int C = m->C, D = m->D;
    const bool env = m->Wt > 2 + 3 * D;
    SortedX sz, ss;
    std::vector<GTile> tuu, tuf, tus;
    std::vector<int> psuu, psuf;
    SnScalars sc;
    RC(snelson_front(m, M, Z, noise_var, jitter, kff_diag, sz, tuu, psuu, tuf, psuf, sc, info, false, sharded));
    TitsiasWork& t = *m->tw;
    const int64_t Mpad = t.Mpad;
    RC(sort_inputs(Xs, S, D, C, MOGP_TILE, ss));
    const int64_t Spad = ss.Mpad;
    const int mt = (int)(Mpad / MOGP_TILE), st = (int)(Spad / MOGP_TILE);
    build_rect_tiles(sz.off, ss.off, C, tus);
    RC(t.Kus.ensure((size_t)Mpad * Spad)); RC(t.Aus.ensure((size_t)Mpad * Spad)); RC(t.Bus.ensure((size_t)Mpad * Spad));
    RC(m->d_xs.ensure((size_t)D * Spad)); RC(m->d_ptiles.ensure(tus.size()));
    RC(m->d_mu.ensure(Spad)); RC(m->d_var.ensure(2 * Spad));
    HIP_TRY(hipMemcpyAsync(m->d_xs.p, ss.xs.data(), (size_t)D * Spad * sizeof(double), hipMemcpyHostToDevice, m->st));
    HIP_TRY(hipMemcpyAsync(m->d_ptiles.p, tus.data(), tus.size() * sizeof(GTile), hipMemcpyHostToDevice, m->st));
    HIP_TRY(hipMemsetAsync(t.Kus.p, 0, (size_t)Mpad * Spad * sizeof(double), m->st));
    GramArgs ga{};
    ga.tiles = m->d_ptiles.p; ga.xr = t.zx.p; ga.ldxr = Mpad; ga.xc = m->d_xs.p; ga.ldxc = Spad; ga.nrows = M; ga.ncols = S;
    RC(t.ph_zs.prepare(sz.off, ss.off, C, m->T, Mpad, Spad, m->st, ga.ph));
    ga.table = m->d_table.p; ga.T = m->T; ga.D = D; ga.C = C; ga.W = m->Wt; ga.out = t.Kus.p; ga.ldo = Spad; ga.mirror = 0;
    RC(launch_gram(ga, (int)tus.size(), m->st));
    HIP_TRY(hipMemcpyAsync(t.Aus.p, t.Kus.p, (size_t)Mpad * Spad * sizeof(double), hipMemcpyDeviceToDevice, m->st));
    RC(trsm_lower(m, t.a.A.p, Mpad, mt, t.Aus.p, Spad, Spad, false));                                      // a = L^-1 Kus
    GemmArgs g = make_gemm(t.Wq, Mpad, 0, t.Aus.p, Spad, 1, t.Bus.p, Spad, 1.0, GM_KHI_I, mt, st, Mpad);          // b = Lq^-1 a
    RC(gemm_call(m, g, gemm_flops(g, nullptr)));
    double* vGy = t.vec.p;
    double* cvec = t.vec.p + 4 * Mpad;
    RC(launch_trmv_lower(t.Wq, Mpad, Mpad, vGy, cvec, t.vec.p + 6 * Mpad, m->st));                         // c = Lq^-1 v G y
    RC(launch_gemv_cols(t.Bus.p, Spad, Mpad, Spad, cvec, m->d_mu.p, t.scratch.p, m->st));                  // mu = b^T c
    RC(launch_gemv_cols(t.Aus.p, Spad, Mpad, Spad, nullptr, m->d_var.p, t.scratch.p, m->st));
    RC(launch_gemv_cols(t.Bus.p, Spad, Mpad, Spad, nullptr, m->d_var.p + Spad, t.scratch.p, m->st));
    std::vector<double> hmu(Spad), hv(2 * Spad);
    HIP_TRY(hipMemcpyAsync(hmu.data(), m->d_mu.p, Spad * sizeof(double), hipMemcpyDeviceToHost, m->st));
    HIP_TRY(hipMemcpyAsync(hv.data(), m->d_var.p, 2 * Spad * sizeof(double), hipMemcpyDeviceToHost, m->st));
    HIP_TRY(hipStreamSynchronize(m->st));
    RC(sparse_timeout_check(m));
    for (int c = 0; c < C; ++c)
        for (int pos = ss.off[c]; pos < ss.off[c + 1]; ++pos) {
            mu[ss.perm[pos]] = hmu[pos];
            var[ss.perm[pos]] = (env ? kss_diag[ss.perm[pos]] : kss_diag[c]) - hv[pos] + hv[Spad + pos];     // envelope: K_ss,diag per test point
        }
    return MOGP_OK;
}

}  // namespace
