// oa.hip -- the variational Gaussian approximation of Opper & Archambeau (2009) at the data points, the O(N^3) algebra around ANY likelihood.
// Reference: gpr/model.py:578-668 (OpperArchambeau): q(f) = N(K nu, (K^-1 + Lambda^2)^-1), nu and lambda one value per data point.
//
// Two calls per evaluation, the likelihood in between (host, O(N): its expectation E(mu, var) and e = dE/dmu, f = dE/dvar per point):
//   forward   B = Lambda K Lambda + I = L L^T  (no jitter, as in the reference),  B^-1 by the fused factorisation + inversion (potri.hip),
//             mu = K nu,   var = (1 - diag B^-1) / lambda^2,   kl = nu^T K nu + log det B + tr B^-1 - N        (ELBO = E - kl / 2)
//   backward  dK      = 1/2 (e nu^T + nu e^T) - 1/2 nu nu^T + Lambda (B^-1 diag(w) B^-1 - 1/2 B^-1) Lambda,   w = f / lambda^2 + 1/2
//             dnu     = K (e - nu)
//             dlambda = -2 f (1 - b) / lambda^3 + (2 / lambda) (d b - r) - (1 - b) / lambda + (b - s) / lambda,    d = f / lambda^2,
//                       b = diag B^-1,  s = diag B^-2,  r = diag(B^-1 diag(d) B^-1)   (K Lambda = Lambda^-1 (B - I): only diagonals are needed)
//   dK is contracted with the kernel derivatives by the dense-mode moment kernel over the symmetric tiles of (X, X).
// The result does not depend on the order of the points (no whitening), so they are taken in the model's channel-sorted order.
// Checked against the reference's autograd through the numpy twin (oracle/table_model.py:oa_forward / oa_backward) and on the device.
#include "mogp_model.h"

#include <cmath>
#include <cstring>
#include <limits>

using namespace mogp;

#define RC(x) do { int r__ = (x); if (r__) return r__; } while (0)

namespace {

// A[i][j] = lam[i] lam[j] K[i][j] (+ 1 on the diagonal) over the lower triangle's tiles (rows >= cols by tile; whole diagonal tiles)
__global__ void k_oa_scale(const double* __restrict__ K, double* __restrict__ A, int64_t ld, const double* __restrict__ lam) {
    const int64_t i = blockIdx.x;
    const int64_t jend = (i / MOGP_TILE + 1) * MOGP_TILE;
    const double li = lam[i];
    for (int64_t j = threadIdx.x; j < jend; j += blockDim.x) {
        const int64_t lo = j <= i ? i * ld + j : j * ld + i;      // K holds its lower triangle
        A[i * ld + j] = li * lam[j] * K[lo] + (i == j ? 1.0 : 0.0);
    }
}

// per row i of the (symmetric, full) inverse: b = Binv[i][i], s = sum_j Binv[i][j]^2, r = sum_j Binv[i][j]^2 d[j]
__global__ void k_oa_rowstats(const double* __restrict__ Binv, int64_t ld, int64_t n, const double* __restrict__ d,
                              double* __restrict__ s_out, double* __restrict__ r_out) {
    const int64_t i = blockIdx.x;
    __shared__ double rs[256], rr[256];
    double s = 0.0, r = 0.0;
    for (int64_t j = threadIdx.x; j < n; j += 256) {
        const double v = Binv[i * ld + j];
        s += v * v; r += v * v * d[j];
    }
    rs[threadIdx.x] = s; rr[threadIdx.x] = r;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
        if ((int)threadIdx.x < k) { rs[threadIdx.x] += rs[threadIdx.x + k]; rr[threadIdx.x] += rr[threadIdx.x + k]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { s_out[i] = rs[0]; r_out[i] = rr[0]; }
}

// out[i][j] = in[i][j] * w[j]
__global__ void k_oa_scale_cols(const double* __restrict__ in, double* __restrict__ out, int64_t ld, int64_t n, const double* __restrict__ w) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const int64_t i = blockIdx.y;
    out[i * ld + j] = in[i * ld + j] * w[j];
}

// G[i][j] = lam[i] lam[j] (Y[i][j] - Binv[i][j] / 2) + (e[i] nu[j] + nu[i] e[j] - nu[i] nu[j]) / 2      (in place over Y)
__global__ void k_oa_adjoint(double* __restrict__ Y, const double* __restrict__ Binv, int64_t ld, int64_t n, const double* __restrict__ lam,
                             const double* __restrict__ nu, const double* __restrict__ e) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const int64_t i = blockIdx.y;
    Y[i * ld + j] = lam[i] * lam[j] * (Y[i * ld + j] - 0.5 * Binv[i * ld + j]) + 0.5 * (e[i] * nu[j] + nu[i] * e[j] - nu[i] * nu[j]);
}


// vector slots of OaWork::vec (each Npad)
enum { V_NU = 0, V_LAM, V_MU, V_B, V_E, V_F, V_D, V_W, V_S, V_R, V_EN, V_GNU, V_COUNT };

int upload_sorted(mogp_model* m, const double* src, double* dst, double pad) {
    std::vector<double> h(m->Npad, pad);
    for (int64_t pos = 0; pos < m->N; ++pos) h[pos] = src[m->sx.perm[pos]];
    HIP_TRY(dev_upload(dst, h.data(), m->Npad * sizeof(double)));
    return 0;
}

}  // namespace

extern "C" {

int mogp_oa_forward(mogp_model* m, const double* q_nu, const double* q_lambda, double* mu, double* var, double* kl, int64_t* info) {
    if (!m || !q_nu || !q_lambda || !mu || !var || !kl) return fail(MOGP_EINVAL, "mogp_oa_forward: bad argument");
    RC(use_device(m->ctx));
    if (info) *info = 0;
    const int C = m->C, D = m->D;
    const int64_t N = m->N, Npad = m->Npad;
    if (m->T <= 0) return fail(MOGP_EINVAL, "mogp_model_set_terms must be called before an evaluation");
    if (m->Wt != 2 + 3 * D) return fail(MOGP_EINVAL, "the Opper-Archambeau path does not take terms with an envelope (MOHSM): exact inference only");
    for (int64_t i = 0; i < N; ++i)
        if (!(q_lambda[i] > 0.0)) return fail(MOGP_EINVAL, "mogp_oa_forward: q_lambda must be positive");
    OaWork& o = m->oa;
    o.valid = false;
    one_gpu_call(m);
    RC(ensure_system(m));
    m->have_W = m->have_Kinv = false;
    m->kinv_sparse = false;                    // this model reads every entry of the inverse (the exact model's gradient plan does not apply)
    m->gemm_ev_used = 0; m->gemm_launches = 0; m->gemm_flops = 0.0;
    RC(o.K.ensure((size_t)Npad * Npad));
    RC(o.vec.ensure((size_t)V_COUNT * Npad));
    const int nchunks = (int)((Npad + 511) / 512);
    RC(m->d_symv.ensure((size_t)(4 + nchunks) * Npad));
    double* nu = o.vec.p + V_NU * Npad;
    double* lam = o.vec.p + V_LAM * Npad;
    RC(upload_sorted(m, q_nu, nu, 0.0));
    RC(upload_sorted(m, q_lambda, lam, 0.0));
    const unsigned long long big = std::numeric_limits<unsigned long long>::max();
    HIP_TRY(hipMemcpyAsync(m->d_info.p, &big, sizeof(big), hipMemcpyHostToDevice, m->st));

    // K (lower triangle's tiles), no noise, no jitter
    GramArgs ga{};
    ga.tiles = m->d_tiles.p; ga.xr = m->d_x.p; ga.xc = m->d_x.p; ga.ldxr = ga.ldxc = Npad; ga.nrows = ga.ncols = N;
    RC(m->ph_xx.prepare(m->sx.off, m->sx.off, C, m->T, Npad, Npad, m->st, ga.ph));
    ga.table = m->d_table.p; ga.T = m->T; ga.D = D; ga.C = C; ga.W = m->Wt;
    ga.out = o.K.p; ga.ldo = Npad; ga.noise = nullptr; ga.dvar = nullptr; ga.jitter_abs = 0.0; ga.mirror = 0;
    m->strip.attach(ga);
    RC(launch_gram(ga, (int)m->tiles.size(), m->st));
    RC(launch_pad_identity(o.K.p, Npad, N, Npad, m->st));          // padded rows / columns: zero off the diagonal (lambda = 0 there anyway)
    // mu = K nu
    double* dmu = o.vec.p + V_MU * Npad;
    RC(launch_symv_lower(o.K.p, Npad, Npad, nu, dmu, m->d_symv.p, 1.0, m->st, 0, 0));
    std::vector<double> hmu(Npad), hb(Npad), hl(m->nb);
    HIP_TRY(hipMemcpyAsync(hmu.data(), dmu, Npad * sizeof(double), hipMemcpyDeviceToHost, m->st));
    // B = Lambda K Lambda + I -> B^-1 (m->k.B: lower tiles, then mirrored), log det from the factorisation
    hipLaunchKernelGGL(k_oa_scale, dim3((unsigned)Npad), dim3(256), 0, m->st, o.K.p, m->k.A.p, Npad, lam);
    HIP_TRY(hipGetLastError());
    if (m->nb <= (flow_enabled(m, m->k) ? 112 : 80)) {             // the same choice of schedule as mogp_exact_eval (112 tile rows as dataflow, 80 as streams)
        RC(spd_potri_fused(m, m->k));
        RC(spd_potri_fused_finish(m, m->k));
    } else {
        RC(spd_potrf(m, m->k)); RC(spd_trtri(m, m->k)); RC(spd_lauum(m, m->k));
    }
    RC(launch_symmetrize(m->k.B.p, Npad, Npad, m->st));
    double* db = o.vec.p + V_B * Npad;
    RC(launch_get_diag(m->k.B.p, Npad, Npad, db, m->st));
    unsigned long long hinfo = 0;
    HIP_TRY(hipMemcpyAsync(hb.data(), db, Npad * sizeof(double), hipMemcpyDeviceToHost, m->st));
    HIP_TRY(hipMemcpyAsync(hl.data(), m->k.logdet.p, m->nb * sizeof(double), hipMemcpyDeviceToHost, m->st));
    HIP_TRY(hipMemcpyAsync(&hinfo, m->d_info.p, sizeof(hinfo), hipMemcpyDeviceToHost, m->st));
    HIP_TRY(hipStreamSynchronize(m->st));
    if (hinfo == MOGP_INFO_CHAIN_TIMEOUT) {                 // see chain_fallback: repeat on the launch-per-step chain
        RC(chain_fallback(m));
        return mogp_oa_forward(m, q_nu, q_lambda, mu, var, kl, info);
    }
    if (hinfo != big) {
        if (info) *info = (int64_t)hinfo;
        return fail(MOGP_ENOTPD, "linalg.cholesky: The factorization could not be completed because the input is not "
                                 "positive-definite (the leading minor of order " + std::to_string(hinfo) + " is not positive-definite).");
    }
    double logdet = 0.0, tr = 0.0, maha = 0.0;
    for (double v : hl) logdet += v;                              // sum log L_ii (the padding contributes log 1)
    for (int64_t pos = 0; pos < N; ++pos) {
        const int64_t dst = m->sx.perm[pos];
        mu[dst] = hmu[pos];
        var[dst] = (1.0 - hb[pos]) / (q_lambda[dst] * q_lambda[dst]);
        tr += hb[pos];
        maha += q_nu[dst] * hmu[pos];
    }
    *kl = maha + 2.0 * logdet + tr - (double)N;
    o.valid = true;
    return MOGP_OK;
}

int mogp_oa_backward(mogp_model* m, const double* e, const double* f, double* moments, double* g_nu, double* g_lambda) {
    if (!m || !e || !f || !moments || !g_nu || !g_lambda) return fail(MOGP_EINVAL, "mogp_oa_backward: bad argument");
    RC(use_device(m->ctx));
    OaWork& o = m->oa;
    if (!o.valid) return fail(MOGP_EINVAL, "mogp_oa_backward: no forward pass precedes it");
    o.valid = false;
    const int C = m->C, D = m->D, W = m->Wt, T = m->T, P = C * (C + 1) / 2;
    const int64_t N = m->N, Npad = m->Npad;
    const int nt = (int)(Npad / MOGP_TILE);
    RC(o.Sc.ensure((size_t)Npad * Npad)); RC(o.Y.ensure((size_t)Npad * Npad));
    double* v = o.vec.p;
    double *nu = v + V_NU * Npad, *lam = v + V_LAM * Npad, *de = v + V_E * Npad, *df = v + V_F * Npad, *dd = v + V_D * Npad,
           *dw = v + V_W * Npad, *ds = v + V_S * Npad, *dr = v + V_R * Npad, *den = v + V_EN * Npad, *dgnu = v + V_GNU * Npad;
    // e, f, d = f / lambda^2, w = d + 1/2, e - nu in the device's order (all zero on the padding)
    std::vector<double> hnu(Npad), hlam(Npad);
    HIP_TRY(hipMemcpy(hnu.data(), nu, Npad * sizeof(double), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(hlam.data(), lam, Npad * sizeof(double), hipMemcpyDeviceToHost));
    std::vector<double> he(Npad, 0.0), hf(Npad, 0.0), hd(Npad, 0.0), hw(Npad, 0.0), hen(Npad, 0.0);
    for (int64_t pos = 0; pos < N; ++pos) {
        he[pos] = e[m->sx.perm[pos]]; hf[pos] = f[m->sx.perm[pos]];
        hd[pos] = hf[pos] / (hlam[pos] * hlam[pos]); hw[pos] = hd[pos] + 0.5; hen[pos] = he[pos] - hnu[pos];
    }
    HIP_TRY(hipMemcpyAsync(de, he.data(), Npad * sizeof(double), hipMemcpyHostToDevice, m->st));
    HIP_TRY(hipMemcpyAsync(df, hf.data(), Npad * sizeof(double), hipMemcpyHostToDevice, m->st));
    HIP_TRY(hipMemcpyAsync(dd, hd.data(), Npad * sizeof(double), hipMemcpyHostToDevice, m->st));
    HIP_TRY(hipMemcpyAsync(dw, hw.data(), Npad * sizeof(double), hipMemcpyHostToDevice, m->st));
    HIP_TRY(hipMemcpyAsync(den, hen.data(), Npad * sizeof(double), hipMemcpyHostToDevice, m->st));
    const double* Binv = m->k.B.p;
    // the diagonals the lambda gradient needs, and dnu = K (e - nu)
    hipLaunchKernelGGL(k_oa_rowstats, dim3((unsigned)Npad), dim3(256), 0, m->st, Binv, Npad, Npad, dd, ds, dr);
    HIP_TRY(hipGetLastError());
    RC(launch_symv_lower(o.K.p, Npad, Npad, den, dgnu, m->d_symv.p, 1.0, m->st, 0, 0));
    // Y = B^-1 diag(w) B^-1 (lower tiles, mirrored), then the adjoint of K in place
    const dim3 gnn((unsigned)((Npad + 255) / 256), (unsigned)Npad);
    hipLaunchKernelGGL(k_oa_scale_cols, gnn, dim3(256), 0, m->st, Binv, o.Sc.p, Npad, Npad, dw);
    HIP_TRY(hipGetLastError());
    GemmArgs g = make_gemm(o.Sc.p, Npad, 0, Binv, Npad, 0, o.Y.p, Npad, 1.0, GM_LOWER, nt, nt, Npad);
    RC(gemm_call(m, g, gemm_flops(g, nullptr)));
    RC(launch_symmetrize(o.Y.p, Npad, Npad, m->st));
    hipLaunchKernelGGL(k_oa_adjoint, gnn, dim3(256), 0, m->st, o.Y.p, Binv, Npad, Npad, lam, nu, de);
    HIP_TRY(hipGetLastError());

    MomentArgs ma{};
    ma.tiles = m->d_tiles.p; ma.ntiles = (int)m->tiles.size();
    ma.x = m->d_x.p; ma.ldx = Npad; ma.nrows = ma.ncols = N; ma.xc = nullptr; ma.ldxc = 0;
    RC(m->ph_xx.prepare(m->sx.off, m->sx.off, C, T, Npad, Npad, m->st, ma.ph));
    ma.table = m->d_table.p; ma.T = T; ma.D = D; ma.C = C; ma.W = W;
    ma.G = o.Y.p; ma.ldg = Npad; ma.ru = nu; ma.rw = nu; ma.rcoef = 0.0; ma.sym = 1;
    ma.gzr = nullptr; ma.gzc = nullptr; ma.ldgz = Npad;
    ma.partial = m->d_partial.p;
    ma.phases_ready = 1;                                        // the forward pass's Gram filled ph_xx for these inputs and this table
    RC(launch_moments(ma, m->st));
    RC(launch_moment_reduce(m->d_partial.p, m->d_pair_start.p, P, T, W, D, m->d_moments.p, m->st, 1));

    std::vector<double> hb(Npad), hs(Npad), hr(Npad), hg(Npad);
    HIP_TRY(hipMemcpyAsync(moments, m->d_moments.p, (size_t)P * T * W * sizeof(double), hipMemcpyDeviceToHost, m->st));
    HIP_TRY(hipMemcpyAsync(hb.data(), v + V_B * Npad, Npad * sizeof(double), hipMemcpyDeviceToHost, m->st));
    HIP_TRY(hipMemcpyAsync(hs.data(), ds, Npad * sizeof(double), hipMemcpyDeviceToHost, m->st));
    HIP_TRY(hipMemcpyAsync(hr.data(), dr, Npad * sizeof(double), hipMemcpyDeviceToHost, m->st));
    HIP_TRY(hipMemcpyAsync(hg.data(), dgnu, Npad * sizeof(double), hipMemcpyDeviceToHost, m->st));
    HIP_TRY(hipStreamSynchronize(m->st));
    for (int64_t pos = 0; pos < N; ++pos) {
        const int64_t dst = m->sx.perm[pos];
        const double l = hlam[pos], b = hb[pos];
        g_nu[dst] = hg[pos];
        g_lambda[dst] = -2.0 * hf[pos] * (1.0 - b) / (l * l * l) + (2.0 / l) * (hd[pos] * b - hr[pos]) - (1.0 - b) / l + (b - hs[pos]) / l;
    }
    return MOGP_OK;
}

}  // extern "C"
