// gram.hip -- spectral Gram builder and gradient-moment pass (gfx950).
//
// Every MOSM / SM / CSM channel-pair block is   K_ab = sum_t A_t exp(-1/2 sum_d V_td u_d^2) cos(2 pi (sum_d M_td u_d + Psi_t)),
// u_d = x_a,d - x_b,d + Delta_td   (reference: gpr/multioutput.py:182-204, :432-449; gpr/singleoutput.py:594-600).
//
// Cosine: splits per point,  cos(2 pi (p_a - q_b)) = cos(2 pi p_a) cos(2 pi q_b) + sin(2 pi p_a) sin(2 pi q_b)  with
// p_a = sum_d M_d (x_a,d + Delta_d) + Psi,  q_b = sum_d M_d x_b,d.  The per-point factors are computed ONCE per launch into a phase
// table (k_phase_table: C T N sincospi instead of one per point and tile) and read back by the tiles.
//
// Gaussian: one 64x64 tile of sorted time series spans a small input range, so with the tile centres c_r, c_c
// (p = x_a - c_r, q = x_b - c_c, s = c_r - c_c + Delta, u = p - q + s)
//     exp(-1/2 V u^2) = exp(-1/2 V s^2) * exp(-1/2 V (p^2 + 2 p s)) * exp(-1/2 V (q^2 - 2 q s)) * exp(V p q)
//                       per tile            per row                     per column                   per entry, |V p q| small.
// The per-row / per-column factors are folded into the staged cosine factors (128 exp per tile and term instead of 4096) and the
// cross term is a Taylor polynomial whose degree (4 .. 14) is chosen per tile and term from max |V p q| (truncation < 2e-17): an entry
// costs one multiply, N FMAs and three more FMAs instead of a library exp.  Tiles whose inputs are spread too far for that (unsorted
// rows, huge gaps) take the general path (exp per entry); a term whose Gaussian is below e^-50 everywhere in the tile is skipped
// (absolute error < 2e-22 A).  One workgroup = one 64x64 tile of one channel-pair block, 256 threads, 4x4 entries per thread; full
// interior tiles store 32-byte runs.
#include "mogp_internal.h"

#include <algorithm>
#include <vector>
#include <cstdlib>

// measurement switches of the tile kernels (no stores / no terms / constant staging: tools/r4_gram_lds.sh) exist only in a build with
// -DMOGP_GRAM_DEBUG; in the product kernels the conditions are the constant 0 and the branches are gone
#ifdef MOGP_GRAM_DEBUG
#define GRAM_DBG(a, bit) ((a).dbg & (bit))
#else
#define GRAM_DBG(a, bit) 0
#endif

namespace mogp {

typedef double d2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ double frac_turn(double p) { return p - rint(p); }

#define GT_SKIP (-1)     // term negligible in this tile
#define GT_GENERAL 0     // exp per entry
#define GT_SKIP_EXPONENT 50.0
// Taylor degree of the cross term exp(V p q) from max |V p q| over the tile: the smallest N with zmax^(N+1) / (N+1)! <= 2e-17 (round 6: every degree from 4 to 12 instead
// of 4 / 6 / 8 / 11 / 14 -- at configs[1] (zmax 0.012 .. 0.12) an entry's polynomial is 1 - 2 terms shorter)
#define GT_DEGREE(zmax) ((zmax) <= 1.19e-3 ? 4 : (zmax) <= 4.93e-3 ? 5 : (zmax) <= 0.0139 ? 6 : (zmax) <= 0.0308 ? 7 : (zmax) <= 0.0578 ? 8 : (zmax) <= 0.0968 ? 9 : \
                         (zmax) <= 0.149 ? 10 : (zmax) <= 0.2147 ? 11 : (zmax) <= 0.294 ? 12 : 14)
#define GT_DEGREE_CASES(X) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(14)

// e^x for |x| < 700: Cody-Waite reduction, degree-13 Taylor on |r| <= ln2 / 2 (truncation 4e-18), ldexp
__device__ __forceinline__ double fast_exp(double x) {
    const double n = rint(x * 1.44269504088896338700e+00);
    double r = fma(-n, 6.93147180369123816490e-01, x);
    r = fma(-n, 1.90821492927058770002e-10, r);
    double p = 1.6059043836821613e-10;                  // 1/13!
    p = fma(p, r, 2.08767569878681e-09);                // 1/12!
    p = fma(p, r, 2.505210838544172e-08);               // 1/11!
    p = fma(p, r, 2.755731922398589e-07);               // 1/10!
    p = fma(p, r, 2.7557319223985893e-06);              // 1/9!
    p = fma(p, r, 2.48015873015873e-05);                // 1/8!
    p = fma(p, r, 1.984126984126984e-04);               // 1/7!
    p = fma(p, r, 1.388888888888889e-03);               // 1/6!
    p = fma(p, r, 8.333333333333333e-03);               // 1/5!
    p = fma(p, r, 4.1666666666666664e-02);              // 1/4!
    p = fma(p, r, 1.6666666666666666e-01);              // 1/3!
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    return ldexp(p, (int)n);
}

// sum_{k <= N} z^k / k!   (Horner)
template <int N>
__device__ __forceinline__ double exp_taylor(double z) {
    constexpr double c[15] = {1.0, 1.0, 0.5, 1.6666666666666666e-01, 4.1666666666666664e-02, 8.333333333333333e-03,
                              1.388888888888889e-03, 1.984126984126984e-04, 2.48015873015873e-05, 2.7557319223985893e-06,
                              2.755731922398589e-07, 2.505210838544172e-08, 2.08767569878681e-09, 1.6059043836821613e-10,
                              1.1470745597729725e-11};
    double p = c[N];
#pragma unroll
    for (int k = N - 1; k >= 0; --k) p = fma(p, z, c[k]);
    return p;
}

// ---- phase table: cos / sin of the per-point phases, once per launch --------------------------------------------------------
// role 0 (rows):    cs[(o T + t) ld + a] = cos 2 pi (sum_d M_d (x_a,d + Delta_d) + Psi)   of pair (chan(a), o), term t
// role 1 (columns): cs[(o T + t) ld + b] = cos 2 pi (sum_d M_d x_b,d)                     of pair (o, chan(b)), term t
struct PhaseArgs {
    const double* x; int64_t ld; const int* off; const double* table; int T, D, C, W, role; double* cs; double* sn;
};
__global__ __launch_bounds__(256) void k_phase_table(PhaseArgs a) {
    const int64_t pt = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (pt >= a.off[a.C]) return;
    const int o = blockIdx.y, t = blockIdx.z, D = a.D, W = a.W;
    int c = 0;
    while (pt >= a.off[c + 1]) ++c;
    const int pair = a.role == 0 ? c * a.C + o : o * a.C + c;
    const double* row = a.table + ((size_t)pair * a.T + t) * W;
    double ph = a.role == 0 ? row[1] : 0.0;
    for (int d = 0; d < D; ++d) {
        const double m = row[2 + D + d], xv = a.x[(size_t)d * a.ld + pt];
        ph = a.role == 0 ? fma(m, xv + row[2 + 2 * D + d], ph) : fma(m, xv, ph);
    }
    double sn, cs;
    sincospi(2.0 * frac_turn(ph), &sn, &cs);
    const size_t idx = ((size_t)o * a.T + t) * a.ld + pt;
    a.cs[idx] = cs;
    a.sn[idx] = sn;
}

// centre and half span of every 64-point block of every channel (the row / column ranges of the tiles), stored at the index of the
// block's first point:  cen[d * ld + start], half[d * ld + start].  One wave per block.
struct CentreArgs { const double* x; int64_t ld; const int* off; int C, D; double* cen; double* half; };
__global__ __launch_bounds__(64) void k_block_centres(CentreArgs a) {
    int b = blockIdx.x, c = 0;
    for (; c < a.C; ++c) {
        const int nb = (a.off[c + 1] - a.off[c] + MOGP_GT - 1) / MOGP_GT;
        if (b < nb) break;
        b -= nb;
    }
    if (c >= a.C) return;
    const int start = a.off[c] + b * MOGP_GT, n = min(MOGP_GT, a.off[c + 1] - start), lane = threadIdx.x;
    for (int d = 0; d < a.D; ++d) {
        const double v = a.x[(size_t)d * a.ld + start + (lane < n ? lane : 0)];
        double lo = v, hi = v;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { lo = fmin(lo, __shfl_xor(lo, off, 64)); hi = fmax(hi, __shfl_xor(hi, off, 64)); }
        if (lane == 0) { a.cen[(size_t)d * a.ld + start] = 0.5 * (lo + hi); a.half[(size_t)d * a.ld + start] = 0.5 * (hi - lo); }
    }
}

// workspace layout: [row cos | row sin] (C T ldr each) [col cos | col sin] (C T ldc each) [row centres | row half spans] (D ldr each)
// [col centres | col half spans] (D ldc each)
size_t phase_ws_doubles(int C, int T, int64_t ldr, int64_t ldc) { return (size_t)2 * (C * T + MOGP_MAXD) * (size_t)(ldr + ldc); }   // sized for any D

struct PhaseView {                       // device pointers into the workspace of one launch
    const double *rcs, *rsn, *ccs, *csn, *rcen, *rhalf, *ccen, *chalf;
};
__host__ __device__ static inline PhaseView phase_view(double* ws, int C, int T, int D, int64_t ldr, int64_t ldc) {
    PhaseView v;
    double* p = ws;
    v.rcs = p; p += (size_t)C * T * ldr; v.rsn = p; p += (size_t)C * T * ldr;
    v.ccs = p; p += (size_t)C * T * ldc; v.csn = p; p += (size_t)C * T * ldc;
    v.rcen = p; p += (size_t)D * ldr; v.rhalf = p; p += (size_t)D * ldr;
    v.ccen = p; p += (size_t)D * ldc; v.chalf = p;
    return v;
}

static int launch_phase_tables(const PhaseRef& ph, const double* xr, int64_t ldxr, int64_t nr, const double* xc, int64_t ldxc, int64_t nc,
                               const double* table, int T, int D, int C, int W, hipStream_t s) {
    if (!ph.ws || !ph.offr || !ph.offc) { set_error("Gram / moment launch without a phase workspace"); return -1; }
    const PhaseView v = phase_view(ph.ws, C, T, D, ldxr, ldxc);
    PhaseArgs a;
    a.table = table; a.T = T; a.D = D; a.C = C; a.W = W;
    a.x = xr; a.ld = ldxr; a.off = ph.offr; a.role = 0; a.cs = const_cast<double*>(v.rcs); a.sn = const_cast<double*>(v.rsn);
    if (nr > 0) hipLaunchKernelGGL(k_phase_table, dim3((unsigned)((nr + 255) / 256), C, T), dim3(256), 0, s, a);
    a.x = xc; a.ld = ldxc; a.off = ph.offc; a.role = 1; a.cs = const_cast<double*>(v.ccs); a.sn = const_cast<double*>(v.csn);
    if (nc > 0) hipLaunchKernelGGL(k_phase_table, dim3((unsigned)((nc + 255) / 256), C, T), dim3(256), 0, s, a);
    CentreArgs c;
    c.C = C; c.D = D;
    c.x = xr; c.ld = ldxr; c.off = ph.offr; c.cen = const_cast<double*>(v.rcen); c.half = const_cast<double*>(v.rhalf);
    if (nr > 0) hipLaunchKernelGGL(k_block_centres, dim3((unsigned)((nr + MOGP_GT - 1) / MOGP_GT + C)), dim3(64), 0, s, c);
    c.x = xc; c.ld = ldxc; c.off = ph.offc; c.cen = const_cast<double*>(v.ccen); c.half = const_cast<double*>(v.chalf);
    if (nc > 0) hipLaunchKernelGGL(k_block_centres, dim3((unsigned)((nc + MOGP_GT - 1) / MOGP_GT + C)), dim3(64), 0, s, c);
    HIP_TRY(hipGetLastError());
    return 0;
}

// ---- per-tile staging shared by the Gram and the moment kernel -----------------------------------------------------------------
// Everything a tile needs from global memory depends on its descriptor only and is requested in ONE batch (the thread's own centred
// inputs, and for each of its staging items the term-table row, the point's input and its phase factors); every thread derives the
// term's mode and tile scalars itself (a handful of flops, identical in all threads), so a chunk of terms costs one barrier.
template <int DM>
struct TileLds {
    double cu[MOGP_TC][MOGP_GT], su[MOGP_TC][MOGP_GT], cw[MOGP_TC][MOGP_GT], sw[MOGP_TC][MOGP_GT];
    double V[MOGP_TC][DM], s[MOGP_TC][DM], M[MOGP_TC][DM], A[MOGP_TC];
    double Kp[MOGP_TC][DM];                                // coefficient of the cross term p q: V - L / 4 (= V without an envelope)
    double L[MOGP_TC][DM], e[MOGP_TC][DM];                 // envelope precision and offset: (x_a + x_b)/2 - c = (p + q)/2 + e
    int deg[MOGP_TC];
};
template <int DM>
struct TileCtx { double cr[DM], cc[DM], hr[DM], hc[DM]; };      // tile centres and half spans (rows, columns)

template <int DM>
__device__ __forceinline__ void tile_centres(TileCtx<DM>& X, const GTile& tl, int D, const PhaseView& v, int64_t ldxr, int64_t ldxc) {
    for (int d = 0; d < D; ++d) {
        X.cr[d] = v.rcen[(size_t)d * ldxr + tl.r0]; X.hr[d] = v.rhalf[(size_t)d * ldxr + tl.r0];
        X.cc[d] = v.ccen[(size_t)d * ldxc + tl.c0]; X.hc[d] = v.chalf[(size_t)d * ldxc + tl.c0];
    }
}

template <int DM>
__device__ __forceinline__ void tile_context(TileCtx<DM>& X, double (&p)[4][DM], double (&q)[4][DM], const GTile& tl, int D,
                                             const PhaseView& v, const double* __restrict__ xr, int64_t ldxr,
                                             const double* __restrict__ xc, int64_t ldxc, int rg, int cg) {
    tile_centres<DM>(X, tl, D, v, ldxr, ldxc);
#pragma unroll
    for (int m = 0; m < 4; ++m)
        for (int d = 0; d < D; ++d) {
            p[m][d] = xr[(size_t)d * ldxr + tl.r0 + min(rg * 4 + m, tl.nr - 1)] - X.cr[d];
            q[m][d] = xc[(size_t)d * ldxc + tl.c0 + min(cg * 4 + m, tl.nc - 1)] - X.cc[d];
        }
}

constexpr int STAGE_ITEMS = MOGP_TC * MOGP_GT * 2 / 256;      // staging items per thread and chunk (4)
template <int DM>
struct StageItem { double x[DM], cs, sn; };                   // what is prefetched per item: the point's input and its phase factors

template <int DM>
__device__ __forceinline__ void stage_item_load(StageItem<DM>& I, int which, int t, int pnt, const GTile& tl, int D, int C, int T, int t0,
                                                const PhaseView& v, const double* __restrict__ xr, int64_t ldxr,
                                                const double* __restrict__ xc, int64_t ldxc) {
    const int i = tl.pair / C, j = tl.pair - i * C;
    if (which == 0) {
        const int64_t g = tl.r0 + min(pnt, tl.nr - 1);
        const size_t k = ((size_t)j * T + t0 + t) * ldxr + g;             // rows: table (j, t, point)
        I.cs = v.rcs[k]; I.sn = v.rsn[k];
        for (int d = 0; d < D; ++d) I.x[d] = xr[(size_t)d * ldxr + g];
    } else {
        const int64_t g = tl.c0 + min(pnt, tl.nc - 1);
        const size_t k = ((size_t)i * T + t0 + t) * ldxc + g;             // columns: table (i, t, point)
        I.cs = v.ccs[k]; I.sn = v.csn[k];
        for (int d = 0; d < D; ++d) I.x[d] = xc[(size_t)d * ldxc + g];
    }
}

// AMP: fold the amplitude A_t into the row factors (Gram); otherwise unit amplitude (moments).  `tab`: the pair's term rows (LDS copy or
// global memory -- a generic pointer), W doubles each.
template <int DM, bool AMP>
__device__ __forceinline__ void stage_item_compute(TileLds<DM>& L, const StageItem<DM>& I, const TileCtx<DM>& X, const double* tab, int W,
                                                   int which, int t, int pnt, int D, int t0) {
    const double* row = tab + (size_t)(t0 + t) * W;
    const double A = row[0];
    const bool env = W > 2 + 3 * D;
    double zmax = 0.0, es = 0.0, emin = 0.0, efac = 0.0, s[DM], V[DM], Lv[DM], er[DM], ec[DM];
    for (int d = 0; d < D; ++d) {
        V[d] = row[2 + d];
        Lv[d] = env ? row[2 + 3 * D + d] : 0.0;
        const double cn = env ? row[2 + 4 * D + d] : 0.0;
        er[d] = X.cr[d] - cn; ec[d] = X.cc[d] - cn;
        const double hr = X.hr[d], hc = X.hc[d];
        s[d] = (X.cr[d] - X.cc[d]) + row[2 + 2 * D + d];
        zmax += fabs(V[d] - 0.25 * Lv[d]) * hr * hc;
        es += V[d] * s[d] * s[d];
        const double mu = fmax(0.0, fabs(s[d]) - hr - hc);                      // smallest |u_d| in the tile
        emin += V[d] * mu * mu;
        const double amax = 0.5 * (fabs(er[d]) + hr + fabs(ec[d]) + hc);          // largest |midpoint - c| in the tile
        efac += fabs(V[d]) * (hr * hr + hc * hc + 2.0 * (hr + hc) * fabs(s[d])) + Lv[d] * amax * amax;    // bound on the row / column exponents
    }
    int deg;
    if ((AMP && A == 0.0) || 0.5 * emin > GT_SKIP_EXPONENT) deg = GT_SKIP;
    else if (!(zmax <= 0.45) || !(0.5 * es < 600.0) || !(0.5 * efac < 600.0)) deg = GT_GENERAL;
    else deg = GT_DEGREE(zmax);
    if (pnt == 0 && which == 0) {
        L.deg[t] = deg; L.A[t] = A;
        for (int d = 0; d < D; ++d) {
            L.V[t][d] = V[d]; L.s[t][d] = s[d]; L.M[t][d] = row[2 + D + d];
            L.Kp[t][d] = V[d] - 0.25 * Lv[d]; L.L[t][d] = Lv[d]; L.e[t][d] = 0.5 * (er[d] + ec[d]);
        }
    }
    if (deg == GT_SKIP) return;
    double f = 1.0;
    if (deg != GT_GENERAL) {
        // exponent = -1/2 V (p - q + s)^2 - L/8 ((p + er) + (q + ec))^2, split into a row part, a column part and the cross term (V - L/4) p q
        double e = 0.0;
        if (which == 0) {                                  // rows: the tile scalars exp(-1/2 V s^2 - L/4 er ec) included
            for (int d = 0; d < D; ++d) {
                const double pp = I.x[d] - X.cr[d];
                e = fma(V[d], fma(pp, pp, s[d] * s[d]) + 2.0 * pp * s[d], e);
                const double pe = pp + er[d];
                e = fma(0.25 * Lv[d], pe * pe + 2.0 * ec[d] * pe, e);
            }
        } else {                                           // columns
            for (int d = 0; d < D; ++d) {
                const double qq = I.x[d] - X.cc[d];
                e = fma(V[d], qq * qq - 2.0 * qq * s[d], e);
                const double qe = qq + ec[d];
                e = fma(0.25 * Lv[d], qe * qe + 2.0 * qq * er[d], e);
            }
        }
        f = fast_exp(-0.5 * e);
    }
    if (which == 0) {
        if (AMP) f *= A;
        L.cu[t][pnt] = f * I.cs; L.su[t][pnt] = f * I.sn;
    } else {
        L.cw[t][pnt] = f * I.cs; L.sw[t][pnt] = f * I.sn;
    }
}

// Staging items of a chunk: (rows | columns) x term x point.  Wave w of the workgroup handles rows (w even) or columns (w odd) of the terms
// w / 2, w / 2 + 2, w / 2 + 4, w / 2 + 6, lane = point: no index arithmetic, and which / t are wave-uniform.
#define STAGE_MAP(tid) const int st_wave = __builtin_amdgcn_readfirstlane((tid) >> 6), st_which = st_wave & 1, st_tb = st_wave >> 1, st_pnt = (tid) & 63

// one chunk of terms into LDS; BATCH: all items' loads first (one memory round trip), else item by item (large D: registers)
template <int DM, bool AMP, bool BATCH>
__device__ __forceinline__ void stage_chunk(TileLds<DM>& L, const TileCtx<DM>& X, const GTile& tl, const double* tab, int W, int D,
                                            int C, int T, int t0, int nt, const PhaseView& v, const double* __restrict__ xr, int64_t ldxr,
                                            const double* __restrict__ xc, int64_t ldxc, int tid) {
    STAGE_MAP(tid);
    if (BATCH) {
        StageItem<DM> I[STAGE_ITEMS];
#pragma unroll
        for (int k = 0; k < STAGE_ITEMS; ++k)
            if (st_tb + 2 * k < nt) stage_item_load<DM>(I[k], st_which, st_tb + 2 * k, st_pnt, tl, D, C, T, t0, v, xr, ldxr, xc, ldxc);
#pragma unroll
        for (int k = 0; k < STAGE_ITEMS; ++k)
            if (st_tb + 2 * k < nt) stage_item_compute<DM, AMP>(L, I[k], X, tab, W, st_which, st_tb + 2 * k, st_pnt, D, t0);
    } else {
        for (int t = st_tb; t < nt; t += 2) {
            StageItem<DM> I;
            stage_item_load<DM>(I, st_which, t, st_pnt, tl, D, C, T, t0, v, xr, ldxr, xc, ldxc);
            stage_item_compute<DM, AMP>(L, I, X, tab, W, st_which, t, st_pnt, D, t0);
        }
    }
}

// Gaussian factor of one entry on the general path
template <int DM>
__device__ __forceinline__ double gauss_general(const double (&p)[DM], const double (&q)[DM], const double* V, const double* s,
                                                const double* Lv, const double* ev, int D) {
    double arg = 0.0;
    for (int d = 0; d < D; ++d) {
        const double u = (p[d] - q[d]) + s[d];
        arg = fma(V[d] * u, u, arg);
        const double a = 0.5 * (p[d] + q[d]) + ev[d];          // envelope on the midpoint (L = 0 without one)
        arg = fma(Lv[d] * a, a, arg);
    }
    return exp(-0.5 * arg);
}

template <int DM, int N>
__device__ __forceinline__ void gram_term(double (&acc)[4][4], const double (&p)[4][DM], const double (&q)[4][DM], const TileLds<DM>& L, int t,
                                          int D, const double (&cu)[4], const double (&su)[4], const double (&cw)[4], const double (&sw)[4]) {
    double vp[4][DM];
    if (N > 0) {
#pragma unroll
        for (int m = 0; m < 4; ++m)
            for (int d = 0; d < D; ++d) vp[m][d] = L.Kp[t][d] * p[m][d];
    }
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        if (m == 2) __builtin_amdgcn_sched_barrier(0);      // two groups of eight independent Horner chains: enough to cover the FMA
                                                            // latency, half the live registers of sixteen
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            double e;
            if (N > 0) {
                double z = vp[m][0] * q[n][0];
                for (int d = 1; d < D; ++d) z = fma(vp[m][d], q[n][d], z);
                e = exp_taylor<N>(z);
            } else {
                e = gauss_general<DM>(p[m], q[n], L.V[t], L.s[t], L.L[t], L.e[t], D);
            }
            acc[m][n] = fma(e, fma(cu[m], cw[n], su[m] * sw[n]), acc[m][n]);
        }
    }
}

// tile epilogue: full interior tiles store 32-byte runs; diagonal / ragged tiles take the per-element path (augmentation, lower part only)
__device__ __forceinline__ void gram_store(const GramArgs& a, const GTile& tl, const double (&acc)[4][4], int rg, int cg) {
    const bool full = tl.nr == MOGP_GT && tl.nc == MOGP_GT && !(tl.flags & GT_DIAG) && ((tl.c0 & 1) == 0) && ((a.ldo & 1) == 0);
    if (full) {
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const int64_t at = (int64_t)(tl.r0 + rg * 4 + m) * a.ldo + tl.c0 + cg * 4;
            double* o = a.out + at;
            *reinterpret_cast<d2_t*>(o) = (d2_t){acc[m][0], acc[m][1]};
            *reinterpret_cast<d2_t*>(o + 2) = (d2_t){acc[m][2], acc[m][3]};
            if (a.out2) {
                *reinterpret_cast<d2_t*>(a.out2 + at) = (d2_t){acc[m][0], acc[m][1]};
                *reinterpret_cast<d2_t*>(a.out2 + at + 2) = (d2_t){acc[m][2], acc[m][3]};
            }
        }
        if (a.mirror && (tl.flags & GT_MIRROR)) {
#pragma unroll
            for (int n = 0; n < 4; ++n)
#pragma unroll
                for (int m = 0; m < 4; ++m) a.out[(int64_t)(tl.c0 + cg * 4 + n) * a.ldo + tl.r0 + rg * 4 + m] = acc[m][n];
        }
        return;
    }
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const int lr = rg * 4 + m;
        if (lr >= tl.nr) continue;
        const int64_t r = tl.r0 + lr;
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            const int lc = cg * 4 + n;
            if (lc >= tl.nc) continue;
            const int64_t c = tl.c0 + lc;
            if ((tl.flags & GT_DIAG) && lc > lr) continue;       // diagonal tiles: lower part computed, upper part mirrored
            double v = acc[m][n];
            if (a.noise != nullptr && r == c) {
                v += a.noise[tl.pair / a.C] + a.jitter_abs;
                if (a.dvar != nullptr) v += a.dvar[r];
            }
            a.out[r * a.ldo + c] = v;
            if (a.out2) a.out2[r * a.ldo + c] = v;
            if (a.mirror && ((tl.flags & GT_MIRROR) || ((tl.flags & GT_DIAG) && lr > lc))) a.out[c * a.ldo + r] = v;
        }
    }
}

// Persistent workgroups (two per CU), software pipelined over tiles blockIdx.x, + gridDim.x, ...: a tile's global reads (its centred
// inputs and staging items, all functions of the descriptor alone) are issued one tile ahead, before the previous tile's main loop, and
// the descriptor itself two tiles ahead -- a tile-per-workgroup launch spends more time in these two dependent memory round trips than
// in its arithmetic (measured: 66 of 126 us).  The staged factors are double buffered in LDS: one barrier per tile.
template <int DT>
__global__ __launch_bounds__(256, 2) void k_gram(GramArgs a, int ntiles) {
    constexpr int DM = DT > 0 ? DT : MOGP_MAXD;
    constexpr bool BATCH = DT == 1;
    const int D = DT > 0 ? DT : a.D;
    const int W = a.W;
    const int tid = threadIdx.x;
    const int cg = tid & 15, rg = tid >> 4;
    __shared__ TileLds<DM> Lb[2];
    extern __shared__ __attribute__((aligned(16))) double s_tab[];     // the whole term table when it fits (a.tab_lds)
    const PhaseView v = phase_view(a.ph.ws, a.C, a.T, D, a.ldxr, a.ldxc);
    const int stride = gridDim.x, nt0 = min(MOGP_TC, a.T);
    STAGE_MAP(tid);

    int tile = blockIdx.x;
    if (tile >= ntiles) return;
    if (a.tab_lds) {
        for (int e = tid; e < a.C * a.C * a.T * W; e += 256) s_tab[e] = a.table[e];
        __syncthreads();
    }
    const double* tabbase = a.tab_lds ? s_tab : a.table;
    GTile tl = a.tiles[tile];
    GTile tl1 = a.tiles[min(tile + stride, ntiles - 1)];
    TileCtx<DM> X;
    double p[4][DM], q[4][DM];
    StageItem<DM> I[BATCH ? STAGE_ITEMS : 1];
    tile_context<DM>(X, p, q, tl, D, v, a.xr, a.ldxr, a.xc, a.ldxc, rg, cg);
    if (BATCH) {
#pragma unroll
        for (int k = 0; k < STAGE_ITEMS; ++k)
            if (st_tb + 2 * k < nt0) stage_item_load<DM>(I[k], st_which, st_tb + 2 * k, st_pnt, tl, D, a.C, a.T, 0, v, a.xr, a.ldxr, a.xc, a.ldxc);
    }
    int buf = 0;
    while (true) {
        TileLds<DM>& L = Lb[buf];
        const GTile cur = tl;
        const double* tab = tabbase + (size_t)cur.pair * a.T * W;
        // stage this tile's first chunk of terms (from the prefetched registers when batched)
        if (BATCH) {
#pragma unroll
            for (int k = 0; k < STAGE_ITEMS; ++k)
                if (st_tb + 2 * k < nt0) stage_item_compute<DM, true>(L, I[k], X, tab, W, st_which, st_tb + 2 * k, st_pnt, D, 0);
        } else {
            stage_chunk<DM, true, false>(L, X, cur, tab, W, D, a.C, a.T, 0, nt0, v, a.xr, a.ldxr, a.xc, a.ldxc, tid);
        }
        double pc[4][DM], qc[4][DM];
#pragma unroll
        for (int m = 0; m < 4; ++m)
            for (int d = 0; d < D; ++d) { pc[m][d] = p[m][d]; qc[m][d] = q[m][d]; }
        // prefetch the next tile (its descriptor arrived during the previous iteration), and the descriptor after it
        const int nxt = tile + stride;
        const bool more = nxt < ntiles;
        if (more) {
            tl = tl1;
            tl1 = a.tiles[min(nxt + stride, ntiles - 1)];
            tile_context<DM>(X, p, q, tl, D, v, a.xr, a.ldxr, a.xc, a.ldxc, rg, cg);
            if (BATCH) {
#pragma unroll
                for (int k = 0; k < STAGE_ITEMS; ++k)
                    if (st_tb + 2 * k < nt0) stage_item_load<DM>(I[k], st_which, st_tb + 2 * k, st_pnt, tl, D, a.C, a.T, 0, v, a.xr, a.ldxr, a.xc, a.ldxc);
            }
        }
        __syncthreads();

        double acc[4][4];
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int n = 0; n < 4; ++n) acc[m][n] = 0.0;
        for (int t0 = 0; t0 < a.T; t0 += MOGP_TC) {
            const int nt = min(MOGP_TC, a.T - t0);
            if (t0 > 0) {                                   // more than MOGP_TC terms: the later chunks are staged in place
                __syncthreads();
                TileCtx<DM> Xc;
                tile_centres<DM>(Xc, cur, D, v, a.ldxr, a.ldxc);
                stage_chunk<DM, true, false>(L, Xc, cur, tab, W, D, a.C, a.T, t0, nt, v, a.xr, a.ldxr, a.xc, a.ldxc, tid);
                __syncthreads();
            }
            for (int t = 0; t < nt; ++t) {
                const int deg = L.deg[t];
                if (deg == GT_SKIP || GRAM_DBG(a, 2)) continue;
                double cu[4], su[4], cw[4], sw[4];
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    cu[m] = L.cu[t][rg * 4 + m]; su[m] = L.su[t][rg * 4 + m];
                    cw[m] = L.cw[t][cg * 4 + m]; sw[m] = L.sw[t][cg * 4 + m];
                }
                switch (deg) {
                    #define GT_CASE(N) case N: gram_term<DM, N>(acc, pc, qc, L, t, D, cu, su, cw, sw); break;
                    GT_DEGREE_CASES(GT_CASE)
#undef GT_CASE
                    default: gram_term<DM, 0>(acc, pc, qc, L, t, D, cu, su, cw, sw); break;
                }
            }
        }
        if (a.T > MOGP_TC) __syncthreads();                // the in-place chunks above must be consumed before this buffer is restaged
        if (!(GRAM_DBG(a, 1) && acc[0][0] != 12345.678)) gram_store(a, cur, acc, rg, cg);
        if (!more) break;
        tile = nxt;
        buf ^= 1;
    }
}

// ---- strip kernel: D = 1, no envelope, runs of full interior tiles ---------------------------------------------------------------
// The general kernel above pays two dependent global round trips per tile (descriptor -> inputs and phase factors) or, software
// pipelined, the registers to hide them (it spills).  Training data are channel-sorted series: almost every tile is a full 64 x 64
// interior tile, and consecutive tiles of the list share their 64 rows.  One workgroup takes a RUN of such tiles: the rows' inputs and
// phase factors are read once, the next tile's 64 columns (input + 2 T phase factors: at most five doubles per thread) are fetched under
// the current tile's arithmetic, and the tile-centred factors are staged from LDS.  Same arithmetic as the general kernel, term by term.
// TC = terms a launch may carry (LDS is sized by it); raw rows of a block of 64 points: x | cos_t | sin_t.  Two workgroups per CU: the
// arithmetic needs ~200 VGPRs, and three or four waves per SIMD bought with spills ran 1.3x / 2x slower.
#define GS_TC_MAX 8
// NC = columns per thread: 4 (256 threads, 4 x 4 entries each: ~200 VGPRs, two waves per SIMD) or 2 (512 threads, 4 x 2 entries: under 128 VGPRs,
// FOUR waves per SIMD -- round 6: the kernel's waves were parked at s_waitcnt / s_barrier 36 % of their cycles with two)
template <int TC, int NC = 4>
struct StripLds {
    static constexpr int NT = 64 * (MOGP_GT / NC) / 4;
    static constexpr int RAW = 1 + 2 * TC, PF = (RAW * MOGP_GT + NT - 1) / NT;      // PF: prefetch registers per thread
    // column-side arrays in two planes: a thread's four columns 4 cg .. 4 cg + 3 are two 16-byte reads, and with the points in order lane cg's reads sat
    // 32 bytes apart -- every fourth lane on the same banks (SQ_LDS_BANK_CONFLICT: 54 % of the kernel's LDS cycles).  Columns 4 cg, 4 cg + 1 at
    // [2 cg], columns 4 cg + 2, 4 cg + 3 at [CP + 2 cg]: each read is 16 contiguous lanes x 16 bytes.
    // (NC = 2: a thread's two columns are ONE 16-byte read, 32 lanes x 16 bytes contiguous in the natural order)
    static constexpr int CP = 36, CL = 72;
    __device__ static __forceinline__ int cs(int pnt) { return NC == 2 ? pnt : ((pnt & 2) ? CP : 0) + ((pnt >> 2) << 1) + (pnt & 1); }
    double rowraw[RAW][MOGP_GT];
    double colraw[2][RAW][CL];
    double cu[TC][MOGP_GT], su[TC][MOGP_GT], cw[TC][CL], sw[TC][CL];
    double tab[TC][3];                            // A, V, Delta of the pair's terms
    double V[TC], s[TC];
    int deg[TC];
};

template <int N, int NC>
__device__ __forceinline__ void strip_term(double (&acc)[4][NC], const double (&p)[4], const double (&q)[NC], double V, double s,
                                           const double (&cu)[4], const double (&su)[4], const double (&cw)[NC], const double (&sw)[NC]) {
    double vp[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) vp[m] = V * p[m];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        if (m == 2 && NC == 4) __builtin_amdgcn_sched_barrier(0);      // two groups of eight independent Horner chains (see gram_term)
#pragma unroll
        for (int n = 0; n < NC; ++n) {
            double e;
            if (N > 0) {
                e = exp_taylor<N>(vp[m] * q[n]);
            } else {
                const double u = (p[m] - q[n]) + s;
                e = fast_exp(fmax(-0.5 * (V * u) * u, -745.0));
            }
            acc[m][n] = fma(e, fma(cu[m], cw[n], su[m] * sw[n]), acc[m][n]);
        }
    }
}

template <int TC, int NC>
__global__ __launch_bounds__(64 * (MOGP_GT / NC) / 4, NC == 2 ? 4 : 2) void k_gram_strip(GramArgs a, const GSeg* __restrict__ segs) {
    using SL = StripLds<TC, NC>;
    constexpr int GS_TC = TC, GS_PF = SL::PF, NT = SL::NT, NCG = MOGP_GT / NC;
    __shared__ SL L;
    const int tid = threadIdx.x, cg = tid % NCG, rg = tid / NCG;
    const GSeg sg = segs[blockIdx.x];
    const int T = a.T, W = a.W;
    const int ci = sg.pair / a.C, cj = sg.pair - ci * a.C;
    const PhaseView v = phase_view(a.ph.ws, a.C, T, 1, a.ldxr, a.ldxc);
    const double* __restrict__ tab = a.table + (size_t)sg.pair * T * W;
    const int nraw = (1 + 2 * T) * MOGP_GT;
    // raw row k of a block of points: 0 = input, 1 .. T = cos of term k - 1, T + 1 .. 2 T = sin of term k - 1 - T; LDS slot of row k
    auto slot = [&](int k) { return k <= T ? k : k - T + GS_TC; };

    for (int e = tid; e < nraw; e += NT) {               // the row block, once per run (rows take the table of the COLUMN channel)
        const int k = e >> 6, pnt = e & 63;
        const int64_t gp = sg.r0 + pnt;
        double val;
        if (k == 0) val = a.xr[gp];
        else if (k <= T) val = v.rcs[((size_t)cj * T + (k - 1)) * a.ldxr + gp];
        else val = v.rsn[((size_t)cj * T + (k - 1 - T)) * a.ldxr + gp];
        L.rowraw[slot(k)][pnt] = val;
    }
    if (tid < 3 * T) { const int t = tid / 3, k = tid - 3 * t; L.tab[t][k] = tab[(size_t)t * W + (k == 0 ? 0 : 2 * k)]; }      // columns 0, 2, 4
    const double cr = v.rcen[sg.r0], hr = v.rhalf[sg.r0];
    double pf[GS_PF], cc_n, hc_n;
    auto col_fetch = [&](int c0) {                        // columns take the table of the ROW channel
#pragma unroll
        for (int k5 = 0; k5 < GS_PF; ++k5) {
            const int e = tid + NT * k5;
            if (e < nraw) {
                const int k = e >> 6, pnt = e & 63;
                const int64_t gp = c0 + pnt;
                if (k == 0) pf[k5] = a.xc[gp];
                else if (k <= T) pf[k5] = v.ccs[((size_t)ci * T + (k - 1)) * a.ldxc + gp];
                else pf[k5] = v.csn[((size_t)ci * T + (k - 1 - T)) * a.ldxc + gp];
            }
        }
        cc_n = v.ccen[c0]; hc_n = v.chalf[c0];
    };
    // gfx9 counts loads and stores in ONE counter and the compiler, seeing both kinds pending, waits for all of them (vmcnt(0)): the next
    // tile's columns go to LDS BEFORE the current tile is stored, so that wait never includes a store just issued.
    auto col_commit = [&](int b) {
#pragma unroll
        for (int k5 = 0; k5 < GS_PF; ++k5) {
            const int e = tid + NT * k5;
            if (e < nraw) L.colraw[b][slot(e >> 6)][SL::cs(e & 63)] = pf[k5];
        }
    };
    col_fetch(sg.c0);
    col_commit(0);
    double cc = cc_n, hc = hc_n;
    __syncthreads();
    if (sg.n > 1) col_fetch(sg.c0 + MOGP_GT);
    double p[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) p[m] = L.rowraw[0][rg * 4 + m] - cr;

    for (int u = 0; u < sg.n; ++u) {
        const int b = u & 1, c0 = sg.c0 + u * MOGP_GT;
        // ---- stage the tile-centred factors: (rows | columns) x term x point ----
        for (int e = tid; e < 2 * T * MOGP_GT; e += NT) {
            const int which = e >= T * MOGP_GT, rem = e - which * T * MOGP_GT, t = rem >> 6, pnt = rem & 63;
            const double A = L.tab[t][0], V = L.tab[t][1], s = (cr - cc) + L.tab[t][2];
            const double zmax = fabs(V) * hr * hc, es = V * s * s;
            const double mu = fmax(0.0, fabs(s) - hr - hc), emin = V * mu * mu;
            const double efac = fabs(V) * (hr * hr + hc * hc + 2.0 * (hr + hc) * fabs(s));
            int deg;
            if (A == 0.0 || 0.5 * emin > GT_SKIP_EXPONENT) deg = GT_SKIP;
            else if (!(zmax <= 0.45) || !(0.5 * es < 600.0) || !(0.5 * efac < 600.0)) deg = GT_GENERAL;
            else deg = GT_DEGREE(zmax);
            if (pnt == 0 && which == 0) { L.deg[t] = deg; L.V[t] = V; L.s[t] = s; }
            if (deg == GT_SKIP) continue;
            double f = 1.0;
            if (GRAM_DBG(a, 4)) { L.cu[t][pnt] = 1.0; L.su[t][pnt] = 0.5; L.cw[t][SL::cs(pnt)] = 0.25; L.sw[t][SL::cs(pnt)] = 2.0; continue; }
            if (which == 0) {
                if (deg != GT_GENERAL) {
                    const double pp = L.rowraw[0][pnt] - cr;
                    f = fast_exp(-0.5 * (V * (fma(pp, pp, s * s) + 2.0 * pp * s)));
                }
                f *= A;
                L.cu[t][pnt] = f * L.rowraw[1 + t][pnt]; L.su[t][pnt] = f * L.rowraw[1 + GS_TC + t][pnt];
            } else {
                if (deg != GT_GENERAL) {
                    const double qq = L.colraw[b][0][SL::cs(pnt)] - cc;
                    f = fast_exp(-0.5 * (V * (qq * qq - 2.0 * qq * s)));
                }
                { const int cp = SL::cs(pnt); L.cw[t][cp] = f * L.colraw[b][1 + t][cp]; L.sw[t][cp] = f * L.colraw[b][1 + GS_TC + t][cp]; }
            }
        }
        __syncthreads();
        double q[NC], acc[4][NC];
#pragma unroll
        for (int n = 0; n < NC; ++n) q[n] = L.colraw[b][0][SL::cs(cg * NC + n)] - cc;
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int n = 0; n < NC; ++n) acc[m][n] = 0.0;
        for (int t = 0; t < T; ++t) {
            const int deg = __builtin_amdgcn_readfirstlane(L.deg[t]);
            if (deg == GT_SKIP || GRAM_DBG(a, 2)) continue;
            const double V = L.V[t], s = L.s[t];
            double cu[4], su[4], cw[NC], sw[NC];
#pragma unroll
            for (int m = 0; m < 4; ++m) { cu[m] = L.cu[t][rg * 4 + m]; su[m] = L.su[t][rg * 4 + m]; }
#pragma unroll
            for (int n = 0; n < NC; ++n) { cw[n] = L.cw[t][SL::cs(cg * NC + n)]; sw[n] = L.sw[t][SL::cs(cg * NC + n)]; }
            switch (deg) {
                #define GT_CASE(N) case N: strip_term<N, NC>(acc, p, q, V, s, cu, su, cw, sw); break;
                GT_DEGREE_CASES(GT_CASE)
#undef GT_CASE
                default: strip_term<0, NC>(acc, p, q, V, s, cu, su, cw, sw); break;
            }
        }
        if (u + 1 < sg.n) { col_commit(b ^ 1); cc = cc_n; hc = hc_n; }
        if (sg.diag && u + 1 == sg.n) {
            // the run's last tile sits on the matrix diagonal: the general kernel's epilogue for such a tile (gram_store) -- lower part only,
            // noise + jitter (+ per-point variance) on the diagonal entries
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const int lr = rg * 4 + m;
                const int64_t r = sg.r0 + lr;
#pragma unroll
                for (int n = 0; n < NC; ++n) {
                    const int lc = cg * NC + n;
                    if (lc > lr) continue;
                    double val = acc[m][n];
                    if (a.noise != nullptr && lc == lr) {
                        val += a.noise[ci] + a.jitter_abs;
                        if (a.dvar != nullptr) val += a.dvar[r];
                    }
                    a.out[r * a.ldo + c0 + lc] = val;
                    if (a.out2) a.out2[r * a.ldo + c0 + lc] = val;
                }
            }
        } else if (!(GRAM_DBG(a, 1) && acc[0][0] != 12345.678)) {
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const int64_t at = (int64_t)(sg.r0 + rg * 4 + m) * a.ldo + c0 + cg * NC;
                double* o = a.out + at;
#pragma unroll
                for (int n = 0; n < NC; n += 2) *reinterpret_cast<d2_t*>(o + n) = (d2_t){acc[m][n], acc[m][n + 1]};
                if (a.out2) {                              // the sparse models' working copy of K_uf (same leading dimension)
                    double* o2 = a.out2 + at;
#pragma unroll
                    for (int n = 0; n < NC; n += 2) *reinterpret_cast<d2_t*>(o2 + n) = (d2_t){acc[m][n], acc[m][n + 1]};
                }
            }
        }
        __syncthreads();                                   // the next tile's columns are in LDS; this tile's factors are consumed
        if (u + 2 < sg.n) col_fetch(c0 + 2 * MOGP_GT);
    }
}

// ---- strip kernel, second form (round 6): ONE barrier per tile ---------------------------------------------------------------------
// k_gram_strip above synchronises twice per tile (staged factors written -> read; read -> the buffers are restaged) and its waves were parked at
// s_waitcnt / s_barrier 36 % of their cycles.  Here the tile-centred factors are double buffered and staged one tile AHEAD: iteration u stages tile
// u + 1 into F[b ^ 1], computes tile u from F[b], commits the columns of tile u + 2 (prefetched an iteration earlier) and the scalars of tile u + 2
// (degree, centre distance: computed ONCE per tile and term by wave 0, not by every staging item), stores tile u; one barrier; the columns of tile
// u + 3 are requested behind it.  The 2 T 64 staging items of a tile do not divide over 256 threads for T = 3 (384): the half wave-pair that takes a
// second item alternates with the tile's parity.  The addresses of the column prefetch and the LDS slots of its commit are per-thread constants,
// worked out once per run.  Same arithmetic, entry by entry, as k_gram_strip (the same row / column factors, the same polynomial): same bits.
template <int TC>
struct Strip2Lds {
    static constexpr int RAW = 1 + 2 * TC, PF = (RAW * MOGP_GT + 255) / 256;
    static constexpr int CP = 36, CL = 72;
    __device__ static __forceinline__ int cs(int pnt) { return ((pnt & 2) ? CP : 0) + ((pnt >> 2) << 1) + (pnt & 1); }
    double rowraw[RAW][MOGP_GT];
    double colraw[2][RAW][CL];
    double cu[2][TC][MOGP_GT], su[2][TC][MOGP_GT], cw[2][TC][CL], sw[2][TC][CL];
    double qv[2][CL];                             // q = x_b - (column tile centre) of the staged tile, in the column planes' order
    double tab[TC][3];                            // A, V, Delta of the pair's terms
    double s[3][TC];                              // centre distance + delay of tile u mod 3 (written two tiles ahead of its last reader)
    int deg[3][TC];
};

template <int TC>
__global__ __launch_bounds__(256, 2) void k_gram_strip2(GramArgs a, const GSeg* __restrict__ segs) {
    using SL = Strip2Lds<TC>;
    constexpr int PF = SL::PF, NT = 256;
    __shared__ SL L;
    const int tid = threadIdx.x, cg = tid & 15, rg = tid >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const GSeg sg = segs[blockIdx.x];
    const int T = a.T, W = a.W;
    const int ci = sg.pair / a.C, cj = sg.pair - ci * a.C;
    const PhaseView v = phase_view(a.ph.ws, a.C, T, 1, a.ldxr, a.ldxc);
    const double* __restrict__ tab = a.table + (size_t)sg.pair * T * W;
    const int nraw = (1 + 2 * T) * MOGP_GT;
    auto slot = [&](int k) { return k <= T ? k : k - T + TC; };

    for (int e = tid; e < nraw; e += NT) {               // the row block, once per run (rows take the table of the COLUMN channel)
        const int k = e >> 6, pnt = e & 63;
        const int64_t gp = sg.r0 + pnt;
        double val;
        if (k == 0) val = a.xr[gp];
        else if (k <= T) val = v.rcs[((size_t)cj * T + (k - 1)) * a.ldxr + gp];
        else val = v.rsn[((size_t)cj * T + (k - 1 - T)) * a.ldxr + gp];
        L.rowraw[slot(k)][pnt] = val;
    }
    if (tid < 3 * T) { const int t = tid / 3, k = tid - 3 * t; L.tab[t][k] = tab[(size_t)t * W + (k == 0 ? 0 : 2 * k)]; }      // columns 0, 2, 4
    const double cr = v.rcen[sg.r0], hr = v.rhalf[sg.r0];
    // column prefetch: per-thread constants (source array + point, LDS slot), columns take the table of the ROW channel
    const double* psrc[PF];
    int pdst[PF];
#pragma unroll
    for (int k5 = 0; k5 < PF; ++k5) {
        const int e = tid + NT * k5, k = e >> 6, pnt = e & 63;
        psrc[k5] = nullptr; pdst[k5] = 0;
        if (e < nraw) {
            if (k == 0) psrc[k5] = a.xc + pnt;
            else if (k <= T) psrc[k5] = v.ccs + ((size_t)ci * T + (k - 1)) * a.ldxc + pnt;
            else psrc[k5] = v.csn + ((size_t)ci * T + (k - 1 - T)) * a.ldxc + pnt;
            pdst[k5] = slot(k) * SL::CL + SL::cs(pnt);
        }
    }
    double pf[PF], cc_n = 0.0, hc_n = 0.0;
    auto col_fetch = [&](int c0) {
#pragma unroll
        for (int k5 = 0; k5 < PF; ++k5) if (psrc[k5]) pf[k5] = psrc[k5][c0];
        cc_n = v.ccen[c0]; hc_n = v.chalf[c0];
    };
    auto col_commit = [&](int b) {
        double* plane = &L.colraw[b][0][0];
#pragma unroll
        for (int k5 = 0; k5 < PF; ++k5) if (psrc[k5]) plane[pdst[k5]] = pf[k5];
    };
    // degree / centre distance of tile `u` (column centre cc, half span hc): ONE wave, one lane per term
    auto scalars = [&](int u, double cc, double hc) {
        if (wave == 0 && tid < T) {
            const double A = L.tab[tid][0], V = L.tab[tid][1], s = (cr - cc) + L.tab[tid][2];
            const double zmax = fabs(V) * hr * hc, es = V * s * s;
            const double mu = fmax(0.0, fabs(s) - hr - hc), emin = V * mu * mu;
            const double efac = fabs(V) * (hr * hr + hc * hc + 2.0 * (hr + hc) * fabs(s));
            int deg;
            if (A == 0.0 || 0.5 * emin > GT_SKIP_EXPONENT) deg = GT_SKIP;
            else if (!(zmax <= 0.45) || !(0.5 * es < 600.0) || !(0.5 * efac < 600.0)) deg = GT_GENERAL;
            else deg = GT_DEGREE(zmax);
            L.deg[u % 3][tid] = deg; L.s[u % 3][tid] = s;
        }
    };
    // the tile-centred factors of tile u (columns in colraw[u & 1], centre cc) into F[u & 1]
    auto stage = [&](int u, double cc) {
        const int b = u & 1, u3 = u % 3;
        auto item = [&](int e) {
            const int which = e >= T * MOGP_GT, rem = e - which * T * MOGP_GT, t = rem >> 6, pnt = rem & 63;
            const int deg = L.deg[u3][t];
            if (deg == GT_SKIP) return;
            if (GRAM_DBG(a, 4)) { L.cu[b][t][pnt] = 1.0; L.su[b][t][pnt] = 0.5; L.cw[b][t][SL::cs(pnt)] = 0.25; L.sw[b][t][SL::cs(pnt)] = 2.0; return; }
            const double V = L.tab[t][1], s = L.s[u3][t];
            double f = 1.0;
            if (which == 0) {
                if (deg != GT_GENERAL) {
                    const double pp = L.rowraw[0][pnt] - cr;
                    f = fast_exp(-0.5 * (V * (fma(pp, pp, s * s) + 2.0 * pp * s)));
                }
                f *= L.tab[t][0];
                L.cu[b][t][pnt] = f * L.rowraw[1 + t][pnt]; L.su[b][t][pnt] = f * L.rowraw[1 + TC + t][pnt];
            } else {
                const int cp = SL::cs(pnt);
                const double qq = L.colraw[b][0][cp] - cc;
                if (deg != GT_GENERAL) f = fast_exp(-0.5 * (V * (qq * qq - 2.0 * qq * s)));
                L.cw[b][t][cp] = f * L.colraw[b][1 + t][cp]; L.sw[b][t][cp] = f * L.colraw[b][1 + TC + t][cp];
            }
        };
        const int items = 2 * T * MOGP_GT, full = items / NT;                // (items come in multiples of 128: which / t are wave-uniform)
        for (int r = 0; r < full; ++r) item(tid + NT * r);
        if ((items & (NT - 1)) && (wave & 2) == ((u & 1) << 1)) item(NT * full + (tid & 127));      // T odd: 128 items left, the wave pairs take turns
        if (tid < MOGP_GT) { const int cp = SL::cs(tid); L.qv[b][cp] = L.colraw[b][0][cp] - cc; }
    };

    // ---- prologue: tile 0 staged, tile 1's columns and scalars in LDS, tile 2's columns requested.  The columns of tiles 0 and 1 are requested TOGETHER
    // (one memory round trip behind the run descriptor's, not two) ----
    double pf1[PF], cc0, hc0, cc1 = 0.0, hc1 = 0.0;
    col_fetch(sg.c0);
    if (sg.n > 1) {
#pragma unroll
        for (int k5 = 0; k5 < PF; ++k5) if (psrc[k5]) pf1[k5] = psrc[k5][sg.c0 + MOGP_GT];
        cc1 = v.ccen[sg.c0 + MOGP_GT]; hc1 = v.chalf[sg.c0 + MOGP_GT];
    }
    col_commit(0);
    cc0 = cc_n; hc0 = hc_n;
    if (sg.n > 1) {
        double* plane = &L.colraw[1][0][0];
#pragma unroll
        for (int k5 = 0; k5 < PF; ++k5) if (psrc[k5]) plane[pdst[k5]] = pf1[k5];
    }
    __syncthreads();                                    // rowraw, tab, colraw[0], colraw[1]
    scalars(0, cc0, hc0);
    if (sg.n > 1) scalars(1, cc1, hc1);
    if (sg.n > 2) col_fetch(sg.c0 + 2 * MOGP_GT);
    __syncthreads();                                    // the scalars of tiles 0 and 1
    stage(0, cc0);
    __syncthreads();                                    // F[0]
    double p[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) p[m] = L.rowraw[0][rg * 4 + m] - cr;

    for (int u = 0; u < sg.n; ++u) {
        const int b = u & 1, u3 = u % 3, c0 = sg.c0 + u * MOGP_GT;
        if (u + 1 < sg.n) stage(u + 1, cc1);
        double q[4], acc[4][4];
#pragma unroll
        for (int n = 0; n < 4; ++n) q[n] = L.qv[b][SL::cs(cg * 4 + n)];
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int n = 0; n < 4; ++n) acc[m][n] = 0.0;
        for (int t = 0; t < T; ++t) {
            const int deg = __builtin_amdgcn_readfirstlane(L.deg[u3][t]);
            if (deg == GT_SKIP || GRAM_DBG(a, 2)) continue;
            const double V = L.tab[t][1], s = L.s[u3][t];
            double cu[4], su[4], cw[4], sw[4];
#pragma unroll
            for (int m = 0; m < 4; ++m) { cu[m] = L.cu[b][t][rg * 4 + m]; su[m] = L.su[b][t][rg * 4 + m]; }
#pragma unroll
            for (int n = 0; n < 4; ++n) { cw[n] = L.cw[b][t][SL::cs(cg * 4 + n)]; sw[n] = L.sw[b][t][SL::cs(cg * 4 + n)]; }
            switch (deg) {
                #define GT_CASE(N) case N: strip_term<N, 4>(acc, p, q, V, s, cu, su, cw, sw); break;
                GT_DEGREE_CASES(GT_CASE)
#undef GT_CASE
                default: strip_term<0, 4>(acc, p, q, V, s, cu, su, cw, sw); break;
            }
        }
        if (u + 2 < sg.n) {                                // the columns of tile u + 2 (requested an iteration ago) and its scalars
            col_commit(b);
            scalars(u + 2, cc_n, hc_n);
        }
        cc0 = cc1; cc1 = cc_n;
        if (sg.diag && u + 1 == sg.n) {
            // the run's last tile sits on the matrix diagonal: lower part only, noise + jitter (+ per-point variance) on the diagonal entries
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const int lr = rg * 4 + m;
                const int64_t r = sg.r0 + lr;
#pragma unroll
                for (int n = 0; n < 4; ++n) {
                    const int lc = cg * 4 + n;
                    if (lc > lr) continue;
                    double val = acc[m][n];
                    if (a.noise != nullptr && lc == lr) {
                        val += a.noise[ci] + a.jitter_abs;
                        if (a.dvar != nullptr) val += a.dvar[r];
                    }
                    a.out[r * a.ldo + c0 + lc] = val;
                    if (a.out2) a.out2[r * a.ldo + c0 + lc] = val;
                }
            }
        } else if (!(GRAM_DBG(a, 1) && acc[0][0] != 12345.678)) {
            double* o = a.out + (int64_t)(sg.r0 + rg * 4) * a.ldo + c0 + cg * 4;
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                *reinterpret_cast<d2_t*>(o + m * a.ldo) = (d2_t){acc[m][0], acc[m][1]};
                *reinterpret_cast<d2_t*>(o + m * a.ldo + 2) = (d2_t){acc[m][2], acc[m][3]};
            }
            if (a.out2) {                                  // the sparse models' working copy of K_uf (same leading dimension)
                double* o2 = a.out2 + (int64_t)(sg.r0 + rg * 4) * a.ldo + c0 + cg * 4;
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    *reinterpret_cast<d2_t*>(o2 + m * a.ldo) = (d2_t){acc[m][0], acc[m][1]};
                    *reinterpret_cast<d2_t*>(o2 + m * a.ldo + 2) = (d2_t){acc[m][2], acc[m][3]};
                }
            }
        }
        // F[b ^ 1], colraw[b], the scalars of tile u + 2: written; F[b], qv[b]: read.  An LDS-only barrier: __syncthreads() would also wait for the
        // tile's stores to land, which nobody here reads
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (u + 3 < sg.n) col_fetch(c0 + 3 * MOGP_GT);
    }
}

void split_strip_tiles(const std::vector<GTile>& tiles, int maxrun, std::vector<GSeg>& segs, std::vector<GTile>& rest) {
    segs.clear(); rest.clear();
    for (const GTile& t : tiles) {
        // round 5: a full-size tile ON the diagonal rides at the end of its row block's run (the strip kernel's store handles it: lower part only,
        // noise + jitter on the diagonal) instead of going to the general kernel in a launch of its own -- 128 tiles, 17 us at the head of every
        // configs[1] evaluation.  (The strip kernel never mirrors: launch_gram takes it only when a.mirror is 0.)
        const bool fast = t.nr == MOGP_GT && t.nc == MOGP_GT && (t.c0 & 1) == 0;
        if (!fast) { rest.push_back(t); continue; }
        const int dg = (t.flags & GT_DIAG) ? 1 : 0;
        if (!segs.empty()) {
            GSeg& g = segs.back();
            if (!g.diag && g.pair == t.pair && g.r0 == t.r0 && g.c0 + g.n * MOGP_GT == t.c0 && g.n < maxrun) { ++g.n; g.diag = dg; continue; }
        }
        segs.push_back(GSeg{t.r0, t.c0, 1, t.pair, dg, {0, 0, 0}});
    }
}

int launch_gram(const GramArgs& a0, int ntiles, hipStream_t s) {
    if (ntiles <= 0) return 0;
    GramArgs a = a0;
    if (a.W <= 0) a.W = 2 + 3 * a.D;
#ifdef MOGP_GRAM_DEBUG
    static const int dbg = []() { const char* e = std::getenv("MOGP_GRAM_DBG"); return e ? std::atoi(e) : 0; }();
    a.dbg = dbg;
#else
    a.dbg = 0;
#endif
    int rc = a.phases_ready ? 0 : launch_phase_tables(a.ph, a.xr, a.ldxr, a.nrows, a.xc, a.ldxc, a.ncols, a.table, a.T, a.D, a.C, a.W, s);
    if (rc) return rc;
    static const int ncu = []() { int dev = 0; hipDeviceProp_t pr; if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&pr, dev) != hipSuccess) return 256; return pr.multiProcessorCount; }();
    const int grid = std::min(ntiles, 2 * ncu);              // persistent: two workgroups per CU (__launch_bounds__(256, 2))
    const size_t tab_bytes = (size_t)a.C * a.C * a.T * a.W * sizeof(double);
    a.tab_lds = tab_bytes <= 24 * 1024;                      // the term table rides in LDS when small (it is read by every staging item)
    const size_t dyn = a.tab_lds ? tab_bytes : 0;
    if (a.ev0) HIP_TRY(hipEventRecord(a.ev0, s));
    static const bool strip_on = !(std::getenv("MOGP_GRAM_STRIP") && std::atoi(std::getenv("MOGP_GRAM_STRIP")) == 0);
    if (strip_on && a.segs && a.nsegs > 0 && a.D == 1 && a.W == 5 && a.T <= GS_TC_MAX && !a.mirror && (a.ldo & 1) == 0) {
        static const int strip_nc = []() { const char* e = std::getenv("MOGP_GRAM_NC"); const int v = e ? std::atoi(e) : 4; return v == 2 ? 2 : 4; }();      // MOGP_GRAM_NC=2: 4 x 2 entries per thread on 512 threads (four waves per SIMD; round 6: 148 vs 159 us on one box, 154 vs 146 on another -- not kept as the default)
        static const bool strip2 = !(std::getenv("MOGP_GRAM_STRIP2") && std::atoi(std::getenv("MOGP_GRAM_STRIP2")) == 0);      // the one-barrier form (k_gram_strip2)
        if (strip2 && strip_nc == 4) {
            if (a.T <= 4) hipLaunchKernelGGL((k_gram_strip2<4>), dim3(a.nsegs), dim3(256), 0, s, a, a.segs);
            else hipLaunchKernelGGL((k_gram_strip2<8>), dim3(a.nsegs), dim3(256), 0, s, a, a.segs);
        } else if (strip_nc == 2) {
            if (a.T <= 4) hipLaunchKernelGGL((k_gram_strip<4, 2>), dim3(a.nsegs), dim3(512), 0, s, a, a.segs);
            else hipLaunchKernelGGL((k_gram_strip<8, 2>), dim3(a.nsegs), dim3(512), 0, s, a, a.segs);
        } else {
            if (a.T <= 4) hipLaunchKernelGGL((k_gram_strip<4, 4>), dim3(a.nsegs), dim3(256), 0, s, a, a.segs);
            else hipLaunchKernelGGL((k_gram_strip<8, 4>), dim3(a.nsegs), dim3(256), 0, s, a, a.segs);
        }
        if (a.nrest > 0) {
            a.tiles = a.rest;
            hipLaunchKernelGGL(k_gram<1>, dim3(std::min(a.nrest, 2 * ncu)), dim3(256), dyn, s, a, a.nrest);
        }
        if (a.ev1) HIP_TRY(hipEventRecord(a.ev1, s));
        HIP_TRY(hipGetLastError());
        return 0;
    }
    switch (a.D) {
        case 1: hipLaunchKernelGGL(k_gram<1>, dim3(grid), dim3(256), dyn, s, a, ntiles); break;
        case 2: hipLaunchKernelGGL(k_gram<2>, dim3(grid), dim3(256), dyn, s, a, ntiles); break;
        case 3: hipLaunchKernelGGL(k_gram<3>, dim3(grid), dim3(256), dyn, s, a, ntiles); break;
        default: hipLaunchKernelGGL(k_gram<0>, dim3(grid), dim3(256), dyn, s, a, ntiles); break;
    }
    if (a.ev1) HIP_TRY(hipEventRecord(a.ev1, s));
    HIP_TRY(hipGetLastError());
    return 0;
}

// ---- gradient moments ---------------------------------------------------------------------------------
// partial[tile][t][w], w = [ m0 = sum g E cos, m4 = sum g E sin, m1_d = sum g u_d^2 E cos, m2_d = sum g u_d E cos,
//                           m3_d = sum g u_d E sin ].
// Exact mode (DENSE = false):  g = weight * 1/2 (alpha_a alpha_b - Kinv_ab), weight = 2 (symmetric double count: strictly
// lower entries of diagonal channel blocks, every entry of off-diagonal channel blocks -- reference kernel.py:466-467
// writes k and k.T), 1 on the matrix diagonal, 0 above it.
// Dense mode (DENSE = true, Titsias):  g = weight * (G[a][b] + rcoef ru[a] rw[b]); weight as above when a.sym, else 1.
// ZG: also accumulate the gradient w.r.t. the row / column INPUTS (inducing points):
//     dK_ab/dx_a,d = sum_t A_t E [ -V_d u_d cos - 2 pi M_d sin ] = - dK_ab/dx_b,d.
// A thread accumulates the W moments of its 16 entries of one term in registers; the workgroup reduction goes through LDS (16 slice sums
// of 16 threads each, then a 4-step butterfly: fixed order, bit-reproducible) with ONE barrier per term.
template <int DM, int N, bool ZG, bool ENV>
__device__ __forceinline__ void moment_term(double* mom, const double (&g)[4][4], const double (&p)[4][DM], const double (&q)[4][DM],
                                            const TileLds<DM>& L, int t, int D,
                                            const double (&cu)[4], const double (&su)[4], const double (&cw)[4], const double (&sw)[4],
                                            double (&zr)[4][DM], double (&zc)[4][DM]) {
    const double* V = L.V[t];
    const double* s = L.s[t];
    const double* Mv = L.M[t];
    const double A = L.A[t];
    double vp[4][DM];
    if (N > 0) {
#pragma unroll
        for (int m = 0; m < 4; ++m)
            for (int d = 0; d < D; ++d) vp[m][d] = L.Kp[t][d] * p[m][d];
    }
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            double u[DM];
            for (int d = 0; d < D; ++d) u[d] = (p[m][d] - q[n][d]) + s[d];
            double e;
            if (N > 0) {
                double z = vp[m][0] * q[n][0];
                for (int d = 1; d < D; ++d) z = fma(vp[m][d], q[n][d], z);
                e = exp_taylor<N>(z);
            } else {
                e = gauss_general<DM>(p[m], q[n], V, s, L.L[t], L.e[t], D);
            }
            const double ge = g[m][n] * e;
            const double kc = ge * fma(cu[m], cw[n], su[m] * sw[n]);
            const double ks = ge * fma(su[m], cw[n], -cu[m] * sw[n]);
            mom[0] += kc;
            mom[1] += ks;
            for (int d = 0; d < D; ++d) {
                const double uk = u[d] * kc;
                mom[2 + d] = fma(u[d], uk, mom[2 + d]);
                mom[2 + D + d] += uk;
                mom[2 + 2 * D + d] = fma(u[d], ks, mom[2 + 2 * D + d]);
                if (ENV) {                                 // envelope moments: a = midpoint - centre
                    const double a = 0.5 * (p[m][d] + q[n][d]) + L.e[t][d];
                    const double ak = a * kc;
                    mom[2 + 3 * D + d] = fma(a, ak, mom[2 + 3 * D + d]);
                    mom[2 + 4 * D + d] += ak;
                }
                if (ZG) {
                    const double j = -A * fma(V[d], uk, 6.283185307179586476925286766559 * Mv[d] * ks);
                    zr[m][d] += j;
                    zc[n][d] -= j;
                    if (ENV) {                             // the envelope sits on the MIDPOINT: d/dx_a = d/dx_b = -1/2 L_d a_d K
                        const double je = -0.5 * A * L.L[t][d] * (0.5 * (p[m][d] + q[n][d]) + L.e[t][d]) * kc;
                        zr[m][d] += je;
                        zc[n][d] += je;
                    }
                }
            }
        }
}

// Two workgroups per CU wherever the registers allow it with a handful of spills (D = 1: the input-gradient variant needs 268 VGPRs
// unconstrained, i.e. ONE wave per SIMD and nothing to hide its loads behind).
// (round 4: cutting the registers to three or four waves per SIMD -- __launch_bounds__(256, 3 / 4) -- costs 81 / 92 spilled VGPRs; the pressure is
// the tile's adjoint block (g: 32 VGPRs), the staged factors and the staging prefetch, not the Horner chains)
template <int DT, bool DENSE, bool ZG, bool ENV>
__global__ __launch_bounds__(256, (DT == 1 ? 2 : 1)) void k_moments(MomentArgs a) {
    constexpr int DM = DT > 0 ? DT : MOGP_MAXD;
    constexpr int WM = 2 + (ENV ? 5 : 3) * DM;
    constexpr int RSTRIDE = 256 + 16;                        // one moment of all threads, +1 per 16 (the slice reads hit distinct banks)
    constexpr int NBUF = WM <= 11 ? 2 : 1;                   // double-buffered reduction staging: one barrier per term instead of two
    const int D = DT > 0 ? DT : a.D;
    const int W = a.W;
    const GTile tl = a.tiles[blockIdx.x];
    const double* tab = a.table + (size_t)tl.pair * a.T * W;
    const int tid = threadIdx.x;
    const int cg = tid & 15, rg = tid >> 4;
    const double* xcol = a.xc ? a.xc : a.x;
    const int64_t ldxc = a.xc ? a.ldxc : a.ldx;

    __shared__ TileLds<DM> L;
    __shared__ double s_red[NBUF][WM * RSTRIDE];
    __shared__ double s_gr[ZG ? DM : 1][MOGP_GT], s_gc[ZG ? DM : 1][MOGP_GT];

    double* outp = a.partial + (size_t)blockIdx.x * a.T * W;
    bool any = false;
    for (int t = 0; t < a.T; ++t) any |= (tab[(size_t)t * W] != 0.0);
    if (!any) {
        for (int idx = tid; idx < a.T * W; idx += 256) outp[idx] = 0.0;
        return;
    }

    const PhaseView v = phase_view(a.ph.ws, a.C, a.T, D, a.ldx, ldxc);
    TileCtx<DM> X;
    double p[4][DM], q[4][DM];
    tile_context<DM>(X, p, q, tl, D, v, a.x, a.ldx, xcol, ldxc, rg, cg);
    if (ZG) {
        for (int idx = tid; idx < D * MOGP_GT; idx += 256) { s_gr[idx / MOGP_GT][idx % MOGP_GT] = 0.0; s_gc[idx / MOGP_GT][idx % MOGP_GT] = 0.0; }
    }

    // g for this thread's 4x4 entries
    double g[4][4];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const int lr = rg * 4 + m;
        const int64_t r = tl.r0 + lr;
        double ar = 0.0;
        if (lr < tl.nr) ar = DENSE ? (a.ru ? a.rcoef * a.ru[r] : 0.0) : a.alpha[r];
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            const int lc = cg * 4 + n;
            const int64_t c = tl.c0 + lc;
            double w = 0.0, v = 0.0;
            if (lr < tl.nr && lc < tl.nc) {
                w = (!DENSE || a.sym) ? 2.0 : 1.0;
                if ((!DENSE || a.sym) && (tl.flags & GT_DIAG)) w = r > c ? 2.0 : (r == c ? 1.0 : 0.0);
                if (DENSE) {
                    const int64_t hi = (a.sym && c > r) ? c : r, lo = (a.sym && c > r) ? r : c;
                    v = a.G[hi * a.ldg + lo] + (a.ru ? ar * a.rw[c] : 0.0);
                } else {
                    const int64_t hi = r > c ? r : c, lo = r > c ? c : r;
                    if (a.row_mod > 1 && (int)((hi / MOGP_TILE) % a.row_mod) != a.row_rem) w = 0.0;       // row owned by another rank
                    else v = 0.5 * (ar * a.alpha[c] - a.kinv_sign * a.kinv[hi * a.ld + lo]);
                }
            }
            g[m][n] = w * v;
        }
    }

    double zr[4][DM], zc[4][DM];
    if (ZG) {
#pragma unroll
        for (int m = 0; m < 4; ++m)
            for (int d = 0; d < D; ++d) { zr[m][d] = 0.0; zc[m][d] = 0.0; }
    }

    int nred = 0;                                            // reductions done so far (selects the staging buffer)
    for (int t0 = 0; t0 < a.T; t0 += MOGP_TC) {
        const int nt = min(MOGP_TC, a.T - t0);
        if (t0 > 0) __syncthreads();                         // the first chunk has nothing to wait for
        stage_chunk<DM, false, (DT == 1)>(L, X, tl, tab, W, D, a.C, a.T, t0, nt, v, a.x, a.ldx, xcol, ldxc, tid);
        __syncthreads();
        for (int t = 0; t < nt; ++t) {
            const int deg = L.deg[t];
            if (deg == GT_SKIP) {                            // uniform across the workgroup
                if (tid < W) outp[(size_t)(t0 + t) * W + tid] = 0.0;
                continue;
            }
            double mom[WM];
#pragma unroll
            for (int w = 0; w < WM; ++w) mom[w] = 0.0;
            double cu[4], su[4], cw[4], sw[4];
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                cu[m] = L.cu[t][rg * 4 + m]; su[m] = L.su[t][rg * 4 + m];
                cw[m] = L.cw[t][cg * 4 + m]; sw[m] = L.sw[t][cg * 4 + m];
            }
            switch (deg) {
                #define GT_CASE(N) case N: moment_term<DM, N, ZG, ENV>(mom, g, p, q, L, t, D, cu, su, cw, sw, zr, zc); break;
                GT_DEGREE_CASES(GT_CASE)
#undef GT_CASE
                default: moment_term<DM, 0, ZG, ENV>(mom, g, p, q, L, t, D, cu, su, cw, sw, zr, zc); break;
            }
            // workgroup reduction of the W moments of this term through LDS, fixed order: thread (w, i) adds the values of threads
            // 16 i .. 16 i + 15, a 4-step butterfly over i finishes.  With two staging buffers the next term's writes need no second
            // barrier (a buffer is reused two terms later, behind the barrier of the term in between).
            double* rb = s_red[nred % NBUF];
            ++nred;
            for (int w = 0; w < W; ++w) rb[w * RSTRIDE + tid + (tid >> 4)] = mom[w];
            __syncthreads();
            if (tid < 16 * W) {
                const int w = tid >> 4, i = tid & 15;
                const double* src = rb + w * RSTRIDE + i * 17;
                double v = 0.0;
#pragma unroll
                for (int k = 0; k < 16; ++k) v += src[k];
#pragma unroll
                for (int off = 8; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
                if (i == 0) outp[(size_t)(t0 + t) * W + w] = v;
            }
            if (NBUF == 1) __syncthreads();
        }
    }

    if (ZG) {
        // rows: the 16 threads of a row group (cg = 0..15) are consecutive lanes -> butterfly; columns: the 16 row groups' values of a column
        // go through LDS and are added in row-group order.  The tile's sums land in ITS slot of the scratch (no atomics anywhere: the order
        // of every sum is fixed); k_gz_reduce adds the slots of a block.
        double* stage = s_red[0];                              // [16 row groups][64 columns]
        for (int d = 0; d < D; ++d) {
            __syncthreads();
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                double v = zr[m][d];
#pragma unroll
                for (int off = 8; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
                if (cg == 0) s_gr[d][rg * 4 + m] = v;
                stage[rg * MOGP_GT + cg * 4 + m] = zc[m][d];
            }
            __syncthreads();
            if (tid < MOGP_GT) {
                double v = 0.0;
#pragma unroll
                for (int r = 0; r < 16; ++r) v += stage[r * MOGP_GT + tid];
                s_gc[d][tid] = v;
            }
        }
        __syncthreads();
        double* slot_r = a.gzp + ((size_t)tl.rb * a.ncb + tl.cb) * D * MOGP_GT;
        double* slot_c = a.gzp + (size_t)a.nrb * a.ncb * D * MOGP_GT + ((size_t)tl.cb * a.nrb + tl.rb) * D * MOGP_GT;
        for (int idx = tid; idx < D * MOGP_GT; idx += 256) {
            const int d = idx / MOGP_GT, pnt = idx - d * MOGP_GT;
            if (a.gzr) slot_r[idx] = pnt < tl.nr ? s_gr[d][pnt] : 0.0;
            if (a.gzc) slot_c[idx] = pnt < tl.nc ? s_gc[d][pnt] : 0.0;
        }
    }
}

// ---- exact mode, D = 1, no envelope: the persistent, software-pipelined form (round 4) -------------------------------------------------------
// k_moments above is one workgroup per tile: descriptor -> inputs and adjoint entries -> staging are three DEPENDENT memory round trips (12 of
// the 16 us a tile takes), two tiles in flight per CU, the VALUs busy half the time.  Here one 512-thread workgroup per CU walks over tiles
// blockIdx.x, + gridDim.x, ...: everything tile i + 1 needs from memory (its 64 x 64 block of Kj^-1: 8 entries per thread, alpha, inputs, phase
// factors; the descriptor came one tile earlier still) is REQUESTED BEFORE tile i's arithmetic and lands under it -- in registers: with one
// workgroup per CU a wave may use 256 VGPRs, which the one-tile-per-workgroup form could not afford.  A thread owns 2 x 4 entries, accumulates
// the five moments of every term over the whole tile in registers, and the workgroup reduces them ONCE per tile (wave butterflies, then eight
// partial sums through LDS in wave order: fixed order, bit-reproducible) instead of once per term.  Same arithmetic per entry as moment_term.
template <int N>
__device__ __forceinline__ void mx_term(double (&mom)[5], const double (&g)[2][4], const double (&p)[2], const double (&q)[4], double V, double s,
                                        const double (&cu)[2], const double (&su)[2], const double (&cw)[4], const double (&sw)[4]) {
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        const double vp = V * p[m];
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            const double u = (p[m] - q[n]) + s;
            double e;
            if (N > 0) e = exp_taylor<N>(vp * q[n]);
            else e = exp(-0.5 * (V * u) * u);
            const double ge = g[m][n] * e;
            const double kc = ge * fma(cu[m], cw[n], su[m] * sw[n]);
            const double ks = ge * fma(su[m], cw[n], -cu[m] * sw[n]);
            mom[0] += kc;
            mom[1] += ks;
            const double uk = u * kc;
            mom[2] = fma(u, uk, mom[2]);
            mom[3] += uk;
            mom[4] = fma(u, ks, mom[4]);
        }
    }
}

#define MX_T 8                                             // terms at most (registers: 5 moments per term)
struct MxFetch {                                           // what a thread requests for a tile one iteration ahead
    double kv[2][4], ar[2], ac[4], xr[2], xc[4];
    StageItem<1> it[2];
    double cr, hr, cc, hc;
};

template <int TT>
__global__ __launch_bounds__(512, 2) void k_moments_x(MomentArgs a) {
    const int T = a.T, W = a.W, tid = threadIdx.x;
    const int cg = tid & 15, rg = tid >> 4;                  // rows 2 rg, 2 rg + 1; columns 4 cg .. 4 cg + 3
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int st_which = wave & 1, st_tb = wave >> 1, st_pnt = lane;      // staging: wave -> (rows | columns) of the terms wave / 2 and wave / 2 + 4
    __shared__ TileLds<1> Lb[2];
    constexpr int MXS = 512 + 32;                          // one moment of all threads, +1 per 16 (the slice reads hit distinct banks)
    __shared__ double s_red[TT * 5 * MXS];
    const PhaseView v = phase_view(a.ph.ws, a.C, T, 1, a.ldx, a.ldx);
    // round 5: a workgroup walks a CONTIGUOUS chunk of the tile list and keeps the moments in registers from tile to tile for as long as the
    // channel pair stays the same (the tiles of a pair are contiguous in the list: a chunk crosses a pair boundary once or twice); the LDS
    // reduction and its two barriers then happen once per run instead of once per tile, and the slots of the run's other tiles receive zeros
    // (k_moment_reduce adds the slots of a pair in fixed order, as before).  The counters had the kernel's waves parked at s_waitcnt / s_barrier
    // 41 % of their cycles (SQ_WAIT_ANY; profiles/r5_pmc_tile_kernels.txt), the vector ALUs busy half the time.
    const int per = (a.ntiles + (int)gridDim.x - 1) / (int)gridDim.x;
    int tile = blockIdx.x * per;
    const int tend = min(tile + per, a.ntiles);
    if (tile >= tend) return;

    auto fetch = [&](MxFetch& F, const GTile& tl) {
        F.cr = v.rcen[tl.r0]; F.hr = v.rhalf[tl.r0]; F.cc = v.ccen[tl.c0]; F.hc = v.chalf[tl.c0];
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const int64_t r = tl.r0 + min(2 * rg + m, tl.nr - 1);
            F.ar[m] = a.alpha[r];
            F.xr[m] = a.x[r];
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                const int64_t c = tl.c0 + min(4 * cg + n, tl.nc - 1);
                const int64_t hi = r > c ? r : c, lo = r > c ? c : r;      // the lower triangle holds the matrix
                F.kv[m][n] = a.kinv[hi * a.ld + lo];
            }
        }
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            const int64_t c = tl.c0 + min(4 * cg + n, tl.nc - 1);
            F.ac[n] = a.alpha[c];
            F.xc[n] = a.x[c];
        }
#pragma unroll
        for (int k = 0; k < 2; ++k)
            if (st_tb + 4 * k < T) stage_item_load<1>(F.it[k], st_which, st_tb + 4 * k, st_pnt, tl, 1, a.C, T, 0, v, a.x, a.ldx, a.x, a.ldx);
    };

    GTile tl = a.tiles[tile];
    GTile tl1 = a.tiles[min(tile + 1, tend - 1)];
    MxFetch F;
    fetch(F, tl);
    int buf = 0;
    double mom[TT][5];
#pragma unroll
    for (int t = 0; t < TT; ++t)
#pragma unroll
        for (int w = 0; w < 5; ++w) mom[t][w] = 0.0;
    while (true) {
        TileLds<1>& L = Lb[buf];
        const GTile cur = tl;
        const double* tab = a.table + (size_t)cur.pair * T * W;
        double* outp = a.partial + (size_t)tile * T * W;
        // ---- this tile out of the registers its requests landed in ----
        TileCtx<1> X;
        X.cr[0] = F.cr; X.hr[0] = F.hr; X.cc[0] = F.cc; X.hc[0] = F.hc;
#pragma unroll
        for (int k = 0; k < 2; ++k)
            if (st_tb + 4 * k < T) stage_item_compute<1, false>(L, F.it[k], X, tab, W, st_which, st_tb + 4 * k, st_pnt, 1, 0);
        double g[2][4], p[2], q[4];
#pragma unroll
        for (int n = 0; n < 4; ++n) q[n] = F.xc[n] - F.cc;
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            p[m] = F.xr[m] - F.cr;
            const int lr = 2 * rg + m;
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                const int lc = 4 * cg + n;
                double w = 0.0;
                if (lr < cur.nr && lc < cur.nc) {
                    w = 2.0;
                    if (cur.flags & GT_DIAG) w = lr > lc ? 2.0 : (lr == lc ? 1.0 : 0.0);
                }
                g[m][n] = w * 0.5 * (F.ar[m] * F.ac[n] - a.kinv_sign * F.kv[m][n]);
            }
        }
        // ---- the next tile's requests go out now, under this tile's arithmetic ----
        const int nxt = tile + 1;
        const bool more = nxt < tend;
        if (more) {
            tl = tl1;
            tl1 = a.tiles[min(nxt + 1, tend - 1)];
            fetch(F, tl);
        }
        const bool flush = !more || tl.pair != cur.pair;      // the run of this channel pair ends with this tile
        // LDS-only barrier: __syncthreads() would also wait for the requests just issued -- the whole point is that they stay in flight
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // the staged factors of this tile are in L
#pragma unroll
        for (int t = 0; t < TT; ++t) {
            if (t >= T) break;
            const int deg = __builtin_amdgcn_readfirstlane(L.deg[t]);
            if (deg == GT_SKIP) continue;
            const double V = L.V[t][0], sft = L.s[t][0];
            double cu[2], su[2], cw[4], sw[4];
#pragma unroll
            for (int m = 0; m < 2; ++m) { cu[m] = L.cu[t][2 * rg + m]; su[m] = L.su[t][2 * rg + m]; }
#pragma unroll
            for (int n = 0; n < 4; ++n) { cw[n] = L.cw[t][4 * cg + n]; sw[n] = L.sw[t][4 * cg + n]; }
            switch (deg) {
                #define GT_CASE(N) case N: mx_term<N>(mom[t], g, p, q, V, sft, cu, su, cw, sw); break;
                GT_DEGREE_CASES(GT_CASE)
#undef GT_CASE
                default: mx_term<0>(mom[t], g, p, q, V, sft, cu, su, cw, sw); break;
            }
        }
        if (!flush) {
            if (tid < T * 5) outp[tid] = 0.0;                 // this tile's moments ride on in registers: its slot adds nothing
        } else {
            // ---- one reduction per RUN through LDS, fixed order: thread (w, i) adds the values of threads 16 i .. 16 i + 15 of moment w, a
            // 5-step butterfly over i finishes (32 slices)
#pragma unroll
            for (int t = 0; t < TT; ++t) {
                if (t >= T) break;
#pragma unroll
                for (int w = 0; w < 5; ++w) { s_red[(t * 5 + w) * MXS + tid + (tid >> 4)] = mom[t][w]; mom[t][w] = 0.0; }
            }
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            for (int e = tid; e < T * 5 * 32; e += 512) {
                const int w = e >> 5, i = e & 31;
                const double* src = s_red + w * MXS + i * 17;
                double x = 0.0;
#pragma unroll
                for (int k = 0; k < 16; ++k) x += src[k];
#pragma unroll
                for (int off = 16; off > 0; off >>= 1) x += __shfl_xor(x, off, 64);
                if (i == 0) outp[w] = x;                     // (a skipped term left zeros)
            }
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // s_red is read: the next run may write it
        }
        if (!more) break;
        tile = nxt;
        buf ^= 1;                                            // the other staging buffer: its last readers passed the barrier above
    }
}

// out[d][first + pnt] += sum over the `nminor` slots of block b (slot layout [block][minor][d][64]), in slot order: four interleaved partial
// sums per point, combined as (s0 + s1) + (s2 + s3).  Slots no tile wrote are zero (the scratch is cleared per launch).
__global__ __launch_bounds__(256) void k_gz_reduce(const double* __restrict__ gzp, int nminor, int D, const int* __restrict__ blk,
                                                   double* __restrict__ out, int64_t ld) {
    const int b = blockIdx.x, pnt = threadIdx.x & 63, part = threadIdx.x >> 6;
    const int first = blk[2 * b], n = blk[2 * b + 1];
    __shared__ double red[4][MOGP_GT], redc[4][MOGP_GT];
    const double* base = gzp + (size_t)b * nminor * D * MOGP_GT;
    // The slots of one inducing point cancel to a small fraction of their size (at configs[4] the derivative with respect to an inducing input
    // is O(1e-2), the residue of sums many orders larger), so the long sums here are compensated (Neumaier): the result is the correctly
    // rounded sum of the slots to within a few ulp of the RESULT, whatever the number of slots.
    for (int d = 0; d < D; ++d) {
        double s = 0.0, c = 0.0;
        for (int k = part; k < nminor; k += 4) {
            const double x = base[((size_t)k * D + d) * MOGP_GT + pnt];
            const double t = s + x;
            c += fabs(s) >= fabs(x) ? (s - t) + x : (x - t) + s;
            s = t;
        }
        __syncthreads();
        red[part][pnt] = s; redc[part][pnt] = c;
        __syncthreads();
        if (part == 0 && pnt < n) {
            double hs = 0.0, hc = 0.0;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const double x = red[q][pnt];
                const double t = hs + x;
                hc += fabs(hs) >= fabs(x) ? (hs - t) + x : (x - t) + hs;
                hs = t;
                hc += redc[q][pnt];
            }
            out[(size_t)d * ld + first + pnt] += hs + hc;
        }
    }
}

template <bool DENSE, bool ZG, bool ENV>
static int launch_moments_t(const MomentArgs& a, hipStream_t s) {
    switch (a.D) {
        case 1: hipLaunchKernelGGL((k_moments<1, DENSE, ZG, ENV>), dim3(a.ntiles), dim3(256), 0, s, a); break;
        case 2: hipLaunchKernelGGL((k_moments<2, DENSE, ZG, ENV>), dim3(a.ntiles), dim3(256), 0, s, a); break;
        case 3: hipLaunchKernelGGL((k_moments<3, DENSE, ZG, ENV>), dim3(a.ntiles), dim3(256), 0, s, a); break;
        default: hipLaunchKernelGGL((k_moments<0, DENSE, ZG, ENV>), dim3(a.ntiles), dim3(256), 0, s, a); break;
    }
    HIP_TRY(hipGetLastError());
    return 0;
}

int launch_moments(const MomentArgs& a0, hipStream_t s) {
    if (a0.ntiles <= 0) return 0;
    MomentArgs a = a0;
    if (a.W <= 0) a.W = 2 + 3 * a.D;
    const bool env = a.W > 2 + 3 * a.D;
    const double* xc = a.xc ? a.xc : a.x;
    const int64_t ldxc = a.xc ? a.ldxc : a.ldx;
    int rc = a.phases_ready ? 0 : launch_phase_tables(a.ph, a.x, a.ldx, a.nrows, xc, ldxc, a.xc ? a.ncols : a.nrows, a.table, a.T, a.D, a.C, a.W, s);
    if (rc) return rc;
    if (a.G == nullptr) {
        if (a.ev0) HIP_TRY(hipEventRecord(a.ev0, s));
        // D = 1 without an envelope, every row this rank's own: the persistent pipelined kernel (MOGP_MOM_X=0: the tile-per-workgroup one)
        static const bool mx_on = !(std::getenv("MOGP_MOM_X") && std::atoi(std::getenv("MOGP_MOM_X")) == 0);
        if (mx_on && !env && a.D == 1 && a.W == 5 && a.T <= 4 && a.row_mod <= 1 && a.xc == nullptr) {
            static const int ncu = []() { int dev = 0; hipDeviceProp_t pr; if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&pr, dev) != hipSuccess) return 256; return pr.multiProcessorCount; }();
            // (the reduction staging of eight terms does not fit in LDS; three terms -- every BASELINE config of the exact model -- get an instantiation of their
            // own: the moments now live in registers across tiles, and five fewer accumulators keep the kernel clear of spills)
            if (a.T <= 3) hipLaunchKernelGGL(k_moments_x<3>, dim3(std::min(a.ntiles, ncu)), dim3(512), 0, s, a);
            else hipLaunchKernelGGL(k_moments_x<4>, dim3(std::min(a.ntiles, ncu)), dim3(512), 0, s, a);
            HIP_TRY(hipGetLastError());
            rc = 0;
        } else
        rc = env ? launch_moments_t<false, false, true>(a, s) : launch_moments_t<false, false, false>(a, s);
        if (!rc && a.ev1) HIP_TRY(hipEventRecord(a.ev1, s));
        return rc;
    }
    if (a.gzr || a.gzc) {
        if (!a.gzp || !a.rblk || !a.cblk || a.nrb <= 0 || a.ncb <= 0) { set_error("launch_moments: the input-gradient pass needs its scratch and block tables"); return -1; }
        const size_t half = (size_t)a.nrb * a.ncb * a.D * MOGP_GT;
        HIP_TRY(hipMemsetAsync(a.gzp, 0, 2 * half * sizeof(double), s));
        if ((rc = env ? launch_moments_t<true, true, true>(a, s) : launch_moments_t<true, true, false>(a, s))) return rc;
        if (a.gzr) hipLaunchKernelGGL(k_gz_reduce, dim3(a.nrb), dim3(256), 0, s, a.gzp, a.ncb, a.D, a.rblk, a.gzr, a.ldgz);
        if (a.gzc) hipLaunchKernelGGL(k_gz_reduce, dim3(a.ncb), dim3(256), 0, s, a.gzp + half, a.nrb, a.D, a.cblk, a.gzc, a.ldgz);
        HIP_TRY(hipGetLastError());
        return 0;
    }
    return env ? launch_moments_t<true, false, true>(a, s) : launch_moments_t<true, false, false>(a, s);
}

// one workgroup per (lower channel pair, moment entry): 256 threads stride over that pair's tiles, then a fixed-shape
// LDS tree -- the summation order depends only on the tile list, so results are bit-reproducible.
// On a diagonal channel block (i == j) the moments that are odd in tau (m4 = sum g E sin, m2_d = sum g u_d E cos)
// cancel between (a, b) and (b, a) in the full symmetric sum; the lower-triangle pass cannot see that, so they are
// set to their exact value, zero, here.
__global__ __launch_bounds__(256) void k_moment_reduce(const double* __restrict__ partial, const int* __restrict__ pair_start,
                                                       int TW, int W, int D, int lower_pairs, double* __restrict__ out) {
    const int p = blockIdx.x, tw = blockIdx.y;
    const int b = pair_start[p], e = pair_start[p + 1];
    __shared__ double red[256];
    double s = 0.0;
    for (int t = b + threadIdx.x; t < e; t += 256) s += partial[(size_t)t * TW + tw];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        int i = (int)((sqrt(8.0 * (double)p + 1.0) - 1.0) * 0.5);
        while ((i + 1) * (i + 2) / 2 <= p) ++i;
        while (i * (i + 1) / 2 > p) --i;
        const bool diag = lower_pairs && (p - i * (i + 1) / 2) == i;
        const int w = tw % W;
        double v = red[0];
        if (diag && (w == 1 || (w >= 2 + D && w < 2 + 2 * D))) v = 0.0;
        out[(size_t)p * TW + tw] = v;
    }
}

int launch_moment_reduce(const double* partial, const int* pair_start, int npairs, int T, int W, int D, double* out, hipStream_t s, int lower_pairs) {
    hipLaunchKernelGGL(k_moment_reduce, dim3(npairs, T * W), dim3(256), 0, s, partial, pair_start, T * W, W, D, lower_pairs, out);
    HIP_TRY(hipGetLastError());
    return 0;
}

// out[c] = sum_{k in channel c} 1/2 (alpha_k^2 - kinv_kk); one workgroup per channel
__global__ void k_diagG(const double* __restrict__ kinv, int64_t ld, const double* __restrict__ alpha,
                        const int* __restrict__ chan_off, double* __restrict__ out, double kinv_sign, int row_mod, int row_rem) {
    const int c = blockIdx.x;
    __shared__ double red[256];
    double s = 0.0;
    for (int64_t k = chan_off[c] + threadIdx.x; k < chan_off[c + 1]; k += 256)
        if (row_mod <= 1 || (int)((k / MOGP_TILE) % row_mod) == row_rem) s += 0.5 * (alpha[k] * alpha[k] - kinv_sign * kinv[k * ld + k]);
    red[threadIdx.x] = s;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[c] = red[0];
}

int launch_diagG(const double* kinv, int64_t ld, const double* alpha, const int* chan_off, int C, double* out, hipStream_t s,
                 double kinv_sign, int row_mod, int row_rem) {
    hipLaunchKernelGGL(k_diagG, dim3(C), dim3(256), 0, s, kinv, ld, alpha, chan_off, out, kinv_sign, row_mod, row_rem);
    HIP_TRY(hipGetLastError());
    return 0;
}

}  // namespace mogp
