// gram.hip -- spectral Gram builder and gradient-moment pass (gfx950).
//
// Every MOSM / SM / CSM channel-pair block is   K_ab = sum_t A_t exp(-1/2 sum_d V_td u_d^2) cos(2 pi (sum_d M_td u_d + Psi_t)),
// u_d = x_a,d - x_b,d + Delta_td   (reference: gpr/multioutput.py:182-204, :432-449; gpr/singleoutput.py:594-600).
// The cosine splits per point:  cos(2 pi (p_a - q_b)) = cos(2 pi p_a) cos(2 pi q_b) + sin(2 pi p_a) sin(2 pi q_b)  with
// p_a = sum_d M_d (x_a,d + Delta_d) + Psi,  q_b = sum_d M_d x_b,d, so a 64x64 tile needs 2*64*T sincos (staged in LDS)
// and one exp + a handful of FMAs per entry and term.  One workgroup = one 64x64 tile of one channel-pair block,
// 256 threads, 4x4 entries per thread; rows of the output are written in 32-byte runs (coalesced across 16 lanes).
#include "mogp_internal.h"

namespace mogp {

__device__ __forceinline__ double frac_turn(double p) { return p - rint(p); }

// stage per-point phase factors of one chunk of terms into LDS
template <int DM>
__device__ __forceinline__ void stage_phases(const double* __restrict__ tab, int W, int D, int t0, int nt, bool unit_amp,
                                             const double (*s_xr)[MOGP_GT], const double (*s_xc)[MOGP_GT],
                                             double (*s_cu)[MOGP_GT], double (*s_su)[MOGP_GT],
                                             double (*s_cw)[MOGP_GT], double (*s_sw)[MOGP_GT],
                                             double (*s_V)[DM], double (*s_Dl)[DM], int tid,
                                             double (*s_M)[DM] = nullptr, double* s_A = nullptr) {
    for (int idx = tid; idx < nt * MOGP_GT * 2; idx += 256) {
        const int which = idx / (nt * MOGP_GT);
        const int rem = idx - which * nt * MOGP_GT;
        const int t = rem / MOGP_GT, p = rem - t * MOGP_GT;
        const double* row = tab + (size_t)(t0 + t) * W;
        double ph = which == 0 ? row[1] : 0.0;
        for (int d = 0; d < D; ++d) {
            const double m = row[2 + D + d];
            ph = which == 0 ? fma(m, s_xr[d][p] + row[2 + 2 * D + d], ph) : fma(m, s_xc[d][p], ph);
        }
        double sn, cs;
        sincospi(2.0 * frac_turn(ph), &sn, &cs);
        if (which == 0) {
            const double amp = unit_amp ? 1.0 : row[0];
            s_cu[t][p] = amp * cs;
            s_su[t][p] = amp * sn;
        } else {
            s_cw[t][p] = cs;
            s_sw[t][p] = sn;
        }
    }
    for (int idx = tid; idx < nt * D; idx += 256) {
        const int t = idx / D, d = idx - t * D;
        const double* row = tab + (size_t)(t0 + t) * W;
        s_V[t][d] = row[2 + d];
        s_Dl[t][d] = row[2 + 2 * D + d];
        if (s_M) s_M[t][d] = row[2 + D + d];
        if (s_A && d == 0) s_A[t] = row[0];
    }
}

template <int DT>
__global__ __launch_bounds__(256) void k_gram(GramArgs a) {
    constexpr int DM = DT > 0 ? DT : MOGP_MAXD;
    const int D = DT > 0 ? DT : a.D;
    const int W = 2 + 3 * D;
    const GTile tl = a.tiles[blockIdx.x];
    const double* tab = a.table + (size_t)tl.pair * a.T * W;
    const int tid = threadIdx.x;
    const int cg = tid & 15, rg = tid >> 4;

    __shared__ double s_xr[DM][MOGP_GT], s_xc[DM][MOGP_GT];
    __shared__ double s_cu[MOGP_TC][MOGP_GT], s_su[MOGP_TC][MOGP_GT], s_cw[MOGP_TC][MOGP_GT], s_sw[MOGP_TC][MOGP_GT];
    __shared__ double s_V[MOGP_TC][DM], s_Dl[MOGP_TC][DM];

    for (int idx = tid; idx < D * MOGP_GT * 2; idx += 256) {
        const int which = idx / (D * MOGP_GT);
        const int rem = idx - which * D * MOGP_GT;
        const int d = rem / MOGP_GT, p = rem - d * MOGP_GT;
        if (which == 0) s_xr[d][p] = p < tl.nr ? a.xr[(size_t)d * a.ldxr + tl.r0 + p] : 0.0;
        else            s_xc[d][p] = p < tl.nc ? a.xc[(size_t)d * a.ldxc + tl.c0 + p] : 0.0;
    }
    bool any = false;
    for (int t = 0; t < a.T; ++t) any |= (tab[(size_t)t * W] != 0.0);

    double acc[4][4];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[m][n] = 0.0;

    if (any) {
        for (int t0 = 0; t0 < a.T; t0 += MOGP_TC) {
            const int nt = min(MOGP_TC, a.T - t0);
            __syncthreads();
            stage_phases<DM>(tab, W, D, t0, nt, false, s_xr, s_xc, s_cu, s_su, s_cw, s_sw, s_V, s_Dl, tid);
            __syncthreads();
            double xr[4][DM], xc[4][DM];
#pragma unroll
            for (int m = 0; m < 4; ++m)
                for (int d = 0; d < D; ++d) { xr[m][d] = s_xr[d][rg * 4 + m]; xc[m][d] = s_xc[d][cg * 4 + m]; }
            for (int t = 0; t < nt; ++t) {
                double cu[4], su[4], cw[4], sw[4];
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    cu[m] = s_cu[t][rg * 4 + m]; su[m] = s_su[t][rg * 4 + m];
                    cw[m] = s_cw[t][cg * 4 + m]; sw[m] = s_sw[t][cg * 4 + m];
                }
#pragma unroll
                for (int m = 0; m < 4; ++m)
#pragma unroll
                    for (int n = 0; n < 4; ++n) {
                        double arg = 0.0;
                        for (int d = 0; d < D; ++d) {
                            const double u = (xr[m][d] - xc[n][d]) + s_Dl[t][d];
                            arg = fma(s_V[t][d] * u, u, arg);
                        }
                        const double e = exp(-0.5 * arg);
                        acc[m][n] = fma(e, fma(cu[m], cw[n], su[m] * sw[n]), acc[m][n]);
                    }
            }
        }
    }

    // epilogue: diagonal augmentation + stores
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
            if (a.mirror && ((tl.flags & GT_MIRROR) || ((tl.flags & GT_DIAG) && lr > lc))) a.out[c * a.ldo + r] = v;
        }
    }
}

int launch_gram(const GramArgs& a, int ntiles, hipStream_t s) {
    if (ntiles <= 0) return 0;
    switch (a.D) {
        case 1: hipLaunchKernelGGL(k_gram<1>, dim3(ntiles), dim3(256), 0, s, a); break;
        case 2: hipLaunchKernelGGL(k_gram<2>, dim3(ntiles), dim3(256), 0, s, a); break;
        case 3: hipLaunchKernelGGL(k_gram<3>, dim3(ntiles), dim3(256), 0, s, a); break;
        default: hipLaunchKernelGGL(k_gram<0>, dim3(ntiles), dim3(256), 0, s, a); break;
    }
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
template <int DT, bool DENSE, bool ZG>
__global__ __launch_bounds__(256) void k_moments(MomentArgs a) {
    constexpr int DM = DT > 0 ? DT : MOGP_MAXD;
    constexpr int WM = 2 + 3 * DM;
    const int D = DT > 0 ? DT : a.D;
    const int W = 2 + 3 * D;
    const GTile tl = a.tiles[blockIdx.x];
    const double* tab = a.table + (size_t)tl.pair * a.T * W;
    const int tid = threadIdx.x;
    const int cg = tid & 15, rg = tid >> 4;
    const int lane = tid & 63, wave = tid >> 6;
    const double* xcol = a.xc ? a.xc : a.x;
    const int64_t ldxc = a.xc ? a.ldxc : a.ldx;

    __shared__ double s_xr[DM][MOGP_GT], s_xc[DM][MOGP_GT];
    __shared__ double s_cu[MOGP_TC][MOGP_GT], s_su[MOGP_TC][MOGP_GT], s_cw[MOGP_TC][MOGP_GT], s_sw[MOGP_TC][MOGP_GT];
    __shared__ double s_V[MOGP_TC][DM], s_Dl[MOGP_TC][DM], s_M[MOGP_TC][DM], s_A[MOGP_TC];
    __shared__ double s_red[4][WM];
    __shared__ double s_gr[ZG ? DM : 1][MOGP_GT], s_gc[ZG ? DM : 1][MOGP_GT];

    double* outp = a.partial + (size_t)blockIdx.x * a.T * W;
    bool any = false;
    for (int t = 0; t < a.T; ++t) any |= (tab[(size_t)t * W] != 0.0);
    if (!any) {
        for (int idx = tid; idx < a.T * W; idx += 256) outp[idx] = 0.0;
        return;
    }

    for (int idx = tid; idx < D * MOGP_GT * 2; idx += 256) {
        const int which = idx / (D * MOGP_GT);
        const int rem = idx - which * D * MOGP_GT;
        const int d = rem / MOGP_GT, p = rem - d * MOGP_GT;
        if (which == 0) s_xr[d][p] = p < tl.nr ? a.x[(size_t)d * a.ldx + tl.r0 + p] : 0.0;
        else            s_xc[d][p] = p < tl.nc ? xcol[(size_t)d * ldxc + tl.c0 + p] : 0.0;
    }
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

    for (int t0 = 0; t0 < a.T; t0 += MOGP_TC) {
        const int nt = min(MOGP_TC, a.T - t0);
        __syncthreads();
        stage_phases<DM>(tab, W, D, t0, nt, true, s_xr, s_xc, s_cu, s_su, s_cw, s_sw, s_V, s_Dl, tid, s_M, s_A);
        __syncthreads();
        double xr[4][DM], xc[4][DM];
#pragma unroll
        for (int m = 0; m < 4; ++m)
            for (int d = 0; d < D; ++d) { xr[m][d] = s_xr[d][rg * 4 + m]; xc[m][d] = s_xc[d][cg * 4 + m]; }
        for (int t = 0; t < nt; ++t) {
            double mom[WM];
            for (int w = 0; w < W; ++w) mom[w] = 0.0;
            double cu[4], su[4], cw[4], sw[4];
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                cu[m] = s_cu[t][rg * 4 + m]; su[m] = s_su[t][rg * 4 + m];
                cw[m] = s_cw[t][cg * 4 + m]; sw[m] = s_sw[t][cg * 4 + m];
            }
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int n = 0; n < 4; ++n) {
                    double u[DM];
                    double arg = 0.0;
                    for (int d = 0; d < D; ++d) {
                        u[d] = (xr[m][d] - xc[n][d]) + s_Dl[t][d];
                        arg = fma(s_V[t][d] * u[d], u[d], arg);
                    }
                    const double ge = g[m][n] * exp(-0.5 * arg);
                    const double kc = ge * fma(cu[m], cw[n], su[m] * sw[n]);
                    const double ks = ge * fma(su[m], cw[n], -cu[m] * sw[n]);
                    mom[0] += kc;
                    mom[1] += ks;
                    for (int d = 0; d < D; ++d) {
                        mom[2 + d] = fma(u[d] * u[d], kc, mom[2 + d]);
                        mom[2 + D + d] = fma(u[d], kc, mom[2 + D + d]);
                        mom[2 + 2 * D + d] = fma(u[d], ks, mom[2 + 2 * D + d]);
                        if (ZG) {
                            const double j = -s_A[t] * fma(s_V[t][d] * u[d], kc, 6.283185307179586476925286766559 * s_M[t][d] * ks);
                            zr[m][d] += j;
                            zc[n][d] -= j;
                        }
                    }
                }
            // workgroup reduction (fixed order: butterfly inside the wave, then waves 0..3)
            for (int w = 0; w < W; ++w) {
                double v = mom[w];
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
                if (lane == 0) s_red[wave][w] = v;
            }
            __syncthreads();
            if (tid < W) outp[(size_t)(t0 + t) * W + tid] = (s_red[0][tid] + s_red[1][tid]) + (s_red[2][tid] + s_red[3][tid]);
            __syncthreads();
        }
    }

    if (ZG) {
        // rows: the 16 threads of a row group (cg = 0..15) are consecutive lanes -> butterfly, then one LDS add;
        // columns: LDS atomics; finally one global atomic per point and dimension
#pragma unroll
        for (int m = 0; m < 4; ++m)
            for (int d = 0; d < D; ++d) {
                double v = zr[m][d];
#pragma unroll
                for (int off = 8; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
                if (cg == 0) s_gr[d][rg * 4 + m] = v;
                atomicAdd(&s_gc[d][cg * 4 + m], zc[m][d]);
            }
        __syncthreads();
        for (int idx = tid; idx < D * MOGP_GT; idx += 256) {
            const int d = idx / MOGP_GT, p = idx - d * MOGP_GT;
            if (a.gzr && p < tl.nr) atomicAdd(&a.gzr[(size_t)d * a.ldgz + tl.r0 + p], s_gr[d][p]);
            if (a.gzc && p < tl.nc) atomicAdd(&a.gzc[(size_t)d * a.ldgz + tl.c0 + p], s_gc[d][p]);
        }
    }
}

template <bool DENSE, bool ZG>
static int launch_moments_t(const MomentArgs& a, hipStream_t s) {
    switch (a.D) {
        case 1: hipLaunchKernelGGL((k_moments<1, DENSE, ZG>), dim3(a.ntiles), dim3(256), 0, s, a); break;
        case 2: hipLaunchKernelGGL((k_moments<2, DENSE, ZG>), dim3(a.ntiles), dim3(256), 0, s, a); break;
        case 3: hipLaunchKernelGGL((k_moments<3, DENSE, ZG>), dim3(a.ntiles), dim3(256), 0, s, a); break;
        default: hipLaunchKernelGGL((k_moments<0, DENSE, ZG>), dim3(a.ntiles), dim3(256), 0, s, a); break;
    }
    HIP_TRY(hipGetLastError());
    return 0;
}

int launch_moments(const MomentArgs& a, hipStream_t s) {
    if (a.ntiles <= 0) return 0;
    if (a.G == nullptr) return launch_moments_t<false, false>(a, s);
    if (a.gzr || a.gzc) return launch_moments_t<true, true>(a, s);
    return launch_moments_t<true, false>(a, s);
}

// one workgroup per (lower channel pair, moment entry): 256 threads stride over that pair's tiles, then a fixed-shape
// LDS tree -- the summation order depends only on the tile list, so results are bit-reproducible.
// On a diagonal channel block (i == j) the moments that are odd in tau (m4 = sum g E sin, m2_d = sum g u_d E cos)
// cancel between (a, b) and (b, a) in the full symmetric sum; the lower-triangle pass cannot see that, so they are
// set to their exact value, zero, here.
__global__ __launch_bounds__(256) void k_moment_reduce(const double* __restrict__ partial, const int* __restrict__ pair_start,
                                                       int TW, int W, int lower_pairs, double* __restrict__ out) {
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
        const int D = (W - 2) / 3, w = tw % W;
        double v = red[0];
        if (diag && (w == 1 || (w >= 2 + D && w < 2 + 2 * D))) v = 0.0;
        out[(size_t)p * TW + tw] = v;
    }
}

int launch_moment_reduce(const double* partial, const int* pair_start, int npairs, int T, int W, double* out, hipStream_t s, int lower_pairs) {
    hipLaunchKernelGGL(k_moment_reduce, dim3(npairs, T * W), dim3(256), 0, s, partial, pair_start, T * W, W, lower_pairs, out);
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
