// trsm.hip -- triangular solves with the Cholesky factor itself: X = L^-1 B and X = L^-T B by blocked SUBSTITUTION.
//
// Why: the Titsias bound differentiates through K_uu^-1 with cond(K_uu) ~ 1e11 at BASELINE configs[4] (512 grid inducing points per
// channel, 0.2 apart).  The reference uses torch.linalg.solve_triangular (gpr/model.py:711,715,746-748) -- backward stable; products
// with an explicit inverse factor W = L^-1 are not: W B carries an error ~eps |W| |B| although the solution L^-1 B is 1e5 times
// smaller than that, and dELBO/dZ (the residue of O(1e4) terms cancelling) comes out with 12-19 % error.  With substitution it matches
// an 80-bit evaluation to 4e-5 (tools/titsias_numerics.py), the level the reference itself reaches.  Explicit inverses of the 128 x 128
// diagonal tiles are NOT enough (their own condition number is ~1e10): the substitution has to go down to single rows.
//
// Blocked left-looking form over 128-row blocks:  B_i -= L[i, <i] X[<i]  on the fp64 MFMA GEMM (k_gemm, K growing with i), then
// X_i = L_ii^-1 B_i in k_trsm_leaf: one COLUMN of the right-hand side per thread (no cross-thread dependency at all), the 128 x 128 tile of
// L in LDS (every read is a wave-wide broadcast), forward / backward substitution in 16-row sub-blocks: the current 16 unknowns in
// registers, the already solved sub-blocks re-read from the (L1 / L2 resident) right-hand side.  2 * 8256 flops per column and block row:
// 1.3e10 flops for the 2048 x 100 000 solve of configs[4] next to 4.2e11 in the GEMMs.
#include "mogp_model.h"

using namespace mogp;

#define RC(x) do { int r__ = (x); if (r__) return r__; } while (0)
#define TS_T 128
#define TS_SB 16
#define TS_LDS_BYTES ((TS_T * TS_T + TS_T) * 8)

typedef double d2_t __attribute__((ext_vector_type(2)));

// X_i = L_ii^-1 B_i (TRANS: L_ii^-T B_i), in place in B (rows r0 .. r0+127), one column per thread; 16-row sub-blocks (32-row ones with
// software-prefetched re-reads were measured slower: 256 VGPRs, 214 vs 78 us)
template <bool TRANS, int THREADS>
__global__ __launch_bounds__(THREADS) void k_trsm_leaf(const double* __restrict__ Lt, int64_t ldl, double* __restrict__ B, int64_t ldb, int64_t ncols) {
    extern __shared__ __attribute__((aligned(16))) double ts_lds[];
    double* Ls = ts_lds;                  // [128][128] row-major (upper part never read)
    double* inv = ts_lds + TS_T * TS_T;   // [128] 1 / L_kk
    const int tid = threadIdx.x;
    for (int idx = tid; idx < TS_T * TS_T / 2; idx += THREADS) {
        const int r = idx >> 6, c = (idx & 63) * 2;
        *reinterpret_cast<d2_t*>(Ls + r * TS_T + c) = *reinterpret_cast<const d2_t*>(Lt + (int64_t)r * ldl + c);
    }
    if (tid < TS_T) inv[tid] = 1.0 / Lt[(int64_t)tid * ldl + tid];
    if (THREADS < TS_T) inv[tid + 64] = 1.0 / Lt[(int64_t)(tid + 64) * ldl + tid + 64];
    __syncthreads();
    const int64_t col = (int64_t)blockIdx.x * THREADS + tid;
    if (col >= ncols) return;
    double* b = B + col;

    if (!TRANS) {
        for (int sb = 0; sb < TS_T / TS_SB; ++sb) {
            double x[TS_SB];
#pragma unroll
            for (int r = 0; r < TS_SB; ++r) x[r] = b[(int64_t)(sb * TS_SB + r) * ldb];
            // x -= L[sb][pb] x_pb over the solved sub-blocks, re-read from the (L1 / L2 resident) right-hand side ONE SUB-BLOCK AHEAD: the 16
            // loads of sub-block pb + 1 fly under the 256 FMAs of sub-block pb (unpipelined, each of the 120 (sb, pb) pairs of a tile paid a
            // cache round trip: 150 us per block row at configs[4], of which ~35 are arithmetic)
            auto fetch = [&](double (&xp)[TS_SB], int pb) {
#pragma unroll
                for (int c = 0; c < TS_SB; ++c) xp[c] = b[(int64_t)(pb * TS_SB + c) * ldb];
            };
            auto apply = [&](const double (&xp)[TS_SB], int pb) {
                const double* Lb = Ls + (sb * TS_SB) * TS_T + pb * TS_SB;
#pragma unroll
                for (int r = 0; r < TS_SB; ++r)
#pragma unroll
                    for (int c = 0; c < TS_SB; c += 2) {
                        const d2_t l = *reinterpret_cast<const d2_t*>(Lb + r * TS_T + c);
                        x[r] = fma(-l[0], xp[c], x[r]);
                        x[r] = fma(-l[1], xp[c + 1], x[r]);
                    }
            };
            double xa[TS_SB], xb[TS_SB];
            if (sb > 0) fetch(xa, 0);
            for (int pb = 0; pb < sb; pb += 2) {
                if (pb + 1 < sb) fetch(xb, pb + 1);
                apply(xa, pb);
                if (pb + 1 < sb) {
                    if (pb + 2 < sb) fetch(xa, pb + 2);
                    apply(xb, pb + 1);
                }
            }
            const double* Ld = Ls + (sb * TS_SB) * TS_T + sb * TS_SB;    // diagonal sub-block: forward substitution
#pragma unroll
            for (int c = 0; c < TS_SB; ++c) {
                x[c] *= inv[sb * TS_SB + c];
#pragma unroll
                for (int r = c + 1; r < TS_SB; ++r) x[r] = fma(-Ld[r * TS_T + c], x[c], x[r]);
            }
#pragma unroll
            for (int r = 0; r < TS_SB; ++r) b[(int64_t)(sb * TS_SB + r) * ldb] = x[r];
        }
    } else {
        for (int sb = TS_T / TS_SB - 1; sb >= 0; --sb) {
            double x[TS_SB];
#pragma unroll
            for (int r = 0; r < TS_SB; ++r) x[r] = b[(int64_t)(sb * TS_SB + r) * ldb];
            // x -= L[pb][sb]^T x_pb, the solved sub-blocks re-read one ahead (see the forward form)
            auto fetch = [&](double (&xp)[TS_SB], int pb) {
#pragma unroll
                for (int c = 0; c < TS_SB; ++c) xp[c] = b[(int64_t)(pb * TS_SB + c) * ldb];
            };
            auto apply = [&](const double (&xp)[TS_SB], int pb) {
                const double* Lb = Ls + (pb * TS_SB) * TS_T + sb * TS_SB;
#pragma unroll
                for (int c = 0; c < TS_SB; ++c)
#pragma unroll
                    for (int r = 0; r < TS_SB; r += 2) {
                        const d2_t l = *reinterpret_cast<const d2_t*>(Lb + c * TS_T + r);
                        x[r] = fma(-l[0], xp[c], x[r]);
                        x[r + 1] = fma(-l[1], xp[c], x[r + 1]);
                    }
            };
            constexpr int LAST = TS_T / TS_SB - 1;
            double xa[TS_SB], xb[TS_SB];
            if (LAST > sb) fetch(xa, LAST);
            for (int pb = LAST; pb > sb; pb -= 2) {
                if (pb - 1 > sb) fetch(xb, pb - 1);
                apply(xa, pb);
                if (pb - 1 > sb) {
                    if (pb - 2 > sb) fetch(xa, pb - 2);
                    apply(xb, pb - 1);
                }
            }
            const double* Ld = Ls + (sb * TS_SB) * TS_T + sb * TS_SB;    // diagonal sub-block: backward substitution with L^T
#pragma unroll
            for (int c = TS_SB - 1; c >= 0; --c) {
                x[c] *= inv[sb * TS_SB + c];
#pragma unroll
                for (int r = 0; r < c; ++r) x[r] = fma(-Ld[c * TS_T + r], x[c], x[r]);
            }
#pragma unroll
            for (int r = 0; r < TS_SB; ++r) b[(int64_t)(sb * TS_SB + r) * ldb] = x[r];
        }
    }
}

template <bool TRANS, int THREADS>
static int launch_leaf_t(const double* Lt, int64_t ldl, double* B, int64_t ldb, int64_t ncols, hipStream_t s) {
    static std::atomic<unsigned long long> attr_done{0ull};                 // one bit per device
    { int r__ = set_max_dynamic_lds(reinterpret_cast<const void*>(k_trsm_leaf<TRANS, THREADS>), TS_LDS_BYTES, attr_done); if (r__) return r__; }
    hipLaunchKernelGGL((k_trsm_leaf<TRANS, THREADS>), dim3((unsigned)((ncols + THREADS - 1) / THREADS)), dim3(THREADS), TS_LDS_BYTES, s, Lt, ldl, B, ldb, ncols);
    HIP_TRY(hipGetLastError());
    return 0;
}
// the 132 KB tile image allows one workgroup per CU: 512 threads (two waves per SIMD cover each other's load latency, and 100 000 columns
// are one round of 196 workgroups instead of two rounds of 391) for wide right-hand sides, one wave per workgroup for narrow ones (more CUs)
template <bool TRANS>
static int launch_leaf(const double* Lt, int64_t ldl, double* B, int64_t ldb, int64_t ncols, hipStream_t s) {
    if (ncols >= 65536) return launch_leaf_t<TRANS, 512>(Lt, ldl, B, ldb, ncols, s);
    return launch_leaf_t<TRANS, 64>(Lt, ldl, B, ldb, ncols, s);
}

__global__ __launch_bounds__(256) void k_transpose(double* __restrict__ dst, const double* __restrict__ src, int64_t ld) {
    __shared__ double t[64][65];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int64_t r0 = (int64_t)blockIdx.y * 64, c0 = (int64_t)blockIdx.x * 64;
    for (int r = ty; r < 64; r += 4) t[r][tx] = src[(r0 + r) * ld + c0 + tx];
    __syncthreads();
    for (int r = ty; r < 64; r += 4) dst[(c0 + r) * ld + r0 + tx] = t[tx][r];
}

// lower(A) <- scale * (A + A^T) / 2: one workgroup per lower 64 x 64 tile pair
__global__ __launch_bounds__(256) void k_sym_lower_avg(double* __restrict__ A, int64_t ld, double scale) {
    __shared__ double t[64][65];
    int b = blockIdx.x, ti = 0;
    while (b > ti) { b -= ti + 1; ++ti; }
    const int tj = b, tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int r = ty; r < 64; r += 4) t[r][tx] = A[(int64_t)(tj * 64 + r) * ld + ti * 64 + tx];       // the mirrored tile (tj, ti)
    __syncthreads();
    for (int r = ty; r < 64; r += 4) {
        double* a = A + (int64_t)(ti * 64 + r) * ld + tj * 64 + tx;
        *a = 0.5 * scale * (*a + t[tx][r]);
    }
}

namespace mogp {

int launch_transpose(double* dst, const double* src, int64_t ld, int64_t n, hipStream_t s) {
    hipLaunchKernelGGL(k_transpose, dim3((unsigned)(n / 64), (unsigned)(n / 64)), dim3(256), 0, s, dst, src, ld);
    HIP_TRY(hipGetLastError());
    return 0;
}

int launch_sym_lower_avg(double* A, int64_t ld, int64_t n, double scale, hipStream_t s) {
    const int nt = (int)(n / 64);
    hipLaunchKernelGGL(k_sym_lower_avg, dim3(nt * (nt + 1) / 2), dim3(256), 0, s, A, ld, scale);
    HIP_TRY(hipGetLastError());
    return 0;
}

int trsm_lower(mogp_model* m, const double* L, int64_t ldl, int nb, double* B, int64_t ldb, int64_t ncols, bool trans, hipStream_t st) {
    if (!st) st = m->st;
    if (ncols % MOGP_TILE) return fail(MOGP_EINVAL, "trsm_lower: the number of right-hand sides must be a multiple of 128");
    const int nt = (int)(ncols / MOGP_TILE);
    // Wide right-hand sides (N columns): left-looking -- block row i is updated once, by one GEMM with K = 128 i, and B is swept once.
    // Narrow ones (M x M): right-looking -- every solved block updates all remaining block rows at once (K = 128, but (nb - i) nt
    // workgroups per launch instead of nt; the matrix stays in the Infinity Cache).
    const bool right = nt <= 32;
    for (int step = 0; step < nb; ++step) {
        const int i = trans ? nb - 1 - step : step;
        double* Bi = B + (int64_t)i * MOGP_TILE * ldb;
        if (!right && step > 0) {
            GemmArgs g{};
            if (!trans) {            // B_i -= L[i, 0:i] X[0:i]
                g.A = L + (int64_t)i * MOGP_TILE * ldl; g.lda = ldl; g.a_kmajor = 0;
                g.B = B; g.ldb = ldb; g.b_kmajor = 1;
            } else {                 // B_i -= L[i+1:, i]^T X[i+1:]
                g.A = L + (int64_t)(i + 1) * MOGP_TILE * ldl + (int64_t)i * MOGP_TILE; g.lda = ldl; g.a_kmajor = 1;
                g.B = B + (int64_t)(i + 1) * MOGP_TILE * ldb; g.ldb = ldb; g.b_kmajor = 1;
            }
            g.C = Bi; g.ldc = ldb; g.alpha = -1.0; g.beta = 1.0;
            g.mode = GM_RECT; g.mt = 1; g.nt = nt; g.K = step * MOGP_TILE;
            g.sk_hint = 1;           // nt tiles per launch (782 at N = 100000: 1.53 rounds of the chip), nothing else running: stream-K form
            RC(gemm_call(m, g, gemm_flops(g, nullptr), st));
        }
        const double* Lii = L + (int64_t)i * MOGP_TILE * (ldl + 1);
        RC(trans ? launch_leaf<true>(Lii, ldl, Bi, ldb, ncols, st) : launch_leaf<false>(Lii, ldl, Bi, ldb, ncols, st));
        const int rest = nb - 1 - step;
        if (right && rest > 0) {
            GemmArgs g{};
            if (!trans) {            // B[i+1:] -= L[i+1:, i] X_i
                g.A = L + (int64_t)(i + 1) * MOGP_TILE * ldl + (int64_t)i * MOGP_TILE; g.lda = ldl; g.a_kmajor = 0;
                g.C = B + (int64_t)(i + 1) * MOGP_TILE * ldb;
            } else {                 // B[0:i] -= L[i, 0:i]^T X_i
                g.A = L + (int64_t)i * MOGP_TILE * ldl; g.lda = ldl; g.a_kmajor = 1;
                g.C = B;
            }
            g.B = Bi; g.ldb = ldb; g.b_kmajor = 1;
            g.ldc = ldb; g.alpha = -1.0; g.beta = 1.0;
            g.mode = GM_RECT; g.mt = rest; g.nt = nt; g.K = MOGP_TILE;
            RC(gemm_call(m, g, gemm_flops(g, nullptr), st));
        }
    }
    return 0;
}

}  // namespace mogp
