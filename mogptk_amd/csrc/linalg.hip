// linalg.hip -- fp64 dense linear algebra for the exact-GP path on gfx950:
//   * k_gemm            : 128x128-tile GEMM on v_mfma_f64_16x16x4_f64, four operand layouts, tile-grid decodes for the
//                         shapes the factorisation needs (panel solve, SYRK trailing update, TRTRI levels, LAUUM, TRMM)
//   * k_potrf_trtri_tile: register-resident Cholesky of one 128x128 diagonal tile + its inverse (rank-1 sweeps, one
//                         LDS-published column/row per step)
//   * small bandwidth-bound vector kernels (triangular mat-vec for alpha, row reductions for the predictive variance)
// These replace torch.linalg.cholesky / cholesky_solve / solve_triangular at reference gpr/model.py:246,452,470-472 and
// the O(N^3) dense solves of their autograd backward nodes.
#include "mogp_internal.h"

namespace mogp {

typedef double d4_t __attribute__((ext_vector_type(4)));
typedef double d2_t __attribute__((ext_vector_type(2)));

#define GEMM_BK 16
#define LDS_ROWK 18     // [128][18] doubles: row stride == 2 (mod 32) in 8-byte units -> conflict-free ds_read_b64 fragments
#define LDS_COLK 144    // [16][144] doubles: row stride == 16 (mod 32)
#define LDS_OPER 2304   // doubles per operand per buffer (128*18 == 16*144)
#define GEMM_LDS_BYTES (2 * 2 * LDS_OPER * 8)

__device__ __forceinline__ int tri_row(int b) {
    int r = (int)((sqrt(8.0 * (double)b + 1.0) - 1.0) * 0.5);
    while ((r + 1) * (r + 2) / 2 <= b) ++r;
    while (r * (r + 1) / 2 > b) --r;
    return r;
}

// C(128x128 tile) = alpha * sum_k A[i,k] B[j,k] + beta * C.   4 waves as 2x2, each wave 64x64 = 4x4 MFMA tiles.
template <int AKM, int BKM>
__global__ __launch_bounds__(256, 2) void k_gemm(GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) double gemm_lds[];     // [2 buffers][2 operands][LDS_OPER]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wi = wave >> 1, wj = wave & 1;

    // ---- which tile, which k range ----
    const double* Ap;
    const double* Bp;
    double* Cp;
    int kt;
    if (g.mode == GM_TASKS) {
        const GemmTask t = g.tasks[blockIdx.x];
        Ap = g.A + t.a_off; Bp = g.B + t.b_off; Cp = g.C + t.c_off; kt = t.kt;
    } else {
        int ti, tj;
        if (g.mode == GM_RECT || g.mode == GM_KHI_J) { ti = blockIdx.x / g.nt; tj = blockIdx.x - ti * g.nt; }
        else { ti = tri_row(blockIdx.x); tj = blockIdx.x - ti * (ti + 1) / 2; }
        int64_t k0 = 0, k1 = g.K;
        if (g.mode == GM_LAUUM) k0 = (int64_t)ti * MOGP_TILE;
        if (g.mode == GM_KHI_J) k1 = min((int64_t)g.K, (int64_t)(tj + 1) * MOGP_TILE);
        kt = (int)((k1 - k0) / GEMM_BK);
        Ap = g.A + (AKM ? k0 * g.lda + (int64_t)ti * MOGP_TILE : (int64_t)ti * MOGP_TILE * g.lda + k0);
        Bp = g.B + (BKM ? k0 * g.ldb + (int64_t)tj * MOGP_TILE : (int64_t)tj * MOGP_TILE * g.ldb + k0);
        Cp = g.C + (int64_t)ti * MOGP_TILE * g.ldc + (int64_t)tj * MOGP_TILE;
    }

    // ---- global -> register staging map: 8 doubles (4 x 16 B) per thread per operand ----
    // k-contiguous operand: thread -> (row = tid/2, 8 k's);  k-major operand: thread -> (k row = tid/16, 8 i's)
    const int64_t a_g = AKM ? (int64_t)(tid >> 4) * g.lda + (tid & 15) * 8 : (int64_t)(tid >> 1) * g.lda + (tid & 1) * 8;
    const int64_t b_g = BKM ? (int64_t)(tid >> 4) * g.ldb + (tid & 15) * 8 : (int64_t)(tid >> 1) * g.ldb + (tid & 1) * 8;
    const int a_l = AKM ? (tid >> 4) * LDS_COLK + (tid & 15) * 8 : (tid >> 1) * LDS_ROWK + (tid & 1) * 8;
    const int b_l = BKM ? (tid >> 4) * LDS_COLK + (tid & 15) * 8 : (tid >> 1) * LDS_ROWK + (tid & 1) * 8;
    const int64_t a_step = AKM ? (int64_t)GEMM_BK * g.lda : GEMM_BK;
    const int64_t b_step = BKM ? (int64_t)GEMM_BK * g.ldb : GEMM_BK;

    d4_t acc[4][4];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[m][n] = (d4_t){0.0, 0.0, 0.0, 0.0};

    d2_t ra[4], rb[4];
    if (kt > 0) {
        const d2_t* pa = reinterpret_cast<const d2_t*>(Ap + a_g);
        const d2_t* pb = reinterpret_cast<const d2_t*>(Bp + b_g);
#pragma unroll
        for (int q = 0; q < 4; ++q) { ra[q] = pa[q]; rb[q] = pb[q]; }
    }
    // fragment read offsets inside an operand buffer (per m / n add 16 rows)
    const int fa = AKM ? (lane >> 4) * LDS_COLK + wi * 64 + (lane & 15) : (wi * 64 + (lane & 15)) * LDS_ROWK + (lane >> 4);
    const int fb = BKM ? (lane >> 4) * LDS_COLK + wj * 64 + (lane & 15) : (wj * 64 + (lane & 15)) * LDS_ROWK + (lane >> 4);
    const int fa_m = AKM ? 16 : 16 * LDS_ROWK, fa_k = AKM ? 4 * LDS_COLK : 4;
    const int fb_n = BKM ? 16 : 16 * LDS_ROWK, fb_k = BKM ? 4 * LDS_COLK : 4;

    for (int kb = 0; kb < kt; ++kb) {
        double* sa = gemm_lds + (kb & 1) * 2 * LDS_OPER;
        double* sb = sa + LDS_OPER;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            *reinterpret_cast<d2_t*>(sa + a_l + 2 * q) = ra[q];
            *reinterpret_cast<d2_t*>(sb + b_l + 2 * q) = rb[q];
        }
        __syncthreads();
        if (kb + 1 < kt) {
            const d2_t* pa = reinterpret_cast<const d2_t*>(Ap + a_g + (int64_t)(kb + 1) * a_step);
            const d2_t* pb = reinterpret_cast<const d2_t*>(Bp + b_g + (int64_t)(kb + 1) * b_step);
#pragma unroll
            for (int q = 0; q < 4; ++q) { ra[q] = pa[q]; rb[q] = pb[q]; }
        }
#pragma unroll
        for (int k4 = 0; k4 < GEMM_BK / 4; ++k4) {
            double av[4], bv[4];
#pragma unroll
            for (int m = 0; m < 4; ++m) av[m] = sa[fa + m * fa_m + k4 * fa_k];
#pragma unroll
            for (int n = 0; n < 4; ++n) bv[n] = sb[fb + n * fb_n + k4 * fb_k];
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int n = 0; n < 4; ++n)
                    acc[m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[m], bv[n], acc[m][n], 0, 0, 0);
        }
    }

    // ---- epilogue: C/D layout of v_mfma_f64_16x16x4_f64: col = lane & 15, row = (lane >> 4) + 4 * reg ----
    const int crow = wi * 64 + (lane >> 4), ccol = wj * 64 + (lane & 15);
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                double* p = Cp + (int64_t)(crow + m * 16 + 4 * r) * g.ldc + ccol + n * 16;
                double v = g.alpha * acc[m][n][r];
                if (g.beta != 0.0) v = fma(g.beta, *p, v);
                *p = v;
            }
}

int launch_gemm(const GemmArgs& a, hipStream_t s) {
    int grid;
    switch (a.mode) {
        case GM_RECT: case GM_KHI_J: grid = a.mt * a.nt; break;
        case GM_LOWER: case GM_LAUUM: grid = a.mt * (a.mt + 1) / 2; break;
        default: grid = a.ntasks; break;
    }
    if (grid <= 0) return 0;
    const int v = (a.a_kmajor ? 2 : 0) | (a.b_kmajor ? 1 : 0);
    static bool attr_set = false;
    if (!attr_set) {
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_gemm<0, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS_BYTES));
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_gemm<0, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS_BYTES));
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_gemm<1, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS_BYTES));
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_gemm<1, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS_BYTES));
        attr_set = true;
    }
    switch (v) {
        case 0: hipLaunchKernelGGL((k_gemm<0, 0>), dim3(grid), dim3(256), GEMM_LDS_BYTES, s, a); break;
        case 1: hipLaunchKernelGGL((k_gemm<0, 1>), dim3(grid), dim3(256), GEMM_LDS_BYTES, s, a); break;
        case 2: hipLaunchKernelGGL((k_gemm<1, 0>), dim3(grid), dim3(256), GEMM_LDS_BYTES, s, a); break;
        default: hipLaunchKernelGGL((k_gemm<1, 1>), dim3(grid), dim3(256), GEMM_LDS_BYTES, s, a); break;
    }
    HIP_TRY(hipGetLastError());
    return 0;
}

double gemm_flops(const GemmArgs& a, const std::vector<GemmTask>* host_tasks) {
    const double tile = 2.0 * MOGP_TILE * MOGP_TILE;
    double k = 0.0;
    switch (a.mode) {
        case GM_RECT: k = (double)a.mt * a.nt * a.K; break;
        case GM_LOWER: k = (double)a.mt * (a.mt + 1) / 2 * a.K; break;
        case GM_LAUUM: for (int ti = 0; ti < a.mt; ++ti) k += (double)(ti + 1) * (a.K - ti * MOGP_TILE); break;
        case GM_KHI_J: for (int tj = 0; tj < a.nt; ++tj) k += (double)a.mt * ((tj + 1) * MOGP_TILE < a.K ? (tj + 1) * MOGP_TILE : a.K); break;
        default: if (host_tasks) for (const auto& t : *host_tasks) k += (double)t.kt * GEMM_BK; break;
    }
    return tile * k;
}

// ---- leaf: Cholesky + inverse of one 128x128 diagonal tile -----------------------------------------------------
// 256 threads; thread (ti = tid >> 4, tj = tid & 15) owns elements (ti + 16 m, tj + 16 n), m, n = 0..7, in registers.
// Right-looking column sweep: the owners of column k publish it to LDS, one barrier, everybody applies the rank-1
// update to its registers.  The outer 16-blocks are unrolled at compile time so every register index is static.
template <int KB>
__device__ __forceinline__ void potrf_block(double (&a)[8][8], double (*colbuf)[MOGP_TILE], int& cur, int ti, int tj,
                                            double& logsum, int& fail) {
#pragma nounroll
    for (int kk = 0; kk < 16; ++kk) {
        const int k = KB * 16 + kk;
        if (tj == kk) {
#pragma unroll
            for (int m = KB; m < 8; ++m) colbuf[cur][ti + 16 * m] = a[m][KB];
        }
        __syncthreads();
        const double d = colbuf[cur][k];
        if (!(d > 0.0) && fail < 0) fail = k;
        const double inv = 1.0 / sqrt(d);
        logsum += log(d);
        double ci[8], cj[8];
#pragma unroll
        for (int m = KB; m < 8; ++m) { ci[m] = colbuf[cur][ti + 16 * m] * inv; cj[m] = colbuf[cur][tj + 16 * m] * inv; }
        // column k itself
        if (tj == kk) {
#pragma unroll
            for (int m = KB; m < 8; ++m) {
                const int i = ti + 16 * m;
                a[m][KB] = i > k ? ci[m] : (i == k ? d * inv : 0.0);
            }
        }
        // trailing update of columns j > k, lower 16-blocks only (m >= n)
#pragma unroll
        for (int n = KB; n < 8; ++n) {
            const bool live = (n > KB) || (tj > kk);
            if (live) {
#pragma unroll
                for (int m = n; m < 8; ++m) a[m][n] = fma(-ci[m], cj[n], a[m][n]);
            }
        }
        cur ^= 1;
    }
}

// W = L^-1 by right-looking forward substitution on all 128 columns at once: row k of W is final after division by
// L[k][k]; its owners publish it; every row i > k subtracts L[i][k] * W[k,:].   L is read from LDS (ld 129).
template <int KB>
__device__ __forceinline__ void trtri_block(double (&w)[8][8], const double* Ls, double (*rowbuf)[MOGP_TILE], int& cur, int ti, int tj) {
#pragma nounroll
    for (int kk = 0; kk < 16; ++kk) {
        const int k = KB * 16 + kk;
        if (ti == kk) {
            const double dinv = 1.0 / Ls[k * 129 + k];
#pragma unroll
            for (int n = 0; n <= KB; ++n) {
                w[KB][n] *= dinv;
                rowbuf[cur][tj + 16 * n] = w[KB][n];
            }
        }
        __syncthreads();
#pragma unroll
        for (int m = KB; m < 8; ++m) {
            const bool live = (m > KB) || (ti > kk);
            if (live) {
                const double l = Ls[(ti + 16 * m) * 129 + k];
#pragma unroll
                for (int n = 0; n <= KB; ++n) {
                    if (n < KB || tj <= kk) w[m][n] = fma(-l, rowbuf[cur][tj + 16 * n], w[m][n]);
                }
            }
        }
        cur ^= 1;
    }
}

__global__ __launch_bounds__(256) void k_potrf_trtri_tile(double* A, int64_t ld, int t, double* invd, double* logdet,
                                                          unsigned long long* info) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double (*buf)[MOGP_TILE] = reinterpret_cast<double (*)[MOGP_TILE]>(smem);          // [2][128]
    double* Ls = smem + 2 * MOGP_TILE;                                                  // [128][129]
    const int tid = threadIdx.x, ti = tid >> 4, tj = tid & 15;
    double* At = A + (int64_t)t * MOGP_TILE * ld + (int64_t)t * MOGP_TILE;

    double a[8][8];
#pragma unroll
    for (int m = 0; m < 8; ++m)
#pragma unroll
        for (int n = 0; n < 8; ++n) a[m][n] = (m >= n) ? At[(int64_t)(ti + 16 * m) * ld + tj + 16 * n] : 0.0;

    int cur = 0, fail = -1;
    double logsum = 0.0;
    potrf_block<0>(a, buf, cur, ti, tj, logsum, fail);
    potrf_block<1>(a, buf, cur, ti, tj, logsum, fail);
    potrf_block<2>(a, buf, cur, ti, tj, logsum, fail);
    potrf_block<3>(a, buf, cur, ti, tj, logsum, fail);
    potrf_block<4>(a, buf, cur, ti, tj, logsum, fail);
    potrf_block<5>(a, buf, cur, ti, tj, logsum, fail);
    potrf_block<6>(a, buf, cur, ti, tj, logsum, fail);
    potrf_block<7>(a, buf, cur, ti, tj, logsum, fail);

    if (tid == 0) {
        logdet[t] = 0.5 * logsum;
        if (fail >= 0) atomicMin(info, (unsigned long long)((int64_t)t * MOGP_TILE + fail + 1));
    }
    // L back to global (strict upper part of the tile zeroed) and into LDS for the inversion
#pragma unroll
    for (int m = 0; m < 8; ++m)
#pragma unroll
        for (int n = 0; n < 8; ++n) {
            const int i = ti + 16 * m, j = tj + 16 * n;
            const double v = (i >= j) ? a[m][n] : 0.0;
            At[(int64_t)i * ld + j] = v;
            Ls[i * 129 + j] = v;
        }
    __syncthreads();

    double w[8][8];
#pragma unroll
    for (int m = 0; m < 8; ++m)
#pragma unroll
        for (int n = 0; n < 8; ++n) w[m][n] = (m == n && ti == tj) ? 1.0 : 0.0;
    cur = 0;
    trtri_block<0>(w, Ls, buf, cur, ti, tj);
    trtri_block<1>(w, Ls, buf, cur, ti, tj);
    trtri_block<2>(w, Ls, buf, cur, ti, tj);
    trtri_block<3>(w, Ls, buf, cur, ti, tj);
    trtri_block<4>(w, Ls, buf, cur, ti, tj);
    trtri_block<5>(w, Ls, buf, cur, ti, tj);
    trtri_block<6>(w, Ls, buf, cur, ti, tj);
    trtri_block<7>(w, Ls, buf, cur, ti, tj);

    double* Wt = invd + (int64_t)t * MOGP_TILE * MOGP_TILE;
#pragma unroll
    for (int m = 0; m < 8; ++m)
#pragma unroll
        for (int n = 0; n < 8; ++n) Wt[(ti + 16 * m) * MOGP_TILE + tj + 16 * n] = w[m][n];
}

static const int POTRF_LDS = (2 * MOGP_TILE + MOGP_TILE * 129) * (int)sizeof(double);

int launch_potrf_trtri_tile(double* A, int64_t ld, int t, double* invd, double* logdet, unsigned long long* info, hipStream_t s) {
    static bool attr_set = false;
    if (!attr_set) {
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_potrf_trtri_tile), hipFuncAttributeMaxDynamicSharedMemorySize, POTRF_LDS));
        attr_set = true;
    }
    hipLaunchKernelGGL(k_potrf_trtri_tile, dim3(1), dim3(256), POTRF_LDS, s, A, ld, t, invd, logdet, info);
    HIP_TRY(hipGetLastError());
    return 0;
}

// ---- small helpers ---------------------------------------------------------------------------------------------
__global__ void k_put_diag_tiles(double* A, int64_t ld, const double* invd) {
    const int t = blockIdx.x;
    double* At = A + (int64_t)t * MOGP_TILE * ld + (int64_t)t * MOGP_TILE;
    const double* W = invd + (int64_t)t * MOGP_TILE * MOGP_TILE;
    for (int idx = threadIdx.x; idx < MOGP_TILE * MOGP_TILE; idx += blockDim.x)
        At[(int64_t)(idx >> 7) * ld + (idx & 127)] = W[idx];
}
int launch_put_diag_tiles(double* A, int64_t ld, int nt, const double* invd, hipStream_t s) {
    hipLaunchKernelGGL(k_put_diag_tiles, dim3(nt), dim3(256), 0, s, A, ld, invd);
    HIP_TRY(hipGetLastError());
    return 0;
}

// rows N..Npad-1: zeros left of the diagonal, one on it (the Gram kernel never writes them)
__global__ void k_pad_identity(double* A, int64_t ld, int64_t N) {
    const int64_t r = N + blockIdx.x;
    for (int64_t c = threadIdx.x; c <= r; c += blockDim.x) A[r * ld + c] = (c == r) ? 1.0 : 0.0;
}
int launch_pad_identity(double* A, int64_t ld, int64_t N, int64_t Npad, hipStream_t s) {
    if (Npad > N) {
        hipLaunchKernelGGL(k_pad_identity, dim3((unsigned)(Npad - N)), dim3(256), 0, s, A, ld, N);
        HIP_TRY(hipGetLastError());
    }
    return 0;
}

// z_i = sum_{k<=i} W[i][k] y[k]; one wave per row, 4 rows per workgroup; zz_partial[block] = sum of z_i^2 of its rows
__global__ __launch_bounds__(256) void k_trmv_lower(const double* __restrict__ W, int64_t ld, int64_t n, const double* __restrict__ y,
                                                    double* __restrict__ z, double* __restrict__ zz_partial) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t i = (int64_t)blockIdx.x * 4 + wave;
    __shared__ double zz[4];
    double s = 0.0;
    if (i < n) {
        const double* row = W + i * ld;
        for (int64_t k = lane; k <= i; k += 64) s = fma(row[k], y[k], s);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
        if (lane == 0) z[i] = s;
    }
    if (lane == 0) zz[wave] = (i < n) ? s * s : 0.0;
    __syncthreads();
    if (threadIdx.x == 0) zz_partial[blockIdx.x] = (zz[0] + zz[1]) + (zz[2] + zz[3]);
}
int launch_trmv_lower(const double* W, int64_t ld, int64_t n, const double* y, double* z, double* zz_partial, hipStream_t s) {
    hipLaunchKernelGGL(k_trmv_lower, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, s, W, ld, n, y, z, zz_partial);
    HIP_TRY(hipGetLastError());
    return 0;
}

// a_j = sum_{i>=j} W[i][j] z_i.  grid (column blocks of 64, row chunks of 512); partial[chunk][j]; fixed-order second pass.
__global__ __launch_bounds__(256) void k_trmv_lower_t_part(const double* __restrict__ W, int64_t ld, int64_t n,
                                                           const double* __restrict__ z, double* __restrict__ part) {
    const int64_t j = (int64_t)blockIdx.x * 64 + (threadIdx.x & 63);
    const int sub = threadIdx.x >> 6;
    const int64_t r0 = (int64_t)blockIdx.y * 512, r1 = min(n, r0 + 512);
    __shared__ double red[4][64];
    double s = 0.0;
    if (j < n) {
        const int64_t start = max(r0, j);
        for (int64_t i = start + sub; i < r1; i += 4) s = fma(W[i * ld + j], z[i], s);
    }
    red[sub][threadIdx.x & 63] = s;
    __syncthreads();
    if (sub == 0 && j < n) part[(int64_t)blockIdx.y * n + j] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}
__global__ void k_sum_chunks(const double* __restrict__ part, int64_t n, int nchunks, double* __restrict__ out) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    double s = 0.0;
    for (int c = (int)(j / 512); c < nchunks; ++c) s += part[(int64_t)c * n + j];
    out[j] = s;
}
int launch_trmv_lower_t(const double* W, int64_t ld, int64_t n, const double* z, double* a, hipStream_t s) {
    // partial buffer lives right behind z's vector block: the caller passes `a` with room for (1 + nchunks) * n doubles
    const int nchunks = (int)((n + 511) / 512);
    double* part = a + n;
    hipLaunchKernelGGL(k_trmv_lower_t_part, dim3((unsigned)((n + 63) / 64), (unsigned)nchunks), dim3(256), 0, s, W, ld, n, z, part);
    hipLaunchKernelGGL(k_sum_chunks, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, part, n, nchunks, a);
    HIP_TRY(hipGetLastError());
    return 0;
}

__global__ __launch_bounds__(256) void k_gemv_rows(const double* __restrict__ M, int64_t ld, int64_t rows, int64_t n,
                                                   const double* __restrict__ v, double* __restrict__ out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t i = (int64_t)blockIdx.x * 4 + wave;
    if (i >= rows) return;
    const double* row = M + i * ld;
    double s = 0.0;
    for (int64_t k = lane; k < n; k += 64) s = fma(row[k], v[k], s);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if (lane == 0) out[i] = s;
}
int launch_gemv_rows(const double* M, int64_t ld, int64_t rows, int64_t n, const double* v, double* out, hipStream_t s) {
    hipLaunchKernelGGL(k_gemv_rows, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, M, ld, rows, n, v, out);
    HIP_TRY(hipGetLastError());
    return 0;
}

__global__ __launch_bounds__(256) void k_row_sqnorm_sub(const double* __restrict__ M, int64_t ld, int64_t rows, int64_t n,
                                                        const double* __restrict__ base, double* __restrict__ out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t i = (int64_t)blockIdx.x * 4 + wave;
    if (i >= rows) return;
    const double* row = M + i * ld;
    double s = 0.0;
    for (int64_t k = lane; k < n; k += 64) s = fma(row[k], row[k], s);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if (lane == 0) out[i] = base[i] - s;
}
int launch_row_sqnorm_sub(const double* M, int64_t ld, int64_t rows, int64_t n, const double* base, double* out, hipStream_t s) {
    hipLaunchKernelGGL(k_row_sqnorm_sub, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, M, ld, rows, n, base, out);
    HIP_TRY(hipGetLastError());
    return 0;
}

__global__ void k_nonfinite_scan(const double* __restrict__ A, int64_t ld, int64_t n, int* flag) {
    const int64_t r = blockIdx.x;
    int f = 0;
    for (int64_t c = threadIdx.x; c <= r; c += blockDim.x) {
        const double v = A[r * ld + c];
        if (isnan(v)) f |= 1;
        if (isinf(v)) f |= 2;
    }
    if (f) atomicOr(flag, f);
}
int launch_nonfinite_scan(const double* A, int64_t ld, int64_t n, int* flag, hipStream_t s) {
    hipLaunchKernelGGL(k_nonfinite_scan, dim3((unsigned)n), dim3(256), 0, s, A, ld, n, flag);
    HIP_TRY(hipGetLastError());
    return 0;
}

}  // namespace mogp
