// wkk.hip -- W_KK = L_KK^-1 of one outer block (nk <= 4 tiles of 128) in ONE launch, for the critical path of the fused
// factorisation + inversion (potri.hip).  The leaf has already inverted the diagonal tiles (invd); what is left is
//     W(t, s) = -invd_t * sum_{i = s}^{t-1} L(t, i) W(i, s)        for tile rows t > s.
// Column blocks of W are independent of each other, and so are the columns inside one: a workgroup owns a 16-column strip of
// column block s, keeps the strips W(i, s) it has finished in LDS (they are the B operands of everything that follows) and walks
// down the tile rows.  32 workgroups of 256 threads, v_mfma_f64_16x16x4_f64 throughout, A fragments straight from L2-resident global
// memory.  Out of place (the other column blocks still read the L tiles): the result goes to a 512 x 512 store of its own.
// Replaces put_diag_tiles + six dependent 1..6-workgroup GEMM launches (~150 us on the critical path of every outer block).
#include "mogp_internal.h"

namespace mogp {

typedef double d4_t __attribute__((ext_vector_type(4)));

#define WK_COLS 16
#define WK_SLOT (MOGP_TILE * WK_COLS)
#define WK_LDS_BYTES (5 * WK_SLOT * 8)            // four finished strips + the intermediate T: 80 KB

// k-order: a lane (row = lane & 15, g = lane >> 4) fetches FOUR consecutive k of its A row with one 32-byte load (16 rows x 128-byte
// lines per wave instruction instead of 16 x 32-byte fragments), so MFMA j of a 16-wide k block contracts k = 16 kk + 4 g + j in
// slot g.  The B operands in LDS are stored with their rows permuted the same way (physical row 16 kk + 4 j + g holds logical row
// 16 kk + 4 g + j), which keeps every ds_read_b64 of a wave on 64 consecutive doubles.
__device__ __forceinline__ int wk_perm(int row) { return (row & ~15) | ((row & 3) << 2) | ((row >> 2) & 3); }

__device__ __forceinline__ void wk_product(const double* __restrict__ A0, const double* __restrict__ A1, int64_t lda, const double* Bs,
                                           int lr, int lk, double sign, d4_t& acc0, d4_t& acc1) {
    const d4_t* a0p = reinterpret_cast<const d4_t*>(A0 + (int64_t)lr * lda + 4 * lk);
    const d4_t* a1p = reinterpret_cast<const d4_t*>(A1 + (int64_t)lr * lda + 4 * lk);
    d4_t a0[8], a1[8];
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) { a0[kk] = a0p[4 * kk]; a1[kk] = a1p[4 * kk]; }        // all 16 loads in flight at once
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
        const double* bp = Bs + (16 * kk + lk) * WK_COLS + lr;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const double b = bp[4 * j * WK_COLS];
            acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(sign * a0[kk][j], b, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(sign * a1[kk][j], b, acc1, 0, 0, 0);
        }
    }
}

__global__ __launch_bounds__(256) void k_wkk(const double* __restrict__ Ablk, int64_t ld, const double* __restrict__ invd, int nk,
                                             double* __restrict__ Wk, int64_t ldw) {
    extern __shared__ __attribute__((aligned(16))) double wk_lds[];
    double* slot = wk_lds;                        // [4][128 (permuted rows)][16]
    double* Tb = wk_lds + 4 * WK_SLOT;            // [128 (permuted rows)][16]
    const int s = blockIdx.y, c0 = blockIdx.x * WK_COLS;
    if (s >= nk) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 15, lk = lane >> 4;
    __builtin_amdgcn_s_setprio(2);

    // W(s, s) = the leaf's tile inverse
    const double* Ds = invd + (int64_t)s * MOGP_TILE * MOGP_TILE;
    for (int e = tid; e < WK_SLOT; e += 256) {
        const int r = e >> 4, c = e & 15;
        const double v = Ds[r * MOGP_TILE + c0 + c];
        slot[s * WK_SLOT + wk_perm(r) * WK_COLS + c] = v;
        Wk[(int64_t)(s * MOGP_TILE + r) * ldw + s * MOGP_TILE + c0 + c] = v;
    }
    __syncthreads();

    const int ra = 16 * wave, rb = 16 * (wave + 4);           // this wave's two 16-row tiles
    for (int t = s + 1; t < nk; ++t) {
        // T = sum_i L(t, i) W(i, s)
        d4_t acc0 = (d4_t){0.0, 0.0, 0.0, 0.0}, acc1 = acc0;
        for (int i = s; i < t; ++i) {
            const double* L = Ablk + (int64_t)t * MOGP_TILE * ld + (int64_t)i * MOGP_TILE;
            wk_product(L + (int64_t)ra * ld, L + (int64_t)rb * ld, ld, slot + i * WK_SLOT, lr, lk, 1.0, acc0, acc1);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {                          // accumulator r of a lane is row lk + 4 r of the tile
            Tb[wk_perm(ra + lk + 4 * r) * WK_COLS + lr] = acc0[r];
            Tb[wk_perm(rb + lk + 4 * r) * WK_COLS + lr] = acc1[r];
        }
        __syncthreads();
        // W(t, s) = -invd_t T
        const double* Dt = invd + (int64_t)t * MOGP_TILE * MOGP_TILE;
        d4_t o0 = (d4_t){0.0, 0.0, 0.0, 0.0}, o1 = o0;
        wk_product(Dt + ra * MOGP_TILE, Dt + rb * MOGP_TILE, MOGP_TILE, Tb, lr, lk, -1.0, o0, o1);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int r0 = ra + lk + 4 * r, r1 = rb + lk + 4 * r;
            slot[t * WK_SLOT + wk_perm(r0) * WK_COLS + lr] = o0[r];
            slot[t * WK_SLOT + wk_perm(r1) * WK_COLS + lr] = o1[r];
            Wk[(int64_t)(t * MOGP_TILE + r0) * ldw + s * MOGP_TILE + c0 + lr] = o0[r];
            Wk[(int64_t)(t * MOGP_TILE + r1) * ldw + s * MOGP_TILE + c0 + lr] = o1[r];
        }
        __syncthreads();
    }
}

// Ablk: origin of the diagonal block in the matrix (leading dimension ld), holding L_KK's tiles below the diagonal; invd: the nk tile
// inverses; Wk: nk*128 square, leading dimension ldw, tiles above the diagonal untouched (zero from allocation)
int launch_wkk(const double* Ablk, int64_t ld, const double* invd, int nk, double* Wk, int64_t ldw, hipStream_t s) {
    static std::atomic<unsigned long long> attr_done{0ull};                 // one bit per device
    { int r__ = set_max_dynamic_lds(reinterpret_cast<const void*>(k_wkk), WK_LDS_BYTES, attr_done); if (r__) return r__; }
    hipLaunchKernelGGL(k_wkk, dim3(MOGP_TILE / WK_COLS, nk), dim3(256), WK_LDS_BYTES, s, Ablk, ld, invd, nk, Wk, ldw);
    HIP_TRY(hipGetLastError());
    return 0;
}

}  // namespace mogp
