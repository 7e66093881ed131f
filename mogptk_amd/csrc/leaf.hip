// leaf.hip -- Cholesky factor + inverse of one 128x128 diagonal tile, the serial critical path of the blocked
// factorisation (one launch per tile row).  One 256-thread workgroup; the LOWER triangle of the tile lives in LDS as 36
// packed 16x16 blocks (row stride 18 doubles inside a block: == 2 (mod 32) 8-byte units, so 16x4 MFMA fragment reads are
// conflict-free).  83 KB of LDS, so the workgroup fits on a CU next to a resident 74 KB trailing-update workgroup of
// the bulk stream (look-ahead schedule in mogp_api.hip) instead of waiting for an empty CU.
//   POTRF  for sb = 0..7:
//     P1  waves 0-2, one ROW per lane (the 16 diagonal-block rows + 48 panel rows per wave): right-looking Cholesky of
//         the 16x16 diagonal block where L[j][k] is broadcast with v_readlane (compile-time lane) -- the same instruction
//         stream IS the triangular solve for the panel rows riding along in lanes 16..63
//     P3  trailing update C_ij -= P_i P_j^T on v_mfma_f64_16x16x4_f64: the block column the next P1 reads right away (all waves),
//         the rest one step later on the waves that sit out P1 (look-ahead, hidden behind the serial micro-panel)
//   TRTRI  16x16 diagonal inverses (one column per lane), then the off-diagonal blocks by block COLUMN, two columns per wave, no
//          barriers: T = sum_k L_ik W_kj (MFMA), W_ij = -W_ii T (MFMA); finished blocks stay in registers -- accumulator register r
//          of a lane is element (4r + lane/16, lane%16), which is exactly the B-operand element of k-group r
// Outputs: invd (the tile inverse, zeros above the diagonal), logdet[t], the pivot check -- not L_kk (see below).
// Replaces the per-tile share of torch.linalg.cholesky (reference gpr/model.py:246).
#include "leaf_dev.h"

namespace mogp {

__global__ __launch_bounds__(256) void k_leaf128(double* A, int64_t ld, int t, double* invd, double* logdet,
                                                 unsigned long long* info, long long info_base, int store_L) {
    extern __shared__ __attribute__((aligned(16))) double lf[];
    leaf_tile<false>(lf, A, ld, t, invd, logdet, info, info_base, store_L);
}

int launch_potrf_trtri_tile(double* A, int64_t ld, int t, double* invd, double* logdet, unsigned long long* info, hipStream_t s,
                            long long info_base, int store_L) {
    static std::atomic<unsigned long long> attr_done{0ull};                 // one bit per device
    { int r__ = set_max_dynamic_lds(reinterpret_cast<const void*>(k_leaf128), LF_LDS_BYTES, attr_done); if (r__) return r__; }
    hipLaunchKernelGGL(k_leaf128, dim3(1), dim3(256), LF_LDS_BYTES, s, A, ld, t, invd, logdet, info, info_base, store_L);
    HIP_TRY(hipGetLastError());
    return 0;
}

}  // namespace mogp
