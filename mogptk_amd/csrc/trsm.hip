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
// X_i = L_ii^-1 B_i by a leaf kernel: substitution in 16-row sub-blocks through the 128 x 128 tile of L.  Round 3's leaf (k_trsm_leaf: one COLUMN of the
// right-hand side per thread, the tile in LDS, every read a wave-wide broadcast) was bound by that broadcast -- 8 bytes per lane and FMA through LDS.
// Round 4 (k_trsm_leaf_r): the products with the solved sub-blocks on v_mfma_f64_16x16x4_f64, the solved rows kept in registers, the vector ALU only for
// the 16 x 16 diagonal blocks: 114 -> 80 us per block row at configs[4], the same bits.  2 * 8256 flops per column and block row: 1.3e10 flops for the
// 2048 x 100 000 solve of configs[4] next to 4.2e11 in the GEMMs.
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

// (round 4) The same substitution with the tile of L read through the SCALAR path.  k_trsm_leaf delivers every L value to 64 lanes out of LDS:
// 512 bytes per wave and FMA through a 128 B/clk return path that four SIMDs share -- 8256 FMAs x 8 waves x 4 clk = 110 us per block row, which is
// what the kernel took (114 us; its arithmetic is 30).  L is the same for every column, i.e. wave-uniform: read with s_load (uniform addresses,
// const __restrict__, nothing in the kernel writes it: the compiler selects scalar loads) it arrives in SGPRs and enters the FMA as its scalar
// operand at no per-lane cost.  Scalar loads return out of order, so the only wait there is is lgkmcnt(0), which the compiler places in front of the
// first use: the loops below use ONE value of the row in hand, THEN request the next row, then do the rest of the FMAs -- the scheduling barriers
// keep that order (left alone the compiler requests the next row first and waits for both).  No LDS tile.  Same operations in the same order per
// unknown as k_trsm_leaf: bit-identical results.
#define TS_PIN() __builtin_amdgcn_sched_barrier(0)

template <bool TRANS, int THREADS, int NC>
__global__ __launch_bounds__(THREADS) void k_trsm_leaf_s(const double* __restrict__ Lt_, int64_t ldl, double* __restrict__ B, int64_t ldb, int64_t ncols) {
    // L does not change while the kernel runs: read through the constant address space, so that every uniform read of it is a scalar load (behind the
    // buffer stores of the solved rows the compiler no longer proves that for a plain global pointer and falls back to per-lane loads)
    typedef const __attribute__((address_space(4))) double* cptr_t;
    const cptr_t Lt = (cptr_t)Lt_;
    __shared__ double inv[TS_T];
    const unsigned tid = threadIdx.x;
    for (int k = tid; k < TS_T; k += THREADS) inv[k] = 1.0 / Lt_[(int64_t)k * ldl + k];
    __syncthreads();
    // NC columns per thread (col0 + tid + k THREADS): every scalar of L then feeds NC independent FMAs.
    // Rows of the right-hand side through a buffer resource: scalar base (the workgroup's first column) + the row's byte offset in an SGPR + the
    // lane's 8 tid (+ 8 k THREADS as the instruction's immediate) -- one VGPR of address for the whole kernel instead of a 64-bit pointer per row in
    // flight (197 -> 110 VGPRs at NC = 1).
    const int64_t col0 = (int64_t)blockIdx.x * (THREADS * NC);      // the launcher picks a form whose THREADS NC divides ncols: no lane is out of range
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(B + col0, 0, (int)0xffffffffu, 0x00020000);
    const unsigned voff = tid * 8u;
    const unsigned ldb8 = (unsigned)(ldb * 8);                 // the caller guarantees 128 ldb 8 < 2^32
    typedef decltype(__builtin_amdgcn_raw_buffer_load_b64(rs, 0, 0, 0)) raw64_t;
    typedef double xs_t[NC][TS_SB];
    auto ld = [&](int r, int k) -> double { return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rs, voff + k * THREADS * 8, (unsigned)r * ldb8, 0)); };
    auto st = [&](int r, int k, double v) { __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(raw64_t, v), rs, voff + k * THREADS * 8, (unsigned)r * ldb8, 0); };
    auto fetch = [&](xs_t& xp, int pb) {
#pragma unroll
        for (int c = 0; c < TS_SB; ++c)
#pragma unroll
            for (int k = 0; k < NC; ++k) xp[k][c] = ld(pb * TS_SB + c, k);
    };
    auto lrow = [&](double (&d)[8], cptr_t p) {
#pragma unroll
        for (int j = 0; j < 8; ++j) d[j] = p[j];
    };
    // x[.] -= (16 x 16 block of L at Lb, row stride ldl) applied to xp: rows of the block are the FMA's scalar operands.
    // forward form: x[r] -= sum_c Lb[r][c] xp[c];  transposed form: x[r] -= sum_c Lb[c][r] xp[c]  (row c of the block holds the 16 r's)
    auto apply = [&](xs_t& x, const xs_t& xp, cptr_t Lb) {
        double a0[8], a1[8], b0[8], b1[8];
        if (!TRANS) {
            // rows i and i + 1 side by side, half a row at a time: two independent FMA chains per column (a row alone is ONE chain of 16 dependent
            // FMAs, and there are only ~1.5 waves per SIMD to fill the gaps: 100 000 columns are 1563 waves)
            lrow(a0, Lb); lrow(b0, Lb + ldl); TS_PIN();
#pragma unroll
            for (int i = 0; i < TS_SB; i += 2) {
                const cptr_t r0 = Lb + (int64_t)i * ldl;
                const cptr_t r1 = r0 + ldl;
#pragma unroll
                for (int k = 0; k < NC; ++k) x[k][i] = fma(-a0[0], xp[k][0], x[k][i]);
                TS_PIN(); lrow(a1, r0 + 8); lrow(b1, r1 + 8); TS_PIN();
#pragma unroll
                for (int j = 0; j < 8; ++j)
#pragma unroll
                    for (int k = 0; k < NC; ++k) {
                        if (j > 0) x[k][i] = fma(-a0[j], xp[k][j], x[k][i]);
                        x[k][i + 1] = fma(-b0[j], xp[k][j], x[k][i + 1]);
                    }
                TS_PIN();
#pragma unroll
                for (int k = 0; k < NC; ++k) x[k][i] = fma(-a1[0], xp[k][8], x[k][i]);
                TS_PIN();
                if (i + 2 < TS_SB) { lrow(a0, r1 + ldl); lrow(b0, r1 + 2 * ldl); TS_PIN(); }
#pragma unroll
                for (int j = 0; j < 8; ++j)
#pragma unroll
                    for (int k = 0; k < NC; ++k) {
                        if (j > 0) x[k][i] = fma(-a1[j], xp[k][8 + j], x[k][i]);
                        x[k][i + 1] = fma(-b1[j], xp[k][8 + j], x[k][i + 1]);
                    }
                TS_PIN();
            }
        } else {
            lrow(a0, Lb); lrow(a1, Lb + 8); TS_PIN();
#pragma unroll
            for (int i = 0; i < TS_SB; i += 2) {
                const cptr_t n1 = Lb + (int64_t)(i + 1) * ldl;
#pragma unroll
                for (int k = 0; k < NC; ++k) x[k][0] = fma(-a0[0], xp[k][i], x[k][0]);
                TS_PIN(); lrow(b0, n1); lrow(b1, n1 + 8); TS_PIN();
#pragma unroll
                for (int j = 1; j < 8; ++j)
#pragma unroll
                    for (int k = 0; k < NC; ++k) x[k][j] = fma(-a0[j], xp[k][i], x[k][j]);
#pragma unroll
                for (int j = 0; j < 8; ++j)
#pragma unroll
                    for (int k = 0; k < NC; ++k) x[k][8 + j] = fma(-a1[j], xp[k][i], x[k][8 + j]);
                TS_PIN();
#pragma unroll
                for (int k = 0; k < NC; ++k) x[k][0] = fma(-b0[0], xp[k][i + 1], x[k][0]);
                TS_PIN();
                if (i + 2 < TS_SB) { const cptr_t n2 = Lb + (int64_t)(i + 2) * ldl; lrow(a0, n2); lrow(a1, n2 + 8); TS_PIN(); }
#pragma unroll
                for (int j = 1; j < 8; ++j)
#pragma unroll
                    for (int k = 0; k < NC; ++k) x[k][j] = fma(-b0[j], xp[k][i + 1], x[k][j]);
#pragma unroll
                for (int j = 0; j < 8; ++j)
#pragma unroll
                    for (int k = 0; k < NC; ++k) x[k][8 + j] = fma(-b1[j], xp[k][i + 1], x[k][8 + j]);
                TS_PIN();
            }
        }
    };
    if (!TRANS) {
        for (int sb = 0; sb < TS_T / TS_SB; ++sb) {
            xs_t x, xa, xb;
#pragma unroll
            for (int r = 0; r < TS_SB; ++r)
#pragma unroll
                for (int k = 0; k < NC; ++k) x[k][r] = ld(sb * TS_SB + r, k);
            if (sb > 0) fetch(xa, 0);
            for (int pb = 0; pb < sb; pb += 2) {
                if (pb + 1 < sb) fetch(xb, pb + 1);
                apply(x, xa, Lt + (int64_t)(sb * TS_SB) * ldl + pb * TS_SB);
                if (pb + 1 < sb) {
                    if (pb + 2 < sb) fetch(xa, pb + 2);
                    apply(x, xb, Lt + (int64_t)(sb * TS_SB) * ldl + (pb + 1) * TS_SB);
                }
            }
            // diagonal sub-block, row by row (per x[r] the same FMAs in the same order as the column-oriented loop of k_trsm_leaf); row r + 1 is
            // requested behind the first FMA of row r
            const cptr_t Ld = Lt + (int64_t)(sb * TS_SB) * (ldl + 1);
            double iv[TS_SB];
#pragma unroll
            for (int c = 0; c < TS_SB; ++c) iv[c] = inv[sb * TS_SB + c];
            double a0[8], a1[8], b0[8], b1[8];
            lrow(a0, Ld + ldl); lrow(a1, Ld + ldl + 8); TS_PIN();                       // row 0 has nothing left of the diagonal
#pragma unroll
            for (int k = 0; k < NC; ++k) x[k][0] *= iv[0];
#pragma unroll
            for (int r = 1; r < TS_SB; r += 2) {                                        // rows r (in a) and r + 1 (in b)
#pragma unroll
                for (int k = 0; k < NC; ++k) x[k][r] = fma(-a0[0], x[k][0], x[k][r]);
                TS_PIN();
                if (r + 1 < TS_SB) { const cptr_t n1 = Ld + (int64_t)(r + 1) * ldl; lrow(b0, n1); lrow(b1, n1 + 8); TS_PIN(); }
#pragma unroll
                for (int c = 1; c < r; ++c)
#pragma unroll
                    for (int k = 0; k < NC; ++k) x[k][r] = fma(-(c < 8 ? a0[c & 7] : a1[c & 7]), x[k][c], x[k][r]);
#pragma unroll
                for (int k = 0; k < NC; ++k) x[k][r] *= iv[r];
                TS_PIN();
                if (r + 1 < TS_SB) {
#pragma unroll
                    for (int k = 0; k < NC; ++k) x[k][r + 1] = fma(-b0[0], x[k][0], x[k][r + 1]);
                    TS_PIN();
                    if (r + 2 < TS_SB) { const cptr_t n2 = Ld + (int64_t)(r + 2) * ldl; lrow(a0, n2); lrow(a1, n2 + 8); TS_PIN(); }
#pragma unroll
                    for (int c = 1; c < r + 1; ++c)
#pragma unroll
                        for (int k = 0; k < NC; ++k) x[k][r + 1] = fma(-(c < 8 ? b0[c & 7] : b1[c & 7]), x[k][c], x[k][r + 1]);
#pragma unroll
                    for (int k = 0; k < NC; ++k) x[k][r + 1] *= iv[r + 1];
                    TS_PIN();
                }
            }
#pragma unroll
            for (int r = 0; r < TS_SB; ++r)
#pragma unroll
                for (int k = 0; k < NC; ++k) st(sb * TS_SB + r, k, x[k][r]);
        }
    } else {
        constexpr int LAST = TS_T / TS_SB - 1;
        for (int sb = LAST; sb >= 0; --sb) {
            xs_t x, xa, xb;
#pragma unroll
            for (int r = 0; r < TS_SB; ++r)
#pragma unroll
                for (int k = 0; k < NC; ++k) x[k][r] = ld(sb * TS_SB + r, k);
            if (LAST > sb) fetch(xa, LAST);
            for (int pb = LAST; pb > sb; pb -= 2) {
                if (pb - 1 > sb) fetch(xb, pb - 1);
                apply(x, xa, Lt + (int64_t)(pb * TS_SB) * ldl + sb * TS_SB);
                if (pb - 1 > sb) {
                    if (pb - 2 > sb) fetch(xa, pb - 2);
                    apply(x, xb, Lt + (int64_t)((pb - 1) * TS_SB) * ldl + sb * TS_SB);
                }
            }
            // diagonal sub-block: x[c] final, then row c of L (its entries left of the diagonal) takes it out of the unknowns above
            const cptr_t Ld = Lt + (int64_t)(sb * TS_SB) * (ldl + 1);
            double iv[TS_SB];
#pragma unroll
            for (int c = 0; c < TS_SB; ++c) iv[c] = inv[sb * TS_SB + c];
            double a0[8], a1[8], b0[8], b1[8];
            { const cptr_t n0 = Ld + (int64_t)(TS_SB - 1) * ldl; lrow(a0, n0); lrow(a1, n0 + 8); TS_PIN(); }
#pragma unroll
            for (int c = TS_SB - 1; c >= 1; c -= 2) {                                   // rows c (in a) and c - 1 (in b)
#pragma unroll
                for (int k = 0; k < NC; ++k) x[k][c] *= iv[c];
#pragma unroll
                for (int k = 0; k < NC; ++k) x[k][0] = fma(-a0[0], x[k][c], x[k][0]);
                TS_PIN();
                { const cptr_t n1 = Ld + (int64_t)(c - 1) * ldl; lrow(b0, n1); lrow(b1, n1 + 8); TS_PIN(); }
#pragma unroll
                for (int r = 1; r < c; ++r)
#pragma unroll
                    for (int k = 0; k < NC; ++k) x[k][r] = fma(-(r < 8 ? a0[r & 7] : a1[r & 7]), x[k][c], x[k][r]);
                TS_PIN();
#pragma unroll
                for (int k = 0; k < NC; ++k) x[k][c - 1] *= iv[c - 1];
                if (c - 1 > 0) {
#pragma unroll
                    for (int k = 0; k < NC; ++k) x[k][0] = fma(-b0[0], x[k][c - 1], x[k][0]);
                    TS_PIN();
                    if (c - 2 > 0) { const cptr_t n2 = Ld + (int64_t)(c - 2) * ldl; lrow(a0, n2); lrow(a1, n2 + 8); TS_PIN(); }
#pragma unroll
                    for (int r = 1; r < c - 1; ++r)
#pragma unroll
                        for (int k = 0; k < NC; ++k) x[k][r] = fma(-(r < 8 ? b0[r & 7] : b1[r & 7]), x[k][c - 1], x[k][r]);
                    TS_PIN();
                }
            }
#pragma unroll
            for (int r = 0; r < TS_SB; ++r)
#pragma unroll
                for (int k = 0; k < NC; ++k) st(sb * TS_SB + r, k, x[k][r]);
        }
    }
}

// (round 4, the form in use for wide right-hand sides) The off-diagonal part of the substitution on the matrix cores.  Both kernels above are bound by
// how the L operand reaches the FMA -- 8 bytes per lane out of LDS (k_trsm_leaf: 110 us of LDS return path per block row) or one scalar-cache miss per
// 16 FMAs (k_trsm_leaf_s: ~700 clk each; measured 165 us).  v_mfma_f64_16x16x4_f64 takes a 16 x 4 piece of L spread over the lanes (one double a lane, read
// once from the L2-resident factor) against a 4 x 16 piece of the solved rows and keeps the 16 x 16 result in registers: the same FMA rate as the vector
// ALU, no operand traffic to speak of.  One wave = 64 columns = four 16-column groups; per 16-row sub-block sb it accumulates
// X_sb -= L[sb, pb] X_pb over the solved sub-blocks pb in the kernels' order (pb, then c ascending), passes the 16 x 64 block through a wave-private LDS
// slab into the column-per-lane layout and finishes with the diagonal 16 x 16 block by substitution on the vector ALU exactly as k_trsm_leaf does
// (the eight diagonal blocks and the reciprocal pivots sit in LDS for the whole workgroup: 17 KB).  Nothing waits for another wave after the first barrier.
#define TM_SLAB (TS_SB * 65)
template <bool TRANS>
__global__ __launch_bounds__(256, 2) void k_trsm_leaf_m(const double* __restrict__ Lt, int64_t ldl, double* __restrict__ B, int64_t ldb, int64_t ncols) {
    __shared__ double dg[TS_T / TS_SB][TS_SB][TS_SB];        // diagonal 16 x 16 blocks of the tile
    __shared__ double inv[TS_T];
    __shared__ double slab[4][TM_SLAB];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    for (int idx = tid; idx < TS_T * TS_SB; idx += 256) {
        const int r = idx >> 4, c = idx & 15, sb = r >> 4;
        dg[sb][r & 15][c] = Lt[(int64_t)r * ldl + sb * TS_SB + c];
    }
    if (tid < TS_T) inv[tid] = 1.0 / Lt[(int64_t)tid * ldl + tid];
    __syncthreads();
    const int64_t col0 = (int64_t)blockIdx.x * 256 + wv * 64;
    if (col0 >= ncols) return;                                 // ncols is a multiple of 128 and 64 divides it: whole waves only
    const int ln = lane & 15, lk = lane >> 4;
    // every global access as buffer base + per-lane offset (one VGPR each for the three patterns) + a uniform row offset in an SGPR + an immediate:
    // with 64-bit pointers per access the kernel needed 224 / 348 VGPRs
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(B + col0, 0, (int)0xffffffffu, 0x00020000);
    const __amdgpu_buffer_rsrc_t rl = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(Lt), 0, (int)0xffffffffu, 0x00020000);
    typedef decltype(__builtin_amdgcn_raw_buffer_load_b64(rb, 0, 0, 0)) raw64_t;
    const unsigned ldb8 = (unsigned)(ldb * 8), ldl8 = (unsigned)(ldl * 8);      // the caller guarantees 128 ldb 8 < 2^32
    const unsigned vB = (unsigned)lk * ldb8 + (unsigned)ln * 8u;               // row lk, column ln of the wave's 64 columns
    const unsigned vC = (unsigned)lane * 8u;                                     // column `lane`
    const unsigned vA = TRANS ? (unsigned)lk * ldl8 + (unsigned)ln * 8u : (unsigned)ln * ldl8 + (unsigned)lk * 8u;
    auto ldB = [&](unsigned voff, unsigned row, int imm) -> double { return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rb, voff + imm, row * ldb8, 0)); };
    double* sl = slab[wv];
    constexpr int NSB = TS_T / TS_SB;
    for (int step = 0; step < NSB; ++step) {
        const int sb = TRANS ? NSB - 1 - step : step;
        // the sub-block's right-hand side in the accumulator layout: reg v of group g = row lk + 4 v, column 16 g + ln
        typedef double d4_t __attribute__((ext_vector_type(4)));
        d4_t acc[4];
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int v = 0; v < 4; ++v) acc[g][v] = ldB(vB, sb * TS_SB + 4 * v, 128 * g);
        // operands of one solved sub-block pb: a[ks] = -L piece (row ln of the sub-block, k = 4 ks + lk), b[ks][g] = X_pb row 4 ks + lk, column 16 g + ln
        auto load_ab = [&](double (&a)[4], double (&b)[4][4], int pb) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const unsigned so = TRANS ? (unsigned)(pb * TS_SB + 4 * ks) * ldl8 + (unsigned)(sb * TS_SB * 8) : (unsigned)(sb * TS_SB) * ldl8 + (unsigned)((pb * TS_SB + 4 * ks) * 8);
                a[ks] = -__builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rl, vA, so, 0));
#pragma unroll
                for (int g = 0; g < 4; ++g) b[ks][g] = ldB(vB, pb * TS_SB + 4 * ks, 128 * g);
            }
        };
        auto mm = [&](const double (&a)[4], const double (&b)[4][4]) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int g = 0; g < 4; ++g) acc[g] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[ks], b[ks][g], acc[g], 0, 0, 0);
        };
        double a0[4], b0[4][4], a1[4], b1[4][4];
        if (step > 0) load_ab(a0, b0, TRANS ? NSB - 1 : 0);
        for (int q = 0; q < step; q += 2) {                     // solved sub-blocks in the order of k_trsm_leaf, operands one sub-block ahead
            const int pb = TRANS ? NSB - 1 - q : q;
            const int pn = TRANS ? pb - 1 : pb + 1;
            if (q + 1 < step) load_ab(a1, b1, pn);
            mm(a0, b0);
            if (q + 1 < step) {
                if (q + 2 < step) load_ab(a0, b0, TRANS ? pn - 1 : pn + 1);
                mm(a1, b1);
            }
        }
        // accumulator layout -> one column per lane, through the wave's slab
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int v = 0; v < 4; ++v) sl[(lk + 4 * v) * 65 + 16 * g + ln] = acc[g][v];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        double x[TS_SB];
#pragma unroll
        for (int r = 0; r < TS_SB; ++r) x[r] = sl[r * 65 + lane];
        const double (*Ld)[TS_SB] = dg[sb];
        if (!TRANS) {
#pragma unroll
            for (int c = 0; c < TS_SB; ++c) {
                x[c] *= inv[sb * TS_SB + c];
#pragma unroll
                for (int r = c + 1; r < TS_SB; ++r) x[r] = fma(-Ld[r][c], x[c], x[r]);
            }
        } else {
#pragma unroll
            for (int c = TS_SB - 1; c >= 0; --c) {
                x[c] *= inv[sb * TS_SB + c];
#pragma unroll
                for (int r = 0; r < c; ++r) x[r] = fma(-Ld[c][r], x[c], x[r]);
            }
        }
#pragma unroll
        for (int r = 0; r < TS_SB; ++r)
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(raw64_t, x[r]), rb, vC, (unsigned)(sb * TS_SB + r) * ldb8, 0);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");       // the stores are read back (other lanes, next sub-block) by this wave only
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
}

// (round 4, second step) k_trsm_leaf_m still takes 125 us: every sub-block is a chain of global round trips -- its rows in, the solved rows of all earlier
// sub-blocks in again as MFMA operands (448 loads per lane and block row), the result out, and the next sub-block cannot request its operands before
// those stores have landed.  Here a wave keeps what it has solved IN REGISTERS: the accumulator layout of v_mfma_f64_16x16x4_f64 (lane (n, j), register v:
// row j + 4 v, column n) is also its B-operand layout (lane (n, k): row 4 ks + k), so a solved 16 x 64 block, turned from one column per lane back into
// that layout through the wave's LDS slab, IS the operand of every later sub-block.  The first TR_KEEP solved sub-blocks stay in registers for the whole
// tile (16 doubles a lane each), the one solved last comes straight from the slab, and only what lies between is read back from memory (3 block reads a
// block row instead of 28), requested a whole step ahead.  L comes out of LDS: the 28 off-diagonal 16 x 16 blocks as A operands (one double a lane and k
// step, padded rows), the diagonal blocks and reciprocal pivots for the substitution.  512 threads = 512 columns per workgroup, two waves per SIMD (one
// wave's substitution under the other's MFMAs), 144 KB of LDS: 196 workgroups at configs[4], a single round of the chip.
// (First version: everything solved kept in registers, ~330 of them, one wave per SIMD, 256-column workgroups: 391 workgroups on 256 CUs = two rounds of
// 46 us; before that, with the tile copied to LDS by a loop with a `continue`, 122 us: the 64 loads of a thread went out one at a time.)
#define TR_BLK (TS_SB * 17)
#define TR_WAVES 8
#define TR_LDS_BYTES ((28 * TR_BLK + 8 * TS_SB * TS_SB + TS_T + TR_WAVES * TM_SLAB) * 8)
template <bool TRANS, int TR_KEEP>
__global__ __launch_bounds__(64 * TR_WAVES) void k_trsm_leaf_r(const double* __restrict__ Lt, int64_t ldl, double* __restrict__ B, int64_t ldb, int64_t ncols) {
    extern __shared__ __attribute__((aligned(16))) double tr_lds[];
    double* La = tr_lds;                                  // [28][16][17]: block (R, C), R > C, at R (R - 1) / 2 + C; forward: [m][k] = L[16 R + m][16 C + k]; transposed: [m][k] = L[16 R + k][16 C + m]
    double* dg = La + 28 * TR_BLK;                        // [8][16][16] diagonal blocks
    double* inv = dg + 8 * TS_SB * TS_SB;                 // [128]
    double* slabs = inv + TS_T;                           // [waves][16][65]
    constexpr int NT = 64 * TR_WAVES;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    {   // the tile into LDS: all of a thread's loads in flight at once (thread = column c of the tile, rows r0, r0 + NT / 128, ...)
        constexpr int RS = NT / TS_T, NI = TS_T / RS;
        const int c = tid & 127, C = c >> 4, r0 = tid >> 7;
        double v[NI];
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int r = r0 + RS * i;
            v[i] = (r >> 4) >= C ? Lt[(int64_t)r * ldl + c] : 0.0;
        }
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int r = r0 + RS * i, R = r >> 4;
            if (R == C) dg[(R * TS_SB + (r & 15)) * TS_SB + (c & 15)] = v[i];
            else if (R > C) La[(R * (R - 1) / 2 + C) * TR_BLK + (TRANS ? (c & 15) * 17 + (r & 15) : (r & 15) * 17 + (c & 15))] = v[i];
        }
    }
    if (tid < TS_T) inv[tid] = 1.0 / Lt[(int64_t)tid * ldl + tid];
    __syncthreads();
    const int64_t col0 = (int64_t)blockIdx.x * NT + wv * 64;
    if (col0 >= ncols) return;                                 // ncols is a multiple of 128: whole waves only
    const int ln = lane & 15, lk = lane >> 4;
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(B + col0, 0, (int)0xffffffffu, 0x00020000);
    typedef decltype(__builtin_amdgcn_raw_buffer_load_b64(rb, 0, 0, 0)) raw64_t;
    typedef double d4_t __attribute__((ext_vector_type(4)));
    const unsigned ldb8 = (unsigned)(ldb * 8);                                    // the caller guarantees 128 ldb 8 < 2^32
    const unsigned vB = (unsigned)lk * ldb8 + (unsigned)ln * 8u;               // row lk, column ln of the wave's 64 columns
    const unsigned vC = (unsigned)lane * 8u;                                     // column `lane`
    double* sl = slabs + wv * TM_SLAB;
    const double* Aln = La + ln * 17 + lk;                                        // this lane's element of a block's k step 0
    constexpr int NSB = TS_T / TS_SB;
    auto load_rows = [&](d4_t (&d)[4], int sb) {              // sub-block sb in the accumulator = operand layout
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int v = 0; v < 4; ++v)
                d[g][v] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rb, vB + 128 * g, (unsigned)(sb * TS_SB + 4 * v) * ldb8, 0));
    };
    auto wave_sync = [&]() {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    auto sbq = [](int q) { return TRANS ? NSB - 1 - q : q; };   // the q-th sub-block in solving order
    d4_t keep[TR_KEEP][4];                                      // the first TR_KEEP solved sub-blocks
    d4_t last[4];                                               // the one solved in the previous step
    d4_t acc[4], accn[4];
    load_rows(acc, sbq(0));
#pragma unroll
    for (int step = 0; step < NSB; ++step) {
        const int sb = sbq(step);
        d4_t back[NSB > TR_KEEP + 1 ? NSB - TR_KEEP - 1 : 1][4];          // solved sub-blocks TR_KEEP .. step - 2, read back (stored at least a step ago)
#pragma unroll
        for (int q = TR_KEEP; q < step - 1; ++q) load_rows(back[q - TR_KEEP], sbq(q));
        if (step + 1 < NSB) load_rows(accn, sbq(step + 1));
#pragma unroll
        for (int q = 0; q < step; ++q) {                       // the solved sub-blocks in k_trsm_leaf's order
            const int p = sbq(q);
            const int R = TRANS ? p : sb, C = TRANS ? sb : p;
            const double* Ab = Aln + (R * (R - 1) / 2 + C) * TR_BLK;
            const d4_t (&bq)[4] = q < TR_KEEP ? keep[q < TR_KEEP ? q : 0] : (q == step - 1 ? last : back[q >= TR_KEEP ? q - TR_KEEP : 0]);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const double a = -Ab[4 * ks];
#pragma unroll
                for (int g = 0; g < 4; ++g) acc[g] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bq[g][ks], acc[g], 0, 0, 0);
            }
        }
        // accumulator layout -> one column per lane
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int v = 0; v < 4; ++v) sl[(lk + 4 * v) * 65 + 16 * g + ln] = acc[g][v];
        wave_sync();
        double x[TS_SB];
#pragma unroll
        for (int r = 0; r < TS_SB; ++r) x[r] = sl[r * 65 + lane];
        const double* Ld = dg + sb * TS_SB * TS_SB;
        if (!TRANS) {
#pragma unroll
            for (int c = 0; c < TS_SB; ++c) {
                x[c] *= inv[sb * TS_SB + c];
#pragma unroll
                for (int r = c + 1; r < TS_SB; ++r) x[r] = fma(-Ld[r * TS_SB + c], x[c], x[r]);
            }
        } else {
#pragma unroll
            for (int c = TS_SB - 1; c >= 0; --c) {
                x[c] *= inv[sb * TS_SB + c];
#pragma unroll
                for (int r = 0; r < c; ++r) x[r] = fma(-Ld[c * TS_SB + r], x[c], x[r]);
            }
        }
#pragma unroll
        for (int r = 0; r < TS_SB; ++r)
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(raw64_t, x[r]), rb, vC, (unsigned)(sb * TS_SB + r) * ldb8, 0);
        if (step + 1 < NSB) {
            // the solved block back into the accumulator = operand layout, for the sub-blocks still to come
            wave_sync();
#pragma unroll
            for (int r = 0; r < TS_SB; ++r) sl[r * 65 + lane] = x[r];
            wave_sync();
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int v = 0; v < 4; ++v) last[g][v] = sl[(lk + 4 * v) * 65 + 16 * g + ln];
            if (step < TR_KEEP) {
#pragma unroll
                for (int g = 0; g < 4; ++g) keep[step < TR_KEEP ? step : 0][g] = last[g];
            }
            wave_sync();
#pragma unroll
            for (int g = 0; g < 4; ++g) acc[g] = accn[g];
        }
    }
}

template <bool TRANS, int KEEP>
static int launch_leaf_r(const double* Lt, int64_t ldl, double* B, int64_t ldb, int64_t ncols, hipStream_t s) {
    static std::atomic<unsigned long long> attr_done{0ull};                 // one bit per device
    { int r__ = set_max_dynamic_lds(reinterpret_cast<const void*>(k_trsm_leaf_r<TRANS, KEEP>), TR_LDS_BYTES, attr_done); if (r__) return r__; }
    constexpr int NT = 64 * TR_WAVES;
    hipLaunchKernelGGL((k_trsm_leaf_r<TRANS, KEEP>), dim3((unsigned)((ncols + NT - 1) / NT)), dim3(NT), TR_LDS_BYTES, s, Lt, ldl, B, ldb, ncols);
    HIP_TRY(hipGetLastError());
    return 0;
}

template <bool TRANS>
static int launch_leaf_m(const double* Lt, int64_t ldl, double* B, int64_t ldb, int64_t ncols, hipStream_t s) {
    hipLaunchKernelGGL((k_trsm_leaf_m<TRANS>), dim3((unsigned)((ncols + 255) / 256)), dim3(256), 0, s, Lt, ldl, B, ldb, ncols);
    HIP_TRY(hipGetLastError());
    return 0;
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
template <bool TRANS, int THREADS, int NC>
static int launch_leaf_s(const double* Lt, int64_t ldl, double* B, int64_t ldb, int64_t ncols, hipStream_t s) {
    hipLaunchKernelGGL((k_trsm_leaf_s<TRANS, THREADS, NC>), dim3((unsigned)((ncols + THREADS * NC - 1) / (THREADS * NC))), dim3(THREADS), 0, s, Lt, ldl, B, ldb, ncols);
    HIP_TRY(hipGetLastError());
    return 0;
}
template <bool TRANS>
static int launch_leaf(const double* Lt, int64_t ldl, double* B, int64_t ldb, int64_t ncols, hipStream_t s) {
    // MOGP_TRSM_LEAF: 0 the LDS-tile kernel of round 3; 1-4 L through the scalar path; 5 / 6 matrix cores, solved rows re-read from memory (wide / all
    // right-hand sides); 7 / 8 matrix cores, solved rows kept in registers (wide / all); 9 / 10 the same with three kept sub-blocks.  Every form gives
    // the same bits (tools/r4_leaf.sh: the checksum of a configs[4] gradient); 8 is the fastest (configs[4]: 48.3 -> 46.3 ms on one box).
    static const int form = []() { const char* e = std::getenv("MOGP_TRSM_LEAF"); return e ? atoi(e) : 8; }();
    if (form >= 5 && ldb * 8 * TS_T < (int64_t)1 << 32) {     // matrix-core form for wide right-hand sides (6: for the narrow ones as well)
        if (form >= 9) { if (ncols >= 65536 || form == 10) return launch_leaf_r<TRANS, 3>(Lt, ldl, B, ldb, ncols, s); return launch_leaf_t<TRANS, 64>(Lt, ldl, B, ldb, ncols, s); }
        if (form >= 7) { if (ncols >= 65536 || form == 8) return launch_leaf_r<TRANS, 4>(Lt, ldl, B, ldb, ncols, s); return launch_leaf_t<TRANS, 64>(Lt, ldl, B, ldb, ncols, s); }
        if (ncols >= 65536 || form == 6) return launch_leaf_m<TRANS>(Lt, ldl, B, ldb, ncols, s);
        return launch_leaf_t<TRANS, 64>(Lt, ldl, B, ldb, ncols, s);
    }
    if (form == 0 || ldb * 8 * TS_T >= (int64_t)1 << 32) {
        if (ncols >= 65536) return launch_leaf_t<TRANS, 512>(Lt, ldl, B, ldb, ncols, s);
        return launch_leaf_t<TRANS, 64>(Lt, ldl, B, ldb, ncols, s);
    }
    if (ncols >= 65536 && ncols % 256 == 0) {
        if (form == 2) return launch_leaf_s<TRANS, 128, 1>(Lt, ldl, B, ldb, ncols, s);
        if (form == 3) return launch_leaf_s<TRANS, 128, 2>(Lt, ldl, B, ldb, ncols, s);
        if (form == 4) return launch_leaf_s<TRANS, 64, 2>(Lt, ldl, B, ldb, ncols, s);
        return launch_leaf_s<TRANS, 256, 1>(Lt, ldl, B, ldb, ncols, s);
    }
    return launch_leaf_s<TRANS, 64, 1>(Lt, ldl, B, ldb, ncols, s);
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
