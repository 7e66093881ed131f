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

#define TM_SLAB (TS_SB * 65)
// (round 4) The leaf on the matrix cores.  k_trsm_leaf above is bound by how the L operand reaches the FMA: 8 bytes per lane out of LDS for every FMA, through
// a 128 B/clk return path that four SIMDs share -- 8256 FMAs x 8 waves x 4 clk = 110 us per block row, which is what it takes (114 us; its arithmetic is 30).
// Tried on the way (all bit-identical; tools/r4_leaf.sh): L through the SCALAR cache as the FMA's SGPR operand -- one scalar-cache miss of ~700 clk per 16
// FMAs, 165 us; the off-diagonal products on v_mfma_f64_16x16x4_f64 with the solved rows re-read from memory as operands -- every sub-block a chain of global
// round trips, 125 us.  What is here: a wave (64 columns) keeps what it has solved IN REGISTERS.  The accumulator layout of v_mfma_f64_16x16x4_f64 (lane
// (n, j), register v: row j + 4 v, column n) is also its B-operand layout (lane (n, k): row 4 ks + k), so a solved 16 x 64 block, turned from one column per
// lane back into that layout through the wave's LDS slab, IS the operand of every later sub-block.  The first TR_KEEP solved sub-blocks stay in registers for
// the whole tile (16 doubles a lane each), the one solved last comes straight from the slab, and only what lies between is read back from memory (3 block
// reads a block row instead of 28), requested a whole step ahead.  Per sub-block: X_sb -= L[sb, p] X_p over the solved sub-blocks p in k_trsm_leaf's order
// (MFMA: a fused multiply-add per k like the vector ALU's -- same bits), accumulator -> slab -> one column per lane, the 16 x 16 diagonal block by
// substitution on the vector ALU exactly as k_trsm_leaf does.  L comes out of LDS: the 28 off-diagonal 16 x 16 blocks as A operands (one double a lane
// and k step, padded rows), the diagonal blocks and reciprocal pivots for the substitution.  512 threads = 512 columns per workgroup, two waves per SIMD
// (one wave's substitution under the other's MFMAs), 144 KB of LDS: 196 workgroups at configs[4], a single round of the chip.  Every global access is a
// buffer access (scalar base + row offset in an SGPR + lane offset): with 64-bit pointers per row in flight the kernel needed 100 more VGPRs.
// (With everything solved kept in registers -- ~330 of them, one wave per SIMD, 256-column workgroups -- 391 workgroups were two rounds of 46 us; before
// that, with the tile copied to LDS by a loop with a `continue`, 122 us: the 64 loads of a thread went out one at a time.)
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
    // MOGP_TRSM_LEAF: 0 the LDS-tile kernel of round 3; 7 the matrix-core kernel for wide right-hand sides only; 8 (default) for all of them.
    // Same bits either way (tools/r4_leaf.sh: the checksum of a configs[4] gradient); configs[4]: 48.3 -> 46.3 ms on one box.
    static const int form = []() { const char* e = std::getenv("MOGP_TRSM_LEAF"); return e ? atoi(e) : 8; }();
    if (form >= 7 && ldb * 8 * TS_T < (int64_t)1 << 32 && (ncols >= 65536 || form >= 8)) return launch_leaf_r<TRANS, 4>(Lt, ldl, B, ldb, ncols, s);
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

// tri (square right-hand side): the solution is wanted -- forward: IS, the right-hand side being lower triangular -- in its lower block triangle only:
// right-looking, block row i solved in and its update applied to the first i + 1 column tiles (L^-1 I and the lower half of L^-T (L^-1 I): half the
// products of the full solves; the accurate form of the exact evaluation, mogp_api.hip:factorize)
int trsm_lower(mogp_model* m, const double* L, int64_t ldl, int nb, double* B, int64_t ldb, int64_t ncols, bool trans, hipStream_t st, bool tri,
               const hipEvent_t* row_ready) {
    if (!st) st = m->st;
    if (ncols % MOGP_TILE) return fail(MOGP_EINVAL, "trsm_lower: the number of right-hand sides must be a multiple of 128");
    const int nt = (int)(ncols / MOGP_TILE);
    // Wide right-hand sides (N columns): left-looking -- block row i is updated once, by one GEMM with K = 128 i, and B is swept once.
    // Narrow ones (M x M): right-looking -- every solved block updates all remaining block rows at once (K = 128, but (nb - i) nt
    // workgroups per launch instead of nt; the matrix stays in the Infinity Cache).
    if (row_ready && (trans || tri || nt <= 32)) return fail(MOGP_EINVAL, "trsm_lower: following a running factorisation row by row is the forward left-looking form's (wide right-hand sides)");
    const bool right = tri || nt <= 32;        // (tri: a square system -- one tile row per launch would use (i + 1) of the chip's 256 CUs: 75 ms for the two solves at N = 8192, 2/3 of the accurate evaluation)
    for (int step = 0; step < nb; ++step) {
        const int i = trans ? nb - 1 - step : step;
        double* Bi = B + (int64_t)i * MOGP_TILE * ldb;
        if (row_ready) HIP_TRY(hipStreamWaitEvent(st, row_ready[i], 0));
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
        const int ntu = tri ? std::min(nt, i + 1) : nt;        // tri: block row i lives in its first i + 1 column tiles (forward: the rest is zero; transposed: the rest is the upper triangle nobody reads)
        const int64_t lc = (int64_t)ntu * MOGP_TILE;
        RC(trans ? launch_leaf<true>(Lii, ldl, Bi, ldb, lc, st) : launch_leaf<false>(Lii, ldl, Bi, ldb, lc, st));
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
            g.mode = GM_RECT; g.mt = rest; g.nt = ntu; g.K = MOGP_TILE;
            RC(gemm_call(m, g, gemm_flops(g, nullptr), st));
        }
    }
    return 0;
}

}  // namespace mogp
