// leaf_dev.h -- device code of the 128 x 128 tile factorisation + inverse (see leaf.hip for the algorithm); shared by the one-tile
// kernel k_leaf128 (leaf.hip) and the persistent chain kernel (chain.hip).
#pragma once
#include "mogp_internal.h"

namespace mogp {

typedef double d4_t __attribute__((ext_vector_type(4)));
typedef double d2_t __attribute__((ext_vector_type(2)));

#define LF_BS 18                               // row stride inside a 16x16 block
#define LF_BLK (16 * LF_BS)                    // doubles per block
#define LF_MAT (36 * LF_BLK)                   // packed lower triangle
#define LF_LDS_BYTES ((LF_MAT + MOGP_TILE + 8) * 8)

__device__ __forceinline__ int lf_blk(int bi, int bj) { return (bi * (bi + 1) / 2 + bj) * LF_BLK; }
// element (r, c), c's block <= r's block
__device__ __forceinline__ int lf_at(int r, int c) { return lf_blk(r >> 4, c >> 4) + (r & 15) * LF_BS + (c & 15); }

__device__ __forceinline__ double readlane_d(double x, int lane) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(x), lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(x), lane);
    return __hiloint2double(hi, lo);
}

// 1/sqrt(d) for the pivots: hardware seed (v_rsq_f64) + one third-order correction  y (1 + e/2 + 3 e^2/8),  e = 1 - d y^2.
// The pivots are Schur complements of a jittered Gram matrix (no denormal / overflow scaling needed); full double precision for any
// seed good to 2^-18.  Four dependent operations instead of the library routine's ~20 on the serial chain of the factorisation.
__device__ __forceinline__ double fast_rsqrt(double d) {
    const double y = __builtin_amdgcn_rsq(d);
    const double e = fma(-d * y, y, 1.0);
    return fma(y * e, fma(0.375, e, 0.5), y);
}

template <int K>
struct P1Step {
    static __device__ __forceinline__ void run(double (&a)[16], double* invdiag, int sb, bool writer, int& fail) {
        const double d = readlane_d(a[K], K);
        if (!(d > 0.0) && fail < 0) fail = sb * 16 + K;
        const double rs = fast_rsqrt(d);
        a[K] *= rs;
        if (writer) invdiag[sb * 16 + K] = rs;
        // four broadcasts, then four updates: distinct scalar pairs, so the v_readlane -> VALU hazard slots are shared instead of paid
        // per update.  The updates are pinned here: left to itself the optimiser sinks them to the column that first needs a[j], keeps
        // every broadcast (30 SGPRs per column) alive until then, overflows the scalar file and pays a v_writelane / v_readlane pair
        // per value.
#pragma unroll
        for (int j0 = K + 1; j0 < 16; j0 += 4) {
            double l[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) l[u] = (j0 + u < 16) ? readlane_d(a[K], (j0 + u < 16) ? j0 + u : 15) : 0.0;
#pragma unroll
            for (int u = 0; u < 4; ++u) if (j0 + u < 16) a[j0 + u] = fma(-a[K], l[u], a[j0 + u]);
#pragma unroll
            for (int u = 0; u < 4; ++u) if (j0 + u < 16) asm volatile("" : "+v"(a[j0 + u]));
        }
        P1Step<K + 1>::run(a, invdiag, sb, writer, fail);
    }
};
template <>
struct P1Step<16> {
    static __device__ __forceinline__ void run(double (&)[16], double*, int, bool, int&) {}
};

// number of block columns of step s's trailing update that are applied right away (the rest is deferred to the look-ahead)
__device__ __forceinline__ int lf_now(int s) { return s == 0 ? 3 : (s == 1 ? 2 : 1); }

// C(i, j) -= P(i, kc) P(j, kc)^T on 16 x 16 blocks of the packed tile (one wave)
__device__ __forceinline__ void lf_update(double* M, int i, int j, int kc, int lane) {
    double* Cb = M + lf_blk(i, j) + (lane >> 4) * LF_BS + (lane & 15);
    d4_t acc;
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[r] = Cb[4 * r * LF_BS];
    const double* Pa = M + lf_blk(i, kc) + (lane & 15) * LF_BS + (lane >> 4);
    const double* Pb = M + lf_blk(j, kc) + (lane & 15) * LF_BS + (lane >> 4);
#pragma unroll
    for (int k4 = 0; k4 < 4; ++k4) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-Pa[4 * k4], Pb[4 * k4], acc, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 4; ++r) Cb[4 * r * LF_BS] = acc[r];
}


// stores of data another workgroup of the same launch will read: write-through (sc1) -- nothing stays dirty in this XCD's L2, so the
// hand-off needs no release fence (which would write back whatever the bulk GEMMs of the same XCD have dirtied); the asm store is not
// counted by the compiler: the publisher drains vmcnt itself before it raises the flag
template <bool WT> __device__ __forceinline__ void lf_st1(double* p, double v) {
    if constexpr (WT) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else *p = v;
}
template <bool WT> __device__ __forceinline__ void lf_st2(double* p, d2_t v) {
    if constexpr (WT) asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" :: "v"(p), "v"(v) : "memory");
    else *reinterpret_cast<d2_t*>(p) = v;
}

// One 256-thread workgroup; lf = LF_LDS_BYTES of LDS.  WT: the tile inverse leaves the CU with write-through (sc1) stores, for a consumer
// workgroup of the SAME launch (chain.hip: the caller drains vmcnt and publishes a flag); false: plain stores (a kernel boundary follows).
template <bool WT>
__device__ __forceinline__ void leaf_tile(double* lf, double* A, int64_t ld, int t, double* invd, double* logdet,
                                          unsigned long long* info, long long info_base, int store_L) {
    double* M = lf;                       // 36 packed lower blocks
    double* invdiag = lf + LF_MAT;        // [128]  1 / L_kk
    double* red = invdiag + MOGP_TILE;    // [8]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double* At = A + (int64_t)t * MOGP_TILE * ld + (int64_t)t * MOGP_TILE;
    __builtin_amdgcn_s_setprio(3);        // serial critical path: outrank co-resident trailing-update waves

    // ---- load the lower 16-blocks: 16-byte loads, all 32 of a thread in flight at once (one memory round trip) ----
    {
        d2_t v[32];
#pragma unroll
        for (int u = 0; u < 32; ++u) {
            const int idx = u * 256 + tid, r = idx >> 6, c = (idx & 63) * 2;
            v[u] = ((c >> 4) <= (r >> 4)) ? *reinterpret_cast<const d2_t*>(At + (int64_t)r * ld + c) : (d2_t){0.0, 0.0};
        }
#pragma unroll
        for (int u = 0; u < 32; ++u) {
            const int idx = u * 256 + tid, r = idx >> 6, c = (idx & 63) * 2;
            if ((c >> 4) <= (r >> 4)) *reinterpret_cast<d2_t*>(M + lf_at(r, c)) = v[u];
        }
    }
    __syncthreads();

    int fail = -1;
    for (int sb = 0; sb < 8; ++sb) {
        const int c0 = sb * 16;
        // ---- P1: waves 0..2.  Lanes 0-15 of every wave hold the 16 diagonal-block rows (redundantly, so each wave has
        // the L[j][k] broadcasts in its own registers), lanes 16-63 hold 48 panel rows: wave w covers c0+16+48w .. +47.
        const bool act = wave < 3 && (wave == 0 || c0 + 16 + 48 * wave < MOGP_TILE);
        const int R = lane < 16 ? c0 + lane : c0 + 16 + 48 * wave + (lane - 16);
        const bool rowok = act && R < MOGP_TILE;
        double* rowp = M + (rowok ? lf_blk(R >> 4, sb) + (R & 15) * LF_BS : 0);
        double a[16];
#pragma unroll
        for (int c = 0; c < 16; ++c) a[c] = rowok ? rowp[c] : 0.0;
        __syncthreads();          // every wave has its copy of the diagonal block before wave 0 overwrites it
        if (act) {
            P1Step<0>::run(a, invdiag, sb, wave == 0 && lane == 0, fail);
            if (rowok && (wave == 0 || lane >= 16)) {
#pragma unroll
                for (int c = 0; c < 16; ++c) rowp[c] = (lane < 16 && c > lane) ? 0.0 : a[c];
            }
        } else if (sb > 0) {
            // look-ahead: the part of the PREVIOUS step's trailing update that the micro-panel above does not read (block columns
            // >= sb + nc(sb - 1), from panel column sb - 1) runs here, on the waves that have no rows in P1 (wave 3; waves 2 and 1 once
            // their panel rows are gone), hidden behind the serial factorisation
            const int first_idle = sb >= 4 ? 1 : 2, nidle = 4 - first_idle;
            const int j0 = sb + lf_now(sb - 1);                 // first deferred block column
            const int nr = 8 - j0, nb2 = nr > 0 ? nr * (nr + 1) / 2 : 0;
            for (int q = wave - first_idle; q < nb2; q += nidle) {
                int bi = (int)((sqrtf(8.0f * (float)q + 1.0f) - 1.0f) * 0.5f);
                while ((bi + 1) * (bi + 2) / 2 <= q) ++bi;
                while (bi * (bi + 1) / 2 > q) --bi;
                const int bj = q - bi * (bi + 1) / 2;
                lf_update(M, j0 + bi, j0 + bj, sb - 1, lane);
            }
        }
        __syncthreads();
        // ---- P3, the part done right away: block columns sb + 1 .. sb + nc(sb) from panel column sb (the next micro-panel reads the
        // first of them; taking three / two columns in the first two steps keeps the deferred rest within what two idle waves finish
        // behind one P1) ----
        {
            const int jn = min(sb + lf_now(sb), 7);
            int cnt = 0;
            for (int j = sb + 1; j <= jn; ++j)
                for (int i = j; i < 8; ++i, ++cnt)
                    if ((cnt & 3) == wave) lf_update(M, i, j, sb, lane);
        }
        __syncthreads();
    }

    // ---- log-determinant share, failure report ----
    {
        double lg = (tid < MOGP_TILE) ? log(M[lf_at(tid, tid)]) : 0.0;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) lg += __shfl_down(lg, off, 64);
        if (lane == 0) red[wave] = lg;
        __syncthreads();
        if (tid == 0) {
            logdet[t] = red[0] + red[1];
            if (fail >= 0) atomicMin(info, (unsigned long long)(info_base + (int64_t)t * MOGP_TILE + fail + 1));
        }
    }
    // The factor L_kk itself is NOT written back on the exact-GP path: every consumer there works with the tile inverse (panel = panel *
    // invd^T, W_KK rows, put_diag_tiles before TRTRI) and the log-determinant is taken here, so the diagonal tile of A keeps its (consumed)
    // input.  The triangular-solve path (trsm.hip, Titsias K_uu) needs L_kk itself: store_L writes the lower triangle, zeros above.
    if (store_L) {
        for (int it = 0; it < 32; ++it) {
            const int idx = it * 256 + tid, r = idx >> 6, c = (idx & 63) * 2;
            d2_t v = (d2_t){0.0, 0.0};
            if ((c >> 4) <= (r >> 4)) v = *reinterpret_cast<const d2_t*>(M + lf_at(r, c));
            if (c > r) v[0] = 0.0;
            if (c + 1 > r) v[1] = 0.0;
            *reinterpret_cast<d2_t*>(At + (int64_t)r * ld + c) = v;
        }
        __syncthreads();             // the inverse below overwrites the diagonal blocks of the LDS image
    }

    // ---- TRTRI: diagonal 16x16 inverses, one column per lane (8 blocks x 16 columns = waves 0 and 1) ----
    if (tid < MOGP_TILE) {
        const int b = tid >> 4, c = tid & 15;
        double* Db = M + lf_blk(b, b);
        double w[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            double s = (r == c) ? 1.0 : 0.0;
#pragma unroll
            for (int k = 0; k < r; ++k) s = fma(-Db[r * LF_BS + k], w[k], s);
            w[r] = (r < c) ? 0.0 : s * invdiag[b * 16 + r];
        }
        // all 16 lanes of a block are in one wave and every read above precedes these writes in program order
#pragma unroll
        for (int r = 0; r < 16; ++r) Db[r * LF_BS + c] = w[r];
    }
    __syncthreads();

    // ---- TRTRI, off-diagonal blocks: W_ij = -W_ii sum_{k = j}^{i-1} L_ik W_kj.  Block COLUMNS of W are independent, so every wave takes
    // two of them (j and 7 - j: 8 + 1, 7 + 2, ... blocks -- balanced) and walks down the rows with NO barrier: L and the diagonal
    // inverses are only read, and the wave keeps its finished blocks W_kj in registers -- a block in MFMA accumulator layout (register
    // r of a lane = row 4 r + lane / 16, column lane % 16) is exactly the B operand of k-group r.  Results go straight to the global
    // tile inverse; the loop at the end writes only the diagonal blocks and the zeros above them.
    double* Wt = invd + (int64_t)t * MOGP_TILE * MOGP_TILE;
    const int lr = lane & 15, lk = lane >> 4;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        const int j = __builtin_amdgcn_readfirstlane(half == 0 ? wave : 7 - wave);
        d4_t wcol[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) wcol[k] = (d4_t){0.0, 0.0, 0.0, 0.0};
        {
            const double* Dj = M + lf_blk(j, j) + lk * LF_BS + lr;            // W_jj in accumulator layout
            d4_t v;
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = Dj[4 * r * LF_BS];
#pragma unroll
            for (int k = 0; k < 8; ++k) if (k == j) wcol[k] = v;
        }
#pragma unroll
        for (int i = 1; i < 8; ++i) {
            if (i > j) {
                d4_t tacc = (d4_t){0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int k = 0; k < 7; ++k) {
                    if (k >= j && k < i) {
                        const double* La = M + lf_blk(i, k) + lr * LF_BS + lk;
#pragma unroll
                        for (int k4 = 0; k4 < 4; ++k4) tacc = __builtin_amdgcn_mfma_f64_16x16x4f64(La[4 * k4], wcol[k][k4], tacc, 0, 0, 0);
                    }
                }
                d4_t acc = (d4_t){0.0, 0.0, 0.0, 0.0};
                const double* Wa = M + lf_blk(i, i) + lr * LF_BS + lk;
#pragma unroll
                for (int k4 = 0; k4 < 4; ++k4) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-Wa[4 * k4], tacc[k4], acc, 0, 0, 0);
                wcol[i] = acc;
                double* Wo = Wt + (int64_t)(16 * i + lk) * MOGP_TILE + 16 * j + lr;
#pragma unroll
                for (int r = 0; r < 4; ++r) lf_st1<WT>(Wo + 4 * r * MOGP_TILE, acc[r]);
            }
        }
    }

    // diagonal 16 x 16 blocks of the tile inverse (from LDS) and zeros above them; the blocks below were written by the column waves
    for (int it = 0; it < 32; ++it) {
        const int idx = it * 256 + tid, r = idx >> 6, c = (idx & 63) * 2;
        if ((c >> 4) < (r >> 4)) continue;
        d2_t v = (d2_t){0.0, 0.0};
        if ((c >> 4) == (r >> 4)) v = *reinterpret_cast<const d2_t*>(M + lf_at(r, c));
        if (c > r) v[0] = 0.0;
        if (c + 1 > r) v[1] = 0.0;
        lf_st2<WT>(Wt + r * MOGP_TILE + c, v);
    }
}


}  // namespace mogp
