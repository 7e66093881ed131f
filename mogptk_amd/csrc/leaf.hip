// leaf.hip -- Cholesky factor + inverse of one 128x128 diagonal tile, the serial critical path of the blocked
// factorisation (one launch per tile row).  One 256-thread workgroup, the tile lives in LDS ([128][130] fp64, row
// stride == 2 (mod 32) 8-byte units so 16x4 MFMA fragment reads are conflict-free), work is organised in 16x16
// sub-blocks:
//   POTRF  for sb = 0..7:
//     P1  waves 0-2, one ROW per lane (the 16 diagonal-block rows + 48 panel rows per wave): right-looking Cholesky of the
//         16x16 diagonal block where L[j][k] is broadcast with v_readlane (compile-time lane) -- the same instruction
//         stream IS the triangular solve for the panel rows riding along in lanes 16..63
//     P3  trailing update C_ij -= P_i P_j^T on v_mfma_f64_16x16x4_f64, blocks dealt round-robin to the 4 waves
//   TRTRI  16x16 diagonal inverses (one column per lane), then block row i = 1..7 in place:
//          T_j = sum_k L_ik W_kj (MFMA) -> LDS scratch -> W_ij = -W_ii T_j (MFMA)
// Replaces the per-tile share of torch.linalg.cholesky (reference gpr/model.py:246).
#include "mogp_internal.h"

namespace mogp {

typedef double d4_t __attribute__((ext_vector_type(4)));

#define LF_LD 130
#define LF_MAT (MOGP_TILE * LF_LD)             // doubles
#define LF_TS (7 * 256)                        // T scratch: 7 blocks of 16x16
#define LF_LDS_BYTES ((LF_MAT + LF_TS + MOGP_TILE) * 8)

__device__ __forceinline__ double readlane_d(double x, int lane) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(x), lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(x), lane);
    return __hiloint2double(hi, lo);
}

template <int K>
struct P1Step {
    static __device__ __forceinline__ void run(double (&a)[16], double* invdiag, int sb, int lane, int& fail) {
        const double d = readlane_d(a[K], K);
        if (!(d > 0.0) && fail < 0) fail = sb * 16 + K;
        const double rs = rsqrt(d);
        a[K] *= rs;
        if (lane == 0) invdiag[sb * 16 + K] = rs;
#pragma unroll
        for (int j = K + 1; j < 16; ++j) {
            const double ljk = readlane_d(a[K], j);
            a[j] = fma(-a[K], ljk, a[j]);
        }
        P1Step<K + 1>::run(a, invdiag, sb, lane, fail);
    }
};
template <>
struct P1Step<16> {
    static __device__ __forceinline__ void run(double (&)[16], double*, int, int, int&) {}
};

__global__ __launch_bounds__(256) void k_leaf128(double* A, int64_t ld, int t, double* invd, double* logdet,
                                                 unsigned long long* info) {
    extern __shared__ __attribute__((aligned(16))) double lf[];
    double* M = lf;                       // [128][LF_LD]
    double* Ts = lf + LF_MAT;             // [7][16][16]
    double* invdiag = Ts + LF_TS;         // [128]  1 / L_kk
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double* At = A + (int64_t)t * MOGP_TILE * ld + (int64_t)t * MOGP_TILE;

    // ---- load (lower 16-blocks; everything above the block diagonal is zero): 16-byte loads, 8 in flight per thread ----
    typedef double d2_t __attribute__((ext_vector_type(2)));
    for (int it = 0; it < 32; it += 8) {
        d2_t v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int idx = (it + u) * 256 + tid, r = idx >> 6, c = (idx & 63) * 2;
            v[u] = ((c >> 4) <= (r >> 4)) ? *reinterpret_cast<const d2_t*>(At + (int64_t)r * ld + c) : (d2_t){0.0, 0.0};
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int idx = (it + u) * 256 + tid, r = idx >> 6, c = (idx & 63) * 2;
            *reinterpret_cast<d2_t*>(M + r * LF_LD + c) = v[u];
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
        double a[16];
#pragma unroll
        for (int c = 0; c < 16; ++c) a[c] = (act && R < MOGP_TILE) ? M[R * LF_LD + c0 + c] : 0.0;
        __syncthreads();          // every wave has its copy of the diagonal block before wave 0 overwrites it
        if (act) {
            P1Step<0>::run(a, invdiag, sb, wave == 0 ? lane : 1, fail);
            if (R < MOGP_TILE && (wave == 0 || lane >= 16)) {
#pragma unroll
                for (int c = 0; c < 16; ++c) M[R * LF_LD + c0 + c] = (lane < 16 && c > lane) ? 0.0 : a[c];
            }
        }
        __syncthreads();
        // ---- P3: C_ij -= P_i P_j^T for sb < j <= i <= 7 ----
        const int nrem = 7 - sb;
        const int nblk = nrem * (nrem + 1) / 2;
        for (int q = wave; q < nblk; q += 4) {
            int bi = (int)((sqrtf(8.0f * (float)q + 1.0f) - 1.0f) * 0.5f);
            while ((bi + 1) * (bi + 2) / 2 <= q) ++bi;
            while (bi * (bi + 1) / 2 > q) --bi;
            const int bj = q - bi * (bi + 1) / 2;
            const int i = sb + 1 + bi, j = sb + 1 + bj;
            double* Cb = M + (i * 16 + (lane >> 4)) * LF_LD + j * 16 + (lane & 15);
            d4_t acc;
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[r] = Cb[4 * r * LF_LD];
            const double* Pa = M + (i * 16 + (lane & 15)) * LF_LD + c0 + (lane >> 4);
            const double* Pb = M + (j * 16 + (lane & 15)) * LF_LD + c0 + (lane >> 4);
#pragma unroll
            for (int k4 = 0; k4 < 4; ++k4) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-Pa[4 * k4], Pb[4 * k4], acc, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) Cb[4 * r * LF_LD] = acc[r];
        }
        __syncthreads();
    }

    // ---- log-determinant share, failure report, L back to global ----
    {
        double lg = (tid < MOGP_TILE) ? log(M[tid * LF_LD + tid]) : 0.0;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) lg += __shfl_down(lg, off, 64);
        if (lane == 0) Ts[wave] = lg;
        __syncthreads();
        if (tid == 0) {
            logdet[t] = Ts[0] + Ts[1];
            if (fail >= 0) atomicMin(info, (unsigned long long)((int64_t)t * MOGP_TILE + fail + 1));
        }
        __syncthreads();
    }
    for (int it = 0; it < 32; ++it) {
        const int idx = it * 256 + tid, r = idx >> 6, c = (idx & 63) * 2;
        d2_t v = *reinterpret_cast<const d2_t*>(M + r * LF_LD + c);
        if (c > r) v[0] = 0.0;
        if (c + 1 > r) v[1] = 0.0;
        *reinterpret_cast<d2_t*>(At + (int64_t)r * ld + c) = v;
    }

    // ---- TRTRI: diagonal 16x16 inverses, one column per lane (8 blocks x 16 columns = waves 0 and 1) ----
    if (tid < MOGP_TILE) {
        const int b = tid >> 4, c = tid & 15, o = b * 16;
        double w[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            double s = (r == c) ? 1.0 : 0.0;
#pragma unroll
            for (int k = 0; k < r; ++k) s = fma(-M[(o + r) * LF_LD + o + k], w[k], s);
            w[r] = (r < c) ? 0.0 : s * invdiag[o + r];
        }
        // all 16 lanes of a block are in one wave and every read above precedes these writes in program order
#pragma unroll
        for (int r = 0; r < 16; ++r) M[(o + r) * LF_LD + o + c] = w[r];
    }
    __syncthreads();

    // ---- TRTRI: block rows 1..7 in place ----
    for (int i = 1; i < 8; ++i) {
        for (int j = wave; j < i; j += 4) {
            d4_t acc = (d4_t){0.0, 0.0, 0.0, 0.0};
            for (int k = j; k < i; ++k) {
                const double* La = M + (i * 16 + (lane & 15)) * LF_LD + k * 16 + (lane >> 4);
                const double* Wb = M + (k * 16 + (lane >> 4)) * LF_LD + j * 16 + (lane & 15);
#pragma unroll
                for (int k4 = 0; k4 < 4; ++k4) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(La[4 * k4], Wb[4 * k4 * LF_LD], acc, 0, 0, 0);
            }
            double* Tj = Ts + j * 256 + (lane >> 4) * 16 + (lane & 15);
#pragma unroll
            for (int r = 0; r < 4; ++r) Tj[4 * r * 16] = acc[r];
        }
        __syncthreads();
        for (int j = wave; j < i; j += 4) {
            d4_t acc = (d4_t){0.0, 0.0, 0.0, 0.0};
            const double* Wa = M + (i * 16 + (lane & 15)) * LF_LD + i * 16 + (lane >> 4);
            const double* Tb = Ts + j * 256 + (lane >> 4) * 16 + (lane & 15);
#pragma unroll
            for (int k4 = 0; k4 < 4; ++k4) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-Wa[4 * k4], Tb[4 * k4 * 16], acc, 0, 0, 0);
            double* Wo = M + (i * 16 + (lane >> 4)) * LF_LD + j * 16 + (lane & 15);
#pragma unroll
            for (int r = 0; r < 4; ++r) Wo[4 * r * LF_LD] = acc[r];
        }
        __syncthreads();
    }

    double* Wt = invd + (int64_t)t * MOGP_TILE * MOGP_TILE;
    for (int it = 0; it < 32; ++it) {
        const int idx = it * 256 + tid, r = idx >> 6, c = (idx & 63) * 2;
        d2_t v = *reinterpret_cast<const d2_t*>(M + r * LF_LD + c);
        if (c > r) v[0] = 0.0;
        if (c + 1 > r) v[1] = 0.0;
        *reinterpret_cast<d2_t*>(Wt + r * MOGP_TILE + c) = v;
    }
}

int launch_potrf_trtri_tile(double* A, int64_t ld, int t, double* invd, double* logdet, unsigned long long* info, hipStream_t s) {
    static bool attr_set = false;
    if (!attr_set) {
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_leaf128), hipFuncAttributeMaxDynamicSharedMemorySize, LF_LDS_BYTES));
        attr_set = true;
    }
    hipLaunchKernelGGL(k_leaf128, dim3(1), dim3(256), LF_LDS_BYTES, s, A, ld, t, invd, logdet, info);
    HIP_TRY(hipGetLastError());
    return 0;
}

}  // namespace mogp
