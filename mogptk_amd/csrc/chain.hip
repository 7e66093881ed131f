// chain.hip -- the serial chain of ONE outer block of the fused factorisation + inversion (potri.hip) as ONE persistent launch:
//     for k = 0 .. nk-1:   leaf(k): factor + invert the 128 x 128 diagonal tile      (leaf_dev.h)
//                          panel(k): L(i, k) = A(i, k) invd_k^T                      for the tile rows i > k of the block
//                          update(k): A(i, j) -= L(i, k) L(j, k)^T                   for k < j <= i
//     W_KK = L_KK^-1:      W(t, s) = -invd_t sum_{i = s}^{t-1} L(t, i) W(i, s)       (what wkk.hip did in a launch of its own)
// Before: 4 leaf launches, 6 small GEMM launches and k_wkk per block, each of them a dependent launch on the private queue that waits
// for its operands to come back from memory first (35-50 us apiece for 2-4 us of arithmetic: 490 us per block, 16 blocks per
// evaluation at N = 8192).  Here 13 workgroups stay resident on the reserved CUs for the whole block and hand tiles to each other
// through memory with agent-scope flags (MI355X_MICROARCH "inter-workgroup visibility": write-through payload stores, every storing
// wave drains vmcnt, ONE lane raises a counter; the consumer polls that word relaxed, ONE agent acquire, then plain loads):
//   workgroup 0        the leaves, one after the other; leaf(k) starts when the four row slabs of tile (k, k) have arrived
//   workgroups 1..12   worker g owns the 32-row slab g of the 384 rows below the first tile (tile i = 1 + g / 4) for panel / update --
//                      operands straight from memory into MFMA fragments, every load of a product in flight at once -- and two
//                      16-column strips of W_KK, whose finished strips stay in LDS as the B operands of the rows below (wkk.hip's
//                      layout).  The strips' work for tile row k+1 that does not need leaf(k+1) runs WHILE leaf(k+1) runs.
// Per step the critical path is leaf -> flag -> panel slab (4 workgroups) -> flag -> update of the diagonal tile -> flag -> leaf.
// Every wait is bounded: a timeout (a workgroup that never became resident) is reported through `info` instead of hanging the queue.
// Replaces the per-block share of torch.linalg.cholesky (reference gpr/model.py:246) on the critical path.
#include "mogp_model.h"
#include "leaf_dev.h"

#include <cstdlib>

namespace mogp {

#define CH_NWORK 12
#define CH_NWG (CH_NWORK + 1)
#define CH_COLS 16
#define CH_SLOT (MOGP_TILE * CH_COLS)             // one 16-column strip of a 128-row tile
#define CH_STRIP (4 * CH_SLOT)                    // three finished strips W(i, s) + the intermediate T
#define CH_LDS_BYTES (2 * CH_STRIP * 8)           // two strips per worker: 128 KB -- more than half a CU's LDS, so ONE workgroup per CU
#define CH_SPIN_LIMIT 400000u                     // polls of ~0.5 us each before a wait gives up
static_assert(CH_LDS_BYTES >= LF_LDS_BYTES, "the leaf workgroup uses the same allocation");

enum { CF_LEAF = 0, CF_DIAG = 4, CF_PAN = 8 };    // flag words of one block: LEAF[k], DIAG[k], PAN[4 k + j]

struct ChainArgs {
    double* A; int64_t ld;          // the matrix
    int t0, nk;                     // first tile of the block, tiles in the block (1 .. 4)
    double* invd; double* logdet;   // per-tile outputs of the leaves (indexed by the global tile)
    unsigned long long* info; long long info_base;
    double* Wk; int64_t ldw;        // W_KK (nk * 128 square; tiles above the diagonal untouched)
    unsigned* flags;                // MOGP_CHAIN_FLAGS zeroed words of this block
    unsigned* err;                  // one word per evaluation, zero at its start
    // inside the dataflow schedule (flow.hip): wait for the diagonal block's last update, publish W_KK, count the workgroups that are through
    unsigned* wait_flag; unsigned wait_val;
    unsigned* done_flag;
    int wt;                         // W_KK leaves with write-through stores (its readers are workgroups of a kernel that is already running)
    unsigned long long* trace;
    unsigned* diag;                 // FLOW_DIAG_* counters (null outside the dataflow schedule)
};

// ---- hand-off -------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void ch_wait(unsigned* flag, unsigned expect, unsigned* err, unsigned code, unsigned* diag = nullptr) {
    __syncthreads();                                   // this workgroup's own stores and LDS traffic are behind us
    if (threadIdx.x == 0) {
        unsigned spins = 0;
        while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < expect) {
            __builtin_amdgcn_s_sleep(2);
            if ((++spins & 127u) == 0u) {
                if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) break;
                if (spins > CH_SPIN_LIMIT) { __hip_atomic_store(err, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
                if ((spins & 8191u) == 0u) {           // a DEEP poll (~every ms of waiting): a returning read-modify-write instead of the sc1 load
                    const unsigned v = __hip_atomic_fetch_or(flag, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (diag) __hip_atomic_fetch_add(diag + FLOW_DIAG_WAIT_DEEP, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (v >= expect) { if (diag) __hip_atomic_fetch_add(diag + FLOW_DIAG_WAIT_STALE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");      // ONE buffer_inv sc1 after the match; plain loads from here on
    }
    __syncthreads();
}
// the payload went out with write-through stores (lf_st1 / lf_st2 <true>)
__device__ __forceinline__ void ch_signal(unsigned* flag) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // every storing wave drains its own stores (the asm stores are not counted)
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---- slab products: 32 rows x two 16-column tiles per wave, both operands k-contiguous, straight from memory ----------------------
// A lane (row = lane & 15, g = lane >> 4) fetches FOUR consecutive k of its row with one 32-byte load, so MFMA j of a 16-wide k block
// contracts k = 16 kk + 4 g + j in slot g -- for both operands alike, which is all the instruction needs (wkk.hip).
// acc[rt][0] += A[16 rt ..][k] B[16 c_lo ..][k] over kk < n_lo;  acc[rt][1] += A B[16 c_hi ..] over kk < n_hi  (n_lo <= n_hi)
__device__ __forceinline__ void ch_slab_product(const double* A, int64_t lda, const double* B, int64_t ldb,
                                                int c_lo, int c_hi, int n_lo, int n_hi, bool lo_on, bool hi_on, int lr, int lk,
                                                d4_t (&acc)[2][2]) {
    const int na = hi_on ? n_hi : (lo_on ? n_lo : 0);
    const d4_t* ap0 = reinterpret_cast<const d4_t*>(A + (int64_t)lr * lda + 4 * lk);
    const d4_t* ap1 = reinterpret_cast<const d4_t*>(A + (int64_t)(16 + lr) * lda + 4 * lk);
    const d4_t* blp = reinterpret_cast<const d4_t*>(B + (int64_t)(16 * c_lo + lr) * ldb + 4 * lk);
    const d4_t* bhp = reinterpret_cast<const d4_t*>(B + (int64_t)(16 * c_hi + lr) * ldb + 4 * lk);
    d4_t a0[8], a1[8], bl[8], bh[8];
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {                                   // every load of the product in flight at once: one round trip
        if (kk < na) { a0[kk] = ap0[4 * kk]; a1[kk] = ap1[4 * kk]; }
        if (lo_on && kk < n_lo) bl[kk] = blp[4 * kk];
        if (hi_on && kk < n_hi) bh[kk] = bhp[4 * kk];
    }
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
        if (lo_on && kk < n_lo) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[kk][j], bl[kk][j], acc[0][0], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[kk][j], bl[kk][j], acc[1][0], 0, 0, 0);
            }
        }
        if (hi_on && kk < n_hi) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[kk][j], bh[kk][j], acc[0][1], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[kk][j], bh[kk][j], acc[1][1], 0, 0, 0);
            }
        }
    }
}

__device__ __forceinline__ void ch_stw(int wt, double* p, double v) {
    if (wt) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else *p = v;
}

// ---- W_KK strips (the arithmetic of wkk.hip) ---------------------------------------------------------------------------------------
__device__ __forceinline__ int ch_perm(int row) { return (row & ~15) | ((row & 3) << 2) | ((row >> 2) & 3); }

__device__ __forceinline__ void ch_strip_product(const double* A0, const double* A1, int64_t lda, const double* Bs,
                                                 int lr, int lk, double sign, d4_t& acc0, d4_t& acc1) {
    const d4_t* a0p = reinterpret_cast<const d4_t*>(A0 + (int64_t)lr * lda + 4 * lk);
    const d4_t* a1p = reinterpret_cast<const d4_t*>(A1 + (int64_t)lr * lda + 4 * lk);
    d4_t a0[8], a1[8];
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) { a0[kk] = a0p[4 * kk]; a1[kk] = a1p[4 * kk]; }
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
        const double* bp = Bs + (16 * kk + lk) * CH_COLS + lr;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const double b = bp[4 * j * CH_COLS];
            acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(sign * a0[kk][j], b, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(sign * a1[kk][j], b, acc1, 0, 0, 0);
        }
    }
}

// the share of strip (column block s, columns c0 .. c0 + 15) that becomes possible once leaf(k) and the panels of step k are there:
//   W(k, s) = -invd_k T(k, s)  (k > s; T was formed one window earlier)  or  W(s, s) = invd_s  (k == s),
//   then T(k + 1, s) = sum_{i = s}^{k} L(k + 1, i) W(i, s) for the next window -- needs the panel slabs of tile row k + 1 of step k.
// Called by the whole workgroup.
__device__ __forceinline__ void ch_strip_window(const ChainArgs& g, double* ctx, int s, int c0, int k, int wave, int lane) {
    const int tid = threadIdx.x, lr = lane & 15, lk = lane >> 4;
    const int ra = 16 * wave, rb = 16 * (wave + 4);
    double* Tb = ctx + 3 * CH_SLOT;
    double* Ablk = g.A + (int64_t)g.t0 * MOGP_TILE * (g.ld + 1);
    const double* Dk = g.invd + (int64_t)(g.t0 + k) * MOGP_TILE * MOGP_TILE;
    double* Wk = g.Wk;
    if (k == s) {
        for (int e = tid; e < CH_SLOT; e += 256) {
            const int r = e >> 4, c = e & 15;
            const double v = Dk[r * MOGP_TILE + c0 + c];
            ctx[ch_perm(r) * CH_COLS + c] = v;
            ch_stw(g.wt, Wk + (int64_t)(s * MOGP_TILE + r) * g.ldw + s * MOGP_TILE + c0 + c, v);
        }
    } else {
        d4_t o0 = (d4_t){0.0, 0.0, 0.0, 0.0}, o1 = o0;
        ch_strip_product(Dk + ra * MOGP_TILE, Dk + rb * MOGP_TILE, MOGP_TILE, Tb, lr, lk, -1.0, o0, o1);
        double* slot = ctx + (k - s) * CH_SLOT;           // kept only while a row below still needs it (k - s <= 2 then)
        const bool keep = k + 1 < g.nk;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int r0 = ra + lk + 4 * r, r1 = rb + lk + 4 * r;
            if (keep) {
                slot[ch_perm(r0) * CH_COLS + lr] = o0[r];
                slot[ch_perm(r1) * CH_COLS + lr] = o1[r];
            }
            ch_stw(g.wt, Wk + (int64_t)(k * MOGP_TILE + r0) * g.ldw + s * MOGP_TILE + c0 + lr, o0[r]);
            ch_stw(g.wt, Wk + (int64_t)(k * MOGP_TILE + r1) * g.ldw + s * MOGP_TILE + c0 + lr, o1[r]);
        }
    }
    if (k + 1 >= g.nk) return;
    // the four slabs of L(k + 1, k) -- and with them every earlier panel of that tile row: a slab owner takes its steps in order
    ch_wait(g.flags + CF_PAN + 4 * k + (k + 1), 4u, g.err, 0x300u + (unsigned)k, g.diag);
    d4_t acc0 = (d4_t){0.0, 0.0, 0.0, 0.0}, acc1 = acc0;
    for (int i = s; i <= k; ++i) {
        const double* L = Ablk + (int64_t)(k + 1) * MOGP_TILE * g.ld + (int64_t)i * MOGP_TILE;
        ch_strip_product(L + (int64_t)ra * g.ld, L + (int64_t)rb * g.ld, g.ld, ctx + (i - s) * CH_SLOT, lr, lk, 1.0, acc0, acc1);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        Tb[ch_perm(ra + lk + 4 * r) * CH_COLS + lr] = acc0[r];
        Tb[ch_perm(rb + lk + 4 * r) * CH_COLS + lr] = acc1[r];
    }
    __syncthreads();
}

__global__ __launch_bounds__(256, 1) void k_chain(ChainArgs g) {
    extern __shared__ __attribute__((aligned(16))) double ch_lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nk = g.nk;
    if (g.trace && blockIdx.x == 0 && tid == 0) g.trace[0] = wall_clock64();
    // dataflow schedule: the diagonal block is complete when the update tasks of the previous panel have all reported (their tiles were
    // stored write-through by another kernel's workgroups: ch_wait's acquire makes them visible)
    if (g.wait_flag) ch_wait(g.wait_flag, g.wait_val, g.err, 0x500u, g.diag);
    if (g.trace && blockIdx.x == 0 && tid == 0) g.trace[1] = wall_clock64();
    if (blockIdx.x == 0) {
        // ---- the leaves ----
        for (int k = 0; k < nk; ++k) {
            if (k > 0) ch_wait(g.flags + CF_DIAG + k, 4u, g.err, 0x100u + (unsigned)k, g.diag);
            leaf_tile<true>(ch_lds, g.A, g.ld, g.t0 + k, g.invd, g.logdet, g.info, g.info_base, 0);
            ch_signal(g.flags + CF_LEAF + k);
        }
    } else {
        // ---- a worker: row slab `wg` of the rows below the first tile, W strips `wg` and `wg + 12` ----
        __builtin_amdgcn_s_setprio(2);
        const int wg = blockIdx.x - 1;
        const int ti = 1 + (wg >> 2), q = wg & 3;                  // tile row of the slab, slab inside the tile
        const int lr = lane & 15, lk = lane >> 4;
        double* Ablk = g.A + (int64_t)g.t0 * MOGP_TILE * (g.ld + 1);
        double* As = Ablk + (int64_t)(MOGP_TILE * ti + 32 * q) * g.ld;       // the slab's rows, column 0 of the block
        const int nstrips = 8 * (nk - 1);
        for (int k = 0; k < nk; ++k) {
            ch_wait(g.flags + CF_LEAF + k, 1u, g.err, 0x200u + (unsigned)k, g.diag);
            if (ti > k && ti < nk) {
                const double* Dk = g.invd + (int64_t)(g.t0 + k) * MOGP_TILE * MOGP_TILE;
                double* Ps = As + (int64_t)MOGP_TILE * k;
                {   // panel: P_s = A_s[:, tile k] invd_k^T, in place.  invd_k is lower triangular: column tile c needs k blocks <= c;
                    // a wave takes the column tiles w and 7 - w (9 k blocks each way round)
                    d4_t acc[2][2];
#pragma unroll
                    for (int a = 0; a < 2; ++a)
#pragma unroll
                        for (int b = 0; b < 2; ++b) acc[a][b] = (d4_t){0.0, 0.0, 0.0, 0.0};
                    ch_slab_product(Ps, g.ld, Dk, MOGP_TILE, wave, 7 - wave, wave + 1, 8 - wave, true, true, lr, lk, acc);
                    __syncthreads();                                // every wave has read the slab before anyone overwrites it
#pragma unroll
                    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                        for (int h = 0; h < 2; ++h)
#pragma unroll
                            for (int r = 0; r < 4; ++r)
                                lf_st1<true>(Ps + (int64_t)(16 * rt + 4 * r + lk) * g.ld + 16 * (h ? 7 - wave : wave) + lr, acc[rt][h][r]);
                    ch_signal(g.flags + CF_PAN + 4 * k + ti);
                }
                for (int j = k + 1; j <= ti; ++j) {                 // the diagonal tile of the next leaf first
                    ch_wait(g.flags + CF_PAN + 4 * k + j, 4u, g.err, 0x400u + (unsigned)(4 * k + j), g.diag);
                    double* Cs = As + (int64_t)MOGP_TILE * j;
                    const double* Pj = Ablk + (int64_t)MOGP_TILE * j * g.ld + (int64_t)MOGP_TILE * k;
                    const int cmax = (j == ti) ? 2 * q + 1 : 7;     // own tile row: only the columns up to the slab's last row
                    const bool lo_on = wave <= cmax, hi_on = 7 - wave <= cmax;
                    const bool last = (j == ti) && (k == ti - 1);   // these rows go to leaf(ti) next
                    d4_t acc[2][2];
#pragma unroll
                    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            const bool on = h ? hi_on : lo_on;
                            const int c = h ? 7 - wave : wave;
#pragma unroll
                            for (int r = 0; r < 4; ++r)
                                acc[rt][h][r] = on ? -Cs[(int64_t)(16 * rt + 4 * r + lk) * g.ld + 16 * c + lr] : 0.0;
                        }
                    ch_slab_product(Ps, g.ld, Pj, g.ld, wave, 7 - wave, 8, 8, lo_on, hi_on, lr, lk, acc);
#pragma unroll
                    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            const bool on = h ? hi_on : lo_on;
                            const int c = h ? 7 - wave : wave;
                            if (on) {
#pragma unroll
                                for (int r = 0; r < 4; ++r) {
                                    double* p = Cs + (int64_t)(16 * rt + 4 * r + lk) * g.ld + 16 * c + lr;
                                    if (last) lf_st1<true>(p, -acc[rt][h][r]); else *p = -acc[rt][h][r];
                                }
                            }
                        }
                    if (last) ch_signal(g.flags + CF_DIAG + ti);
                }
            }
            // ---- W_KK: what this step makes possible, underneath the next leaf ----
            for (int u = 0; u < 2; ++u) {
                const int sid = wg + CH_NWORK * u;
                if (sid >= nstrips) break;
                const int s = sid >> 3, c0 = CH_COLS * (sid & 7);
                if (k >= s) ch_strip_window(g, ch_lds + u * CH_STRIP, s, c0, k, wave, lane);
            }
            if (k == nk - 1) {      // the last diagonal block of W_KK is the last tile inverse itself (no strip covers it)
                const double* Dk = g.invd + (int64_t)(g.t0 + k) * MOGP_TILE * MOGP_TILE;
                const int nwork = (int)gridDim.x - 1;
                for (int e = tid + 256 * wg; e < MOGP_TILE * MOGP_TILE; e += 256 * nwork) {
                    const int r = e >> 7, c = e & 127;
                    ch_stw(g.wt, g.Wk + (int64_t)(k * MOGP_TILE + r) * g.ldw + k * MOGP_TILE + c, Dk[e]);
                }
            }
        }
    }
    if (g.done_flag && __hip_atomic_load(g.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {              // W_KK (and everything else this workgroup stored) has left the CU before the counter moves
        // (after a time-out anywhere the counter stays put: nothing downstream may start on a void evaluation)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            __hip_atomic_fetch_add(g.done_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (g.trace && blockIdx.x == 0) g.trace[2] = wall_clock64();
        }
    }
    // a timed-out wait anywhere: the results are garbage -- say so through the pivot report (the time-out value is below every pivot index: it wins)
    if (tid == 0 && __hip_atomic_load(g.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)
        atomicMin(g.info, (unsigned long long)MOGP_INFO_CHAIN_TIMEOUT);
}

// MOGP_CHAIN=0: the launch-per-step chain everywhere.  Otherwise the persistent kernel, unless this model fell back (chain_fallback) or the
// private stream has fewer CUs than the kernel has workgroups.
bool chain_enabled(const mogp_model* m) {
    static const bool on = !(std::getenv("MOGP_CHAIN") && std::atoi(std::getenv("MOGP_CHAIN")) == 0);
    return on && !m->no_chain && m->ctx->chain_ok;
}

int launch_chain(double* A, int64_t ld, int t0, int nk, double* invd, double* logdet, unsigned long long* info, long long info_base,
                 double* Wk, int64_t ldw, unsigned* flags, unsigned* err, hipStream_t s, const ChainFlow* flow) {
    static std::atomic<unsigned long long> attr_done{0ull};                 // one bit per device
    { int r__ = set_max_dynamic_lds(reinterpret_cast<const void*>(k_chain), CH_LDS_BYTES, attr_done); if (r__) return r__; }
    if (nk < 1 || nk > 4) { set_error("launch_chain: a block has 1 .. 4 tiles"); return -1; }
    ChainArgs g{};
    g.A = A; g.ld = ld; g.t0 = t0; g.nk = nk; g.invd = invd; g.logdet = logdet; g.info = info; g.info_base = info_base;
    g.Wk = Wk; g.ldw = ldw; g.flags = flags; g.err = err;
    if (flow) { g.wait_flag = flow->wait_flag; g.wait_val = flow->wait_val; g.done_flag = flow->done_flag; g.wt = flow->write_through; g.trace = flow->trace; g.diag = flow->diag; }
    hipLaunchKernelGGL(k_chain, dim3(nk > 1 ? CH_NWG : 2), dim3(256), CH_LDS_BYTES, s, g);
    HIP_TRY(hipGetLastError());
    return 0;
}

}  // namespace mogp
