// linalg.hip -- fp64 dense linear algebra for the exact-GP path on gfx950:
//   * k_gemm            : 128x128-tile GEMM on v_mfma_f64_16x16x4_f64, four operand layouts, tile-grid decodes for the
//                         shapes the factorisation needs (panel solve, SYRK trailing update, TRTRI levels, LAUUM, TRMM)
//   (the 128x128 leaf factor+inverse kernel lives in leaf.hip)
//   * small bandwidth-bound vector kernels (triangular mat-vec for alpha, row reductions for the predictive variance)
// These replace torch.linalg.cholesky / cholesky_solve / solve_triangular at reference gpr/model.py:246,452,470-472 and
// the O(N^3) dense solves of their autograd backward nodes.
#include "mogp_internal.h"

#include <cstdlib>

namespace mogp {

typedef double d4_t __attribute__((ext_vector_type(4)));
typedef double d2_t __attribute__((ext_vector_type(2)));

#define GEMM_BK 16
#define LDS_ROWK 18     // [TM][18] doubles: row stride == 2 (mod 32) in 8-byte units -> conflict-free ds_read_b64 fragments

__device__ __forceinline__ int tri_row(int b) {
    int r = (int)((sqrt(8.0 * (double)b + 1.0) - 1.0) * 0.5);
    while ((r + 1) * (r + 2) / 2 <= b) ++r;
    while (r * (r + 1) / 2 > b) --r;
    return r;
}

template <int WTM, int WTN, int NWJ = 2, int NWI = 2> struct GemmCfg {
    // workgroup tile TMR x TNC: NWI x NWJ waves, each WTM x WTN MFMA tiles of 16x16 (2 x 4 waves of 4 x 2 tiles: 128 x 128 on eight waves)
    static constexpr int NT = 64 * NWI * NWJ;                       // threads
    static constexpr int TMR = 16 * NWI * WTM, TNC = 16 * NWJ * WTN;
    static constexpr int COLK_A = TMR + 16, COLK_B = TNC + 16;     // [16][T+16] doubles: row stride == 16 (mod 32)
    static constexpr int OPER_A = (TMR * LDS_ROWK > 16 * COLK_A) ? TMR * LDS_ROWK : 16 * COLK_A;
    static constexpr int OPER_B = (TNC * LDS_ROWK > 16 * COLK_B) ? TNC * LDS_ROWK : 16 * COLK_B;
    static constexpr int LDS_BYTES = 2 * (OPER_A + OPER_B) * 8;
    static constexpr int EPT_A = TMR * GEMM_BK / NT, EPT_B = TNC * GEMM_BK / NT;     // doubles staged per thread (8, 4 or 2)
};

// C(TMR x TNC tile) = alpha * sum_k A[i,k] B[j,k] + beta * C on v_mfma_f64_16x16x4_f64.
// <4,4>: 128x128 tiles (throughput shape).  <2,4> / <2,2>: 64x128 / 64x64 tiles for the latency-bound launches of the
// Cholesky chain (more workgroups, less MFMA work each); 64x128 keeps the panel solve in place (a workgroup owns its rows).
#ifdef GEMM_TIMING
__device__ unsigned long long g_gemm_tim[8 * 8192];       // per workgroup: entry, loop start, loop end, exit (wall clock, 100 MHz), HW_ID, XCC_ID
#define GEMM_STAMP(i) do { if (threadIdx.x == 0) g_gemm_tim[8 * blockIdx.x + (i)] = wall_clock64(); } while (0)
#else
#define GEMM_STAMP(i)
#endif
template <int AKM, int BKM, int WTM, int WTN, int NWJ = 2, int NWI = 2, bool FLOWH = false>
__global__ __launch_bounds__(64 * NWI * NWJ, NWI * NWJ / 2) void k_gemm(GemmArgs g) {
    GEMM_STAMP(0);
    if (FLOWH && g.fl_nwait > 0) {      // inside the dataflow schedule: the operands come from workgroups of a kernel that is still running
        // (the bound: 1.6 M polls, ~0.28 s.  Round 5's soak -- tools/flow_soak.py, profiles/r5_flow_soak.txt -- saw the dataflow kernel stand still for 60 - 70 ms once in a few
        // thousand evaluations, every one of its workgroups, and go on by itself; with the earlier 400 000 polls this wait gave up just before it did)
        if (threadIdx.x == 0) {
            for (int k = 0; k < g.fl_nwait; ++k) {
                unsigned spins = 0;
                while (__hip_atomic_load(g.fl_flags + g.fl_widx[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < g.fl_wval[k]) {
                    __builtin_amdgcn_s_sleep(2);
                    if ((++spins & 127u) == 0u) {
                        if (__hip_atomic_load(g.fl_err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) break;
                        if (spins > (g.fl_spins ? g.fl_spins : 1600000u)) { __hip_atomic_store(g.fl_err, 0x800u + (unsigned)k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
                        if ((spins & 8191u) == 0u) {       // a DEEP poll (chain.hip:ch_wait): what does the memory side say?
                            const unsigned v = __hip_atomic_fetch_or(g.fl_flags + g.fl_widx[k], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            if (g.fl_diag) __hip_atomic_fetch_add(g.fl_diag + FLOW_DIAG_WAIT_DEEP, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            if (v >= g.fl_wval[k]) { if (g.fl_diag) __hip_atomic_fetch_add(g.fl_diag + FLOW_DIAG_WAIT_STALE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
                        }
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();
        // a wait that gave up (here or anywhere in the schedule): the evaluation is void and will be repeated -- no product of unfinished operands, no signal
        if (__hip_atomic_load(g.fl_err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
            if (threadIdx.x == 0 && g.sk_info) atomicMin(g.sk_info, (unsigned long long)MOGP_INFO_CHAIN_TIMEOUT);
            return;
        }
    }
    using Cfg = GemmCfg<WTM, WTN, NWJ, NWI>;
    constexpr int TMR = Cfg::TMR, TNC = Cfg::TNC, COLK_A = Cfg::COLK_A, COLK_B = Cfg::COLK_B, NT = Cfg::NT;
    constexpr int OPER_A = Cfg::OPER_A, OPER_B = Cfg::OPER_B, EPT_A = Cfg::EPT_A, EPT_B = Cfg::EPT_B;
    constexpr int NQ_A = EPT_A / 2, NQ_B = EPT_B / 2, TPR_A = GEMM_BK / EPT_A, TPR_B = GEMM_BK / EPT_B;
    constexpr int TPK = NT / GEMM_BK;                                     // threads per k row of a k-major operand block
    extern __shared__ __attribute__((aligned(16))) double gemm_lds[];     // [2 buffers][A operand | B operand]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wi = wave / NWJ, wj = wave % NWJ;

    // ---- which tile, which k range ----
    const double* Ap;
    const double* Bp;
    double* Cp;
    int kt;
    bool fresh = false;                  // this tile row starts from zero although the launch accumulates (beta0_from)
    if (g.mode == GM_TASKS) {
        int bid = blockIdx.x;
        if (g.task_chunked && gridDim.x >= 64) {            // equal-cost tasks in row-major tile order: one contiguous chunk of the list per XCD (see below)
            const int q = gridDim.x >> 3, r = gridDim.x & 7, x = bid & 7;
            bid = x * q + min(x, r) + (bid >> 3);
        }
        const GemmTask t = g.tasks[bid];
        Ap = g.A + t.a_off; Bp = g.B + t.b_off; Cp = g.C + t.c_off; kt = t.kt;
        fresh = g.beta0_from > 0 && t.pad >= g.beta0_from;  // pad = tile row + 1 when the list carries it (0: never fresh)
    } else {
        // XCD-aware tile order: workgroup b is dispatched to XCD b % 8, each with its own L2.  Giving every XCD one contiguous
        // chunk of the (row-major) tile list keeps a panel row inside one L2 instead of all eight (measured: 2.8x the algorithmic
        // HBM traffic without it).  Only for the modes whose tiles all cost the same (equal chunks = equal work).
        int bid = blockIdx.x, grid = gridDim.x, ks = 0;
        bool placed = false;
        if (g.ksplit > 1) {            // split K: slice ks of the k range, accumulated into its own copy of C (c_split apart); the caller adds them
            grid /= g.ksplit;
            if (g.ksplit_xcd) {        // slices x, x + 8, ... on XCD x, one after the other: every workgroup resident on an XCD reads the same k window
                const int x = bid & 7, local = bid >> 3, sl = local / grid;      // (slice-major numbering put ~4 slices on every XCD at once: 4 k windows
                ks = x + 8 * sl;                                                   // x 12 row panels per k step against a 4 MB L2 -- 12.4 GB fetched for 1.6 GB of v)
                bid = local - sl * grid;
                placed = true;
            } else {
                ks = bid / grid;
                bid -= ks * grid;
            }
        }
        if (!placed && (g.mode == GM_RECT || g.mode == GM_RECT_LOWER || g.mode == GM_LOWER || g.mode == GM_KLO_J || g.mode == GM_KHI_J)) {
            const int q = grid >> 3, r = grid & 7, x = bid & 7;
            if (grid >= 64) bid = x * q + min(x, r) + (bid >> 3);
        }
        int ti, tj;
        if (g.mode == GM_LOWER || g.mode == GM_LAUUM) { ti = tri_row(bid); tj = bid - ti * (ti + 1) / 2; }
        else if (g.col_major) { tj = bid / g.mt; ti = bid - tj * g.mt; }
        else { ti = bid / g.nt; tj = bid - ti * g.nt; }
        if (g.mode == GM_RECT_LOWER && (ti + 1) * TMR <= tj * TNC) return;        // tile entirely above the diagonal
        fresh = g.beta0_from > 0 && ti >= g.beta0_from - 1;
        if (g.row_mod > 1 && (((ti >> g.row_shift) + g.row_off) % g.row_mod) != g.row_rem) return;  // tile row owned by another rank
        int64_t k0 = 0, k1 = g.K;
        if (g.mode == GM_LAUUM || g.mode == GM_KLO_I) k0 = (int64_t)ti * TMR;
        if (g.mode == GM_KLO_J) k0 = (int64_t)tj * TNC;
        if (g.mode == GM_KHI_J) k1 = min((int64_t)g.K, (int64_t)(tj + 1) * TNC);
        if (g.mode == GM_KHI_I) k1 = min((int64_t)g.K, (int64_t)(ti + 1) * TMR);
        if (g.ksplit > 1) {
            const int64_t len = ((k1 - k0) / g.ksplit) / GEMM_BK * GEMM_BK;
            k0 += ks * len;
            if (ks + 1 < g.ksplit) k1 = k0 + len;
        }
        kt = (int)((k1 - k0) / GEMM_BK);
        Ap = g.A + (AKM ? k0 * g.lda + (int64_t)ti * TMR : (int64_t)ti * TMR * g.lda + k0);
        Bp = g.B + (BKM ? k0 * g.ldb + (int64_t)tj * TNC : (int64_t)tj * TNC * g.ldb + k0);
        Cp = g.C + (int64_t)ti * TMR * g.ldc + (int64_t)tj * TNC + (int64_t)ks * g.c_split;
    }

#ifdef GEMM_TIMING
    if (kt >= 0) GEMM_STAMP(6);              // the kernel arguments have arrived
#endif
    // ---- global -> register staging map: EPT doubles (16-byte loads) per thread per operand ----
    // k-contiguous operand: thread -> (row = tid / TPR, EPT k's);  k-major operand: thread -> (k row = tid / 16, EPT i's)
    const int64_t a_g = AKM ? (int64_t)(tid / TPK) * g.lda + (tid % TPK) * EPT_A : (int64_t)(tid / TPR_A) * g.lda + (tid % TPR_A) * EPT_A;
    const int64_t b_g = BKM ? (int64_t)(tid / TPK) * g.ldb + (tid % TPK) * EPT_B : (int64_t)(tid / TPR_B) * g.ldb + (tid % TPR_B) * EPT_B;
    const int a_l = AKM ? (tid / TPK) * COLK_A + (tid % TPK) * EPT_A : (tid / TPR_A) * LDS_ROWK + (tid % TPR_A) * EPT_A;
    const int b_l = BKM ? (tid / TPK) * COLK_B + (tid % TPK) * EPT_B : (tid / TPR_B) * LDS_ROWK + (tid % TPR_B) * EPT_B;
    const int64_t a_step = AKM ? (int64_t)GEMM_BK * g.lda : GEMM_BK;
    const int64_t b_step = BKM ? (int64_t)GEMM_BK * g.ldb : GEMM_BK;

    // beta != 0: start the accumulators at (beta/alpha) * C so the read of C overlaps the first operand loads and the
    // epilogue is a pure store of alpha * acc.
    const int crow = wi * (TMR / NWI) + (lane >> 4), ccol = wj * (TNC / NWJ) + (lane & 15);
    d4_t acc[WTM][WTN];
    if (g.beta != 0.0 && !fresh) {
        const double sc = g.beta / g.alpha;
#pragma unroll
        for (int m = 0; m < WTM; ++m)
#pragma unroll
            for (int n = 0; n < WTN; ++n)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[m][n][r] = sc * Cp[(int64_t)(crow + m * 16 + 4 * r) * g.ldc + ccol + n * 16];
    } else {
#pragma unroll
        for (int m = 0; m < WTM; ++m)
#pragma unroll
            for (int n = 0; n < WTN; ++n) acc[m][n] = (d4_t){0.0, 0.0, 0.0, 0.0};
    }
#ifdef GEMM_TIMING
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    GEMM_STAMP(7);                           // C has arrived
#endif
    if (WTM < 4) __builtin_amdgcn_s_setprio(2);      // chain (latency-bound) variants outrank co-resident bulk waves

    d2_t ra[NQ_A], rb[NQ_B];
    if (kt > 0) {
        const d2_t* pa = reinterpret_cast<const d2_t*>(Ap + a_g);
        const d2_t* pb = reinterpret_cast<const d2_t*>(Bp + b_g);
#pragma unroll
        for (int q = 0; q < NQ_A; ++q) ra[q] = pa[q];
#pragma unroll
        for (int q = 0; q < NQ_B; ++q) rb[q] = pb[q];
    }
    // fragment read offsets inside an operand buffer (per m / n add 16 rows)
    const int fa = AKM ? (lane >> 4) * COLK_A + wi * (TMR / NWI) + (lane & 15) : (wi * (TMR / NWI) + (lane & 15)) * LDS_ROWK + (lane >> 4);
    const int fb = BKM ? (lane >> 4) * COLK_B + wj * (TNC / NWJ) + (lane & 15) : (wj * (TNC / NWJ) + (lane & 15)) * LDS_ROWK + (lane >> 4);
    constexpr int fa_m = AKM ? 16 : 16 * LDS_ROWK, fa_k = AKM ? 4 * COLK_A : 4;
    constexpr int fb_n = BKM ? 16 : 16 * LDS_ROWK, fb_k = BKM ? 4 * COLK_B : 4;

    // ---- k loop, software pipelined so that nothing but the MFMAs sits on the issue path of a wave:
    //   * fragments are read one k4 step ahead (two register sets), the first step of a block right behind the barrier of the block before
    //     it and under that block's last 16 MFMAs;
    //   * block kb + 1 goes registers -> LDS in the MIDDLE of block kb (its loads were issued a whole block earlier), followed at once by
    //     the global loads of block kb + 2: the ds_write latency and the barrier skew hide under the third MFMA group;
    //   * the barrier waits for LDS traffic only (lgkmcnt) -- __syncthreads() would also drain the global loads just issued.
    // Before: write, barrier, first fragment read were exposed in every block (a lone workgroup on a CU reached 0.79 of the MFMA rate).
    auto load_block = [&](int kb) {
        const d2_t* pa = reinterpret_cast<const d2_t*>(Ap + a_g + (int64_t)kb * a_step);
        const d2_t* pb = reinterpret_cast<const d2_t*>(Bp + b_g + (int64_t)kb * b_step);
#pragma unroll
        for (int q = 0; q < NQ_A; ++q) ra[q] = pa[q];
#pragma unroll
        for (int q = 0; q < NQ_B; ++q) rb[q] = pb[q];
    };
    auto write_block = [&](int buf) {
        double* sa = gemm_lds + buf * (OPER_A + OPER_B);
        double* sb = sa + OPER_A;
#pragma unroll
        for (int q = 0; q < NQ_A; ++q) *reinterpret_cast<d2_t*>(sa + a_l + 2 * q) = ra[q];
#pragma unroll
        for (int q = 0; q < NQ_B; ++q) *reinterpret_cast<d2_t*>(sb + b_l + 2 * q) = rb[q];
    };
    auto read_frag = [&](double (&av)[WTM], double (&bv)[WTN], int buf, int k4) {
        const double* sa = gemm_lds + buf * (OPER_A + OPER_B);
        const double* sb = sa + OPER_A;
#pragma unroll
        for (int m = 0; m < WTM; ++m) av[m] = sa[fa + m * fa_m + k4 * fa_k];
#pragma unroll
        for (int n = 0; n < WTN; ++n) bv[n] = sb[fb + n * fb_n + k4 * fb_k];
    };
    auto mma = [&](const double (&av)[WTM], const double (&bv)[WTN]) {
#pragma unroll
        for (int m = 0; m < WTM; ++m)
#pragma unroll
            for (int n = 0; n < WTN; ++n)
                acc[m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[m], bv[n], acc[m][n], 0, 0, 0);
    };
#define GEMM_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
    static_assert(GEMM_BK == 16, "the pipeline below is written for four k4 steps per block");
    if (kt > 0) {
        write_block(0);
        load_block(min(1, kt - 1));
        GEMM_LDS_BARRIER();
        GEMM_STAMP(1);
        double a0[WTM], b0[WTN], a1[WTM], b1[WTN];
        read_frag(a0, b0, 0, 0);
        for (int kb = 0; kb < kt; ++kb) {
            const int buf = kb & 1;
            read_frag(a1, b1, buf, 1);
            __builtin_amdgcn_sched_barrier(0);
            mma(a0, b0);
            __builtin_amdgcn_sched_barrier(0);
            read_frag(a0, b0, buf, 2);
            __builtin_amdgcn_sched_barrier(0);
            mma(a1, b1);
            __builtin_amdgcn_sched_barrier(0);
            write_block(buf ^ 1);                      // unconditional (the last block rewrites what nobody reads; its loads are clamped):
            load_block(min(kb + 2, kt - 1));           // the compiler then counts the LDS queue exactly and waits for the fragments only
            read_frag(a1, b1, buf, 3);
            __builtin_amdgcn_sched_barrier(0);
            mma(a0, b0);
            __builtin_amdgcn_sched_barrier(0);
            GEMM_LDS_BARRIER();
            read_frag(a0, b0, buf ^ 1, 0);
            __builtin_amdgcn_sched_barrier(0);
            mma(a1, b1);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
#undef GEMM_LDS_BARRIER
    GEMM_STAMP(2);

    // ---- epilogue: C/D layout of v_mfma_f64_16x16x4_f64: col = lane & 15, row = (lane >> 4) + 4 * reg ----
#pragma unroll
    for (int m = 0; m < WTM; ++m)
#pragma unroll
        for (int n = 0; n < WTN; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                double* cp = Cp + (int64_t)(crow + m * 16 + 4 * r) * g.ldc + ccol + n * 16;
                if (FLOWH && g.fl_wt) __hip_atomic_store(cp, g.alpha * acc[m][n][r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                else *cp = g.alpha * acc[m][n][r];
            }
    if (FLOWH && g.fl_sig) {                // the tile has left the CU (write-through stores, drained by every wave) before its counter moves
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) {
            const int trow = (int)((Cp - g.C) / ((int64_t)TMR * g.ldc));
            __hip_atomic_fetch_add(g.fl_flags + g.fl_sig_base + (unsigned)(trow >> g.fl_sig_shift), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (__hip_atomic_load(g.fl_err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u && g.sk_info) atomicMin(g.sk_info, (unsigned long long)MOGP_INFO_CHAIN_TIMEOUT);
        }
    }
#ifdef GEMM_TIMING
    __builtin_amdgcn_s_waitcnt(0);          // the stores are acknowledged
    GEMM_STAMP(3);
    if (threadIdx.x == 0) { g_gemm_tim[8 * blockIdx.x + 4] = __builtin_amdgcn_s_getreg((31 << 11) | 4); g_gemm_tim[8 * blockIdx.x + 5] = __builtin_amdgcn_s_getreg((3 << 11) | 20); }
#endif
}

// ---- stream-K form of the 128 x 128-tile kernel -------------------------------------------------------------------------------------
// A launch of T equal tiles on S workgroup slots takes ceil(T / S) rounds: 2080 tiles on 512 slots pay five rounds for 4.06 rounds of work,
// the 782-tile launches of the sparse models' substitutions two rounds for 1.53, and a launch of 16 tiles on the critical path of the
// factorisation takes a whole tile's latency however many CUs idle.  Here the launch is S workgroups, and workgroup s takes the s-th
// S-th of the launch's k ITERATIONS (tile by tile, k block by k block): at most one tile that started in the span before it, whole tiles,
// and at most one tile that ends in a later span.  A tile cut that way is finished by its OWNER, the span holding its first k block:
// the spans after it compute their k ranges first thing (nothing to wait for), leave alpha * (their sum) in a slot of a workspace
// (write-through stores, then one agent-scope flag per span -- chain.hip's hand-off), and the owner, which reaches that tile LAST in its
// own span, adds the slots in span order to its own sum and writes C.  Fixed order: bit-reproducible (not bit-identical to the unsplit
// kernel: a cut tile's sum is associated differently).  No workgroup ever waits before it has produced what others wait for, and a span's
// successors have higher span numbers inside an XCD's chunk: a launch that does not get all S slots at once still drains.
// Supported: GM_RECT, GM_RECT_LOWER, GM_LOWER (equal k range per tile), GM_KHI_J / GM_KLO_J (k range by tile column); eight-wave tile only.
struct SkTile { int ti, tj, kt, kbase, start; bool skip; };
#define SK_U(x) __builtin_amdgcn_readfirstlane(x)      // launch-uniform values: keep them in scalar registers

template <int TMR, int TNC>
__device__ __forceinline__ int sk_col_kt(const GemmArgs& g, int tj) {
    if (g.mode == GM_KHI_J) return (int)(min((int64_t)g.K, (int64_t)(tj + 1) * TNC) / GEMM_BK);
    if (g.mode == GM_KLO_J) return (int)((g.K - (int64_t)tj * TNC) / GEMM_BK);
    return g.K / GEMM_BK;
}
// the tile holding iteration `it` of the launch's flattened (tile, k block) space
template <int TMR, int TNC>
__device__ __forceinline__ SkTile sk_locate(const GemmArgs& g, int it, int rowcost) {
    SkTile t;
    t.kbase = 0; t.skip = false;
    if (g.mode == GM_KHI_J || g.mode == GM_KLO_J) {
        t.ti = SK_U(it / rowcost);
        int r = it - t.ti * rowcost, tj = 0, c = sk_col_kt<TMR, TNC>(g, 0);
        t.start = t.ti * rowcost;
        while (r >= c) { r -= c; t.start += c; ++tj; c = sk_col_kt<TMR, TNC>(g, tj); }
        t.tj = SK_U(tj); t.kt = SK_U(c); t.start = SK_U(t.start);
        if (g.mode == GM_KLO_J) t.kbase = tj * TNC / GEMM_BK;
        return t;
    }
    const int ktu = g.K / GEMM_BK;
    const int b = SK_U(it / ktu);
    t.start = b * ktu; t.kt = ktu;
    if (g.mode == GM_LOWER) { t.ti = SK_U(tri_row(b)); t.tj = b - t.ti * (t.ti + 1) / 2; }
    else { t.ti = SK_U(b / g.nt); t.tj = b - t.ti * g.nt; }
    t.skip = g.mode == GM_RECT_LOWER && (t.ti + 1) * TMR <= t.tj * TNC;
    return t;
}

template <int AKM, int BKM>
__global__ __launch_bounds__(512, 4) void k_gemm_sk(GemmArgs g) {
    constexpr int WTM = 4, WTN = 2, NWJ = 4, NWI = 2;
    using Cfg = GemmCfg<WTM, WTN, NWJ, NWI>;
    constexpr int TMR = Cfg::TMR, TNC = Cfg::TNC, COLK_A = Cfg::COLK_A, COLK_B = Cfg::COLK_B, NT = Cfg::NT;
    constexpr int OPER_A = Cfg::OPER_A, OPER_B = Cfg::OPER_B, EPT_A = Cfg::EPT_A, EPT_B = Cfg::EPT_B;
    constexpr int NQ_A = EPT_A / 2, NQ_B = EPT_B / 2, TPR_A = GEMM_BK / EPT_A, TPR_B = GEMM_BK / EPT_B;
    constexpr int TPK = NT / GEMM_BK;
    constexpr int TILE_ELEMS = TMR * TNC;
    extern __shared__ __attribute__((aligned(16))) double gemm_lds[];
    const int tid = threadIdx.x;

    const int S = gridDim.x;
    int sp = blockIdx.x;                                   // span number: an XCD (blockIdx % 8) takes a contiguous run of spans
    if (S >= 64 && (S & 7) == 0) sp = (sp & 7) * (S >> 3) + (sp >> 3);
    sp = SK_U(sp);
    int rowcost = 0, TOT;                                  // the launcher keeps the iteration count below 2^31
    if (g.mode == GM_KHI_J || g.mode == GM_KLO_J) {
        for (int tj = 0; tj < g.nt; ++tj) rowcost += sk_col_kt<TMR, TNC>(g, tj);
        TOT = g.mt * rowcost;
    } else {
        TOT = (g.mode == GM_LOWER ? g.mt * (g.mt + 1) / 2 : g.mt * g.nt) * (g.K / GEMM_BK);
    }
    const int span_q = SK_U(TOT / S), span_r = TOT - span_q * S;
    auto span_start = [&](int s) { return SK_U(span_q * s + span_r * s / S); };       // floor(TOT s / S) without 64-bit arithmetic (span_r s < S^2)
    int it = span_start(sp);
    const int it1 = span_start(sp + 1);

    while (it < it1) {
        const SkTile t = sk_locate<TMR, TNC>(g, it, rowcost);
        const int kb0 = it - t.start;
        const int kb1 = min(t.kt, kb0 + (it1 - it));
        it += kb1 - kb0;
        if (t.skip) continue;
        const bool partial = kb0 > 0;                       // the tile began in an earlier span: this one contributes a slot
        const bool owner = !partial && kb1 < t.kt;          // the tile ends in a later span: wait for the slots, then write C
        const int kt = kb1 - kb0;
        const int64_t k0 = (int64_t)(t.kbase + kb0) * GEMM_BK;
        const double* Ap = g.A + (AKM ? k0 * g.lda + (int64_t)t.ti * TMR : (int64_t)t.ti * TMR * g.lda + k0);
        const double* Bp = g.B + (BKM ? k0 * g.ldb + (int64_t)t.tj * TNC : (int64_t)t.tj * TNC * g.ldb + k0);
        double* Cp = g.C + (int64_t)t.ti * TMR * g.ldc + (int64_t)t.tj * TNC;
        const bool fresh = g.beta0_from > 0 && t.ti >= g.beta0_from - 1;
        // per-thread offsets are derived from an opaque copy of the thread id INSIDE the loop: hoisted out of it (32 64-bit C offsets among
        // them) they would not fit next to the accumulators and spill
        int tl = tid;
        asm volatile("" : "+v"(tl));
        const int ln = tl & 63, wv = tl >> 6, wi = wv / NWJ, wj = wv % NWJ;
        const int64_t a_g = AKM ? (int64_t)(tl / TPK) * g.lda + (tl % TPK) * EPT_A : (int64_t)(tl / TPR_A) * g.lda + (tl % TPR_A) * EPT_A;
        const int64_t b_g = BKM ? (int64_t)(tl / TPK) * g.ldb + (tl % TPK) * EPT_B : (int64_t)(tl / TPR_B) * g.ldb + (tl % TPR_B) * EPT_B;
        const int a_l = AKM ? (tl / TPK) * COLK_A + (tl % TPK) * EPT_A : (tl / TPR_A) * LDS_ROWK + (tl % TPR_A) * EPT_A;
        const int b_l = BKM ? (tl / TPK) * COLK_B + (tl % TPK) * EPT_B : (tl / TPR_B) * LDS_ROWK + (tl % TPR_B) * EPT_B;
        const int64_t a_step = AKM ? (int64_t)GEMM_BK * g.lda : GEMM_BK;
        const int64_t b_step = BKM ? (int64_t)GEMM_BK * g.ldb : GEMM_BK;
        const int crow = wi * (TMR / NWI) + (ln >> 4), ccol = wj * (TNC / NWJ) + (ln & 15);
        const int fa = AKM ? (ln >> 4) * COLK_A + wi * (TMR / NWI) + (ln & 15) : (wi * (TMR / NWI) + (ln & 15)) * LDS_ROWK + (ln >> 4);
        const int fb = BKM ? (ln >> 4) * COLK_B + wj * (TNC / NWJ) + (ln & 15) : (wj * (TNC / NWJ) + (ln & 15)) * LDS_ROWK + (ln >> 4);
        constexpr int fa_m = AKM ? 16 : 16 * LDS_ROWK, fa_k = AKM ? 4 * COLK_A : 4;
        constexpr int fb_n = BKM ? 16 : 16 * LDS_ROWK, fb_k = BKM ? 4 * COLK_B : 4;


        d4_t acc[WTM][WTN];
        if (!partial && g.beta != 0.0 && !fresh) {
            const double sc = g.beta / g.alpha;
#pragma unroll
            for (int m = 0; m < WTM; ++m)
#pragma unroll
                for (int n = 0; n < WTN; ++n)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[m][n][r] = sc * Cp[(int64_t)(crow + m * 16 + 4 * r) * g.ldc + ccol + n * 16];
        } else {
#pragma unroll
            for (int m = 0; m < WTM; ++m)
#pragma unroll
                for (int n = 0; n < WTN; ++n) acc[m][n] = (d4_t){0.0, 0.0, 0.0, 0.0};
        }
        d2_t ra[NQ_A], rb[NQ_B];
        auto load_block = [&](int kb) {
            const d2_t* pa = reinterpret_cast<const d2_t*>(Ap + a_g + (int64_t)kb * a_step);
            const d2_t* pb = reinterpret_cast<const d2_t*>(Bp + b_g + (int64_t)kb * b_step);
#pragma unroll
            for (int q = 0; q < NQ_A; ++q) ra[q] = pa[q];
#pragma unroll
            for (int q = 0; q < NQ_B; ++q) rb[q] = pb[q];
        };
        auto write_block = [&](int buf) {
            double* sa = gemm_lds + buf * (OPER_A + OPER_B);
            double* sb = sa + OPER_A;
#pragma unroll
            for (int q = 0; q < NQ_A; ++q) *reinterpret_cast<d2_t*>(sa + a_l + 2 * q) = ra[q];
#pragma unroll
            for (int q = 0; q < NQ_B; ++q) *reinterpret_cast<d2_t*>(sb + b_l + 2 * q) = rb[q];
        };
        auto read_frag = [&](double (&av)[WTM], double (&bv)[WTN], int buf, int k4) {
            const double* sa = gemm_lds + buf * (OPER_A + OPER_B);
            const double* sb = sa + OPER_A;
#pragma unroll
            for (int m = 0; m < WTM; ++m) av[m] = sa[fa + m * fa_m + k4 * fa_k];
#pragma unroll
            for (int n = 0; n < WTN; ++n) bv[n] = sb[fb + n * fb_n + k4 * fb_k];
        };
        auto mma = [&](const double (&av)[WTM], const double (&bv)[WTN]) {
#pragma unroll
            for (int m = 0; m < WTM; ++m)
#pragma unroll
                for (int n = 0; n < WTN; ++n)
                    acc[m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[m], bv[n], acc[m][n], 0, 0, 0);
        };
#define GEMM_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
        __syncthreads();                                   // the previous segment's last fragment reads are done
        load_block(0);
        write_block(0);
        load_block(min(1, kt - 1));
        GEMM_LDS_BARRIER();
        {
            double a0[WTM], b0[WTN], a1[WTM], b1[WTN];
            read_frag(a0, b0, 0, 0);
            for (int kb = 0; kb < kt; ++kb) {               // the pipeline of k_gemm
                const int buf = kb & 1;
                read_frag(a1, b1, buf, 1);
                __builtin_amdgcn_sched_barrier(0);
                mma(a0, b0);
                __builtin_amdgcn_sched_barrier(0);
                read_frag(a0, b0, buf, 2);
                __builtin_amdgcn_sched_barrier(0);
                mma(a1, b1);
                __builtin_amdgcn_sched_barrier(0);
                write_block(buf ^ 1);
                load_block(min(kb + 2, kt - 1));
                read_frag(a1, b1, buf, 3);
                __builtin_amdgcn_sched_barrier(0);
                mma(a0, b0);
                __builtin_amdgcn_sched_barrier(0);
                GEMM_LDS_BARRIER();
                read_frag(a0, b0, buf ^ 1, 0);
                __builtin_amdgcn_sched_barrier(0);
                mma(a1, b1);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#undef GEMM_LDS_BARRIER
#pragma unroll
        for (int m = 0; m < WTM; ++m)
#pragma unroll
            for (int n = 0; n < WTN; ++n) acc[m][n] *= g.alpha;
        if (partial) {
            // slot sp: [register][thread] -- every store a contiguous 4 KB row; write-through, drained, then ONE flag store
            double* slot = g.sk_ws + (size_t)sp * TILE_ELEMS;
#pragma unroll
            for (int m = 0; m < WTM; ++m)
#pragma unroll
                for (int n = 0; n < WTN; ++n)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        __hip_atomic_store(slot + (size_t)((m * WTN + n) * 4 + r) * NT + tl, acc[m][n][r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) __hip_atomic_store(g.sk_flags + sp, g.sk_epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            continue;
        }
        if (owner) {
            const int tile_end = t.start + t.kt;
            int last = sp + 1;
            while (last + 1 < S && span_start(last + 1) < tile_end) ++last;      // spans sp + 1 .. last hold the rest of this tile
            if (tid == 0) {
                for (int q = sp + 1; q <= last; ++q) {
                    unsigned spins = 0;
                    while (__hip_atomic_load(g.sk_flags + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != g.sk_epoch) {
                        __builtin_amdgcn_s_sleep(8);
                        if (++spins > (1u << 24)) {           // seconds: the span never ran.  Reported through the pivot word (the evaluation fails loudly)
                            if (g.sk_info) atomicMin(g.sk_info, (unsigned long long)MOGP_INFO_CHAIN_TIMEOUT);
                            break;
                        }
                    }
                }
            }
            __syncthreads();
            for (int q = sp + 1; q <= last; ++q) {
                const double* slot = g.sk_ws + (size_t)q * TILE_ELEMS;
#pragma unroll
                for (int m = 0; m < WTM; ++m)
#pragma unroll
                    for (int n = 0; n < WTN; ++n)
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            acc[m][n][r] += __hip_atomic_load(slot + (size_t)((m * WTN + n) * 4 + r) * NT + tl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
#pragma unroll
        for (int m = 0; m < WTM; ++m)
#pragma unroll
            for (int n = 0; n < WTN; ++n)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    Cp[(int64_t)(crow + m * 16 + 4 * r) * g.ldc + ccol + n * 16] = acc[m][n][r];
    }
}

template <int AKM, int BKM>
static int launch_gemm_sk_t(const GemmArgs& a, hipStream_t s) {
    constexpr int lds_bytes = GemmCfg<4, 2, 4, 2>::LDS_BYTES;
    static std::atomic<unsigned long long> attr_done{0ull};                 // one bit per device
    { int r__ = set_max_dynamic_lds(reinterpret_cast<const void*>(k_gemm_sk<AKM, BKM>), lds_bytes, attr_done); if (r__) return r__; }
    hipLaunchKernelGGL((k_gemm_sk<AKM, BKM>), dim3(a.sk_spans), dim3(512), lds_bytes, s, a);
    HIP_TRY(hipGetLastError());
    return 0;
}

template <int AKM, int BKM, int WTM, int WTN, int NWJ = 2, int NWI = 2, bool FLOWH = false>
static int launch_gemm_t(const GemmArgs& a, int grid, hipStream_t s) {
    constexpr int lds_bytes = GemmCfg<WTM, WTN, NWJ, NWI>::LDS_BYTES;
    static std::atomic<unsigned long long> attr_done{0ull};                 // one bit per device
    { int r__ = set_max_dynamic_lds(reinterpret_cast<const void*>(k_gemm<AKM, BKM, WTM, WTN, NWJ, NWI, FLOWH>), lds_bytes, attr_done); if (r__) return r__; }
    hipLaunchKernelGGL((k_gemm<AKM, BKM, WTM, WTN, NWJ, NWI, FLOWH>), dim3(grid), dim3(64 * NWI * NWJ), lds_bytes, s, a);
    HIP_TRY(hipGetLastError());
    return 0;
}

int launch_gemm(const GemmArgs& a0, hipStream_t s) {
    GemmArgs a = a0;
    // MOGP_FAKE_K=d (measurement only, WRONG results): the 128 x 128-tile launches contract 1/d of their k range -- is an evaluation bound
    // by what the GEMM streams deliver (time falls with d) or by its dependency chain (it does not)?
    static const int fake_k = std::getenv("MOGP_FAKE_K") ? std::atoi(std::getenv("MOGP_FAKE_K")) : 0;
    if (fake_k > 1 && !a.small && a.mode != GM_TASKS && a.K >= 512) a.K = (a.K / fake_k) / GEMM_BK * GEMM_BK;
    int grid;
    switch (a.mode) {
        case GM_LOWER: case GM_LAUUM: grid = a.mt * (a.mt + 1) / 2; break;
        case GM_TASKS: grid = a.ntasks; break;
        default: grid = a.mt * a.nt; break;
    }
    if (grid <= 0) return 0;
    if (a.ksplit > 1) grid *= a.ksplit;
    const int v = (a.a_kmajor ? 2 : 0) | (a.b_kmajor ? 1 : 0);
    if (a.small) {
        const bool rect = a.mode == GM_RECT || a.mode == GM_RECT_LOWER;
        // (eight waves on the 64-row tiles as well: no difference -- 12.97 vs 12.97 ms at configs[1], 47.4 vs 47.3 at configs[3])
        if (a.small == 1 && v == 1 && (rect || a.mode == GM_KLO_J || a.mode == GM_KHI_I)) return launch_gemm_t<0, 1, 2, 4>(a, grid, s);
        if (a.fl_flags) {               // the two products between chain kernels of the dataflow schedule (flow.hip): same tiles, hand-off hooks compiled in
            if (a.small == 1 && v == 0 && a.mode == GM_KHI_J) return launch_gemm_t<0, 0, 2, 4, 2, 2, true>(a, grid, s);
            if (a.small == 2 && v == 0 && a.mode == GM_RECT_LOWER) return launch_gemm_t<0, 0, 2, 2, 2, 2, true>(a, grid, s);
            set_error("launch_gemm: the dataflow hooks exist for the mini-panel and the next-diagonal update only");
            return -1;
        }
        if (a.small == 1 && v == 0 && a.mode == GM_KHI_J) return launch_gemm_t<0, 0, 2, 4>(a, grid, s);
        if (v != 0 || !rect) {
            set_error("launch_gemm: small-tile variants are built for A k-contiguous and rectangular grids only");
            return -1;
        }
        return a.small == 1 ? launch_gemm_t<0, 0, 2, 4>(a, grid, s) : launch_gemm_t<0, 0, 2, 2>(a, grid, s);
    }
    // The 128 x 128 tile runs on EIGHT waves (4 x 2 MFMA tiles each, 512 threads, 128 VGPRs): two workgroups per CU as before, but four
    // waves per SIMD instead of two to pick MFMAs from while one waits for LDS, the barrier or its C tile.  Same arithmetic in the same
    // order (results bit-identical to the four-wave kernel); measured round 3 on one box: rank-512 lower-triangle update 57.3 -> 58.4
    // TFLOP/s (k-contiguous), 54.7 -> 58.9 (k-major); configs[1] 13.46 -> 12.96 ms, configs[2] 555 -> 546, configs[3] 48.8 -> 47.4.
    // MOGP_GEMM8=0: the four-wave kernel (4 x 4 MFMA tiles a wave, 224 VGPRs).
    static const bool eight = !(std::getenv("MOGP_GEMM8") && std::atoi(std::getenv("MOGP_GEMM8")) == 0);
    if (a.sk_spans > 0 && eight && a.ksplit <= 1 && a.row_mod <= 1 && a.sk_ws && a.sk_flags &&
        (a.mode == GM_RECT || a.mode == GM_RECT_LOWER || a.mode == GM_LOWER || a.mode == GM_KHI_J || a.mode == GM_KLO_J)) {
        switch (v) {
            case 0: return launch_gemm_sk_t<0, 0>(a, s);
            case 1: return launch_gemm_sk_t<0, 1>(a, s);
            case 2: return launch_gemm_sk_t<1, 0>(a, s);
            default: return launch_gemm_sk_t<1, 1>(a, s);
        }
    }
    // (Sixteen waves of 2 x 2 tiles -- eight per SIMD, 64 VGPRs -- spill and read LDS twice as often: 55-56 TFLOP/s, slower than either.)
    if (eight) {
        switch (v) {
            case 0: return launch_gemm_t<0, 0, 4, 2, 4>(a, grid, s);
            case 1: return launch_gemm_t<0, 1, 4, 2, 4>(a, grid, s);
            case 2: return launch_gemm_t<1, 0, 4, 2, 4>(a, grid, s);
            default: return launch_gemm_t<1, 1, 4, 2, 4>(a, grid, s);
        }
    }
    switch (v) {
        case 0: return launch_gemm_t<0, 0, 4, 4>(a, grid, s);
        case 1: return launch_gemm_t<0, 1, 4, 4>(a, grid, s);
        case 2: return launch_gemm_t<1, 0, 4, 4>(a, grid, s);
        default: return launch_gemm_t<1, 1, 4, 4>(a, grid, s);
    }
}

double gemm_flops(const GemmArgs& a, const std::vector<GemmTask>* host_tasks) {
    const int TM = a.small ? 64 : MOGP_TILE;                 // tile rows
    const int TN = a.small == 2 ? 64 : MOGP_TILE;            // tile columns
    const double tile = 2.0 * TM * TN;
    double k = 0.0;
    switch (a.mode) {
        case GM_RECT: k = (double)a.mt * a.nt * a.K; break;
        case GM_RECT_LOWER: for (int tj = 0; tj < a.nt; ++tj) for (int ti = 0; ti < a.mt; ++ti) if ((ti + 1) * TM > tj * TN) k += a.K; break;
        case GM_LOWER: k = (double)a.mt * (a.mt + 1) / 2 * a.K; break;
        case GM_LAUUM: for (int ti = 0; ti < a.mt; ++ti) k += (double)(ti + 1) * (a.K - ti * TM); break;
        case GM_KHI_I: for (int ti = 0; ti < a.mt; ++ti) k += (double)a.nt * ((ti + 1) * TM < a.K ? (ti + 1) * TM : a.K); break;
        case GM_KLO_J: for (int tj = 0; tj < a.nt; ++tj) k += (double)a.mt * (a.K - tj * TN); break;
        case GM_KLO_I: for (int ti = 0; ti < a.mt; ++ti) k += (double)a.nt * (a.K - ti * TM); break;
        case GM_KHI_J: for (int tj = 0; tj < a.nt; ++tj) k += (double)a.mt * ((tj + 1) * TN < a.K ? (tj + 1) * TN : a.K); break;
        default: if (host_tasks) for (const auto& t : *host_tasks) k += (double)t.kt * GEMM_BK; break;
    }
    return tile * k;
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
                                                    double* __restrict__ z, double* __restrict__ zz_partial, int row_mod, int row_rem) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t i = (int64_t)blockIdx.x * 4 + wave;
    __shared__ double zz[4];
    double s = 0.0;
    const bool mine = row_mod <= 1 || (int)((i / MOGP_TILE) % row_mod) == row_rem;
    if (i < n && !mine && lane == 0) z[i] = 0.0;
    if (i < n && mine) {
        const double* row = W + i * ld;
        for (int64_t k = lane; k <= i; k += 64) s = fma(row[k], y[k], s);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
        if (lane == 0) z[i] = s;
    }
    if (lane == 0) zz[wave] = (i < n && mine) ? s * s : 0.0;
    __syncthreads();
    if (threadIdx.x == 0) zz_partial[blockIdx.x] = (zz[0] + zz[1]) + (zz[2] + zz[3]);
}
int launch_trmv_lower(const double* W, int64_t ld, int64_t n, const double* y, double* z, double* zz_partial, hipStream_t s,
                      int row_mod, int row_rem) {
    hipLaunchKernelGGL(k_trmv_lower, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, s, W, ld, n, y, z, zz_partial, row_mod, row_rem);
    HIP_TRY(hipGetLastError());
    return 0;
}

// a_j = sum_{i>=j} W[i][j] z_i.  grid (column blocks of 64, row chunks of 512); partial[chunk][j]; fixed-order second pass.
__global__ __launch_bounds__(256) void k_trmv_lower_t_part(const double* __restrict__ W, int64_t ld, int64_t n,
                                                           const double* __restrict__ z, double* __restrict__ part, int row_mod, int row_rem) {
    const int64_t j = (int64_t)blockIdx.x * 64 + (threadIdx.x & 63);
    const int sub = threadIdx.x >> 6;
    const int64_t r0 = (int64_t)blockIdx.y * 512, r1 = min(n, r0 + 512);
    __shared__ double red[4][64];
    double s = 0.0;
    if (j < n) {
        const int64_t start = max(r0, j);
        for (int64_t i = start + sub; i < r1; i += 4)
            if (row_mod <= 1 || (int)((i / MOGP_TILE) % row_mod) == row_rem) s = fma(W[i * ld + j], z[i], s);
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
int launch_trmv_lower_t(const double* W, int64_t ld, int64_t n, const double* z, double* a, hipStream_t s, int row_mod, int row_rem) {
    // partial buffer lives right behind z's vector block: the caller passes `a` with room for (1 + nchunks) * n doubles
    const int nchunks = (int)((n + 511) / 512);
    double* part = a + n;
    hipLaunchKernelGGL(k_trmv_lower_t_part, dim3((unsigned)((n + 63) / 64), (unsigned)nchunks), dim3(256), 0, s, W, ld, n, z, part, row_mod, row_rem);
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

__global__ void k_symv_combine(int64_t n, const double* a, const double* b, const double* A, int64_t ld, const double* y, double sign, double* out,
                               int row_mod, int row_rem) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const bool mine = row_mod <= 1 || (int)((i / MOGP_TILE) % row_mod) == row_rem;      // the diagonal was counted twice only on its owner
    out[i] = sign * (a[i] + b[i] - (mine ? A[i * ld + i] * y[i] : 0.0));
}
int launch_symv_lower(const double* A, int64_t ld, int64_t n, const double* y, double* out, double* scratch, double sign, hipStream_t s,
                      int row_mod, int row_rem) {
    double* z1 = scratch;                 // tril(A) y            [n] (+ n/4 partials behind it, unused here)
    double* zz = scratch + n;             // [n/4 + 1]
    double* z2 = scratch + 2 * n;         // tril(A)^T y          [n] + row-chunk partials
    int rc;
    if ((rc = launch_trmv_lower(A, ld, n, y, z1, zz, s, row_mod, row_rem))) return rc;
    if ((rc = launch_trmv_lower_t(A, ld, n, y, z2, s, row_mod, row_rem))) return rc;
    hipLaunchKernelGGL(k_symv_combine, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, n, z1, z2, A, ld, y, sign, out, row_mod, row_rem);
    HIP_TRY(hipGetLastError());
    return 0;
}

__global__ void k_copy2d(double* dst, int64_t ldd, const double* src, int64_t lds, int64_t cols, double scale) {
    const int64_t r = blockIdx.x;
    for (int64_t c = threadIdx.x; c < cols; c += blockDim.x) dst[r * ldd + c] = scale * src[r * lds + c];
}
int launch_copy2d(double* dst, int64_t ldd, const double* src, int64_t lds, int64_t rows, int64_t cols, double scale, hipStream_t s) {
    if (rows <= 0 || cols <= 0) return 0;
    hipLaunchKernelGGL(k_copy2d, dim3((unsigned)rows), dim3(256), 0, s, dst, ldd, src, lds, cols, scale);
    HIP_TRY(hipGetLastError());
    return 0;
}

__global__ void k_add_diag(double* A, int64_t ld, int64_t n, double val) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) A[i * ld + i] += val;
}
int launch_add_diag(double* A, int64_t ld, int64_t n, double val, hipStream_t s) {
    hipLaunchKernelGGL(k_add_diag, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, A, ld, n, val);
    HIP_TRY(hipGetLastError());
    return 0;
}

// upper <- lower^T, 64x64 tiles through LDS (one workgroup per strictly-lower or diagonal tile)
__global__ __launch_bounds__(256) void k_symmetrize(double* A, int64_t ld) {
    __shared__ double t[64][65];
    const int ti = tri_row(blockIdx.x), tj = blockIdx.x - ti * (ti + 1) / 2;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int r = ty; r < 64; r += 4) t[r][tx] = A[(int64_t)(ti * 64 + r) * ld + tj * 64 + tx];
    __syncthreads();
    for (int r = ty; r < 64; r += 4) {
        const int64_t gr = tj * 64 + r, gc = ti * 64 + tx;
        if (gc > gr) A[gr * ld + gc] = t[tx][r];
    }
}
int launch_symmetrize(double* A, int64_t ld, int64_t n, hipStream_t s) {
    const int nt = (int)(n / 64);
    hipLaunchKernelGGL(k_symmetrize, dim3(nt * (nt + 1) / 2), dim3(256), 0, s, A, ld);
    HIP_TRY(hipGetLastError());
    return 0;
}

__global__ void k_combine(double* out, const double* P, const double* Q, int64_t ld, int64_t n, double ca, double cp, double cq) {
    const int64_t r = blockIdx.x;
    for (int64_t c = threadIdx.x; c < n; c += blockDim.x) {
        double v = (r == c ? ca : 0.0) - cp * P[r * ld + c];
        if (Q) v -= cq * Q[r * ld + c];
        out[r * ld + c] = v;
    }
}
int launch_combine(double* out, const double* P, const double* Q, int64_t ld, int64_t n, double ca, double cp, double cq, hipStream_t s) {
    hipLaunchKernelGGL(k_combine, dim3((unsigned)n), dim3(256), 0, s, out, P, Q, ld, n, ca, cp, cq);
    HIP_TRY(hipGetLastError());
    return 0;
}

// column reductions of a dense rows x n matrix: partial over row chunks of 256, then a fixed-order sum
__global__ __launch_bounds__(256) void k_gemv_cols_part(const double* __restrict__ M, int64_t ld, int64_t rows, int64_t n,
                                                        const double* __restrict__ v, double* __restrict__ part) {
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= n) return;
    const int64_t r0 = (int64_t)blockIdx.y * 256, r1 = min(rows, r0 + 256);
    double s = 0.0;
    if (v) { for (int64_t i = r0; i < r1; ++i) s = fma(M[i * ld + j], v[i], s); }
    else   { for (int64_t i = r0; i < r1; ++i) { const double x = M[i * ld + j]; s = fma(x, x, s); } }
    part[(int64_t)blockIdx.y * n + j] = s;
}
__global__ void k_sum_parts(const double* __restrict__ part, int64_t n, int nparts, double* __restrict__ out) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    double s = 0.0;
    for (int c = 0; c < nparts; ++c) s += part[(int64_t)c * n + j];
    out[j] = s;
}
int launch_gemv_cols(const double* M, int64_t ld, int64_t rows, int64_t n, const double* v, double* out, double* scratch, hipStream_t s) {
    const int nparts = (int)((rows + 255) / 256);       // scratch: nparts * n doubles
    hipLaunchKernelGGL(k_gemv_cols_part, dim3((unsigned)((n + 255) / 256), (unsigned)nparts), dim3(256), 0, s, M, ld, rows, n, v, scratch);
    hipLaunchKernelGGL(k_sum_parts, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, scratch, n, nparts, out);
    HIP_TRY(hipGetLastError());
    return 0;
}

__global__ void k_get_diag(const double* A, int64_t ld, int64_t n, double* out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = A[i * ld + i];
}
int launch_get_diag(const double* A, int64_t ld, int64_t n, double* out, hipStream_t s) {
    hipLaunchKernelGGL(k_get_diag, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, A, ld, n, out);
    HIP_TRY(hipGetLastError());
    return 0;
}
__global__ void k_axpby(int64_t n, double a, const double* x, double b, const double* y, double* out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = a * x[i] + b * y[i];
}
int launch_axpby(int64_t n, double a, const double* x, double b, const double* y, double* out, hipStream_t s) {
    hipLaunchKernelGGL(k_axpby, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, n, a, x, b, y, out);
    HIP_TRY(hipGetLastError());
    return 0;
}

// out[i] = sum_k slices[k * stride + i]   (split-K partial results, fixed order)
__global__ void k_sum_slices(const double* __restrict__ slices, int64_t n, int ks, double* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double s = 0.0;
    for (int k = 0; k < ks; ++k) s += slices[(int64_t)k * n + i];
    out[i] = s;
}
int launch_sum_slices(const double* slices, int64_t n, int ks, double* out, hipStream_t s) {
    hipLaunchKernelGGL(k_sum_slices, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, slices, n, ks, out);
    HIP_TRY(hipGetLastError());
    return 0;
}

// between two factorisations of one evaluation: the pivot word back to "no failure" -- unless it holds a time-out, which must survive until the
// host looks (a stream-K hand-off of a triangular solve that timed out used to be overwritten here: ADVICE round 3)
__global__ void k_info_rearm(unsigned long long* info) {
    if (*info != MOGP_INFO_CHAIN_TIMEOUT) *info = ~0ull;
}
// the same, keeping what the first factorisation reported in info[1]: the host then looks at both words ONCE, at the synchronisation it needs anyway,
// instead of stopping the device behind every factorisation (two host round trips of ~250 us per evaluation of the sparse bound)
__global__ void k_info_stash(unsigned long long* info) {
    info[1] = info[0];
    if (info[0] != MOGP_INFO_CHAIN_TIMEOUT) info[0] = ~0ull;
}
int launch_info_stash(unsigned long long* info, hipStream_t s) {
    hipLaunchKernelGGL(k_info_stash, dim3(1), dim3(1), 0, s, info);
    HIP_TRY(hipGetLastError());
    return 0;
}
int launch_info_rearm(unsigned long long* info, hipStream_t s) {
    hipLaunchKernelGGL(k_info_rearm, dim3(1), dim3(1), 0, s, info);
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
// smallest and largest diagonal entry of the Cholesky factor, read off the tile inverses (diag W_kk = 1 / diag L_kk): out[0] = min, out[1] = max over the
// first n rows (the padding rows carry ones).  One workgroup; the factorisation's cheap condition signal (mogp_model_pivot_range).
__global__ __launch_bounds__(1024) void k_pivot_range(const double* __restrict__ invd, int64_t n, double* __restrict__ out) {
    __shared__ double smin[16], smax[16];
    double lo = 1e300, hi = 0.0;
    for (int64_t r0 = threadIdx.x; r0 < n; r0 += 8 * 1024) {            // eight independent loads in flight per thread
        double w[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int64_t r = r0 + (int64_t)u * 1024;
            w[u] = r < n ? fabs(invd[(r >> 7) * (int64_t)(MOGP_TILE * MOGP_TILE) + (r & 127) * (MOGP_TILE + 1)]) : -1.0;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (w[u] < 0.0) continue;
            const double l = w[u] > 0.0 ? 1.0 / w[u] : 1e300;
            lo = fmin(lo, l); hi = fmax(hi, l);
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { lo = fmin(lo, __shfl_xor(lo, off, 64)); hi = fmax(hi, __shfl_xor(hi, off, 64)); }
    if ((threadIdx.x & 63) == 0) { smin[threadIdx.x >> 6] = lo; smax[threadIdx.x >> 6] = hi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < 16; ++i) { lo = fmin(lo, smin[i]); hi = fmax(hi, smax[i]); }
        out[0] = lo; out[1] = hi;
    }
}
int launch_pivot_range(const double* invd, int64_t n, double* out, hipStream_t s) {
    hipLaunchKernelGGL(k_pivot_range, dim3(1), dim3(1024), 0, s, invd, n, out);
    HIP_TRY(hipGetLastError());
    return 0;
}

int launch_nonfinite_scan(const double* A, int64_t ld, int64_t n, int* flag, hipStream_t s) {
    hipLaunchKernelGGL(k_nonfinite_scan, dim3((unsigned)n), dim3(256), 0, s, A, ld, n, flag);
    HIP_TRY(hipGetLastError());
    return 0;
}

}  // namespace mogp
