// flow.hip -- the fused factorisation + inversion of potri.hip as TILE DATAFLOW: one resident kernel instead of five streams of launches.
//
// Why (round 3's evidence): with the inverse streamed behind the Cholesky chain as rank-512 launches on five streams an
// N = 8192 evaluation takes 13 ms although its dependency chain alone takes 7.7 ms and its GEMM work 8.5-9.3 ms at lone-launch rates: tall
// launches on the block cycle and bulk launches steal each other's workgroup slots, and every cross-stream event waits for slots that free
// in bursts.  Fifteen stream-level variants did not move it.  Here the SAME tile products (same operands, same k order: results identical
// per tile) are tasks of a static graph:
//   * k_flow: 2 workgroups per CU on the CUs the chain does not own, resident for the whole evaluation.  A workgroup looks at the HEAD of
//     every task queue (queues in priority order: what the next chain kernel needs first, the inverse's accumulations last), takes the
//     first head whose dependency counters are satisfied (compare-and-swap on the queue's head word), runs the 128 x 128 tile with the
//     k loop of linalg.hip:k_gemm (eight waves, BK = 16, double-buffered LDS), stores C write-through, bumps the task's counters.
//   * the chain kernels (chain.hip) stay the producers of L_KK / W_KK on the reserved CUs, one launch per outer block on the private
//     stream; they now wait for "my diagonal block has all its updates" and report "W_KK is there" through the same counters.
// Hand-off = the guide's recipe (MI355X_MICROARCH "inter-workgroup visibility"): payload with sc1 stores, every storing wave drains vmcnt,
// ONE lane bumps an agent-scope counter; the consumer polls relaxed, ONE agent acquire after the match, barrier, plain loads.
// A workgroup never holds a task that is not ready, so any number of resident workgroups makes progress; every queue is sorted by the
// sequential algorithm's order (block, phase), so the globally first unfinished task is always at the head of its queue with its
// dependencies met: no deadlock (tests/test_flow_plan_cpu.py replays the graph on numpy tiles in random orders and checks both).
// Every wait is bounded; a time-out is reported through the pivot word like the chain kernel's and the evaluation is repeated on the
// stream schedule.
// Buffers (all Npad x Npad, leading dimension Npad): A = Schur data (in place), L = panels L[>K, K] at their natural position (no rotating
// buffers: nothing is ever overwritten while it can still be read), Wt = the running product of the elementary block-column inverses,
// Wm = W = L^-1 (final row blocks; the chain kernels write W_KK straight into its diagonal blocks), B = Kj^-1 accumulator.
// Replaces torch.linalg.cholesky + the O(N^3) solves of its autograd backward (reference gpr/model.py:242-246, :291).
#include "mogp_model.h"
#include <unistd.h>
#include <chrono>

#include <cstdio>
#include <cstdlib>

namespace mogp {

typedef double d4_t __attribute__((ext_vector_type(4)));
typedef double d2_t __attribute__((ext_vector_type(2)));

// ---- the tile product: linalg.hip's eight-wave 128 x 128 tile (2 x 4 waves of 4 x 2 MFMA tiles), lda = ldb = ldc = ld ----------------
#define FL_BK 16
#define FL_ROWK 18
#define FL_WTM 4
#define FL_WTN 2
#define FL_NWJ 4
#define FL_NWI 2
#define FL_NT 512
#define FL_COLK (MOGP_TILE + 16)
#define FL_OPER (MOGP_TILE * FL_ROWK)            // == 16 * FL_COLK
#define FL_LDS_DOUBLES (2 * 2 * FL_OPER)
#define FL_LDS_BYTES (FL_LDS_DOUBLES * 8 + 16 + 3 * 64 * 4 + 16)   // + the workgroup's pick words + what wave 0's lanes remember between looks (k_flow: lst)
static_assert(MOGP_TILE * FL_ROWK == 16 * FL_COLK, "one LDS operand slot serves both layouts");

#define FL_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

// C = alpha * (sum_k A B) + (fresh ? 0 : C); alpha = +-1.  Stores are write-through (sc1): another workgroup of this launch reads them.
template <int AKM, int BKM, bool MARK = false>
__device__ __forceinline__ void flow_tile(const double* Ap, const double* Bp, double* Cp, const int64_t ld, const int kt, const bool fresh,
                                          const double alpha, double* gemm_lds, unsigned long long* tr, unsigned* prog, double* const* Cpp, const double* alphap) {
    constexpr int WTM = FL_WTM, WTN = FL_WTN, NWJ = FL_NWJ, NWI = FL_NWI, NT = FL_NT;
    constexpr int TMR = MOGP_TILE, TNC = MOGP_TILE, COLK_A = FL_COLK, COLK_B = FL_COLK, OPER_A = FL_OPER, OPER_B = FL_OPER;
    constexpr int EPT_A = TMR * FL_BK / NT, EPT_B = TNC * FL_BK / NT;
    constexpr int NQ_A = EPT_A / 2, NQ_B = EPT_B / 2, TPR_A = FL_BK / EPT_A, TPR_B = FL_BK / EPT_B, TPK = NT / FL_BK;
    // per-thread offsets from an opaque copy of the thread id (hoisted out of the task loop they would not fit next to the accumulators)
    int tl = threadIdx.x;
    asm volatile("" : "+v"(tl));
    const int ln = tl & 63, wv = tl >> 6, wi = wv / NWJ, wj = wv % NWJ;
    const int64_t a_g = AKM ? (int64_t)(tl / TPK) * ld + (tl % TPK) * EPT_A : (int64_t)(tl / TPR_A) * ld + (tl % TPR_A) * EPT_A;
    const int64_t b_g = BKM ? (int64_t)(tl / TPK) * ld + (tl % TPK) * EPT_B : (int64_t)(tl / TPR_B) * ld + (tl % TPR_B) * EPT_B;
    const int a_l = AKM ? (tl / TPK) * COLK_A + (tl % TPK) * EPT_A : (tl / TPR_A) * FL_ROWK + (tl % TPR_A) * EPT_A;
    const int b_l = BKM ? (tl / TPK) * COLK_B + (tl % TPK) * EPT_B : (tl / TPR_B) * FL_ROWK + (tl % TPR_B) * EPT_B;
    const int64_t a_step = AKM ? (int64_t)FL_BK * ld : FL_BK;
    const int64_t b_step = BKM ? (int64_t)FL_BK * ld : FL_BK;
    const int crow = wi * (TMR / NWI) + (ln >> 4), ccol = wj * (TNC / NWJ) + (ln & 15);
    const int fa = AKM ? (ln >> 4) * COLK_A + wi * (TMR / NWI) + (ln & 15) : (wi * (TMR / NWI) + (ln & 15)) * FL_ROWK + (ln >> 4);
    const int fb = BKM ? (ln >> 4) * COLK_B + wj * (TNC / NWJ) + (ln & 15) : (wj * (TNC / NWJ) + (ln & 15)) * FL_ROWK + (ln >> 4);
    constexpr int fa_m = AKM ? 16 : 16 * FL_ROWK, fa_k = AKM ? 4 * COLK_A : 4;
    constexpr int fb_n = BKM ? 16 : 16 * FL_ROWK, fb_k = BKM ? 4 * COLK_B : 4;

    d4_t acc[WTM][WTN];
    if (!fresh) {                                        // start from (1 / alpha) C: the epilogue is a pure store of alpha * acc (k_gemm's form)
#pragma unroll
        for (int m = 0; m < WTM; ++m)
#pragma unroll
            for (int n = 0; n < WTN; ++n)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[m][n][r] = alpha * Cp[(int64_t)(crow + m * 16 + 4 * r) * ld + ccol + n * 16];
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
    // (debugging: prog[wave] = how far this wave has come -- 0x10000 entered, 0x20000 + kb: the loads of k block kb + 1 have returned and gone to LDS,
    // 0x30000 past the k loop, 0x40000 stores issued; prog[8 + wave]: the clock of that mark)
#define FL_MARK(v) do { if constexpr (MARK) { if (prog && ln == 0) { const unsigned v__ = (v); prog[wv] = v__; \
        if (wv == 0) { const unsigned kb__ = v__ & 0xffffu, now__ = (unsigned)(wall_clock64() >> 4); \
            if (v__ == 0x10000u) prog[8] = now__; else if (v__ == 0x30000u) prog[13] = now__; else if (v__ == 0x40000u) prog[14] = now__; \
            else if ((v__ >> 16) == 2u && (kb__ & 7u) == 0u) prog[9 + (kb__ >> 3)] = now__; } } } } while (0)
    FL_MARK(0x10000u);
    load_block(0);
    write_block(0);
    load_block(min(1, kt - 1));
    FL_LDS_BARRIER();
    if constexpr (MARK) { if (tr) tr[2] = wall_clock64(); }
    {
        double a0[WTM], b0[WTN], a1[WTM], b1[WTN];
        read_frag(a0, b0, 0, 0);
        for (int kb = 0; kb < kt; ++kb) {               // the pipeline of k_gemm (linalg.hip)
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
            FL_MARK(0x20000u + (unsigned)kb);
            load_block(min(kb + 2, kt - 1));
            read_frag(a1, b1, buf, 3);
            __builtin_amdgcn_sched_barrier(0);
            mma(a0, b0);
            __builtin_amdgcn_sched_barrier(0);
            FL_LDS_BARRIER();
            read_frag(a0, b0, buf ^ 1, 0);
            __builtin_amdgcn_sched_barrier(0);
            mma(a1, b1);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    FL_MARK(0x30000u);
    if constexpr (MARK) { if (tr) tr[3] = wall_clock64(); }
    // the C coordinates again from a fresh opaque copy of the thread id, the C pointer and the sign from an opaque copy of the caller's: what is only
    // needed behind the k loop must not be kept alive across it (it was spilled to scratch)
    int t2 = threadIdx.x;
    asm volatile("" : "+v"(t2));
    const int ln2 = t2 & 63, wv2 = t2 >> 6;
    const int crow2 = (wv2 / NWJ) * (TMR / NWI) + (ln2 >> 4), ccol2 = (wv2 % NWJ) * (TNC / NWJ) + (ln2 & 15);
    double* Cq = *Cpp;
    const double alpha2 = *alphap;
#pragma unroll
    for (int m = 0; m < WTM; ++m)
#pragma unroll
        for (int n = 0; n < WTN; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                __hip_atomic_store(Cq + (int64_t)(crow2 + m * 16 + 4 * r) * ld + ccol2 + n * 16, alpha2 * acc[m][n][r], __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
    FL_MARK(0x40000u);
#undef FL_MARK
}

struct FlowArgs {
    double* bA; double* bL; double* bWt; double* bWm; double* bB;     // buffer ids 0 .. 4
    int64_t ld;
    const FlowTask* tasks;
    const int* qmeta;                 // [2 nq]: first task, number of tasks of every queue
    unsigned* flags;                  // dependency counters, then the queue heads, then the error word (all zero at the start of an evaluation)
    int nq, ncas, base_heads, base_err;
    unsigned nap_max;                 // an idle workgroup sleeps 2^1 .. 2^nap_max microseconds between looks
    unsigned nap_calm;                // ... up to 2^nap_calm once it has been idle for FL_CALM_AFTER looks
    int nhi;                          // queues below this index always hold a task per workgroup (claimed at the next look after one was taken)
    int claim_one;                    // nothing ready: take from ONE queue per look (the highest priority with a free slot) instead of from all
    int refill;                       // a taken eager slot is refilled at once (0: only when the workgroup finds nothing ready)
    const double* vy; double* vz; double* vzz; double* vpart;         // z = W y and alpha = W^T z as tasks (null: those tasks only count)
    int64_t npad;
    unsigned long long* info;
    int post_base;                    // (the second, small instance of the kernel reports behind the first one's workgroups)
    unsigned char* done;              // optional (MOGP_FLOW_DEBUG): one byte per task, set when the task has signalled
    unsigned* post;                   // optional post-mortem (MOGP_FLOW_DEBUG): [workgroups][FLOW_POST_W]: exit code + 4, idle looks, state, last task, 64 x (held ticket + 1)
    unsigned* diag;                   // FLOW_DIAG_WORDS counters that outlive the evaluation (deep looks, what they found)
    unsigned long long* trace;        // optional: [FLOW_TRACE_W ntasks] look, claimed, k loop from, to, signalled (100 MHz wall clock), XCC << 16 | workgroup
};

// what every look and every task set-up reads: by VALUE (kernel-argument registers); the rest of FlowArgs is read through the pointer where it is used
struct FlowHot {
    double* bA; double* bL; double* bWt; double* bWm; double* bB; int64_t ld;
    const FlowTask* tasks; const int* qmeta; unsigned* flags;
    int nq, ncas, base_heads, base_err;
    unsigned nap_max, nap_calm;
};

// ---- z = W y and alpha = W^T z inside the dataflow (they were three memory-bound launches behind the last accumulation: 0.4 ms) ----------
// z rows: tile row i of W is final when its row block's T6 tasks are; one task = the 128 dot products of that tile row (a wave takes 16 rows,
// four at a time, lanes along k with 16-byte loads).  zz[i] = sum of the 128 z^2 (the caller adds the tile rows up).
__device__ __forceinline__ void flow_zrow(const FlowArgs& g, const int i, double* lds) {
    int tid = threadIdx.x;                                // (opaque: nothing of this rare task is computed ahead at kernel entry and kept -- spilled -- for the kernel's life)
    asm volatile("" : "+v"(tid));
    const int lane = tid & 63, wave = tid >> 6;
    const int K = (i + 1) * MOGP_TILE;
    const double* Wr = g.bWm + (int64_t)(i * MOGP_TILE + wave * 16) * g.ld + 2 * lane;
    const double* yp = g.vy + 2 * lane;
    double zsq = 0.0;
    for (int r4 = 0; r4 < 16; r4 += 4) {
        const double* p0 = Wr + (int64_t)r4 * g.ld;
        d2_t a0 = (d2_t){0.0, 0.0}, a1 = a0, a2 = a0, a3 = a0;
#pragma unroll 2
        for (int k = 0; k < K; k += MOGP_TILE) {
            const d2_t yv = *reinterpret_cast<const d2_t*>(yp + k);
            const d2_t w0 = *reinterpret_cast<const d2_t*>(p0 + k);
            const d2_t w1 = *reinterpret_cast<const d2_t*>(p0 + g.ld + k);
            const d2_t w2 = *reinterpret_cast<const d2_t*>(p0 + 2 * g.ld + k);
            const d2_t w3 = *reinterpret_cast<const d2_t*>(p0 + 3 * g.ld + k);
            a0 += w0 * yv; a1 += w1 * yv; a2 += w2 * yv; a3 += w3 * yv;
        }
        double s[4] = {a0[0] + a0[1], a1[0] + a1[1], a2[0] + a2[1], a3[0] + a3[1]};
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) s[u] += __shfl_down(s[u], off, 64);
            if (lane == 0) {
                __hip_atomic_store(g.vz + (int64_t)i * MOGP_TILE + wave * 16 + r4 + u, s[u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                zsq += s[u] * s[u];
            }
        }
    }
    if (lane == 0) lds[wave] = zsq;
    __syncthreads();
    if (tid == 0) {
        double t = 0.0;
        for (int w = 0; w < FL_NT / 64; ++w) t += lds[w];
        __hip_atomic_store(g.vzz + i, t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
// alpha, one row block at a time: part[r][col] = sum over the rows of block r of W[row][col] z[row], a thread per column, 512 columns per task
// (the entries of W above its diagonal are zeros: no triangle logic).  alpha[col] = sum_r part[r][col] is taken by k_alpha_sum afterwards.
__device__ __forceinline__ void flow_apart(const FlowArgs& g, const int r0, const int nrows, const int rblk, const int jg, double* lds) {
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    if (tid < nrows) lds[tid] = g.vz[r0 + tid];              // vector loads: z came from other workgroups of this launch
    __syncthreads();
    const int64_t col = (int64_t)jg * FL_NT + tid;
    if (col < g.npad) {
        const double* Wc = g.bWm + (int64_t)r0 * g.ld + col;
        double a[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll 2
        for (int r = 0; r < nrows; r += 4) {
#pragma unroll
            for (int u = 0; u < 4; ++u) a[u] = fma(Wc[(int64_t)(r + u) * g.ld], lds[r + u], a[u]);
        }
        __hip_atomic_store(g.vpart + (int64_t)rblk * g.npad + col, (a[0] + a[1]) + (a[2] + a[3]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

#define FL_IDLE_LIMIT 4000u           // idle looks (1 .. 16 us apart: ~65 ms) before a workgroup gives up.  Round 5: 60000 (0.9 s) was what the rare stall of tools/flow_soak.py cost; a time-out is a detour since (chain_fallback), so a false one is cheap and a true one should be
#define FL_CALM_AFTER 48u             // idle looks (~0.6 ms) after which a waiting workgroup polls at the calmer rate
#define FL_DEEP_AFTER 128u            // idle looks (~2 ms) before the first deep look of a wait, then one in FL_DEEP_EVERY
#define FL_DEEP_EVERY 32u
#define FL_LA 8                       // positions behind the head of a compare-and-swap queue whose readiness a look already knows

// How a workgroup gets its next tile.  Wave 0 looks at every queue at once, one lane per candidate:
//   * the first `ncas` queues (the few tiles the next chain kernel waits for) are taken READY ONLY: a lane per position head .. head + 7 reads
//     that task's counters; if the head is ready, compare-and-swap head -> head + 1; a failed swap returns the new head, whose readiness the
//     look already knows, so the retry costs one round trip, not a new look (16 tiles that become ready together are out in ~16 round trips);
//   * every other queue is taken EAGERLY: fetch-add on its head (no contention: a look of 4 dependent round trips between two swaps of one
//     word was the whole dispatch rate of the first version, 1 task per 6 us for 456 workgroups) and the workgroup keeps the index as its
//     PENDING task of that queue -- one per queue -- until the counters allow it; meanwhile it runs whatever else it holds or can take.
// Of everything ready it takes the queue with the highest priority.  No workgroup ever waits holding a slot it could use otherwise, and the
// globally first unfinished task is either somebody's pending task with its counters met, or at the head of a queue nobody holds a
// pending task of: progress with any number of resident workgroups.
template <bool MARK>          // MARK: the debugging build of the same kernel (MOGP_FLOW_DEBUG) whose tile bodies leave per-wave progress marks
__global__ __launch_bounds__(FL_NT, 4) void k_flow(const FlowArgs* __restrict__ gp, const FlowHot ha) {
    // (the arguments by POINTER: as a by-value struct their ~60 scalar registers were live for the whole kernel, 30-odd of them spilled into vector lanes)
    const FlowArgs& g = *gp;
    extern __shared__ __attribute__((aligned(16))) double gemm_lds[];
    int* pick = reinterpret_cast<int*>(gemm_lds + FL_LDS_DOUBLES);
    // Round 6: NOTHING of the look is live in registers across a tile.  What wave 0's lanes remember between looks -- the ticket each eager lane holds,
    // the task it looked at last and how many of its dependencies it has seen met, whether its queue is used up -- lives in LDS (`lst`), the lane roles are
    // recomputed per look.  With that state in VGPRs across flow_tile the kernel needed 39 spilled VGPRs and a scratch load INSIDE the k loop; k_gemm's
    // identical tile body has none.  (Scratch accesses are also where the rare stall of rounds 5 - 6 caught its workgroups: DESIGN section 9.)
    int* lst = pick + 4;                                  // [3][64]: pend, met_idx, met_n | exhausted << 8
    double** ep_C = reinterpret_cast<double**>(lst + 192);          // the running tile's C pointer and sign, for its epilogue
    double* ep_alpha = reinterpret_cast<double*>(lst + 194);
    if (threadIdx.x < 64) { lst[threadIdx.x] = -1; lst[64 + threadIdx.x] = -1; lst[128 + threadIdx.x] = 0; }
    for (;;) {
        int tid = threadIdx.x;                            // (re-derived per iteration from an opaque copy: not live across the tile body)
        asm volatile("" : "+v"(tid));
        const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
        unsigned long long t_look = 0;
        if (MARK && g.trace && tid == 0) t_look = wall_clock64();
        if (MARK && g.post && tid == 0) {                                  // looking (word 72: the FIRST look's clock, 74 / 75: where the workgroup runs)
            unsigned* po = g.post + (size_t)(g.post_base + blockIdx.x) * FLOW_POST_W;
            const unsigned now = (unsigned)(wall_clock64() >> 4);
            po[2] = 1u; po[4] = now;
            if (po[72] == 0u) { po[72] = now | 1u; po[74] = __builtin_amdgcn_s_getreg((31 << 11) | 4); po[75] = __builtin_amdgcn_s_getreg((3 << 11) | 20); }
        }
        if (wave == 0) {
            unsigned* heads = ha.flags + ha.base_heads;
            unsigned* err = ha.flags + ha.base_err;
            // wave 0's lane roles
            const int ncl = ha.ncas * FL_LA, nlanes = ncl + (ha.nq - ha.ncas);
            const bool is_cas = lane < ncl, is_eager = lane >= ncl && lane < nlanes;
            const int myq = is_cas ? lane / FL_LA : (is_eager ? ha.ncas + (lane - ncl) : 0), myk = is_cas ? lane % FL_LA : 0;
            int qbase = 0, qsize = 0;
            if (lane < nlanes) { qbase = ha.qmeta[2 * myq]; qsize = ha.qmeta[2 * myq + 1]; }
            int pend = lst[lane];             // eager lanes: the index (inside the queue) this workgroup holds
            int met_idx = lst[64 + lane], met_n = lst[128 + lane] & 0xff;      // the task this lane looked at last, and how many of its leading dependencies it has seen met
            bool exhausted = (lst[128 + lane] >> 8) != 0;                       // eager lanes: the queue has nothing left to take
            unsigned long long cas_heads = 0; // bit q * FL_LA for every compare-and-swap queue
            for (int q = 0; q < ha.ncas; ++q) cas_heads |= 1ull << (q * FL_LA);
            const unsigned long long eager_mask = nlanes >= 64 ? ~0ull << ncl : ((1ull << nlanes) - 1ull) & ~((1ull << ncl) - 1ull);
            unsigned idle = 0, naps = 0;
            int res = -1;
            unsigned nap = 0;
            for (;;) {
                int h = 0, idx = -1;
                // the few queues right behind the critical one are LOOKED AT by every workgroup at every look (their head's readiness, like the
                // compare-and-swap queue's) but taken by fetch-add only when the head is ready: taken only by idle workgroups (like the bulk queues)
                // they starved, kept filled per workgroup their ready tasks sat in busy workgroups' slots
                const bool peek = is_eager && myq < g.nhi && pend < 0 && !exhausted;
                if (is_cas) {
                    h = (int)__hip_atomic_load(heads + myq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    h = __shfl(h, lane - myk, 64);                             // one head value per queue
                    idx = h + myk < qsize ? h + myk : -1;
                } else if (peek) {
                    h = (int)__hip_atomic_load(heads + myq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (h >= qsize) exhausted = true; else idx = h;
                } else if (is_eager) {
                    idx = pend;
                }
                bool ready = false;
                // a DEEP look (round 6): a workgroup that has found nothing for a few ms reads its counters with returning read-modify-writes
                // (fetch-or 0: answered where the atomics are performed) instead of sc1 loads (answered by the XCD's L2), and counts every
                // answer that differs -- see FLOW_DIAG_* and DESIGN section 4 "the stall"
                const bool deep = idle >= FL_DEEP_AFTER && (idle & (FL_DEEP_EVERY - 1u)) == 0u;
                if (deep && is_cas) {
                    int h2 = (int)__hip_atomic_fetch_or(heads + myq, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    h2 = __shfl(h2, lane - myk, 64);
                    if (h2 != h && myk == 0) __hip_atomic_fetch_add(g.diag + FLOW_DIAG_HEAD_STALE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    h = h2;
                    idx = h + myk < qsize ? h + myk : -1;
                }
                if (idx >= 0) {
                    const FlowTask* t = ha.tasks + qbase + idx;
                    const int nd = t->ndep;
                    // counters only grow: a dependency seen met stays met, so a lane that keeps looking at the SAME task re-reads only what it has not yet
                    // seen met (met_n leading dependencies of task met_idx) -- a third of the counter loads of a waiting grid (round 6)
                    if (idx != met_idx) { met_idx = idx; met_n = 0; }
                    ready = true;
                    int lead = met_n;
                    for (int d = 0; d < 4; ++d)
                        if (d >= met_n && d < nd) {
                            if (__hip_atomic_load(ha.flags + t->dep[d], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)t->need[d]) ready = false;
                            else if (lead == d) lead = d + 1;
                        }
                    met_n = lead;
                    if (deep && !ready) {
                        bool ready2 = true;
                        unsigned seen = 0, which = 0;
                        for (int d = 0; d < 4; ++d)
                            if (d < nd) {
                                const unsigned v = __hip_atomic_fetch_or(ha.flags + t->dep[d], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                if (v < (unsigned)t->need[d]) ready2 = false; else { seen = v; which = t->dep[d]; }
                            }
                        if (ready2) {                                          // the loads said "not yet", the memory side says "all there"
                            __hip_atomic_fetch_add(g.diag + FLOW_DIAG_DEP_STALE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            __hip_atomic_store(g.diag + FLOW_DIAG_LAST, which, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            __hip_atomic_store(g.diag + FLOW_DIAG_LAST + 1, seen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            __hip_atomic_store(g.diag + FLOW_DIAG_LAST + 2, (unsigned)blockIdx.x | ((unsigned)__builtin_amdgcn_s_getreg((3 << 11) | 20) << 16), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            ready = true;
                        }
                    }
                }
                if (deep && lane == 0) __hip_atomic_fetch_add(g.diag + FLOW_DIAG_DEEP, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned long long mready = __ballot(ready);
                if (MARK && g.post) {                                  // debugging: freeze at once when anybody has given up (the state the host dumps is then the state of the stall)
                    if (lane == 0) { g.post[(size_t)(g.post_base + blockIdx.x) * FLOW_POST_W + 6] = (unsigned)mready; g.post[(size_t)(g.post_base + blockIdx.x) * FLOW_POST_W + 7] = (unsigned)(mready >> 32); }
                    if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) { res = -3; break; }
                }
                const unsigned long long cand = mready & (cas_heads | eager_mask);
                if (cand) {
                    const int wl = __ffsll((long long)cand) - 1;               // lanes are in priority order
                    int ok = 0;
                    if (lane == wl) {
                        if (is_cas) {
                            unsigned cur = (unsigned)h;
                            for (;;) {
                                unsigned expect = cur;
                                if (__hip_atomic_compare_exchange_strong(heads + myq, &expect, cur + 1u, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                                                         __HIP_MEMORY_SCOPE_AGENT)) { ok = 1; res = qbase + (int)cur; break; }
                                cur = expect;                                  // somebody else moved the head: is the task it now points at one this look saw ready?
                                const unsigned off = cur - (unsigned)h;
                                if (off >= FL_LA || !((mready >> (lane + off)) & 1ull)) break;
                            }
                        } else if (peek) {                                      // the head looked ready: take a ticket; if others were faster the ticket is a
                            const unsigned hh = __hip_atomic_fetch_add(heads + myq, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // later task: hold it
                            if (hh == (unsigned)h) { ok = 1; res = qbase + h; }
                            else if (hh < (unsigned)qsize) { pend = (int)hh; if (MARK && g.trace) g.trace[FLOW_TRACE_W * (size_t)(qbase + (int)hh) + 5] = (1ull << 63) | blockIdx.x; }
                            else exhausted = true;
                        } else {
                            ok = 1; res = qbase + pend;
                            // refill the slot at once (the answer is not needed before the next look) -- except near the end of the
                            // queue, where a task held by a busy workgroup is a task an idle one cannot take
                            if (g.refill && pend + 2 * (int)gridDim.x < qsize) {
                                const unsigned hh = __hip_atomic_fetch_add(heads + myq, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                if (hh < (unsigned)qsize) { pend = (int)hh; if (MARK && g.trace) g.trace[FLOW_TRACE_W * (size_t)(qbase + (int)hh) + 5] = (1ull << 63) | blockIdx.x; } else { pend = -1; exhausted = true; }
                            } else pend = -1;
                        }
                    }
                    ok = __shfl(ok, wl, 64);
                    res = __shfl(res, wl, 64);
                    if (ok) break;
                    continue;                                                  // the head ran away: look again
                }
                // nothing this workgroup holds or may take is ready: take what can be taken eagerly
                bool took = false;
                bool want = is_eager && myq >= g.nhi && pend < 0 && !exhausted;      // (the looked-at queues are taken ready-only, above)
                if (g.claim_one) {                                             // one queue per look, the highest priority first: a workgroup then holds at
                    const unsigned long long mw = __ballot(want);              // most one READY task it is not running (held ready tasks are tasks idle
                    want = want && mw && lane == __ffsll((long long)mw) - 1;   // workgroups cannot take)
                }
                if (want) {
                    const unsigned hh = __hip_atomic_fetch_add(heads + myq, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (hh < (unsigned)qsize) { pend = (int)hh; took = true; if (MARK && g.trace) g.trace[FLOW_TRACE_W * (size_t)(qbase + (int)hh) + 5] = (1ull << 63) | blockIdx.x; } else exhausted = true;      // (trace: who HOLDS the task; overwritten when it starts)
                }
                const bool open = is_cas ? (myk == 0 && h < qsize) : (is_eager && (pend >= 0 || !exhausted));
                if (!__ballot(open)) { res = -2; break; }                      // every queue is empty and nothing is held: done
                if (__ballot(took)) continue;
                // back off: idle workgroups must not crowd the memory system.  Two stages (round 6): naps of up to 2^nap_max units while the wait
                // is an ordinary one (a look that finds nothing waits ~175 us on average), up to 2^nap_calm once it has lasted FL_CALM_AFTER looks -- when most of
                // the grid waits for a few tasks, ~480 workgroups x ~100 counter loads per look are what those few tasks' operand loads queue behind
                const unsigned cap = idle >= FL_CALM_AFTER ? ha.nap_calm : ha.nap_max;
                nap = nap < cap ? nap + 1u : cap;
                for (unsigned z = 0; z < (1u << nap); ++z) __builtin_amdgcn_s_sleep(32);
                ++naps;
                if ((++idle & 15u) == 0u) {
                    if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) { res = -3; break; }
                    if (idle > FL_IDLE_LIMIT) {
                        __hip_atomic_store(err, 0x700u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        res = -3;
                        break;
                    }
                }
            }
            lst[lane] = pend; lst[64 + lane] = met_idx; lst[128 + lane] = met_n | (exhausted ? 0x100 : 0);
            if (lane == 0) {
                pick[0] = res;
                pick[1] = (int)naps;
                if (res >= 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");      // ONE buffer_inv sc1 behind the satisfied counters
            }
            if (MARK && g.post && res < 0) {                           // post-mortem: what this workgroup still held when it left, and why it left
                unsigned* po = g.post + (size_t)(g.post_base + blockIdx.x) * FLOW_POST_W;
                po[8 + lane] = is_eager ? (unsigned)(pend + 1) : 0u;
                if (lane == 0) { po[0] = (unsigned)(res + 4); po[1] = idle; po[2] = 3u; po[5] = (unsigned)(wall_clock64() >> 4); }
            }
        }
        __syncthreads();
        const int ti = __builtin_amdgcn_readfirstlane(pick[0]);
        if (ti < 0) {
            if (ti == -3 && tid == 0) atomicMin(g.info, (unsigned long long)MOGP_INFO_CHAIN_TIMEOUT);
            break;
        }
        const FlowTask* tp = ha.tasks + ti;
        const int var = tp->var, kt = tp->kt;
        const int ab = tp->abuf, bb = tp->bbuf, cb = tp->cbuf;
        const double* Ab = ab == 0 ? ha.bA : ab == 1 ? ha.bL : ab == 2 ? ha.bWt : ab == 3 ? ha.bWm : ha.bB;
        const double* Bb = bb == 0 ? ha.bA : bb == 1 ? ha.bL : bb == 2 ? ha.bWt : bb == 3 ? ha.bWm : ha.bB;
        double* Cb = cb == 0 ? ha.bA : cb == 1 ? ha.bL : cb == 2 ? ha.bWt : cb == 3 ? ha.bWm : ha.bB;
        const double* Ap = Ab + ((int64_t)tp->ar * ha.ld + tp->ac) * MOGP_TILE;
        const double* Bp = Bb + ((int64_t)tp->br * ha.ld + tp->bc) * MOGP_TILE;
        double* Cp = Cb + ((int64_t)tp->cr * ha.ld + tp->cc) * MOGP_TILE;
        const bool fresh = (var & 4) != 0;
        const double alpha = (var & 8) ? -1.0 : 1.0;
        unsigned long long* tr = (MARK && g.trace && tid == 0) ? g.trace + FLOW_TRACE_W * (size_t)ti : nullptr;      // (time stamps: the debugging build only)
        if (tr) { tr[0] = t_look; tr[1] = wall_clock64(); }
        unsigned* prog = (MARK && g.post) ? g.post + (size_t)(g.post_base + blockIdx.x) * FLOW_POST_W + 80 : nullptr;
        if (MARK && g.post && tid == 0) {                                  // running task ti (word 73: tasks so far, 76: this one's start)
            unsigned* po = g.post + (size_t)(g.post_base + blockIdx.x) * FLOW_POST_W;
            po[2] = 2u; po[3] = (unsigned)ti; po[73] += 1u; po[76] = (unsigned)(wall_clock64() >> 4);
        }
        if (var & 32) {                                            // vector task: tile row ar (z) or row block ar .. ar + kt - 1, column group ac (alpha)
            if (g.vy) {
                if (var & 1) flow_apart(g, tp->ar * MOGP_TILE, kt * MOGP_TILE, tp->br, tp->ac, gemm_lds);
                else flow_zrow(g, tp->ar, gemm_lds);
            }
        } else {
        if (var & 16) __builtin_amdgcn_s_setprio(2);
        if (tid == 0) { *ep_C = Cp; *ep_alpha = alpha; }           // what the tile's epilogue needs, parked in LDS across the k loop (whose barriers publish it)
        switch (var & 3) {
            case 0: flow_tile<0, 0, MARK>(Ap, Bp, Cp, ha.ld, kt, fresh, alpha, gemm_lds, tr, prog, ep_C, ep_alpha); break;
            case 1: flow_tile<0, 1, MARK>(Ap, Bp, Cp, ha.ld, kt, fresh, alpha, gemm_lds, tr, prog, ep_C, ep_alpha); break;
            default: flow_tile<1, 1, MARK>(Ap, Bp, Cp, ha.ld, kt, fresh, alpha, gemm_lds, tr, prog, ep_C, ep_alpha); break;
        }
        if (var & 16) __builtin_amdgcn_s_setprio(0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // every storing wave drains its own write-through stores
        __syncthreads();                                           // ... and the tile's last LDS reads are behind us
        if (tid == 0) {
            const unsigned s0 = tp->sig[0], s1 = tp->sig[1];
            if (s0 != FLOW_NOSIG) __hip_atomic_fetch_add(ha.flags + s0, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (s1 != FLOW_NOSIG) __hip_atomic_fetch_add(ha.flags + s1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (MARK && g.done) g.done[ti] = 1;
            if (MARK && g.post) g.post[(size_t)(g.post_base + blockIdx.x) * FLOW_POST_W + 77] = (unsigned)(wall_clock64() >> 4);
            if (tr) {
                tr[4] = wall_clock64();
                tr[5] = ((unsigned long long)(unsigned)pick[1] << 32) | ((unsigned long long)__builtin_amdgcn_s_getreg((3 << 11) | 20) << 16) | blockIdx.x;
            }
        }
    }
}


// ---- the task graph (host) ------------------------------------------------------------------------------------------------------------
// Flag words.  S[i][j]: panel updates applied to Schur tile (i, j);  R[i][c]: the same summed over row i's tiles in column block c (i outside
// that block);  DG[c]: summed over the diagonal block c;  PN[b][i]: tiles of panel row i of block b that exist;  CH[b]: workgroups of chain
// kernel b that have finished;  WT[i][j]: versions of running-product tile (i, j);  WC[r][j]: the same summed over the tile rows of row block
// r;  WF[b][j]: finished tile rows of W[b, j];  KV[i][j]: accumulations into inverse tile (i, j).
namespace {
// The sequential algorithm the keys follow: superstep r = [ the running product's updates INTO row block r, by source block b' < r (T7 / T8:
// left-looking on the inverse side, so that nothing of it is due before its row block is) | chain(r) | T6(r): the final row block r of W |
// panel(r) | update(r) (right-looking on the Schur side) | T9(r) ].  key = FLOW_KEY_STEP r + position inside the superstep.
enum { PH_INTO = 0, PH_CHAIN = 600, PH_T6 = 601, PH_PANEL = 602, PH_UPDATE = 603, PH_T9 = 604, PH_ZROW = 605, PH_APART = 606, PH_XS = 607, PH_TU = 608 };
enum { Q_LOOK2 = 0, Q_INVCRIT = 1, Q_SEMI = 2 };
enum { BUF_A = 0, BUF_L = 1, BUF_WT = 2, BUF_WM = 3, BUF_B = 4 };
}

void flow_build(int nb, int ob, FlowPlan& p, int rhs_nt, bool replay) {
    p = FlowPlan();
    p.nb = nb; p.ob = ob; p.rhs_nt = rhs_nt; p.replay = replay;
    const int no = (nb + ob - 1) / ob;
    p.nouter = no;
    const uint32_t base_S = 0, base_R = base_S + (uint32_t)nb * nb, base_DG = base_R + (uint32_t)nb * no, base_PN = base_DG + no,
                   base_CH = base_PN + (uint32_t)no * nb, base_WT = base_CH + no, base_WC = base_WT + (uint32_t)nb * nb,
                   base_WF = base_WC + (uint32_t)no * nb, base_KV = base_WF + (uint32_t)no * nb, base_WR = base_KV + (uint32_t)nb * nb,
                   base_ZR = base_WR + no, base_TS = base_ZR + no, base_TR = base_TS + (uint32_t)rhs_nt * nb, base_XN = base_TR + (uint32_t)rhs_nt * no,
                   base_end = base_XN + (uint32_t)no * rhs_nt;
    p.base_heads = (int)base_end; p.base_err = p.base_heads + FLOW_MAXQ; p.nflags = p.base_err + 1;
    auto S = [&](int i, int j) { return base_S + (uint32_t)i * nb + j; };
    auto R = [&](int i, int c) { return base_R + (uint32_t)i * no + c; };
    auto DG = [&](int c) { return base_DG + (uint32_t)c; };
    auto PN = [&](int b, int i) { return base_PN + (uint32_t)b * nb + i; };
    auto CH = [&](int b) { return base_CH + (uint32_t)b; };
    auto WT = [&](int i, int j) { return base_WT + (uint32_t)i * nb + j; };
    auto WC = [&](int r, int j) { return base_WC + (uint32_t)r * nb + j; };
    auto WF = [&](int b, int j) { return base_WF + (uint32_t)b * nb + j; };
    auto KV = [&](int i, int j) { return base_KV + (uint32_t)i * nb + j; };
    auto WR = [&](int b) { return base_WR + (uint32_t)b; };          // finished T6 tasks of row block b
    auto ZR = [&](int b) { return base_ZR + (uint32_t)b; };          // finished z tile rows of row block b
    auto TS = [&](int i, int j) { return base_TS + (uint32_t)i * nb + j; };       // right-hand sides: updates applied to tile (i, j) of T
    auto TR = [&](int i, int c) { return base_TR + (uint32_t)i * no + c; };       // ... summed over row i's tiles in column block c
    auto XN = [&](int b, int i) { return base_XN + (uint32_t)b * rhs_nt + i; };   // solved tiles X[i][block b] that exist
    auto k0 = [&](int b) { return b * ob; };
    auto k1 = [&](int b) { return std::min(nb, (b + 1) * ob); };
    auto nk = [&](int b) { return k1(b) - k0(b); };
    auto blk = [&](int i) { return i / ob; };
    auto chain_wgs = [&](int b) { return nk(b) > 1 ? 13u : 2u; };          // chain.hip: CH_NWG workgroups, 2 for a one-tile block

    // queues 0 .. 3 as named above; then, by the superstep d that needs them (earliest deadline first): INTO[d] = the running product's updates
    // into row block d from the source blocks <= d - 2, and TRAIL[d - 3] = the trailing update of panel d - 3 right of its two look-ahead
    // column blocks (its first columns are block d's); last the accumulations of the inverse, which nothing waits for.
    std::vector<FlowTask> q[3], acc, vec;
    std::vector<std::vector<FlowTask>> into(no), trail(no);
    auto mk = [&](int b, int phase, int var, int kt) {
        FlowTask t{};
        t.var = (uint8_t)var; t.kt = (uint16_t)kt; t.ndep = 0; t.sig[0] = t.sig[1] = FLOW_NOSIG; t.key = (uint32_t)(b * FLOW_KEY_STEP + phase);
        return t;
    };
    auto dep = [&](FlowTask& t, uint32_t idx, unsigned need) {
        if (need == 0) return;
        t.dep[t.ndep] = idx; t.need[t.ndep] = (uint16_t)need; ++t.ndep;
    };
    auto opA = [](FlowTask& t, int buf, int r, int c) { t.abuf = (uint8_t)buf; t.ar = (uint16_t)r; t.ac = (uint16_t)c; };
    auto opB = [](FlowTask& t, int buf, int r, int c) { t.bbuf = (uint8_t)buf; t.br = (uint16_t)r; t.bc = (uint16_t)c; };
    auto opC = [](FlowTask& t, int buf, int r, int c) { t.cbuf = (uint8_t)buf; t.cr = (uint16_t)r; t.cc = (uint16_t)c; };
    const int KB = MOGP_TILE / FL_BK;                 // k blocks per tile
    double tiles_k = 0.0;                             // sum over tasks of k blocks

    // The two products a chain kernel waits for -- the panel rows of block b + 1 ("mini-panel") and the last update of its diagonal block -- are NOT
    // tasks: they run as launches on the private stream between the chain kernels, on 64-row tiles on the reserved CUs as in potri.hip (as
    // 128 x 128 tasks next to the bulk work they took 230 us per block against 140 us there, and the chain phase is where workgroups wait).
    // The mini-panel is 64 x 128 tiles: PN(b, i) of its rows counts two tiles per column tile.
    // replay (mogp_model_flow_replay, a measurement mode): those two products ARE tasks -- 128 x 128 like everything else --, at the head of the
    // critical queue, and the host presets the chain kernels' counters: with W_KK left in place by an earlier evaluation the dataflow kernel then
    // runs the whole evaluation ALONE, which is what a serialising profiler (rocprofv3 --pmc) needs to count its traffic.
    auto crit_row = [&](int b, int i) { return !replay && i >= k1(b) && i < std::min(nb, k1(b) + ob); };
    auto pn_need = [&](int b, int i) { return (unsigned)((crit_row(b, i) ? 2 : 1) * nk(b)); };
    // panel tile (b, i, c): L[i][k0 + c] = sum_{cc <= c} A[i][k0 + cc] W_bb[c][cc]^T
    auto panel = [&](int b, int i, int c, bool prio) {
        FlowTask t = mk(b, PH_PANEL, 0 | 4 | (prio ? 16 : 0), KB * (c + 1));
        opA(t, BUF_A, i, k0(b)); opB(t, BUF_WM, k0(b) + c, k0(b)); opC(t, BUF_L, i, k0(b) + c);
        dep(t, CH(b), chain_wgs(b));
        dep(t, R(i, b), (unsigned)(nk(b) * b));
        t.sig[0] = PN(b, i);
        tiles_k += t.kt;
        return t;
    };
    // update (b, i, j): A[i][j] -= L[i][K] L[j][K]^T
    auto update = [&](int b, int i, int j, bool prio) {
        FlowTask t = mk(b, PH_UPDATE, 0 | 8 | (prio ? 16 : 0), KB * nk(b));
        opA(t, BUF_L, i, k0(b)); opB(t, BUF_L, j, k0(b)); opC(t, BUF_A, i, j);
        dep(t, PN(b, i), pn_need(b, i));
        if (j != i) dep(t, PN(b, j), pn_need(b, j));
        dep(t, S(i, j), (unsigned)b);
        t.sig[0] = S(i, j);
        t.sig[1] = blk(i) == blk(j) ? DG(blk(j)) : R(i, blk(j));
        tiles_k += t.kt;
        return t;
    };
    // T7 (b, i, c): Wt[i][k0 + c] = -L[i][K, from tile c on] W_bb[from tile row c on, tile column c]
    auto t7 = [&](int b, int i, int c) {
        FlowTask t = mk(blk(i), PH_INTO + 2 * b, 1 | 4 | 8, KB * (nk(b) - c));
        opA(t, BUF_L, i, k0(b) + c); opB(t, BUF_WM, k0(b) + c, k0(b) + c); opC(t, BUF_WT, i, k0(b) + c);
        dep(t, PN(b, i), pn_need(b, i));
        t.sig[0] = WT(i, k0(b) + c); t.sig[1] = WC(blk(i), k0(b) + c);
        tiles_k += t.kt;
        return t;
    };
    // T8 (b, i, j): Wt[i][j] -= L[i][K] W[K][j]      (j < k0)
    auto t8 = [&](int b, int i, int j) {
        FlowTask t = mk(blk(i), PH_INTO + 2 * b + 1, 1 | 8, KB * nk(b));
        opA(t, BUF_L, i, k0(b)); opB(t, BUF_WM, k0(b), j); opC(t, BUF_WT, i, j);
        dep(t, PN(b, i), pn_need(b, i));
        dep(t, WF(b, j), (unsigned)nk(b));
        dep(t, WT(i, j), (unsigned)(b - blk(j)));
        t.sig[0] = WT(i, j); t.sig[1] = WC(blk(i), j);
        tiles_k += t.kt;
        return t;
    };
    // T6 (b, ti, j): W[k0 + ti][j] = sum_{tt <= ti} W_bb[ti][tt] Wt[k0 + tt][j]      (j < k0)
    auto t6 = [&](int b, int ti, int j) {
        FlowTask t = mk(b, PH_T6, 1 | 4, KB * (ti + 1));
        opA(t, BUF_WM, k0(b) + ti, k0(b)); opB(t, BUF_WT, k0(b), j); opC(t, BUF_WM, k0(b) + ti, j);
        dep(t, CH(b), chain_wgs(b));
        dep(t, WC(b, j), (unsigned)(nk(b) * (b - blk(j))));
        t.sig[0] = WF(b, j); t.sig[1] = WR(b);
        tiles_k += t.kt;
        return t;
    };
    // T9 (b, i, j): B[i][j] (+)= W[K][i]^T W[K][j]      (j <= i < k1)
    // A tile column i INSIDE the block is a column of the lower-triangular W_KK: its tile rows above i are zeros, so the product starts at tile row i
    // (same bits -- the skipped k blocks add exact zeros -- and 2.3 % fewer k blocks per evaluation at 64 tile rows; MOGP_FLOW_T9FULL=1: the full range)
    static const bool t9_full = std::getenv("MOGP_FLOW_T9FULL") && std::atoi(std::getenv("MOGP_FLOW_T9FULL")) != 0;
    auto t9 = [&](int b, int i, int j) {
        const int r0 = (i >= k0(b) && !t9_full) ? i : k0(b);
        FlowTask t = mk(b, PH_T9, 2 | (i >= k0(b) ? 4 : 0), KB * (k1(b) - r0));
        opA(t, BUF_WM, r0, i); opB(t, BUF_WM, r0, j); opC(t, BUF_B, i, j);
        bool ch = false;
        if (i >= k0(b)) { dep(t, CH(b), chain_wgs(b)); ch = true; } else dep(t, WF(b, i), (unsigned)nk(b));
        if (j != i) { if (j >= k0(b)) { if (!ch) dep(t, CH(b), chain_wgs(b)); } else dep(t, WF(b, j), (unsigned)nk(b)); }
        dep(t, KV(i, j), (unsigned)(b - blk(i)));
        t.sig[0] = KV(i, j);
        tiles_k += t.kt;
        return t;
    };

    // ---- right-hand sides (rhs_nt > 0: the prediction's X L^T = T, rows = tile rows of test points, no inverse).  T lives where the running product
    // does otherwise (BUF_WT), X where the inverse does (BUF_B): same leading dimension, the kernel needs no change.
    // XS (b, i, c): X[i][k0 + c] = sum_{cc <= c} T[i][k0 + cc] W_bb[c][cc]^T        -- the panel's product on a row of right-hand sides
    auto xs = [&](int b, int i, int c) {
        FlowTask t = mk(b, PH_XS, 0 | 4, KB * (c + 1));
        opA(t, BUF_WT, i, k0(b)); opB(t, BUF_WM, k0(b) + c, k0(b)); opC(t, BUF_B, i, k0(b) + c);
        dep(t, CH(b), chain_wgs(b));
        dep(t, TR(i, b), (unsigned)(nk(b) * b));
        t.sig[0] = XN(b, i);
        tiles_k += t.kt;
        return t;
    };
    // TU (b, i, j): T[i][j] -= X[i][K] L[j][K]^T      (j >= k1), keyed by its SOURCE block (right-looking): everything a solved block can update is ready at
    // once and fills the chip.  (Keyed by the column block that needs the tile -- deadline first, in ONE backlog queue -- only the head of the queue
    // was ever eligible and two thirds of the workgroups idled: configs[3] 62.3 instead of 45.6 ms.)
    auto tu = [&](int b, int i, int j) {
        FlowTask t = mk(b, PH_TU, 0 | 8, KB * nk(b));
        opA(t, BUF_B, i, k0(b)); opB(t, BUF_L, j, k0(b)); opC(t, BUF_WT, i, j);
        dep(t, XN(b, i), (unsigned)nk(b));
        dep(t, PN(b, j), pn_need(b, j));
        dep(t, TS(i, j), (unsigned)b);
        t.sig[0] = TS(i, j); t.sig[1] = TR(i, blk(j));
        tiles_k += t.kt;
        return t;
    };
    std::vector<FlowTask> rrest;       // (the updates into the next 3 / 6 / 12 column blocks in a queue of their own ahead of the rest: 45.6 / 45.6 / 45.9 vs 45.6 ms, dropped)

    p.chain.resize(no);
    for (int b = 0; b < no; ++b) {
        const int a0 = k1(b), a1 = std::min(nb, a0 + ob), a2 = std::min(nb, a1 + ob);        // rows of block b+1: [a0, a1), of block b+2: [a1, a2)
        FlowPlan::Chain& c = p.chain[b];
        c = FlowPlan::Chain{};
        c.done_idx = CH(b); c.expect = chain_wgs(b);
        // the mini-panel reads A[rows of block b + 1][columns of block b]: every update of those tiles (from the panels before b) must be in
        c.t1_nwait = 0;
        for (int i = a0; i < a1 && b > 0; ++i) { c.t1_widx[c.t1_nwait] = R(i, b); c.t1_wval[c.t1_nwait] = (uint32_t)(nk(b) * b); ++c.t1_nwait; }
        c.t1_sig_base = PN(b, a0); c.t1_sig_per_row = 2u * (uint32_t)nk(b);
        // the next-diagonal update is the LAST update of block b + 1's diagonal block: the b earlier ones (dataflow tasks) first
        if (a1 > a0) { const int n1 = a1 - a0; c.t2_widx = DG(b + 1); c.t2_wval = (uint32_t)(n1 * (n1 + 1) / 2 * b); }
        if (replay) {
            for (int i = a0; i < a1; ++i) for (int cc = 0; cc < nk(b); ++cc) q[Q_LOOK2].push_back(panel(b, i, cc, true));
            for (int i = a0; i < a1; ++i) for (int j = a0; j <= i; ++j) q[Q_LOOK2].push_back(update(b, i, j, true));
        }
        // Q_LOOK2: what the critical tasks of block b + 1 wait for
        for (int i = a1; i < a2; ++i) for (int cc = 0; cc < nk(b); ++cc) q[Q_LOOK2].push_back(panel(b, i, cc, true));
        for (int i = a1; i < a2; ++i) for (int j = a0; j < a1; ++j) q[Q_LOOK2].push_back(update(b, i, j, true));
        for (int i = a1; i < a2; ++i) for (int j = a1; j <= i; ++j) q[Q_LOOK2].push_back(update(b, i, j, true));
        // Q_SEMI: the rest of the panel, of the next block's columns and of the columns of the block after it
        for (int i = a2; i < nb; ++i) for (int cc = 0; cc < nk(b); ++cc) q[Q_SEMI].push_back(panel(b, i, cc, false));
        for (int i = a2; i < nb; ++i) for (int j = a0; j < a2; ++j) q[Q_SEMI].push_back(update(b, i, j, false));
        // TRAIL[b]: everything to the right
        for (int i = a2; i < nb; ++i) for (int j = a2; j <= i; ++j) trail[b].push_back(update(b, i, j, false));
        if (rhs_nt > 0) {
            // the substitution's own cycle (solved block b -> the next block's columns of T) in the inverse cycle's place; the rest is backlog
            for (int i = 0; i < rhs_nt; ++i) for (int cc = 0; cc < nk(b); ++cc) q[Q_INVCRIT].push_back(xs(b, i, cc));
            for (int i = 0; i < rhs_nt; ++i) for (int j = a0; j < a1; ++j) q[Q_INVCRIT].push_back(tu(b, i, j));
            for (int i = 0; i < rhs_nt; ++i) for (int j = a1; j < nb; ++j) rrest.push_back(tu(b, i, j));
            continue;
        }
        // Q_INVCRIT: the serial cycle of the inverse: final row block b -> running product of the next block's rows
        for (int j = 0; j < k0(b); ++j) for (int ti = 0; ti < nk(b); ++ti) q[Q_INVCRIT].push_back(t6(b, ti, j));
        for (int i = a0; i < a1; ++i) for (int cc = 0; cc < nk(b); ++cc) q[Q_INVCRIT].push_back(t7(b, i, cc));
        for (int i = a0; i < a1; ++i) for (int j = 0; j < k0(b); ++j) q[Q_INVCRIT].push_back(t8(b, i, j));
        // INTO[b]: the running product of row block b from the source blocks b' <= b - 2 (the last one, b' = b - 1, is on the serial cycle above)
        for (int bs = 0; bs + 2 <= b; ++bs) {
            for (int i = k0(b); i < k1(b); ++i) for (int cc = 0; cc < nk(bs); ++cc) into[b].push_back(t7(bs, i, cc));
            for (int i = k0(b); i < k1(b); ++i) for (int j = 0; j < k0(bs); ++j) into[b].push_back(t8(bs, i, j));
        }
        for (int i = 0; i < k1(b); ++i) for (int j = 0; j <= i; ++j) acc.push_back(t9(b, i, j));
        // z = W y for the tile rows of row block b as soon as they are final, then this row block's share of alpha = W^T z
        for (int i = k0(b); i < k1(b); ++i) {
            FlowTask t = mk(b, PH_ZROW, 32, 1);
            t.ar = (uint16_t)i;
            dep(t, CH(b), chain_wgs(b));
            dep(t, WR(b), (unsigned)(nk(b) * k0(b)));
            t.sig[0] = ZR(b);
            vec.push_back(t);
        }
        for (int jg = 0; jg * 4 < k1(b); ++jg) {
            FlowTask t = mk(b, PH_APART, 32 | 1, nk(b));
            t.ar = (uint16_t)k0(b); t.br = (uint16_t)b; t.ac = (uint16_t)jg;
            dep(t, ZR(b), (unsigned)nk(b));
            vec.push_back(t);
        }
    }
    std::vector<const std::vector<FlowTask>*> order = {&q[Q_LOOK2], &q[Q_SEMI], &q[Q_INVCRIT], &vec};      // look-ahead, the rest of the Schur side next to the chain, then the inverse's cycle, z / alpha
    for (int d = 2; d < no + 3; ++d) {
        if (d < no && !into[d].empty()) order.push_back(&into[d]);
        if (d - 3 >= 0 && d - 3 < no && !trail[d - 3].empty()) order.push_back(&trail[d - 3]);
    }
    order.push_back(rhs_nt > 0 ? &rrest : &acc);
    for (size_t k = order.size(); k-- > 1;) if (order[k]->empty()) order.erase(order.begin() + (long)k);      // (the first queue keeps its place: it is the compare-and-swap one)
    if ((int)order.size() > FLOW_MAXQ) {          // more deadline queues than the kernel has lanes for (above 85 tile rows): fold the farthest deadlines into one queue
        std::vector<FlowTask> rest;
        // merged in superstep order so that the folded queue stays sorted by key
        std::vector<const std::vector<FlowTask>*> tail(order.begin() + (FLOW_MAXQ - 2), order.end() - 1);
        for (auto* v : tail) rest.insert(rest.end(), v->begin(), v->end());
        std::stable_sort(rest.begin(), rest.end(), [](const FlowTask& x, const FlowTask& y) { return x.key < y.key; });
        p.folded = rest;
        order.erase(order.begin() + (FLOW_MAXQ - 2), order.end() - 1);
        order.insert(order.end() - 1, &p.folded);
    }
    p.nq = (int)order.size();
    for (int qi = 0; qi < p.nq; ++qi) {
        p.qbase[qi] = (int)p.tasks.size(); p.qsize[qi] = (int)order[qi]->size();
        p.tasks.insert(p.tasks.end(), order[qi]->begin(), order[qi]->end());
    }
    p.flops = 2.0 * MOGP_TILE * MOGP_TILE * FL_BK * tiles_k;
}

// alpha[col] = sum over the row blocks r >= col's block of part[r][col], in that order
__global__ void k_alpha_sum(const double* __restrict__ part, int64_t npad, int no, int ob, double* __restrict__ out) {
    const int64_t col = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (col >= npad) return;
    double s = 0.0;
    for (int r = (int)(col / MOGP_TILE) / ob; r < no; ++r) s += part[(int64_t)r * npad + col];
    out[col] = s;
}
int launch_flow_alpha_sum(const Spd& w, double* alpha, hipStream_t st) {
    const FlowPlan& p = w.flow_cur ? *w.flow_cur : w.flow;          // the plan the kernel just ran (the measurement plan has its own object)
    hipLaunchKernelGGL(k_alpha_sum, dim3((unsigned)((w.Npad + 255) / 256)), dim3(256), 0, st, w.vec_part, w.Npad, p.nouter, p.ob, alpha);
    HIP_TRY(hipGetLastError());
    return 0;
}

// ---- the schedule -----------------------------------------------------------------------------------------------------------------------
// MOGP_FLOW=0: the stream schedule of potri.hip everywhere.  MOGP_FLOW_MIN=n: the smallest number of 128-row tiles that takes the dataflow
// form.  With the two products between chain kernels as launches on the private stream (64-row tiles on the reserved CUs, as in the stream
// schedule) the dataflow form is the faster one at every size measured: 900 points 1.07 vs 1.11 ms, 1700: 1.73 vs 1.92, 4097: 4.02 vs 4.37,
// 8192: 10.5 vs 12.8.
bool flow_enabled(const mogp_model* m, const Spd& w) {
    const char* eo = std::getenv("MOGP_FLOW");                 // read per call: tests switch it inside one process
    const char* en = std::getenv("MOGP_FLOW_MIN");
    const int on = eo ? std::atoi(eo) : 1, nmin = en ? std::atoi(en) : 8;          // (two outer blocks: below that there is nothing to overlap)
    if (!on || (m->no_flow && m->n_fact < m->flow_retry_at) || !chain_enabled(m) || !m->ctx->st_priv) return false;      // (after a time-out: the stream schedule until flow_retry_at, then another try)
    if (m->kinv_sparse && &w == &m->k) return false;          // a planned (partial) inverse: the stream schedule knows how
    return w.nb >= nmin && w.nb <= 0xfff0;
}

int spd_potri_flow(mogp_model* m, Spd& w, const FlowRhs* rhs) {
    const int nb = w.nb, ob = 4;
    const int64_t ld = w.Npad;
    int rc;
    const auto host_t0 = std::chrono::steady_clock::now();     // how long the HOST takes to enqueue this evaluation (chain_fallback reports it: a stalled host looks like a stalled device)
    hipStream_t crit = m->st, priv = m->st_priv, bulk = m->st2;
    if (w.Wm.n < (size_t)ld * ld) {                      // nothing ever writes above the block diagonal of W: keep it zero
        if ((rc = w.Wm.ensure((size_t)ld * ld))) return rc;
        HIP_TRY(hipMemsetAsync(w.Wm.p, 0, (size_t)ld * ld * sizeof(double), crit));
    }
    if ((rc = w.Lm.ensure((size_t)ld * ld))) return rc;
    if (!rhs && (rc = w.Wt.ensure((size_t)ld * ld))) return rc;
    // two plans per system: the gradient's (factorisation + inversion) and the prediction's (factorisation + substitution of right-hand sides)
    const bool replay = m->replay_flow && !rhs && &w == &m->k;        // measurement mode: the dataflow kernel alone on the replay plan
    FlowPlan& plan = rhs ? w.flow_rhs : replay ? w.flow_replay : w.flow;
    DevBuf<FlowTask>& d_tasks = rhs ? w.flow_tasks_rhs : replay ? w.flow_tasks_replay : w.flow_tasks;
    DevBuf<int>& d_qmeta = rhs ? w.flow_qmeta_rhs : replay ? w.flow_qmeta_replay : w.flow_qmeta;
    const int rhs_nt = rhs ? rhs->nt : 0;
    if (plan.nb != nb || plan.ob != ob || plan.rhs_nt != rhs_nt) {
        flow_build(nb, ob, plan, rhs_nt, replay);
        if ((rc = d_tasks.ensure(plan.tasks.size()))) return rc;
        if ((rc = d_qmeta.ensure(2 * FLOW_MAXQ))) return rc;
        int qm[2 * FLOW_MAXQ] = {0};
        for (int q = 0; q < plan.nq; ++q) { qm[2 * q] = plan.qbase[q]; qm[2 * q + 1] = plan.qsize[q]; }
        HIP_TRY(hipStreamSynchronize(crit));               // (a previous run may still read the old copy)
        HIP_TRY(dev_upload(d_tasks.p, plan.tasks.data(), plan.tasks.size() * sizeof(FlowTask)));
        HIP_TRY(dev_upload(d_qmeta.p, qm, sizeof(qm)));
    }
    if ((rc = w.flow_flags.ensure((size_t)plan.nflags))) return rc;
    if (!w.flow_diag.p) {                                // diagnostic counters that outlive an evaluation (mogp_model_flow_diag)
        if ((rc = w.flow_diag.ensure(FLOW_DIAG_WORDS))) return rc;
        HIP_TRY(hipMemsetAsync(w.flow_diag.p, 0, FLOW_DIAG_WORDS * sizeof(unsigned), crit));
    }
    w.flow_cur = &plan;
    const FlowPlan& p = plan;
    const int nouter = p.nouter;
    if ((rc = w.chain_flags.ensure((size_t)(nouter + 1) * MOGP_CHAIN_FLAGS))) return rc;
    HIP_TRY(hipMemsetAsync(w.chain_flags.p, 0, (size_t)(nouter + 1) * MOGP_CHAIN_FLAGS * sizeof(unsigned), crit));
    HIP_TRY(hipMemsetAsync(w.flow_flags.p, 0, (size_t)p.nflags * sizeof(unsigned), crit));
    if (replay) {                                        // every chain kernel "has finished": W_KK is what the last evaluation left in Wm's diagonal blocks
        std::vector<unsigned> preset((size_t)p.nflags, 0u);
        for (const FlowPlan::Chain& c : p.chain) preset[c.done_idx] = c.expect;
        HIP_TRY(hipMemcpyAsync(w.flow_flags.p, preset.data(), preset.size() * sizeof(unsigned), hipMemcpyHostToDevice, crit));
        HIP_TRY(hipStreamSynchronize(crit));
    }
    if (!rhs && w.want_vec && w.vec_zz) HIP_TRY(hipMemsetAsync(w.vec_zz, 0, (size_t)((ld + 3) / 4) * sizeof(double), crit));   // the tile rows' z^T z parts land in the first nb entries
    static const bool want_trace = std::getenv("MOGP_FLOW_TRACE") && std::atoi(std::getenv("MOGP_FLOW_TRACE")) != 0;
    if (want_trace) {
        if ((rc = w.flow_trace.ensure(FLOW_TRACE_W * p.tasks.size() + 4 * (size_t)nouter))) return rc;
        HIP_TRY(hipMemsetAsync(w.flow_trace.p, 0, w.flow_trace.n * sizeof(unsigned long long), crit));
    }
    while ((int)w.inv_ev.size() < 4) {
        hipEvent_t e;
        HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        w.inv_ev.push_back(e);
    }
    hipEvent_t start = w.inv_ev[0], e_flow = w.inv_ev[1], e_chain = w.inv_ev[2];
    HIP_TRY(hipEventRecord(start, crit));                // the Gram matrix is in place, every counter is zero
    HIP_TRY(hipStreamWaitEvent(bulk, start, 0));
    HIP_TRY(hipStreamWaitEvent(priv, start, 0));

    static std::atomic<unsigned long long> attr_done{0ull};                 // one bit per device
    static std::atomic<unsigned long long> attr_done_dbg{0ull};
    { int r__ = set_max_dynamic_lds(reinterpret_cast<const void*>(k_flow<false>), FL_LDS_BYTES, attr_done); if (r__) return r__; }
    { int r__ = set_max_dynamic_lds(reinterpret_cast<const void*>(k_flow<true>), FL_LDS_BYTES, attr_done_dbg); if (r__) return r__; }
    static const int wg_per_cu = std::getenv("MOGP_FLOW_WGS") ? std::max(1, std::atoi(std::getenv("MOGP_FLOW_WGS"))) : 2;
    int cus = (m->ctx->ncu > 0 ? m->ctx->ncu : 256) - m->ctx->ncu_reserved;
    // Round 6: the kernel does NOT fill the bulk CUs.  With two workgroups on every one of them (60 per XCD: the whole vector register file and 149 of
    // 160 KB of LDS of each CU) about one evaluation in 7000 stood still -- the memory operations of the LAST-dispatched workgroup of every XCD (the 57th: index 450 .. 455 of 480,
    // whatever it was doing: operand loads, write-through stores, the polls of a look) stopped completing until the other workgroups left the kernel at a time-out.  56 per XCD:
    // none in 40 000 + 100 000 soaked evaluations (58 and 59 per XCD: 5 in 30 000 each).  MOGP_FLOW_DROP_CUS=0 fills them again.
    { static const int drop = std::getenv("MOGP_FLOW_DROP_CUS") ? std::atoi(std::getenv("MOGP_FLOW_DROP_CUS")) : 16;
      if (drop > 0 && drop < cus) cus -= drop; }
    FlowArgs g{};
    g.bA = w.A.p; g.bL = w.Lm.p; g.bWt = rhs ? rhs->T : w.Wt.p; g.bWm = w.Wm.p; g.bB = rhs ? rhs->X : w.B.p; g.ld = ld;       // (right-hand sides: T where the running product is, X where the inverse is)
    g.tasks = d_tasks.p; g.qmeta = d_qmeta.p; g.flags = w.flow_flags.p;
    g.nq = p.nq; g.ncas = FLOW_NCAS; g.base_heads = p.base_heads; g.base_err = p.base_err; g.info = m->d_info.p;
    g.trace = want_trace ? w.flow_trace.p : nullptr;
    g.diag = w.flow_diag.p;
    static const bool want_post = std::getenv("MOGP_FLOW_DEBUG") && std::atoi(std::getenv("MOGP_FLOW_DEBUG")) != 0;
    if (want_post) {
        const size_t nwg = (size_t)wg_per_cu * ((size_t)cus + (size_t)m->ctx->ncu_reserved);
        if ((rc = w.flow_post.ensure(nwg * FLOW_POST_W))) return rc;
        HIP_TRY(hipMemsetAsync(w.flow_post.p, 0, w.flow_post.n * sizeof(unsigned), bulk));     // (bulk: in front of the kernel, behind `start`)
        g.post = w.flow_post.p;
        if ((rc = w.flow_done.ensure(p.tasks.size()))) return rc;
        HIP_TRY(hipMemsetAsync(w.flow_done.p, 0, p.tasks.size(), bulk));
        g.done = w.flow_done.p;
    }
    g.npad = ld;
    { const char* e = std::getenv("MOGP_FLOW_REFILL"); g.refill = e ? std::atoi(e) : 0; }
    { const char* e = std::getenv("MOGP_FLOW_CLAIM1"); g.claim_one = e ? std::atoi(e) : 0; }
    { const char* e = std::getenv("MOGP_FLOW_NHI"); g.nhi = e ? std::atoi(e) : 0; }            // measured 3 / 4 (semi, the inverse cycle, z and alpha looked at by everybody): 10.74-10.79 vs 10.52-10.65 ms
    { const char* e = std::getenv("MOGP_FLOW_NAP"); g.nap_max = e ? (unsigned)std::max(0, std::atoi(e)) : 4u; }
    { const char* e = std::getenv("MOGP_FLOW_NAP_CALM"); g.nap_calm = e ? (unsigned)std::max(0, std::atoi(e)) : 0u; if (g.nap_calm < g.nap_max) g.nap_calm = g.nap_max; }      // (off by default: the calmer rate did not prevent the stall it was built against, and costs reaction time)
    w.vec_done = false;
    if (!rhs && w.want_vec && w.vec_y && w.vec_z && w.vec_zz && w.vec_part) {       // z = W y, alpha = W^T z as tasks of the same kernel
        g.vy = w.vec_y; g.vz = w.vec_z; g.vzz = w.vec_zz; g.vpart = w.vec_part;
        w.vec_done = true;
    }
    hipEvent_t pe0 = nullptr, pe1 = nullptr;
    if (m->profiling) {
        if (m->gemm_ev_used + 2 > m->gemm_ev.size())
            for (int i = 0; i < 64; ++i) { hipEvent_t e; HIP_TRY(hipEventCreate(&e)); m->gemm_ev.push_back(e); }
        pe0 = m->gemm_ev[m->gemm_ev_used++]; pe1 = m->gemm_ev[m->gemm_ev_used++];
        HIP_TRY(hipEventRecord(pe0, bulk));
    }
    if (rhs && rhs->ready) HIP_TRY(hipStreamWaitEvent(bulk, rhs->ready, 0));          // the right-hand sides are in place (the chain kernels need not wait for them)
    const bool dbg_kernel = g.post != nullptr || g.trace != nullptr;        // time stamps and the post-mortem: the debugging build of the kernel
    // the arguments travel through device memory (k_flow takes a pointer): slot 0 for this launch, slot 1 for the small instance behind the chain.  The host
    // copies live in the workspace (an asynchronous copy from pageable memory is staged before it returns, but nothing here relies on it)
    static_assert(sizeof(FlowArgs) <= 512, "Spd::flow_args slots");
    if ((rc = w.flow_args.ensure(1024))) return rc;
    if (w.flow_args_h.size() != 2048) w.flow_args_h.assign(2048, (char)0xff);       // [0, 1024): what the device copies hold; [1024, 2048): scratch for the comparison
    // (the same buffers, plan and switches evaluation after evaluation: the copy is made when something changed, not per evaluation -- two copies from
    // pageable memory were 0.13 ms of host time per step)
    auto upload = [&](size_t slot, const FlowArgs& a, hipStream_t st) -> int {
        char* held = w.flow_args_h.data() + slot;
        char* fresh_ = w.flow_args_h.data() + 1024 + slot;
        std::memset(fresh_, 0, 512);
        std::memcpy(fresh_, &a, sizeof(a));
        if (std::memcmp(held, fresh_, 512) == 0) return 0;
        HIP_TRY(hipStreamSynchronize(st));                   // (a kernel of an earlier evaluation may still read the old copy)
        HIP_TRY(dev_upload(w.flow_args.p + slot, fresh_, 512));
        std::memcpy(held, fresh_, 512);
        return 0;
    };
    if ((rc = upload(0, g, bulk))) return rc;
    const FlowArgs* d_g = reinterpret_cast<const FlowArgs*>(w.flow_args.p);
    FlowHot hot{};
    hot.bA = g.bA; hot.bL = g.bL; hot.bWt = g.bWt; hot.bWm = g.bWm; hot.bB = g.bB; hot.ld = g.ld; hot.tasks = g.tasks; hot.qmeta = g.qmeta; hot.flags = g.flags;
    hot.nq = g.nq; hot.ncas = g.ncas; hot.base_heads = g.base_heads; hot.base_err = g.base_err; hot.nap_max = g.nap_max; hot.nap_calm = g.nap_calm;
    if (dbg_kernel) hipLaunchKernelGGL(k_flow<true>, dim3(wg_per_cu * cus), dim3(FL_NT), FL_LDS_BYTES, bulk, d_g, hot);
    else hipLaunchKernelGGL(k_flow<false>, dim3(wg_per_cu * cus), dim3(FL_NT), FL_LDS_BYTES, bulk, d_g, hot);
    HIP_TRY(hipGetLastError());
    if (pe1) HIP_TRY(hipEventRecord(pe1, bulk));
    HIP_TRY(hipEventRecord(e_flow, bulk));
    m->gemm_launches += 1;
    m->gemm_flops += p.flops;

    unsigned* ferr = w.flow_flags.p + p.base_err;        // ONE error word for both kernels: whoever times out first stops the other
    for (int kb = 0; kb < nouter && !replay; ++kb) {
        const int k0 = kb * ob, k1 = std::min(k0 + ob, nb), nk = k1 - k0, na = std::min(ob, nb - k1);
        const FlowPlan::Chain& pc = p.chain[kb];
        ChainFlow cf{};
        cf.done_flag = w.flow_flags.p + pc.done_idx; cf.write_through = 1;       // (its diagonal block is complete in stream order: the update below)
        cf.trace = want_trace ? w.flow_trace.p + FLOW_TRACE_W * p.tasks.size() + 4 * (size_t)kb : nullptr;
        cf.diag = w.flow_diag.p;
        if ((rc = launch_chain(w.A.p, ld, k0, nk, w.invd.p, w.logdet.p, m->d_info.p, 0, w.Wm.p + (int64_t)k0 * MOGP_TILE * (ld + 1), ld,
                               w.chain_flags.p + (size_t)kb * MOGP_CHAIN_FLAGS, ferr, priv, &cf))) return rc;
        if (na <= 0) continue;
        // mini-panel  L[rows of block kb + 1][K] = A[rows][K] W_KK^T  on 64 x 128 tiles: waits for those rows' last updates from the dataflow
        // kernel, leaves with write-through stores and reports per row (the look-ahead tasks and the inverse's cycle read it)
        GemmArgs t1{};
        t1.A = w.A.p + (int64_t)k1 * MOGP_TILE * ld + (int64_t)k0 * MOGP_TILE; t1.lda = ld; t1.a_kmajor = 0;
        t1.B = w.Wm.p + (int64_t)k0 * MOGP_TILE * (ld + 1); t1.ldb = ld; t1.b_kmajor = 0;
        t1.C = w.Lm.p + (int64_t)k1 * MOGP_TILE * ld + (int64_t)k0 * MOGP_TILE; t1.ldc = ld; t1.alpha = 1.0; t1.beta = 0.0;
        t1.mode = GM_KHI_J; t1.small = 1; t1.mt = 2 * na; t1.nt = nk; t1.K = nk * MOGP_TILE;
        t1.fl_flags = w.flow_flags.p; t1.fl_nwait = pc.t1_nwait;
        for (int k = 0; k < pc.t1_nwait; ++k) { t1.fl_widx[k] = pc.t1_widx[k]; t1.fl_wval[k] = pc.t1_wval[k]; }
        static const unsigned hook_spins = std::getenv("MOGP_FLOW_HOOK_SPINS") ? (unsigned)std::atoll(std::getenv("MOGP_FLOW_HOOK_SPINS")) : 0u;
        t1.fl_spins = hook_spins; t1.fl_diag = w.flow_diag.p;
        t1.fl_wt = 1; t1.fl_sig = 1; t1.fl_sig_base = pc.t1_sig_base; t1.fl_sig_shift = 1; t1.fl_err = ferr; t1.sk_info = m->d_info.p;
        if ((rc = launch_gemm(t1, priv))) return rc;
        // (the first launch of this stream that touches A beyond its first 512 columns: the rest of the Gram matrix may still be on its way)
        if (kb == 0 && w.tail_ready) HIP_TRY(hipStreamWaitEvent(priv, w.tail_ready, 0));
        // next-diagonal update  D_{K+1,K+1} -= P P^T  on 64 x 64 tiles: the block's last update -- the earlier ones are dataflow tasks
        GemmArgs t2{};
        t2.A = t1.C; t2.lda = ld; t2.a_kmajor = 0; t2.B = t1.C; t2.ldb = ld; t2.b_kmajor = 0;
        t2.C = w.A.p + (int64_t)k1 * MOGP_TILE * (ld + 1); t2.ldc = ld; t2.alpha = -1.0; t2.beta = 1.0;
        t2.mode = GM_RECT_LOWER; t2.small = 2; t2.mt = 2 * na; t2.nt = 2 * na; t2.K = nk * MOGP_TILE;
        t2.fl_flags = w.flow_flags.p; t2.fl_nwait = pc.t2_wval ? 1 : 0; t2.fl_widx[0] = pc.t2_widx; t2.fl_wval[0] = pc.t2_wval;
        t2.fl_err = ferr; t2.sk_info = m->d_info.p; t2.fl_spins = hook_spins; t2.fl_diag = w.flow_diag.p;
        if ((rc = launch_gemm(t2, priv))) return rc;
        m->gemm_flops += gemm_flops(t1, nullptr) + gemm_flops(t2, nullptr);
    }
    // behind the last chain kernel the reserved CUs have nothing left to do while a quarter of the tile work is still queued: a second, small
    // instance of the dataflow kernel on the private stream takes tasks from the same queues until they are empty (MOGP_FLOW_TAIL=0: off)
    static const bool tail_on = !(std::getenv("MOGP_FLOW_TAIL") && std::atoi(std::getenv("MOGP_FLOW_TAIL")) == 0);
    if (tail_on && m->ctx->ncu_reserved > 0 && !replay) {
        if (rhs && rhs->ready) HIP_TRY(hipStreamWaitEvent(priv, rhs->ready, 0));
        g.post_base = wg_per_cu * cus;
        if ((rc = upload(512, g, priv))) return rc;
        const FlowArgs* d_g2 = reinterpret_cast<const FlowArgs*>(w.flow_args.p + 512);
        if (dbg_kernel) hipLaunchKernelGGL(k_flow<true>, dim3(wg_per_cu * m->ctx->ncu_reserved), dim3(FL_NT), FL_LDS_BYTES, priv, d_g2, hot);
        else hipLaunchKernelGGL(k_flow<false>, dim3(wg_per_cu * m->ctx->ncu_reserved), dim3(FL_NT), FL_LDS_BYTES, priv, d_g2, hot);
        HIP_TRY(hipGetLastError());
    }
    HIP_TRY(hipEventRecord(e_chain, priv));
    HIP_TRY(hipStreamWaitEvent(crit, e_flow, 0));
    HIP_TRY(hipStreamWaitEvent(crit, e_chain, 0));
    w.fused_last_inv = nullptr; w.fused_last_wt = nullptr;
    w.flow_used = true;
    m->flow_ran = true;
    m->flow_enqueue_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - host_t0).count();
    m->flow_enqueue_us_max = std::max(m->flow_enqueue_us_max, m->flow_enqueue_us);
    { const char* e = std::getenv("MOGP_FLOW_DEBUG");
      if (e && std::atoi(e) == 2 && &w == &m->k) { usleep(60000); fprintf(stderr, "mogp: dataflow state 60 ms after the evaluation was enqueued\n"); flow_debug_dump(m); } }
    return 0;
}

}  // namespace mogp

// the time stamps of the last dataflow evaluation of the exact system (MOGP_FLOW_TRACE=1): [FLOW_TRACE_W ntasks] (see FlowArgs::trace)
// in the order of mogp_flow_plan's task rows, then [4 nouter] chain kernel launch, wait over, end, spare
extern "C" int mogp_flow_trace(mogp_model* m, int64_t* out, int64_t cap, int64_t* count) {
    if (!m || !count) return fail(MOGP_EINVAL, "mogp_flow_trace: null argument");
    int rc;
    if ((rc = use_device(m->ctx))) return rc;
    const Spd& w = m->k;
    *count = 0;
    if (!w.flow_used || !w.flow_trace.p || !w.flow_cur) return MOGP_OK;
    const int64_t n = FLOW_TRACE_W * (int64_t)w.flow_cur->tasks.size() + 4 * (int64_t)w.flow_cur->nouter;
    *count = n;
    if (!out) return MOGP_OK;
    if (cap < n) return fail(MOGP_EINVAL, "mogp_flow_trace: the output is too small");
    HIP_TRY(hipStreamSynchronize(m->st));
    HIP_TRY(hipMemcpy(out, w.flow_trace.p, (size_t)n * sizeof(int64_t), hipMemcpyDeviceToHost));
    return MOGP_OK;
}

// ---- the plan as numbers: tests replay it on the CPU (no device work) ----------------------------------------------------------------
extern "C" int mogp_flow_plan_rhs(int nb, int rhs_nt, int64_t* out, int64_t cap, int64_t* count);
extern "C" int mogp_flow_plan(int nb, int64_t* out, int64_t cap, int64_t* count) { return mogp_flow_plan_rhs(nb, 0, out, cap, count); }
// rhs_nt > 0: the plan of factorisation + forward substitution of rhs_nt tile rows of right-hand sides (the prediction; T in buffer 2, X in buffer 4)
extern "C" int mogp_flow_plan_rhs(int nb, int rhs_nt, int64_t* out, int64_t cap, int64_t* count) {
    if (nb <= 0 || nb > 4096 || rhs_nt < 0 || rhs_nt > 4096 || !count) return fail(MOGP_EINVAL, "mogp_flow_plan: bad argument");
    FlowPlan p;
    flow_build(nb, 4, p, rhs_nt);
    int nlaunch = 0;
    for (int b = 0; b < p.nouter; ++b) nlaunch += (std::min(nb, b * 4 + 4) < nb) ? 3 : 1;
    const int64_t W = 24, rows = (int64_t)p.tasks.size() + nlaunch;
    *count = rows;
    if (!out) return MOGP_OK;
    if (cap < rows * W) return fail(MOGP_EINVAL, "mogp_flow_plan: the output holds fewer than 24 * count numbers");
    int64_t* o = out;
    for (int b = 0; b < p.nouter; ++b) {                  // the private stream, in its order: chain kernel (-1), mini-panel (-2), next-diagonal update (-3)
        const int k0 = b * 4, k1 = std::min(nb, k0 + 4), na = std::min(4, nb - k1);
        const FlowPlan::Chain& c = p.chain[b];
        std::fill(o, o + W, 0);
        o[0] = -1; o[1] = (int64_t)b * FLOW_KEY_STEP + PH_CHAIN; o[2] = b; o[3] = k0; o[4] = k1 - k0; o[22] = c.done_idx; o[23] = c.expect;
        o += W;
        if (na <= 0) continue;
        std::fill(o, o + W, 0);
        o[0] = -2; o[1] = (int64_t)b * FLOW_KEY_STEP + PH_PANEL; o[2] = b; o[3] = k0; o[4] = k1 - k0; o[5] = k1; o[6] = na; o[13] = c.t1_nwait;
        for (int d = 0; d < c.t1_nwait; ++d) { o[14 + d] = c.t1_widx[d]; o[18 + d] = c.t1_wval[d]; }
        o[22] = c.t1_sig_base; o[23] = c.t1_sig_per_row;
        o += W;
        std::fill(o, o + W, 0);
        o[0] = -3; o[1] = (int64_t)b * FLOW_KEY_STEP + PH_UPDATE; o[2] = b; o[3] = k0; o[4] = k1 - k0; o[5] = k1; o[6] = na; o[13] = c.t2_wval ? 1 : 0;
        o[14] = c.t2_widx; o[18] = c.t2_wval; o[22] = -1;
        o += W;
    }
    for (int q = 0; q < p.nq; ++q)
        for (int k = 0; k < p.qsize[q]; ++k, o += W) {
            const FlowTask& t = p.tasks[p.qbase[q] + k];
            o[0] = q; o[1] = t.key; o[2] = t.abuf; o[3] = t.ar; o[4] = t.ac; o[5] = t.bbuf; o[6] = t.br; o[7] = t.bc;
            o[8] = t.cbuf; o[9] = t.cr; o[10] = t.cc; o[11] = t.var; o[12] = t.kt; o[13] = t.ndep;
            for (int d = 0; d < 4; ++d) { o[14 + d] = d < t.ndep ? t.dep[d] : 0; o[18 + d] = d < t.ndep ? t.need[d] : 0; }
            o[22] = t.sig[0] == FLOW_NOSIG ? -1 : (int64_t)t.sig[0]; o[23] = t.sig[1] == FLOW_NOSIG ? -1 : (int64_t)t.sig[1];
        }
    return MOGP_OK;
}
