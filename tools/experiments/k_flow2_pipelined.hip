// NOT BUILT.  Round 5's experiment, kept as a record (results: profiles/r5_flow_pipe_ab.txt): the dataflow kernel of
// mogptk_amd/csrc/flow.hip software-pipelined across tasks.  It is bit-for-bit correct (the GPU parity suite passed with it as the default) and
// SLOWER than k_flow in every variant measured, so the library does not carry it.  To try it again: paste this block in front of
// "---- the task graph (host)" in flow.hip and select it where k_flow is launched (it takes the same FlowArgs plus `fast`, `refill_from`).
// ---- round 5: the same kernel, software-pipelined ACROSS tasks (k_flow2; MOGP_FLOW_PIPE=0 gives k_flow back) ---------------------------------
// Round 4's time stamps: around a 98 us k loop a task spent 13.0 us looking for work + 5.5 us from claim to k loop + 5.2 us storing, draining and
// signalling -- three dependent memory round trips of the look, one for the descriptor, the drain of 128 KB of write-through stores, all with
// seven of the eight waves parked at a barrier.  Here:
//   * wave 0's lanes keep the DESCRIPTOR of the task they hold (or, the compare-and-swap queue's lanes: of a window of eight tasks by absolute
//     index) in an LDS slot of their own, filled when the task is claimed; a look is one round trip (the counters), the chosen descriptor goes to
//     the other waves through LDS, not through memory;
//   * that one round trip is requested BEFORE the tile's stores go out (loads and stores share vmcnt and retire in order: the answer is back
//     while the stores drain) and evaluated right behind them: if something wave 0 holds is ready the workgroup is in its next task ~2 us
//     after its last MFMA;
//   * the stores of task n drain UNDER the first loads of task n + 1 (C tile, first operand blocks): every wave waits for its own vmcnt(0)
//     where it needs those loads anyway, and task n's counters are bumped behind the k loop's first LDS barrier;
//   * only if nothing held is ready (or the critical queue's window has run away) the workgroup drains, signals and looks the long way
//     (refresh the window, claim, nap) exactly as k_flow does -- a workgroup never sleeps on a signal it still owes.
// Same tile bodies, same k order: results bit for bit those of k_flow.
#define FL2_SLOT_INTS 16
// vmcnt(0) through the builtin, not inline asm: the compiler's wait-count pass reads it and knows nothing is in flight behind it (an asm wait is
// invisible to it -- it then keeps "pending" stores on its books across the loop's back edges and answers with vmcnt(0) at the next merge)
#define FL_VMCNT0() __builtin_amdgcn_s_waitcnt(0x0F70)        // gfx9 encoding: vmcnt = 0, expcnt = 7, lgkmcnt = 15 (not waited for)
#define FL2_LDS_BYTES (FL_LDS_DOUBLES * 8 + 16 + FL2_SLOT_INTS * 4 + 64 * FL2_SLOT_INTS * 4)     // tile operands, pick word, current descriptor, 64 slots
static_assert(sizeof(FlowTask) == FL2_SLOT_INTS * 4, "a slot holds one descriptor");
static_assert(2 * FL2_LDS_BYTES <= 160 * 1024, "two workgroups per CU");

// byte address inside the workgroup's LDS of a pointer into a __shared__ array (what M0 takes for an LDS-DMA load)
__device__ __forceinline__ unsigned lds_address(const void* p) {
    return (unsigned)(unsigned long long)(__attribute__((address_space(3))) const char*)p;
}

struct FlowOwed {                      // what a workgroup still owes for the tile whose stores are draining
    unsigned s0, s1;                   // its counters (FLOW_NOSIG: none)
    unsigned long long* tr;            // its trace row (thread 0)
};
__device__ __forceinline__ void flow_pay(const FlowArgs& g, FlowOwed& o) {      // thread 0, behind a barrier that follows every wave's vmcnt(0)
    if (o.s0 != FLOW_NOSIG) __hip_atomic_fetch_add(g.flags + o.s0, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (o.s1 != FLOW_NOSIG) __hip_atomic_fetch_add(g.flags + o.s1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (o.tr) o.tr[4] = wall_clock64();
}

// flow_tile without its epilogue: the accumulators stay in registers (the caller stores them: flow_store), and the PREVIOUS tile's counters are
// bumped here, behind the first LDS barrier -- every wave has waited for vmcnt(0) (its own stores of that tile included) to fill the first block
template <int AKM, int BKM, typename Hook>
__device__ __forceinline__ void flow_compute(const double* Ap, const double* Bp, const double* Cp, const int64_t ld, const int kt, const bool fresh,
                                             const double alpha, double* gemm_lds, d4_t (&acc)[FL_WTM][FL_WTN], const FlowArgs& g, FlowOwed& owed,
                                             unsigned long long* tr, Hook&& after_first_wait) {
    constexpr int WTM = FL_WTM, WTN = FL_WTN, NWJ = FL_NWJ, NWI = FL_NWI, NT = FL_NT;
    constexpr int TMR = MOGP_TILE, TNC = MOGP_TILE, COLK_A = FL_COLK, COLK_B = FL_COLK, OPER_A = FL_OPER, OPER_B = FL_OPER;
    constexpr int EPT_A = TMR * FL_BK / NT, EPT_B = TNC * FL_BK / NT;
    constexpr int NQ_A = EPT_A / 2, NQ_B = EPT_B / 2, TPR_A = FL_BK / EPT_A, TPR_B = FL_BK / EPT_B, TPK = NT / FL_BK;
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

    if (!fresh) {
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
    load_block(0);
    FL_VMCNT0();     // the first block is here -- and so is everything this wave had in flight: the previous tile's stores
    after_first_wait();                                  // (the replacement claim issued at the top of the task is back as well: its descriptor's copy starts here)
    write_block(0);
    load_block(min(1, kt - 1));
    FL_LDS_BARRIER();
    if (threadIdx.x == 0) flow_pay(g, owed);             // every wave has passed its vmcnt(0): the previous tile is in memory
    owed.s0 = owed.s1 = FLOW_NOSIG; owed.tr = nullptr;
    if (tr) tr[2] = wall_clock64();
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
    // the last iterations' operand prefetches were never used but are still "in flight" as far as the compiler's counter model goes: retire them
    // here (they were issued a k block ago: no wait in practice), or the first reuse of their registers puts a vmcnt(0) BEHIND the requests below
    FL_VMCNT0();
    if (tr) tr[3] = wall_clock64();
}
// The tile's 32 write-through stores per lane as inline asm: hipcc's wait-count pass does not see them.  That is the point -- with stores it can
// see still in flight at the task loop's back edge it answers every merge with s_waitcnt vmcnt(0) (measured in the ISA: at the top of the loop,
// in front of the look), which is exactly the drain this kernel wants to overlap.  The waits that matter are placed by hand: vmcnt(32) behind the
// stores for the five requests issued in front of them (flow_requests_wait), vmcnt(0) through the builtin where the next tile needs its first
// operand block (flow_compute) -- the hardware counter holds these stores whether hipcc knows of them or not.
__device__ __forceinline__ void flow_store(double* Cp, const int64_t ld, const double alpha, const d4_t (&acc)[FL_WTM][FL_WTN]) {
    int tl = threadIdx.x;
    asm volatile("" : "+v"(tl));
    const int ln = tl & 63, wv = tl >> 6, wi = wv / FL_NWJ, wj = wv % FL_NWJ;
    const int crow = wi * (MOGP_TILE / FL_NWI) + (ln >> 4), ccol = wj * (MOGP_TILE / FL_NWJ) + (ln & 15);
    static_assert(FL_WTN == 2, "the second column tile is the first one's address + 128 bytes");
#pragma unroll
    for (int m = 0; m < FL_WTM; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            double* q = Cp + (int64_t)(crow + m * 16 + 4 * r) * ld + ccol;
            const double x0 = alpha * acc[m][0][r], x1 = alpha * acc[m][1][r];
            asm volatile("global_store_dwordx2 %0, %1, off sc1\n\tglobal_store_dwordx2 %0, %2, off offset:128 sc1\n\ts_nop 0"
                         :: "v"(q), "v"(x0), "v"(x1) : "memory");
        }
}
#define FL2_STORES_PER_LANE (FL_WTM * FL_WTN * 4)
static_assert(FL2_STORES_PER_LANE == 32, "flow_requests_wait counts the stores by hand");

__global__ __launch_bounds__(FL_NT, 4) void k_flow2(FlowArgs g) {
    extern __shared__ __attribute__((aligned(16))) double gemm_lds[];
    int* pick = reinterpret_cast<int*>(gemm_lds + FL_LDS_DOUBLES);     // [0] the task (or -2 done, -3 error, -4 "nothing at hand: drain and signal first"), [1] the lane that held it
    int* cur = pick + 4;                                               // [16]: the descriptor of the task about to run (copied out of its lane's slot, which may be refilled at once)
    int* slots = cur + FL2_SLOT_INTS;                                  // [64][16]: wave 0's lanes, one descriptor each
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    unsigned* heads = g.flags + g.base_heads;
    unsigned* err = g.flags + g.base_err;
    // wave 0's lane roles (as in k_flow): FL_LA lanes per compare-and-swap queue, one lane per fetch-add queue
    const int ncl = g.ncas * FL_LA, nlanes = ncl + (g.nq - g.ncas);
    const bool is_cas = lane < ncl, is_eager = lane >= ncl && lane < nlanes;
    const int myq = is_cas ? lane / FL_LA : (is_eager ? g.ncas + (lane - ncl) : 0), myk = is_cas ? lane % FL_LA : 0, lb = lane - myk;
    int qbase = 0, qsize = 0;
    if (wave == 0 && lane < nlanes) { qbase = g.qmeta[2 * myq]; qsize = g.qmeta[2 * myq + 1]; }
    int* myslot = slots + FL2_SLOT_INTS * lane;
    int pend = -1;                    // fetch-add lanes: the index (inside the queue) this workgroup holds; its descriptor is in the lane's slot
    bool exhausted = false;
    int c0 = -1;                      // compare-and-swap lanes: the slots of the queue's lanes hold the descriptors of tasks c0 .. c0 + FL_LA - 1 (-1: nothing)
    const unsigned long long eager_mask = nlanes >= 64 ? ~0ull << ncl : ((1ull << nlanes) - 1ull) & ~((1ull << ncl) - 1ull);
    unsigned idle = 0, naps = 0;
    FlowOwed owed{FLOW_NOSIG, FLOW_NOSIG, nullptr};
    unsigned long long t_look = 0;

    auto fill_slot = [&](int idx) {                   // descriptor of task idx of my queue -> my slot (64 bytes, static data: plain loads)
        const int4* s4 = reinterpret_cast<const int4*>(g.tasks + qbase + idx);
        const int4 a = s4[0], b = s4[1], c = s4[2], d = s4[3];
        int4* d4 = reinterpret_cast<int4*>(myslot);
        d4[0] = a; d4[1] = b; d4[2] = c; d4[3] = d;
    };
    // FlowTask as ints: [4] = kt | ndep << 16, [5 .. 8] = dep, [9 .. 10] = need (two 16-bit values each; 0 for an absent dependency: any value passes)
    auto satisfied = [&](unsigned v0, unsigned v1, unsigned v2, unsigned v3) {
        const unsigned n01 = (unsigned)myslot[9], n23 = (unsigned)myslot[10];
        return v0 >= (n01 & 0xffffu) && v1 >= (n01 >> 16) && v2 >= (n23 & 0xffffu) && v3 >= (n23 >> 16);
    };
    // the winner of a look: the lane `wl` takes its task (compare-and-swap lanes: the head, and what the head has become if others were faster and
    // this look saw that task ready too); -> ok, res (task index), src (lane whose slot holds the descriptor)
    auto take = [&](int wl, int h, unsigned long long mready, int& res, int& src) {
        int ok = 0;
        if (lane == wl) {
            if (is_cas) {
                unsigned cur = (unsigned)h;
                for (;;) {
                    unsigned expect = cur;
                    if (__hip_atomic_compare_exchange_strong(heads + myq, &expect, cur + 1u, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                                             __HIP_MEMORY_SCOPE_AGENT)) { ok = 1; res = qbase + (int)cur; src = lane + (int)(cur - (unsigned)h); break; }
                    cur = expect;
                    const unsigned off = cur - (unsigned)h;
                    if (myk + off >= FL_LA || !((mready >> (lane + off)) & 1ull)) break;
                }
            } else {
                ok = 1; res = qbase + pend; src = lane; pend = -1;
            }
        }
        ok = __shfl(ok, wl, 64);
        res = __shfl(res, wl, 64);
        src = __shfl(src, wl, 64);
        return ok;
    };

    for (;;) {
        // ================= the long way: nothing is owed, nothing was asked for ==========================================================
        if (g.trace && tid == 0) t_look = wall_clock64();
        if (wave == 0) {
            int res = -1, src = 0;
            unsigned nap = 0;
            naps = 0;
            for (;;) {
                int h = 0;
                bool have = false, ready = false;
                if (is_cas) {
                    h = (int)__hip_atomic_load(heads + myq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    h = __shfl(h, lb, 64);                                     // one head value per queue
                    if (h < qsize && !(c0 >= 0 && h >= c0 && h < c0 + FL_LA)) {   // the window has run away (or there is none yet): the next eight from the head
                        c0 = h;
                        if (c0 + myk < qsize) fill_slot(c0 + myk);
                    }
                    have = c0 >= 0 && h >= c0 && h < c0 + FL_LA && c0 + myk >= h && c0 + myk < qsize;
                } else if (is_eager) {
                    have = pend >= 0;
                }
                if (have) {
                    const int nd = myslot[4] >> 16;
                    unsigned v0 = 0u, v1 = 0u, v2 = 0u, v3 = 0u;
                    if (nd > 0) v0 = __hip_atomic_load(g.flags + (unsigned)myslot[5], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (nd > 1) v1 = __hip_atomic_load(g.flags + (unsigned)myslot[6], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (nd > 2) v2 = __hip_atomic_load(g.flags + (unsigned)myslot[7], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (nd > 3) v3 = __hip_atomic_load(g.flags + (unsigned)myslot[8], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    ready = satisfied(v0, v1, v2, v3);
                }
                const unsigned long long mready = __ballot(ready);
                // candidates in priority order: the head of every compare-and-swap queue (the lane that holds task h), then the fetch-add lanes
                const bool is_head = is_cas && have && c0 + myk == h;
                const unsigned long long cand = mready & (__ballot(is_head) | eager_mask);
                if (cand) {
                    if (take(__ffsll((long long)cand) - 1, h, mready, res, src)) break;
                    continue;                                                  // the head ran away: look again
                }
                // nothing this workgroup holds or may take is ready: take what can be taken eagerly
                bool took = false;
                bool want = is_eager && pend < 0 && !exhausted;
                if (g.claim_one) {
                    const unsigned long long mw = __ballot(want);
                    want = want && mw && lane == __ffsll((long long)mw) - 1;
                }
                if (want) {
                    const unsigned hh = __hip_atomic_fetch_add(heads + myq, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (hh < (unsigned)qsize) { pend = (int)hh; took = true; fill_slot(pend); } else exhausted = true;
                }
                const bool open = is_cas ? (myk == 0 && h < qsize) : (is_eager && (pend >= 0 || !exhausted));
                if (!__ballot(open)) { res = -2; break; }                      // every queue is empty and nothing is held: done
                if (__ballot(took)) continue;
                nap = nap < g.nap_max ? nap + 1u : g.nap_max;
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
            if (res >= 0 && lane < FL2_SLOT_INTS) cur[lane] = slots[FL2_SLOT_INTS * src + lane];
            if (lane == 0) {
                pick[0] = res; pick[1] = src;
                if (res >= 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");      // ONE buffer_inv sc1 behind the satisfied counters (nothing of this wave's is in flight here)
            }
        }
        __syncthreads();
        int ti = __builtin_amdgcn_readfirstlane(pick[0]);
        if (ti < 0) {
            if (ti == -3 && tid == 0) atomicMin(g.info, (unsigned long long)MOGP_INFO_CHAIN_TIMEOUT);
            break;
        }
        // ================= a chain of tasks: each one's stores drain under the next one's first loads =====================================
        for (;;) {
            idle = 0;
            const int* sd = cur;
            const unsigned w0 = (unsigned)__builtin_amdgcn_readfirstlane(sd[0]), w1 = (unsigned)__builtin_amdgcn_readfirstlane(sd[1]),
                           w2 = (unsigned)__builtin_amdgcn_readfirstlane(sd[2]), w3 = (unsigned)__builtin_amdgcn_readfirstlane(sd[3]),
                           w4 = (unsigned)__builtin_amdgcn_readfirstlane(sd[4]);
            const unsigned sg0 = (unsigned)__builtin_amdgcn_readfirstlane(sd[11]), sg1 = (unsigned)__builtin_amdgcn_readfirstlane(sd[12]);
            const int t_ar = (int)(w0 & 0xffffu), t_ac = (int)(w0 >> 16), t_br = (int)(w1 & 0xffffu), t_bc = (int)(w1 >> 16), t_cr = (int)(w2 & 0xffffu), t_cc = (int)(w2 >> 16);
            const int ab = (int)(w3 & 0xffu), bb = (int)((w3 >> 8) & 0xffu), cb = (int)((w3 >> 16) & 0xffu), var = (int)(w3 >> 24), kt = (int)(w4 & 0xffffu);
            const double* Ab = ab == 0 ? g.bA : ab == 1 ? g.bL : ab == 2 ? g.bWt : ab == 3 ? g.bWm : g.bB;
            const double* Bb = bb == 0 ? g.bA : bb == 1 ? g.bL : bb == 2 ? g.bWt : bb == 3 ? g.bWm : g.bB;
            double* Cb = cb == 0 ? g.bA : cb == 1 ? g.bL : cb == 2 ? g.bWt : cb == 3 ? g.bWm : g.bB;
            const double* Ap = Ab + ((int64_t)t_ar * g.ld + t_ac) * MOGP_TILE;
            const double* Bp = Bb + ((int64_t)t_br * g.ld + t_bc) * MOGP_TILE;
            double* Cp = Cb + ((int64_t)t_cr * g.ld + t_cc) * MOGP_TILE;
            const bool fresh = (var & 4) != 0;
            const double alpha = (var & 8) ? -1.0 : 1.0;
            unsigned long long* tr = (g.trace && tid == 0) ? g.trace + FLOW_TRACE_W * (size_t)ti : nullptr;
            unsigned long long t_claim = 0;
            if (tr) {
                t_claim = wall_clock64();
                tr[0] = t_look; tr[1] = t_claim;
                tr[5] = ((unsigned long long)naps << 32) | ((unsigned long long)__builtin_amdgcn_s_getreg((3 << 11) | 20) << 16) | blockIdx.x;
            }
            if (var & 32) {                                            // vector task (rare): as in k_flow, between two full drains
                FL_VMCNT0();
                __syncthreads();
                if (tid == 0) flow_pay(g, owed);
                if (g.vy) {
                    if (var & 1) flow_apart(g, t_ar * MOGP_TILE, kt * MOGP_TILE, t_br, t_ac, gemm_lds);
                    else flow_zrow(g, t_ar, gemm_lds);
                }
                if (tr) { tr[2] = t_claim; tr[3] = wall_clock64(); }
                owed.s0 = sg0; owed.s1 = sg1; owed.tr = tr;
                break;                                                 // -> drain, pay, the long way
            }
            // ---- a replacement for the task just taken (fetch-add queues from refill_from on: the backlog, where the next task is ready long
            // before anybody gets to it): claimed NOW, its descriptor copied into the lane's slot by an LDS-DMA load issued behind the tile's first
            // wait -- both round trips ride under the prologue's own.  At the end of this tile wave 0 then HOLDS a task whose counters it can ask for
            // in front of the stores; without it the short look found nothing (what a workgroup keeps holding are the tasks that are NOT ready)
            // and every task went the long way: 21 us between two k loops against round 4's 20.
            const int srcl = __builtin_amdgcn_readfirstlane(pick[1]);
            int hh = -1;
            const bool refill = wave == 0 && lane == srcl && is_eager && myq >= g.refill_from && !exhausted && pend < 0 &&
                                (ti - qbase) + 2 * (int)gridDim.x < qsize;          // (near the end of a queue a task held by a busy workgroup is a task an idle one cannot take)
            if (refill) hh = (int)__hip_atomic_fetch_add(heads + myq, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            auto after_first_wait = [&]() {
                if (wave != 0) return;
                if (refill) { if (hh < qsize) pend = hh; else { exhausted = true; hh = -1; } }
                const int hs = __shfl(hh, srcl, 64), qb = __shfl(qbase, srcl, 64);
                if (hs >= 0 && lane < 4) {                          // 4 lanes x 16 bytes: LDS destination = M0 + lane * 16 (wave-uniform base), source per lane
                    const char* gsrc = reinterpret_cast<const char*>(g.tasks + qb + hs) + 16 * lane;
                    const unsigned ldst = lds_address(slots + FL2_SLOT_INTS * srcl);
                    unsigned keep;
                    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                                 : "=&s"(keep) : "v"(gsrc), "s"(ldst) : "memory");
                }
            };
            d4_t acc[FL_WTM][FL_WTN];
            if (var & 16) __builtin_amdgcn_s_setprio(2);
            switch (var & 3) {
                case 0: flow_compute<0, 0>(Ap, Bp, Cp, g.ld, kt, fresh, alpha, gemm_lds, acc, g, owed, tr, after_first_wait); break;
                case 1: flow_compute<0, 1>(Ap, Bp, Cp, g.ld, kt, fresh, alpha, gemm_lds, acc, g, owed, tr, after_first_wait); break;
                default: flow_compute<1, 1>(Ap, Bp, Cp, g.ld, kt, fresh, alpha, gemm_lds, acc, g, owed, tr, after_first_wait); break;
            }
            if (var & 16) __builtin_amdgcn_s_setprio(0);
            owed.s0 = sg0; owed.s1 = sg1; owed.tr = tr;
            if (!g.fast) {                                             // (measurement switch: the long way after every tile)
                flow_store(Cp, g.ld, alpha, acc);
                break;
            }
            // ---- wave 0 asks for the counters of what it holds NOW, in front of the stores (loads and stores retire in order through vmcnt: the
            // answers are back long before the stores have drained).  Branch-free, executed by every wave (lanes with nothing to ask read the
            // error word), and in inline asm together with the stores: see flow_store.
            if (g.trace && tid == 0) t_look = wall_clock64();
            FL_VMCNT0();           // (spill reloads of the look's state: cache hits)
            unsigned pre_h, pv0, pv1, pv2, pv3;
            const bool asked = wave == 0 && (is_cas ? (c0 >= 0 && c0 + myk < qsize) : (is_eager && pend >= 0));
            {
                const unsigned safe = (unsigned)g.base_err;
                const int nd = asked ? (myslot[4] >> 16) : 0;
                const unsigned* a0 = g.flags + (nd > 0 ? (unsigned)myslot[5] : safe);
                const unsigned* a1 = g.flags + (nd > 1 ? (unsigned)myslot[6] : safe);
                const unsigned* a2 = g.flags + (nd > 2 ? (unsigned)myslot[7] : safe);
                const unsigned* a3 = g.flags + (nd > 3 ? (unsigned)myslot[8] : safe);
                const unsigned* ah = g.flags + ((wave == 0 && is_cas) ? (unsigned)(g.base_heads + myq) : safe);
                asm volatile("global_load_dword %0, %5, off sc1\n\tglobal_load_dword %1, %6, off sc1\n\tglobal_load_dword %2, %7, off sc1\n\t"
                             "global_load_dword %3, %8, off sc1\n\tglobal_load_dword %4, %9, off sc1"
                             : "=&v"(pre_h), "=&v"(pv0), "=&v"(pv1), "=&v"(pv2), "=&v"(pv3) : "v"(ah), "v"(a0), "v"(a1), "v"(a2), "v"(a3) : "memory");
            }
            flow_store(Cp, g.ld, alpha, acc);
            // the five answers are older than the 32 stores behind them: "at most 32 outstanding" means they are back, whatever the stores are doing
            asm volatile("s_waitcnt vmcnt(32)" : "+v"(pre_h), "+v"(pv0), "+v"(pv1), "+v"(pv2), "+v"(pv3) :: "memory");
            // ---- the short look: ONE pass over those answers -- no refresh, no claim, no nap: the workgroup still owes this tile's signal
            if (wave == 0) {
                int res = -4, src = 0, h = 0;
                bool have = false;
                if (is_cas) {
                    h = __shfl((int)pre_h, lb, 64);
                    have = asked && h >= c0 && h < c0 + FL_LA && c0 + myk >= h;
                } else if (is_eager) {
                    have = asked;
                }
                // a critical queue whose window has run away is unknown territory: the long way refreshes it (and keeps its priority)
                const bool lost = is_cas && myk == 0 && h < qsize && !(c0 >= 0 && h >= c0 && h < c0 + FL_LA);
                const bool ready = have && satisfied(pv0, pv1, pv2, pv3);
                const unsigned long long mlost = __ballot(lost), mready = __ballot(ready);
                const bool is_head = is_cas && have && c0 + myk == h;
                const unsigned long long cand = mready & (__ballot(is_head) | eager_mask);
                if (!mlost && cand) {
                    if (!take(__ffsll((long long)cand) - 1, h, mready, res, src)) res = -4;
                }
                if (res >= 0 && lane < FL2_SLOT_INTS) cur[lane] = slots[FL2_SLOT_INTS * src + lane];
                if (lane == 0) {
                    pick[0] = res; pick[1] = src;
                    // the agent acquire behind the satisfied counters WITHOUT the s_waitcnt vmcnt(0) the fence builtin puts in front of it: the
                    // counters were consumed above (they are back); what is still in flight are this wave's stores of the tile just finished
                    if (res >= 0) asm volatile("buffer_inv sc1" ::: "memory");
                }
                naps = 0;
            }
            FL_LDS_BARRIER();          // (not __syncthreads: the stores may still be on their way; the tile's last LDS reads are behind every wave)
            ti = __builtin_amdgcn_readfirstlane(pick[0]);
            if (ti < 0) break;
        }
        // nothing at hand (or a vector task ran): drain this workgroup's stores, signal, and look the long way
        FL_VMCNT0();
        __syncthreads();
        if (tid == 0) flow_pay(g, owed);
        owed.s0 = owed.s1 = FLOW_NOSIG; owed.tr = nullptr;
    }
}

