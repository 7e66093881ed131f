// potri.hip -- Cholesky factorisation AND inverse of the SPD matrix in ONE schedule (the default gradient path).
//
// Why: at N = 8192 the blocked Cholesky is bound by its serial chain (64 leaf factorisations and the small kernels between
// them), not by flops: it leaves three quarters of the chip idle, and TRTRI + LAUUM (2/3 of all flops) used to run afterwards.
// Here the inverse is streamed BEHIND the chain as rank-512 updates, and the chain itself is kept off the CUs the bulk work uses.
//
// Outer blocks K of 4 tiles (Kd = 512 columns).  Per block, on four streams:
//   priv  (CU mask = the reserved CUs only)   the whole critical path, in ONE stream:
//            intra-block chain on the 512 x 512 diagonal block D_KK:
//              4 x [ leaf (factor + inverse of a 128 tile) -> panel inside the block -> update inside the block ], then W_KK = L_KK^-1
//              in one launch (wkk.hip: 32 workgroups, each a 16-column strip of one column block; out of place)
//            mini-panel            P[K+1 rows] = A[K+1 rows, K] W_KK^T       (the next block's rows only)
//            next-diagonal update  D_{K+1,K+1} -= P[K+1] P[K+1]^T            -> the chain of block K+1 follows in the same stream
//            Every launch is <= 64 small workgroups; alone on their CUs they run at their unloaded latency.
//   crit  (all CUs, high priority)            rest of the panel     P[> K+1]      = A[> K+1, K] W_KK^T
//                                             next-block columns    A[> K+1, K+1] -= P[> K+1] P[K+1]^T
//   bulk  (all but the reserved CUs)          A[> K+1, > K+1] -= P[> K+1] P[> K+1]^T   (trailing update, K = 512; block K+2's columns first)
//   inv   (all but the reserved CUs)          W = L^-1 by elementary block-column inverses, and the inverse itself:
//            W[K, <K]   = W_KK Wt[K, <K]                            finalise the row block (Wt = running product, in w.Wm)
//            Wt[>K, K]  = -P W_KK ;  Wt[>K, <K] -= P W[K, <K]       rank-512 update of all rows below
//            Kinv[<=K, <=K] += W[K, <=K]^T W[K, <=K]                rank-512 update of the inverse (w.B)
// The panel P = L[>K, K] is consumed only by block K's own updates, so it lives in MOGP_NPANEL rotating Npad x 512 buffers, not in w.A.
// Critical path per block = intra-block chain + mini-panel + next-diagonal update; everything of N-proportional size is off it.  On return: w.Wm = W = L^-1 (lower), w.B = (L L^T)^-1 (lower tiles, full diagonal tiles),
// w.logdet / w.invd / the pivot check as in spd_potrf; w.A is consumed (Schur data; its diagonal blocks keep L_KK).
#include "mogp_model.h"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>

using namespace mogp;

namespace mogp {
int launch_potrf_trtri_tile(double* A, int64_t ld, int t, double* invd, double* logdet, unsigned long long* info, hipStream_t s,
                            long long info_base = 0, int store_L = 0);
}

#define RC(x) do { int r__ = (x); if (r__) return r__; } while (0)
#define FZ_OB 4
#define FZ_KD (FZ_OB * MOGP_TILE)

namespace {

enum { EV_BLK = 0, EV_DIAG = 1, EV_PANEL = 2, EV_BULK = 3, EV_INV = 4, EV_REST = 5, EV_WROW = 6, EV_ACC = 7, EV_PER_BLOCK = 8 };

// factor the diagonal block [k0, k1) of w.A in place (L_KK) and put W_KK = L_KK^-1 (lower; zeros above) into Wk (leading dimension FZ_KD)
int intra_block_chain(mogp_model* m, Spd& w, double* Wk, int k0, int k1, hipStream_t q, int kb) {
    const int64_t ld = w.Npad;
    // MOGP_CHAIN=0: the launch-per-step form below (4 leaves, 6 small GEMMs, k_wkk) instead of the persistent kernel of chain.hip
    if (chain_enabled(m)) {
        const int nouter = (w.nb + FZ_OB - 1) / FZ_OB;
        return launch_chain(w.A.p, ld, k0, k1 - k0, w.invd.p, w.logdet.p, m->d_info.p, 0, Wk, FZ_KD,
                            w.chain_flags.p + (size_t)kb * MOGP_CHAIN_FLAGS, w.chain_flags.p + (size_t)nouter * MOGP_CHAIN_FLAGS, q);
    }
    for (int k = k0; k < k1; ++k) {
        RC(launch_potrf_trtri_tile(w.A.p, ld, k, w.invd.p, w.logdet.p, m->d_info.p, q, 0));
        const int ri = k1 - k - 1;                       // tile rows below, inside the block
        if (ri <= 0) break;
        double* panel = w.A.p + (int64_t)(k + 1) * MOGP_TILE * ld + (int64_t)k * MOGP_TILE;
        GemmArgs g{};
        g.A = panel; g.lda = ld; g.a_kmajor = 0;
        g.B = w.invd.p + (int64_t)k * MOGP_TILE * MOGP_TILE; g.ldb = MOGP_TILE; g.b_kmajor = 0;
        g.C = panel; g.ldc = ld; g.alpha = 1.0; g.beta = 0.0;
        g.mode = GM_RECT; g.small = 1; g.mt = 2 * ri; g.nt = 1; g.K = MOGP_TILE;               // 64 x 128 tiles: in place
        RC(gemm_call(m, g, gemm_flops(g, nullptr), q));
        GemmArgs u{};
        u.A = panel; u.lda = ld; u.a_kmajor = 0; u.B = panel; u.ldb = ld; u.b_kmajor = 0;
        u.C = w.A.p + (int64_t)(k + 1) * MOGP_TILE * (ld + 1); u.ldc = ld; u.alpha = -1.0; u.beta = 1.0;
        u.mode = GM_RECT_LOWER; u.small = 2; u.mt = 2 * ri; u.nt = 2 * ri; u.K = MOGP_TILE;
        RC(gemm_call(m, u, gemm_flops(u, nullptr), q));
    }
    // W_KK = L_KK^-1 in one launch, out of place (wkk.hip)
    return launch_wkk(w.A.p + (int64_t)k0 * MOGP_TILE * (ld + 1), ld, w.invd.p + (int64_t)k0 * MOGP_TILE * MOGP_TILE, k1 - k0, Wk, FZ_KD, q);
}

// P[rows r0 .. r0+nr) = A[rows, K] * W_KK^T  (rows in tiles; P has leading dimension FZ_KD and is indexed by the global row)
int panel_rows(mogp_model* m, Spd& w, double* P, const double* Wk, int k0, int nk, int r0, int nr, bool small, hipStream_t q) {
    const int64_t ld = w.Npad;
    GemmArgs g{};
    g.A = w.A.p + (int64_t)r0 * MOGP_TILE * ld + (int64_t)k0 * MOGP_TILE; g.lda = ld; g.a_kmajor = 0;
    g.B = Wk; g.ldb = FZ_KD; g.b_kmajor = 0;                                              // W_KK as [j][k], k <= j
    g.C = P + (int64_t)r0 * MOGP_TILE * FZ_KD; g.ldc = FZ_KD; g.alpha = 1.0; g.beta = 0.0;
    g.mode = GM_KHI_J; g.small = small ? 1 : 0; g.mt = small ? 2 * nr : nr; g.nt = nk; g.K = nk * MOGP_TILE;
    return gemm_call(m, g, gemm_flops(g, nullptr), q);
}

// the inverse stream's share of block K (see the header); Lp = P[k1 tile row], leading dimension FZ_KD
int inverse_step(mogp_model* m, Spd& w, const double* Wkk, int k0, int k1, const double* Lp, hipStream_t q, hipStream_t qacc, hipEvent_t w_row,
                 hipEvent_t w_final) {
    const int64_t ld = w.Npad;
    const int nk = k1 - k0, rem = w.nb - k1;
    const int64_t Kd = (int64_t)nk * MOGP_TILE, c0 = (int64_t)k0 * MOGP_TILE;
    double* Wrow = w.Wm.p + c0 * ld;                 // W[K, 0]
    double* Brow = w.B.p + c0 * ld;                  // scratch now, Kinv[K, 0] afterwards
    if (k0 > 0) {                                    // finalise the row block through the scratch (not in place)
        GemmArgs g{};
        g.A = Wkk; g.lda = FZ_KD; g.a_kmajor = 0; g.B = Wrow; g.ldb = ld; g.b_kmajor = 1;
        g.C = Brow; g.ldc = ld; g.alpha = 1.0; g.beta = 0.0;
        g.mode = GM_KHI_I; g.small = 1; g.mt = 2 * nk; g.nt = k0; g.K = (int)Kd;
        RC(gemm_call(m, g, gemm_flops(g, nullptr), q));
        RC(launch_copy2d(Wrow, ld, Brow, ld, Kd, c0, 1.0, q));
    }
    RC(launch_copy2d(Wrow + c0, ld, Wkk, FZ_KD, Kd, Kd, 1.0, q));
    if (w_final) HIP_TRY(hipEventRecord(w_final, q));           // the last row block: W = L^-1 is complete
    HIP_TRY(hipEventRecord(w_row, q));                          // W[K, <=K] is final: its accumulation into the inverse may start
    if (rem > 0) {
        double* Wt = w.Wm.p + (int64_t)k1 * MOGP_TILE * ld;                    // Wt[>K, 0]
        GemmArgs g{};
        g.A = Lp; g.lda = FZ_KD; g.a_kmajor = 0; g.B = Wkk; g.ldb = FZ_KD; g.b_kmajor = 1;
        g.C = Wt + c0; g.ldc = ld; g.alpha = -1.0; g.beta = 0.0;
        g.mode = GM_KLO_J; g.mt = rem; g.nt = nk; g.K = (int)Kd;
        RC(gemm_call(m, g, gemm_flops(g, nullptr), q));
        if (k0 > 0) {
            GemmArgs u{};
            u.A = Lp; u.lda = FZ_KD; u.a_kmajor = 0; u.B = Wrow; u.ldb = ld; u.b_kmajor = 1;
            u.C = Wt; u.ldc = ld; u.alpha = -1.0; u.beta = 1.0;
            u.mode = GM_RECT; u.mt = rem; u.nt = k0; u.K = (int)Kd;
            RC(gemm_call(m, u, gemm_flops(u, nullptr), q));
        }
    }
    // Kinv[<=K, <=K] += W[K, <=K]^T W[K, <=K] on its own stream: the big launch runs next to the small, serial launches of the
    // following row blocks instead of in front of them (the accumulations only order among themselves).  (Two row blocks per launch --
    // rank 1024, half the passes over the accumulator -- was measured SLOWER, 13.31 vs 12.93 ms: the deferred block's work is missing
    // from what fills the chip next to the chain.)
    HIP_TRY(hipStreamWaitEvent(qacc, w_row, 0));
    GemmArgs g{};
    g.A = Wrow; g.lda = ld; g.a_kmajor = 1; g.B = Wrow; g.ldb = ld; g.b_kmajor = 1;
    g.C = w.B.p; g.ldc = ld; g.alpha = 1.0; g.beta = 1.0;
    g.mode = GM_LOWER; g.mt = g.nt = k1; g.K = (int)Kd;
    g.beta0_from = k0 + 1;                         // the row block K of the inverse is new (it held scratch): written, not accumulated -- no memset
    if (m->kinv_sparse && &w == &m->k && Kd == 4 * MOGP_TILE && k1 < (int)m->kinv_prefix.size()) {
        // only the tiles of the inverse the gradient reads (mogp_api.hip:kinv_plan): the planned tiles of the rows above k1, same arithmetic
        g.mode = GM_TASKS; g.tasks = m->d_kinv_acc.p; g.ntasks = m->kinv_prefix[k1]; g.task_chunked = 1; g.mt = g.nt = 0;
        return gemm_call(m, g, 2.0 * MOGP_TILE * MOGP_TILE * (double)Kd * g.ntasks, qacc);
    }
    return gemm_call(m, g, gemm_flops(g, nullptr), qacc);
}

}  // namespace

namespace mogp {

int spd_potri_fused(mogp_model* m, Spd& w) {
    w.flow_used = false;
    if (flow_enabled(m, w)) return spd_potri_flow(m, w);           // flow.hip: the same tile products as one resident dataflow kernel
    const int nb = w.nb;
    const int64_t ld = w.Npad;
    const int nouter = (nb + FZ_OB - 1) / FZ_OB;
    const auto t_host0 = std::chrono::steady_clock::now();
    hipStream_t crit = m->st, priv = m->st_priv ? m->st_priv : m->st, bulk = m->st2, inv = m->st3;
    static const bool split_acc = !(std::getenv("MOGP_ACC_STREAM") && std::atoi(std::getenv("MOGP_ACC_STREAM")) == 0);
    hipStream_t acc = (m->st4 && split_acc) ? m->st4 : m->st3;

    if (w.Wm.n < (size_t)ld * ld) {                      // nothing ever writes above the block diagonal of W: keep it finite
        RC(w.Wm.ensure((size_t)ld * ld));
        HIP_TRY(hipMemsetAsync(w.Wm.p, 0, (size_t)ld * ld * sizeof(double), crit));
    }
    for (auto& b : w.Pb) RC(b.ensure((size_t)ld * FZ_KD));
    if (w.Wd.n < (size_t)nouter * FZ_KD * FZ_KD) {       // W_KK store: tiles above the diagonal are never written and must be zero
        RC(w.Wd.ensure((size_t)nouter * FZ_KD * FZ_KD));
        HIP_TRY(hipMemsetAsync(w.Wd.p, 0, (size_t)nouter * FZ_KD * FZ_KD * sizeof(double), crit));
    }
    RC(w.chain_flags.ensure((size_t)(nouter + 1) * MOGP_CHAIN_FLAGS));          // hand-off words of the persistent chain kernels: zero per evaluation
    HIP_TRY(hipMemsetAsync(w.chain_flags.p, 0, (size_t)(nouter + 1) * MOGP_CHAIN_FLAGS * sizeof(unsigned), crit));
    auto Wk = [&](int kb) { return w.Wd.p + (int64_t)kb * FZ_KD * FZ_KD; };
    while ((int)w.inv_ev.size() < EV_PER_BLOCK * nouter + 2) {
        hipEvent_t e;
        HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        w.inv_ev.push_back(e);
    }
    auto ev = [&](int kb, int which) { return w.inv_ev[EV_PER_BLOCK * kb + which]; };
    hipEvent_t start = w.inv_ev[EV_PER_BLOCK * nouter], w_ready = w.inv_ev[EV_PER_BLOCK * nouter + 1];
    HIP_TRY(hipEventRecord(start, crit));                // the Gram matrix is in place

    // Host order matters as much as stream order: a late block takes about as long on the GPU as its ~45 launches take to enqueue,
    // so the chain of block kb+1 is enqueued before the bulk work of block kb -- but never a wait before the record it refers to.
    auto geom = [&](int kb, int& k0, int& k1, int& nk, int& rem, int& na, int& k2) {
        k0 = kb * FZ_OB; k1 = std::min(k0 + FZ_OB, nb); nk = k1 - k0; rem = nb - k1; na = std::min(FZ_OB, rem); k2 = k1 + na;
    };
    auto chain = [&](int kb) -> int {                    // priv: factor D_KK, W_KK (its inputs are ordered by the stream itself)
        int k0, k1, nk, rem, na, k2; geom(kb, k0, k1, nk, rem, na, k2);
        RC(intra_block_chain(m, w, Wk(kb), k0, k1, priv, kb));
        HIP_TRY(hipEventRecord(ev(kb, EV_BLK), priv));
        return 0;
    };
    auto next_diag = [&](int kb) -> int {                // priv: the next block's panel rows and the next diagonal block
        int k0, k1, nk, rem, na, k2; geom(kb, k0, k1, nk, rem, na, k2);
        if (rem <= 0) return 0;
        double* P = w.Pb[kb % MOGP_NPANEL].p;
        if (kb >= MOGP_NPANEL) HIP_TRY(hipStreamWaitEvent(priv, ev(kb - MOGP_NPANEL, EV_INV), 0));   // the panel buffer is free again
        if (kb >= 1) HIP_TRY(hipStreamWaitEvent(priv, ev(kb - 1, EV_BULK), 0));         // bulk1(kb-1): columns K+1;  implies bulk1(kb-2): columns K
        if (kb >= 1) HIP_TRY(hipStreamWaitEvent(priv, ev(kb - 1, EV_PANEL), 0));        // crit(kb-1): columns K below the diagonal block
        RC(panel_rows(m, w, P, Wk(kb), k0, nk, k1, na, true, priv));
        const double* Pn = P + (int64_t)k1 * MOGP_TILE * FZ_KD;
        GemmArgs u{};
        u.A = Pn; u.lda = FZ_KD; u.a_kmajor = 0; u.B = Pn; u.ldb = FZ_KD; u.b_kmajor = 0;
        u.C = w.A.p + (int64_t)k1 * MOGP_TILE * (ld + 1); u.ldc = ld; u.alpha = -1.0; u.beta = 1.0;
        u.mode = GM_RECT_LOWER; u.small = 2; u.mt = 2 * na; u.nt = 2 * na; u.K = nk * MOGP_TILE;
        RC(gemm_call(m, u, gemm_flops(u, nullptr), priv));
        HIP_TRY(hipEventRecord(ev(kb, EV_DIAG), priv));
        return 0;
    };
    HIP_TRY(hipStreamWaitEvent(priv, start, 0));
    // Tried on top of the persistent chain kernel (245 us per block instead of 490) and measured SLOWER than this schedule, round 3:
    //   * a look-ahead of two outer blocks (panel b applied to the rows of blocks b+1 and b+2 by small launches of their own, the tall
    //     launches covering rows > b+2): 14.8 vs 13.4 ms -- the small launches queue on the critical stream behind its tall ones;
    //   * the trailing update in three pieces (columns of block K+2, of block K+3, remainder on a stream of its own): 14.1-14.9 vs 13.4 ms,
    //     and a FIFTH CU-masked stream slows everything on the reserved CUs (chain kernel 520-680 us instead of 245);
    //   * the inverse's streams confined to a subset of the CUs: 15.7 ms (160 CUs) ... 54 ms (48 CUs) -- the panel buffers couple the chain
    //     to the inverse.
    //   * device-side hand-offs instead of events on the block cycle (a one-thread kernel raising a flag behind the producer, a spinning
    //     one-wave kernel in front of the consumer): the kernel behind such a gate starts without the cache invalidation the runtime
    //     attaches to a cross-queue wait and reads stale panel rows (wrong pivots from N = 4096 on); and with the GEMM work taken out
    //     (MOGP_FAKE_K=8, tools/fake_k.py) a block period is 475 us = chain 245 + mini-panel 140 + 90 for three hops and two tiny launches,
    //     so an event costs ~20 us, not the 45-55 us seen on a loaded chip -- those are slots that free in bursts.
    //   * the next block's columns and those of the block after it in ONE launch on the critical stream (high priority, one event fewer):
    //     13.25 vs 12.99 ms -- that launch has to wait for the whole remainder of the previous panel.
    //   * s_setprio 3 for the waves of the three tall launches on the block cycle (a 48-tile launch late in the factorisation takes 160-200 us
    //     next to the remainders' workgroups): 13.05 vs 13.01 ms -- it is not the MFMA issue slots they wait for.
    // What the traces say (profiles/r3_c1_timeline.txt): a block period is mini-panel (140 us) -> rest of the panel (~200) -> columns of the
    // block after next (~200) -> mini-panel, plus ~45 us per cross-stream event; the chain kernel runs next to the two tall launches, off
    // that cycle.  And the chip is busy with GEMM tiles throughout (47-58 TFLOP/s in every 500 us window): the evaluation is bound by what
    // five interleaved GEMM streams deliver, not by the chain.
    RC(chain(0));
    RC(next_diag(0));
    for (int kb = 0; kb < nouter; ++kb) {
        int k0, k1, nk, rem, na, k2; geom(kb, k0, k1, nk, rem, na, k2);
        const int Kd = nk * MOGP_TILE;
        double* P = w.Pb[kb % MOGP_NPANEL].p;
        const double* Pn = P + (int64_t)k1 * MOGP_TILE * FZ_KD;
        const double* Pr = P + (int64_t)k2 * MOGP_TILE * FZ_KD;
        const int nr = rem - na, n1 = std::min(FZ_OB, std::max(nr, 0));
        // 1. priv: the chain of the next block (needs only the next-diagonal update of this block, which is ahead of it in the stream)
        if (kb + 1 < nouter) RC(chain(kb + 1));
        // 2. crit: the rest of the panel, then the next block's columns below its diagonal block
        if (rem > 0) {
            HIP_TRY(hipStreamWaitEvent(crit, ev(kb, EV_DIAG), 0));                       // W_KK, a free buffer, the mini-panel
            if (nr > 0) RC(panel_rows(m, w, P, Wk(kb), k0, nk, k2, nr, false, crit));
            HIP_TRY(hipEventRecord(ev(kb, EV_REST), crit));
            if (nr > 0) {
                GemmArgs u{};
                u.A = Pr; u.lda = FZ_KD; u.a_kmajor = 0; u.B = Pn; u.ldb = FZ_KD; u.b_kmajor = 0;
                u.C = w.A.p + (int64_t)k2 * MOGP_TILE * ld + (int64_t)k1 * MOGP_TILE; u.ldc = ld; u.alpha = -1.0; u.beta = 1.0;
                u.mode = GM_RECT; u.mt = nr; u.nt = na; u.K = Kd;
                RC(gemm_call(m, u, gemm_flops(u, nullptr), crit));
            }
            HIP_TRY(hipEventRecord(ev(kb, EV_PANEL), crit));
        }
        // 3. bulk: trailing update, the columns of block K+2 first (the chain needs them next)
        if (nr > 0) {
            HIP_TRY(hipStreamWaitEvent(bulk, ev(kb, EV_REST), 0));
            GemmArgs u{};
            u.A = Pr; u.lda = FZ_KD; u.a_kmajor = 0; u.B = Pr; u.ldb = FZ_KD; u.b_kmajor = 0;
            u.C = w.A.p + (int64_t)k2 * MOGP_TILE * (ld + 1); u.ldc = ld; u.alpha = -1.0; u.beta = 1.0;
            u.mode = GM_RECT_LOWER; u.mt = nr; u.nt = n1; u.K = Kd;
            RC(gemm_call(m, u, gemm_flops(u, nullptr), bulk));
        }
        HIP_TRY(hipEventRecord(ev(kb, EV_BULK), bulk));
        // 4. priv: mini-panel and next-diagonal update of the next block
        if (kb + 1 < nouter) RC(next_diag(kb + 1));
        // 5. bulk: the rest of the trailing update
        if (nr > n1) {
            const double* Pq = Pr + (int64_t)n1 * MOGP_TILE * FZ_KD;
            GemmArgs v{};
            v.A = Pq; v.lda = FZ_KD; v.a_kmajor = 0; v.B = Pq; v.ldb = FZ_KD; v.b_kmajor = 0;
            v.C = w.A.p + (int64_t)(k2 + n1) * MOGP_TILE * (ld + 1); v.ldc = ld; v.alpha = -1.0; v.beta = 1.0;
            v.mode = GM_LOWER; v.mt = v.nt = nr - n1; v.K = Kd;
            RC(gemm_call(m, v, gemm_flops(v, nullptr), bulk));
        }
        // 6. inv
        HIP_TRY(hipStreamWaitEvent(inv, ev(kb, EV_BLK), 0));
        if (rem > 0) HIP_TRY(hipStreamWaitEvent(inv, ev(kb, EV_REST), 0));
        RC(inverse_step(m, w, Wk(kb), k0, k1, P + (int64_t)k1 * MOGP_TILE * FZ_KD, inv, acc, ev(kb, EV_WROW), kb == nouter - 1 ? w_ready : nullptr));
        HIP_TRY(hipEventRecord(ev(kb, EV_INV), inv));
        HIP_TRY(hipEventRecord(ev(kb, EV_ACC), acc));
    }
    HIP_TRY(hipEventRecord(ev(nouter - 1, EV_DIAG), bulk));                              // reuse: everything on the bulk stream
    HIP_TRY(hipStreamWaitEvent(crit, ev(nouter - 1, EV_BLK), 0));
    HIP_TRY(hipStreamWaitEvent(crit, ev(nouter - 1, EV_DIAG), 0));
    // W is complete one step before the inverse: the caller's W y / W^T z run next to the last accumulate of the inverse and
    // spd_potri_fused_finish() joins the inverse stream afterwards
    HIP_TRY(hipStreamWaitEvent(crit, w_ready, 0));
    w.fused_last_inv = ev(nouter - 1, EV_ACC);
    w.fused_last_wt = ev(nouter - 1, EV_INV);
    if (std::getenv("MOGP_DEBUG_HOST")) {
        const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_host0).count();
        fprintf(stderr, "spd_potri_fused: host enqueue %.0f us, %d outer blocks\n", us, nouter);
    }
    return 0;
}


// the inverse (w.B) is complete on the critical stream after this
int spd_potri_fused_finish(mogp_model* m, Spd& w) {
    if (w.fused_last_inv) HIP_TRY(hipStreamWaitEvent(m->st, w.fused_last_inv, 0));
    if (w.fused_last_wt) HIP_TRY(hipStreamWaitEvent(m->st, w.fused_last_wt, 0));
    return 0;
}

}  // namespace mogp
