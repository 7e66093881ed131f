// sweep.hip -- single-sweep blocked inversion of the SPD matrix Kj (block Gauss-Jordan / "sweep operator"):
//   for each 512-wide pivot block K:   S = A_KK (the current Schur complement: the same SPD block a Cholesky would meet)
//       P = S^-1 (Cholesky + inverse of a 512 x 512 block: the ONLY serial part; log|Kj| comes from its pivots)
//       X = A_OK P                      panels (all other rows O, one launch each for the column part and the row part)
//       A_OO -= X A_KO                  rank-512 update of EVERYTHING else (swept and unswept alike), bulk stream
//       A_OK = X, A_KK = -P
//   after the last block A = -Kj^-1 (lower triangle).
// Same N^3 flops as POTRF + TRTRI + LAUUM, but one pass whose serial chain is nb/4 small block inversions instead of nb
// leaf->panel->update steps three times over, and whose bulk work is always the full triangle -- the chip stays busy.
// Look-ahead: the critical stream applies the update to the NEXT pivot block's panels first (a1, b1) and goes on to invert
// it while the bulk stream applies the rest (a2, b2, c).  Old panels are copied out (double buffered) so the new ones can
// be written in place.
// Owned-rows form (m->sh_owned: every sharded evaluation over more than one rank): a rank touches the matrix ONLY in the tile rows it owns.  What
// a block needs from the other ranks never enters the matrix: the exchange is unpacked straight into the block's work buffers -- the pivot block into the
// Schur workspace (where the chain factors it), the old column panel into Uc, the old row panel into Ur -- and the new row panel, which every rank forms
// for all four pivot tile rows, goes to a buffer of its own (Xr) with the owned rows copied into the matrix.  The matrix can then live in a RowBacked range
// with physical memory under the owned rows only (mogp_model.h).
#include "mogp_model.h"

#include <cstdlib>
#include <limits>

using namespace mogp;

#define RC(x) do { int r__ = (x); if (r__) return r__; } while (0)
#define SW_OB 4

static GemmArgs upd(const double* A, int64_t lda, int akm, const double* B, int64_t ldb, int bkm, double* C, int64_t ldc,
                    int mode, int mt, int nt, int64_t K, int small = 0) {
    GemmArgs g{};
    g.A = A; g.lda = lda; g.a_kmajor = akm; g.B = B; g.ldb = ldb; g.b_kmajor = bkm; g.C = C; g.ldc = ldc;
    g.alpha = -1.0; g.beta = 1.0; g.mode = mode; g.mt = mt; g.nt = nt; g.K = (int)K; g.small = small;
    return g;
}

namespace mogp {

int sweep_prepare(mogp_model* m, Spd& w) {
    const int nb = w.nb;
    const int64_t ld = w.Npad;
    const int nouter = (nb + SW_OB - 1) / SW_OB;
    RC(spd_alloc(m->ws, (int64_t)std::min(SW_OB, nb) * MOGP_TILE));
    if (nb % SW_OB && nb > SW_OB) RC(spd_alloc(m->ws_tail, (int64_t)(nb % SW_OB) * MOGP_TILE));
    for (int b = 0; b < 2; ++b) {
        RC(m->swU[b].ensure((size_t)ld * SW_OB * MOGP_TILE));
        RC(m->swUr[b].ensure((size_t)ld * SW_OB * MOGP_TILE));
        if (m->sh_owned) RC(m->swXr[b].ensure((size_t)ld * SW_OB * MOGP_TILE));
    }
    while ((int)m->sw_ev.size() < 2 * nouter) {
        hipEvent_t e;
        HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        m->sw_ev.push_back(e);
    }
    return 0;
}

int sweep_nblocks(const Spd& w) { return (w.nb + SW_OB - 1) / SW_OB; }

// one pivot block.  With m->sh_n > 1 (sharded evaluation) the chain (P, panels) is repeated by every rank from the assembled
// authoritative panel and the rank-Kd update touches only the tile rows this rank owns (row i is owned by rank i % sh_n).
int sweep_block(mogp_model* m, Spd& w, int kb, hipEvent_t* prof, hipEvent_t panel_ready, hipEvent_t* stall) {
    const int nb = w.nb;
    const int64_t ld = w.Npad;
    const int rm = m->sh_n > 1 ? m->sh_n : 0, rr = m->sh_rank;
    const bool ow = m->sh_owned;                                 // owned-rows form (head of this file)
    auto owned = [&](int i) { return rm == 0 || i % rm == rr; };
    // bulk stream: see spd_potrf; a rank of a sharded evaluation updates 1 / sh_n of the rows, so its chain matters at every size
    static const int force_masked = std::getenv("MOGP_SWEEP_MASKED") ? std::atoi(std::getenv("MOGP_SWEEP_MASKED")) : 0;    // measurement switch
    hipStream_t q1 = m->st, q2 = (rm == 0 && !force_masked && w.nb > MOGP_CHAIN_BOUND_TILES && m->st2u) ? m->st2u : m->st2;
    double* A = w.A.p;
    const int k0 = kb * SW_OB, k1 = std::min(k0 + SW_OB, nb), nk = k1 - k0;
    const int64_t Kd = (int64_t)nk * MOGP_TILE;
    const int below = nb - k1;                                   // tile rows under the pivot block
    Spd& s = (nk == m->ws.nb) ? m->ws : m->ws_tail;
    double* Akk = A + (int64_t)k0 * MOGP_TILE * (ld + 1);
    double* Acol = A + (int64_t)k1 * MOGP_TILE * ld + (int64_t)k0 * MOGP_TILE;      // A[O>][K]
    double* Arow = A + (int64_t)k0 * MOGP_TILE * ld;                                  // A[K][O<]
    // ---- P = S^-1 on the critical stream.  With the persistent chain kernel (chain.hip): ONE launch factors the block in place and leaves
    // W = L^-1, then P = W^T W -- instead of a copy, 4 leaves + 6 small GEMMs, 5 launches of the triangular inverse and the LAUUM product
    // (~0.8 ms of dependent launches per block: the part of a sharded evaluation that every rank repeats).  The kernel's workgroups need
    // CUs with 128 KB of LDS free: they find them on the reserved CUs as long as the bulk stream is masked off those (q2 == st2) and the
    // ranks do not share a GPU (external communicator = the test suite's ranks on one device: two such kernels would starve each other).
    const bool chain = chain_enabled(m) && q2 == m->st2 && m->st_priv && m->ctx->comm.kind != MOGP_COMM_EXTERNAL;
    // MOGP_SHARD_FACTOR_ONCE=1 (an A/B switch): only the rank that owns the pivot block's first tile row inverts the Schur block; the others
    // contribute zeros to an all-reduce of [P | log-det parts | pivot report] -- "all-gather of panel factors" in BASELINE.json's wording.
    // Same bits (x + 0), one more collective on the critical stream per block; the default repeats the 0.3 ms inversion on every rank instead.
    const bool once = rm > 0 && m->sh_factor_once;
    const bool mine = !once || (k0 % rm) == rr;
    if (chain && kb == 0) {                              // (also on the ranks that, factoring once, run their first chain kernel at a later block)
        const int nouter = (nb + SW_OB - 1) / SW_OB;
        RC(w.chain_flags.ensure((size_t)(nouter + 1) * MOGP_CHAIN_FLAGS));
        HIP_TRY(hipMemsetAsync(w.chain_flags.p, 0, (size_t)(nouter + 1) * MOGP_CHAIN_FLAGS * sizeof(unsigned), q1));
    }
    if (!mine) {
        HIP_TRY(hipMemsetAsync(s.B.p, 0, (size_t)Kd * Kd * sizeof(double), q1));
    } else if (chain) {
        const int nouter = (nb + SW_OB - 1) / SW_OB;
        if (s.Wm.n < (size_t)Kd * Kd) {
            RC(s.Wm.ensure((size_t)Kd * Kd));
            HIP_TRY(hipMemsetAsync(s.Wm.p, 0, (size_t)Kd * Kd * sizeof(double), q1));     // the tiles above the diagonal are never written
        }
        // (owned-rows form: the block is in the Schur workspace, Kd x Kd; the kernel addresses its block as base + k0 * 128 * (ld + 1) and nothing else)
        double* cA = ow ? s.A.p - (int64_t)k0 * MOGP_TILE * (Kd + 1) : A;
        RC(launch_chain(cA, ow ? Kd : ld, k0, nk, w.invd.p, w.logdet.p, m->d_info.p, 0, s.Wm.p, Kd, w.chain_flags.p + (size_t)kb * MOGP_CHAIN_FLAGS,
                        w.chain_flags.p + (size_t)nouter * MOGP_CHAIN_FLAGS, q1));
        GemmArgs g{};
        g.A = s.Wm.p; g.lda = Kd; g.a_kmajor = 1; g.B = s.Wm.p; g.ldb = Kd; g.b_kmajor = 1;
        g.C = s.B.p; g.ldc = Kd; g.alpha = 1.0; g.beta = 0.0;
        g.mode = GM_LAUUM; g.mt = g.nt = nk; g.K = (int)Kd;
        RC(gemm_call(m, g, gemm_flops(g, nullptr), q1));
    } else {
        if (!ow) RC(launch_copy2d(s.A.p, Kd, Akk, ld, Kd, Kd, 1.0, q1));
        RC(spd_potrf(m, s, (long long)k0 * MOGP_TILE));
        HIP_TRY(hipMemcpyAsync(w.logdet.p + k0, s.logdet.p, nk * sizeof(double), hipMemcpyDeviceToDevice, q1));
        RC(spd_trtri(m, s));
        RC(spd_lauum(m, s));
    }
    if (mine) RC(launch_symmetrize(s.B.p, Kd, Kd, q1));
    if (once) RC(shard_factor_bcast(m, w, s, k0, nk, mine, q1));
    const double* P = s.B.p;
    if (stall) HIP_TRY(hipEventRecord(stall[0], q1));
    if (panel_ready) HIP_TRY(hipStreamWaitEvent(q1, panel_ready, 0));        // the rest of the panel (other ranks' rows) arrives on the communication stream
    if (stall) HIP_TRY(hipEventRecord(stall[1], q1));
    // ---- old panels out, new panels X = U P in place, diagonal block = -P
    double* Uc = m->swU[kb & 1].p;                                                     // [below*128][Kd]
    double* Ur = m->swUr[kb & 1].p;                                                    // [Kd][ld] (first k0*128 columns used)
    double* Xrow = ow ? m->swXr[kb & 1].p : Arow;                                     // the new row part [Kd][ld]
    if (!ow) {                                                                         // (owned-rows form: the unpack has put the old panels there)
        RC(launch_copy2d(Uc, Kd, Acol, ld, (int64_t)below * MOGP_TILE, Kd, 1.0, q1));
        RC(launch_copy2d(Ur, ld, Arow, ld, Kd, (int64_t)k0 * MOGP_TILE, 1.0, q1));
    }
    if (below > 0) {
        GemmArgs g = upd(Uc, Kd, 0, P, Kd, 0, Acol, ld, GM_RECT, 2 * below, nk, Kd, 1);
        g.alpha = 1.0; g.beta = 0.0;
        g.row_mod = rm; g.row_rem = rr; g.row_off = k1; g.row_shift = 1;    // sharded: X = U P only for the tile rows this rank updates
        RC(gemm_call(m, g, gemm_flops(g, nullptr), q1));
    }
    if (k0 > 0) {
        GemmArgs g = upd(P, Kd, 0, Ur, ld, 1, Xrow, ld, GM_RECT, nk, k0, Kd);
        g.alpha = 1.0; g.beta = 0.0;
        RC(gemm_call(m, g, gemm_flops(g, nullptr), q1));
    }
    if (!ow) RC(launch_copy2d(Akk, ld, P, Kd, Kd, Kd, -1.0, q1));
    else
        for (int i = k0; i < k1; ++i) {
            if (!owned(i)) continue;
            const int64_t r = (int64_t)(i - k0) * MOGP_TILE;
            if (k0 > 0) RC(launch_copy2d(A + (int64_t)i * MOGP_TILE * ld, ld, Xrow + r * ld, ld, MOGP_TILE, (int64_t)k0 * MOGP_TILE, 1.0, q1));
            RC(launch_copy2d(A + (int64_t)i * MOGP_TILE * ld + (int64_t)k0 * MOGP_TILE, ld, P + r * Kd, Kd, MOGP_TILE, Kd, -1.0, q1));
        }
    HIP_TRY(hipEventRecord(m->sw_ev[2 * kb], q1));                                     // X(kb) ready
    if (prof) HIP_TRY(hipEventRecord(prof[0], q1));
    // ---- rank-Kd update of everything outside the pivot block
    const int nk2 = std::min(SW_OB, below);                                            // tile columns of the next pivot block
    if (kb > 0) HIP_TRY(hipStreamWaitEvent(q1, m->sw_ev[2 * (kb - 1) + 1], 0));        // a1/b1 share tiles with bulk(kb-1)
    if (nk2 > 0) {
        GemmArgs a1 = upd(Acol, ld, 0, Uc, Kd, 0, A + (int64_t)k1 * MOGP_TILE * (ld + 1), ld, GM_RECT_LOWER, below, nk2, Kd);
        a1.row_mod = rm; a1.row_rem = rr; a1.row_off = k1;
        RC(gemm_call(m, a1, gemm_flops(a1, nullptr), q1));
        if (k0 > 0) {
            GemmArgs b1 = upd(Acol, ld, 0, Ur, ld, 1, A + (int64_t)k1 * MOGP_TILE * ld, ld, GM_RECT, nk2, k0, Kd);
            b1.row_mod = rm; b1.row_rem = rr; b1.row_off = k1;
            RC(gemm_call(m, b1, gemm_flops(b1, nullptr), q1));
        }
    }
    if (prof) HIP_TRY(hipEventRecord(prof[1], q1));
    HIP_TRY(hipStreamWaitEvent(q2, m->sw_ev[2 * kb], 0));
    if (prof) HIP_TRY(hipEventRecord(prof[2], q2));
    const int rest = below - nk2;
    if (rest > 0) {
        const int64_t r0 = (int64_t)(k1 + nk2) * MOGP_TILE;
        const double* Xr2 = A + r0 * ld + (int64_t)k0 * MOGP_TILE;
        GemmArgs a2 = upd(Xr2, ld, 0, Uc + (int64_t)nk2 * MOGP_TILE * Kd, Kd, 0, A + r0 * (ld + 1), ld, GM_LOWER, rest, rest, Kd);
        a2.row_mod = rm; a2.row_rem = rr; a2.row_off = k1 + nk2;
        RC(gemm_call(m, a2, gemm_flops(a2, nullptr), q2));
        if (k0 > 0) {
            GemmArgs b2 = upd(Xr2, ld, 0, Ur, ld, 1, A + r0 * ld, ld, GM_RECT, rest, k0, Kd);
            b2.row_mod = rm; b2.row_rem = rr; b2.row_off = k1 + nk2;
            RC(gemm_call(m, b2, gemm_flops(b2, nullptr), q2));
        }
    }
    if (k0 > 0) {
        GemmArgs c = upd(Xrow, ld, 1, Ur, ld, 1, A, ld, GM_LOWER, k0, k0, Kd);
        c.row_mod = rm; c.row_rem = rr; c.row_off = 0;
        RC(gemm_call(m, c, gemm_flops(c, nullptr), q2));
    }
    if (prof) HIP_TRY(hipEventRecord(prof[3], q2));
    HIP_TRY(hipEventRecord(m->sw_ev[2 * kb + 1], q2));                                 // bulk(kb) done
    return 0;
}

int sweep_finish(mogp_model* m, Spd& w) {
    const int nouter = sweep_nblocks(w);
    HIP_TRY(hipStreamWaitEvent(m->st, m->sw_ev[2 * (nouter - 1) + 1], 0));
    return 0;
}

int spd_sweep(mogp_model* m, Spd& w) {
    RC(sweep_prepare(m, w));
    for (int kb = 0; kb < sweep_nblocks(w); ++kb) RC(sweep_block(m, w, kb));
    return sweep_finish(m, w);
}

// ---- sharded evaluation: assembling the authoritative panel of pivot block kb with ONE all-gather ------------------------
// column part: tile rows i >= k0, 128 x Kd each, owner i % P; rank r's rows are first_r + idx * P
// (grid.y = 8 slabs of 16 rows per tile; 16-byte accesses along the block's columns -- Kd and ld are multiples of 128)
__global__ __launch_bounds__(256) void k_shard_pack(const double* __restrict__ A, int64_t ld, int k0, int row_lo, int row_hi, int P, int rank, int64_t Kd,
                                                    double* __restrict__ send) {
    const int first = row_lo + ((rank - row_lo % P) + P) % P;
    const int i = first + (int)blockIdx.x * P;
    if (i >= row_hi) return;
    const int r0 = 16 * (int)blockIdx.y, kd2 = (int)(Kd >> 1);
    const double* src = A + ((int64_t)i * MOGP_TILE + r0) * ld + (int64_t)k0 * MOGP_TILE;
    double* dst = send + ((int64_t)blockIdx.x * MOGP_TILE + r0) * Kd;
    for (int e = threadIdx.x; e < 16 * kd2; e += 256) {
        const int r = e / kd2, c = 2 * (e - r * kd2);
        *reinterpret_cast<double2*>(dst + (int64_t)r * Kd + c) = *reinterpret_cast<const double2*>(src + (int64_t)r * ld + c);
    }
}
// (the rows of rank `own` are skipped: they are where they were packed from)
__global__ __launch_bounds__(256) void k_shard_unpack(double* __restrict__ A, int64_t ld, int k0, int row_lo, int row_hi, int P, int64_t Kd, int64_t chunk,
                                                      const double* __restrict__ recv, int own) {
    const int r = blockIdx.z;
    if (r == own) return;
    const int first = row_lo + ((r - row_lo % P) + P) % P;
    const int i = first + (int)blockIdx.x * P;
    if (i >= row_hi) return;
    const int r0 = 16 * (int)blockIdx.y, kd2 = (int)(Kd >> 1);
    double* dst = A + ((int64_t)i * MOGP_TILE + r0) * ld + (int64_t)k0 * MOGP_TILE;
    const double* src = recv + (int64_t)r * chunk + ((int64_t)blockIdx.x * MOGP_TILE + r0) * Kd;
    for (int e = threadIdx.x; e < 16 * kd2; e += 256) {
        const int rr = e / kd2, c = 2 * (e - rr * kd2);
        *reinterpret_cast<double2*>(dst + (int64_t)rr * ld + c) = *reinterpret_cast<const double2*>(src + (int64_t)rr * Kd + c);
    }
}

// owned-rows form: EVERY rank's rows (this rank's own included) from the receive buffer into the block's work buffers, never into the matrix: tile rows of
// the pivot block -> the Schur workspace S (Kd x Kd), tile rows below it -> the old column panel Uc ([below * 128][Kd]); 128 x Kd contiguous on both sides
__global__ __launch_bounds__(256) void k_shard_unpack_owned(const double* __restrict__ recv, int64_t chunk, int row_lo, int row_hi, int P, int k0, int k1, int64_t Kd,
                                                            double* __restrict__ S, double* __restrict__ Uc) {
    const int r = blockIdx.z;
    const int first = row_lo + ((r - row_lo % P) + P) % P;
    const int i = first + (int)blockIdx.x * P;
    if (i >= row_hi) return;
    const int64_t slab = 16 * Kd, off = (int64_t)blockIdx.y * slab;
    const double* src = recv + (int64_t)r * chunk + (int64_t)blockIdx.x * MOGP_TILE * Kd + off;
    double* dst = (i < k1 ? S + (int64_t)(i - k0) * MOGP_TILE * Kd : Uc + (int64_t)(i - k1) * MOGP_TILE * Kd) + off;
    for (int64_t e = 2 * (int64_t)threadIdx.x; e < slab; e += 512) *reinterpret_cast<double2*>(dst + e) = *reinterpret_cast<const double2*>(src + e);
}

// factor-once: [P (Kd x Kd) | nk log-det parts | pivot report as a double (0: none, else index + 1)] in one buffer, so that ONE all-reduce carries
// the owner's result to everybody (the other ranks hold zeros)
__global__ void k_factor_pack(double* __restrict__ buf, int64_t kd2, const double* __restrict__ logdet, int nk, const unsigned long long* __restrict__ info, int mine) {
    const int t = threadIdx.x;
    if (t < nk) buf[kd2 + t] = mine ? logdet[t] : 0.0;
    if (t == 0) {
        const unsigned long long v = *info;
        buf[kd2 + nk] = (mine && v != ~0ull) ? (double)(v + 1ull) : 0.0;
    }
}
__global__ void k_factor_unpack(const double* __restrict__ buf, int64_t kd2, double* __restrict__ logdet, int nk, unsigned long long* __restrict__ info) {
    const int t = threadIdx.x;
    if (t < nk) logdet[t] = buf[kd2 + t];
    if (t == 0 && buf[kd2 + nk] > 0.0) atomicMin(info, (unsigned long long)(buf[kd2 + nk] - 1.0));
}

// chunk of one rank for pivot block kb: [maxrows column tiles, 128 x Kd each][maxpiv row tiles, 128 x (k0*128) each]
// (row tiles: the part LEFT of the pivot block of the pivot tile rows this rank owns -- the transposed half of the panel)
struct ShardGeom { int k0, k1, nk, maxrows, maxpiv; int64_t Kd, cols, rowoff, chunk; };
static ShardGeom shard_geometry(const Spd& w, int kb, int P) {
    ShardGeom g;
    g.k0 = kb * SW_OB;
    g.k1 = std::min(g.k0 + SW_OB, w.nb);
    g.nk = g.k1 - g.k0;
    g.Kd = (int64_t)g.nk * MOGP_TILE;
    g.cols = (int64_t)g.k0 * MOGP_TILE;
    g.maxrows = (w.nb - g.k0 + P - 1) / P;
    g.maxpiv = (g.nk + P - 1) / P;
    g.rowoff = (int64_t)g.maxrows * MOGP_TILE * g.Kd;
    g.chunk = g.rowoff + (int64_t)g.maxpiv * MOGP_TILE * g.cols;
    return g;
}
static inline int first_owned(int k0, int P, int r) { return k0 + ((r - k0 % P) + P) % P; }

// part: 0 = the whole panel in one message (the stage API and MOGP_SHARD_SPLIT=0), 1 = the pivot block's own tile rows (all the serial part
// needs: 2 MB), 2 = the rest (the column part below the block + the row part left of it: up to 134 MB at configs[2], needed by the panels only)
static void part_rows(const ShardGeom& g, int nb, int part, int& lo, int& hi) {
    lo = part == 2 ? g.k1 : g.k0;
    hi = part == 1 ? g.k1 : nb;
}
static int64_t part_chunk(const ShardGeom& g, int nb, int P, int part, int& maxrows, int64_t& rowoff) {
    int lo, hi;
    part_rows(g, nb, part, lo, hi);
    maxrows = (hi - lo + P - 1) / P;
    rowoff = (int64_t)maxrows * MOGP_TILE * g.Kd;
    return rowoff + (part == 1 ? 0 : (int64_t)g.maxpiv * MOGP_TILE * g.cols);
}

int shard_pack_part(mogp_model* m, Spd& w, int kb, int part, DevBuf<double>& sendb, DevBuf<double>& recvb, int64_t* count, hipStream_t st) {
    const int P = m->sh_n;
    const ShardGeom g = shard_geometry(w, kb, P);
    int maxrows, lo, hi; int64_t rowoff;
    const int64_t chunk = part_chunk(g, w.nb, P, part, maxrows, rowoff);
    part_rows(g, w.nb, part, lo, hi);
    RC(sendb.ensure((size_t)std::max<int64_t>(chunk, 1))); RC(recvb.ensure((size_t)std::max<int64_t>(chunk, 1) * P));
    *count = chunk;
    if (maxrows > 0) {
        hipLaunchKernelGGL(k_shard_pack, dim3(maxrows, 8), dim3(256), 0, st, w.A.p, w.Npad, g.k0, lo, hi, P, m->sh_rank, g.Kd, sendb.p);
        HIP_TRY(hipGetLastError());
    }
    if (part != 1 && g.cols > 0) {
        int idx = 0;
        for (int i = first_owned(g.k0, P, m->sh_rank); i < g.k1; i += P, ++idx)
            RC(launch_copy2d(sendb.p + rowoff + (int64_t)idx * MOGP_TILE * g.cols, g.cols, w.A.p + (int64_t)i * MOGP_TILE * w.Npad, w.Npad,
                             MOGP_TILE, g.cols, 1.0, st));
    }
    return 0;
}

int shard_unpack_part(mogp_model* m, Spd& w, int kb, int part, DevBuf<double>& recvb, hipStream_t st) {
    const int P = m->sh_n;
    const ShardGeom g = shard_geometry(w, kb, P);
    int maxrows, lo, hi; int64_t rowoff;
    const int64_t chunk = part_chunk(g, w.nb, P, part, maxrows, rowoff);
    part_rows(g, w.nb, part, lo, hi);
    if (m->sh_owned) {
        Spd& s = (g.nk == m->ws.nb) ? m->ws : m->ws_tail;
        if (maxrows > 0) {
            hipLaunchKernelGGL(k_shard_unpack_owned, dim3(maxrows, 8, P), dim3(256), 0, st, recvb.p, chunk, lo, hi, P, g.k0, g.k1, g.Kd, s.A.p, m->swU[kb & 1].p);
            HIP_TRY(hipGetLastError());
        }
        if (part != 1 && g.cols > 0)
            for (int i = g.k0; i < g.k1; ++i) {
                const int r = i % P, idx = (i - first_owned(g.k0, P, r)) / P;
                RC(launch_copy2d(m->swUr[kb & 1].p + (int64_t)(i - g.k0) * MOGP_TILE * w.Npad, w.Npad,
                                 recvb.p + (int64_t)r * chunk + rowoff + (int64_t)idx * MOGP_TILE * g.cols, g.cols, MOGP_TILE, g.cols, 1.0, st));
            }
        return 0;
    }
    if (maxrows > 0) {
        if (P > 1) hipLaunchKernelGGL(k_shard_unpack, dim3(maxrows, 8, P), dim3(256), 0, st, w.A.p, w.Npad, g.k0, lo, hi, P, g.Kd, chunk, recvb.p, m->sh_rank);
        HIP_TRY(hipGetLastError());
    }
    if (part != 1 && g.cols > 0) {
        for (int i = g.k0; i < g.k1; ++i) {
            const int r = i % P, idx = (i - first_owned(g.k0, P, r)) / P;
            if (r == m->sh_rank) continue;                    // this rank's own rows never left
            RC(launch_copy2d(w.A.p + (int64_t)i * MOGP_TILE * w.Npad, w.Npad, recvb.p + (int64_t)r * chunk + rowoff + (int64_t)idx * MOGP_TILE * g.cols,
                             g.cols, MOGP_TILE, g.cols, 1.0, st));
        }
    }
    return 0;
}

int shard_pack(mogp_model* m, Spd& w, int kb, double** send, double** recv, int64_t* count) {
    RC(shard_pack_part(m, w, kb, 0, m->sh_send, m->sh_recv, count, m->st));
    *send = m->sh_send.p; *recv = m->sh_recv.p;
    return 0;
}

int shard_unpack(mogp_model* m, Spd& w, int kb) { return shard_unpack_part(m, w, kb, 0, m->sh_recv, m->st); }

// factor-once (sweep_block): the owner's P, log-det parts and pivot report to every rank by ONE all-reduce (everybody else adds zeros)
int shard_factor_bcast(mogp_model* m, Spd& w, Spd& s, int k0, int nk, bool mine, hipStream_t st) {
    const int64_t kd2 = (int64_t)nk * MOGP_TILE * nk * MOGP_TILE;
    RC(m->sh_fact.ensure((size_t)kd2 + nk + 1));
    HIP_TRY(hipMemcpyAsync(m->sh_fact.p, s.B.p, (size_t)kd2 * sizeof(double), hipMemcpyDeviceToDevice, st));
    hipLaunchKernelGGL(k_factor_pack, dim3(1), dim3(64), 0, st, m->sh_fact.p, kd2, w.logdet.p + k0, nk, m->d_info.p, mine ? 1 : 0);
    HIP_TRY(hipGetLastError());
    RC(comm_allreduce(m->ctx, m->sh_fact.p, kd2 + nk + 1, st));
    HIP_TRY(hipMemcpyAsync(s.B.p, m->sh_fact.p, (size_t)kd2 * sizeof(double), hipMemcpyDeviceToDevice, st));
    hipLaunchKernelGGL(k_factor_unpack, dim3(1), dim3(64), 0, st, m->sh_fact.p, kd2, w.logdet.p + k0, nk, m->d_info.p);
    HIP_TRY(hipGetLastError());
    return 0;
}

}  // namespace mogp
