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
#include "mogp_model.h"

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

int spd_sweep(mogp_model* m, Spd& w) {
    const int nb = w.nb;
    const int64_t ld = w.Npad;
    const int nouter = (nb + SW_OB - 1) / SW_OB;
    RC(spd_alloc(m->ws, (int64_t)std::min(SW_OB, nb) * MOGP_TILE));
    if (nb % SW_OB && nb > SW_OB) RC(spd_alloc(m->ws_tail, (int64_t)(nb % SW_OB) * MOGP_TILE));
    for (int b = 0; b < 2; ++b) {
        RC(m->swU[b].ensure((size_t)ld * SW_OB * MOGP_TILE));
        RC(m->swUr[b].ensure((size_t)ld * SW_OB * MOGP_TILE));
    }
    while ((int)m->sw_ev.size() < 2 * nouter) {
        hipEvent_t e;
        HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        m->sw_ev.push_back(e);
    }
    hipStream_t q1 = m->st, q2 = m->st2;
    double* A = w.A.p;
    for (int kb = 0; kb < nouter; ++kb) {
        const int k0 = kb * SW_OB, k1 = std::min(k0 + SW_OB, nb), nk = k1 - k0;
        const int64_t Kd = (int64_t)nk * MOGP_TILE;
        const int below = nb - k1;                                   // tile rows under the pivot block
        Spd& s = (nk == m->ws.nb) ? m->ws : m->ws_tail;
        double* Akk = A + (int64_t)k0 * MOGP_TILE * (ld + 1);
        double* Acol = A + (int64_t)k1 * MOGP_TILE * ld + (int64_t)k0 * MOGP_TILE;      // A[O>][K]
        double* Arow = A + (int64_t)k0 * MOGP_TILE * ld;                                  // A[K][O<]
        // ---- P = S^-1 on the critical stream
        RC(launch_copy2d(s.A.p, Kd, Akk, ld, Kd, Kd, 1.0, q1));
        RC(spd_potrf(m, s, (long long)k0 * MOGP_TILE));
        HIP_TRY(hipMemcpyAsync(w.logdet.p + k0, s.logdet.p, nk * sizeof(double), hipMemcpyDeviceToDevice, q1));
        RC(spd_trtri(m, s));
        RC(spd_lauum(m, s));
        RC(launch_symmetrize(s.B.p, Kd, Kd, q1));
        const double* P = s.B.p;
        // ---- old panels out, new panels X = U P in place, diagonal block = -P
        double* Uc = m->swU[kb & 1].p;                                                     // [below*128][Kd]
        double* Ur = m->swUr[kb & 1].p;                                                    // [Kd][ld] (first k0*128 columns used)
        RC(launch_copy2d(Uc, Kd, Acol, ld, (int64_t)below * MOGP_TILE, Kd, 1.0, q1));
        RC(launch_copy2d(Ur, ld, Arow, ld, Kd, (int64_t)k0 * MOGP_TILE, 1.0, q1));
        if (below > 0) {
            GemmArgs g = upd(Uc, Kd, 0, P, Kd, 0, Acol, ld, GM_RECT, 2 * below, nk, Kd, 1);
            g.alpha = 1.0; g.beta = 0.0;
            RC(gemm_call(m, g, gemm_flops(g, nullptr), q1));
        }
        if (k0 > 0) {
            GemmArgs g = upd(P, Kd, 0, Ur, ld, 1, Arow, ld, GM_RECT, nk, k0, Kd);
            g.alpha = 1.0; g.beta = 0.0;
            RC(gemm_call(m, g, gemm_flops(g, nullptr), q1));
        }
        RC(launch_copy2d(Akk, ld, P, Kd, Kd, Kd, -1.0, q1));
        HIP_TRY(hipEventRecord(m->sw_ev[2 * kb], q1));                                     // X(kb) ready
        // ---- rank-Kd update of everything outside the pivot block
        const int nk2 = std::min(SW_OB, below);                                            // tile columns of the next pivot block
        if (kb > 0) HIP_TRY(hipStreamWaitEvent(q1, m->sw_ev[2 * (kb - 1) + 1], 0));        // a1/b1 share tiles with bulk(kb-1)
        if (nk2 > 0) {
            GemmArgs a1 = upd(Acol, ld, 0, Uc, Kd, 0, A + (int64_t)k1 * MOGP_TILE * (ld + 1), ld, GM_RECT_LOWER, below, nk2, Kd);
            RC(gemm_call(m, a1, gemm_flops(a1, nullptr), q1));
            if (k0 > 0) {
                GemmArgs b1 = upd(Acol, ld, 0, Ur, ld, 1, A + (int64_t)k1 * MOGP_TILE * ld, ld, GM_RECT, nk2, k0, Kd);
                RC(gemm_call(m, b1, gemm_flops(b1, nullptr), q1));
            }
        }
        HIP_TRY(hipStreamWaitEvent(q2, m->sw_ev[2 * kb], 0));
        const int rest = below - nk2;
        if (rest > 0) {
            const int64_t r0 = (int64_t)(k1 + nk2) * MOGP_TILE;
            const double* Xr2 = A + r0 * ld + (int64_t)k0 * MOGP_TILE;
            GemmArgs a2 = upd(Xr2, ld, 0, Uc + (int64_t)nk2 * MOGP_TILE * Kd, Kd, 0, A + r0 * (ld + 1), ld, GM_LOWER, rest, rest, Kd);
            RC(gemm_call(m, a2, gemm_flops(a2, nullptr), q2));
            if (k0 > 0) {
                GemmArgs b2 = upd(Xr2, ld, 0, Ur, ld, 1, A + r0 * ld, ld, GM_RECT, rest, k0, Kd);
                RC(gemm_call(m, b2, gemm_flops(b2, nullptr), q2));
            }
        }
        if (k0 > 0) {
            GemmArgs c = upd(Arow, ld, 1, Ur, ld, 1, A, ld, GM_LOWER, k0, k0, Kd);
            RC(gemm_call(m, c, gemm_flops(c, nullptr), q2));
        }
        HIP_TRY(hipEventRecord(m->sw_ev[2 * kb + 1], q2));                                 // bulk(kb) done
    }
    HIP_TRY(hipStreamWaitEvent(q1, m->sw_ev[2 * (nouter - 1) + 1], 0));
    return 0;
}

}  // namespace mogp
