// titsias.hip -- the Titsias sparse variational bound and its gradient on the device (BASELINE.json configs[4]).
// Reference: gpr/model.py:700-724 (elbo), :730-765 (predict_f); the gradient replaces autograd through that code.
//
// Whitened, cancellation-free forms (A = Kuu + jitter mean(diag Kuu) I = L L^T, B = Kuf, s2 = sigma^2):
//   v = L^-1 B,  Qs = v v^T / s2 + I = Lq Lq^T,  Pq = Qs^-1,  t1 = Pq (v y),  beta = L^-T t1
//   ELBO      = -N/2 log 2pi - sum log Lq_kk - N log sigma - y^T y /(2 s2) + t1.(v y) /(2 s2^2) - (sum Kff_diag - tr Q)/(2 s2)
//   dELBO/dB  = L^-T (I - Pq) v / s2 + beta (y / s2^2 - B^T beta / s2^3)^T
//   dELBO/dA  = 1/2 L^-T (2 I - Pq - Qs) L^-1 - 1/2 beta beta^T / s2^2
// Everything that goes through K_uu^-1 is a triangular SOLVE with L (trsm.hip: blocked substitution, as the reference's
// solve_triangular at gpr/model.py:711,715): K_uu has condition number ~1e11 at configs[4] and products with an explicit L^-1 lose
// dELBO/dZ (12-19 % error; 4e-5 with solves -- tools/titsias_numerics.py).  The inner system Qs = I + v v^T / s2 is well conditioned:
// its inverse Pq is formed explicitly (Cholesky, triangular inverse, W^T W on the MFMA GEMM).  The two adjoints are contracted with
// the kernel derivatives by the dense-mode moment kernel, which also accumulates the gradient w.r.t. the inducing inputs.
#include "mogp_model.h"

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <limits>

namespace mogp {
int launch_potrf_trtri_tile(double* A, int64_t ld, int t, double* invd, double* logdet, unsigned long long* info, hipStream_t s,
                            long long info_base = 0, int store_L = 0);
}
using namespace mogp;

#define RC(x) do { int r__ = (x); if (r__) return r__; } while (0)



namespace mogp {

int side_fork(mogp_model* m, TitsiasWork& t, hipStream_t* side) {
    static const bool on = !(std::getenv("MOGP_SIDE_STREAM") && std::atoi(std::getenv("MOGP_SIDE_STREAM")) == 0);
    *side = m->st;
    if (!on || !m->st3) return 0;
    for (auto& e : t.side_ev) if (!e) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    HIP_TRY(hipEventRecord(t.side_ev[0], m->st));
    HIP_TRY(hipStreamWaitEvent(m->st3, t.side_ev[0], 0));
    *side = m->st3;
    return 0;
}

int side_join(mogp_model* m, TitsiasWork& t, hipStream_t side) {
    if (side == m->st) return 0;
    HIP_TRY(hipEventRecord(t.side_ev[1], side));
    HIP_TRY(hipStreamWaitEvent(m->st, t.side_ev[1], 0));
    return 0;
}

// defer_check: the caller reads the pivot word itself at its next synchronisation (a failed factorisation then runs on with garbage: bounded, harmless)
int spd_invert(mogp_model* m, Spd& w, const char* which, int64_t* info, const double** W, bool defer_check) {
    static const bool fused = std::getenv("MOGP_SPARSE_FUSED") && std::atoi(std::getenv("MOGP_SPARSE_FUSED")) != 0;
    if (fused && w.nb <= 80) {
        RC(spd_potri_fused(m, w));
        RC(spd_potri_fused_finish(m, w));
        if (!defer_check) RC(spd_check_info(m, which, info));
        *W = w.Wm.p;
        return 0;
    }
    RC(spd_potrf(m, w));
    if (!defer_check) RC(spd_check_info(m, which, info));
    RC(spd_trtri(m, w));                                                        // w.A = L^-1
    RC(spd_lauum(m, w));                                                        // w.B = inverse (lower)
    *W = w.A.p;
    return 0;
}

int gz_prepare(mogp_model* m, TitsiasWork& t, const std::vector<int>& offz, int D) {
    std::vector<int> hz, hx;
    tile_blocks(offz, m->C, hz);
    tile_blocks(m->sx.off, m->C, hx);
    if (hz != t.hblk_z) {
        RC(t.blk_z.ensure(hz.size()));
        t.hblk_z = hz;
        HIP_TRY(hipMemcpyAsync(t.blk_z.p, t.hblk_z.data(), hz.size() * sizeof(int), hipMemcpyHostToDevice, m->st));
    }
    if (hx != t.hblk_x) {
        RC(t.blk_x.ensure(hx.size()));
        t.hblk_x = hx;
        HIP_TRY(hipMemcpyAsync(t.blk_x.p, t.hblk_x.data(), hx.size() * sizeof(int), hipMemcpyHostToDevice, m->st));
    }
    const int nbz = (int)hz.size() / 2, nbx = (int)hx.size() / 2;
    return t.gzp.ensure(gz_scratch_doubles(nbz, std::max(nbz, nbx), D));
}

void gz_attach(const TitsiasWork& t, MomentArgs& ma, bool zx) {
    ma.gzp = t.gzp.p;
    ma.nrb = (int)t.hblk_z.size() / 2; ma.rblk = t.blk_z.p;
    ma.ncb = zx ? (int)t.hblk_x.size() / 2 : ma.nrb; ma.cblk = zx ? t.blk_x.p : t.blk_z.p;
}

int spd_check_info(mogp_model* m, const char* which, int64_t* info) {
    unsigned long long hinfo = 0;
    HIP_TRY(hipMemcpyAsync(&hinfo, m->d_info.p, sizeof(hinfo), hipMemcpyDeviceToHost, m->st));
    HIP_TRY(hipStreamSynchronize(m->st));
    return spd_info_verdict(m, which, hinfo, info);
}

int spd_info_verdict(mogp_model* m, const char* which, unsigned long long hinfo, int64_t* info) {
    if (hinfo == MOGP_INFO_CHAIN_TIMEOUT) {
        m->no_chain = true;                                  // gates the chain kernel AND the stream-K launches (mogp_api.hip:stream_k_setup)
        return fail(MOGP_EHIP, "a hand-off between workgroups timed out (chain kernel, chain.hip, or a stream-K GEMM, linalg.hip:k_gemm_sk: the GPU is "
                               "shared with another process?).  The model has switched both forms off, as MOGP_CHAIN=0 and MOGP_SK=0 do: repeat the call");
    }
    if (hinfo != std::numeric_limits<unsigned long long>::max()) {
        if (info) *info = (int64_t)hinfo;
        return fail(MOGP_ENOTPD, std::string("linalg.cholesky: ") + which + " is not positive-definite (the leading minor of order " +
                                 std::to_string(hinfo) + " is not positive-definite).");
    }
    return 0;
}

int sparse_timeout_check(mogp_model* m) {
    unsigned long long hinfo = 1;
    HIP_TRY(hipMemcpy(&hinfo, m->d_info.p, sizeof(hinfo), hipMemcpyDeviceToHost));     // the caller has just synchronised the stream
    if (hinfo != MOGP_INFO_CHAIN_TIMEOUT) return 0;
    m->no_chain = true;
    return fail(MOGP_EHIP, "a hand-off between workgroups timed out during this evaluation (stream-K GEMM of a triangular solve, linalg.hip:k_gemm_sk, "
                           "or the chain kernel: the GPU is shared with another process?): its result is not valid.  The model has switched both forms "
                           "off, as MOGP_SK=0 and MOGP_CHAIN=0 do: repeat the call");
}

// mt (mt + 1) / 2 = 136 tiles at configs[4] would leave half the chip idle over K = N: K is cut into ks slices, each slice into a block of
// its own, and the blocks are summed.  ks minimises rounds(tiles ks / 512 slots) / ks: 136 tiles -> ks = 15, 2040 workgroups = four
// full rounds (two slices left every second CU with two workgroups and the rest with one: 12.5 ms; fifteen: see DESIGN 4b)
int mm_lower_splitk(mogp_model* m, TitsiasWork& t, const double* A, const double* B, double* out, int mt, int64_t Mpad, int64_t ldk, int64_t K,
                    double alpha) {
    GemmArgs g = make_gemm(A, ldk, 0, B, ldk, 0, out, Mpad, alpha, GM_LOWER, mt, mt, K);
    const int tiles_q = mt * (mt + 1) / 2;
    int ks = 1;
    static const int ks_env = []() { const char* e = std::getenv("MOGP_SYRK_KS"); return e ? atoi(e) : 0; }();      // > 0: that many slices, slice-major; < 0: |.| slices, one XCD each
    if (tiles_q < 512 && K >= 4096) {
        double best = 1e30;
        for (int c = 1; c <= 16; ++c) {
            if (K / c < 2048) break;
            const double cost = std::ceil((double)tiles_q * c / 512.0) / c;
            if (cost < best - 1e-12) { best = cost; ks = c; }
        }
        // (round 4: eight or sixteen slices, each on ONE XCD -- k_gemm: ksplit_xcd, the workgroups that share an L2 then share a k window as well --
        // measured no faster than fifteen slice-major ones, 45.4 vs 45.3 ms at configs[4], although those fetch 12.4 GB for 1.6 GB of v: the product is
        // not bound by that traffic; kept as a switch)
        if (ks_env > 0) ks = ks_env;
        if (ks_env < 0) { ks = -ks_env; g.ksplit_xcd = (ks % 8 == 0); }
    }
    if (ks > 1) {
        if (t.kslices.n < (size_t)ks * Mpad * Mpad) {         // the upper tiles are never written: keep them finite
            RC(t.kslices.ensure((size_t)ks * Mpad * Mpad));
            HIP_TRY(hipMemsetAsync(t.kslices.p, 0, (size_t)ks * Mpad * Mpad * sizeof(double), m->st));
        }
        g.C = t.kslices.p; g.ksplit = ks; g.c_split = (int64_t)Mpad * Mpad;
    }
    RC(gemm_call(m, g, gemm_flops(g, nullptr)));
    if (ks > 1) RC(launch_sum_slices(t.kslices.p, (int64_t)Mpad * Mpad, ks, out, m->st));
    return 0;
}

}  // namespace mogp

// Common front end: sort Z, build tiles, Kuu -> W (in tw.a.A), Kuf -> tw.B, v -> tw.v, Qs -> tw.Qs, Wq (tw.q.A), Pq (tw.q.B, full),
// vy, t1 in tw.vec[0 : Mpad], tw.vec[Mpad : 2 Mpad].  Host scalars through `sc`.
struct TitsiasScalars { double logdet_q, yy, t1vy, t1t1, trPq, trQs, jit, ntot, kff; };

// sharded: this handle holds ONE SHARD of the training points (the ranks of the context's communicator hold the others; Z, sigma and the
// terms are the same everywhere).  Everything that sums over data points -- v v^T, v y, y^T y, N, sum K_ff,nn -- is all-reduced, after which
// every M x M quantity is the full model's on every rank (two collectives: Mpad^2 doubles and Mpad + 3).
static int titsias_front(mogp_model* m, int64_t M, const double* Z, double sigma, double jitter, SortedX& sz,
                         std::vector<GTile>& tuu, std::vector<int>& psuu, std::vector<GTile>& tuf, std::vector<int>& psuf,
                         TitsiasScalars& sc, int64_t* info, bool need_moment_tiles, bool sharded = false, const double* kff_diag = nullptr) {
    const int C = m->C, D = m->D, W = m->Wt;                   // 2 + 3 D, or 2 + 5 D: terms with an envelope on the input midpoint (MOHSM)
    const bool env = W > 2 + 3 * D;
    const int64_t Npad = m->Npad;
    if (m->T <= 0) return fail(MOGP_EINVAL, "mogp_model_set_terms must be called before an evaluation");
    if (!(sigma > 0.0)) return fail(MOGP_EINVAL, "sigma must be positive");
    RC(sort_inputs(Z, M, D, C, MOGP_TILE, sz));
    const int64_t Mpad = sz.Mpad;
    if (!m->tw) m->tw = new TitsiasWork();
    TitsiasWork& t = *m->tw;
    const int mt = (int)(Mpad / MOGP_TILE), nt = (int)(Npad / MOGP_TILE);
    if (t.Mpad != Mpad) {
        t.Mpad = Mpad;
        RC(spd_alloc(t.a, Mpad)); RC(spd_alloc(t.q, Mpad));
        RC(t.zx.ensure((size_t)D * Mpad));
        RC(t.B.ensure((size_t)Mpad * Npad)); RC(t.v.ensure((size_t)Mpad * Npad));
        RC(t.Qs.ensure((size_t)Mpad * Mpad));
        RC(t.vec.ensure((size_t)8 * Mpad + 4 * Npad));
        RC(t.scratch.ensure((size_t)(Mpad / 256 + 2) * std::max(Npad, Mpad) + (size_t)(Mpad / 512 + 2) * Mpad));
        RC(t.zero_noise.ensure(C));
        { int r__ = dev_fill_zero(t.zero_noise.p, C * sizeof(double)); if (r__) return r__; }
        { int r__ = dev_fill_zero(t.B.p, (size_t)Mpad * Npad * sizeof(double)); if (r__) return r__; }      // padding of Kuf stays zero: the Gram kernel never writes it
        { int r__ = dev_fill_zero(t.v.p, (size_t)Mpad * Npad * sizeof(double)); if (r__) return r__; }      // ... nor that of its working copy (the solves keep zeros zero)
    }
    m->gemm_ev_used = 0; m->gemm_launches = 0; m->gemm_flops = 0.0;
    HIP_TRY(hipMemcpyAsync(t.zx.p, sz.xs.data(), (size_t)D * Mpad * sizeof(double), hipMemcpyHostToDevice, m->st));
    // the tile lists depend on the channel offsets of Z and X only: built and uploaded when those change, not per evaluation (50 000 tiles and 1.2 MB
    // of pageable copies at configs[4], all of it in front of the evaluation's first kernel); the (Z, X) list also as strip-kernel runs
    std::vector<int> key(sz.off);
    key.insert(key.end(), m->sx.off.begin(), m->sx.off.end());
    if (key != t.tile_key) {
        t.tile_key.clear();
        build_sym_tiles(sz.off, C, tuu, psuu);
        build_rect_tiles(sz.off, m->sx.off, C, tuf, &psuf);
        HIP_TRY(hipStreamSynchronize(m->st));                   // a previous evaluation's kernels may still read the lists (first call / a new Z layout only)
        RC(t.tiles_uu.ensure(tuu.size())); RC(t.tiles_uf.ensure(tuf.size()));
        HIP_TRY(dev_upload(t.tiles_uu.p, tuu.data(), tuu.size() * sizeof(GTile)));
        HIP_TRY(dev_upload(t.tiles_uf.p, tuf.data(), tuf.size() * sizeof(GTile)));
        RC(t.strip_uf.build(tuf));
        RC(t.ps_uu.ensure(psuu.size())); RC(t.ps_uf.ensure(psuf.size()));
        HIP_TRY(dev_upload(t.ps_uu.p, psuu.data(), psuu.size() * sizeof(int)));
        HIP_TRY(dev_upload(t.ps_uf.p, psuf.data(), psuf.size() * sizeof(int)));
        t.n_tuu = tuu.size(); t.n_tuf = tuf.size();
        t.tile_key = key;
    }
    if (need_moment_tiles) {
        RC(t.partial_uu.ensure(t.n_tuu * (size_t)m->T * W)); RC(t.partial_uf.ensure(t.n_tuf * (size_t)m->T * W));
        RC(t.mom_uu.ensure((size_t)(C * (C + 1) / 2) * m->T * W)); RC(t.mom_uf.ensure((size_t)C * C * m->T * W));
    }
    const unsigned long long big = std::numeric_limits<unsigned long long>::max();
    HIP_TRY(hipMemcpyAsync(m->d_info.p, &big, sizeof(big), hipMemcpyHostToDevice, m->st));

    // relative jitter on Kuu (reference gpr/model.py:710 -> :244)
    sc.jit = jitter * table_diag_points(m, sz) / (double)M;    // with an envelope the diagonal of Kuu follows the inducing inputs

    GramArgs ga{};
    ga.tiles = t.tiles_uu.p; ga.xr = t.zx.p; ga.xc = t.zx.p; ga.ldxr = ga.ldxc = Mpad; ga.nrows = ga.ncols = M;
    RC(t.ph_zz.prepare(sz.off, sz.off, C, m->T, Mpad, Mpad, m->st, ga.ph));
    ga.table = m->d_table.p; ga.T = m->T; ga.D = D; ga.C = C; ga.W = W; ga.out = t.a.A.p; ga.ldo = Mpad;
    ga.noise = t.zero_noise.p; ga.dvar = nullptr; ga.jitter_abs = sc.jit; ga.mirror = 0;
    RC(launch_gram(ga, (int)t.n_tuu, m->st));
    RC(launch_pad_identity(t.a.A.p, Mpad, M, Mpad, m->st));
    // Kuf and its working copy on the side stream, underneath the (chain-bound) factorisation of Kuu
    hipStream_t side;
    RC(side_fork(m, t, &side));
    ga.tiles = t.tiles_uf.p; ga.xc = m->d_x.p; ga.ldxc = Npad; ga.ncols = m->N; ga.out = t.B.p; ga.ldo = Npad; ga.noise = nullptr; ga.jitter_abs = 0.0;
    ga.out2 = t.v.p;                                                            // the copy the solve below works in, written by the same kernel
    t.strip_uf.attach(ga);                                                      // full interior tiles in runs of four on the strip kernel
    RC(t.ph_zx.prepare(sz.off, m->sx.off, C, m->T, Mpad, Npad, side, ga.ph));
    RC(launch_gram(ga, (int)t.n_tuf, side));

    t.a.keep_L = true;                                                          // the solves below need L itself, diagonal tiles included
    t.a.refine_panels = !(std::getenv("MOGP_REFINE_PANELS") && std::atoi(std::getenv("MOGP_REFINE_PANELS")) == 0);   // K_uu + jitter is ill-conditioned: mogp_api.hip:spd_potrf
    // (round 5, measured and dropped: the chain of this factorisation on the CU-masked private stream, so that its one-workgroup kernels do not share
    // a CU with the K_uf Gram waves -- configs[4] 39.7-39.9 vs 39.3-39.5 ms, bit-identical: the panel and next-column products of the chain are
    // slower on 16 CUs than the leaves gain)
    // Round 6: v = L^-1 B FOLLOWS the factorisation block row by block row instead of waiting for all of it.  Block row i of L is final behind the leaf of
    // tile column i (right-looking: the panels left of it were finished by the columns before), and the left-looking substitution needs exactly that row for
    // its step i -- so the substitution (M^2 N flop, 9.3 ms at configs[4]) runs on the side stream behind the K_uf Gram, waiting per step for the event the
    // factorisation records, and the factorisation's 16 x 7 dependent small launches (3.6 ms by themselves) run underneath it: the side stream is masked off
    // the reserved CUs, so the chain's one-workgroup kernels always find a free CU.  Measured at configs[4] (profiles/r6_cfg5_follow.txt): 38.3 against 39.7 ms
    // per step, ELBO and kernel gradients unchanged -- but NOT the default: on the masked stream the substitution's stream-K products split their k range over
    // 480 workgroups instead of 512, a different summation order, and dELBO/dZ -- which sits at the rounding floor of this conditioning (cond K_uu ~ 1e11) --
    // lands 2.62e-3 of the tensor from the 80-bit truth instead of 1.59e-3 (the reference's two runs: 1.44e-3, 1.93e-3).  Both are inside the scatter of the
    // reference itself; the test holds the device to "no further away than the reference", so the schedule that is there stays.  MOGP_TITSIAS_FOLLOW=1 switches this on.
    static const bool follow = std::getenv("MOGP_TITSIAS_FOLLOW") && std::atoi(std::getenv("MOGP_TITSIAS_FOLLOW")) != 0;
    const bool fl = follow && side != m->st && Npad / MOGP_TILE > 32;
    t.a.want_row_ev = fl;
    RC(spd_potrf(m, t.a));                                                      // its pivot report is read with the scalars at the end of this function
    t.a.want_row_ev = false;
    const double s2 = sigma * sigma;
    if (fl) {
        RC(trsm_lower(m, t.a.A.p, Mpad, mt, t.v.p, Npad, Npad, false, side, false, t.a.row_ev.data()));
        RC(side_join(m, t, side));
    } else {
        RC(side_join(m, t, side));
        RC(trsm_lower(m, t.a.A.p, Mpad, mt, t.v.p, Npad, Npad, false));          // v = L^-1 B   (reference gpr/model.py:711)
    }
    (void)nt;
    // Qs = v v^T / s2 + I, with v y (one memory-bound pass over v) underneath the compute-bound product
    double* vy = t.vec.p;
    RC(side_fork(m, t, &side));
    RC(launch_gemv_rows(t.v.p, Npad, Mpad, Npad, m->d_y.p, vy, side));
    RC(mm_lower_splitk(m, t, t.v.p, t.v.p, t.q.A.p, mt, Mpad, Npad, Npad, 1.0 / s2));
    RC(side_join(m, t, side));
    sc.yy = 0.0; for (int64_t i = 0; i < m->N; ++i) sc.yy += m->hy[i] * m->hy[i];
    sc.ntot = (double)m->N;
    sc.kff = 0.0;
    if (kff_diag && env) for (int64_t i = 0; i < m->N; ++i) sc.kff += kff_diag[i];                  // per training point
    else if (kff_diag) for (int c = 0; c < C; ++c) sc.kff += (double)(m->sx.off[c + 1] - m->sx.off[c]) * kff_diag[c];
    if (sharded) {
        RC(comm_allreduce(m->ctx, t.q.A.p, Mpad * Mpad, m->st));
        RC(t.red.ensure((size_t)Mpad + 4));
        double hs[3] = {sc.yy, sc.ntot, sc.kff};
        HIP_TRY(hipMemcpyAsync(t.red.p, vy, Mpad * sizeof(double), hipMemcpyDeviceToDevice, m->st));
        HIP_TRY(hipMemcpyAsync(t.red.p + Mpad, hs, sizeof(hs), hipMemcpyHostToDevice, m->st));
        HIP_TRY(hipStreamSynchronize(m->st));                                   // hs is a stack buffer
        RC(comm_allreduce(m->ctx, t.red.p, Mpad + 3, m->st));
        HIP_TRY(hipMemcpyAsync(vy, t.red.p, Mpad * sizeof(double), hipMemcpyDeviceToDevice, m->st));
        HIP_TRY(hipMemcpyAsync(hs, t.red.p + Mpad, sizeof(hs), hipMemcpyDeviceToHost, m->st));
        HIP_TRY(hipStreamSynchronize(m->st));
        sc.yy = hs[0]; sc.ntot = hs[1]; sc.kff = hs[2];
    }
    RC(launch_add_diag(t.q.A.p, Mpad, Mpad, 1.0, m->st));
    HIP_TRY(hipMemcpyAsync(t.Qs.p, t.q.A.p, (size_t)Mpad * Mpad * sizeof(double), hipMemcpyDeviceToDevice, m->st));
    RC(launch_info_stash(m->d_info.p, m->st));              // Kuu's report (and a time-out of the wide solve above) -> info[1]; info[0] armed for Qs
    RC(spd_invert(m, t.q, "Q/sigma^2 + I", info, &t.Wq, true));                 // t.Wq = Lq^-1, t.q.B = Pq (lower)
    RC(launch_symmetrize(t.q.B.p, Mpad, Mpad, m->st));
    double* t1 = t.vec.p + Mpad;
    double* dg = t.vec.p + 2 * Mpad;                                            // diag Pq, diag Qs
    // t1 = Pq (v y): the explicit inverse plus ONE step of iterative refinement against Qs.  Of the three places Pq enters the gradient,
    // this vector is the one where the explicitly formed inverse costs accuracy on dELBO/dZ (tools/titsias_numerics.py: 1.3e-4 -> 7e-5, the
    // same as two triangular solves with Lq), and three M x M mat-vecs are far cheaper than 2 nb dependent launches of a vector solve
    RC(launch_symmetrize(t.Qs.p, Mpad, Mpad, m->st));
    {
        double* tmp = t.vec.p + 5 * Mpad;
        double* res = t.vec.p + 6 * Mpad;
        RC(launch_gemv_rows(t.q.B.p, Mpad, Mpad, Mpad, vy, t1, m->st));
        RC(launch_gemv_rows(t.Qs.p, Mpad, Mpad, Mpad, t1, tmp, m->st));
        RC(launch_axpby(Mpad, 1.0, vy, -1.0, tmp, res, m->st));
        RC(launch_gemv_rows(t.q.B.p, Mpad, Mpad, Mpad, res, tmp, m->st));
        RC(launch_axpby(Mpad, 1.0, t1, 1.0, tmp, t1, m->st));
    }
    RC(launch_get_diag(t.q.B.p, Mpad, Mpad, dg, m->st));
    RC(launch_get_diag(t.Qs.p, Mpad, Mpad, dg + Mpad, m->st));
    const int nbq = t.q.nb;
    std::vector<double> hv((size_t)4 * Mpad), hl(nbq);
    HIP_TRY(hipMemcpyAsync(hv.data(), t.vec.p, (size_t)4 * Mpad * sizeof(double), hipMemcpyDeviceToHost, m->st));
    HIP_TRY(hipMemcpyAsync(hl.data(), t.q.logdet.p, nbq * sizeof(double), hipMemcpyDeviceToHost, m->st));
    unsigned long long hinfo[2] = {0, 0};
    HIP_TRY(hipMemcpyAsync(hinfo, m->d_info.p, sizeof(hinfo), hipMemcpyDeviceToHost, m->st));
    HIP_TRY(hipStreamSynchronize(m->st));
    RC(spd_info_verdict(m, "Kuu", hinfo[1], info));                             // the one synchronisation of the forward part: both factorisations' reports
    RC(spd_info_verdict(m, "Q/sigma^2 + I", hinfo[0], info));
    sc.logdet_q = 0.0; for (double x : hl) sc.logdet_q += x;
    sc.t1vy = sc.t1t1 = sc.trPq = sc.trQs = 0.0;
    for (int64_t i = 0; i < M; ++i) {
        sc.t1vy += hv[Mpad + i] * hv[i]; sc.t1t1 += hv[Mpad + i] * hv[Mpad + i];
        sc.trPq += hv[2 * Mpad + i]; sc.trQs += hv[3 * Mpad + i];
    }
    return 0;
}

static int titsias_eval_impl(mogp_model* m, int64_t M, const double* Z, double sigma, double jitter, const double* kff_diag, int flags,
                             double* elbo, double* mom_uu, double* mom_uf, double* gZ, double* trGA, double* dsigma,
                             double* jitter_abs, int64_t* info, bool sharded) {
    if (!m || !Z || !kff_diag || !elbo || M <= 0) return fail(MOGP_EINVAL, "mogp_titsias_eval: bad argument");
    RC(use_device(m->ctx));
    if (info) *info = 0;
    const int C = m->C, D = m->D, W = m->Wt, T = m->T, P = C * (C + 1) / 2;
    const int64_t N = m->N, Npad = m->Npad;
    const bool grad = (flags & MOGP_EVAL_GRAD) != 0;
    SortedX sz;
    std::vector<GTile> tuu, tuf;
    std::vector<int> psuu, psuf;
    TitsiasScalars sc;
    RC(titsias_front(m, M, Z, sigma, jitter, sz, tuu, psuu, tuf, psuf, sc, info, grad, sharded, kff_diag));
    TitsiasWork& t = *m->tw;
    const int64_t Mpad = t.Mpad;
    const int mt = (int)(Mpad / MOGP_TILE), nt = (int)(Npad / MOGP_TILE);
    const double s2 = sigma * sigma;
    const double kff = sc.kff, Ntot = sc.ntot;                   // sums over ALL data points (all-reduced when the data is sharded)
    const double trQ = s2 * (sc.trQs - (double)M);
    *elbo = -0.5 * Ntot * std::log(2.0 * M_PI) - sc.logdet_q - Ntot * std::log(sigma) - 0.5 * sc.yy / s2
            + 0.5 * sc.t1vy / (s2 * s2) - 0.5 * (kff - trQ) / s2;
    if (jitter_abs) *jitter_abs = sc.jit;
    if (!grad) return sparse_timeout_check(m);
    if (!mom_uu || !mom_uf || !gZ || !trGA || !dsigma) return fail(MOGP_EINVAL, "mogp_titsias_eval: gradient outputs are null");

    // d ELBO / d s2, then d sigma  (tr(Pq Q) = s2 (M - tr Pq);  vy^T Pq Q Pq vy = s2 (t1.vy - t1.t1))
    const double ds2 = -0.5 * Ntot / s2 + 0.5 * ((double)M - sc.trPq) / s2 + 0.5 * sc.yy / (s2 * s2) - sc.t1vy / (s2 * s2 * s2)
                       + 0.5 * (sc.t1vy - sc.t1t1) / (s2 * s2 * s2) + 0.5 * (kff - trQ) / (s2 * s2);
    *dsigma = 2.0 * sigma * ds2;

    RC(t.GB.ensure((size_t)Mpad * Npad)); RC(t.E.ensure((size_t)Mpad * Mpad)); RC(t.R.ensure((size_t)Mpad * Mpad));
    RC(t.GA.ensure((size_t)Mpad * Mpad));
    RC(t.gz.ensure((size_t)D * Mpad));
    if (t.zero_col.n < (size_t)Mpad) { RC(t.zero_col.ensure(Mpad)); HIP_TRY(hipMemsetAsync(t.zero_col.p, 0, Mpad * sizeof(double), m->st)); }
    double* vy = t.vec.p; (void)vy;
    double* t1 = t.vec.p + Mpad;
    double* beta = t.vec.p + 4 * Mpad;            // [Mpad] (+ chunk scratch behind it, see launch_trmv_lower_t)
    double* dga = t.vec.p + 2 * Mpad;             // reuse: diag of GA
    double* btb = t.vec.p + 8 * Mpad;             // [Npad]
    double* r = btb + Npad;                       // [Npad]
    // GA = 1/2 L^-T E L^-1 (symmetric), E = 2I - Pq - Qs: T1 = L^-T E in place, then L^-T T1^T, then the lower triangle of the symmetrised half.
    // M x M only, underneath the M x N work below.  Round 6 (experiment, see prod_aside): the adjoint chain (~70 launches of 4 .. 200 workgroups) stays on the MAIN stream (highest priority) and the
    // M x M x N product goes to the all-CU stream of normal priority instead -- on a lower-priority side stream (rounds 3 - 5) a 4-workgroup leaf that arrived while the
    // product's 12 000 workgroups were queued was not dispatched until the product had drained: 12.8 ms for a 67 us kernel in EVERY configs[4] evaluation
    // (profiles/r5_cfg5_bench_kernel_stats.csv: max 12.9 ms, median 67 us), i.e. the chain ran BEHIND the product, not under it.  MOGP_SIDE_URGENT=1 switches this on.
    static const bool prod_aside = std::getenv("MOGP_SIDE_URGENT") && std::atoi(std::getenv("MOGP_SIDE_URGENT")) != 0;      // measured: 42.8 vs 40.0 ms -- the chain is hidden, the product and the memory-bound passes behind the chain slow each other by more.  OFF by default
    // MOGP_SIDE_URGENT=2 (round 6, second session): the two on DISJOINT CUs -- the product on the bulk stream (masked off the reserved CUs), the chain on the private
    // stream (the reserved CUs only): neither priority nor slots are shared.  Same tiles, same bits.
    static const int urgent_mode = std::getenv("MOGP_SIDE_URGENT") ? std::atoi(std::getenv("MOGP_SIDE_URGENT")) : 0;
    const bool disjoint = urgent_mode == 2 && m->st_priv && m->st2;
    const bool aside = (prod_aside && m->st2u != nullptr) || disjoint;
    hipStream_t prod_q = disjoint ? m->st2 : m->st2u;
    hipStream_t side = m->st;
    if (disjoint) {
        for (auto& e : t.side_ev) if (!e) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        HIP_TRY(hipEventRecord(t.side_ev[0], m->st));
        HIP_TRY(hipStreamWaitEvent(m->st_priv, t.side_ev[0], 0));
        side = m->st_priv;
    } else if (!aside) RC(side_fork(m, t, &side));
    // GB = L^-T (R v) / s2, and beta = L^-T t1 riding along: when N is not a multiple of 128 the right-hand side has zero padding columns,
    // and t1 travels through the blocked solve in the first of them (a vector solve of its own is 2 nb dependent, almost empty launches).
    // The large product is ENQUEUED before the side chain's ~70 launches, so that a slow host does not hold it back.
    RC(launch_combine(t.R.p, t.q.B.p, nullptr, Mpad, Mpad, 1.0, 1.0, 0.0, m->st));           // R = I - Pq
    // Round 4: GB = (L^-T R) v / s2 -- the M x M solve FIRST (16 block rows of a 2048-column right-hand side), then ONE M x M x N product, instead of
    // the product followed by the M x N solve (8.6 of configs[4]'s 45 ms).  This is not the explicit-inverse product that cost round 1 its dELBO/dZ:
    // L^-T R comes from a backward-stable substitution, v is the accurately solved L^-1 B, and the product is the LAST step before the contraction
    // with the kernel derivatives -- no solve behind it amplifies its rounding.  Against the 80-bit truth dELBO/dZ is 2.17e-3 of the tensor off
    // (M x N solve: 2.37e-3, the reference's fp64: 2.40e-3); kernel gradients 9.6e-9 (9.2e-9).  MOGP_TITSIAS_GB=0: the M x N solve.
    static const bool gb_first = !(std::getenv("MOGP_TITSIAS_GB") && atoi(std::getenv("MOGP_TITSIAS_GB")) == 0);
    if (gb_first) RC(trsm_lower(m, t.a.A.p, Mpad, mt, t.R.p, Mpad, Mpad, true));
    GemmArgs g = make_gemm(t.R.p, Mpad, 0, t.v.p, Npad, 1, t.GB.p, Npad, 1.0 / s2, GM_RECT, mt, nt, Mpad);
    // tiles down the columns: the 64 workgroups an XCD holds are then 64 / mt column blocks of v against all of R (M x M: cache-resident), and v comes
    // in from memory once -- row by row it came in mt times (27 GB fetched at configs[4] for a 1.6 GB panel)
    static const bool rv_cols = !(std::getenv("MOGP_RV_COLS") && atoi(std::getenv("MOGP_RV_COLS")) == 0);
    g.col_major = rv_cols ? 1 : 0;
    if (aside) {
        for (auto& e : t.prod_ev) if (!e) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        HIP_TRY(hipEventRecord(t.prod_ev[0], m->st));
        HIP_TRY(hipStreamWaitEvent(prod_q, t.prod_ev[0], 0));
        RC(gemm_call(m, g, gemm_flops(g, nullptr), prod_q));
        HIP_TRY(hipEventRecord(t.prod_ev[1], prod_q));        // (the main stream waits for it right in front of the (Z, X) moment pass, its first reader)
    } else
    RC(gemm_call(m, g, gemm_flops(g, nullptr)));
    RC(launch_combine(t.E.p, t.q.B.p, t.Qs.p, Mpad, Mpad, 2.0, 1.0, 1.0, side));
    RC(trsm_lower(m, t.a.A.p, Mpad, mt, t.E.p, Mpad, Mpad, true, side));
    RC(launch_transpose(t.GA.p, t.E.p, Mpad, Mpad, side));
    RC(trsm_lower(m, t.a.A.p, Mpad, mt, t.GA.p, Mpad, Mpad, true, side));
    RC(launch_sym_lower_avg(t.GA.p, Mpad, Mpad, 0.5, side));
    RC(launch_get_diag(t.GA.p, Mpad, Mpad, dga, side));
    const bool ride = Npad > N && !gb_first;
    if (ride) RC(launch_copy2d(t.GB.p + N, Npad, t1, 1, Mpad, 1, 1.0, m->st));
    if (!gb_first) RC(trsm_lower(m, t.a.A.p, Mpad, mt, t.GB.p, Npad, Npad, true));
    if (ride) {
        RC(launch_copy2d(beta, 1, t.GB.p + N, Npad, Mpad, 1, 1.0, m->st));
        RC(launch_copy2d(t.GB.p + N, Npad, t.zero_col.p, 1, Mpad, 1, 1.0, m->st));          // the padding column is zero again
    } else {
        RC(t.Hm.ensure((size_t)Mpad * MOGP_TILE));
        HIP_TRY(hipMemsetAsync(t.Hm.p, 0, (size_t)Mpad * MOGP_TILE * sizeof(double), m->st));
        RC(launch_copy2d(t.Hm.p, MOGP_TILE, t1, 1, Mpad, 1, 1.0, m->st));
        RC(trsm_lower(m, t.a.A.p, Mpad, mt, t.Hm.p, MOGP_TILE, MOGP_TILE, true));
        RC(launch_copy2d(beta, 1, t.Hm.p, MOGP_TILE, Mpad, 1, 1.0, m->st));
    }
    RC(launch_gemv_cols(t.B.p, Npad, Mpad, Npad, beta, btb, t.scratch.p, m->st));                           // B^T beta
    RC(launch_axpby(Npad, 1.0 / (s2 * s2), m->d_y.p, -1.0 / (s2 * s2 * s2), btb, r, m->st));
    HIP_TRY(hipMemsetAsync(t.gz.p, 0, (size_t)D * Mpad * sizeof(double), m->st));
    RC(gz_prepare(m, t, sz.off, D));

    MomentArgs ma{};
    gz_attach(t, ma, true);
    ma.tiles = t.tiles_uf.p; ma.ntiles = (int)t.n_tuf; ma.x = t.zx.p; ma.ldx = Mpad; ma.xc = m->d_x.p; ma.ldxc = Npad;
    ma.nrows = M; ma.ncols = N;
    RC(t.ph_zx.prepare(sz.off, m->sx.off, C, T, Mpad, Npad, m->st, ma.ph));
    ma.table = m->d_table.p; ma.T = T; ma.D = D; ma.C = C; ma.W = W;
    ma.G = t.GB.p; ma.ldg = Npad; ma.ru = beta; ma.rw = r; ma.rcoef = 1.0; ma.sym = 0;
    ma.gzr = t.gz.p; ma.gzc = nullptr; ma.ldgz = Mpad; ma.partial = t.partial_uf.p;
    if (aside) HIP_TRY(hipStreamWaitEvent(m->st, t.prod_ev[1], 0));       // GB is there
    RC(launch_moments(ma, m->st));
    RC(launch_moment_reduce(t.partial_uf.p, t.ps_uf.p, C * C, T, W, D, t.mom_uf.p, m->st, 0));
    if (sharded) {                       // the (Z, X) moments and their share of d/dZ are sums over data points; the (Z, Z) pass below is not
        RC(comm_allreduce(m->ctx, t.mom_uf.p, (int64_t)C * C * T * W, m->st));
        RC(comm_allreduce(m->ctx, t.gz.p, (int64_t)D * Mpad, m->st));
    }
    RC(side_join(m, t, side));
    ma.tiles = t.tiles_uu.p; ma.ntiles = (int)t.n_tuu; ma.xc = nullptr; ma.ldxc = 0; ma.ncols = M;
    RC(t.ph_zz.prepare(sz.off, sz.off, C, T, Mpad, Mpad, m->st, ma.ph));
    ma.G = t.GA.p; ma.ldg = Mpad; ma.ru = beta; ma.rw = beta; ma.rcoef = -0.5 / (s2 * s2); ma.sym = 1;
    ma.gzr = t.gz.p; ma.gzc = t.gz.p; ma.partial = t.partial_uu.p;
    gz_attach(t, ma, false);
    RC(launch_moments(ma, m->st));
    RC(launch_moment_reduce(t.partial_uu.p, t.ps_uu.p, P, T, W, D, t.mom_uu.p, m->st, 1));

    std::vector<double> hgz((size_t)D * Mpad), hb(Mpad), hd(Mpad);
    HIP_TRY(hipMemcpyAsync(mom_uu, t.mom_uu.p, (size_t)P * T * W * sizeof(double), hipMemcpyDeviceToHost, m->st));
    HIP_TRY(hipMemcpyAsync(mom_uf, t.mom_uf.p, (size_t)C * C * T * W * sizeof(double), hipMemcpyDeviceToHost, m->st));
    HIP_TRY(hipMemcpyAsync(hgz.data(), t.gz.p, hgz.size() * sizeof(double), hipMemcpyDeviceToHost, m->st));
    HIP_TRY(hipMemcpyAsync(hb.data(), beta, Mpad * sizeof(double), hipMemcpyDeviceToHost, m->st));
    HIP_TRY(hipMemcpyAsync(hd.data(), dga, Mpad * sizeof(double), hipMemcpyDeviceToHost, m->st));
    HIP_TRY(hipStreamSynchronize(m->st));
    RC(sparse_timeout_check(m));
    for (int64_t pos = 0; pos < M; ++pos)
        for (int d = 0; d < D; ++d) gZ[sz.perm[pos] * D + d] = hgz[(size_t)d * Mpad + pos];
    double tr = 0.0;
    for (int64_t i = 0; i < M; ++i) tr += hd[i] - 0.5 * hb[i] * hb[i] / (s2 * s2);
    *trGA = tr;
    (void)N;
    return MOGP_OK;
}

static int titsias_predict_impl(mogp_model* m, int64_t M, const double* Z, double sigma, double jitter, const double* kss_diag,
                                int64_t S, const double* Xs, double* mu, double* var, int64_t* info, bool sharded);

extern "C" {

int mogp_titsias_eval(mogp_model* m, int64_t M, const double* Z, double sigma, double jitter, const double* kff_diag, int flags,
                      double* elbo, double* mom_uu, double* mom_uf, double* gZ, double* trGA, double* dsigma,
                      double* jitter_abs, int64_t* info) {
    return titsias_eval_impl(m, M, Z, sigma, jitter, kff_diag, flags, elbo, mom_uu, mom_uf, gZ, trGA, dsigma, jitter_abs, info, false);
}

int mogp_titsias_fetch(mogp_model* m, int which, int64_t M, double* out) {
    if (!m || !out || which < 0 || which > 8 || M <= 0) return fail(MOGP_EINVAL, "mogp_titsias_fetch: bad argument");
    if (!m->tw || m->tw->Mpad <= 0 || m->tw->GA.n == 0) return fail(MOGP_EINVAL, "mogp_titsias_fetch: no gradient evaluation of the Titsias bound on this handle yet");
    RC(use_device(m->ctx));
    TitsiasWork& t = *m->tw;
    const int64_t Mpad = t.Mpad, Npad = m->Npad, N = m->N;
    if (M > Mpad) return fail(MOGP_EINVAL, "mogp_titsias_fetch: M is larger than the last evaluation's");
    HIP_TRY(hipStreamSynchronize(m->st));
    const double* src = nullptr; int64_t rows = 0, cols = 0, ld = 0;
    switch (which) {
        case 0: src = t.GA.p; rows = M; cols = M; ld = Mpad; break;
        case 1: src = t.GB.p; rows = M; cols = N; ld = Npad; break;
        case 2: src = t.vec.p + 4 * Mpad; rows = 1; cols = M; ld = Mpad; break;
        case 3: src = t.vec.p + 8 * Mpad + Npad; rows = 1; cols = N; ld = Npad; break;
        case 4: src = t.v.p; rows = M; cols = N; ld = Npad; break;
        case 5: src = t.a.A.p; rows = M; cols = M; ld = Mpad; break;
        case 6: src = t.Qs.p; rows = M; cols = M; ld = Mpad; break;
        case 7: src = t.q.B.p; rows = M; cols = M; ld = Mpad; break;
        default: src = t.vec.p + Mpad; rows = 1; cols = M; ld = Mpad; break;
    }
    HIP_TRY(hipMemcpy2D(out, (size_t)cols * sizeof(double), src, (size_t)ld * sizeof(double), (size_t)cols * sizeof(double), (size_t)rows, hipMemcpyDeviceToHost));
    return MOGP_OK;
}

int mogp_titsias_eval_sharded(mogp_model* m, int64_t M, const double* Z, double sigma, double jitter, const double* kff_diag, int flags,
                              double* elbo, double* mom_uu, double* mom_uf, double* gZ, double* trGA, double* dsigma,
                              double* jitter_abs, int64_t* info) {
    return titsias_eval_impl(m, M, Z, sigma, jitter, kff_diag, flags, elbo, mom_uu, mom_uf, gZ, trGA, dsigma, jitter_abs, info, true);
}

int mogp_titsias_predict(mogp_model* m, int64_t M, const double* Z, double sigma, double jitter, const double* kss_diag,
                         int64_t S, const double* Xs, double* mu, double* var, int64_t* info) {
    return titsias_predict_impl(m, M, Z, sigma, jitter, kss_diag, S, Xs, mu, var, info, false);
}

int mogp_titsias_predict_sharded(mogp_model* m, int64_t M, const double* Z, double sigma, double jitter, const double* kss_diag,
                                 int64_t S, const double* Xs, double* mu, double* var, int64_t* info) {
    return titsias_predict_impl(m, M, Z, sigma, jitter, kss_diag, S, Xs, mu, var, info, true);
}

}  // extern "C"

static int titsias_predict_impl(mogp_model* m, int64_t M, const double* Z, double sigma, double jitter, const double* kss_diag,
                                int64_t S, const double* Xs, double* mu, double* var, int64_t* info, bool sharded) {
    if (!m || !Z || !kss_diag || !Xs || !mu || !var || M <= 0 || S <= 0) return fail(MOGP_EINVAL, "mogp_titsias_predict: bad argument");
    RC(use_device(m->ctx));
    if (info) *info = 0;
    const int C = m->C, D = m->D;
    const bool env = m->Wt > 2 + 3 * D;
    SortedX sz, ss;
    std::vector<GTile> tuu, tuf, tus;
    std::vector<int> psuu, psuf;
    TitsiasScalars sc;
    RC(titsias_front(m, M, Z, sigma, jitter, sz, tuu, psuu, tuf, psuf, sc, info, false, sharded));
    TitsiasWork& t = *m->tw;
    const int64_t Mpad = t.Mpad;
    const double s2 = sigma * sigma;
    RC(sort_inputs(Xs, S, D, C, MOGP_TILE, ss));
    t.pred_valid = false;
    const int64_t Spad = ss.Mpad;
    const int mt = (int)(Mpad / MOGP_TILE), st = (int)(Spad / MOGP_TILE);
    build_rect_tiles(sz.off, ss.off, C, tus);
    RC(t.Kus.ensure((size_t)Mpad * Spad)); RC(t.Aus.ensure((size_t)Mpad * Spad)); RC(t.Bus.ensure((size_t)Mpad * Spad));
    RC(m->d_xs.ensure((size_t)D * Spad)); RC(m->d_ptiles.ensure(tus.size()));
    RC(m->d_mu.ensure(Spad)); RC(m->d_var.ensure(2 * Spad));
    HIP_TRY(hipMemcpyAsync(m->d_xs.p, ss.xs.data(), (size_t)D * Spad * sizeof(double), hipMemcpyHostToDevice, m->st));
    HIP_TRY(hipMemcpyAsync(m->d_ptiles.p, tus.data(), tus.size() * sizeof(GTile), hipMemcpyHostToDevice, m->st));
    HIP_TRY(hipMemsetAsync(t.Kus.p, 0, (size_t)Mpad * Spad * sizeof(double), m->st));
    GramArgs ga{};
    ga.tiles = m->d_ptiles.p; ga.xr = t.zx.p; ga.ldxr = Mpad; ga.xc = m->d_xs.p; ga.ldxc = Spad; ga.nrows = M; ga.ncols = S;
    RC(t.ph_zs.prepare(sz.off, ss.off, C, m->T, Mpad, Spad, m->st, ga.ph));
    ga.table = m->d_table.p; ga.T = m->T; ga.D = D; ga.C = C; ga.W = m->Wt; ga.out = t.Kus.p; ga.ldo = Spad; ga.mirror = 0;
    RC(launch_gram(ga, (int)tus.size(), m->st));
    HIP_TRY(hipMemcpyAsync(t.Aus.p, t.Kus.p, (size_t)Mpad * Spad * sizeof(double), hipMemcpyDeviceToDevice, m->st));
    RC(trsm_lower(m, t.a.A.p, Mpad, mt, t.Aus.p, Spad, Spad, false));                                      // a = L^-1 Kus
    GemmArgs g = make_gemm(t.Wq, Mpad, 0, t.Aus.p, Spad, 1, t.Bus.p, Spad, 1.0, GM_KHI_I, mt, st, Mpad);                    // b = Wq a
    RC(gemm_call(m, g, gemm_flops(g, nullptr)));
    double* vy = t.vec.p;
    double* cvec = t.vec.p + 4 * Mpad;
    RC(launch_trmv_lower(t.Wq, Mpad, Mpad, vy, cvec, t.vec.p + 6 * Mpad, m->st));                             // c s2 = Wq vy
    RC(launch_gemv_cols(t.Bus.p, Spad, Mpad, Spad, cvec, m->d_mu.p, t.scratch.p, m->st));                    // mu s2 = b^T (Wq vy)
    RC(launch_gemv_cols(t.Aus.p, Spad, Mpad, Spad, nullptr, m->d_var.p, t.scratch.p, m->st));               // colsum a^2
    RC(launch_gemv_cols(t.Bus.p, Spad, Mpad, Spad, nullptr, m->d_var.p + Spad, t.scratch.p, m->st));        // colsum b^2
    std::vector<double> hmu(Spad), hv(2 * Spad);
    HIP_TRY(hipMemcpyAsync(hmu.data(), m->d_mu.p, Spad * sizeof(double), hipMemcpyDeviceToHost, m->st));
    HIP_TRY(hipMemcpyAsync(hv.data(), m->d_var.p, 2 * Spad * sizeof(double), hipMemcpyDeviceToHost, m->st));
    HIP_TRY(hipStreamSynchronize(m->st));
    RC(sparse_timeout_check(m));
    for (int c = 0; c < C; ++c)
        for (int pos = ss.off[c]; pos < ss.off[c + 1]; ++pos) {
            mu[ss.perm[pos]] = hmu[pos] / s2;
            var[ss.perm[pos]] = (env ? kss_diag[ss.perm[pos]] : kss_diag[c]) - hv[pos] + hv[Spad + pos];     // envelope: K_ss,diag per test point
        }
    t.pred_ss = ss; t.pred_valid = true;
    return MOGP_OK;
}

extern "C" {

// Full predictive covariance of the LAST sparse prediction on this handle (mogp_titsias_predict, mogp_svgp_forward at test inputs; their
// sharded forms): K_ss - a^T a + b^T b with a = L^-1 K_us and b as that call left them (reference gpr/model.py:758-760, 870-872), S x S in
// the caller's order of the test points.  One Gram over the test inputs and two S x S x M products.
int mogp_sparse_predict_cov(mogp_model* m, int64_t S, double* cov) {
    if (!m || !cov || S <= 0) return fail(MOGP_EINVAL, "mogp_sparse_predict_cov: bad argument");
    RC(use_device(m->ctx));
    if (!m->tw || !m->tw->pred_valid || m->tw->pred_ss.M != S)
        return fail(MOGP_EINVAL, "mogp_sparse_predict_cov: no sparse prediction at S test points precedes it on this handle");
    TitsiasWork& t = *m->tw;
    const SortedX& ss = t.pred_ss;
    const int C = m->C, D = m->D;
    const int64_t Mpad = t.Mpad, Spad = ss.Mpad;
    std::vector<GTile> st_tiles;
    std::vector<int> ps;
    build_sym_tiles(ss.off, C, st_tiles, ps);
    RC(m->d_Kss.ensure((size_t)Spad * Spad)); RC(m->d_ptiles.ensure(st_tiles.size()));
    HIP_TRY(hipMemsetAsync(m->d_Kss.p, 0, (size_t)Spad * Spad * sizeof(double), m->st));
    HIP_TRY(hipMemcpyAsync(m->d_ptiles.p, st_tiles.data(), st_tiles.size() * sizeof(GTile), hipMemcpyHostToDevice, m->st));
    GramArgs ga{};
    ga.tiles = m->d_ptiles.p; ga.xr = m->d_xs.p; ga.ldxr = Spad; ga.xc = m->d_xs.p; ga.ldxc = Spad; ga.nrows = S; ga.ncols = S;
    RC(m->ph_ss.prepare(ss.off, ss.off, C, m->T, Spad, Spad, m->st, ga.ph));
    ga.table = m->d_table.p; ga.T = m->T; ga.D = D; ga.C = C; ga.W = m->Wt; ga.out = m->d_Kss.p; ga.ldo = Spad; ga.mirror = 1;
    RC(launch_gram(ga, (int)st_tiles.size(), m->st));
    const int st = (int)(Spad / MOGP_TILE);
    GemmArgs g = make_gemm(t.Aus.p, Spad, 1, t.Aus.p, Spad, 1, m->d_Kss.p, Spad, -1.0, GM_RECT, st, st, Mpad);
    g.beta = 1.0;
    RC(gemm_call(m, g, gemm_flops(g, nullptr)));
    g = make_gemm(t.Bus.p, Spad, 1, t.Bus.p, Spad, 1, m->d_Kss.p, Spad, 1.0, GM_RECT, st, st, Mpad);
    g.beta = 1.0;
    RC(gemm_call(m, g, gemm_flops(g, nullptr)));
    std::vector<double> hc((size_t)Spad * Spad);
    HIP_TRY(hipMemcpyAsync(hc.data(), m->d_Kss.p, hc.size() * sizeof(double), hipMemcpyDeviceToHost, m->st));
    HIP_TRY(hipStreamSynchronize(m->st));
    for (int64_t a = 0; a < S; ++a)
        for (int64_t b = 0; b < S; ++b) cov[ss.perm[a] * S + ss.perm[b]] = hc[(size_t)a * Spad + b];
    return MOGP_OK;
}

}  // extern "C"
