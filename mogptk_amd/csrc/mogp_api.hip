// mogp_api.hip -- C ABI of libmogp_hip.so (see include/mogp_hip.h) and the per-evaluation orchestration:
//   Gram (lower tiles, noise + jitter fused on the diagonal) -> blocked Cholesky -> level-batched triangular inverse
//   -> alpha / log-det -> LAUUM (K^-1) -> gradient-moment pass -> a few hundred doubles back to the host.
#include "mogp_model.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <cstdlib>
#include <limits>
#include <numeric>
#include <unistd.h>
#include <map>
#include <set>

namespace mogp {

static thread_local std::string g_err;
void set_error(const std::string& msg) { g_err = msg; }
int hip_fail(hipError_t e, const char* what, const char* file, int line) {
    g_err = std::string("HIP error '") + hipGetErrorString(e) + "' in " + what + " (" + file + ":" + std::to_string(line) + ")";
    return MOGP_EHIP;
}
int launch_potrf_trtri_tile(double* A, int64_t ld, int t, double* invd, double* logdet, unsigned long long* info, hipStream_t s,
                            long long info_base = 0, int store_L = 0);

int fail(int code, const std::string& msg) { g_err = msg; return code; }

int sort_inputs(const double* X, int64_t M, int D, int C, int64_t pad_to, SortedX& o) {
    o.M = M;
    o.Mpad = round_up(std::max<int64_t>(M, 1), pad_to);
    o.perm.resize(M);
    o.off.assign(C + 1, 0);
    std::vector<int> chan(M);
    for (int64_t r = 0; r < M; ++r) {
        const double c = X[r * (1 + D)];
        if (!(c >= 0.0) || c >= (double)C || c != std::floor(c))
            return fail(MOGP_EINVAL, "X must have integers in [0, output_dims) for the channel IDs in the first input dimension");
        chan[r] = (int)c;
        o.off[chan[r] + 1]++;
    }
    for (int c = 0; c < C; ++c) o.off[c + 1] += o.off[c];
    std::vector<int> cur(o.off.begin(), o.off.end() - 1);
    o.identity = true;
    for (int64_t r = 0; r < M; ++r) {
        const int64_t pos = cur[chan[r]]++;
        o.perm[pos] = r;
        if (pos != r) o.identity = false;
    }
    o.xs.assign((size_t)D * o.Mpad, 0.0);
    for (int64_t pos = 0; pos < M; ++pos)
        for (int d = 0; d < D; ++d) o.xs[(size_t)d * o.Mpad + pos] = X[o.perm[pos] * (1 + D) + 1 + d];
    return 0;
}

// tiles of the symmetric Gram (lower channel pairs, lower tiles inside diagonal channel blocks), grouped by pair
void tile_blocks(const std::vector<int>& off, int C, std::vector<int>& blk) {
    blk.clear();
    for (int c = 0; c < C; ++c)
        for (int b = 0; b * MOGP_GT < off[c + 1] - off[c]; ++b) {
            blk.push_back(off[c] + b * MOGP_GT);
            blk.push_back(std::min(MOGP_GT, off[c + 1] - off[c] - b * MOGP_GT));
        }
}
static std::vector<int> block_base(const std::vector<int>& off, int C) {         // index of channel c's first block
    std::vector<int> base(C + 1, 0);
    for (int c = 0; c < C; ++c) base[c + 1] = base[c] + (off[c + 1] - off[c] + MOGP_GT - 1) / MOGP_GT;
    return base;
}

void build_sym_tiles(const std::vector<int>& off, int C, std::vector<GTile>& tiles, std::vector<int>& pair_start) {
    tiles.clear();
    pair_start.assign(1, 0);
    const std::vector<int> rbase = block_base(off, C), cbase = rbase;
    for (int i = 0; i < C; ++i)
        for (int j = 0; j <= i; ++j) {
            const int ni = off[i + 1] - off[i], nj = off[j + 1] - off[j];
            for (int bi = 0; bi * MOGP_GT < ni; ++bi)
                for (int bj = 0; bj * MOGP_GT < nj; ++bj) {
                    if (i == j && bj > bi) continue;
                    GTile t;
                    t.r0 = off[i] + bi * MOGP_GT; t.c0 = off[j] + bj * MOGP_GT;
                    t.nr = std::min(MOGP_GT, ni - bi * MOGP_GT); t.nc = std::min(MOGP_GT, nj - bj * MOGP_GT);
                    t.pair = i * C + j;
                    t.flags = (i == j && bi == bj) ? GT_DIAG : GT_MIRROR;
                    t.rb = rbase[i] + bi; t.cb = cbase[j] + bj;
                    tiles.push_back(t);
                }
            pair_start.push_back((int)tiles.size());
        }
}

void build_rect_tiles(const std::vector<int>& offr, const std::vector<int>& offc, int C, std::vector<GTile>& tiles,
                      std::vector<int>* pair_start) {
    tiles.clear();
    if (pair_start) pair_start->assign(1, 0);
    const std::vector<int> rbase = block_base(offr, C), cbase = block_base(offc, C);
    for (int i = 0; i < C; ++i)
        for (int j = 0; j < C; ++j) {
            const int ni = offr[i + 1] - offr[i], nj = offc[j + 1] - offc[j];
            for (int bi = 0; bi * MOGP_GT < ni; ++bi)
                for (int bj = 0; bj * MOGP_GT < nj; ++bj) {
                    GTile t;
                    t.r0 = offr[i] + bi * MOGP_GT; t.c0 = offc[j] + bj * MOGP_GT;
                    t.nr = std::min(MOGP_GT, ni - bi * MOGP_GT); t.nc = std::min(MOGP_GT, nj - bj * MOGP_GT);
                    t.pair = i * C + j;
                    t.flags = 0;
                    t.rb = rbase[i] + bi; t.cb = cbase[j] + bj;
                    tiles.push_back(t);
                }
            if (pair_start) pair_start->push_back((int)tiles.size());
        }
}

}  // namespace mogp

using namespace mogp;

// The ONE wait of an evaluation.  hipStreamSynchronize sleeps on an interrupt; on a shared, loaded host the wake-up is what the wall clock
// of a 13 ms evaluation then waits for.  Polling the stream costs one busy core for the duration and returns within microseconds.
static int wait_stream(hipStream_t st) {
    static const bool spin = !(std::getenv("MOGP_SPIN_WAIT") && std::atoi(std::getenv("MOGP_SPIN_WAIT")) == 0);
    if (!spin) { HIP_TRY(hipStreamSynchronize(st)); return 0; }
    for (;;) {
        const hipError_t e = hipStreamQuery(st);
        if (e == hipSuccess) return 0;
        if (e != hipErrorNotReady) return hip_fail(e, "hipStreamQuery", __FILE__, __LINE__);
    }
}

int StripTiles::build(const std::vector<GTile>& tiles) {
    static const int maxrun = []() { const char* e = std::getenv("MOGP_STRIP_RUN"); const int v = e ? std::atoi(e) : 0; return v > 0 ? v : 8; }();      // (round 6, with the graded tail and the one-barrier kernel: 103 us at 8 against 105 at 4 and 108 at 6, configs[1])
    split_strip_tiles(tiles, maxrun, segs, rest);
    // The hardware hands out workgroups in index order as slots free up: the launch ends when its LAST runs end, so those should be short.  The runs
    // that would be dispatched last (the final `tail` tiles' worth) are cut into single tiles, the `tail` tiles before them into pairs; a list with
    // fewer runs than there are workgroup slots is cut into single tiles altogether (the head launch of the dataflow schedule: 8 column tiles a row).
    static const int grade = []() { const char* e = std::getenv("MOGP_STRIP_GRADE"); return e ? std::atoi(e) : 1; }();
    if (grade > 0 && maxrun > 1 && !segs.empty()) {
        static const int slots = []() { int dev = 0; hipDeviceProp_t pr; if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&pr, dev) != hipSuccess) return 512; return 2 * pr.multiProcessorCount; }();
        const long tail = (long)grade * slots;
        long total = 0;
        for (const GSeg& g : segs) total += g.n;
        std::vector<GSeg> out;
        out.reserve(segs.size() * 2);
        long seen = 0;
        const bool all_single = (long)segs.size() < 2L * slots;
        for (const GSeg& g : segs) {
            const long left = total - seen;                      // tiles from this run to the end of the list
            const int cut = (all_single || left <= tail) ? 1 : (left <= 2 * tail ? 2 : maxrun);
            for (int o = 0; o < g.n; o += cut) {
                GSeg h = g;
                h.c0 = g.c0 + o * MOGP_GT; h.n = std::min(cut, g.n - o); h.diag = (o + cut >= g.n) ? g.diag : 0;
                out.push_back(h);
            }
            seen += g.n;
        }
        segs.swap(out);
    }
    int rc;
    if ((rc = d_segs.ensure(std::max<size_t>(segs.size(), 1)))) return rc;
    if ((rc = d_rest.ensure(std::max<size_t>(rest.size(), 1)))) return rc;
    if (!segs.empty()) HIP_TRY(dev_upload(d_segs.p, segs.data(), segs.size() * sizeof(GSeg)));
    if (!rest.empty()) HIP_TRY(dev_upload(d_rest.p, rest.data(), rest.size() * sizeof(GTile)));
    return 0;
}

static int g_outer = 4;    // outer Cholesky block in tiles (x128 columns); MOGP_OUTER env var overrides (tuning)
#define MOGP_OUTER g_outer

namespace mogp { int use_device(mogp_ctx* c) { HIP_TRY(hipSetDevice(c->device)); return 0; } }

// ---------------------------------------------------------------------------------------------------------------
extern "C" {

const char* mogp_version(void) { return "mogp_hip 0.1 (gfx950, fp64)"; }
const char* mogp_last_error(void) { return g_err.c_str(); }

int mogp_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int mogp_ctx_create(int device, mogp_ctx** out) {
    if (!out) return fail(MOGP_EINVAL, "mogp_ctx_create: out is null");
    int n = mogp_device_count();
    if (n <= 0) return fail(MOGP_ENODEVICE, "no HIP device visible: mogptk_amd has no CPU path");
    if (device < 0 || device >= n) return fail(MOGP_EINVAL, "mogp_ctx_create: device ordinal out of range");
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    mogp_ctx* c = new mogp_ctx();
    c->device = device;
    c->name = std::string(prop.name) + " (" + prop.gcnArchName + ")";
    {   // One device-to-host copy of more than 16 KB into pinned memory NOW.  The first such copy of a process sets something up inside the runtime, and when
        // that happened while the dataflow kernel and its chain kernels were running (the z^T z parts of a model with more than 8192 points: 18 KB), they
        // stalled until their waits gave up: every first evaluation above N = 8192 fell back to the stream schedule (tools/r4_first.py; 16 KB pieces do not
        // trigger it, a sleep in front of the copy avoids it).  Found in round 4 when the dataflow default went to 96 tile rows.
        // ON THIS CONTEXT'S DEVICE (whatever the calling thread's current device is -- if the set-up is per device, GPUs 1 .. n need it as well), on
        // that device's null stream (the context's own streams are created with its first model), and the caller's current device is restored.
        int prev = -1;
        hipError_t e = hipGetDevice(&prev); (void)e;
        if (hipSetDevice(device) == hipSuccess) {
            void* dsrc = nullptr; void* hdst = nullptr;
            const size_t nbytes = 32u << 20;                     // and at three sizes: the runtime picks its copy path by size (the prediction brings back 33 KB, a fetch 512 MB)
            if (hipMalloc(&dsrc, nbytes) == hipSuccess && hipHostMalloc(&hdst, nbytes, hipHostMallocDefault) == hipSuccess) {
                e = hipMemsetAsync(dsrc, 0, nbytes, nullptr); (void)e;
                for (size_t nb : {(size_t)64 << 10, (size_t)2 << 20, nbytes}) { e = hipMemcpyAsync(hdst, dsrc, nb, hipMemcpyDeviceToHost, nullptr); (void)e; }
                e = hipStreamSynchronize(nullptr); (void)e;
            }
            if (hdst) { e = hipHostFree(hdst); (void)e; }
            if (dsrc) { e = hipFree(dsrc); (void)e; }
            if (prev >= 0 && prev != device) { e = hipSetDevice(prev); (void)e; }
        }
    }
    *out = c;
    return MOGP_OK;
}

int mogp_ctx_destroy(mogp_ctx* ctx) {
    if (!ctx) return MOGP_OK;
    for (hipStream_t q : {ctx->st, ctx->st2, ctx->st2u, ctx->st3, ctx->st4, ctx->st5, ctx->st_priv}) if (q) { hipError_t e = hipStreamSynchronize(q); (void)e; e = hipStreamDestroy(q); (void)e; }
    for (auto& kv : ctx->sk) { kv.second.ws.release(); kv.second.flags.release(); }
    ctx->sk.clear();
    delete ctx;
    return MOGP_OK;
}

int mogp_ctx_device_name(mogp_ctx* ctx, char* buf, int buflen) {
    if (!ctx || !buf || buflen <= 0) return fail(MOGP_EINVAL, "mogp_ctx_device_name: bad argument");
    std::snprintf(buf, (size_t)buflen, "%s", ctx->name.c_str());
    return MOGP_OK;
}

}  // extern "C"

// ---- TRTRI level tasks ------------------------------------------------------------------------------------------
// Bottom-up pairing of tile ranges: at level l (block size s = 2^(l-1) tiles) node b owns tiles [2sb, 2sb+2s); its left
// half [lo, mid) and right half [mid, hi) are already inverted, and  W21 = -W22 * (L21 * W11)  fills the off-diagonal part.
namespace mogp { void build_trtri_levels(Spd& w) {
    const int nb = w.nb;
    const int64_t ld = w.Npad;
    w.levels.clear();
    for (int s = 1; s < nb; s *= 2) {
        TrtriLevel lv;
        for (int lo = 0; lo < nb; lo += 2 * s) {
            const int mid = lo + s, hi = std::min(lo + 2 * s, nb);
            if (mid >= hi) continue;
            for (int ti = mid; ti < hi; ++ti)
                for (int tj = lo; tj < mid; ++tj) {
                    GemmTask t1;      // T[ti][tj] = sum_{k = tj..mid} L21[ti][k] * W11[k][tj]      (W11 lower: k >= tj)
                    t1.a_off = (int64_t)ti * MOGP_TILE * ld + (int64_t)tj * MOGP_TILE;
                    t1.b_off = (int64_t)tj * MOGP_TILE * ld + (int64_t)tj * MOGP_TILE;
                    t1.c_off = (int64_t)ti * MOGP_TILE * ld + (int64_t)tj * MOGP_TILE;
                    t1.kt = (mid - tj) * (MOGP_TILE / 16); t1.pad = 0;
                    lv.h1.push_back(t1);
                    GemmTask t2;      // W21[ti][tj] = - sum_{k = mid..ti} W22[ti][k] * T[k][tj]   (W22 lower: k <= ti)
                    t2.a_off = (int64_t)ti * MOGP_TILE * ld + (int64_t)mid * MOGP_TILE;
                    t2.b_off = (int64_t)mid * MOGP_TILE * ld + (int64_t)tj * MOGP_TILE;
                    t2.c_off = t1.c_off;
                    t2.kt = (ti - mid + 1) * (MOGP_TILE / 16); t2.pad = 0;
                    lv.h2.push_back(t2);
                }
        }
        auto by_k = [](const GemmTask& a, const GemmTask& b) { return a.kt > b.kt; };
        std::stable_sort(lv.h1.begin(), lv.h1.end(), by_k);
        std::stable_sort(lv.h2.begin(), lv.h2.end(), by_k);
        for (auto& t : lv.h1) lv.flops1 += 2.0 * MOGP_TILE * MOGP_TILE * 16.0 * t.kt;
        for (auto& t : lv.h2) lv.flops2 += 2.0 * MOGP_TILE * MOGP_TILE * 16.0 * t.kt;
        w.levels.push_back(std::move(lv));
    }
}
}  // namespace mogp

// Stream-K form for the launches whose tile count sits badly on the workgroup slots of their stream (linalg.hip:k_gemm_sk).
// Measured inside the schedules (round 3, tools/r3_x2.sh): the wide triangular solves of the sparse models -- 782-tile updates that have the
// chip to themselves -- gain (configs[4] 49.5 -> 47.9 ms); the fused factorisation + inversion and the prediction LOSE (configs[1] 12.9 ->
// 13.9 ms, configs[3] 47.2 -> 50.4): next to other launches a partly filled round is filled by them anyway, and workgroups that hold their
// slots four tiles long delay the critical stream's launches.  So: only where the caller asks (GemmArgs::sk_hint).
// MOGP_SK (experiments): 0 = never; bit 0 = also launches of at least half the slots whose last round would be less than MOGP_SK_FILL
// percent full; bit 1 = also launches of fewer tiles than a quarter of the slots on the critical stream; bit 2 = every eligible launch.
namespace mogp { static int stream_k_setup(mogp_model* m, GemmArgs& g, hipStream_t st) {
    static const int sk_mode = std::getenv("MOGP_SK") ? std::atoi(std::getenv("MOGP_SK")) : 8;        // 8: hinted launches only
    static const int sk_min = std::getenv("MOGP_SK_MIN") ? std::max(1, std::atoi(std::getenv("MOGP_SK_MIN"))) : 8;       // k blocks per span at least
    static const int sk_fill = std::getenv("MOGP_SK_FILL") ? std::atoi(std::getenv("MOGP_SK_FILL")) : 60;
    g.sk_spans = 0;
    if (!sk_mode || m->no_chain || g.small || g.ksplit > 1 || g.row_mod > 1 || g.K % 16) return 0;
    if (!(g.mode == GM_RECT || g.mode == GM_RECT_LOWER || g.mode == GM_LOWER || g.mode == GM_KHI_J || g.mode == GM_KLO_J)) return 0;
    mogp_ctx* ctx = m->ctx;
    const int ncu = ctx->ncu > 0 ? ctx->ncu : 256;
    int cus = ncu;
    if (ctx->st_priv) {
        if (st == ctx->st_priv) cus = ctx->ncu_reserved;
        else if (st == ctx->st2 || st == ctx->st3 || st == ctx->st4) cus = ncu - ctx->ncu_reserved;
    }
    const int slots = 2 * cus;
    const long long T = g.mode == GM_LOWER ? (long long)g.mt * (g.mt + 1) / 2 : (long long)g.mt * g.nt;
    long long tot;
    if (g.mode == GM_KHI_J || g.mode == GM_KLO_J) { long long row = 0; for (int tj = 0; tj < g.nt; ++tj) row += (g.mode == GM_KHI_J ? std::min<long long>(g.K, (long long)(tj + 1) * MOGP_TILE) : g.K - (long long)tj * MOGP_TILE) / 16; tot = row * g.mt; }
    else tot = T * (g.K / 16);
    if (tot <= 0 || tot >= (1ll << 31) || (long long)slots * slots >= (1ll << 31)) return 0;
    bool want = (sk_mode & 4) != 0;
    if (((sk_mode & 1) || g.sk_hint) && 2 * T >= slots) {
        const long long last = T % slots;                       // tiles of the last round
        if (last != 0 && 100 * last < (long long)sk_fill * slots) want = true;
    }
    if ((sk_mode & 2) && st == ctx->st && 4 * T < slots) want = true;
    if (!want) return 0;
    const long long spans = std::min<long long>(slots, tot / sk_min);
    if (spans < 2) return 0;
    mogp_ctx::SkWs& w = ctx->sk[st];
    if (w.flags.n < (size_t)slots) {
        HIP_TRY(hipStreamSynchronize(st));
        int rc;
        if ((rc = w.ws.ensure((size_t)slots * MOGP_TILE * MOGP_TILE))) return rc;
        if ((rc = w.flags.ensure((size_t)slots))) return rc;
        HIP_TRY(hipMemsetAsync(w.flags.p, 0, (size_t)slots * sizeof(unsigned), st));      // in stream order before the launch (the streams are non-blocking)
        w.epoch = 0;
    }
    g.sk_spans = (int)spans; g.sk_ws = w.ws.p; g.sk_flags = w.flags.p; g.sk_epoch = ++w.epoch; g.sk_info = m->d_info.p;
    return 0;
} }

namespace mogp { int gemm_call(mogp_model* m, const GemmArgs& g, double flops, hipStream_t st) {
    if (!st) st = m->st;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (m->profiling) {
        if (m->gemm_ev_used + 2 > m->gemm_ev.size()) {
            for (int i = 0; i < 64; ++i) { hipEvent_t e; HIP_TRY(hipEventCreate(&e)); m->gemm_ev.push_back(e); }
        }
        e0 = m->gemm_ev[m->gemm_ev_used++];
        e1 = m->gemm_ev[m->gemm_ev_used++];
        HIP_TRY(hipEventRecord(e0, st));
    }
    GemmArgs gs = g;
    { int r__ = stream_k_setup(m, gs, st); if (r__) return r__; }
    int rc = launch_gemm(gs, st);
    if (rc) return rc;
    if (m->profiling) HIP_TRY(hipEventRecord(e1, st));
    m->gemm_launches++;
    m->gemm_flops += flops;
    return 0;
}
}  // namespace mogp

namespace mogp { int mark(mogp_model* m, int idx) {
    if (!m->profiling) return 0;
    while ((int)m->ev.size() <= idx) { hipEvent_t e; HIP_TRY(hipEventCreate(&e)); m->ev.push_back(e); }
    HIP_TRY(hipEventRecord(m->ev[idx], m->st));
    return 0;
}
// events 7 .. 10 bracket the Gram and the moment tile kernels alone (handed to the launchers)
static hipEvent_t prof_event(mogp_model* m, int idx) {
    if (!m->profiling) return nullptr;
    while ((int)m->ev.size() <= idx) { hipEvent_t e; if (hipEventCreate(&e) != hipSuccess) return nullptr; m->ev.push_back(e); }
    return m->ev[idx];
}
}  // namespace mogp

// Cholesky of w.A (lower) in place; w.invd gets the inverses of the diagonal 128-tiles, w.logdet the per-tile sums of
// log L_kk; a non-positive pivot is reported through m->d_info (atomicMin of the 1-based index).
namespace mogp { int spd_potrf(mogp_model* m, Spd& w, long long info_base) {
    int rc;
    hipStream_t cq = m->st;
    // Bulk stream: the one masked to everything but the reserved CUs while the serial chain matters -- the chain's small kernels (this
    // stream, all CUs) then find idle CUs instead of sharing one with GEMM waves: 15.9 vs 21.1 ms per evaluation at N = 8192, 74 vs 82 ms
    // for the N = 16384 prediction.  Once the work is flop-bound the 6 % of CUs matter more (sweep at N = 32768: 597 vs 638 ms): all CUs.
    static const int bound_tiles = std::getenv("MOGP_CHAIN_BOUND") ? std::atoi(std::getenv("MOGP_CHAIN_BOUND")) : MOGP_CHAIN_BOUND_TILES;      // (experiment switch: tile rows up to which the bulk stream stays off the reserved CUs)
    hipStream_t bulk_q = (w.nb > bound_tiles && m->st2u) ? m->st2u : m->st2;
    // ---- two-level blocked right-looking Cholesky with look-ahead.
    // Outer blocks of MOGP_OUTER tiles.  "chain(kb)" = for each 128-column of the block: leaf (factor + inverse) -> panel =
    // panel * inv(Lkk)^T for ALL rows below -> update of the block's remaining columns (64x64-tile GEMMs: latency-bound).
    // The trailing matrix gets one K = MOGP_OUTER*128 SYRK per outer block, split in two:
    //   A(kb): the next block's columns, on the critical stream (chain(kb+1) needs them);
    //   B(kb): everything to the right, on the bulk stream, overlapping chain(kb+1).
    // A(kb) and B(kb-1) accumulate into the same tiles, so A(kb) waits for B(kb-1).
    { const char* e = std::getenv("MOGP_OUTER"); if (e && std::atoi(e) > 0) g_outer = std::atoi(e); }
    const int nb = w.nb;
    const int nouter = (nb + MOGP_OUTER - 1) / MOGP_OUTER;
    while ((int)w.sync_ev.size() < 2 * nouter + 2) {
        hipEvent_t e;
        HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        w.sync_ev.push_back(e);
    }
    int last_bulk = -1;
    if (w.want_row_ev)
        while ((int)w.row_ev.size() < nb) {
            hipEvent_t e;
            HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            w.row_ev.push_back(e);
        }
    for (int kb = 0; kb < nouter; ++kb) {
        const int k0 = kb * MOGP_OUTER, k1 = std::min(k0 + MOGP_OUTER, nb);
        for (int k = k0; k < k1; ++k) {
            if ((rc = launch_potrf_trtri_tile(w.A.p, w.Npad, k, w.invd.p, w.logdet.p, m->d_info.p, cq, info_base, w.keep_L ? 1 : 0))) return rc;
            if (w.want_row_ev) HIP_TRY(hipEventRecord(w.row_ev[k], cq));      // block row k of L: the panels of the columns left of it are behind us on this stream, L_kk is this leaf's
            const int rem = nb - k - 1;
            if (rem <= 0) break;
            double* panel = w.A.p + (int64_t)(k + 1) * MOGP_TILE * w.Npad + (int64_t)k * MOGP_TILE;
            GemmArgs g{};
            g.A = panel; g.lda = w.Npad; g.a_kmajor = 0;
            g.B = w.invd.p + (int64_t)k * MOGP_TILE * MOGP_TILE; g.ldb = MOGP_TILE; g.b_kmajor = 0;
            g.C = panel; g.ldc = w.Npad; g.alpha = 1.0; g.beta = 0.0;
            g.mode = GM_RECT; g.small = 1; g.mt = 2 * rem; g.nt = 1; g.K = MOGP_TILE;      // 64x128 tiles: in place
            if (w.refine_panels) {            // keep the panel as it came: the residual below is taken against it
                if ((rc = w.pscr.ensure((size_t)w.Npad * MOGP_TILE))) return rc;
                if ((rc = launch_copy2d(w.pscr.p, MOGP_TILE, panel, w.Npad, (int64_t)rem * MOGP_TILE, MOGP_TILE, 1.0, cq))) return rc;
            }
            if ((rc = gemm_call(m, g, gemm_flops(g, nullptr), cq))) return rc;
            if (w.refine_panels) {
                // (round 5) P = A0 W^T is only as good as the explicit tile inverse: with L_kk of condition 4e4 (a 128-point stretch of the inducing
                // grid of configs[4]) the leaf's W has |I - L W| = 3e-11 and the factor a backward error |L L^T - A| / |A| = 1.3e-11 -- four orders
                // above LAPACK's, 5 % of the smallest eigenvalue of K_uu + jitter, and the reason dELBO/dZ sat twice as far from the 80-bit truth
                // as the reference (tools/titsias_stage_errors.py, tools/titsias_chol_residual.py).  One step of iterative refinement against
                // the factor itself, P += (A0 - P L_kk^T) W^T, brings the residual to 6e-16.  Two more small GEMMs per tile column; asked for by
                // the sparse models' K_uu only (Spd::refine_panels; needs keep_L: the diagonal tile of A holds L_kk).
                GemmArgs r1 = g;
                r1.A = panel; r1.lda = w.Npad; r1.B = w.A.p + (int64_t)k * MOGP_TILE * (w.Npad + 1); r1.ldb = w.Npad;
                r1.C = w.pscr.p; r1.ldc = MOGP_TILE; r1.alpha = -1.0; r1.beta = 1.0;
                if ((rc = gemm_call(m, r1, gemm_flops(r1, nullptr), cq))) return rc;
                GemmArgs r2 = g;
                r2.A = w.pscr.p; r2.lda = MOGP_TILE; r2.C = panel; r2.ldc = w.Npad; r2.alpha = 1.0; r2.beta = 1.0;
                if ((rc = gemm_call(m, r2, gemm_flops(r2, nullptr), cq))) return rc;
            }
            const int inner = k1 - k - 1;            // columns k+1 .. k1-1 of this outer block
            if (inner > 0) {
                GemmArgs u{};
                u.A = panel; u.lda = w.Npad; u.a_kmajor = 0; u.B = panel; u.ldb = w.Npad; u.b_kmajor = 0;
                u.C = w.A.p + (int64_t)(k + 1) * MOGP_TILE * (w.Npad + 1); u.ldc = w.Npad; u.alpha = -1.0; u.beta = 1.0;
                u.mode = GM_RECT_LOWER; u.small = 2; u.mt = 2 * rem; u.nt = 2 * inner; u.K = MOGP_TILE;
                if ((rc = gemm_call(m, u, gemm_flops(u, nullptr), cq))) return rc;
            }
        }
        const int rem = nb - k1;
        HIP_TRY(hipEventRecord(w.sync_ev[2 * kb], cq));                       // chain(kb) done
        if (rem <= 0) break;
        double* blockp = w.A.p + (int64_t)k1 * MOGP_TILE * w.Npad + (int64_t)k0 * MOGP_TILE;
        const int K = (k1 - k0) * MOGP_TILE;
        const int na = std::min(MOGP_OUTER, rem);      // tile columns of the next outer block
        if (rem > na) {                                                           // B(kb) on the bulk stream
            HIP_TRY(hipStreamWaitEvent(bulk_q, w.sync_ev[2 * kb], 0));
            double* bp = blockp + (int64_t)na * MOGP_TILE * w.Npad;
            GemmArgs u{};
            u.A = bp; u.lda = w.Npad; u.a_kmajor = 0; u.B = bp; u.ldb = w.Npad; u.b_kmajor = 0;
            u.C = w.A.p + (int64_t)(k1 + na) * MOGP_TILE * (w.Npad + 1); u.ldc = w.Npad; u.alpha = -1.0; u.beta = 1.0;
            u.mode = GM_LOWER; u.mt = rem - na; u.nt = rem - na; u.K = K;
            if ((rc = gemm_call(m, u, gemm_flops(u, nullptr), bulk_q))) return rc;
        }
        if (last_bulk >= 0) HIP_TRY(hipStreamWaitEvent(cq, w.sync_ev[2 * last_bulk + 1], 0));   // A(kb) after B(kb-1)
        if (rem > na) { HIP_TRY(hipEventRecord(w.sync_ev[2 * kb + 1], bulk_q)); last_bulk = kb; }
        {
            GemmArgs u{};                                                         // A(kb): columns k1 .. k1+na-1, rows >= column
            u.A = blockp; u.lda = w.Npad; u.a_kmajor = 0; u.B = blockp; u.ldb = w.Npad; u.b_kmajor = 0;
            u.C = w.A.p + (int64_t)k1 * MOGP_TILE * (w.Npad + 1); u.ldc = w.Npad; u.alpha = -1.0; u.beta = 1.0;
            u.mode = GM_RECT_LOWER; u.mt = rem; u.nt = na; u.K = K;
            if ((rc = gemm_call(m, u, gemm_flops(u, nullptr), cq))) return rc;
        }
    }
    if (last_bulk >= 0) HIP_TRY(hipStreamWaitEvent(cq, w.sync_ev[2 * last_bulk + 1], 0));
    return 0;
}
}  // namespace mogp

// w.A: L -> W = L^-1 (lower), level-batched; uses w.B as scratch
namespace mogp { int spd_trtri(mogp_model* m, Spd& w) {
    int rc;
    const int nb = w.nb;
    // ---- W = L^-1, level by level (all nodes of one level in one launch)
    if ((rc = launch_put_diag_tiles(w.A.p, w.Npad, nb, w.invd.p, m->st))) return rc;
    for (auto& lv : w.levels) {
        GemmArgs g{};
        g.A = w.A.p; g.lda = w.Npad; g.a_kmajor = 0; g.B = w.A.p; g.ldb = w.Npad; g.b_kmajor = 1;
        g.C = w.B.p; g.ldc = w.Npad; g.alpha = 1.0; g.beta = 0.0;
        g.mode = GM_TASKS; g.mt = g.nt = 0; g.K = 0; g.tasks = lv.d1.p; g.ntasks = (int)lv.h1.size();
        if ((rc = gemm_call(m, g, lv.flops1))) return rc;
        g.B = w.B.p; g.C = w.A.p; g.alpha = -1.0; g.tasks = lv.d2.p; g.ntasks = (int)lv.h2.size();
        if ((rc = gemm_call(m, g, lv.flops2))) return rc;
    }
    return 0;
}
}  // namespace mogp

// w.B (lower tiles, full diagonal tiles) = W^T W with W = w.A lower triangular: ONE LAUUM-mode GEMM launch
namespace mogp { int spd_lauum(mogp_model* m, Spd& w) {
    GemmArgs g{};
    g.A = w.A.p; g.lda = w.Npad; g.a_kmajor = 1; g.B = w.A.p; g.ldb = w.Npad; g.b_kmajor = 1;
    g.C = w.B.p; g.ldc = w.Npad; g.alpha = 1.0; g.beta = 0.0;
    if (m->kinv_sparse && &w == &m->k && !m->kinv_lauum_tasks.empty()) {          // only the tiles the gradient reads (kinv_plan), longest k range first
        g.mode = GM_TASKS; g.tasks = m->d_kinv_lauum.p; g.ntasks = (int)m->kinv_lauum_tasks.size(); g.mt = g.nt = 0; g.K = 0;
        double fl = 0.0;
        for (const GemmTask& t : m->kinv_lauum_tasks) fl += 2.0 * MOGP_TILE * MOGP_TILE * 16.0 * t.kt;
        return gemm_call(m, g, fl);
    }
    g.mode = GM_LAUUM; g.mt = g.nt = w.nb; g.K = (int)w.Npad;
    return gemm_call(m, g, gemm_flops(g, nullptr));
}
}  // namespace mogp

namespace mogp { int spd_alloc(Spd& w, int64_t Npad, int owned_rows_device) {
    if (w.Npad == Npad) return 0;
    w.release();
    w.Npad = Npad; w.nb = (int)(Npad / MOGP_TILE);
    int rc;
    bool owned = owned_rows_device >= 0;
    if (owned && w.Arows.reserve((size_t)Npad * Npad * sizeof(double), owned_rows_device)) {
        // a runtime without virtual memory management (or out of address space): the ordinary allocation -- the owned-rows code path does not care
        // whether the rows it never touches exist
        w.Arows.release();
        (void)hipGetLastError();
        owned = false;
    }
    if (owned) {
        // the work matrix of a sharded evaluation: the whole address range, physical memory only where mogp_shard_config asks for it; no B
        // (nothing of the sharded gradient evaluation uses it -- the sharded prediction allocates it when it comes)
        w.A.p = reinterpret_cast<double*>(w.Arows.base); w.A.n = (size_t)Npad * Npad; w.A.borrowed = true;
        w.owned_rows = true;
    } else {
        if ((rc = w.A.ensure((size_t)Npad * Npad))) return rc;
        if ((rc = w.B.ensure((size_t)Npad * Npad))) return rc;
    }
    if ((rc = w.invd.ensure((size_t)w.nb * MOGP_TILE * MOGP_TILE))) return rc;
    if ((rc = w.logdet.ensure(w.nb))) return rc;
    build_trtri_levels(w);
    for (auto& lv : w.levels) {
        if ((rc = lv.d1.ensure(std::max<size_t>(lv.h1.size(), 1)))) return rc;
        if ((rc = lv.d2.ensure(std::max<size_t>(lv.h2.size(), 1)))) return rc;
        HIP_TRY(dev_upload(lv.d1.p, lv.h1.data(), lv.h1.size() * sizeof(GemmTask)));
        HIP_TRY(dev_upload(lv.d2.p, lv.h2.data(), lv.h2.size() * sizeof(GemmTask)));
    }
    // nothing ever writes above the block diagonal of A / B; keep it finite (an owned-rows A is zeroed granule by granule as it is backed)
    if (owned) return 0;
    { int r__ = dev_fill_zero(w.A.p, (size_t)Npad * Npad * sizeof(double)); if (r__) return r__; }
    { int r__ = dev_fill_zero(w.B.p, (size_t)Npad * Npad * sizeof(double)); if (r__) return r__; }
    return 0;
}
int spd_make_whole(Spd& w) {
    int rc;
    if (!w.owned_rows) return 0;
    if ((rc = w.Arows.back(0, (size_t)w.Npad * w.Npad * sizeof(double)))) return rc;
    if (!w.B.p) {
        if ((rc = w.B.ensure((size_t)w.Npad * w.Npad))) return rc;
        if ((rc = dev_fill_zero(w.B.p, (size_t)w.Npad * w.Npad * sizeof(double)))) return rc;
    }
    w.owned_rows = false;
    return 0;
}
}  // namespace mogp

// diagonal value of channel block (c, c) implied by the table (Delta = Psi = 0 there for every kernel on the path)
namespace mogp { double table_diag(const mogp_model* m, int c) {
    const int D = m->D, W = m->Wt;
    const double* tab = m->table.data() + (size_t)(c * m->C + c) * m->T * W;
    double s = 0.0;
    for (int t = 0; t < m->T; ++t) {
        const double* r = tab + (size_t)t * W;
        double arg = 0.0, ph = r[1];
        for (int d = 0; d < D; ++d) { arg += r[2 + d] * r[2 + 2 * D + d] * r[2 + 2 * D + d]; ph += r[2 + D + d] * r[2 + 2 * D + d]; }
        s += r[0] * std::exp(-0.5 * arg) * std::cos(2.0 * M_PI * ph);
    }
    return s;
}
}  // namespace mogp

// sum over the points of `pts` of the kernel diagonal K(x, x) implied by the table: n_c table_diag(c) per channel, or -- with an envelope on
// the input midpoint (rows of width 2 + 5 D, MOHSM) -- sum_t A_t exp(-1/2 sum_d L_d (x_d - c_d)^2) point by point
namespace mogp { double table_diag_points(const mogp_model* m, const SortedX& pts) {
    const int D = m->D, W = m->Wt, C = m->C;
    double s = 0.0;
    if (W == 2 + 3 * D) {
        for (int c = 0; c < C; ++c) s += (double)(pts.off[c + 1] - pts.off[c]) * table_diag(m, c);
        return s;
    }
    for (int c = 0; c < C; ++c) {
        const double* tab = m->table.data() + (size_t)(c * C + c) * m->T * W;
        for (int pos = pts.off[c]; pos < pts.off[c + 1]; ++pos)
            for (int t = 0; t < m->T; ++t) {
                const double* r = tab + (size_t)t * W;
                double arg = 0.0, ph = r[1], env = 0.0;
                for (int d = 0; d < D; ++d) {
                    arg += r[2 + d] * r[2 + 2 * D + d] * r[2 + 2 * D + d];
                    ph += r[2 + D + d] * r[2 + 2 * D + d];
                    const double a = pts.xs[(size_t)d * pts.Mpad + pos] - r[2 + 4 * D + d];
                    env += r[2 + 3 * D + d] * a * a;
                }
                s += r[0] * std::exp(-0.5 * (arg + env)) * std::cos(2.0 * M_PI * ph);
            }
    }
    return s;
}
}  // namespace mogp

// every entry point that evaluates on this GPU alone: whatever a sharded evaluation of the same model left behind (row ownership: the moment pass,
// the alpha sums and the sweep's updates mask by it; the owned-rows form of the work matrix) no longer applies
namespace mogp { void one_gpu_call(mogp_model* m) { m->sh_n = 1; m->sh_rank = 0; m->sh_owned = false; } }

namespace mogp { int ensure_system(mogp_model* m) {
    int rc;
    if (m->tiles.empty()) {
        build_sym_tiles(m->sx.off, m->C, m->tiles, m->pair_start);
        if ((rc = m->d_tiles.ensure(m->tiles.size()))) return rc;
        if ((rc = m->d_pair_start.ensure(m->pair_start.size()))) return rc;
        HIP_TRY(dev_upload(m->d_tiles.p, m->tiles.data(), m->tiles.size() * sizeof(GTile)));
        if ((rc = m->strip.build(m->tiles))) return rc;
        for (const GTile& t : m->tiles) (t.c0 < 4 * MOGP_TILE ? m->tiles_head : m->tiles_tail).push_back(t);
        if (!m->tiles_head.empty() && !m->tiles_tail.empty()) {
            if ((rc = m->d_tiles_head.ensure(m->tiles_head.size()))) return rc;
            if ((rc = m->d_tiles_tail.ensure(m->tiles_tail.size()))) return rc;
            HIP_TRY(dev_upload(m->d_tiles_head.p, m->tiles_head.data(), m->tiles_head.size() * sizeof(GTile)));
            HIP_TRY(dev_upload(m->d_tiles_tail.p, m->tiles_tail.data(), m->tiles_tail.size() * sizeof(GTile)));
            if ((rc = m->strip_head.build(m->tiles_head))) return rc;
            if ((rc = m->strip_tail.build(m->tiles_tail))) return rc;
        }
        HIP_TRY(dev_upload(m->d_pair_start.p, m->pair_start.data(), m->pair_start.size() * sizeof(int)));
    }
    if ((rc = m->d_partial.ensure(m->tiles.size() * (size_t)std::max(m->T, 1) * (size_t)std::max(m->Wt, 1)))) return rc;
    // a model whose FIRST evaluation is a sharded one gets its work matrix in the owned-rows form (mogp_shard_config backs the rows); the first
    // one-GPU call on such a model makes it whole
    if (m->k.Npad == m->Npad) return (m->k.owned_rows && !m->sh_owned) ? spd_make_whole(m->k) : 0;
    return spd_alloc(m->k, m->Npad, m->sh_owned ? m->ctx->device : -1);
} }

// ---- which tiles of Kj^-1 a gradient evaluation needs ------------------------------------------------------------------------------------
// The gradient is 1/2 sum_ab (alpha_a alpha_b - Kinv_ab) dK_ab/dtheta.  The moment kernel (gram.hip:k_moments) drops a term in a 64 x 64
// tile when the smallest exponent it can reach there is below -50 (the rule the Gram build uses for K itself: gram.hip:stage_item_compute),
// so where ALL terms of a tile are dropped the entries of Kj^-1 under it are never read -- and the accumulation Kj^-1 = W^T W need not
// form them.  For stationary kernels on long series that is most of the matrix: at BASELINE configs[1] (2048 points per channel over
// [0, 100], spectral variances ~0.03: a support of +-10) 65 % of the 128 x 128 tiles, i.e. 22 % of all flops of the evaluation.
// The plan is made on the host from the same numbers the device uses (block centres and half spans, the term table) with a stricter
// threshold (52 instead of 50), so it can only keep MORE tiles than the kernel reads.  Exact: the dropped terms are below 2e-22 of a
// tile's peak either way.  MOGP_FULL_INVERSE=1 forms every tile; mogp_model_fetch(which = 1) completes a planned inverse on demand.
static void kinv_block_ranges(mogp_model* m) {
    std::vector<int> blk;
    tile_blocks(m->sx.off, m->C, blk);
    const int nblk = (int)blk.size() / 2, D = m->D;
    m->blk_cen.assign((size_t)D * nblk, 0.0); m->blk_half.assign((size_t)D * nblk, 0.0);
    for (int b = 0; b < nblk; ++b)
        for (int d = 0; d < D; ++d) {
            const double* x = m->sx.xs.data() + (size_t)d * m->sx.Mpad + blk[2 * b];
            double lo = x[0], hi = x[0];
            for (int i = 1; i < blk[2 * b + 1]; ++i) { lo = std::fmin(lo, x[i]); hi = std::fmax(hi, x[i]); }
            m->blk_cen[(size_t)d * nblk + b] = 0.5 * (lo + hi); m->blk_half[(size_t)d * nblk + b] = 0.5 * (hi - lo);
        }
}

static int kinv_plan(mogp_model* m, bool want) {
    static const bool full = std::getenv("MOGP_FULL_INVERSE") && std::atoi(std::getenv("MOGP_FULL_INVERSE")) != 0;
    m->kinv_sparse = false; m->kinv_fraction = 1.0;
    if (!want || full || m->sh_n > 1 || m->tiles.empty()) return 0;
    const int nb = m->nb, D = m->D, T = m->T, W = m->Wt;
    const int64_t ld = m->Npad;
    if (m->blk_cen.empty()) kinv_block_ranges(m);
    const int nblk = (int)(m->blk_cen.size() / std::max(D, 1));
    std::vector<char> need((size_t)nb * nb, 0);
    for (const GTile& t : m->tiles) {
        const double* tab = m->table.data() + (size_t)t.pair * T * W;
        bool read = false;
        for (int k = 0; k < T && !read; ++k) {
            const double* row = tab + (size_t)k * W;
            double emin = 0.0;
            for (int d = 0; d < D; ++d) {
                const double sd = (m->blk_cen[(size_t)d * nblk + t.rb] - m->blk_cen[(size_t)d * nblk + t.cb]) + row[2 + 2 * D + d];
                const double mu = std::fmax(0.0, std::fabs(sd) - m->blk_half[(size_t)d * nblk + t.rb] - m->blk_half[(size_t)d * nblk + t.cb]);
                emin += row[2 + d] * mu * mu;
            }
            read = !(0.5 * emin > 52.0);                      // NaN -> read
        }
        if (!read) continue;
        const int i0 = t.r0 / MOGP_TILE, i1 = (t.r0 + t.nr - 1) / MOGP_TILE, j0 = t.c0 / MOGP_TILE, j1 = (t.c0 + t.nc - 1) / MOGP_TILE;
        for (int i = i0; i <= i1; ++i) for (int j = j0; j <= j1; ++j) if (j <= i) need[(size_t)i * nb + j] = 1;
    }
    for (int i = 0; i < nb; ++i) need[(size_t)i * nb + i] = 1;            // the diagonal tiles always (trace term)
    std::vector<GemmTask> acc, lau;
    std::vector<int> prefix(nb + 1, 0);
    for (int i = 0; i < nb; ++i) {
        for (int j = 0; j <= i; ++j) {
            if (!need[(size_t)i * nb + j]) continue;
            GemmTask a;
            a.a_off = (int64_t)i * MOGP_TILE; a.b_off = (int64_t)j * MOGP_TILE; a.c_off = (int64_t)i * MOGP_TILE * ld + (int64_t)j * MOGP_TILE;
            a.kt = 4 * MOGP_TILE / 16; a.pad = i + 1;
            acc.push_back(a);
            GemmTask l;                                                     // LAUUM: sum over k >= 128 i of W[k, i]^T W[k, j]  (both k-major)
            l.a_off = (int64_t)i * MOGP_TILE * ld + (int64_t)i * MOGP_TILE; l.b_off = (int64_t)i * MOGP_TILE * ld + (int64_t)j * MOGP_TILE;
            l.c_off = a.c_off; l.kt = (int)((ld - (int64_t)i * MOGP_TILE) / 16); l.pad = 0;
            lau.push_back(l);
        }
        prefix[i + 1] = (int)acc.size();
    }
    const double frac = (double)acc.size() / ((double)nb * (nb + 1) / 2);
    m->kinv_fraction = frac;
    if (frac > 0.85) return 0;                                              // little to gain: the dense launches
    const bool same = acc.size() == m->kinv_acc_tasks.size() && (acc.empty() || std::memcmp(acc.data(), m->kinv_acc_tasks.data(), acc.size() * sizeof(GemmTask)) == 0);
    if (!same || m->d_kinv_acc.n < acc.size()) {
        int rc;
        if ((rc = m->d_kinv_acc.ensure(std::max<size_t>(acc.size(), 1)))) return rc;
        if ((rc = m->d_kinv_lauum.ensure(std::max<size_t>(lau.size(), 1)))) return rc;
        // (pageable source: the copy is staged before the call returns, so the vectors may be replaced afterwards)
        HIP_TRY(hipMemcpyAsync(m->d_kinv_acc.p, acc.data(), acc.size() * sizeof(GemmTask), hipMemcpyHostToDevice, m->st));
        HIP_TRY(hipMemcpyAsync(m->d_kinv_lauum.p, lau.data(), lau.size() * sizeof(GemmTask), hipMemcpyHostToDevice, m->st));
        HIP_TRY(hipStreamSynchronize(m->st));
        m->kinv_acc_tasks.swap(acc); m->kinv_lauum_tasks.swap(lau);
    }
    m->kinv_prefix.swap(prefix);
    m->kinv_sparse = true;
    return 0;
}

// Gram + factorisation + inverse factor + alpha.  On return d_A holds W = L^-1, d_alpha = Kj^-1 y.
static int pin_ensure(mogp_model* m, size_t n) {
    if (n <= m->h_pin_n) return 0;
    if (m->h_pin) { hipError_t e = hipHostFree(m->h_pin); (void)e; m->h_pin = nullptr; m->h_pin_n = 0; }
    HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&m->h_pin), n * sizeof(double), hipHostMallocDefault));
    m->h_pin_n = n;
    return 0;
}

static int factorize_finish(mogp_model* m, const GramArgs& ga, double* lml, int64_t* info);

// MOGP_FLOW_DEBUG: where every queue of the dataflow schedule stands, what its next tasks wait for, the private stream's waits; with MOGP_FLOW_TRACE=1
// also the tasks finished per 5 ms.  (1: after a time-out; 2: 60 ms after the evaluation was enqueued, while whatever is stuck is still stuck)
namespace mogp { void flow_debug_dump(mogp_model* m) {
    if (!m->k.flow_flags.p || !m->k.flow_cur) return;
    const FlowPlan& p = *m->k.flow_cur;
    std::vector<unsigned> fl((size_t)p.nflags);
    hipError_t e = hipMemcpy(fl.data(), m->k.flow_flags.p, fl.size() * sizeof(unsigned), hipMemcpyDeviceToHost); (void)e;
    std::vector<unsigned long long> tr;
    if (m->k.flow_trace.p && m->k.flow_trace.n >= FLOW_TRACE_W * p.tasks.size()) {
        tr.resize(FLOW_TRACE_W * p.tasks.size());
        e = hipMemcpy(tr.data(), m->k.flow_trace.p, tr.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost); (void)e;
    }
    fprintf(stderr, "  error word 0x%x\n", fl[p.base_err]);
    std::set<unsigned> holders;
    for (int q = 0; q < p.nq; ++q) {
        const unsigned h = fl[p.base_heads + q];
        const unsigned claimed = std::min<unsigned>(h, (unsigned)p.qsize[q]);
        unsigned done = 0, shown = 0;
        fprintf(stderr, "  queue %2d: head %u of %d", q, h, p.qsize[q]);
        for (unsigned hh = 0; hh < (unsigned)p.qsize[q]; ++hh) {
            const size_t ti = (size_t)p.qbase[q] + hh;
            const FlowTask& t = p.tasks[ti];
            const bool fin = !tr.empty() && tr[FLOW_TRACE_W * ti + 4] != 0;
            if (fin) { ++done; continue; }
            if (tr.empty() && hh + 4 < claimed) continue;              // without the trace: the last claimed ones and the head
            if (hh > claimed || shown >= 6) continue;
            ++shown;
            fprintf(stderr, "\n      %s task %u key %u C(buf %d %d,%d) A(buf %d %d,%d) B(buf %d %d,%d) kt %d var %d:", hh < claimed ? "CLAIMED" : "head   ", hh, t.key,
                    t.cbuf, t.cr, t.cc, t.abuf, t.ar, t.ac, t.bbuf, t.br, t.bc, t.kt, t.var);
            for (int d = 0; d < t.ndep; ++d) fprintf(stderr, " flag[%u]=%u/%u", t.dep[d], fl[t.dep[d]], (unsigned)t.need[d]);
            if (!tr.empty()) {
                const unsigned long long w5 = tr[FLOW_TRACE_W * ti + 5];
                if (w5 >> 63) { fprintf(stderr, "  (HELD by wg %llu%s)", w5 & 0xffff, tr[FLOW_TRACE_W * ti + 1] ? ": STARTED, not finished" : ""); holders.insert((unsigned)(w5 & 0xffff)); }
                else fprintf(stderr, "  (taken by wg %llu: %s)", w5 & 0xffff, tr[FLOW_TRACE_W * ti + 1] ? "running" : "not started");
            }
        }
        if (!tr.empty()) fprintf(stderr, "\n      finished %u", done);
        fprintf(stderr, "\n");
    }
    if (!tr.empty() && !holders.empty()) {                                  // what the workgroups that HOLD unfinished tasks did last
        unsigned long long t0 = ~0ull, tmax = 0;
        for (size_t i = 0; i < p.tasks.size(); ++i) {
            const unsigned long long s1 = tr[FLOW_TRACE_W * i + 1];
            if (s1) { t0 = std::min(t0, s1); tmax = std::max(tmax, std::max(s1, tr[FLOW_TRACE_W * i + 4])); }
        }
        fprintf(stderr, "  last time stamp of the evaluation: %.1f us after its first task\n", (double)(tmax - t0) / 100.0);
        int shown = 0;
        for (unsigned wg : holders) {
            size_t last = (size_t)-1, held = 0;
            for (size_t i = 0; i < p.tasks.size(); ++i) {
                const unsigned long long w5 = tr[FLOW_TRACE_W * i + 5];
                if ((unsigned)(w5 & 0xffff) != wg) continue;
                if (w5 >> 63) { ++held; if (!tr[FLOW_TRACE_W * i + 1]) continue; }
                if (tr[FLOW_TRACE_W * i + 1] && (last == (size_t)-1 || tr[FLOW_TRACE_W * i + 1] > tr[FLOW_TRACE_W * last + 1])) last = i;
            }
            if (shown++ >= 24) break;
            if (last == (size_t)-1) { fprintf(stderr, "  holder wg %u: holds %zu, has started NO task\n", wg, held); continue; }
            const FlowTask& t = p.tasks[last];
            fprintf(stderr, "  holder wg %u (xcc %llu): holds %zu; its last task %zu var %d kt %d: looked %.1f taken %.1f k loop %.1f .. %.1f signalled %.1f us%s\n", wg,
                    (tr[FLOW_TRACE_W * last + 5] >> 16) & 0xff, held, last, t.var, t.kt,
                    (double)((long long)(tr[FLOW_TRACE_W * last + 0] - t0)) / 100.0, (double)((long long)(tr[FLOW_TRACE_W * last + 1] - t0)) / 100.0,
                    (double)((long long)(tr[FLOW_TRACE_W * last + 2] - t0)) / 100.0, (double)((long long)(tr[FLOW_TRACE_W * last + 3] - t0)) / 100.0,
                    tr[FLOW_TRACE_W * last + 4] ? (double)((long long)(tr[FLOW_TRACE_W * last + 4] - t0)) / 100.0 : -1.0, tr[FLOW_TRACE_W * last + 4] ? "" : "  <- NOT FINISHED");
        }
    }
    if (!tr.empty()) {                                                      // tasks finished and workgroups seen per 5 ms
        unsigned long long t0 = ~0ull;
        for (size_t i = 0; i < p.tasks.size(); ++i) if (tr[FLOW_TRACE_W * i + 1] && tr[FLOW_TRACE_W * i + 1] < t0) t0 = tr[FLOW_TRACE_W * i + 1];
        std::map<long, std::pair<int, std::set<unsigned>>> win;
        for (size_t i = 0; i < p.tasks.size(); ++i) {
            const unsigned long long en = tr[FLOW_TRACE_W * i + 4];
            if (!en) continue;
            auto& w = win[(long)((en - t0) / 500000ull)];
            ++w.first; w.second.insert((unsigned)(tr[FLOW_TRACE_W * i + 5] & 0xffff));
        }
        for (auto& kv : win) fprintf(stderr, "  %4ld ms: %6d tasks done by %3zu workgroups\n", kv.first * 5, kv.second.first, kv.second.second.size());
    }
    if (m->k.flow_post.p) {             // the workgroups' own reports: why each left, what it still held -- and who never reported
        std::vector<unsigned> po(m->k.flow_post.n);
        e = hipMemcpy(po.data(), m->k.flow_post.p, po.size() * sizeof(unsigned), hipMemcpyDeviceToHost); (void)e;
        std::vector<unsigned char> dn(p.tasks.size(), 0);
        if (m->k.flow_done.p && m->k.flow_done.n >= dn.size()) { e = hipMemcpy(dn.data(), m->k.flow_done.p, dn.size(), hipMemcpyDeviceToHost); (void)e; }
        const size_t nwg = po.size() / FLOW_POST_W;
        const int ncl = FLOW_NCAS * 8;
        size_t n_to = 0, n_done = 0, n_silent = 0, n_never = 0;
        std::map<size_t, std::vector<unsigned>> holder;          // task -> workgroups that hold its ticket
        for (size_t wgi = 0; wgi < nwg; ++wgi) {
            const unsigned* r = po.data() + wgi * FLOW_POST_W;
            if (r[0] == 1) ++n_to; else if (r[0] == 2) ++n_done; else if (r[2] == 0) ++n_never; else {
                ++n_silent;
                fprintf(stderr, "  workgroup %zu NEVER LEFT: state %u (1 looking, 2 in task %u), last look at clock %u\n", wgi, r[2], r[3], r[4]);
            }
            if (r[0] != 1 && r[0] != 2) continue;
            for (int lane = ncl; lane < 64; ++lane) {
                if (!r[8 + lane]) continue;
                const int q = FLOW_NCAS + (lane - ncl);
                if (q >= p.nq) continue;
                holder[(size_t)p.qbase[q] + (r[8 + lane] - 1)].push_back((unsigned)wgi);
            }
        }
        fprintf(stderr, "  post-mortem: %zu workgroups left on the time-out, %zu left done, %zu never started, %zu never left; %zu tickets held\n", n_to, n_done, n_never, n_silent, holder.size());
        // every task that has NOT signalled although its counters are met: who holds it, and what did the holder's last look say?
        size_t nready = 0, nundone = 0;
        std::set<unsigned> suspects;
        for (int q = 0; q < p.nq; ++q) {
            const unsigned h = fl[p.base_heads + q];
            for (int k = 0; k < p.qsize[q]; ++k) {
                const size_t ti = (size_t)p.qbase[q] + k;
                if (dn[ti]) continue;
                ++nundone;
                const FlowTask& t = p.tasks[ti];
                bool ready = true;
                for (int d = 0; d < t.ndep; ++d) if (fl[t.dep[d]] < (unsigned)t.need[d]) ready = false;
                if (!ready) continue;
                if (nready++ >= 40) continue;
                fprintf(stderr, "  READY BUT NOT DONE: queue %d task %d (head %u) key %u var %d C(buf %d %d,%d):", q, k, h, t.key, t.var, t.cbuf, t.cr, t.cc);
                for (int d = 0; d < t.ndep; ++d) fprintf(stderr, " flag[%u]=%u/%u", t.dep[d], fl[t.dep[d]], (unsigned)t.need[d]);
                auto it = holder.find(ti);
                if (q < FLOW_NCAS) fprintf(stderr, "  [compare-and-swap queue: %s]", (unsigned)k == h ? "AT THE HEAD" : ((unsigned)k < h ? "taken, running or lost" : "behind the head"));
                else if (it == holder.end()) fprintf(stderr, "  [%s]", (unsigned)k < h ? "TICKET GIVEN OUT, NO HOLDER REPORTED IT (running when the kernel froze, or lost)" : "no ticket given out yet");
                else for (unsigned wgi : it->second) {
                    const unsigned* r = po.data() + (size_t)wgi * FLOW_POST_W;
                    const int lane = ncl + (q - FLOW_NCAS);
                    const unsigned long long mr = ((unsigned long long)r[7] << 32) | r[6];
                    fprintf(stderr, "  [held by workgroup %u: %u idle looks, its last look saw this task %s]", wgi, r[1], ((mr >> lane) & 1ull) ? "READY" : "not ready");
                    suspects.insert(wgi);
                }
                fprintf(stderr, "\n");
            }
        }
        fprintf(stderr, "  %zu tasks have not signalled; %zu of them have their counters met\n", nundone, nready);
        // the life of every workgroup that holds such a task, and of the whole grid in numbers (clock: 0.16 us units since the first workgroup's first look)
        unsigned t0 = ~0u;
        for (size_t wgi = 0; wgi < nwg; ++wgi) { const unsigned* r = po.data() + wgi * FLOW_POST_W; if (r[72] && r[72] < t0) t0 = r[72]; }
        auto us = [&](unsigned c) { return c ? 0.16 * (double)(int)(c - t0) : -1.0; };
        for (unsigned wgi : suspects) {
            const unsigned* r = po.data() + (size_t)wgi * FLOW_POST_W;
            fprintf(stderr, "  workgroup %u: XCC %u HW_ID 0x%x (SE %u CU %u); first look at %.0f us, %u tasks run, last task %u started %.0f finished %.0f us, last look %.0f us, left %.0f us after %u idle looks\n",
                    wgi, r[75] & 0xf, r[74], (r[74] >> 13) & 7, (r[74] >> 8) & 15, us(r[72]), r[73], r[3], us(r[76]), us(r[77]), us(r[4]), us(r[5]), r[1]);
            fprintf(stderr, "      its waves' last marks (0x1.. entered, 0x2.. + k block whose successor's loads are in, 0x3.. past the k loop, 0x4.. stored) and when:");
            for (int wv = 0; wv < 8; ++wv) fprintf(stderr, " %x", r[80 + wv]);
            fprintf(stderr, "; wave 0 of its last task: entered %.0f, k block 0 / 8 / 16 / 24 at %.0f / %.0f / %.0f / %.0f, past the loop %.0f, stored %.0f us\n",
                    us(r[88]), us(r[89]), us(r[90]), us(r[91]), us(r[92]), us(r[93]), us(r[94]));
        }
        {
            std::vector<double> first, ntask;
            for (size_t wgi = 0; wgi < nwg; ++wgi) { const unsigned* r = po.data() + wgi * FLOW_POST_W; if (r[72]) { first.push_back(us(r[72])); ntask.push_back((double)r[73]); } }
            std::sort(first.begin(), first.end()); std::sort(ntask.begin(), ntask.end());
            if (!first.empty()) fprintf(stderr, "  first looks: median %.0f us, latest %.0f us; tasks run per workgroup: least %.0f, median %.0f, most %.0f\n", first[first.size() / 2], first.back(), ntask.front(), ntask[ntask.size() / 2], ntask.back());
        }
        // workgroups in the middle of a task when the kernel froze
        for (size_t wgi = 0; wgi < nwg; ++wgi) {
            const unsigned* r = po.data() + wgi * FLOW_POST_W;
            if (r[2] == 2 && r[3] < p.tasks.size() && !dn[r[3]]) fprintf(stderr, "  workgroup %zu was INSIDE task %u when it was last heard of\n", wgi, r[3]);
        }
    }
    for (size_t b = 0; b < p.chain.size(); ++b) {
        const FlowPlan::Chain& c = p.chain[b];
        fprintf(stderr, "  chain %zu done flag[%u]=%u/%u; mini-panel waits", b, c.done_idx, fl[c.done_idx], c.expect);
        for (int k = 0; k < c.t1_nwait; ++k) fprintf(stderr, " flag[%u]=%u/%u", c.t1_widx[k], fl[c.t1_widx[k]], c.t1_wval[k]);
        fprintf(stderr, "; signals from %u:", c.t1_sig_base);
        for (int k = 0; k < 4; ++k) fprintf(stderr, " %u", fl[c.t1_sig_base + k]);
        fprintf(stderr, " (of %u); next-diagonal update waits flag[%u]=%u/%u\n", c.t1_sig_per_row, c.t2_widx, fl[c.t2_widx], c.t2_wval);
    }
} }

// A hand-off inside the persistent chain kernel (chain.hip) timed out: its 13 workgroups were not all resident -- another process sharing the
// GPU holds part of the reserved CUs with its own chain kernel (two such kernels can each hold some of the 16 CUs and wait for the rest).
// Nothing is wrong with the data: drain the streams and repeat the evaluation on the launch-per-step chain, which this model keeps from now on.
#define MOGP_RETRY_NO_CHAIN 0x7e7e
namespace mogp { int chain_fallback(mogp_model* m) {
    if (m->flow_ran) {                           // the dataflow schedule (flow.hip) was on: drop IT first, the chain kernel stays
        // (round 5) ... for a while, not for good: a soak of configs[1] (tools/flow_soak.py) sees one stall of 60 - 900 ms in 2000 - 4000 evaluations on an
        // otherwise idle box -- every workgroup of every kernel of the process standing still, then going on -- and a model that stayed on the stream schedule
        // from its first time-out on trained 20 % slower for the rest of its life.  The stream schedule for the next `flow_backoff` evaluations, four times
        // as many after every further time-out (64, 256, ... 16384): a GPU that really is shared ends up there for good, a hiccup costs one repeated evaluation.
        m->no_flow = true; m->flow_ran = false;
        m->flow_timeouts++;
        m->flow_retry_at = m->n_fact + m->flow_backoff;
        m->flow_backoff = std::min(m->flow_backoff * 4, 16384);
        for (hipStream_t q : {m->st, m->st2, m->st3, m->st4, m->ctx->st5, m->st_priv}) if (q) HIP_TRY(hipStreamSynchronize(q));
        static bool said_flow = false;
        if (!said_flow) {
            said_flow = true;
            unsigned code = 0;                       // which wait gave up: 0x700 an idle workgroup of the dataflow kernel, 0x800 + k a hook of a private-stream launch, else a chain kernel's
            if (m->k.flow_flags.p && m->k.flow_cur && m->k.flow_cur->base_err > 0) { hipError_t e = hipMemcpy(&code, m->k.flow_flags.p + m->k.flow_cur->base_err, sizeof(code), hipMemcpyDeviceToHost); (void)e; }
            fprintf(stderr, "mogp: the dataflow kernel timed out (wait 0x%x; GPU shared with another process?); using the stream schedule for the next %d evaluations (said once)\n", code, (int)(m->flow_retry_at - m->n_fact));
            fprintf(stderr, "mogp: the host enqueued that evaluation in %.0f us (longest so far %.0f us)\n", m->flow_enqueue_us, m->flow_enqueue_us_max);
        }
        if (std::getenv("MOGP_FLOW_DEBUG")) { fprintf(stderr, "mogp: dataflow time-out %d of this model\n", m->flow_timeouts); flow_debug_dump(m); }
        return 0;
    }
    if (m->no_chain) return fail(MOGP_EHIP, "chain kernel: a hand-off timed out although the model is on the launch-per-step chain");
    m->no_chain = true;
    for (hipStream_t q : {m->st, m->st2, m->st3, m->st4, m->ctx->st5, m->st_priv}) if (q) HIP_TRY(hipStreamSynchronize(q));
    static bool said = false;
    if (!said) { said = true; fprintf(stderr, "mogp: the persistent chain kernel timed out (GPU shared with another process?); using the launch-per-step chain\n"); }
    return 0;
} }

// defer: enqueue only -- the scalars travel to the pinned block asynchronously and factorize_finish() (after the caller's ONE stream sync)
// turns them into the LML / the failure report
static int factorize(mogp_model* m, const double* noise_var, const double* data_var, double jitter,
                     double* lml, double* jitter_abs, int64_t* info, bool fuse_inverse = false, bool defer = false, GramArgs* ga_out = nullptr,
                     bool factor_only = false, bool want_inverse = true) {
    const int C = m->C, D = m->D;
    const int64_t N = m->N, Npad = m->Npad;
    if (m->T <= 0) return fail(MOGP_EINVAL, "mogp_model_set_terms must be called before an evaluation");
    if (!noise_var) return fail(MOGP_EINVAL, "noise_var is null");
    { int r__ = ensure_system(m); if (r__) return r__; }
    m->n_fact++;
    if (m->no_flow && m->n_fact >= m->flow_retry_at) m->no_flow = false;        // the dataflow schedule gets another try (chain_fallback)
    m->have_W = m->have_Kinv = false;
    m->factor_only = factor_only;
    m->gemm_ev_used = 0; m->gemm_launches = 0; m->gemm_flops = 0.0;

    // host scalars: mean of the diagonal for the relative jitter (reference gpr/model.py:244)
    double dsum = 0.0;
    if (!m->point_diag.empty()) {       // non-stationary kernels: the caller supplied K_diag per point (mogp_model_set_point_diag)
        for (int c = 0; c < C; ++c)
            for (int k = m->sx.off[c]; k < m->sx.off[c + 1]; ++k) dsum += m->point_diag[k] + noise_var[c];
    } else
    for (int c = 0; c < C; ++c) dsum += (double)(m->sx.off[c + 1] - m->sx.off[c]) * (table_diag(m, c) + noise_var[c]);
    std::vector<double> dv;
    if (data_var) {
        dv.resize(Npad, 0.0);
        for (int64_t pos = 0; pos < N; ++pos) { dv[pos] = data_var[m->sx.perm[pos]]; dsum += dv[pos]; }
        { int r__ = m->d_dvar.ensure(Npad); if (r__) return r__; }
        HIP_TRY(hipMemcpyAsync(m->d_dvar.p, dv.data(), Npad * sizeof(double), hipMemcpyHostToDevice, m->st));
    }
    const double jabs = jitter * dsum / (double)N;
    if (jitter_abs) *jitter_abs = jabs;

    HIP_TRY(hipMemcpyAsync(m->d_noise.p, noise_var, C * sizeof(double), hipMemcpyHostToDevice, m->st));
    const unsigned long long big = std::numeric_limits<unsigned long long>::max();
    HIP_TRY(hipMemcpyAsync(m->d_info.p, &big, sizeof(big), hipMemcpyHostToDevice, m->st));

    int rc;
    if ((rc = mark(m, 0))) return rc;
    GramArgs ga{};
    ga.tiles = m->d_tiles.p; ga.xr = m->d_x.p; ga.xc = m->d_x.p; ga.ldxr = ga.ldxc = Npad; ga.nrows = ga.ncols = N;
    if ((rc = m->ph_xx.prepare(m->sx.off, m->sx.off, C, m->T, Npad, Npad, m->st, ga.ph))) return rc;
    ga.table = m->d_table.p; ga.T = m->T; ga.D = D; ga.C = C; ga.W = m->Wt;
    ga.out = m->k.A.p; ga.ldo = Npad; ga.noise = m->d_noise.p; ga.dvar = data_var ? m->d_dvar.p : nullptr;
    ga.jitter_abs = jabs; ga.mirror = 0;
    ga.ev0 = prof_event(m, 7); ga.ev1 = prof_event(m, 8);
    m->strip.attach(ga);
    // Dataflow schedule: the first chain kernel and the first panel read the first 512 columns only, so the Gram matrix is built in two
    // launches -- those columns on this stream, the rest on the bulk stream in front of the dataflow kernel, i.e. UNDERNEATH the first chain
    // kernel (whose 240 us every workgroup of the dataflow kernel used to sit out after the whole Gram build).  MOGP_GRAM_SPLIT=0: one launch.
    static const bool split_on = !(std::getenv("MOGP_GRAM_SPLIT") && std::atoi(std::getenv("MOGP_GRAM_SPLIT")) == 0);
    // (the prediction's dataflow schedule gains nothing from the split: 45.55 vs 45.59 ms at configs[3], it is throughput-bound)
    const bool split = split_on && fuse_inverse && !factor_only && flow_enabled(m, m->k) && !m->tiles_head.empty() && !m->tiles_tail.empty() && m->st2;
    if (split) {
        GramArgs gh = ga, gt = ga;
        gh.tiles = m->d_tiles_head.p; m->strip_head.attach(gh); gh.ev1 = nullptr;
        gt.tiles = m->d_tiles_tail.p; m->strip_tail.attach(gt); gt.ev0 = nullptr; gt.phases_ready = 1;
        if ((rc = launch_gram(gh, (int)m->tiles_head.size(), m->st))) return rc;
        if ((rc = launch_pad_identity(m->k.A.p, Npad, N, Npad, m->st))) return rc;
        if (!m->gram_ev) HIP_TRY(hipEventCreateWithFlags(&m->gram_ev, hipEventDisableTiming));
        HIP_TRY(hipEventRecord(m->gram_ev, m->st));
        HIP_TRY(hipStreamWaitEvent(m->st2, m->gram_ev, 0));
        if ((rc = launch_gram(gt, (int)m->tiles_tail.size(), m->st2))) return rc;       // spd_potri_flow enqueues the dataflow kernel behind it
        // ... and makes the private stream wait for this event before the first launch that reads beyond the first 512 columns (round 4: with
        // four processes on one GPU the next-diagonal update of block 0 ran BEFORE this launch had written its block: "not positive definite")
        if (!m->gram_tail_ev) HIP_TRY(hipEventCreateWithFlags(&m->gram_tail_ev, hipEventDisableTiming));
        HIP_TRY(hipEventRecord(m->gram_tail_ev, m->st2));
        m->k.tail_ready = m->gram_tail_ev;
    } else {
        if ((rc = launch_gram(ga, (int)m->tiles.size(), m->st))) return rc;
        if ((rc = launch_pad_identity(m->k.A.p, Npad, N, Npad, m->st))) return rc;
    }
    ga.ev0 = ga.ev1 = nullptr;
    if ((rc = mark(m, 1))) return rc;

    // every allocation of this evaluation BEFORE the co-operating kernels are enqueued
    // (the two pivot doubles and the accurate form's right-hand-side block included: a hipHostMalloc / hipMalloc behind the enqueue of kernels that
    // wait for each other is the stall mogp_ctx_create's comment describes)
    if ((rc = pin_ensure(m, (size_t)m->nb + (size_t)((Npad + 3) / 4) + 1 + (size_t)(C * (C + 1) / 2) * m->T * m->Wt + C + 2))) return rc;
    if ((rc = m->d_pivots.ensure(2))) return rc;
    if (m->accurate && (rc = m->acc_rhs.ensure((size_t)Npad * MOGP_TILE))) return rc;
    m->k.flow_used = false;                           // (mogp_model_schedule reports the LAST evaluation: set again by spd_potri_flow)
    m->flow_ran = false;                              // ... and chain_fallback decides from THIS evaluation which schedule to drop, not from an earlier one
    m->k.want_vec = fuse_inverse && !factor_only;     // the dataflow schedule (flow.hip) also forms z = W y and alpha = W^T z
    m->k.vec_y = m->d_y.p; m->k.vec_z = m->d_z.p; m->k.vec_zz = m->d_zz.p; m->k.vec_part = m->d_alpha.p + Npad;
    // (round 6, measured and dropped -- profiles/r6_exact_illcond.txt: the refined factorisation followed by the phases schedule's TRTRI / LAUUM products instead of
    // the two substitutions repairs the LML (2e-10 at cond 7e7) but NOT the gradient (2.8e-4, the fast schedules' 2.0e-4; the substitutions: 5.9e-6): it is the
    // inverse formed through explicit block inverses that costs the gradient its digits, so Kj^-1 stays with trsm.hip here)
    const bool accurate = m->accurate && !fuse_inverse && !factor_only;
    m->accurate_ran = accurate;
    if (accurate) {
        // (round 5) The backward-stable form, for matrices outside the envelope of the schedules below: the launch-per-step Cholesky with
        // every panel refined against L_kk (Spd::refine_panels), then Kj^-1 = L^-T (L^-1 I) by two blocked SUBSTITUTIONS (trsm.hip) instead of
        // products with explicit block inverses, z and alpha by the same substitution on a 128-column block.  2 1/3 N^3 flop at the solves' rate
        // instead of N^3 at the products', behind a launch-per-step factorisation: 43 ms against 10 at N = 8192 (12 factorisation, 24 the two solves in their triangular form, 10 the two vector solves).  mogp_model_set_accurate; the host side switches to it when the pivot range says so.
        m->k.keep_L = true; m->k.refine_panels = true;
        m->k.want_vec = false;
        rc = spd_potrf(m, m->k);
        m->k.keep_L = false; m->k.refine_panels = false;
        m->k.tail_ready = nullptr;
        if (rc) return rc;
        if ((rc = mark(m, 2))) return rc;
        // Round 6: (i) the matrix solve and the two vector solves (128 dependent leaf + update steps, ~10 ms at N = 8192, latency-bound) overlap -- the SMALL launches stay
        // on the main stream (highest priority), the matrix solve goes to the all-CU stream of normal priority: the other way round a 4-workgroup leaf waits until the
        // large launch's queued workgroups have drained (titsias.hip found the same in configs[4]); (ii) Kj^-1 = W^T W with W = L^-1 from ONE substitution (every column a
        // backward-stable solve) and one LAUUM-mode product at the matrix cores' rate, instead of a second substitution L^-T W.  42.7 -> 25 ms (profiles/r6_exact_illcond.txt).
        static const bool acc_aside = !(std::getenv("MOGP_ACC_ASIDE") && std::atoi(std::getenv("MOGP_ACC_ASIDE")) == 0);
        hipStream_t ms = (want_inverse && m->st2u && acc_aside) ? m->st2u : m->st;
        if (want_inverse) {          // (an LML-only evaluation needs L, z and the log-determinant: not the N^2 fill and the N^3 solve nothing would read)
            if (ms != m->st) {
                while ((int)m->k.inv_ev.size() < 4) { hipEvent_t e; HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming)); m->k.inv_ev.push_back(e); }
                HIP_TRY(hipEventRecord(m->k.inv_ev[0], m->st));                  // the factor is complete
                HIP_TRY(hipStreamWaitEvent(ms, m->k.inv_ev[0], 0));
            }
            if ((rc = m->k.Wm.ensure((size_t)Npad * Npad))) return rc;
            HIP_TRY(hipMemsetAsync(m->k.Wm.p, 0, (size_t)Npad * Npad * sizeof(double), ms));      // (above its block diagonal W stays zero: flow.hip relies on it)
            if ((rc = launch_add_diag(m->k.Wm.p, Npad, Npad, 1.0, ms))) return rc;
            if ((rc = trsm_lower(m, m->k.A.p, Npad, m->nb, m->k.Wm.p, Npad, Npad, false, ms, true))) return rc;      // W = L^-1 I, lower block triangle only (trsm.hip: tri)
            GemmArgs g{};
            g.A = m->k.Wm.p; g.lda = Npad; g.a_kmajor = 1; g.B = m->k.Wm.p; g.ldb = Npad; g.b_kmajor = 1;
            g.C = m->k.B.p; g.ldc = Npad; g.alpha = 1.0; g.beta = 0.0; g.mode = GM_LAUUM; g.mt = g.nt = m->nb; g.K = (int)Npad;
            if ((rc = gemm_call(m, g, gemm_flops(g, nullptr), ms))) return rc;
            if (ms != m->st) HIP_TRY(hipEventRecord(m->k.inv_ev[1], ms));
        }
        if ((rc = m->acc_rhs.ensure((size_t)Npad * MOGP_TILE))) return rc;
        HIP_TRY(hipMemsetAsync(m->acc_rhs.p, 0, (size_t)Npad * MOGP_TILE * sizeof(double), m->st));
        if ((rc = launch_copy2d(m->acc_rhs.p, MOGP_TILE, m->d_y.p, 1, Npad, 1, 1.0, m->st))) return rc;
        if ((rc = trsm_lower(m, m->k.A.p, Npad, m->nb, m->acc_rhs.p, MOGP_TILE, MOGP_TILE, false))) return rc;
        if ((rc = launch_copy2d(m->d_z.p, 1, m->acc_rhs.p, MOGP_TILE, Npad, 1, 1.0, m->st))) return rc;
        HIP_TRY(hipMemsetAsync(m->d_zz.p, 0, (size_t)((Npad + 3) / 4) * sizeof(double), m->st));
        if ((rc = launch_gemv_rows(m->d_z.p, Npad, 1, Npad, m->d_z.p, m->d_zz.p, m->st))) return rc;       // z^T z into the first part
        if ((rc = trsm_lower(m, m->k.A.p, Npad, m->nb, m->acc_rhs.p, MOGP_TILE, MOGP_TILE, true))) return rc;
        if ((rc = launch_copy2d(m->d_alpha.p, 1, m->acc_rhs.p, MOGP_TILE, Npad, 1, 1.0, m->st))) return rc;
        if (want_inverse && ms != m->st) HIP_TRY(hipStreamWaitEvent(m->st, m->k.inv_ev[1], 0));
        if ((rc = mark(m, 3))) return rc;
    } else {
    if (factor_only && !fuse_inverse && m->rhs_job && flow_enabled(m, m->k)) rc = spd_potri_flow(m, m->k, m->rhs_job);      // the prediction: factor + substitute as dataflow
    else rc = fuse_inverse ? spd_potri_fused(m, m->k) : spd_potrf(m, m->k);
    m->k.want_vec = false; m->k.tail_ready = nullptr;
    if (rc) return rc;
    if ((rc = mark(m, 2))) return rc;

    if (!fuse_inverse && !factor_only && (rc = spd_trtri(m, m->k))) return rc;
    if ((rc = mark(m, 3))) return rc;
    }

    // ---- z = W y, alpha = W^T z   (factor_only: the caller solves with L itself; the LML is not formed)
    const int nzz_clear = (int)((Npad + 3) / 4);
    if (factor_only) {
        HIP_TRY(hipMemsetAsync(m->d_zz.p, 0, nzz_clear * sizeof(double), m->st));
    } else if (!accurate) {
        const double* Wp = fuse_inverse ? m->k.Wm.p : m->k.A.p;
        m->w_in_Wm = fuse_inverse;
        if (fuse_inverse && m->k.flow_used && m->k.vec_done) {
            if ((rc = launch_flow_alpha_sum(m->k, m->d_alpha.p, m->st))) return rc;
        } else {
            if ((rc = launch_trmv_lower(Wp, Npad, Npad, m->d_y.p, m->d_z.p, m->d_zz.p, m->st))) return rc;
            if ((rc = launch_trmv_lower_t(Wp, Npad, Npad, m->d_z.p, m->d_alpha.p, m->st))) return rc;
        }
    }
    if (fuse_inverse && (rc = spd_potri_fused_finish(m, m->k))) return rc;
    if ((rc = mark(m, 4))) return rc;
    {   // test hook (tests/test_gpu_parity.py: the detour test): every dataflow evaluation reports a hand-off time-out, as if one of its waits had given up
        static const bool fault = std::getenv("MOGP_FLOW_FAULT") && std::atoi(std::getenv("MOGP_FLOW_FAULT")) != 0;
        static const unsigned long long timed_out = MOGP_INFO_CHAIN_TIMEOUT;
        if (fault && m->k.flow_used) HIP_TRY(hipMemcpyAsync(m->d_info.p, &timed_out, sizeof(timed_out), hipMemcpyHostToDevice, m->st));
    }

    // scalars back: [nb log-det parts][nzz z^T z parts][pivot report] through the pinned block
    const int nzz = (int)((Npad + 3) / 4);
    const int nb = m->nb;
    const size_t pin_n = (size_t)nb + nzz + 1 + (size_t)(C * (C + 1) / 2) * m->T * m->Wt + C;
    if ((rc = pin_ensure(m, pin_n + 2))) return rc;
    // the factor's smallest and largest diagonal entry ride back with the scalars (two doubles behind everything else in the block)
    if ((rc = m->d_pivots.ensure(2))) return rc;
    if ((rc = launch_pivot_range(m->k.invd.p, N, m->d_pivots.p, m->st))) return rc;
    HIP_TRY(hipMemcpyAsync(m->h_pin + pin_n, m->d_pivots.p, 2 * sizeof(double), hipMemcpyDeviceToHost, m->st));
    m->pin_pivots = pin_n;
    HIP_TRY(hipMemcpyAsync(m->h_pin, m->k.logdet.p, nb * sizeof(double), hipMemcpyDeviceToHost, m->st));
    static const int zz_piece = []() { const char* e = std::getenv("MOGP_D2H_CHUNK"); const int v = e ? std::atoi(e) : 2048; return v > 0 ? v : (1 << 30); }();
    for (int o = 0; o < nzz; o += zz_piece)               // in pieces of 16 KB: see mogp_ctx_create on larger device-to-host copies next to running co-operating kernels
        HIP_TRY(hipMemcpyAsync(m->h_pin + nb + o, m->d_zz.p + o, std::min(zz_piece, nzz - o) * sizeof(double), hipMemcpyDeviceToHost, m->st));
    HIP_TRY(hipMemcpyAsync(m->h_pin + nb + nzz, m->d_info.p, sizeof(unsigned long long), hipMemcpyDeviceToHost, m->st));
    if (ga_out) *ga_out = ga;
    if (defer) return 0;
    HIP_TRY(hipStreamSynchronize(m->st));
    return factorize_finish(m, ga, lml, info);
}

static int factorize_finish(mogp_model* m, const GramArgs& ga, double* lml, int64_t* info) {
    const int64_t N = m->N, Npad = m->Npad;
    const int nzz = (int)((Npad + 3) / 4), nb = m->nb;
    const unsigned long long big = std::numeric_limits<unsigned long long>::max();
    unsigned long long hinfo = 0;
    std::memcpy(&hinfo, m->h_pin + nb + nzz, sizeof(hinfo));
    int rc;
    if (hinfo == MOGP_INFO_CHAIN_TIMEOUT) return MOGP_RETRY_NO_CHAIN;      // the caller repeats the evaluation on the launch-per-step chain
    static const bool fake = std::getenv("MOGP_FAKE_K") && std::atoi(std::getenv("MOGP_FAKE_K")) > 1;    // timing experiment: the numbers are wrong on purpose
    if (hinfo != big && !fake) {
        if (info) *info = (int64_t)hinfo;
        // distinguish NaN / Inf in the Gram from a plain indefinite matrix (reference prints which, gpr/model.py:249-252)
        int flag = 0;
        HIP_TRY(hipMemsetAsync(m->d_flag.p, 0, sizeof(int), m->st));
        if ((rc = launch_gram(ga, (int)m->tiles.size(), m->st))) return rc;
        if ((rc = launch_nonfinite_scan(m->k.A.p, Npad, N, m->d_flag.p, m->st))) return rc;
        HIP_TRY(hipMemcpyAsync(&flag, m->d_flag.p, sizeof(int), hipMemcpyDeviceToHost, m->st));
        HIP_TRY(hipStreamSynchronize(m->st));
        if (flag & 1) return fail(MOGP_ENONFINITE, "linalg.cholesky: kernel matrix has NaNs!");
        if (flag & 2) return fail(MOGP_ENONFINITE, "linalg.cholesky: kernel matrix has infinities!");
        return fail(MOGP_ENOTPD, "linalg.cholesky: The factorization could not be completed because the input is not "
                                 "positive-definite (the leading minor of order " + std::to_string(hinfo) + " is not positive-definite).");
    }
    m->pivot_min = m->h_pin[m->pin_pivots]; m->pivot_max = m->h_pin[m->pin_pivots + 1];
    double logdet = 0.0, zz = 0.0;
    for (int i = 0; i < nb; ++i) logdet += m->h_pin[i];
    for (int i = 0; i < nzz; ++i) zz += m->h_pin[nb + i];
    if (lml) *lml = -0.5 * (double)N * std::log(2.0 * M_PI) - logdet - 0.5 * zz;
    m->have_W = !m->factor_only && !m->accurate_ran;     // (the accurate form keeps L, not W = L^-1)
    return 0;
}

static void collect_timing(mogp_model* m, int last_mark) {
    if (!m->profiling) return;
    auto el = [&](int a, int b) { float t = 0.f; if (hipEventElapsedTime(&t, m->ev[a], m->ev[b]) != hipSuccess) t = 0.f; return (double)t; };
    std::fill(m->ms, m->ms + MOGP_ST_COUNT, 0.0);
    m->ms[MOGP_ST_GRAM] = el(0, 1);
    m->ms[MOGP_ST_POTRF] = el(1, 2);
    m->ms[MOGP_ST_TRTRI] = el(2, 3);
    m->ms[MOGP_ST_SOLVE] = el(3, 4);
    if (last_mark >= 6) { m->ms[MOGP_ST_LAUUM] = el(4, 5); m->ms[MOGP_ST_MOMENTS] = el(5, 6); }
    m->ms[MOGP_ST_TOTAL] = el(0, last_mark);
    if ((int)m->ev.size() > 8) m->ms[MOGP_ST_GRAM_KERNEL] = el(7, 8);
    if (last_mark >= 6 && (int)m->ev.size() > 10) m->ms[MOGP_ST_MOMENT_KERNEL] = el(9, 10);
    double gsum = 0.0;
    for (size_t i = 0; i + 1 < m->gemm_ev_used; i += 2) {
        float t = 0.f;
        if (hipEventElapsedTime(&t, m->gemm_ev[i], m->gemm_ev[i + 1]) == hipSuccess) gsum += t;
    }
    m->ms[MOGP_ST_GEMM_KERNEL] = gsum;
}

// ---- gradient evaluation on the sweep inversion: Gram -> A = -Kj^-1 (one sweep) -> alpha, LML -------------------------
// sweep_eval_begin: uploads, Gram (lower, noise + jitter on the diagonal), padding.  m->sh_jabs keeps the absolute jitter.
static int sweep_eval_begin(mogp_model* m, const double* noise_var, const double* data_var, double jitter) {
    const int C = m->C, D = m->D;
    const int64_t N = m->N, Npad = m->Npad;
    if (m->T <= 0) return fail(MOGP_EINVAL, "mogp_model_set_terms must be called before an evaluation");
    if (!noise_var) return fail(MOGP_EINVAL, "noise_var is null");
    { int r__ = ensure_system(m); if (r__) return r__; }
    m->have_W = m->have_Kinv = false;
    m->gemm_ev_used = 0; m->gemm_launches = 0; m->gemm_flops = 0.0;
    double dsum = 0.0;
    if (!m->point_diag.empty()) {
        for (int c = 0; c < C; ++c)
            for (int k = m->sx.off[c]; k < m->sx.off[c + 1]; ++k) dsum += m->point_diag[k] + noise_var[c];
    } else
    for (int c = 0; c < C; ++c) dsum += (double)(m->sx.off[c + 1] - m->sx.off[c]) * (table_diag(m, c) + noise_var[c]);
    m->sh_dvar = data_var != nullptr;
    if (data_var) {
        std::vector<double> dv(Npad, 0.0);
        for (int64_t pos = 0; pos < N; ++pos) { dv[pos] = data_var[m->sx.perm[pos]]; dsum += dv[pos]; }
        { int r__ = m->d_dvar.ensure(Npad); if (r__) return r__; }
        HIP_TRY(dev_upload(m->d_dvar.p, dv.data(), Npad * sizeof(double)));
    }
    m->sh_jabs = jitter * dsum / (double)N;
    HIP_TRY(hipMemcpyAsync(m->d_noise.p, noise_var, C * sizeof(double), hipMemcpyHostToDevice, m->st));
    const unsigned long long big = std::numeric_limits<unsigned long long>::max();
    HIP_TRY(hipMemcpyAsync(m->d_info.p, &big, sizeof(big), hipMemcpyHostToDevice, m->st));
    int rc;
    if ((rc = mark(m, 0))) return rc;
    GramArgs ga{};
    ga.tiles = m->d_tiles.p; ga.xr = m->d_x.p; ga.xc = m->d_x.p; ga.ldxr = ga.ldxc = Npad; ga.nrows = ga.ncols = N;
    if ((rc = m->ph_xx.prepare(m->sx.off, m->sx.off, C, m->T, Npad, Npad, m->st, ga.ph))) return rc;
    ga.table = m->d_table.p; ga.T = m->T; ga.D = D; ga.C = C; ga.W = m->Wt;
    ga.out = m->k.A.p; ga.ldo = Npad; ga.noise = m->d_noise.p; ga.dvar = data_var ? m->d_dvar.p : nullptr;
    ga.jitter_abs = m->sh_jabs; ga.mirror = 0;
    const bool own = m->sh_n > 1 && m->own_n == m->sh_n && m->own_rank == m->sh_rank;
    if (own) ga.tiles = m->d_tiles_own.p;
    (own ? m->strip_own : m->strip).attach(ga);
    if ((rc = launch_gram(ga, (int)(own ? m->tiles_own.size() : m->tiles.size()), m->st))) return rc;
    // (owned-rows form: the padding rows lie in the last tile row -- its owner's business; everybody else gets them with the pivot block)
    if (!(m->sh_owned && m->sh_n > 1 && (m->nb - 1) % m->sh_n != m->sh_rank))
        if ((rc = launch_pad_identity(m->k.A.p, Npad, N, Npad, m->st))) return rc;
    if ((rc = mark(m, 1))) return rc;
    return 0;
}

// alpha (or this rank's partial sums of it) = -A y into m->d_alpha
static int sweep_eval_alpha(mogp_model* m) {
    const int64_t Npad = m->Npad;
    int rc;
    if ((rc = mark(m, 2))) return rc;
    if ((rc = mark(m, 3))) return rc;
    const int nchunks = (int)((Npad + 511) / 512);
    if ((rc = m->d_symv.ensure((size_t)(4 + nchunks) * Npad))) return rc;
    const int rm = m->sh_n > 1 ? m->sh_n : 0;
    if ((rc = launch_symv_lower(m->k.A.p, Npad, Npad, m->d_y.p, m->d_alpha.p, m->d_symv.p, -1.0, m->st, rm, m->sh_rank))) return rc;
    if ((rc = mark(m, 4))) return rc;
    return 0;
}

// scalars back: failure report, log-det (every rank factors every pivot block, so it is complete everywhere), y^T alpha
static int sweep_eval_scalars(mogp_model* m, double* lml, int64_t* info) {
    const int64_t N = m->N, Npad = m->Npad;
    const int nb = m->nb;
    const unsigned long long big = std::numeric_limits<unsigned long long>::max();
    std::vector<double> hl(nb), ha(Npad);
    unsigned long long hinfo = 0;
    HIP_TRY(hipMemcpyAsync(hl.data(), m->k.logdet.p, nb * sizeof(double), hipMemcpyDeviceToHost, m->st));
    HIP_TRY(hipMemcpyAsync(ha.data(), m->d_alpha.p, Npad * sizeof(double), hipMemcpyDeviceToHost, m->st));
    HIP_TRY(hipMemcpyAsync(&hinfo, m->d_info.p, sizeof(hinfo), hipMemcpyDeviceToHost, m->st));
    HIP_TRY(hipStreamSynchronize(m->st));
    if (hinfo == MOGP_INFO_CHAIN_TIMEOUT) return MOGP_RETRY_NO_CHAIN;
    if (hinfo != big) {
        if (info) *info = (int64_t)hinfo;
        return fail(MOGP_ENOTPD, "linalg.cholesky: The factorization could not be completed because the input is not "
                                 "positive-definite (the leading minor of order " + std::to_string(hinfo) + " is not positive-definite).");
    }
    double logdet = 0.0, ya = 0.0;
    for (double v : hl) logdet += v;
    for (int64_t i = 0; i < N; ++i) ya += m->hy[i] * ha[i];
    if (lml) *lml = -0.5 * (double)N * std::log(2.0 * M_PI) - logdet - 0.5 * ya;
    return 0;
}

// gradient-moment pass over this rank's rows of Kj^-1 (all rows when not sharded): results in m->d_moments / m->d_diagG
static int moment_pass_device(mogp_model* m, const double* kinv, double ksign) {
    const int C = m->C, D = m->D, W = m->Wt, T = m->T, P = C * (C + 1) / 2;
    const int64_t Npad = m->Npad;
    const int rm = m->sh_n > 1 ? m->sh_n : 0;
    const bool own = m->sh_n > 1 && m->own_n == m->sh_n && m->own_rank == m->sh_rank;
    int rc;
    MomentArgs ma{};
    ma.tiles = own ? m->d_tiles_own.p : m->d_tiles.p; ma.ntiles = (int)(own ? m->tiles_own.size() : m->tiles.size());
    ma.x = m->d_x.p; ma.ldx = Npad; ma.nrows = ma.ncols = m->N;
    if ((rc = m->ph_xx.prepare(m->sx.off, m->sx.off, C, T, Npad, Npad, m->st, ma.ph))) return rc;
    ma.table = m->d_table.p; ma.T = T; ma.D = D; ma.C = C; ma.W = W; ma.kinv = kinv; ma.kinv_sign = ksign; ma.ld = Npad; ma.alpha = m->d_alpha.p;
    ma.row_mod = rm; ma.row_rem = m->sh_rank;
    ma.partial = m->d_partial.p;
    ma.phases_ready = 1;                       // ph_xx was filled by this evaluation's Gram launch: same inputs, same table
    ma.ev0 = prof_event(m, 9); ma.ev1 = prof_event(m, 10);
    if ((rc = launch_moments(ma, m->st))) return rc;
    if ((rc = launch_moment_reduce(m->d_partial.p, own ? m->d_pair_start_own.p : m->d_pair_start.p, P, T, W, D, m->d_moments.p, m->st))) return rc;
    if ((rc = launch_diagG(kinv, Npad, m->d_alpha.p, m->d_chan_off.p, C, m->d_diagG.p, m->st, ksign, rm, m->sh_rank))) return rc;
    return mark(m, 6);
}

static int moment_pass(mogp_model* m, const double* kinv, double ksign, double* moments, double* diagG) {
    const int C = m->C, W = m->Wt, T = m->T, P = C * (C + 1) / 2;
    int rc;
    if ((rc = moment_pass_device(m, kinv, ksign))) return rc;
    HIP_TRY(hipMemcpyAsync(moments, m->d_moments.p, (size_t)P * T * W * sizeof(double), hipMemcpyDeviceToHost, m->st));
    HIP_TRY(hipMemcpyAsync(diagG, m->d_diagG.p, C * sizeof(double), hipMemcpyDeviceToHost, m->st));
    HIP_TRY(hipStreamSynchronize(m->st));
    return 0;
}

static int eval_sweep(mogp_model* m, const double* noise_var, const double* data_var, double jitter,
                      double* lml, double* jitter_abs, int64_t* info) {
    int rc;
    if ((rc = sweep_eval_begin(m, noise_var, data_var, jitter))) return rc;
    if (jitter_abs) *jitter_abs = m->sh_jabs;
    if ((rc = spd_sweep(m, m->k))) return rc;
    if ((rc = sweep_eval_alpha(m))) return rc;
    return sweep_eval_scalars(m, lml, info);
}

extern "C" {

// the context's streams: critical (high priority, all CUs), private (reserved CUs only), two bulk streams (everything else)
static int ctx_streams(mogp_ctx* ctx) {
    if (ctx->streams_ready) return 0;
    int lo_prio = 0;
    {
        int lo = 0, hi = 0;
        HIP_TRY(hipDeviceGetStreamPriorityRange(&lo, &hi));
        lo_prio = lo;
        HIP_TRY(hipStreamCreateWithPriority(&ctx->st, hipStreamNonBlocking, hi));
        // MOGP_RESERVE_CUS = R compute units of every XCD are kept for the latency-bound intra-block chain of the fused
        // factorisation + inversion (potri.hip): CU-mask bit i is CU (i / 8) of XCD (i % 8) on gfx950 (tools/micro/cumask.hip),
        // so the first 8 R bits are R CUs from each XCD.  st_priv runs ONLY there, the bulk streams everywhere else: a 1-workgroup
        // leaf that shares its CU with bulk GEMM waves runs 1.6-3x slower (measured), and the dispatcher does not avoid that by itself.
        const char* er = std::getenv("MOGP_RESERVE_CUS");
        const int reserve = er ? std::atoi(er) : 2;
        hipDeviceProp_t prop;
        HIP_TRY(hipGetDeviceProperties(&prop, ctx->device));
        const int ncu = prop.multiProcessorCount;
        const bool masked = reserve > 0 && 16 * reserve < ncu;
        ctx->ncu = ncu; ctx->ncu_reserved = masked ? 8 * reserve : 0;
        ctx->chain_ok = !masked || 8 * reserve >= 13;        // one 128 KB workgroup per CU: fewer reserved CUs than workgroups would never all be resident
        if (masked) {
            std::vector<uint32_t> bulk((ncu + 31) / 32, 0u), priv((ncu + 31) / 32, 0u);
            for (int i = 0; i < ncu; ++i) (i < 8 * reserve ? priv : bulk)[i / 32] |= 1u << (i % 32);
            HIP_TRY(hipExtStreamCreateWithCUMask(&ctx->st_priv, (uint32_t)priv.size(), priv.data()));
            HIP_TRY(hipExtStreamCreateWithCUMask(&ctx->st2, (uint32_t)bulk.size(), bulk.data()));
            HIP_TRY(hipExtStreamCreateWithCUMask(&ctx->st3, (uint32_t)bulk.size(), bulk.data()));
            HIP_TRY(hipExtStreamCreateWithCUMask(&ctx->st4, (uint32_t)bulk.size(), bulk.data()));
        } else {
            HIP_TRY(hipStreamCreateWithPriority(&ctx->st2, hipStreamNonBlocking, (lo + hi) / 2));
            HIP_TRY(hipStreamCreateWithPriority(&ctx->st3, hipStreamNonBlocking, lo));
            HIP_TRY(hipStreamCreateWithPriority(&ctx->st4, hipStreamNonBlocking, lo));
        }
    }
    {   // bulk stream over ALL CUs (full mask), used once an evaluation is flop-bound
        hipDeviceProp_t prop;
        HIP_TRY(hipGetDeviceProperties(&prop, ctx->device));
        std::vector<uint32_t> all((prop.multiProcessorCount + 31) / 32, 0u);
        for (int i = 0; i < prop.multiProcessorCount; ++i) all[i / 32] |= 1u << (i % 32);
        if (ctx->st_priv) HIP_TRY(hipExtStreamCreateWithCUMask(&ctx->st2u, (uint32_t)all.size(), all.data()));
        else HIP_TRY(hipStreamCreateWithPriority(&ctx->st2u, hipStreamNonBlocking, lo_prio));
    }
    ctx->streams_ready = true;
    return 0;
}

int mogp_model_create(mogp_ctx* ctx, int64_t N, int D, int C, const double* X, const double* y, mogp_model** out) {
    if (!ctx || !X || !y || !out) return fail(MOGP_EINVAL, "mogp_model_create: null argument");
    *out = nullptr;
    if (N <= 0 || D <= 0 || D > MOGP_MAXD || C <= 0) return fail(MOGP_EINVAL, "mogp_model_create: need N > 0, 0 < D <= 8, C > 0");
    int rc;
    if ((rc = use_device(ctx))) return rc;
    mogp_model* m = new mogp_model();
    m->ctx = ctx; m->N = N; m->D = D; m->C = C;
    if ((rc = sort_inputs(X, N, D, C, MOGP_TILE, m->sx))) { delete m; return rc; }
    m->Npad = m->sx.Mpad;
    m->nb = (int)(m->Npad / MOGP_TILE);
    const int64_t Npad = m->Npad;
    const int nchunks = (int)((Npad + 511) / 512);
#define MOGP_OUTER_DEFINED 1
#define TRY_RC(x) do { int r__ = (x); if (r__) { mogp_model_destroy(m); return r__; } } while (0)
#define TRY_HIP(x) do { hipError_t e__ = (x); if (e__ != hipSuccess) { int r__ = hip_fail(e__, #x, __FILE__, __LINE__); mogp_model_destroy(m); return r__; } } while (0)
    TRY_RC(ctx_streams(ctx));
    m->st = ctx->st; m->st2 = ctx->st2; m->st2u = ctx->st2u; m->st3 = ctx->st3; m->st4 = ctx->st4; m->st_priv = ctx->st_priv;
    // the N x N system (two Npad^2 matrices: 160 GB at N = 100 000), the tile lists over (X, X) and their partial-moment buffer (N^2 / 4096
    // tiles) are set up by the first call that needs them (ensure_system): the sparse and variational models never do
    TRY_RC(m->d_x.ensure((size_t)D * Npad));
    TRY_RC(m->d_y.ensure(Npad));
    TRY_RC(m->d_noise.ensure(C));
    TRY_RC(m->d_z.ensure(Npad));
    TRY_RC(m->d_alpha.ensure((size_t)(1 + nchunks) * Npad));
    TRY_RC(m->d_zz.ensure((Npad + 3) / 4));
    TRY_RC(m->d_diagG.ensure(C));
    TRY_RC(m->d_info.ensure(2));                      // [1]: the first factorisation's report of a sparse evaluation (launch_info_stash)
    TRY_RC(m->d_flag.ensure(1));
    TRY_RC(m->d_chan_off.ensure(C + 1));
    TRY_HIP(dev_upload(m->d_x.p, m->sx.xs.data(), (size_t)D * Npad * sizeof(double)));
    TRY_HIP(dev_upload(m->d_chan_off.p, m->sx.off.data(), (C + 1) * sizeof(int)));
    TRY_RC(mogp_model_set_y(m, y));
#undef TRY_RC
#undef TRY_HIP
    *out = m;                                   // only a fully built model is handed out (a failure above has destroyed it)
    return MOGP_OK;
}

int mogp_model_destroy(mogp_model* m) {
    if (!m) return MOGP_OK;
    if (m->ctx) { hipError_t e = hipSetDevice(m->ctx->device); (void)e; }
    if (m->st) { hipError_t e = hipStreamSynchronize(m->st); (void)e; }
    for (auto e : m->ev) { hipError_t r = hipEventDestroy(e); (void)r; }
    for (auto e : m->gemm_ev) { hipError_t r = hipEventDestroy(e); (void)r; }
    for (auto& e : m->pred_ev) if (e) { hipError_t r = hipEventDestroy(e); (void)r; e = nullptr; }
    for (hipStream_t q : {m->st2, m->st2u, m->st3, m->st4, m->ctx->st5, m->st_priv}) if (q) { hipError_t e = hipStreamSynchronize(q); (void)e; }
    m->k.release(); m->ws.release(); m->ws_tail.release();
    for (int b = 0; b < 2; ++b) { m->swU[b].release(); m->swUr[b].release(); m->swXr[b].release(); }
    for (auto e : m->sw_ev) { hipError_t r = hipEventDestroy(e); (void)r; }
    for (auto e : m->sh_prof) { hipError_t r = hipEventDestroy(e); (void)r; }
    m->d_symv.release(); m->sh_send.release(); m->sh_recv.release(); m->sh_send1.release(); m->sh_recv1.release(); m->sh_fact.release();
    for (auto e : m->sh_ev) { hipError_t r = hipEventDestroy(e); (void)r; }
    if (m->tw) { m->tw->release(); delete m->tw; m->tw = nullptr; }
    m->oa.release();
    m->d_x.release(); m->d_y.release(); m->d_table.release();
    m->d_noise.release(); m->d_dvar.release(); m->d_z.release(); m->d_alpha.release(); m->d_zz.release();
    m->d_partial.release(); m->d_moments.release(); m->d_diagG.release(); m->d_tiles.release(); m->d_pair_start.release(); m->strip.release(); m->strip_own.release();
    m->d_tiles_head.release(); m->d_tiles_tail.release(); m->strip_head.release(); m->strip_tail.release();
    if (m->gram_ev) { hipError_t r = hipEventDestroy(m->gram_ev); (void)r; m->gram_ev = nullptr; }
    if (m->gram_tail_ev) { hipError_t r = hipEventDestroy(m->gram_tail_ev); (void)r; m->gram_tail_ev = nullptr; }
    m->d_chan_off.release(); m->d_flag.release(); m->d_info.release(); m->d_pivots.release(); m->acc_rhs.release();
    m->d_xs.release(); m->d_Ksf.release(); m->d_Vt.release(); m->d_mu.release(); m->d_var.release(); m->d_kdiag.release();
    m->d_Kss.release(); m->d_ptiles.release(); m->d_pred_tasks.release();
    m->ph_xx.release(); m->ph_sx.release(); m->ph_ss.release();
    if (m->h_pin) { hipError_t e = hipHostFree(m->h_pin); (void)e; m->h_pin = nullptr; }

    delete m;
    return MOGP_OK;
}

int mogp_model_set_y(mogp_model* m, const double* y) {
    if (!m || !y) return fail(MOGP_EINVAL, "mogp_model_set_y: null argument");
    int rc;
    if ((rc = use_device(m->ctx))) return rc;
    std::vector<double> ys(m->Npad, 0.0);
    for (int64_t pos = 0; pos < m->N; ++pos) ys[pos] = y[m->sx.perm[pos]];
    HIP_TRY(dev_upload(m->d_y.p, ys.data(), m->Npad * sizeof(double)));
    m->hy = ys;
    return MOGP_OK;
}

int mogp_model_set_terms_ex(mogp_model* m, int T, int width, const double* table) {
    if (!m || !table || T <= 0) return fail(MOGP_EINVAL, "mogp_model_set_terms: bad argument");
    if (width != 2 + 3 * m->D && width != 2 + 5 * m->D) return fail(MOGP_EINVAL, "mogp_model_set_terms: the row width must be 2 + 3 D or 2 + 5 D");
    int rc;
    if ((rc = use_device(m->ctx))) return rc;
    const int W = width;
    const size_t n = (size_t)m->C * m->C * T * W;
    for (size_t i = 0; i < n; ++i)
        if (!std::isfinite(table[i])) return fail(MOGP_ENONFINITE, "spectral term table has non-finite entries (kernel parameters diverged)");
    m->T = T;
    m->Wt = W;
    if (m->tw) m->tw->pred_valid = false;       // mogp_sparse_predict_cov combines the last prediction's panels with the CURRENT table: a new table ends that
    m->table.assign(table, table + n);
    if ((rc = m->d_table.ensure(n))) return rc;
    if ((rc = m->d_moments.ensure((size_t)(m->C * (m->C + 1) / 2) * T * W))) return rc;
    HIP_TRY(hipMemcpyAsync(m->d_table.p, m->table.data(), n * sizeof(double), hipMemcpyHostToDevice, m->st));
    return MOGP_OK;
}

int mogp_model_set_terms(mogp_model* m, int T, const double* table) {
    if (!m) return fail(MOGP_EINVAL, "mogp_model_set_terms: bad argument");
    return mogp_model_set_terms_ex(m, T, 2 + 3 * m->D, table);
}

int mogp_model_set_point_diag(mogp_model* m, const double* kdiag) {
    if (!m) return fail(MOGP_EINVAL, "mogp_model_set_point_diag: model is null");
    m->point_diag.clear();
    if (kdiag) {
        m->point_diag.resize(m->N);
        for (int64_t pos = 0; pos < m->N; ++pos) m->point_diag[pos] = kdiag[m->sx.perm[pos]];
    }
    return MOGP_OK;
}

int mogp_exact_eval(mogp_model* m, const double* noise_var, const double* data_var, double jitter, int flags,
                    double* lml, double* moments, double* diagG, double* trG, double* jitter_abs, int64_t* info) {
    if (!m) return fail(MOGP_EINVAL, "mogp_exact_eval: model is null");
    int rc;
    if ((rc = use_device(m->ctx))) return rc;
    if (info) *info = 0;
    one_gpu_call(m);
    // Gradient evaluation, three schedules of the same arithmetic (MOGP_GRAD_PATH = fused | phases | sweep overrides the choice):
    //   fused   potri.hip: the inverse streamed behind the Cholesky chain.  Wins while the serial chain dominates: 15.1 vs 15.9 ms at
    //           N = 8192, 20.1 vs 21.1 ms at N = 9216, even at N = 10240 -- the default up to 80 tile rows (112 as dataflow, below).
    //   phases  POTRF, TRTRI, LAUUM one after the other: fewer, larger GEMM launches.  Wins once the evaluation is flop-bound
    //           (38.8 vs 41.8 ms at N = 12288, 80.6 vs 90.1 ms at N = 16384, 569 vs 657 ms at N = 32768) -- the default above.
    //   sweep   sweep.hip: single-sweep blocked inversion; slower on one GPU (47 evals/s at N = 8192) but with one panel
    //           exchange per pivot block, which is what the sharded multi-GPU evaluation (mogp_shard_*) is built on.
    static const std::string grad_path = []() { const char* e = std::getenv("MOGP_GRAD_PATH"); return std::string(e ? e : ""); }();
    const bool sweep = grad_path == "sweep" && (flags & MOGP_EVAL_GRAD);
    const bool grad = (flags & MOGP_EVAL_GRAD) != 0;
    GramArgs ga{};
    if ((rc = ensure_system(m))) return rc;
    if ((rc = kinv_plan(m, grad && !sweep && !m->accurate))) return rc;
    // round 4: as tile dataflow (flow.hip) the fused schedule also beats the phases at 81 .. 112 tile rows (configs[1]'s kernel, tools/r4_sizes.sh:
    // 19.3 vs 22.4 ms at N = 10240, 32.5 vs 35.9 at 12288, 41.0 vs 43.8 at 13312, 50.7 vs 53.0 at 14336; 76.6 vs 75.8 the other way at 16384); where
    // the dataflow kernel is not available (switched off, fallen back, a planned inverse) the stream form keeps its 80
    const int fused_max = flow_enabled(m, m->k) ? 112 : 80;
    const bool fused = !sweep && grad && !m->accurate && (grad_path == "fused" || (grad_path != "phases" && m->nb <= fused_max));
    if (sweep) {
        m->pivot_min = m->pivot_max = 0.0;                     // (the sweep reports no pivot range)
        if ((rc = eval_sweep(m, noise_var, data_var, jitter, lml, jitter_abs, info))) {
            if (rc != MOGP_RETRY_NO_CHAIN) return rc;
            if ((rc = chain_fallback(m))) return rc;
            return mogp_exact_eval(m, noise_var, data_var, jitter, flags, lml, moments, diagG, trG, jitter_abs, info);
        }
    }
    else if ((rc = factorize(m, noise_var, data_var, jitter, lml, jitter_abs, info, fused, grad, &ga, false, grad))) {
        if (rc != MOGP_RETRY_NO_CHAIN) return rc;
        if ((rc = chain_fallback(m))) return rc;
        return mogp_exact_eval(m, noise_var, data_var, jitter, flags, lml, moments, diagG, trG, jitter_abs, info);
    }
    if (!grad) { collect_timing(m, 4); return MOGP_OK; }
    if (!moments || !diagG || !trG) return fail(MOGP_EINVAL, "mogp_exact_eval: gradient outputs are null");

    const int C = m->C, W = m->Wt, T = m->T, P = C * (C + 1) / 2;

    // K^-1: the sweep left -Kj^-1 in k.A; the POTRF path needs W^T W (lower tiles, full diagonal tiles) in k.B
    if (!sweep && !fused && !m->accurate_ran && (rc = spd_lauum(m, m->k))) return rc;          // (the accurate form left Kj^-1 itself in k.B)
    const double* kinv = sweep ? m->k.A.p : m->k.B.p;
    const double ksign = sweep ? -1.0 : 1.0;
    if ((rc = mark(m, 5))) return rc;
    if (sweep) {
        if ((rc = moment_pass(m, kinv, ksign, moments, diagG))) return rc;
    } else {
        // everything of this evaluation is enqueued before the host waits ONCE: scalars and moments come back through the pinned block
        if ((rc = moment_pass_device(m, kinv, ksign))) return rc;
        const size_t off = (size_t)m->nb + (size_t)((m->Npad + 3) / 4) + 1;
        HIP_TRY(hipMemcpyAsync(m->h_pin + off, m->d_moments.p, (size_t)P * T * W * sizeof(double), hipMemcpyDeviceToHost, m->st));
        HIP_TRY(hipMemcpyAsync(m->h_pin + off + (size_t)P * T * W, m->d_diagG.p, C * sizeof(double), hipMemcpyDeviceToHost, m->st));
        if ((rc = wait_stream(m->st))) return rc;
        if ((rc = factorize_finish(m, ga, lml, info))) {
            if (rc != MOGP_RETRY_NO_CHAIN) return rc;
            if ((rc = chain_fallback(m))) return rc;
            return mogp_exact_eval(m, noise_var, data_var, jitter, flags, lml, moments, diagG, trG, jitter_abs, info);
        }
        std::memcpy(moments, m->h_pin + off, (size_t)P * T * W * sizeof(double));
        std::memcpy(diagG, m->h_pin + off + (size_t)P * T * W, C * sizeof(double));
    }
    double tr = 0.0;
    for (int c = 0; c < C; ++c) tr += diagG[c];
    *trG = tr;
    m->have_Kinv = true;
    m->kinv_in_A = sweep;
    collect_timing(m, 6);
    return MOGP_OK;
}

// mean_w (caller order, may be null): the predictive mean is K_sf mean_w instead of K_sf Kj^-1 y (the Opper-Archambeau model, whose mean
// weights are variational parameters; its variance is the exact one with the per-point variances 1 / lambda^2)
static int predict_core(mogp_model* m, const double* noise_var, const double* data_var, double jitter,
                        const double* kss_diag, int64_t S, const double* Xs, int full,
                        double* mu, double* var, int64_t* info, const double* mean_w) {
    int rc;
    if ((rc = use_device(m->ctx))) return rc;
    if (info) *info = 0;
    one_gpu_call(m);
    // Cholesky factor only: the predictive equations need V = L^-1 K_fs and z = L^-1 y, never L^-1 itself (reference gpr/model.py:470-472 solves).
    // Round 1 / 2a formed W = L^-1 (N^3/3 flop) and multiplied; here [V | z] comes from ONE blocked forward substitution, N^2 (S+1) flop,
    // streamed behind the factorisation: block column K is solved as soon as the factorisation's chain has finished block K.
    const int C = m->C, D = m->D, nb = m->nb;
    const int64_t Npad = m->Npad;
    hipStream_t sv = m->st3 ? m->st3 : m->st;                       // test Gram + substitution (bulk CUs, lowest priority)
    for (auto& e : m->pred_ev) if (!e) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    HIP_TRY(hipEventRecord(m->pred_ev[0], m->st));                  // whatever the model's stream still holds comes first
    HIP_TRY(hipStreamWaitEvent(sv, m->pred_ev[0], 0));
    SortedX ss;
    if ((rc = sort_inputs(Xs, S, D, C, MOGP_TILE, ss))) return rc;
    const int64_t Spad = ss.Mpad, Srow = Spad + MOGP_TILE;          // one more tile row: its first row carries y^T through the same solve
    std::vector<GTile> pt;
    build_rect_tiles(ss.off, m->sx.off, C, pt);
    if ((rc = m->d_xs.ensure((size_t)D * Spad))) return rc;
    if ((rc = m->d_Ksf.ensure((size_t)Srow * Npad))) return rc;
    if ((rc = m->d_Vt.ensure((size_t)Srow * Npad))) return rc;
    if ((rc = m->d_mu.ensure(Spad))) return rc;
    if ((rc = m->d_var.ensure(Spad))) return rc;
    if ((rc = m->d_kdiag.ensure(Spad))) return rc;
    if ((rc = m->d_ptiles.ensure(pt.size()))) return rc;
    std::vector<double> kd(Spad, 0.0);
    const bool per_point = m->Wt > 2 + 3 * D;          // terms with an envelope: kss_diag holds one value per test point (caller order)
    for (int c = 0; c < C; ++c)
        for (int pos = ss.off[c]; pos < ss.off[c + 1]; ++pos) kd[pos] = per_point ? kss_diag[ss.perm[pos]] : kss_diag[c];
    HIP_TRY(hipMemcpyAsync(m->d_xs.p, ss.xs.data(), (size_t)D * Spad * sizeof(double), hipMemcpyHostToDevice, sv));
    HIP_TRY(hipMemcpyAsync(m->d_kdiag.p, kd.data(), Spad * sizeof(double), hipMemcpyHostToDevice, sv));
    HIP_TRY(hipMemcpyAsync(m->d_ptiles.p, pt.data(), pt.size() * sizeof(GTile), hipMemcpyHostToDevice, sv));
    // padded rows/columns of Ksf must be zero: rows >= S and columns >= N are never written by the Gram kernel
    HIP_TRY(hipMemsetAsync(m->d_Ksf.p, 0, (size_t)Srow * Npad * sizeof(double), sv));
    HIP_TRY(hipMemcpyAsync(m->d_Ksf.p + Spad * Npad, m->d_y.p, Npad * sizeof(double), hipMemcpyDeviceToDevice, sv));

    // K_sf = K(Xs, X)   (rows: test points, columns: training points; all C*C pairs, reference kernel.py:468-479 transposed)
    GramArgs ga{};
    ga.tiles = m->d_ptiles.p; ga.xr = m->d_xs.p; ga.ldxr = Spad; ga.xc = m->d_x.p; ga.ldxc = Npad; ga.nrows = S; ga.ncols = m->N;
    if ((rc = m->ph_sx.prepare(ss.off, m->sx.off, C, m->T, Spad, Npad, sv, ga.ph))) return rc;
    ga.table = m->d_table.p; ga.T = m->T; ga.D = D; ga.C = C; ga.W = m->Wt; ga.out = m->d_Ksf.p; ga.ldo = Npad;
    ga.noise = nullptr; ga.dvar = nullptr; ga.jitter_abs = 0.0; ga.mirror = 0;
    if ((rc = launch_gram(ga, (int)pt.size(), sv))) return rc;
    if (mean_w) {                                                    // mu = K_sf w, before the substitution consumes K_sf
        std::vector<double> hw(Npad, 0.0);
        for (int64_t pos = 0; pos < m->N; ++pos) hw[pos] = mean_w[m->sx.perm[pos]];
        if ((rc = m->d_z.ensure(Npad))) return rc;
        HIP_TRY(dev_upload(m->d_z.p, hw.data(), Npad * sizeof(double)));
        if ((rc = launch_gemv_rows(m->d_Ksf.p, Npad, Spad, Npad, m->d_z.p, m->d_mu.p, sv))) return rc;
    }

    // Round 4: factorisation AND substitution as ONE tile-dataflow schedule (flow.hip, the prediction's plan: panels, Schur updates, the solved block
    // X[:, K] = T[:, K] W_KK^T and the updates T[:, > K] -= X[:, K] L[> K, K]^T as tasks of the resident kernel, the chain kernels as producers) where the
    // dataflow kernel is available: the two sets of rank-512 launches on their streams got in each other's way like those of round 3's gradient schedule.
    // MOGP_FLOW_PREDICT=0: the stream form below.
    // From 48 tile rows on (CSM, S = N / 4, tools/r4_predict_sizes.py: N = 4096 3.9 vs 3.5 ms -- below, the chain sets the pace and the launch-per-step
    // chain of the stream form is the shorter one -- 8192 8.6 vs 9.5, 12288 21.1 vs 22.8, 16384 45.6 vs 47.8, 20480 86.3 vs 90.3).
    if ((rc = ensure_system(m))) return rc;                                                 // (flow_enabled looks at the system's tile count: a model's first call may be a prediction)
    const char* fpe = std::getenv("MOGP_FLOW_PREDICT");                                     // "0": never; "lo:hi": the range of tile rows (read per call: tests)
    int fp_lo = 48, fp_hi = 160;
    if (fpe && std::strchr(fpe, ':')) { fp_lo = std::atoi(fpe); fp_hi = std::atoi(std::strchr(fpe, ':') + 1); }
    else if (fpe) fp_hi = std::atoi(fpe);
    // mogp_model_set_accurate: the stream form with every panel of the factorisation and every solved block column refined once against L itself
    const bool acc = m->accurate;
    const bool as_flow = !acc && fp_hi > 0 && nb >= fp_lo && nb <= fp_hi && Srow / MOGP_TILE <= 4096 && flow_enabled(m, m->k) && !m->kinv_sparse;
    FlowRhs job{m->d_Ksf.p, m->d_Vt.p, (int)(Srow / MOGP_TILE), nullptr};
    if (as_flow) {
        HIP_TRY(hipEventRecord(m->pred_ev[1], sv));           // K_sf (and y^T in its last tile row) are in place
        job.ready = m->pred_ev[1];
        m->rhs_job = &job;
    }
    // the factorisation: enqueued on the model's streams, not waited for
    GramArgs gaK{};
    if (acc) { m->k.keep_L = true; m->k.refine_panels = true; }
    rc = factorize(m, noise_var, data_var, jitter, nullptr, nullptr, info, false, true, &gaK, true);
    if (acc) { m->k.keep_L = false; m->k.refine_panels = false; }
    m->rhs_job = nullptr;
    if (rc) return rc;
    const bool flowed = as_flow && m->k.flow_used;

    // X L^T = [K_sf ; y^T]  by block columns of 512 (right-looking):  X[:, K] = T[:, K] W_KK^T,  T[:, > K] -= X[:, K] L[> K, K]^T.
    // W_KK = L_KK^-1 of the 512 x 512 diagonal blocks comes from the tile inverses the factorisation leaves behind (wkk.hip).
    if (!flowed) {
        constexpr int OB = 4, KD = OB * MOGP_TILE;
        const int nouter = (nb + OB - 1) / OB, mt = (int)(Srow / MOGP_TILE);
        Spd& w = m->k;
        const bool streamed = MOGP_OUTER == OB && (int)w.sync_ev.size() >= 2 * nouter;     // spd_potrf's outer blocks are these blocks
        if (!streamed) {                                       // other blocking (MOGP_OUTER override): after the whole factorisation
            HIP_TRY(hipEventRecord(m->pred_ev[1], m->st));
            HIP_TRY(hipStreamWaitEvent(sv, m->pred_ev[1], 0));
        }
        if (w.Wd.n < (size_t)nouter * KD * KD) {             // tiles above the diagonal of a W_KK are never written and must be zero
            if ((rc = w.Wd.ensure((size_t)nouter * KD * KD))) return rc;
            HIP_TRY(hipMemsetAsync(w.Wd.p, 0, (size_t)nouter * KD * KD * sizeof(double), sv));
        }
        for (int kb = 0; kb < nouter; ++kb) {
            const int k0 = kb * OB, nk = std::min(OB, nb - k0), k1 = k0 + nk, rem = nb - k1;
            const int64_t c0 = (int64_t)k0 * MOGP_TILE;
            if (streamed) HIP_TRY(hipStreamWaitEvent(sv, w.sync_ev[2 * kb], 0));             // chain(kb): L[>= K, K] and the tile inverses of block K are final
            double* Wk = w.Wd.p + (int64_t)kb * KD * KD;
            if ((rc = launch_wkk(w.A.p + c0 * (Npad + 1), Npad, w.invd.p + (int64_t)k0 * MOGP_TILE * MOGP_TILE, nk, Wk, KD, sv))) return rc;
            GemmArgs g{};
            g.A = m->d_Ksf.p + c0; g.lda = Npad; g.a_kmajor = 0; g.B = Wk; g.ldb = KD; g.b_kmajor = 0;
            g.C = m->d_Vt.p + c0; g.ldc = Npad; g.alpha = 1.0; g.beta = 0.0;
            g.mode = GM_KHI_J; g.small = 1; g.mt = 2 * mt; g.nt = nk; g.K = nk * MOGP_TILE;        // 64 x 128 tiles: twice the workgroups of a launch that fills a quarter of the chip
            if ((rc = gemm_call(m, g, gemm_flops(g, nullptr), sv))) return rc;
            if (acc) {                 // X += (T - X L_KK^T) W_KK^T: the product with the explicit W_KK is only a first approximation of the solve (spd_potrf does the same to its panels)
                GemmArgs r1 = g;
                r1.A = m->d_Vt.p + c0; r1.B = w.A.p + c0 * (Npad + 1); r1.ldb = Npad; r1.C = m->d_Ksf.p + c0; r1.alpha = -1.0; r1.beta = 1.0;
                if ((rc = gemm_call(m, r1, gemm_flops(r1, nullptr), sv))) return rc;
                GemmArgs r2 = g;
                r2.A = m->d_Ksf.p + c0; r2.C = m->d_Vt.p + c0; r2.alpha = 1.0; r2.beta = 1.0;
                if ((rc = gemm_call(m, r2, gemm_flops(r2, nullptr), sv))) return rc;
            }
            if (rem > 0) {
                GemmArgs u{};
                u.A = m->d_Vt.p + c0; u.lda = Npad; u.a_kmajor = 0;
                u.B = w.A.p + (int64_t)k1 * MOGP_TILE * Npad + c0; u.ldb = Npad; u.b_kmajor = 0;
                u.C = m->d_Ksf.p + (int64_t)k1 * MOGP_TILE; u.ldc = Npad; u.alpha = -1.0; u.beta = 1.0;
                u.mode = GM_RECT; u.mt = mt; u.nt = rem; u.K = nk * MOGP_TILE;
                if ((rc = gemm_call(m, u, gemm_flops(u, nullptr), sv))) return rc;
            }
        }
        HIP_TRY(hipEventRecord(m->pred_ev[1], sv));
        HIP_TRY(hipStreamWaitEvent(m->st, m->pred_ev[1], 0));
    }
    // mu = V^T z: the rows of X against its last row (z^T)
    if (!mean_w && (rc = launch_gemv_rows(m->d_Vt.p, Npad, Spad, Npad, m->d_Vt.p + Spad * Npad, m->d_mu.p, m->st))) return rc;

    std::vector<double> hmu(Spad);
    HIP_TRY(hipMemcpyAsync(hmu.data(), m->d_mu.p, Spad * sizeof(double), hipMemcpyDeviceToHost, m->st));
    if (!full) {
        if ((rc = launch_row_sqnorm_sub(m->d_Vt.p, Npad, Spad, Npad, m->d_kdiag.p, m->d_var.p, m->st))) return rc;
        std::vector<double> hv(Spad);
        HIP_TRY(hipMemcpyAsync(hv.data(), m->d_var.p, Spad * sizeof(double), hipMemcpyDeviceToHost, m->st));
        HIP_TRY(hipStreamSynchronize(m->st));
        if ((rc = factorize_finish(m, gaK, nullptr, info))) {                           // the pivot report of the factorisation
            if (rc != MOGP_RETRY_NO_CHAIN) return rc;
            if ((rc = chain_fallback(m))) return rc;                                     // a hand-off of the dataflow schedule timed out: again, on streams
            return predict_core(m, noise_var, data_var, jitter, kss_diag, S, Xs, full, mu, var, info, mean_w);
        }
        for (int64_t pos = 0; pos < S; ++pos) { mu[ss.perm[pos]] = hmu[pos]; var[ss.perm[pos]] = hv[pos]; }
        return MOGP_OK;
    }
    // full covariance: K_ss - V^T V   (reference gpr/model.py:476-478)
    std::vector<GTile> st_tiles;
    std::vector<int> ps;
    build_sym_tiles(ss.off, C, st_tiles, ps);
    if ((rc = m->d_Kss.ensure((size_t)Spad * Spad))) return rc;
    if ((rc = m->d_ptiles.ensure(st_tiles.size()))) return rc;
    HIP_TRY(hipMemsetAsync(m->d_Kss.p, 0, (size_t)Spad * Spad * sizeof(double), m->st));
    HIP_TRY(hipMemcpyAsync(m->d_ptiles.p, st_tiles.data(), st_tiles.size() * sizeof(GTile), hipMemcpyHostToDevice, m->st));
    ga.tiles = m->d_ptiles.p; ga.xc = m->d_xs.p; ga.ldxc = Spad; ga.ncols = S; ga.out = m->d_Kss.p; ga.ldo = Spad; ga.mirror = 1;
    if ((rc = m->ph_ss.prepare(ss.off, ss.off, C, m->T, Spad, Spad, m->st, ga.ph))) return rc;
    if ((rc = launch_gram(ga, (int)st_tiles.size(), m->st))) return rc;
    GemmArgs c{};
    c.A = m->d_Vt.p; c.lda = Npad; c.a_kmajor = 0; c.B = m->d_Vt.p; c.ldb = Npad; c.b_kmajor = 0;
    c.C = m->d_Kss.p; c.ldc = Spad; c.alpha = -1.0; c.beta = 1.0;
    c.mode = GM_RECT; c.mt = c.nt = (int)(Spad / MOGP_TILE); c.K = (int)Npad; c.tasks = nullptr; c.ntasks = 0;
    if ((rc = gemm_call(m, c, gemm_flops(c, nullptr)))) return rc;
    std::vector<double> hc((size_t)Spad * Spad);
    HIP_TRY(hipMemcpyAsync(hc.data(), m->d_Kss.p, hc.size() * sizeof(double), hipMemcpyDeviceToHost, m->st));
    HIP_TRY(hipStreamSynchronize(m->st));
    if ((rc = factorize_finish(m, gaK, nullptr, info))) {
        if (rc != MOGP_RETRY_NO_CHAIN) return rc;
        if ((rc = chain_fallback(m))) return rc;
        return predict_core(m, noise_var, data_var, jitter, kss_diag, S, Xs, full, mu, var, info, mean_w);
    }
    for (int64_t a = 0; a < S; ++a) {
        mu[ss.perm[a]] = hmu[a];
        for (int64_t b = 0; b < S; ++b) var[ss.perm[a] * S + ss.perm[b]] = hc[(size_t)a * Spad + b];
    }
    return MOGP_OK;
}

int mogp_exact_predict(mogp_model* m, const double* noise_var, const double* data_var, double jitter,
                       const double* kss_diag, int64_t S, const double* Xs, int full,
                       double* mu, double* var, int64_t* info) {
    if (!m || !Xs || !mu || !var || !kss_diag || S <= 0) return fail(MOGP_EINVAL, "mogp_exact_predict: bad argument");
    return predict_core(m, noise_var, data_var, jitter, kss_diag, S, Xs, full, mu, var, info, nullptr);
}

// OpperArchambeau.predict_f (reference gpr/model.py:640-668):  mu = K_sf nu,  var = K_ss - K_sf (K + diag(1 / lambda^2))^-1 K_fs, no jitter
int mogp_oa_predict(mogp_model* m, const double* q_nu, const double* q_lambda, const double* kss_diag, int64_t S, const double* Xs, int full,
                    double* mu, double* var, int64_t* info) {
    if (!m || !q_nu || !q_lambda || !Xs || !mu || !var || !kss_diag || S <= 0) return fail(MOGP_EINVAL, "mogp_oa_predict: bad argument");
    std::vector<double> dv(m->N), zero(m->C, 0.0);
    for (int64_t i = 0; i < m->N; ++i) {
        if (!(q_lambda[i] > 0.0)) return fail(MOGP_EINVAL, "mogp_oa_predict: q_lambda must be positive");
        dv[i] = 1.0 / (q_lambda[i] * q_lambda[i]);
    }
    return predict_core(m, zero.data(), dv.data(), 0.0, kss_diag, S, Xs, full, mu, var, info, q_nu);
}

int mogp_gram(mogp_ctx* ctx, int C, int D, int T, const double* table, int64_t M1, const double* X1,
              int64_t M2, const double* X2, double* K_out) {
    return mogp_gram_ex(ctx, C, D, T, 2 + 3 * D, table, M1, X1, M2, X2, K_out);
}

int mogp_gram_ex(mogp_ctx* ctx, int C, int D, int T, int width, const double* table, int64_t M1, const double* X1,
                 int64_t M2, const double* X2, double* K_out) {
    if (!ctx || !table || !X1 || !K_out || M1 <= 0 || T <= 0 || C <= 0 || D <= 0 || D > MOGP_MAXD || (width != 2 + 3 * D && width != 2 + 5 * D))
        return fail(MOGP_EINVAL, "mogp_gram: bad argument");
    int rc;
    if ((rc = use_device(ctx))) return rc;
    const bool sym = (X2 == nullptr);
    SortedX s1, s2;
    if ((rc = sort_inputs(X1, M1, D, C, 1, s1))) return rc;
    if (!sym && (rc = sort_inputs(X2, M2, D, C, 1, s2))) return rc;
    const SortedX& sc = sym ? s1 : s2;
    const int64_t R = s1.M, Cc = sc.M;
    std::vector<GTile> tiles;
    std::vector<int> ps;
    if (sym) build_sym_tiles(s1.off, C, tiles, ps); else build_rect_tiles(s1.off, s2.off, C, tiles);
    const int W = width;
    DevBuf<double> dx1, dx2, dtab, dout;
    DevBuf<GTile> dt;
    PhaseWs ph;
    auto cleanup = [&]() { dx1.release(); dx2.release(); dtab.release(); dout.release(); dt.release(); ph.release(); };
#define G_TRY(x) do { int r__ = (x); if (r__) { cleanup(); return r__; } } while (0)
#define G_HIP(x) do { hipError_t e__ = (x); if (e__ != hipSuccess) { cleanup(); return hip_fail(e__, #x, __FILE__, __LINE__); } } while (0)
    G_TRY(dx1.ensure((size_t)D * s1.Mpad));
    G_TRY(dtab.ensure((size_t)C * C * T * W));
    G_TRY(dout.ensure((size_t)R * Cc));
    G_TRY(dt.ensure(std::max<size_t>(tiles.size(), 1)));
    G_HIP(dev_upload(dx1.p, s1.xs.data(), (size_t)D * s1.Mpad * sizeof(double)));
    if (!sym) {
        G_TRY(dx2.ensure((size_t)D * s2.Mpad));
        G_HIP(dev_upload(dx2.p, s2.xs.data(), (size_t)D * s2.Mpad * sizeof(double)));
    }
    G_HIP(dev_upload(dtab.p, table, (size_t)C * C * T * W * sizeof(double)));
    G_HIP(dev_upload(dt.p, tiles.data(), tiles.size() * sizeof(GTile)));
    GramArgs ga{};
    ga.tiles = dt.p; ga.xr = dx1.p; ga.ldxr = s1.Mpad; ga.xc = sym ? dx1.p : dx2.p; ga.ldxc = sc.Mpad; ga.nrows = R; ga.ncols = Cc;
    G_TRY(ph.prepare(s1.off, sc.off, C, T, s1.Mpad, sc.Mpad, nullptr, ga.ph));
    ga.table = dtab.p; ga.T = T; ga.D = D; ga.C = C; ga.W = W; ga.out = dout.p; ga.ldo = Cc;
    ga.noise = nullptr; ga.dvar = nullptr; ga.jitter_abs = 0.0; ga.mirror = 1;
    G_TRY(launch_gram(ga, (int)tiles.size(), nullptr));
    G_HIP(hipDeviceSynchronize());
    if (s1.identity && sc.identity) {
        G_HIP(hipMemcpy(K_out, dout.p, (size_t)R * Cc * sizeof(double), hipMemcpyDeviceToHost));
    } else {
        std::vector<double> h((size_t)R * Cc);
        G_HIP(hipMemcpy(h.data(), dout.p, h.size() * sizeof(double), hipMemcpyDeviceToHost));
        for (int64_t a = 0; a < R; ++a)
            for (int64_t b = 0; b < Cc; ++b) K_out[s1.perm[a] * Cc + sc.perm[b]] = h[(size_t)a * Cc + b];
    }
    cleanup();
#undef G_TRY
#undef G_HIP
    return MOGP_OK;
}

// A chain-kernel time-out inside a sharded evaluation cannot be repeated here (the other ranks are past their collectives).  One GPU per
// rank means nothing else competes for the reserved CUs, so it is not expected; the rank switches to the launch-per-step chain and reports.
static int sharded_rc(mogp_model* m, int rc) {
    if (rc != MOGP_RETRY_NO_CHAIN) return rc;
    if ((rc = chain_fallback(m))) return rc;
    return fail(MOGP_EHIP, "chain kernel: a hand-off timed out inside a sharded evaluation; this rank uses the launch-per-step chain from now on -- repeat the call on every rank");
}

// ---- sharded evaluation (one process per GPU; collectives are issued by the caller between these calls) --------------------
int mogp_shard_config(mogp_model* m, int rank, int nranks) {
    if (!m || nranks < 1 || rank < 0 || rank >= nranks) return fail(MOGP_EINVAL, "mogp_shard_config: bad argument");
    m->sh_rank = rank; m->sh_n = nranks;
    // MOGP_SHARD_OWNED: 1 (default) the owned-rows form for every group of more than one rank, 0 the replicated-matrix form of rounds 1-5,
    // 2 the owned-rows form for a one-rank group as well (its code path on one GPU: tests)
    static const int owned_mode = []() { const char* e = std::getenv("MOGP_SHARD_OWNED"); return e ? std::atoi(e) : 1; }();
    m->sh_owned = (owned_mode >= 1 && nranks > 1) || owned_mode >= 2;
    { int r__ = use_device(m->ctx); if (r__) return r__; r__ = ensure_system(m); if (r__) return r__; }
    if (nranks > 1 && (m->own_rank != rank || m->own_n != nranks)) {
        // each rank generates exactly the Gram / moment tiles it owns: a 64-row tile is kept if one of the (at most two)
        // 128-row tile rows it touches belongs to this rank; nothing else of the work matrix is ever read on this rank (sweep.hip)
        int rc;
        if ((rc = use_device(m->ctx))) return rc;
        m->tiles_own.clear();
        m->pair_start_own.assign(1, 0);
        for (size_t p = 0; p + 1 < m->pair_start.size(); ++p) {
            for (int t = m->pair_start[p]; t < m->pair_start[p + 1]; ++t) {
                const GTile& g = m->tiles[t];
                const int a = g.r0 / MOGP_TILE, b = (g.r0 + g.nr - 1) / MOGP_TILE;
                if (a % nranks == rank || b % nranks == rank) m->tiles_own.push_back(g);
            }
            m->pair_start_own.push_back((int)m->tiles_own.size());
        }
        if ((rc = m->d_tiles_own.ensure(std::max<size_t>(m->tiles_own.size(), 1)))) return rc;
        if ((rc = m->d_pair_start_own.ensure(m->pair_start_own.size()))) return rc;
        HIP_TRY(dev_upload(m->d_tiles_own.p, m->tiles_own.data(), m->tiles_own.size() * sizeof(GTile)));
        if ((rc = m->strip_own.build(m->tiles_own))) return rc;
        HIP_TRY(dev_upload(m->d_pair_start_own.p, m->pair_start_own.data(), m->pair_start_own.size() * sizeof(int)));
        m->own_rank = rank; m->own_n = nranks;
    }
    if (m->k.owned_rows && (m->backed_rank != rank || m->backed_n != nranks)) {
        // physical memory under this rank's part of the work matrix: its tile rows (i % nranks == rank) and, where a channel does not start on a
        // 128-row boundary, the rows of a neighbouring tile row that one of its 64-row Gram tiles reaches into
        int rc;
        const size_t row_bytes = (size_t)m->Npad * sizeof(double);
        for (int i = rank; i < m->nb; i += nranks)
            if ((rc = m->k.Arows.back((size_t)i * MOGP_TILE * row_bytes, (size_t)MOGP_TILE * row_bytes))) return rc;
        if (nranks > 1)
            for (const GTile& g : m->tiles_own)
                if ((rc = m->k.Arows.back((size_t)g.r0 * row_bytes, (size_t)g.nr * row_bytes))) return rc;
        m->backed_rank = rank; m->backed_n = nranks;
    }
    return MOGP_OK;
}

int mogp_model_work_bytes(mogp_model* m, int64_t* backed, int64_t* whole) {
    if (!m || !backed || !whole) return fail(MOGP_EINVAL, "mogp_model_work_bytes: bad argument");
    const int64_t one = (int64_t)m->k.Npad * m->k.Npad * (int64_t)sizeof(double);
    *whole = one;
    *backed = m->k.Npad == 0 ? 0 : (m->k.owned_rows || m->k.Arows.base ? (int64_t)m->k.Arows.backed_bytes() : one);
    return MOGP_OK;
}

int mogp_shard_begin(mogp_model* m, const double* noise_var, const double* data_var, double jitter, double* jitter_abs, int* nblocks) {
    if (!m || !nblocks) return fail(MOGP_EINVAL, "mogp_shard_begin: bad argument");
    int rc;
    if ((rc = use_device(m->ctx))) return rc;
    if ((rc = sweep_eval_begin(m, noise_var, data_var, jitter))) return rc;
    if ((rc = sweep_prepare(m, m->k))) return rc;
    if (jitter_abs) *jitter_abs = m->sh_jabs;
    *nblocks = sweep_nblocks(m->k);
    return MOGP_OK;
}

int mogp_shard_pack(mogp_model* m, int kb, void** send, void** recv, int64_t* count) {
    if (!m || !send || !recv || !count) return fail(MOGP_EINVAL, "mogp_shard_pack: bad argument");
    int rc;
    if ((rc = use_device(m->ctx))) return rc;
    double *s = nullptr, *r = nullptr;
    if ((rc = shard_pack(m, m->k, kb, &s, &r, count))) return rc;
    HIP_TRY(hipStreamSynchronize(m->st));           // the caller's collective runs outside this stream; the bulk stream keeps running
    *send = s; *recv = r;
    return MOGP_OK;
}

int mogp_shard_unpack(mogp_model* m, int kb) {
    if (!m) return fail(MOGP_EINVAL, "mogp_shard_unpack: bad argument");
    int rc;
    if ((rc = use_device(m->ctx))) return rc;
    return shard_unpack(m, m->k, kb);
}

int mogp_shard_block(mogp_model* m, int kb) {
    if (!m) return fail(MOGP_EINVAL, "mogp_shard_block: bad argument");
    int rc;
    if ((rc = use_device(m->ctx))) return rc;
    return sweep_block(m, m->k, kb);
}

int mogp_shard_alpha(mogp_model* m, void** vec, int64_t* count) {
    if (!m || !vec || !count) return fail(MOGP_EINVAL, "mogp_shard_alpha: bad argument");
    int rc;
    if ((rc = use_device(m->ctx))) return rc;
    if ((rc = sweep_finish(m, m->k))) return rc;
    if ((rc = sweep_eval_alpha(m))) return rc;
    HIP_TRY(hipStreamSynchronize(m->st));
    *vec = m->d_alpha.p; *count = m->Npad;
    return MOGP_OK;
}

int mogp_shard_finish(mogp_model* m, double* lml, double* moments, double* diagG, int64_t* info) {
    if (!m || !lml || !moments || !diagG) return fail(MOGP_EINVAL, "mogp_shard_finish: bad argument");
    int rc;
    if ((rc = use_device(m->ctx))) return rc;
    if (info) *info = 0;
    if ((rc = sweep_eval_scalars(m, lml, info))) return sharded_rc(m, rc);
    if ((rc = mark(m, 5))) return rc;
    if ((rc = moment_pass(m, m->k.A.p, -1.0, moments, diagG))) return rc;
    m->have_Kinv = true; m->kinv_in_A = true;
    return MOGP_OK;
}

// ---- the sharded evaluation as ONE call: the collectives are issued here, on the model's critical stream (comm.hip) ---------------------
static int sharded_inverse(mogp_model* m, const double* noise_var, const double* data_var, double jitter, double* jitter_abs) {
    mogp_comm& c = m->ctx->comm;
    int rc;
    if ((rc = mogp_shard_config(m, c.rank, c.n))) return rc;
    if ((rc = sweep_eval_begin(m, noise_var, data_var, jitter))) return rc;
    if ((rc = sweep_prepare(m, m->k))) return rc;
    if (jitter_abs) *jitter_abs = m->sh_jabs;
    const int nblocks = sweep_nblocks(m->k);
    m->sh_prof_blocks = 0;
    // Round 5: the exchange of a pivot block in TWO messages.  The serial part (Schur block inversion, 0.3 ms, repeated on every rank) needs the pivot
    // block's own tile rows only: 4 tiles of 128 x 512, 2 MB.  The rest of the panel -- the column part below the block and the row part left of it,
    // up to 134 MB at configs[2] -- is needed by the panel products behind it.  So: small message on the critical stream, large message on a
    // communication stream of its own (the context's third stream, idle in this schedule) UNDERNEATH the serial part; the critical stream waits
    // for it only where the panels start.  Both are collectives of the same communicator issued in the same order on every rank.
    // MOGP_SHARD_SPLIT=0: one message on the critical stream, as in rounds 1-4.  (A group of ONE rank runs the same schedule -- its messages are
    // copies -- so that the one-rank time measures what the schedule costs a rank, not a schedule of its own.)
    { const char* e = std::getenv("MOGP_SHARD_SPLIT"); m->sh_split = m->st3 && !(e && std::atoi(e) == 0); }
    { const char* e = std::getenv("MOGP_SHARD_FACTOR_ONCE"); m->sh_factor_once = c.n > 1 && e && std::atoi(e) != 0; }
    const int PEV = 10;                                  // timing events per pivot block
    if (m->profiling) {
        while ((int)m->sh_prof.size() < PEV * nblocks) { hipEvent_t e; HIP_TRY(hipEventCreate(&e)); m->sh_prof.push_back(e); }
        m->sh_prof_blocks = nblocks;
    }
    while ((int)m->sh_ev.size() < 2 * nblocks) { hipEvent_t e; HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming)); m->sh_ev.push_back(e); }
    hipStream_t qc = m->st3;
    for (int kb = 0; kb < nblocks; ++kb) {
        int64_t count = 0;
        hipEvent_t* pe = m->profiling ? m->sh_prof.data() + PEV * kb : nullptr;
        if (pe) HIP_TRY(hipEventRecord(pe[0], m->st));
        if (!m->sh_split) {
            double *send = nullptr, *recv = nullptr;
            if ((rc = shard_pack(m, m->k, kb, &send, &recv, &count))) return rc;
            if ((rc = comm_allgather(m->ctx, send, recv, count, m->st))) return rc;       // stream ordered: no host round trip with RCCL
            if ((rc = shard_unpack(m, m->k, kb))) return rc;
            if (pe) HIP_TRY(hipEventRecord(pe[1], m->st));
            if ((rc = sweep_block(m, m->k, kb, pe ? pe + 2 : nullptr))) return rc;
            continue;
        }
        hipEvent_t packed = m->sh_ev[2 * kb], rest_in = m->sh_ev[2 * kb + 1];
        int64_t count2 = 0;
        if ((rc = shard_pack_part(m, m->k, kb, 1, m->sh_send1, m->sh_recv1, &count, m->st))) return rc;
        if ((rc = shard_pack_part(m, m->k, kb, 2, m->sh_send, m->sh_recv, &count2, m->st))) return rc;     // (both read what the previous block's next-columns update left: this stream)
        HIP_TRY(hipEventRecord(packed, m->st));
        if ((rc = comm_allgather(m->ctx, m->sh_send1.p, m->sh_recv1.p, count, m->st))) return rc;
        if ((rc = shard_unpack_part(m, m->k, kb, 1, m->sh_recv1, m->st))) return rc;
        if (pe) HIP_TRY(hipEventRecord(pe[1], m->st));
        HIP_TRY(hipStreamWaitEvent(qc, packed, 0));
        if (pe) HIP_TRY(hipEventRecord(pe[6], qc));
        if ((rc = comm_allgather(m->ctx, m->sh_send.p, m->sh_recv.p, count2, qc))) return rc;
        if ((rc = shard_unpack_part(m, m->k, kb, 2, m->sh_recv, qc))) return rc;            // other ranks' rows only: nothing this rank's streams touch
        if (pe) HIP_TRY(hipEventRecord(pe[7], qc));
        HIP_TRY(hipEventRecord(rest_in, qc));
        if ((rc = sweep_block(m, m->k, kb, pe ? pe + 2 : nullptr, rest_in, pe ? pe + 8 : nullptr))) return rc;
    }
    if ((rc = sweep_finish(m, m->k))) return rc;
    if ((rc = sweep_eval_alpha(m))) return rc;                                         // owned-row partial sums of alpha
    return comm_allreduce(m->ctx, m->d_alpha.p, m->Npad, m->st);
}

int mogp_exact_eval_sharded(mogp_model* m, const double* noise_var, const double* data_var, double jitter,
                            double* lml, double* moments, double* diagG, double* trG, double* jitter_abs, int64_t* info) {
    if (!m || !lml || !moments || !diagG || !trG) return fail(MOGP_EINVAL, "mogp_exact_eval_sharded: bad argument");
    int rc;
    if ((rc = use_device(m->ctx))) return rc;
    if (info) *info = 0;
    const int C = m->C, W = m->Wt, T = m->T, P = C * (C + 1) / 2;
    m->pivot_min = m->pivot_max = 0.0;
    if ((rc = sharded_inverse(m, noise_var, data_var, jitter, jitter_abs))) return rc;
    if ((rc = mark(m, 5))) return rc;
    if ((rc = moment_pass_device(m, m->k.A.p, -1.0))) return rc;                       // owned rows only
    if ((rc = comm_allreduce(m->ctx, m->d_moments.p, (int64_t)P * T * W, m->st))) return rc;
    if ((rc = comm_allreduce(m->ctx, m->d_diagG.p, C, m->st))) return rc;
    HIP_TRY(hipMemcpyAsync(moments, m->d_moments.p, (size_t)P * T * W * sizeof(double), hipMemcpyDeviceToHost, m->st));
    HIP_TRY(hipMemcpyAsync(diagG, m->d_diagG.p, C * sizeof(double), hipMemcpyDeviceToHost, m->st));
    if ((rc = sweep_eval_scalars(m, lml, info))) return sharded_rc(m, rc);             // syncs the stream
    if (m->sh_prof_blocks > 0) {
        for (hipStream_t q : {m->st2, m->st2u}) if (q) HIP_TRY(hipStreamSynchronize(q));
        if (m->st3) HIP_TRY(hipStreamSynchronize(m->st3));
        double acc6[6] = {0, 0, 0, 0, 0, 0};
        for (int kb = 0; kb < m->sh_prof_blocks; ++kb) {
            hipEvent_t* pe = m->sh_prof.data() + 10 * kb;
            // exchange on the critical stream | serial part (inversion + panels) | next-block columns | bulk | exchange on the communication stream | the critical stream's wait for it
            const int a_[6] = {0, 1, 2, 4, 6, 8}, b_[6] = {1, 2, 3, 5, 7, 9};
            for (int i = 0; i < 6; ++i) {
                if (i >= 4 && !m->sh_split) continue;
                float t = 0.f;
                if (hipEventElapsedTime(&t, pe[a_[i]], pe[b_[i]]) == hipSuccess) acc6[i] += t;
            }
        }
        for (int i = 0; i < 6; ++i) m->sh_ms[i] = acc6[i];
    }
    double tr = 0.0;
    for (int c = 0; c < C; ++c) tr += diagG[c];
    *trG = tr;
    m->have_Kinv = true; m->kinv_in_A = true;
    collect_timing(m, 6);
    return MOGP_OK;
}

// part[s] = sum over this rank's tile rows j (T[s][j*128 + k], the rows' share of K_s. Kj^-1) and k of Ksf[s][row(j)*128 + k] * T[s][j*128 + k]
// (one wave per test point; rows: the tile rows i = rank, rank + P, ...)
__global__ __launch_bounds__(256) void k_owned_quadform(const double* __restrict__ Ksf, int64_t ldk, const double* __restrict__ T, int64_t ldt, int64_t S,
                                                        int nown, int P, int rank, double* __restrict__ part) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t s = (int64_t)blockIdx.x * 4 + wave;
    if (s >= S) return;
    double acc = 0.0;
    for (int j = 0; j < nown; ++j) {
        const double* kr = Ksf + s * ldk + (int64_t)(rank + j * P) * MOGP_TILE;
        const double* tr = T + s * ldt + (int64_t)j * MOGP_TILE;
        acc = fma(kr[lane], tr[lane], acc);
        acc = fma(kr[lane + 64], tr[lane + 64], acc);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if (lane == 0) part[s] = acc;
}
__global__ void k_var_finish(const double* __restrict__ kdiag, const double* __restrict__ part, int64_t S, double* __restrict__ var) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < S) var[i] = kdiag[i] + part[i];               // (the work matrix holds MINUS Kj^-1)
}

int mogp_exact_predict_sharded(mogp_model* m, const double* noise_var, const double* data_var, double jitter,
                               const double* kss_diag, int64_t S, const double* Xs, double* mu, double* var, int64_t* info) {
    if (!m || !Xs || !mu || !var || !kss_diag || S <= 0) return fail(MOGP_EINVAL, "mogp_exact_predict_sharded: bad argument");
    int rc;
    if ((rc = use_device(m->ctx))) return rc;
    if (info) *info = 0;
    mogp_comm& cm = m->ctx->comm;
    const int P = cm.n, rank = cm.rank, C = m->C, D = m->D, nb = m->nb;
    const int64_t Npad = m->Npad;
    // 1. the inversion, sharded exactly like the gradient evaluation: owned tile rows of -Kj^-1 (lower tiles, whole diagonal tiles) in k.A, alpha complete on every rank
    if ((rc = sharded_inverse(m, noise_var, data_var, jitter, nullptr))) return rc;
    double lml = 0.0;
    if ((rc = sweep_eval_scalars(m, &lml, info))) return sharded_rc(m, rc);            // failure report (not positive definite)
    // 2. Round 6: the predictive variance FROM THE OWNED ROWS, no all-gather of Kj^-1 (N^2 doubles, and the whole inverse on every rank, in rounds 3 - 5).
    //    k_ss - K_s. Kj^-1 K_.s is a quadratic form: with the rows a of Kj^-1 dealt out to the ranks,
    //        sum_ab K_sa Kinv_ab K_sb = sum over ranks, over their tile rows i, of  sum_{a in i} K_sa ( 2 sum_{b left of tile i} Kinv_ab K_sb + sum_{b in tile i} Kinv_ab K_sb )
    //    -- the strictly lower tiles count twice, the diagonal tile (held whole) once.  Every rank: the test Gram K_sf for ALL test points, one task-list GEMM
    //    T[:, tile row] = K_sf[:, left of it] A[row, left of it]^T (x 2) behind one for the diagonal tiles, a row-wise dot, and ONE all-reduce of S doubles.
    SortedX ss;
    if ((rc = sort_inputs(Xs, S, D, C, MOGP_TILE, ss))) return rc;
    const int64_t Spad = ss.Mpad;
    const int st = (int)(Spad / MOGP_TILE);
    const int nown = rank < nb ? (nb - rank + P - 1) / P : 0;                          // tile rows rank, rank + P, ...
    const int64_t ldt = (int64_t)std::max(nown, 1) * MOGP_TILE;
    std::vector<GTile> pt;
    build_rect_tiles(ss.off, m->sx.off, C, pt);
    if ((rc = m->d_xs.ensure((size_t)D * Spad))) return rc;
    if ((rc = m->d_Ksf.ensure((size_t)Spad * Npad))) return rc;
    if ((rc = m->d_Vt.ensure((size_t)Spad * ldt))) return rc;
    if ((rc = m->d_mu.ensure(Spad))) return rc;
    if ((rc = m->d_var.ensure(2 * Spad))) return rc;                                   // [variance | this rank's share of the quadratic form]
    if ((rc = m->d_kdiag.ensure(Spad))) return rc;
    if ((rc = m->d_ptiles.ensure(std::max<size_t>(pt.size(), 1)))) return rc;
    std::vector<double> kd(Spad, 0.0);
    const bool per_point = m->Wt > 2 + 3 * D;
    for (int c = 0; c < C; ++c)
        for (int pos = ss.off[c]; pos < ss.off[c + 1]; ++pos) kd[pos] = per_point ? kss_diag[ss.perm[pos]] : kss_diag[c];
    // the two task lists: [diagonal tiles | strictly lower parts], longest k range first within each
    std::vector<GemmTask> tasks;
    for (int j = 0; j < nown; ++j)
        for (int t = 0; t < st; ++t) {
            const int64_t i = rank + (int64_t)j * P;
            GemmTask g{};
            g.a_off = (int64_t)t * MOGP_TILE * Npad + i * MOGP_TILE; g.b_off = i * MOGP_TILE * Npad + i * MOGP_TILE;
            g.c_off = (int64_t)t * MOGP_TILE * ldt + (int64_t)j * MOGP_TILE; g.kt = MOGP_TILE / 16; g.pad = 0;
            tasks.push_back(g);
        }
    const size_t ndiag = tasks.size();
    for (int j = nown - 1; j >= 0; --j)
        for (int t = 0; t < st; ++t) {
            const int64_t i = rank + (int64_t)j * P;
            if (i == 0) continue;
            GemmTask g{};
            g.a_off = (int64_t)t * MOGP_TILE * Npad; g.b_off = i * MOGP_TILE * Npad;
            g.c_off = (int64_t)t * MOGP_TILE * ldt + (int64_t)j * MOGP_TILE; g.kt = (int)(i * MOGP_TILE / 16); g.pad = 0;
            tasks.push_back(g);
        }
    if ((rc = m->d_pred_tasks.ensure(std::max<size_t>(tasks.size(), 1)))) return rc;
    HIP_TRY(hipMemcpyAsync(m->d_xs.p, ss.xs.data(), (size_t)D * Spad * sizeof(double), hipMemcpyHostToDevice, m->st));
    HIP_TRY(hipMemcpyAsync(m->d_kdiag.p, kd.data(), Spad * sizeof(double), hipMemcpyHostToDevice, m->st));
    HIP_TRY(hipMemcpyAsync(m->d_ptiles.p, pt.data(), pt.size() * sizeof(GTile), hipMemcpyHostToDevice, m->st));
    if (!tasks.empty()) HIP_TRY(hipMemcpyAsync(m->d_pred_tasks.p, tasks.data(), tasks.size() * sizeof(GemmTask), hipMemcpyHostToDevice, m->st));
    HIP_TRY(hipMemsetAsync(m->d_Ksf.p, 0, (size_t)Spad * Npad * sizeof(double), m->st));           // padded rows / columns stay zero
    HIP_TRY(hipMemsetAsync(m->d_var.p, 0, 2 * Spad * sizeof(double), m->st));
    GramArgs ga{};
    ga.tiles = m->d_ptiles.p; ga.xr = m->d_xs.p; ga.ldxr = Spad; ga.xc = m->d_x.p; ga.ldxc = Npad; ga.nrows = S; ga.ncols = m->N;
    if ((rc = m->ph_sx.prepare(ss.off, m->sx.off, C, m->T, Spad, Npad, m->st, ga.ph))) return rc;
    ga.table = m->d_table.p; ga.T = m->T; ga.D = D; ga.C = C; ga.W = m->Wt; ga.out = m->d_Ksf.p; ga.ldo = Npad;
    ga.noise = nullptr; ga.dvar = nullptr; ga.jitter_abs = 0.0; ga.mirror = 0;
    if ((rc = launch_gram(ga, (int)pt.size(), m->st))) return rc;
    if ((rc = launch_gemv_rows(m->d_Ksf.p, Npad, Spad, Npad, m->d_alpha.p, m->d_mu.p, m->st))) return rc;            // mu = K_sf alpha (alpha is complete on every rank)
    if (nown > 0) {
        GemmArgs g{};
        g.A = m->d_Ksf.p; g.lda = Npad; g.a_kmajor = 0; g.B = m->k.A.p; g.ldb = Npad; g.b_kmajor = 0;
        g.C = m->d_Vt.p; g.ldc = ldt; g.mode = GM_TASKS; g.mt = g.nt = 0; g.K = 0;
        g.alpha = 1.0; g.beta = 0.0; g.tasks = m->d_pred_tasks.p; g.ntasks = (int)ndiag;
        if ((rc = gemm_call(m, g, 2.0 * MOGP_TILE * MOGP_TILE * MOGP_TILE * (double)ndiag))) return rc;
        if (tasks.size() > ndiag) {
            double fl = 0.0;
            for (size_t k = ndiag; k < tasks.size(); ++k) fl += 2.0 * MOGP_TILE * MOGP_TILE * 16.0 * tasks[k].kt;
            g.alpha = 2.0; g.beta = 1.0; g.tasks = m->d_pred_tasks.p + ndiag; g.ntasks = (int)(tasks.size() - ndiag);
            if ((rc = gemm_call(m, g, fl))) return rc;
        }
        hipLaunchKernelGGL(k_owned_quadform, dim3((unsigned)((Spad + 3) / 4)), dim3(256), 0, m->st, m->d_Ksf.p, Npad, m->d_Vt.p, ldt, Spad, nown, P, rank,
                           m->d_var.p + Spad);
        HIP_TRY(hipGetLastError());
    }
    // 3. the ranks' shares of the quadratic form: one sum of S doubles
    if ((rc = comm_allreduce(m->ctx, m->d_var.p + Spad, Spad, m->st))) return rc;
    hipLaunchKernelGGL(k_var_finish, dim3((unsigned)((Spad + 255) / 256)), dim3(256), 0, m->st, m->d_kdiag.p, m->d_var.p + Spad, Spad, m->d_var.p);
    HIP_TRY(hipGetLastError());
    std::vector<double> hmu(Spad), hv(Spad);
    HIP_TRY(hipMemcpyAsync(hmu.data(), m->d_mu.p, Spad * sizeof(double), hipMemcpyDeviceToHost, m->st));
    HIP_TRY(hipMemcpyAsync(hv.data(), m->d_var.p, Spad * sizeof(double), hipMemcpyDeviceToHost, m->st));
    HIP_TRY(hipStreamSynchronize(m->st));
    for (int64_t pos = 0; pos < S; ++pos) { mu[ss.perm[pos]] = hmu[pos]; var[ss.perm[pos]] = hv[pos]; }
    m->have_Kinv = false; m->have_W = false;
    return MOGP_OK;
}

int mogp_model_set_accurate(mogp_model* m, int on) {
    if (!m) return fail(MOGP_EINVAL, "mogp_model_set_accurate: model is null");
    m->accurate = on != 0;
    return MOGP_OK;
}

int mogp_model_pivot_range(mogp_model* m, double* lmin, double* lmax) {
    if (!m || !lmin || !lmax) return fail(MOGP_EINVAL, "mogp_model_pivot_range: bad argument");
    *lmin = m->pivot_min; *lmax = m->pivot_max;
    return MOGP_OK;
}

int mogp_model_inverse_fraction(mogp_model* m, double* fraction) {
    if (!m || !fraction) return fail(MOGP_EINVAL, "mogp_model_inverse_fraction: bad argument");
    *fraction = m->kinv_sparse ? m->kinv_fraction : 1.0;
    return MOGP_OK;
}

int mogp_model_schedule(mogp_model* m, int* flags) {
    if (!m || !flags) return fail(MOGP_EINVAL, "mogp_model_schedule: null argument");
    *flags = (m->k.flow_used ? MOGP_SCHED_DATAFLOW : 0) | (chain_enabled(m) ? MOGP_SCHED_CHAIN_KERNEL : 0) |
             (m->no_flow ? MOGP_SCHED_DATAFLOW_FELL_BACK : 0) | (m->no_chain ? MOGP_SCHED_CHAIN_FELL_BACK : 0) |
             (std::min(m->flow_timeouts, 0xffff) << MOGP_SCHED_TIMEOUTS_SHIFT);
    return MOGP_OK;
}

int mogp_model_flow_replay(mogp_model* m, int on) {
    if (!m) return fail(MOGP_EINVAL, "mogp_model_flow_replay: null model");
    if (on && !(m->k.Wm.p && m->have_Kinv)) return fail(MOGP_EINVAL, "mogp_model_flow_replay: needs a completed gradient evaluation on this model first (its W_KK blocks are the replay's input)");
    m->replay_flow = on != 0;
    return MOGP_OK;
}

int mogp_model_flow_diag(mogp_model* m, unsigned* out8) {
    if (!m || !out8) return fail(MOGP_EINVAL, "mogp_model_flow_diag: null argument");
    int rc;
    if ((rc = use_device(m->ctx))) return rc;
    std::memset(out8, 0, 8 * sizeof(unsigned));
    if (!m->k.flow_diag.p) return MOGP_OK;
    HIP_TRY(hipStreamSynchronize(m->st));
    HIP_TRY(hipMemcpy(out8, m->k.flow_diag.p, FLOW_DIAG_WORDS * sizeof(unsigned), hipMemcpyDeviceToHost));
    return MOGP_OK;
}

int mogp_shard_stage_ms(mogp_model* m, double* ms) {
    if (!m || !ms) return fail(MOGP_EINVAL, "mogp_shard_stage_ms: bad argument");
    for (int i = 0; i < 6; ++i) ms[i] = m->sh_ms[i];
    return MOGP_OK;
}

int mogp_comm_selftest(mogp_ctx* ctx, int* ranks_seen, int* rank_sum) {
    if (!ctx || !ranks_seen || !rank_sum) return fail(MOGP_EINVAL, "mogp_comm_selftest: bad argument");
    int rc;
    if ((rc = use_device(ctx))) return rc;
    if ((rc = ctx_streams(ctx))) return rc;
    double h[2] = {1.0, (double)(ctx->comm.rank + 1)};
    double* d = nullptr;
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&d), sizeof(h)));
    hipError_t e = hipMemcpyAsync(d, h, sizeof(h), hipMemcpyHostToDevice, ctx->st);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->st);
    if (e == hipSuccess && (rc = comm_allreduce(ctx, d, 2, ctx->st)) == 0) {
        e = hipMemcpyAsync(h, d, sizeof(h), hipMemcpyDeviceToHost, ctx->st);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->st);
    }
    hipError_t e2 = hipFree(d); (void)e2;
    if (rc) return rc;
    HIP_TRY(e);
    *ranks_seen = (int)(h[0] + 0.5);
    *rank_sum = (int)(h[1] + 0.5);
    return MOGP_OK;
}

int mogp_dev_copy(void* dst, const void* src, int64_t bytes, int to_device) {
    if (!dst || !src || bytes < 0) return fail(MOGP_EINVAL, "mogp_dev_copy: bad argument");
    HIP_TRY(hipMemcpy(dst, src, (size_t)bytes, to_device ? hipMemcpyHostToDevice : hipMemcpyDeviceToHost));
    if (to_device) HIP_TRY(hipStreamSynchronize(nullptr));          // see dev_upload
    return MOGP_OK;
}

int mogp_set_profiling(mogp_model* m, int on) {
    if (!m) return fail(MOGP_EINVAL, "mogp_set_profiling: model is null");
    m->profiling = on != 0;
    return MOGP_OK;
}

int mogp_stage_ms(mogp_model* m, double* ms, int64_t* gemm_launches, double* gemm_flops) {
    if (!m || !ms) return fail(MOGP_EINVAL, "mogp_stage_ms: bad argument");
    for (int i = 0; i < MOGP_ST_COUNT; ++i) ms[i] = m->ms[i];
    if (gemm_launches) *gemm_launches = m->gemm_launches;
    if (gemm_flops) *gemm_flops = m->gemm_flops;
    return MOGP_OK;
}

int mogp_model_fetch(mogp_model* m, int which, double* out) {
    if (!m || !out) return fail(MOGP_EINVAL, "mogp_model_fetch: bad argument");
    int rc;
    if ((rc = use_device(m->ctx))) return rc;
    const int64_t N = m->N, Npad = m->Npad;
    if (which == 2) {
        if (!m->have_W && !m->have_Kinv) return fail(MOGP_EINVAL, "mogp_model_fetch: no evaluation has completed yet");
        std::vector<double> h(Npad);
        HIP_TRY(hipMemcpy(h.data(), m->d_alpha.p, Npad * sizeof(double), hipMemcpyDeviceToHost));
        for (int64_t pos = 0; pos < N; ++pos) out[m->sx.perm[pos]] = h[pos];
        return MOGP_OK;
    }
    if (which == 0 && !m->have_W) return fail(MOGP_EINVAL, "mogp_model_fetch: no evaluation has completed yet");
    if (which == 1 && !m->have_Kinv) return fail(MOGP_EINVAL, "mogp_model_fetch: Kj^-1 needs an evaluation with MOGP_EVAL_GRAD");
    if (which != 0 && which != 1) return fail(MOGP_EINVAL, "mogp_model_fetch: which must be 0, 1 or 2");
    if (m->k.owned_rows) return fail(MOGP_EINVAL, "mogp_model_fetch: this rank of a sharded evaluation holds only its own tile rows of the matrix");
    if (which == 1 && m->kinv_sparse && !m->kinv_in_A) {
        // the evaluation formed only the tiles of Kj^-1 its gradient reads (kinv_plan): form all of them now, W^T W from the W it left
        m->kinv_sparse = false;
        GemmArgs g{};
        const double* Wp = m->w_in_Wm ? m->k.Wm.p : m->k.A.p;
        g.A = Wp; g.lda = Npad; g.a_kmajor = 1; g.B = Wp; g.ldb = Npad; g.b_kmajor = 1;
        g.C = m->k.B.p; g.ldc = Npad; g.alpha = 1.0; g.beta = 0.0;
        g.mode = GM_LAUUM; g.mt = g.nt = m->nb; g.K = (int)Npad;
        if ((rc = gemm_call(m, g, gemm_flops(g, nullptr)))) return rc;
        HIP_TRY(hipStreamSynchronize(m->st));
    }
    std::vector<double> h((size_t)Npad * Npad);
    const bool neg = (which == 1 && m->kinv_in_A);
    const double* src = which == 0 ? (m->w_in_Wm ? m->k.Wm.p : m->k.A.p) : (neg ? m->k.A.p : m->k.B.p);
    HIP_TRY(hipMemcpy(h.data(), src, h.size() * sizeof(double), hipMemcpyDeviceToHost));
    if (neg) for (auto& v : h) v = -v;
    for (int64_t a = 0; a < N; ++a)
        for (int64_t b = 0; b < N; ++b) {
            double v;
            if (which == 0) v = (b <= a) ? h[(size_t)a * Npad + b] : 0.0;                               // W = L^-1 (sorted order)
            else v = (b <= a) ? h[(size_t)a * Npad + b] : h[(size_t)b * Npad + a];                        // symmetric Kj^-1
            if (which == 0) out[a * N + b] = v;
            else out[m->sx.perm[a] * N + m->sx.perm[b]] = v;
        }
    return MOGP_OK;
}

}  // extern "C"
