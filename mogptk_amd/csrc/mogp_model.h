// mogp_model.h -- private definitions shared by the translation units of libmogp_hip.so
#pragma once
#include "../../include/mogp_hip.h"
#include "mogp_internal.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <string>
#include <map>
#include <vector>

namespace mogp {

// hipMemset returns before the fill has run (like cudaMemset it is asynchronous with respect to the host) and it runs on the NULL stream, which
// the library's non-blocking streams do not wait for: a kernel launched right behind it on one of them can write the buffer BEFORE the fill does.
// Seen as garbage in the FIRST evaluation after an allocation when eight processes shared one GPU (the fill arrived late): round 3.
inline int dev_fill_zero(void* p, size_t bytes) {
    HIP_TRY(hipMemset(p, 0, bytes));
    HIP_TRY(hipStreamSynchronize(nullptr));
    return 0;
}

// a blocking upload whose data the next kernel on a non-blocking stream may read: the copy and, to be independent of when exactly the runtime
// considers a pageable host-to-device copy finished, the NULL stream it ran on
inline hipError_t dev_upload(void* dst, const void* src, size_t bytes) {
    hipError_t e = hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice);
    return e != hipSuccess ? e : hipStreamSynchronize(nullptr);
}

template <typename T>
struct DevBuf {
    T* p = nullptr;
    size_t n = 0;
    bool borrowed = false;              // p points into memory somebody else owns (Spd::A over a RowBacked range): never freed here, never regrown
    int ensure(size_t count) {
        if (count <= n) return 0;
        if (borrowed) { set_error("DevBuf: a borrowed range cannot grow"); return -1; }
        if (p) { hipError_t e = hipFree(p); (void)e; p = nullptr; n = 0; }
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&p), count * sizeof(T)));
        n = count;
        return 0;
    }
    void release() { if (p && !borrowed) { hipError_t e = hipFree(p); (void)e; } p = nullptr; n = 0; borrowed = false; }
};

// A device address range that is reserved whole and given physical memory granule by granule, where somebody asks for it (HIP's virtual
// memory management: hipMemAddressReserve / hipMemCreate / hipMemMap).  The work matrix of a SHARDED exact evaluation lives in one: every rank
// keeps the global (row, column) -> address map of the one-GPU code, but only the granules under the tile rows it owns exist -- a rank of P
// holds ceil(T / P) tile rows of the N x N matrix (block-cyclic ownership), and a kernel that strays into somebody else's rows
// faults instead of reading stale numbers.  Everything mapped is zero-filled once.
struct RowBacked {
    // One physical allocation per 2 MB granule, every one mapped and given access by itself.  That is the form this runtime takes reliably: with
    // allocations of DIFFERENT sizes mapped into one reserved range hipMemSetAccess answers "invalid argument" from the second or third size on
    // (tools/micro/vmm_probe.hip, ROCm 7.2 on MI355X: runs of 4, 4, 1 granules fail at the third; equal sizes at any offsets never do).  The runtime reports a
    // granularity of 4 KB; 2 MB is the page size the device's address translation wants for a matrix that is streamed.
    static constexpr size_t GRAN = (size_t)2 << 20;
    char* base = nullptr;
    size_t bytes = 0, gran = GRAN;
    int device = 0;
    std::vector<hipMemGenericAllocationHandle_t> handle;          // per granule
    std::vector<char> have;                                       // per granule: mapped
    size_t nbacked = 0;
    int reserve(size_t nbytes, int dev) {
        release();
        bytes = (nbytes + gran - 1) / gran * gran;
        void* q = nullptr;
        HIP_TRY(hipMemAddressReserve(&q, bytes, gran, nullptr, 0));
        base = static_cast<char*>(q); device = dev;
        have.assign(bytes / gran, 0);
        handle.resize(bytes / gran);
        return 0;
    }
    // physical memory (zeroed) under every granule that meets [off, off + len)
    int back(size_t off, size_t len) {
        if (len == 0) return 0;
        if (!base || off + len > bytes) { set_error("RowBacked::back: outside the reserved range"); return -1; }
        const size_t g_lo = off / gran, g_hi = (off + len - 1) / gran;
        hipMemAllocationProp prop{};
        prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = device;
        hipMemAccessDesc acc{};
        acc.location.type = hipMemLocationTypeDevice; acc.location.id = device; acc.flags = hipMemAccessFlagsProtReadWrite;
        for (size_t g = g_lo; g <= g_hi;) {
            if (have[g]) { ++g; continue; }
            size_t e = g;
            for (; e <= g_hi && !have[e]; ++e) {
                HIP_TRY(hipMemCreate(&handle[e], gran, &prop, 0));
                hipError_t me = hipMemMap(base + e * gran, gran, 0, handle[e], 0);
                if (me == hipSuccess) me = hipMemSetAccess(base + e * gran, gran, &acc, 1);
                if (me != hipSuccess) { hipError_t x = hipMemUnmap(base + e * gran, gran); x = hipMemRelease(handle[e]); (void)x; HIP_TRY(me); }
                have[e] = 1; ++nbacked;
            }
            { int z__ = dev_fill_zero(base + g * gran, (e - g) * gran); if (z__) return z__; }
            g = e;
        }
        return 0;
    }
    size_t backed_bytes() const { return nbacked * gran; }
    void release() {
        for (size_t g = 0; g < have.size(); ++g)
            if (have[g]) {
                hipError_t e = hipMemUnmap(base + g * gran, gran); (void)e;
                e = hipMemRelease(handle[g]); (void)e;
            }
        have.clear(); handle.clear(); nbacked = 0;
        if (base) { hipError_t e = hipMemAddressFree(base, bytes); (void)e; }
        base = nullptr; bytes = 0;
    }
};

// phase-table workspace for Gram / moment launches over one (row inputs, column inputs) combination: scratch + the device copies
// of the two channel-offset arrays (uploaded only when they change)
struct PhaseWs {
    DevBuf<double> ws;
    DevBuf<int> offr, offc;
    std::vector<int> hr, hc;
    int prepare(const std::vector<int>& r, const std::vector<int>& c, int C, int T, int64_t ldr, int64_t ldc, hipStream_t s, PhaseRef& out) {
        int rc;
        if ((rc = ws.ensure(phase_ws_doubles(C, T, ldr, ldc)))) return rc;
        if (hr != r) {
            if ((rc = offr.ensure(r.size()))) return rc;
            hr = r;
            HIP_TRY(hipMemcpyAsync(offr.p, hr.data(), hr.size() * sizeof(int), hipMemcpyHostToDevice, s));
        }
        if (hc != c) {
            if ((rc = offc.ensure(c.size()))) return rc;
            hc = c;
            HIP_TRY(hipMemcpyAsync(offc.p, hc.data(), hc.size() * sizeof(int), hipMemcpyHostToDevice, s));
        }
        out.offr = offr.p; out.offc = offc.p; out.ws = ws.p;
        return 0;
    }
    void release() { ws.release(); offr.release(); offc.release(); hr.clear(); hc.clear(); }
};

// channel-sorted view of an input matrix X (M x (1+D)): stable sort by channel id
struct SortedX {
    int64_t M = 0, Mpad = 0;
    std::vector<int64_t> perm;        // sorted position -> original row
    std::vector<int> off;             // [C+1]
    std::vector<double> xs;           // [D][Mpad]
    bool identity = true;
};


int fail(int code, const std::string& msg);
static inline int64_t round_up(int64_t v, int64_t m) { return (v + m - 1) / m * m; }
int sort_inputs(const double* X, int64_t M, int D, int C, int64_t pad_to, SortedX& o);
void build_sym_tiles(const std::vector<int>& off, int C, std::vector<GTile>& tiles, std::vector<int>& pair_start);
void build_rect_tiles(const std::vector<int>& offr, const std::vector<int>& offc, int C, std::vector<GTile>& tiles,
                      std::vector<int>* pair_start = nullptr);

}  // namespace mogp

using namespace mogp;      // private header: the global handle structs below are built from mogp:: types

enum { MOGP_COMM_NONE = 0, MOGP_COMM_RCCL = 1, MOGP_COMM_EXTERNAL = 2 };
struct mogp_comm {                       // communicator of the sharded evaluation (comm.hip)
    int kind = MOGP_COMM_NONE, rank = 0, n = 1;
    void* nccl = nullptr;                // ncclComm_t
    mogp_allgather_cb allgather = nullptr;
    mogp_allreduce_cb allreduce = nullptr;
    void* user = nullptr;
};

struct mogp_ctx {
    int device = 0;
    mogp_comm comm;
    std::string name;
    // streams shared by every model of the context (created once: a CU-masked stream owns a hardware queue, and models come and go)
    hipStream_t st = nullptr, st2 = nullptr, st3 = nullptr, st4 = nullptr, st5 = nullptr, st_priv = nullptr, st2u = nullptr;
    bool streams_ready = false;
    bool chain_ok = true;               // the private stream has room for the 13 workgroups of the persistent chain kernel (chain.hip)
    // stream-K GEMM launches (linalg.hip:k_gemm_sk): one workspace per stream (launches of one stream never overlap)
    struct SkWs { DevBuf<double> ws; DevBuf<unsigned> flags; unsigned epoch = 0; };
    std::map<hipStream_t, SkWs> sk;
    int ncu = 0, ncu_reserved = 0;      // CUs of the device, and how many of them the private stream owns
};

struct TrtriLevel {
    std::vector<GemmTask> h1, h2;
    DevBuf<GemmTask> d1, d2;
    double flops1 = 0, flops2 = 0;
};

// dense SPD workspace: A (Npad x Npad, lower) -> L -> W = L^-1 in place; B = scratch, then W^T W
#define MOGP_CHAIN_BOUND_TILES 160   // up to this many 128-row tiles (N <= 20480) the bulk work stays off the reserved CUs
#define MOGP_NPANEL 4       // the factorisation may run this many outer blocks ahead of the inverse stream
struct Spd {
    int64_t Npad = 0;
    int nb = 0;
    bool keep_L = false;                // spd_potrf also stores the diagonal tiles of L (the triangular-solve path needs the factor itself)
    bool refine_panels = false;         // spd_potrf refines every panel once against L_kk itself (ill-conditioned K_uu of the sparse models; needs keep_L)
    DevBuf<double> pscr;                // its scratch: one panel (Npad x 128)
    DevBuf<double> A, B, invd, logdet;
    RowBacked Arows;                    // owned-rows form (a sharded evaluation's work matrix): A.p points into it, B does not exist until someone needs it
    bool owned_rows = false;
    DevBuf<double> Wm;                  // W = L^-1 when the inverse is streamed behind the factorisation (spd_potrf fuse_inverse)
    std::vector<TrtriLevel> levels;
    std::vector<hipEvent_t> sync_ev;    // cross-stream dependencies of the look-ahead schedule
    std::vector<hipEvent_t> row_ev;     // want_row_ev: spd_potrf records row_ev[k] behind the leaf of tile column k -- block row k of L (its tiles left of the diagonal and
    bool want_row_ev = false;           // L_kk itself) is final then, and a forward substitution on another stream may take it (trsm_lower's row_ready)
    std::vector<hipEvent_t> inv_ev;     // events of the fused schedule (potri.hip)
    hipEvent_t fused_last_inv = nullptr, fused_last_wt = nullptr;
    DevBuf<double> Wd;                  // W_KK = L_KK^-1 of every outer block of the fused schedule (512 x 512 each, wkk.hip)
    DevBuf<double> Pb[MOGP_NPANEL];     // rotating panel buffers L[>K, K] of the fused schedule (Npad x 512 each)
    DevBuf<unsigned> chain_flags;       // hand-off words of the persistent chain kernel (chain.hip): MOGP_CHAIN_FLAGS per outer block + the error word
    // the dataflow form (flow.hip): panels at their natural position, the running product of the block-column inverses, the task graph
    DevBuf<double> Lm, Wt;
    FlowPlan flow;
    DevBuf<FlowTask> flow_tasks;
    DevBuf<int> flow_qmeta;
    FlowPlan flow_replay;                               // the measurement plan of mogp_model_flow_replay and its device copy
    DevBuf<FlowTask> flow_tasks_replay;
    DevBuf<int> flow_qmeta_replay;
    FlowPlan flow_rhs;                                  // the prediction's plan (factorisation + forward substitution of right-hand sides) and its device copy
    DevBuf<FlowTask> flow_tasks_rhs;
    DevBuf<int> flow_qmeta_rhs;
    const FlowPlan* flow_cur = nullptr;                 // the plan of the last dataflow run (debug dump, trace)
    DevBuf<unsigned> flow_flags;
    DevBuf<char> flow_args;             // the dataflow kernel's arguments in device memory (two 512-byte slots: the kernel, its small second instance)
    std::vector<char> flow_args_h;      // ... and their host copies
    DevBuf<unsigned char> flow_done;    // MOGP_FLOW_DEBUG: one byte per task of the last dataflow evaluation, set when it has signalled
    DevBuf<unsigned> flow_post;         // MOGP_FLOW_DEBUG: per-workgroup post-mortem of the last dataflow evaluation (FLOW_POST_W words each)
    DevBuf<unsigned> flow_diag;         // FLOW_DIAG_WORDS counters of the schedule's deep looks (zeroed once, never per evaluation)
    DevBuf<unsigned long long> flow_trace;
    hipEvent_t tail_ready = nullptr;    // set by the caller for ONE factorisation: everything of A right of the first 512 columns is in place after this event
                                        // (the Gram build in two launches: the second one runs on the bulk stream underneath the first chain kernel)
    bool flow_used = false;             // the last fused factorisation + inversion of this workspace ran as dataflow
    // the caller wants z = W y and the row blocks' shares of alpha = W^T z from the same kernel: y, z [Npad], z^T z parts [(Npad + 3) / 4],
    // shares [nouter][Npad] (device); vec_done: the last factorisation delivered them (launch_flow_alpha_sum adds the shares up)
    bool want_vec = false, vec_done = false;
    const double* vec_y = nullptr; double* vec_z = nullptr; double* vec_zz = nullptr; double* vec_part = nullptr;
    void release() {
        for (auto& lv : levels) { lv.d1.release(); lv.d2.release(); }
        for (auto e : sync_ev) { hipError_t r = hipEventDestroy(e); (void)r; }
        for (auto e : row_ev) { hipError_t r = hipEventDestroy(e); (void)r; }
        row_ev.clear();
        for (auto e : inv_ev) { hipError_t r = hipEventDestroy(e); (void)r; }
        inv_ev.clear(); Wm.release(); Wd.release(); chain_flags.release(); for (auto& b : Pb) b.release();
        Lm.release(); Wt.release(); flow_tasks.release(); flow_qmeta.release(); flow_flags.release(); flow_args.release(); flow_args_h.clear(); flow_diag.release(); flow_post.release(); flow_done.release(); flow_trace.release(); flow = FlowPlan();
        flow_tasks_rhs.release(); flow_qmeta_rhs.release(); flow_rhs = FlowPlan(); flow_cur = nullptr;
        flow_tasks_replay.release(); flow_qmeta_replay.release(); flow_replay = FlowPlan();
        levels.clear(); sync_ev.clear();
        A.release(); B.release(); invd.release(); logdet.release(); pscr.release();
        Arows.release(); owned_rows = false;
    }
};

// a tile list split into runs of full interior tiles (strip kernel) and the rest; see split_strip_tiles
struct StripTiles {
    std::vector<mogp::GSeg> segs;
    std::vector<mogp::GTile> rest;
    mogp::DevBuf<mogp::GSeg> d_segs;
    mogp::DevBuf<mogp::GTile> d_rest;
    int build(const std::vector<mogp::GTile>& tiles);      // host split + upload
    void attach(mogp::GramArgs& ga) const { ga.segs = d_segs.p; ga.nsegs = (int)segs.size(); ga.rest = d_rest.p; ga.nrest = (int)rest.size(); }
    void release() { d_segs.release(); d_rest.release(); }
};

// workspaces of the Titsias sparse bound (config 5): two M x M SPD systems and three M x N panels
struct TitsiasWork {
    int64_t Mpad = 0;
    Spd a, q;                                           // Kuu (+jitter) and Qs = v v^T / s2 + I
    DevBuf<double> zx, B, v, GB, Qs, E, R, T1, GA, Hm;  // inputs [D][Mpad]; Kuf, W Kuf, dELBO/dKuf (Mpad x Npad); M x M temporaries
    DevBuf<double> vec, scratch, gz, partial_uu, partial_uf, mom_uu, mom_uf, zero_noise;
    DevBuf<GTile> tiles_uu, tiles_uf;
    DevBuf<int> ps_uu, ps_uf;
    StripTiles strip_uf;                                // (Z, X) tiles as strip-kernel runs + the rest (titsias_front)
    std::vector<int> tile_key;                          // channel offsets of Z and X the device copies of the four lists above were made for
    size_t n_tuu = 0, n_tuf = 0;                        // their lengths (the host lists themselves are built only when the key changes)
    DevBuf<double> Kus, Aus, Bus;                       // prediction panels (Mpad x Spad)
    DevBuf<double> zero_col;                            // Mpad zeros
    DevBuf<double> kslices;                             // split-K partial sums of the Qs SYRK (ks x Mpad x Mpad)
    DevBuf<double> red;                                 // data-sharded evaluation: [v y | y^T y, N, sum K_ff,nn] for the all-reduce
    const double* Wq = nullptr;                         // L_q^-1 of the inner M x M system (where spd_invert left it)
    SortedX pred_ss;                                    // the test inputs of the last sparse prediction (a = L^-1 Kus in Aus, b in Bus): what
    bool pred_valid = false;                            // mogp_sparse_predict_cov needs for the full covariance K_ss - a^T a + b^T b
    hipEvent_t side_ev[2] = {nullptr, nullptr};         // fork / join of the M x M adjoint chain on a side stream (side_fork / side_join)
    hipEvent_t prod_ev[2] = {nullptr, nullptr};         // fork / join of the backward part's M x M x N product on the normal-priority stream (titsias.hip)
    DevBuf<int> blk_z, blk_x;                           // [first point, count] of the 64-point blocks of Z and of X (tile_blocks): the slots of
    std::vector<int> hblk_z, hblk_x;                    // the fixed-order reduction of d/dZ (gz_prepare / gz_attach)
    DevBuf<double> gzp;
    // svgp.hip: what the forward pass at the training inputs leaves for the backward pass
    SortedX sv_sz; std::vector<GTile> sv_tuu, sv_tuf; std::vector<int> sv_psuu, sv_psuf; int64_t sv_M = 0; bool sv_dense = false, sv_valid = false;
    DevBuf<double> kd_point;                            // Snelson / Hensman with enveloped terms: the kernel diagonal per training point (sorted order)
    DevBuf<double> nvec;                                // Snelson: per-point vectors (g, G, G y, sqrt G, v^T r / w, alpha, h) + per-channel inputs
    PhaseWs ph_zz, ph_zx, ph_zs;                        // phase tables: (Z, Z), (Z, X), (Z, Xs)
    void release() {
        ph_zz.release(); ph_zx.release(); ph_zs.release();
        a.release(); q.release();
        zx.release(); B.release(); v.release(); GB.release(); Qs.release(); E.release(); R.release(); T1.release(); GA.release(); Hm.release();
        vec.release(); scratch.release(); gz.release(); partial_uu.release(); partial_uf.release(); mom_uu.release(); mom_uf.release();
        zero_noise.release(); tiles_uu.release(); tiles_uf.release(); ps_uu.release(); ps_uf.release(); strip_uf.release(); tile_key.clear();
        Kus.release(); Aus.release(); Bus.release(); zero_col.release(); kslices.release(); nvec.release(); kd_point.release(); red.release();
        blk_z.release(); blk_x.release(); gzp.release(); hblk_z.clear(); hblk_x.clear();
        for (auto& e : side_ev) if (e) { hipError_t r = hipEventDestroy(e); (void)r; e = nullptr; }
        for (auto& e : prod_ev) if (e) { hipError_t r = hipEventDestroy(e); (void)r; e = nullptr; }
    }
};

// workspaces of the Opper-Archambeau model (oa.hip): K itself, two N x N temporaries, the per-point vectors
struct OaWork {
    mogp::DevBuf<double> K, Sc, Y, vec;
    bool valid = false;                 // a forward pass left its state for the backward call
    void release() { K.release(); Sc.release(); Y.release(); vec.release(); valid = false; }
};

// right-hand sides of the prediction's dataflow schedule: X L^T = T, `nt` tile rows (row-major, the leading dimension of the N x N matrices), `ready`: T is in place
struct FlowRhs { double* T; double* X; int nt; hipEvent_t ready; };
struct mogp_model {
    mogp_ctx* ctx = nullptr;
    int64_t N = 0, Npad = 0;
    int nb = 0, D = 0, C = 0, T = 0;
    int Wt = 0;                         // width of a term-table row = moments per (pair, term): 2 + 3 D, or 2 + 5 D with an envelope (MOHSM)
    std::vector<double> point_diag;     // K_diag per training point (channel-sorted) when the diagonal is not constant per channel
    SortedX sx;
    std::vector<GTile> tiles;
    std::vector<int> pair_start;
    std::vector<double> table;          // host copy [C*C*T*W]
    hipStream_t st = nullptr;           // critical-path stream (high priority)
    hipStream_t st2 = nullptr;          // bulk trailing updates of the fused schedule (CU-masked: everything but the reserved CUs)
    hipStream_t st2u = nullptr;         // bulk trailing updates over ALL CUs, for flop-bound sizes (MOGP_CHAIN_BOUND_TILES)
    hipStream_t st3 = nullptr;          // inverse streamed behind the factorisation (lowest priority)
    hipStream_t st4 = nullptr;          // the inverse's rank-512 accumulations W[K,:]^T W[K,:] (nothing but the result depends on them)
    hipStream_t st_priv = nullptr;      // intra-block chain of the fused factorisation, on the reserved CUs only
    Spd k;                              // the N x N system
    Spd ws, ws_tail;                    // Schur-block workspaces of the sweep inversion (outer block / last partial block)
    DevBuf<double> swU[2], swUr[2];     // old panels of the block being swept (column part, row part), double buffered
    DevBuf<double> swXr[2];             // owned-rows form: the NEW row part of the panel (all four pivot tile rows; the matrix keeps the owned ones)
    std::vector<hipEvent_t> sw_ev;
    int sh_rank = 0, sh_n = 1;          // sharded evaluation: this rank owns tile rows i with i % sh_n == sh_rank
    bool sh_owned = false;              // ... in the owned-rows form: nothing outside the owned tile rows of k.A is read or written (sweep.hip); set by
                                        // mogp_shard_config, cleared by every one-GPU entry point
    int backed_rank = -1, backed_n = 0; // (rank, nranks) k.Arows has its granules for
    std::vector<GTile> tiles_own;       // the Gram / moment tiles that touch an owned tile row (grouped by pair like `tiles`)
    std::vector<int> pair_start_own;
    DevBuf<GTile> d_tiles_own;
    bool factor_only = false;           // the last factorisation stopped at L (prediction): no W, no alpha
    hipEvent_t pred_ev[2] = {nullptr, nullptr};      // fork / join of the prediction's side stream
    StripTiles strip, strip_own;        // the same tile lists (all / owned) split for the Gram strip kernel
    // the Gram build in two launches for the dataflow schedule: the tiles that start in the first 512 columns (all the first chain kernel and the
    // first panel read) and the rest, which is then built on the bulk stream UNDERNEATH the first chain kernel
    std::vector<GTile> tiles_head, tiles_tail;
    DevBuf<GTile> d_tiles_head, d_tiles_tail;
    StripTiles strip_head, strip_tail;
    hipEvent_t gram_ev = nullptr, gram_tail_ev = nullptr;
    DevBuf<int> d_pair_start_own;
    int own_rank = -1, own_n = 0;       // (rank, nranks) the owned lists were built for
    DevBuf<double> sh_send, sh_recv;
    DevBuf<double> sh_send1, sh_recv1;  // the pivot block's own rows (the first, small message of a split exchange)
    DevBuf<double> sh_fact;             // factor-once: [P | log-det parts | pivot report]
    bool sh_factor_once = false;        // MOGP_SHARD_FACTOR_ONCE, read per sharded evaluation
    bool sh_split = false;              // the exchange in two messages, the large one on the communication stream (MOGP_SHARD_SPLIT=0: one message)
    std::vector<hipEvent_t> sh_ev;      // split exchange: per pivot block [packed, rest of the panel in place]
    std::vector<hipEvent_t> sh_prof;    // profiling (mogp_set_profiling) of a sharded evaluation: 6 timing events per pivot block
    double sh_ms[6] = {0, 0, 0, 0, 0, 0};   // ... summed over the blocks: exchange on the critical stream, serial part, next-block columns, bulk update, exchange on the communication stream, the critical stream's wait for it (mogp_shard_stage_ms)
    int sh_prof_blocks = 0;
    double sh_jabs = 0.0;
    bool sh_dvar = false;

    std::vector<double> hy;             // channel-sorted targets (host copy, Npad)
    DevBuf<double> d_symv;
    DevBuf<double> d_x, d_y, d_table, d_noise, d_dvar, d_z, d_alpha, d_zz, d_partial, d_moments, d_diagG;
    DevBuf<GTile> d_tiles;
    // Tiles of Kj^-1 the gradient actually reads (mogp_api.hip:kinv_plan): where every term of dK/dtheta is below e^-50 of its peak in a
    // 64 x 64 tile the moment kernel skips the tile, so the 128 x 128 tiles of the inverse under such tiles only are never formed.
    std::vector<GemmTask> kinv_acc_tasks, kinv_lauum_tasks;        // host lists of the current plan (tile-row-major)
    std::vector<int> kinv_prefix;                                 // [nb + 1]: tasks with tile row < r
    DevBuf<GemmTask> d_kinv_acc, d_kinv_lauum;
    std::vector<double> blk_cen, blk_half;                        // [D][nblk]: centre / half span of every 64-point block (k_block_centres' numbers)
    bool kinv_sparse = false;                                     // the inverse held by k.B lacks the tiles outside the plan
    double kinv_fraction = 1.0;                                   // planned / all lower tiles
    DevBuf<int> d_pair_start, d_chan_off, d_flag;
    DevBuf<double> d_pivots;            // [min, max] diagonal entry of the last factorisation's L (k_pivot_range)
    bool accurate = false, accurate_ran = false;      // mogp_model_set_accurate: gradient evaluations by refined panels + substitutions (factorize); the last one did
    DevBuf<double> acc_rhs;             // its 128-column right-hand-side block (y -> z -> alpha)
    size_t pin_pivots = 0;              // where in the pinned block they come back
    double pivot_min = 0.0, pivot_max = 0.0;      // the same on the host, 0 when the last evaluation did not report them (sweep / sharded)
    DevBuf<unsigned long long> d_info;

    // prediction workspaces
    DevBuf<double> d_xs, d_Ksf, d_Vt, d_mu, d_var, d_kdiag, d_Kss;
    DevBuf<GTile> d_ptiles;
    DevBuf<GemmTask> d_pred_tasks;      // the sharded prediction's task lists (this rank's tile rows x test-point tiles)
    PhaseWs ph_xx, ph_sx, ph_ss;                        // phase tables: (X, X), (Xs, X), (Xs, Xs)

    double* h_pin = nullptr;            // pinned host block for the per-evaluation scalars (log-det parts, z^T z parts, pivot report, moments):
    size_t h_pin_n = 0;                 // asynchronous device-to-host copies need page-locked memory, and with them an evaluation has ONE stream sync

    // profiling
    bool profiling = false;
    std::vector<hipEvent_t> ev;          // stage boundaries
    std::vector<hipEvent_t> gemm_ev;     // pairs around GEMM launches
    size_t gemm_ev_used = 0;
    double ms[MOGP_ST_COUNT] = {0};
    int64_t gemm_launches = 0;
    double gemm_flops = 0.0;
    bool have_W = false, have_Kinv = false, kinv_in_A = false, w_in_Wm = false;
    bool flow_ran = false;              // some factorisation of this model has used the dataflow schedule since the last fallback
    const FlowRhs* rhs_job = nullptr;   // set by the prediction around its factorize() call: solve these right-hand sides inside the dataflow schedule
    bool replay_flow = false;           // mogp_model_flow_replay: the next gradient evaluations run the dataflow kernel ALONE on the measurement plan (no chain kernels)
    bool no_flow = false;               // the dataflow kernel timed out: stream schedule until evaluation number flow_retry_at (chain_fallback)
    long long n_fact = 0, flow_retry_at = 0;      // factorisations so far; when the dataflow schedule is tried again
    int flow_backoff = 64, flow_timeouts = 0;
    double flow_enqueue_us = 0.0, flow_enqueue_us_max = 0.0;      // host time spent enqueuing the last dataflow evaluation / the longest so far
    bool no_chain = false;              // the persistent chain kernel timed out once on this model (another process's chain kernel held the reserved CUs): launch-per-step chain from now on
    TitsiasWork* tw = nullptr;
    OaWork oa;
};


namespace mogp {
int use_device(mogp_ctx* c);
int gemm_call(mogp_model* m, const GemmArgs& g, double flops, hipStream_t st = nullptr);
int mark(mogp_model* m, int idx);
double table_diag(const mogp_model* m, int c);
double table_diag_points(const mogp_model* m, const SortedX& pts);      // sum of K(x, x) over the points (per point when the terms carry an envelope)
int spd_alloc(Spd& w, int64_t Npad, int owned_rows_device = -1);   // >= 0: the owned-rows form (A reserved on that device, nothing backed yet, no B)
int spd_make_whole(Spd& w);           // an owned-rows workspace becomes an ordinary one (every granule of A backed, B allocated)
// helpers shared by the sparse / variational models (titsias.hip)
inline GemmArgs make_gemm(const double* A, int64_t lda, int akm, const double* B, int64_t ldb, int bkm, double* C, int64_t ldc,
                          double alpha, int mode, int mt, int nt, int64_t K) {
    GemmArgs g{};
    g.A = A; g.lda = lda; g.a_kmajor = akm; g.B = B; g.ldb = ldb; g.b_kmajor = bkm; g.C = C; g.ldc = ldc;
    g.alpha = alpha; g.beta = 0.0; g.mode = mode; g.mt = mt; g.nt = nt; g.K = (int)K;
    return g;
}
// The M x M part of a sparse model's backward pass (two triangular solves with 2 nb dependent, nearly empty launches: ~3 ms at M = 2048) depends
// only on M x M inputs; side_fork returns a second stream that starts behind everything enqueued on m->st so far, side_join makes m->st wait
// for it -- the chain then runs underneath the M x N work instead of in front of it.  MOGP_SIDE_STREAM=0: everything on m->st.
int side_fork(mogp_model* m, TitsiasWork& t, hipStream_t* side);
int side_join(mogp_model* m, TitsiasWork& t, hipStream_t side);
// d/dZ of a sparse model: block tables of Z and X and the scratch of the fixed-order reduction (gram.hip: k_gz_reduce); gz_attach points a
// moment pass at them -- zx: rows Z, columns X (the (Z, X) pass); otherwise rows and columns Z
int gz_prepare(mogp_model* m, TitsiasWork& t, const std::vector<int>& offz, int D);
void gz_attach(const TitsiasWork& t, MomentArgs& ma, bool zx);
int spd_check_info(mogp_model* m, const char* which, int64_t* info);
int spd_info_verdict(mogp_model* m, const char* which, unsigned long long hinfo, int64_t* info);   // what spd_check_info concludes from the word
// after the LAST stream sync of a sparse / variational evaluation: did a hand-off between workgroups (stream-K GEMM of a wide triangular solve,
// chain kernel) time out anywhere?  Then the numbers are not valid: the model drops those forms and the call fails, loudly (titsias.hip)
int sparse_timeout_check(mogp_model* m);
bool chain_enabled(const mogp_model* m);   // chain.hip
int chain_fallback(mogp_model* m);     // mogp_api.hip: after MOGP_INFO_CHAIN_TIMEOUT -- drain, switch the model to the launch-per-step chain; the caller repeats the evaluation
// w.A (SPD, lower tiles) -> w.B = its inverse (lower tiles, full diagonal tiles) and *W = L^-1 (lower; in w.A, or in w.Wm on the fused path),
// w.logdet per tile: POTRF, TRTRI, LAUUM.  MOGP_SPARSE_FUSED=1 takes the fused factorisation + inversion schedule of the exact path
// (potri.hip) instead -- measured SLOWER for the 16-tile-row systems of configs[4] (52.0 vs 49.9 ms per evaluation: four outer blocks give
// its streams nothing to overlap), kept as a switch.  A failed factorisation is reported through `info` / MOGP_ENOTPD naming `which`.
int spd_invert(mogp_model* m, Spd& w, const char* which, int64_t* info, const double** W, bool defer_check = false);   // the pivot report of the last factorisation -> MOGP_ENOTPD naming `which`
// out (Mpad x Mpad, lower tiles) = alpha A B^T over K (leading dimension ldk), K cut into slices so that the launch fills the chip
int mm_lower_splitk(mogp_model* m, TitsiasWork& t, const double* A, const double* B, double* out, int mt, int64_t Mpad, int64_t ldk, int64_t K,
                    double alpha = 1.0);
void one_gpu_call(mogp_model* m);   // ownership state of a previous sharded evaluation off (every one-GPU entry point calls this first)
int ensure_system(mogp_model* m);     // the N x N system of the exact / OA paths and the tile lists over (X, X), on first use (mogp_api.hip)
int spd_potrf(mogp_model* m, Spd& w, long long info_base = 0);
int spd_potri_fused(mogp_model* m, Spd& w);
bool flow_enabled(const mogp_model* m, const Spd& w);   // flow.hip
int launch_flow_alpha_sum(const Spd& w, double* alpha, hipStream_t st);
int spd_potri_flow(mogp_model* m, Spd& w, const FlowRhs* rhs = nullptr);
void flow_debug_dump(mogp_model* m);                 // MOGP_FLOW_DEBUG (mogp_api.hip)             // flow.hip: the same result as spd_potri_fused, as tile dataflow
int spd_potri_fused_finish(mogp_model* m, Spd& w);   // joins the inverse stream: call before reading w.B   // potri.hip: w.A (SPD, lower) -> w.Wm = L^-1, w.B = inverse (lower); w.logdet per tile
int spd_trtri(mogp_model* m, Spd& w);
int spd_lauum(mogp_model* m, Spd& w);
int spd_sweep(mogp_model* m, Spd& w);
int sweep_prepare(mogp_model* m, Spd& w);
int sweep_nblocks(const Spd& w);
int sweep_block(mogp_model* m, Spd& w, int kb, hipEvent_t* prof = nullptr, hipEvent_t panel_ready = nullptr, hipEvent_t* stall = nullptr);   // prof (4 timing events or null): panels ready / next-block columns done (critical stream), bulk start / end
int sweep_finish(mogp_model* m, Spd& w);
// B (nb*128 rows x ncols, leading dimension ldb, ncols a multiple of 128) <- L^-1 B  (trans: L^-T B) by blocked substitution, in place;
// L lower triangular nb*128 square with leading dimension ldl, diagonal tiles included (Spd::keep_L).  trsm.hip
int trsm_lower(mogp_model* m, const double* L, int64_t ldl, int nb, double* B, int64_t ldb, int64_t ncols, bool trans, hipStream_t st = nullptr, bool tri = false,
               const hipEvent_t* row_ready = nullptr);   // row_ready[i] (forward, wide right-hand sides): block row i of L is final after this event -- the substitution follows a factorisation still running (Spd::row_ev)
int launch_transpose(double* dst, const double* src, int64_t ld, int64_t n, hipStream_t s);            // dst = src^T, n x n, n % 64 == 0
int launch_sym_lower_avg(double* A, int64_t ld, int64_t n, double scale, hipStream_t s);               // lower(A) <- scale * (A + A^T) / 2
int comm_allgather(mogp_ctx* ctx, const double* send, double* recv, int64_t count, hipStream_t st);   // count doubles per rank, device memory
int comm_allreduce(mogp_ctx* ctx, double* buf, int64_t count, hipStream_t st);                        // sum, in place
int shard_pack(mogp_model* m, Spd& w, int kb, double** send, double** recv, int64_t* count);
int shard_unpack(mogp_model* m, Spd& w, int kb);
int shard_pack_part(mogp_model* m, Spd& w, int kb, int part, DevBuf<double>& sendb, DevBuf<double>& recvb, int64_t* count, hipStream_t st);
int shard_unpack_part(mogp_model* m, Spd& w, int kb, int part, DevBuf<double>& recvb, hipStream_t st);
int shard_factor_bcast(mogp_model* m, Spd& w, Spd& s, int k0, int nk, bool mine, hipStream_t st);
// w.A (SPD, lower) -> -inverse (lower); w.logdet per tile; failure through m->d_info
}  // namespace mogp
