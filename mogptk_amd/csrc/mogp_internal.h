// mogp_internal.h -- shared declarations of libmogp_hip.so (gfx950 only; no other target is supported).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>
#include <string>
#include <vector>

#define MOGP_TILE 128          // linear-algebra tile edge: every dense matrix is padded to a multiple of it
#define MOGP_GT 64             // Gram / moment tile edge (relative to channel blocks)
#define MOGP_TC 8              // spectral terms processed per LDS chunk in the Gram / moment kernels
#define MOGP_MAXD 8            // maximum input dimension

namespace mogp {

// ---- error plumbing --------------------------------------------------------------------------------
void set_error(const std::string& msg);
int hip_fail(hipError_t e, const char* what, const char* file, int line);
#define HIP_TRY(x)                                                        \
    do {                                                                  \
        hipError_t e__ = (x);                                             \
        if (e__ != hipSuccess) return mogp::hip_fail(e__, #x, __FILE__, __LINE__); \
    } while (0)

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-DEVICE attribute: set it once per (kernel, device) -- a process may hold contexts on
// several GPUs, and host threads may evaluate on them at the same time (a function-local `static bool` covered neither: ADVICE round 3)
inline int set_max_dynamic_lds(const void* fn, int bytes, std::atomic<unsigned long long>& done) {
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    const unsigned long long bit = 1ull << (dev & 63);
    if (done.load(std::memory_order_acquire) & bit) return 0;
    HIP_TRY(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    done.fetch_or(bit, std::memory_order_release);
    return 0;
}

// ---- Gram / moment tiles ---------------------------------------------------------------------------
// One descriptor per 64x64 tile of one channel-pair block (never straddles a channel boundary).
struct GTile {
    int r0, c0;      // first row / column (global, channel-sorted order)
    int nr, nc;      // valid rows / columns (<= MOGP_GT)
    int pair;        // i*C + j : row channel i, column channel j
    int flags;       // GT_* bits
    int rb, cb;      // index of the tile's 64-point row / column block among all row / column blocks (channel by channel): the slot of its
};                   // per-point input gradients in the fixed-order reduction (k_gz_reduce)
// a run of n consecutive FULL interior tiles of one row block: columns c0, c0 + 64, ... (the strip kernel of gram.hip)
struct GSeg { int r0, c0, n, pair; int diag, pad[3]; };      // a run of n consecutive full 64 x 64 tiles of one row block; diag: the LAST one sits on the matrix diagonal
enum { GT_MIRROR = 1,     // also write the transpose to (c, r)   (off-diagonal tile of the symmetric Gram)
       GT_DIAG = 2 };     // tile sits on the matrix diagonal (r0 == c0)

// phase-table workspace of one Gram / moment launch (gram.hip): device channel offsets [C+1] of the row / column inputs and
// phase_ws_doubles(C, T, ldxr, ldxc) doubles of scratch
struct PhaseRef {
    const int* offr = nullptr;
    const int* offc = nullptr;
    double* ws = nullptr;
};
size_t phase_ws_doubles(int C, int T, int64_t ldr, int64_t ldc);

struct GramArgs {
    const GTile* tiles;
    const double* xr;      // row inputs   [D][ldxr]
    const double* xc;      // column inputs [D][ldxc]
    int64_t ldxr, ldxc;
    int64_t nrows, ncols;  // number of row / column points (all channels)
    PhaseRef ph;
    const double* table;   // [C*C][T][W]
    int T, D, C;
    int W;                 // width of a table row: 2 + 3 D, or 2 + 5 D when the terms carry a Gaussian envelope on the input midpoint
                           // ([.., L_d, c_d]: K *= exp(-1/2 sum_d L_d ((x_a,d + x_b,d)/2 - c_d)^2), MOHSM); 0 -> 2 + 3 D
    double* out;           // row-major, leading dimension ldo
    int64_t ldo;
    double* out2;          // when non-null: a second copy of every stored element, same layout (rectangular Grams: K_uf and the working copy the
                           // triangular solve overwrites, instead of a 1.6 GB device-to-device copy at configs[4])
    // diagonal augmentation (symmetric training Gram only; null -> none)
    const double* noise;   // [C] sigma_c^2
    const double* dvar;    // [N] per-point variance or null
    double jitter_abs;
    int mirror;            // write the transpose of GT_MIRROR tiles too (full symmetric Gram for Kernel.K)
    hipEvent_t ev0, ev1;   // when non-null: recorded around the tile kernel alone (profiling)
    const GSeg* segs; int nsegs;      // optional split of `tiles`: runs of full interior tiles (strip kernel: D = 1, no envelope, no mirror) ...
    const GTile* rest; int nrest;     // ... and everything else (diagonal, ragged and odd-aligned tiles) for the general kernel
    int dbg;               // measurement only (MOGP_GRAM_DBG): 1 = no stores, 2 = no terms (stores only)
    int tab_lds;           // set by the launcher: the term table is copied to LDS
    int phases_ready;      // the phase workspace already holds this table's phases and block centres (an earlier launch of the same evaluation)
};

struct MomentArgs {
    const GTile* tiles;
    int ntiles;
    const double* x;       // row inputs [D][ldx]
    int64_t ldx;
    const double* xc;      // column inputs [D][ldxc]; null -> the row inputs (symmetric case)
    int64_t ldxc;
    int64_t nrows, ncols;  // number of row / column points (ncols unused when xc is null)
    PhaseRef ph;
    const double* table;
    int T, D, C;
    int W;                 // table row width = number of moments per (pair, term): 2 + 3 D, or 2 + 5 D with the envelope (two more moments
                           // per dimension: m5_d = sum g a_d^2 E cos, m6_d = sum g a_d E cos, a_d = (x_a,d + x_b,d)/2 - c_d); 0 -> 2 + 3 D
    // adjoint source, exact mode (G == null):  g = w * 1/2 (alpha_a alpha_b - kinv_ab), symmetric weights
    const double* kinv;    // lower triangle valid, leading dimension ld
    int64_t ld;
    const double* alpha;   // [N]
    double kinv_sign;      // +1: kinv holds Kj^-1; -1: it holds -Kj^-1 (result of the sweep inversion)
    int row_mod, row_rem;  // row_mod > 1 (exact mode): only entries whose matrix row max(a,b) lies in a 128-tile row owned by this rank
    // adjoint source, dense mode (G != null):  g = w * (G[a][b] + rcoef * ru[a] * rw[b])
    const double* G;
    int64_t ldg;
    const double* ru;      // may be null (no rank-1 term)
    const double* rw;
    double rcoef;
    int sym;               // dense mode: 1 = lower tiles of a symmetric adjoint (weights 2 / 1 on the diagonal / 0 above), 0 = weight 1
    // per-point input gradients (dense mode only, null = skip): gzr[d][row] += sum_b g dK_ab/dx_a,d ; gzc[d][col] -= ...
    // Every tile stores its 64 row sums / column sums in a slot of its own (gzp), a second kernel adds the slots of a block in fixed order:
    // bit-reproducible (round 2 used fp64 atomics here, and the Titsias d/dZ changed from run to run).
    double* gzr;
    double* gzc;
    int64_t ldgz;
    double* gzp;           // scratch, gz_scratch_doubles(nrb, ncb, D) doubles
    int nrb, ncb;          // 64-point blocks of the row / column inputs (tile_blocks)
    const int* rblk;       // device [nrb][2]: first point and number of points of each row block;
    const int* cblk;       //        [ncb][2]: the same for the column blocks
    double* partial;       // [ntiles][T][W] per-tile partial moments (reduced in fixed order afterwards)
    hipEvent_t ev0, ev1;   // when non-null: recorded around the tile kernel alone (profiling)
    int tab_lds;           // set by the launcher: the term table is copied to LDS
    int phases_ready;      // the phase workspace still holds this table's phases and block centres (the Gram launch of the same evaluation filled it)
};

int launch_gram(const GramArgs& a, int ntiles, hipStream_t s);
// split a tile list into runs of at most `maxrun` full interior tiles (same pair and row block, consecutive columns) and the rest
void split_strip_tiles(const std::vector<GTile>& tiles, int maxrun, std::vector<GSeg>& segs, std::vector<GTile>& rest);
int launch_moments(const MomentArgs& a, hipStream_t s);
inline size_t gz_scratch_doubles(int nrb, int ncb, int D) { return (size_t)2 * nrb * ncb * D * MOGP_GT; }
// [first point, number of points] of every 64-point block, channel by channel: the enumeration GTile::rb / cb refers to
void tile_blocks(const std::vector<int>& off, int C, std::vector<int>& blk);
// moments[P][T][W] += fixed-order sum of per-tile partials; tile_pair_lower[t] = p index, tiles grouped by pair
int launch_moment_reduce(const double* partial, const int* pair_start, int npairs, int T, int W, int D, double* out, hipStream_t s,
                         int lower_pairs = 1);
// per-channel sum of G_kk = 1/2(alpha_k^2 - kinv_kk): out[c], chan_off device array [C+1]
int launch_diagG(const double* kinv, int64_t ld, const double* alpha, const int* chan_off, int C, double* out, hipStream_t s,
                 double kinv_sign = 1.0, int row_mod = 0, int row_rem = 0);

// ---- dense linear algebra (fp64, MFMA) -------------------------------------------------------------
enum GemmMode { GM_RECT = 0,      // mt x nt tiles, k in [0, K)
                GM_LOWER = 1,     // lower tiles of an mt x mt grid (ti >= tj), k in [0, K)
                GM_LAUUM = 2,     // lower tiles, k in [ti*128, K)            (C = W^T W with W lower triangular)
                GM_KHI_J = 3,     // RECT, k in [0, (tj+1)*128)               (B lower triangular in [j][k] layout)
                GM_TASKS = 4,     // explicit task list
                GM_RECT_LOWER = 5,  // mt x nt tiles, only tiles with ti >= tj (others exit), k in [0, K)
                GM_KHI_I = 6,     // RECT, k in [0, (ti+1)*TM)                 (A lower triangular in [i][k] layout)
                GM_KLO_J = 7,     // RECT, k in [tj*TN, K)                     (B lower triangular in [k][j] layout)
                GM_KLO_I = 8 };   // RECT, k in [ti*TM, K)                     (A = W^T with W lower triangular, k-major)

struct GemmTask {              // element offsets relative to the launch's base pointers
    int64_t a_off, b_off, c_off;
    int kt;                    // number of 16-wide k blocks
    int pad;                   // 0, or the task's tile row + 1 (GemmArgs::beta0_from applies to task lists that carry it)
};

struct GemmArgs {
    const double* A; int64_t lda; int a_kmajor;   // a_kmajor 0: A[i*lda + k]   1: A[k*lda + i]
    const double* B; int64_t ldb; int b_kmajor;   // b_kmajor 0: B[j*ldb + k]   1: B[k*ldb + j]
    double* C; int64_t ldc;
    double alpha, beta;                            // C = alpha * A.B^T(+layout) + beta * C
    int mode, mt, nt, K;
    const GemmTask* tasks; int ntasks;
    int task_chunked;                              // GM_TASKS: the list is equal-cost tiles in row-major order -- give each XCD a contiguous chunk of it
    int small;                                     // 0: 128x128 tiles; 1: 64x128; 2: 64x64 (mt / nt count tiles of that shape)
    int beta0_from;                                // > 0: tile rows ti >= beta0_from - 1 are written with beta = 0 (fresh rows of an accumulator: no memset)
    int ksplit; int64_t c_split;                   // ksplit > 1: the k range is cut into ksplit slices, slice s accumulates into C + s * c_split (not with GM_TASKS)
    int ksplit_xcd;                                // ksplit a multiple of 8: slice s runs on XCD s mod 8 only (workgroup b -> XCD b mod 8): the 64 workgroups an XCD holds share ONE k window
    int col_major;                                 // GM_RECT: tiles numbered down the columns (the concurrent workgroups of an XCD share B's column panels; A must be cache-resident)
    int row_mod, row_rem, row_off, row_shift;      // row_mod > 1: only tile rows with ((ti >> row_shift) + row_off) % row_mod == row_rem
                                                   // (sharded evaluation; row_shift = 1 when the launch uses 64-row tiles: ownership is per 128 rows)
    // stream-K form (linalg.hip:k_gemm_sk): sk_spans > 0 = launch that many workgroups, each an equal share of the launch's k iterations
    int sk_hint;                                   // the caller asks for the stream-K form when the tile count sits badly on the slots (a launch that has its stream's CUs to itself)
    int sk_spans;
    double* sk_ws;                                 // sk_spans slots of 128 x 128 doubles (one workspace per stream: launches of a stream do not overlap)
    unsigned* sk_flags;                            // one word per span, never reset: a span's flag holds the epoch of the launch that filled its slot
    unsigned sk_epoch;                             // > 0, different from every earlier launch on this workspace
    unsigned long long* sk_info;                   // a hand-off that times out: atomicMin(MOGP_INFO_CHAIN_TIMEOUT)
    // a launch of the small-tile variants INSIDE the dataflow schedule (flow.hip: the two products between chain kernels run as launches on the
    // private stream, their operands and results shared with the resident dataflow kernel through its counters): every workgroup first waits
    // until fl_flags[fl_widx[k]] >= fl_wval[k] for k < fl_nwait (then ONE agent acquire), stores C write-through when fl_wt, and when fl_sig
    // bumps fl_flags[fl_sig_base + (tile row >> fl_sig_shift)] once its tile has left the CU.  fl_err: the schedule's error word.
    unsigned* fl_flags; int fl_nwait; unsigned fl_widx[4]; unsigned fl_wval[4];
    int fl_wt, fl_sig; unsigned fl_sig_base; int fl_sig_shift; unsigned* fl_err; unsigned fl_spins; unsigned* fl_diag;
};
int launch_gemm(const GemmArgs& a, hipStream_t s);
double gemm_flops(const GemmArgs& a, const std::vector<GemmTask>* host_tasks);

// leaf kernels on 128x128 tiles
// factor the diagonal tile A[t] = L L^T in place (strict upper part zeroed), write sum log L_kk to logdet[t],
// record the first non-positive pivot (1-based global index) in *info (atomicMin on a value initialised to INT64_MAX)
int launch_potrf_tile(double* A, int64_t ld, int t, double* logdet, unsigned long long* info, hipStream_t s);
// invd[t] = inverse of the lower-triangular diagonal tile t of A (dense 128x128, upper part zero); batch over tiles [t0, t0+nt)
int launch_trtri_tiles(const double* A, int64_t ld, int t0, int nt, double* invd, hipStream_t s);
// copy the batch of inverted diagonal tiles into the diagonal tiles of A
int launch_put_diag_tiles(double* A, int64_t ld, int nt, const double* invd, hipStream_t s);
int launch_wkk(const double* Ablk, int64_t ld, const double* invd, int nk, double* Wk, int64_t ldw, hipStream_t s);
// chain.hip: the whole serial chain of one outer block (tiles t0 .. t0 + nk - 1, nk <= 4) in ONE persistent launch: the leaves, the panels
// and updates inside the block and W_KK = L_KK^-1 (-> Wk, as launch_wkk leaves it).  flags: MOGP_CHAIN_FLAGS zeroed words of this block;
// err: one zeroed word per evaluation.  A hand-off that times out is reported as MOGP_INFO_CHAIN_TIMEOUT through *info.
#define MOGP_CHAIN_FLAGS 32
// 0: below every pivot index (they are 1-based), so that atomicMin lets a time-out WIN over the non-positive pivots the garbage behind it
// produces (round 4: with several processes on one GPU a timed-out dataflow evaluation was reported as "not positive definite" instead of repeated)
#define MOGP_INFO_CHAIN_TIMEOUT 0ull
// flow (optional, flow.hip): the kernel first waits until *wait_flag >= wait_val (its diagonal block has received every update from the
// dataflow kernel), stores W_KK write-through and every workgroup bumps *done_flag when it is through.
struct ChainFlow {
    unsigned* wait_flag; unsigned wait_val;
    unsigned* done_flag;
    int write_through;
    unsigned long long* trace;         // optional [4]: launch, after the wait, end (100 MHz wall clock), spare
    unsigned* diag;                    // optional: the dataflow schedule's diagnostic counters (FLOW_DIAG_*)
};
int launch_chain(double* A, int64_t ld, int t0, int nk, double* invd, double* logdet, unsigned long long* info, long long info_base,
                 double* Wk, int64_t ldw, unsigned* flags, unsigned* err, hipStream_t s, const ChainFlow* flow = nullptr);

// ---- tile dataflow form of the fused factorisation + inversion (flow.hip) --------------------------------------------------------------
#define FLOW_MAXQ 48                   // queues of a plan at most: 2 * 8 compare-and-swap lanes + one lane per other queue fit one wave
#define FLOW_NCAS 1                    // the first queues (by priority) are taken ready-only by compare-and-swap, the others eagerly (flow.hip:k_flow)
#define FLOW_TRACE_W 6
#define FLOW_POST_W 96                 // words per workgroup of the dataflow kernel's post-mortem (FlowArgs::post)
#define FLOW_KEY_STEP 1024             // FlowTask::key = FLOW_KEY_STEP * superstep + position inside it
#define FLOW_NOSIG 0xffffffffu
struct FlowTask {                      // 64 bytes; static per matrix size
    uint16_t ar, ac, br, bc, cr, cc;   // tile coordinates (row, column) of each operand's first element in its buffer
    uint8_t abuf, bbuf, cbuf, var;     // buffers (0 A, 1 L, 2 Wt, 3 Wm, 4 B); var: bits 0-1 operand layout (0: both k-contiguous, 1: B k-major,
                                       // 2: both k-major), bit 2: beta = 0, bit 3: alpha = -1, bit 4: raised wave priority
    uint16_t kt, ndep;                 // 16-wide k blocks; dependencies
    uint32_t dep[4];                   // counter index ...
    uint16_t need[4];                  // ... and the value it must have reached
    uint32_t sig[2];                   // counters bumped when the tile is stored (FLOW_NOSIG: none)
    uint32_t key;                      // position in the sequential algorithm (every dependency has a smaller key; every queue is sorted by it)
    uint32_t pad[2];
};
static_assert(sizeof(FlowTask) == 64, "FlowTask layout");
// Diagnostic words of the dataflow schedule (Spd::flow_diag; they outlive an evaluation; mogp_model_flow_diag reads them):
// deep looks made by idle workgroups of k_flow; deep looks whose returning atomics saw a queue head / a dependency counter that the sc1 loads
// of the same look did not; the same two for the waits of the chain kernels and of the private stream's hooks; the last such dependency
// (flag index, value seen, workgroup | XCC << 16)
enum { FLOW_DIAG_DEEP = 0, FLOW_DIAG_HEAD_STALE = 1, FLOW_DIAG_DEP_STALE = 2, FLOW_DIAG_WAIT_DEEP = 3, FLOW_DIAG_WAIT_STALE = 4, FLOW_DIAG_LAST = 5, FLOW_DIAG_WORDS = 8 };
struct FlowPlan {
    int nb = 0, ob = 0, nouter = 0, nq = 0, rhs_nt = 0;
    bool replay = false;               // the measurement plan (flow.hip: flow_build): the private stream's products are tasks too, the chain kernels' counters preset
    std::vector<FlowTask> tasks;       // queue after queue, queues in priority order
    std::vector<FlowTask> folded;      // (scratch of flow_build)
    int qbase[FLOW_MAXQ] = {0}, qsize[FLOW_MAXQ] = {0};
    int nflags = 0, base_heads = 0, base_err = 0;
    // per outer block, the private stream's three launches: the chain kernel (reports done_idx += expect workgroups), the mini-panel (rows of
    // the next block: waits for t1_nwait counters, bumps t1_sig_base + row by 2 nk in all) and the next-diagonal update (waits for one counter)
    struct Chain { uint32_t done_idx, expect; int t1_nwait; uint32_t t1_widx[4], t1_wval[4], t1_sig_base, t1_sig_per_row; uint32_t t2_widx, t2_wval; };
    std::vector<Chain> chain;
    double flops = 0.0;
};
// rhs_nt > 0: factorisation + forward substitution of rhs_nt tile rows of right-hand sides X L^T = T (the prediction) instead of the inverse
void flow_build(int nb, int ob, FlowPlan& p, int rhs_nt = 0, bool replay = false);
// rows >= N of the padded matrix: identity (lower part)
int launch_pad_identity(double* A, int64_t ld, int64_t N, int64_t Npad, hipStream_t s);
// z = W y (W lower triangular), and partial[blk] = sum z^2 over the block's rows
int launch_trmv_lower(const double* W, int64_t ld, int64_t n, const double* y, double* z, double* zz_partial, hipStream_t s,
                      int row_mod = 0, int row_rem = 0);
// a = W^T z
int launch_trmv_lower_t(const double* W, int64_t ld, int64_t n, const double* z, double* a, hipStream_t s, int row_mod = 0, int row_rem = 0);
// out[r] = sum_k M[r][k] * v[k]   (dense row-major rows x n)
int launch_gemv_rows(const double* M, int64_t ld, int64_t rows, int64_t n, const double* v, double* out, hipStream_t s);
// out[r] = base[r] - sum_k M[r][k]^2
int launch_row_sqnorm_sub(const double* M, int64_t ld, int64_t rows, int64_t n, const double* base, double* out, hipStream_t s);
// out = sign * (tril(A) y + strict_tril(A)^T y): symmetric mat-vec with a lower-stored matrix (scratch: (2 + n/512 + 1) * n doubles)
int launch_symv_lower(const double* A, int64_t ld, int64_t n, const double* y, double* out, double* scratch, double sign, hipStream_t s,
                      int row_mod = 0, int row_rem = 0);
// dst[r][c] = scale * src[r][c]
int launch_copy2d(double* dst, int64_t ldd, const double* src, int64_t lds, int64_t rows, int64_t cols, double scale, hipStream_t s);
// A[i][i] += val for i < n
int launch_add_diag(double* A, int64_t ld, int64_t n, double val, hipStream_t s);
// upper triangle <- transpose of the lower triangle (n multiple of 64)
int launch_symmetrize(double* A, int64_t ld, int64_t n, hipStream_t s);
// out = ca * I - cp * P - cq * Q (full n x n, all leading dimension ld); Q may be null
int launch_combine(double* out, const double* P, const double* Q, int64_t ld, int64_t n, double ca, double cp, double cq, hipStream_t s);
// out[j] = sum_i M[i][j] * v[i]  (v null: sum_i M[i][j]^2),  rows x n dense row-major
int launch_gemv_cols(const double* M, int64_t ld, int64_t rows, int64_t n, const double* v, double* out, double* scratch, hipStream_t s);
int launch_get_diag(const double* A, int64_t ld, int64_t n, double* out, hipStream_t s);
int launch_axpby(int64_t n, double a, const double* x, double b, const double* y, double* out, hipStream_t s);
// out[i] = sum over ks slices of n doubles each (split-K partial results, summed in slice order)
int launch_sum_slices(const double* slices, int64_t n, int ks, double* out, hipStream_t s);
// the pivot word <- "no failure", unless it holds a time-out (MOGP_INFO_CHAIN_TIMEOUT)
int launch_info_rearm(unsigned long long* info, hipStream_t s);
int launch_info_stash(unsigned long long* info, hipStream_t s);      // info[1] <- info[0], then re-arm info[0]
// non-finite scan of the lower triangle: flag[0] |= 1 if NaN seen, |= 2 if Inf seen
int launch_nonfinite_scan(const double* A, int64_t ld, int64_t n, int* flag, hipStream_t s);
int launch_pivot_range(const double* invd, int64_t n, double* out, hipStream_t s);

}  // namespace mogp
