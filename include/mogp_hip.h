/*
 * mogp_hip.h -- C ABI of libmogp_hip.so: the MI355X (gfx950) exact multi-output GP hot path.
 *
 * The reference (GAMES-UChile/mogptk v0.5.1) has no FFI of its own: its hot path is a chain of Python
 * method calls into torch.  Each entry point below replaces the torch work behind one of those seams;
 * the reference file:line it stands in for is cited per function (paths relative to the reference root).
 * INTEGRATION.md shows the ctypes stub a mogptk maintainer would add to bind them.
 *
 * Conventions
 *   - plain C types only; every pointer argument is a HOST pointer owned by the caller for the duration
 *     of the call only (device-pointer accessors are the explicitly named mogp_dev_* functions);
 *   - matrices are row-major fp64; X is (N, 1+D) with the channel id in column 0 exactly as the
 *     reference's kernel format (mogptk/model.py:585-606); arbitrary row order is accepted
 *     (mogptk/gpr/kernel.py:446-452) -- the library keeps a stable channel sort internally;
 *   - every function returns 0 on success, a negative MOGP_E* code otherwise; mogp_last_error() gives
 *     the message of the calling thread's last failure;
 *   - calls are synchronous (the reference syncs once per iteration at mogptk/model.py:529); a model
 *     handle is not re-entrant; distinct handles on distinct devices may be used concurrently.
 */
#ifndef MOGP_HIP_H
#define MOGP_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MOGP_OK            0
#define MOGP_EINVAL       -1   /* bad argument */
#define MOGP_EHIP         -2   /* HIP runtime error (message has the hipError string) */
#define MOGP_ENOTPD       -3   /* Cholesky met a non-positive pivot; *info = 1-based index (LAPACK dpotrf convention) */
#define MOGP_ENONFINITE   -4   /* NaN/Inf met in the Gram matrix (reference prints which: gpr/model.py:249-252) */
#define MOGP_ENODEVICE    -5   /* no gfx950 device visible */

/* width of one row of the spectral term table for input dimension D:
 *   [ A, Psi, V_0..V_{D-1}, M_0..M_{D-1}, Delta_0..Delta_{D-1} ]                                     */
#define MOGP_TERM_WIDTH(D) (2 + 3 * (D))
/* number of gradient moments per (channel pair, term):
 *   [ m0 = sum G*E*cos, m4 = sum G*E*sin, m1_d = sum G*u_d^2*E*cos, m2_d = sum G*u_d*E*cos, m3_d = sum G*u_d*E*sin ] */
#define MOGP_MOMENT_WIDTH(D) (2 + 3 * (D))

typedef struct mogp_ctx mogp_ctx;       /* one per device */
typedef struct mogp_model mogp_model;   /* one per (X, y) training set: owns all device workspaces */

/* ---- library / device ------------------------------------------------------------------------- */
const char* mogp_version(void);
const char* mogp_last_error(void);
int  mogp_device_count(void);
/* replaces gpr/config.py:41-52 (use_gpu): bind a context to HIP device `device`. */
int  mogp_ctx_create(int device, mogp_ctx** out);
int  mogp_ctx_destroy(mogp_ctx* ctx);
int  mogp_ctx_device_name(mogp_ctx* ctx, char* buf, int buflen);

/* ---- model: data resident in HBM -------------------------------------------------------------- */
/* replaces gpr/model.py:89-118 + :418-436 (gpr.Model/Exact.__init__: X, y to the device; the dense eye(N)
 * of :435 is NOT materialised).  X: N x (1+D), y: N.  C = number of channels (kernel.output_dims). */
int  mogp_model_create(mogp_ctx* ctx, int64_t N, int D, int C, const double* X, const double* y, mogp_model** out);
int  mogp_model_destroy(mogp_model* m);
/* replace y (e.g. y - mean(X), gpr/model.py:445-448) */
int  mogp_model_set_y(mogp_model* m, const double* y);

/* Unified spectral term table: every MOSM / SM / CSM channel-pair block is
 *   K_ab = sum_t A exp(-1/2 sum_d V_d u_d^2) cos(2 pi (sum_d M_d u_d + Psi)),  u_d = x_a,d - x_b,d + Delta_d.
 * table: C x C x T x MOGP_TERM_WIDTH(D) doubles, entry [i][j][t] for rows in channel i / columns in channel j.
 * replaces the parameter algebra at gpr/multioutput.py:182-199 (MOSM), gpr/singleoutput.py:596-600 (SM),
 * gpr/multioutput.py:432-448 (CSM) as evaluated by the host; the O(n_i n_j) part runs on the device. */
int  mogp_model_set_terms(mogp_model* m, int T, const double* table);
/* Terms with a Gaussian envelope on the input MIDPOINT (MultiOutputHarmonizableSpectralKernel, gpr/multioutput.py:295-395):
 *   K_ab = sum_t A exp(-1/2 sum_d V_d u_d^2) cos(2 pi (sum_d M_d u_d + Psi)) exp(-1/2 sum_d L_d ((x_a,d + x_b,d)/2 - c_d)^2)
 * rows of width 2 + 5 D = [ A, Psi, V_d, M_d, Delta_d, L_d, c_d ]; width 2 + 3 D is mogp_model_set_terms.  With the wide rows the gradient
 * has two more moments per dimension, [ .., m5_d = sum G a_d^2 E cos, m6_d = sum G a_d E cos ], a_d = (x_a,d + x_b,d)/2 - c_d (every moment
 * array of mogp_exact_eval then has rows of that width), the kernel's diagonal is no longer constant per channel -- supply it with
 * mogp_model_set_point_diag, and pass kss_diag to mogp_exact_predict per TEST POINT (S values, caller order) instead of per channel.
 * The Titsias entry points take enveloped terms the same way (any kernel under any inference, as in the reference): kff_diag per TRAINING
 * point (N values, caller order), kss_diag per test point, moment rows of width 2 + 5 D; the derivative with respect to the inducing inputs
 * includes the envelope's.  So do the Snelson and Hensman entry points (kff_diag / kss_diag per point; mogp_snelson_eval returns dp/dKff_nn per
 * point); their data-parallel forms and the Opper-Archambeau entry points do not. */
int  mogp_model_set_terms_ex(mogp_model* m, int T, int width, const double* table);
/* K_diag(X) of the N training points in the caller's row order (reference gpr/kernel.py:483-495): enters the relative jitter
 * jitter * mean(diag) (gpr/model.py:244).  NULL: back to the per-channel constant implied by the table. */
int  mogp_model_set_point_diag(mogp_model* m, const double* kdiag);

/* replaces MultiOutputKernel.K (gpr/kernel.py:446-481); stateless, needs only a context and a term table
 * (C x C x T x MOGP_TERM_WIDTH(D)).  X1 is M1 x (1+D).  X2 == NULL: symmetric Gram K(X1), K_out is M1 x M1
 * (lower channel pairs + mirror, :458-467).  Otherwise X2 is M2 x (1+D) and K_out is M1 x M2
 * (all C*C pairs, no symmetry, :468-479).  Rows may be in any order. */
int  mogp_gram(mogp_ctx* ctx, int C, int D, int T, const double* table,
               int64_t M1, const double* X1, int64_t M2, const double* X2, double* K_out);
/* the same with rows of `width` = 2 + 3 D or 2 + 5 D (envelope, see mogp_model_set_terms_ex) */
int  mogp_gram_ex(mogp_ctx* ctx, int C, int D, int T, int width, const double* table,
                  int64_t M1, const double* X1, int64_t M2, const double* X2, double* K_out);

/* flags for mogp_exact_eval */
#define MOGP_EVAL_GRAD   1   /* also compute the gradient moments */

/* replaces Exact.log_marginal_likelihood (gpr/model.py:438-453) and, with MOGP_EVAL_GRAD, the backward pass
 * gpr.Model.loss() obtains from autograd (gpr/model.py:279-292):
 *   Kj = K + diag(noise_var[c(k)]) (+ data_var[k]) + jitter * mean(diag) * I   (gpr/model.py:440-442, :244)
 *   L = chol(Kj); lml = -N/2 log 2pi - sum log L_kk - 1/2 y^T Kj^-1 y
 *   G = 1/2 (alpha alpha^T - Kj^-1);  moments[i>=j][t][:] over the FULL symmetric matrix (off-diagonal
 *   channel blocks counted twice, as autograd sees `res[r1[i],r2[j]] = k; res[r1[j],r2[i]] = k.T`, kernel.py:466-467)
 *   diagG[c] = sum_{k in c} G_kk;  *trG = trace(G);  *jitter_abs = jitter * mean(diag) actually added.
 * moments: (C*(C+1)/2) x T x MOGP_MOMENT_WIDTH(D), pair index p = i*(i+1)/2 + j for i >= j.
 * data_var may be NULL.  moments/diagG/trG may be NULL without MOGP_EVAL_GRAD.
 * On MOGP_ENOTPD *info is the 1-based failing pivot (in channel-sorted order). */
int  mogp_exact_eval(mogp_model* m, const double* noise_var, const double* data_var, double jitter, int flags,
                     double* lml, double* moments, double* diagG, double* trG, double* jitter_abs, int64_t* info);

/* replaces Exact.predict_f (gpr/model.py:455-483): mu = Kfs^T Kj^-1 y, var = Kss_diag - colsum((L^-1 Kfs)^2)
 * (full != 0: var is S x S = Kss - v^T v).  Xs: S x (1+D).  mu: S, var: S or S*S.  No noise added to var.
 * kss_diag[C]: the kernel's K_diag value per channel as the reference returns it (gpr/kernel.py:483-495; for SM
 * with D > 1 the reference's K_diag, singleoutput.py:602-605, differs from diag K and is reproduced as is). */
int  mogp_exact_predict(mogp_model* m, const double* noise_var, const double* data_var, double jitter,
                        const double* kss_diag, int64_t S, const double* Xs, int full,
                        double* mu, double* var, int64_t* info);

/* ---- Titsias sparse variational bound (BASELINE.json configs[4]) ------------------------------------------------ */
/* replaces Titsias.elbo (gpr/model.py:700-724) and, with MOGP_EVAL_GRAD, the autograd backward of gpr.Model.loss():
 *   Z: M x (1+D) inducing inputs (channel id in column 0), sigma: SCALAR noise scale (gpr/model.py:686-689),
 *   kff_diag[C]: the kernel's K_diag value per channel (as for mogp_exact_predict); with enveloped terms (mogp_model_set_terms_ex,
 *   width 2 + 5 D) kff_diag[N]: K_diag per training point, and every moment row has that width.
 *   *elbo as gpr/model.py:718-723.  Gradient outputs:
 *   mom_uu: (C(C+1)/2) x T x MOGP_MOMENT_WIDTH(D) moments of dELBO/dKuu (symmetric double count, as mogp_exact_eval),
 *   mom_uf: (C*C) x T x MOGP_MOMENT_WIDTH(D) moments of dELBO/dKuf, pair index i*C + j (i: inducing channel, j: data channel),
 *   gZ: M x D  dELBO/dZ[:,1:] in the caller's row order, *trGA = trace(dELBO/dKuu_jittered) (for the relative jitter),
 *   *dsigma = dELBO/dsigma, *jitter_abs = jitter * mean(diag Kuu). */
int  mogp_titsias_eval(mogp_model* m, int64_t M, const double* Z, double sigma, double jitter, const double* kff_diag, int flags,
                       double* elbo, double* mom_uu, double* mom_uf, double* gZ, double* trGA, double* dsigma,
                       double* jitter_abs, int64_t* info);
/* what the last mogp_titsias_eval with MOGP_EVAL_GRAD left on the device, in the device's CHANNEL-SORTED order of Z's and X's rows (identical to
 * the caller's when both came sorted by channel), padding removed (M: the rows of that call's Z) -- numerics diagnostics (tools/titsias_stage_errors.py), no reference seam:
 *   which 0: dELBO/dKuu_jittered (M x M; its lower triangle is what the moment pass reads), 1: dELBO/dKuf WITHOUT its rank-one part (M x N),
 *         2: beta (M), 3: r (N)  [the rank-one part is beta r^T], 4: v = L^-1 Kuf (M x N), 5: L, the lower Cholesky factor of Kuu + jitter (M x M),
 *         6: Qs = v v^T / sigma^2 + I (M x M, symmetric), 7: its inverse Pq (M x M, symmetric), 8: t1 = Pq v y (M). */
int  mogp_titsias_fetch(mogp_model* m, int which, int64_t M, double* out);
/* replaces Titsias.predict_f (gpr/model.py:730-765), diagonal variance: mu[S], var[S]. */
int  mogp_titsias_predict(mogp_model* m, int64_t M, const double* Z, double sigma, double jitter, const double* kss_diag,
                          int64_t S, const double* Xs, double* mu, double* var, int64_t* info);
/* The same two calls DATA-PARALLEL over the ranks of the context's communicator (mogp_comm_init_*; new design, the reference has no
 * multi-device path): each rank's handle was created on ITS OWN SHARD of the training points (any split; Z, sigma, the terms and the
 * arguments are the same on every rank).  The sparse bound touches the data only through sums over points: K_uf and v = L^-1 K_uf are built
 * for the local columns, v v^T (M x M), v y, y^T y, N and sum K_ff,nn are all-reduced (M^2 + M + 3 doubles), every M x M stage then runs
 * replicated, the adjoint of K_uf and its moments are local again and the (C C T W) moments and the D x M inducing-input gradient of that
 * part are all-reduced.  Every rank returns the FULL model's ELBO and gradient (mom_uf / gZ already summed over ranks).  The O(N M^2) work
 * is divided by the number of ranks; the O(M^3) work is not.  With one rank (or no communicator) they equal the plain calls. */
int  mogp_titsias_eval_sharded(mogp_model* m, int64_t M, const double* Z, double sigma, double jitter, const double* kff_diag, int flags,
                               double* elbo, double* mom_uu, double* mom_uf, double* gZ, double* trGA, double* dsigma,
                               double* jitter_abs, int64_t* info);
int  mogp_titsias_predict_sharded(mogp_model* m, int64_t M, const double* Z, double sigma, double jitter, const double* kss_diag,
                                  int64_t S, const double* Xs, double* mu, double* var, int64_t* info);

/* ---- sharded exact evaluation / prediction across the GPUs of one node -------------------------------------------
 * The reference has no distributed code at all (no torch.distributed / NCCL call site: SURVEY.md section 5); this is new design behind the
 * same seams as mogp_exact_eval / mogp_exact_predict (gpr/model.py:438-453, :279-292, :455-483).
 * One process per GPU, each with its own context and a model handle on the FULL training set (X, y are O(N)).  Rank r owns the 128-row
 * tile rows i with i % nranks == r: it BUILDS only the Gram / moment tiles of those rows, holds only those rows of the work matrix current,
 * and applies the rank-512 updates of the single-sweep inversion (sweep.hip) to them alone -- the O(N^3) work is divided by nranks.
 * Per 512-wide pivot block ONE all-gather (the block's column panel from the owners of its tile rows + the left part of the pivot rows);
 * once per evaluation an all-reduce of alpha (Npad doubles) and of the gradient moments / diagonal sums (a few kB).  The collectives are
 * issued by the library itself on its critical HIP stream: with RCCL nothing returns to the host between pack, exchange, unpack and the
 * block's arithmetic, and the previous block's trailing update keeps running on the bulk stream underneath the exchange.
 *
 * Communicator (per context).  RCCL: rank 0 calls mogp_comm_unique_id, the caller distributes the 128 bytes (any transport), every rank
 * calls mogp_comm_init_rccl.  External: two callbacks on DEVICE pointers (counts in doubles; the library drains its stream before each
 * call and the callback returns when the result is in place): for transports without a device path and for tests with several ranks on
 * one GPU, which RCCL refuses. */
#define MOGP_COMM_ID_BYTES 128
typedef int (*mogp_allgather_cb)(void* user, const void* send_dev, void* recv_dev, int64_t count);   /* recv holds nranks * count */
typedef int (*mogp_allreduce_cb)(void* user, void* buf_dev, int64_t count);                          /* sum over ranks, in place */
int  mogp_comm_unique_id(void* id128);
int  mogp_comm_init_rccl(mogp_ctx* ctx, const void* id128, int rank, int nranks);
int  mogp_comm_init_external(mogp_ctx* ctx, int rank, int nranks, mogp_allgather_cb allgather, mogp_allreduce_cb allreduce, void* user);
int  mogp_comm_destroy(mogp_ctx* ctx);
/* kind: 0 none, 1 RCCL, 2 external */
int  mogp_comm_info(mogp_ctx* ctx, int* kind, int* rank, int* nranks);
/* One all-reduce ISSUED BY THE LIBRARY over the context's communicator (the call every sharded evaluation makes): every rank contributes
 * (1, rank + 1); *ranks_seen = the number of ranks that took part, *rank_sum = n (n + 1) / 2 when they are the ranks 0 .. n-1.  A context
 * without a communicator reports (1, 1).  What bench.py prints as `rccl_ranks` (there is no reference counterpart: the reference has no
 * multi-GPU path). */
int  mogp_comm_selftest(mogp_ctx* ctx, int* ranks_seen, int* rank_sum);
/* HIP-event times (ms, summed over the pivot blocks) of the last mogp_exact_eval_sharded on this model with profiling on (mogp_set_profiling), SIX numbers:
 * ms[0] exchange on the critical stream (pack, all-gather, unpack -- with the split exchange of round 5 only the pivot block's own rows: 2 MB),
 * ms[1] the part every rank repeats (Schur block inversion and panels), ms[2] the update of the next pivot block's columns (critical stream),
 * ms[3] the rank's share of the bulk update (bulk stream, overlaps the others), ms[4] the rest of the panel on the communication stream (all-gather
 * + unpack, underneath the inversion), ms[5] how long the critical stream then still waited for it (0 when MOGP_SHARD_SPLIT=0). */
int  mogp_shard_stage_ms(mogp_model* m, double* ms);
/* Fraction of the lower 128 x 128 tiles of Kj^-1 the last mogp_exact_eval(..., MOGP_EVAL_GRAD) formed: the gradient
 * 1/2 sum_ab (alpha_a alpha_b - Kinv_ab) dK_ab/dtheta (reference gpr/model.py:291, autograd through :242-246) reads Kinv only where some term
 * of dK/dtheta is above e^-50 of its peak, and for stationary kernels on a series much longer than their support that is a band of tiles.
 * 1.0 = all of them (always so above 0.85, with MOGP_FULL_INVERSE=1, on the sweep schedule and sharded).  mogp_model_fetch(which = 1)
 * completes the inverse on demand. */
int  mogp_model_inverse_fraction(mogp_model* m, double* fraction);
/* Smallest and largest diagonal entry of the Cholesky factor of Kj from the last mogp_exact_eval / mogp_exact_predict (0, 0 when that call was a
 * sweep or sharded evaluation): (lmax / lmin)^2 is a cheap lower estimate of cond(Kj).  The exact path forms its panels with explicit block
 * inverses and stays within the reference's tolerances (LML 1e-9, gradients 1e-5) up to cond(Kj) ~ 1e6 - 1e7; the host side
 * (mogptk_amd.gpr.Exact) warns beyond.  No reference seam: torch.linalg.cholesky (gpr/model.py:246) is backward stable and says nothing. */
int  mogp_model_pivot_range(mogp_model* m, double* lmin, double* lmax);
/* on != 0: mogp_exact_eval evaluates in its backward-stable form -- launch-per-step Cholesky with refined panels, Kj^-1 = L^-T (L^-1 I) and
 * alpha by blocked substitution instead of products with explicit block inverses -- about four times slower (43 ms against 10 at N = 8192), and within the reference's
 * tolerances where the default schedules are not (cond(Kj) >~ 1e7).  gpr.Exact switches it on when the pivot range says so.
 * mogp_exact_predict follows: refined panels, and every solved block column of the forward substitution refined once against L_KK.
 * Not for the sweep / sharded evaluation.  No reference seam (the reference's LAPACK calls are this form already). */
int  mogp_model_set_accurate(mogp_model* m, int on);
/* mogp_exact_eval(..., MOGP_EVAL_GRAD) sharded over the context's communicator: same outputs, identical on every rank. */
int  mogp_exact_eval_sharded(mogp_model* m, const double* noise_var, const double* data_var, double jitter,
                             double* lml, double* moments, double* diagG, double* trG, double* jitter_abs, int64_t* info);
/* mogp_exact_predict (diagonal variance) sharded: the inversion as above; every rank builds the test Gram K_sf for all test points and forms, from the tile rows
 * of Kj^-1 it OWNS, its share of the quadratic form K_s. Kj^-1 K_.s (strictly lower tiles twice, diagonal tiles once); ONE all-reduce of S doubles makes the
 * variances, the mean is K_sf alpha with the all-reduced alpha.  No rank ever holds more of Kj^-1 than its own rows (rounds 3 - 5 all-gathered the whole inverse:
 * N^2 doubles over the links and on every rank).  Same mu / var on every rank. */
int  mogp_exact_predict_sharded(mogp_model* m, const double* noise_var, const double* data_var, double jitter,
                                const double* kss_diag, int64_t S, const double* Xs, double* mu, double* var, int64_t* info);

/* The stages of the sharded evaluation one by one, for a caller that issues the collectives itself (the numpy twin of this protocol in
 * oracle/table_model.py runs under gloo on CPU through exactly this sequence -- mogptk_amd/dist.py:sharded_eval):
 *   config -> begin -> for kb in 0..nblocks-1: { pack; ALL-GATHER(send -> recv, count per rank); unpack; block }
 *   alpha -> ALL-REDUCE(vec, count, sum) -> finish -> ALL-REDUCE(moments), ALL-REDUCE(diagG) on the host arrays.
 * Pointers returned through void** are DEVICE pointers owned by the handle, counts are in doubles. */
int  mogp_shard_config(mogp_model* m, int rank, int nranks);
/* How much of the N x N work matrix this handle holds physical memory for.  A model whose FIRST evaluation is a sharded one over P > 1 ranks gets the
 * matrix in the owned-rows form: the address range whole, memory only under the 128-row tile rows i with i % P == rank (block-cyclic
 * ownership) -- ceil(T / P) of T tile rows, and no second matrix -- so P ranks hold an N that one GPU could not.  The first one-GPU call on such a
 * handle (mogp_exact_eval, mogp_exact_predict, mogp_oa_forward) backs the rest.  backed = bytes under the matrix now, whole = 8 Npad^2.
 * MOGP_SHARD_OWNED=0 in the environment keeps the replicated-matrix form of rounds 1-5.  No reference seam (mogptk is single-device). */
int  mogp_model_work_bytes(mogp_model* m, int64_t* backed, int64_t* whole);
int  mogp_shard_begin(mogp_model* m, const double* noise_var, const double* data_var, double jitter, double* jitter_abs, int* nblocks);
int  mogp_shard_pack(mogp_model* m, int kb, void** send, void** recv, int64_t* count);
int  mogp_shard_unpack(mogp_model* m, int kb);
int  mogp_shard_block(mogp_model* m, int kb);
int  mogp_shard_alpha(mogp_model* m, void** vec, int64_t* count);
int  mogp_shard_finish(mogp_model* m, double* lml, double* moments, double* diagG, int64_t* info);
/* plain device <-> host copy of `bytes` bytes (host staging when the collective backend has no device path) */
int  mogp_dev_copy(void* dst, const void* src, int64_t bytes, int to_device);

/* ---- measurement ------------------------------------------------------------------------------- */
/* stage ids for mogp_stage_ms (HIP-event time of the last mogp_exact_eval on this model's stream) */
enum { MOGP_ST_GRAM = 0, MOGP_ST_POTRF = 1, MOGP_ST_TRTRI = 2, MOGP_ST_LAUUM = 3, MOGP_ST_SOLVE = 4,
       MOGP_ST_MOMENTS = 5, MOGP_ST_TOTAL = 6, MOGP_ST_GEMM_KERNEL = 7,
       MOGP_ST_GRAM_KERNEL = 8, MOGP_ST_MOMENT_KERNEL = 9,   /* the Gram / moment kernel alone (the stage also has the phase-table pre-pass) */
       MOGP_ST_COUNT = 10 };
int  mogp_set_profiling(mogp_model* m, int on);
/* ms[MOGP_ST_COUNT]; MOGP_ST_GEMM_KERNEL = summed duration of every launch of the fp64 MFMA GEMM kernel;
 * *gemm_launches / *gemm_flops = their count and algorithmic flop total in the last eval. */
int  mogp_stage_ms(mogp_model* m, double* ms, int64_t* gemm_launches, double* gemm_flops);

/* How the last mogp_exact_eval(MOGP_EVAL_GRAD) of this model was scheduled -- so that a silent degradation is visible (bench.py prints it,
 * the GPU tests assert it).  The factorisation + inversion (reference gpr/model.py:242-246, :291) runs as ONE resident dataflow kernel fed by
 * persistent chain kernels; if a hand-off inside them times out (their workgroups not all resident: another process on the same GPU) the
 * evaluation is repeated on streams of launches and the model stays there for a while (below).  *flags: OR of the bits below. */
#define MOGP_SCHED_DATAFLOW            1   /* the last fused factorisation + inversion ran as tile dataflow (csrc/flow.hip) */
#define MOGP_SCHED_CHAIN_KERNEL        2   /* the persistent chain kernel (csrc/chain.hip) is in use */
#define MOGP_SCHED_DATAFLOW_FELL_BACK  4   /* a dataflow evaluation timed out and the model is on the stream schedule at the moment: for 64 evaluations after the first
                                            * time-out, four times as many after each further one (up to 16384), then the dataflow kernel gets another try */
#define MOGP_SCHED_TIMEOUTS_SHIFT      8   /* bits 8 .. 23: how many dataflow evaluations of this model have timed out so far */
#define MOGP_SCHED_CHAIN_FELL_BACK     8   /* a chain kernel timed out once: launch-per-step chain from then on */
int  mogp_model_schedule(mogp_model* m, int* flags);

/* Measurement mode for the dataflow kernel (no counterpart in the reference: it has no kernels to profile).  A counter-collecting profiler
 * (rocprofv3 --pmc) serialises dispatches, and the dataflow kernel normally CO-OPERATES with the chain kernels that factor the diagonal blocks,
 * so under such a profiler an evaluation falls back to streams of launches and the kernel that does the N^3 work is never counted.  With
 * on = 1, the following mogp_exact_eval(MOGP_EVAL_GRAD) calls of this model launch the dataflow kernel ALONE on a replay plan: the chain
 * kernels' counters are preset, their results (the W_KK = L_KK^-1 blocks) are the ones the last completed evaluation left in place, and the two
 * products the private stream normally runs are ordinary tile tasks.  Same Gram, same tile products, same inverse (mogp_model_fetch(which = 1)
 * returns the same Kj^-1 -- tools/flow_replay.py checks it); the log-determinant part of the returned LML is NOT recomputed, so the value is
 * the previous evaluation's: use the mode for measurement only.  Requires a completed gradient evaluation on this model.  on = 0: back to normal. */
int  mogp_model_flow_replay(mogp_model* m, int on);

/* Diagnostic counters of the dataflow schedule, summed over every evaluation of this model so far (no counterpart in the reference).  A workgroup
 * (or a chain kernel / private-stream hook) that has waited for a few ms re-reads what it waits for with RETURNING read-modify-writes -- a "deep"
 * look -- next to the sc1 loads every look uses, and counts the answers that differ.  out[0] deep looks of the dataflow kernel, out[1] of them with a
 * different queue head, out[2] with dependency counters the loads called unmet and the atomics met; out[3] deep polls of the other kernels' waits,
 * out[4] of them that ended the wait; out[5..7] the last such dependency (counter index, value, workgroup | XCC << 16).
 * A difference is not by itself a stale read: the two reads are 1 - 2 us apart and counters move (25 differences in 2129 deep looks of a healthy 100 000-evaluation
 * soak); what round 6 used the counters for is the opposite finding -- during a stalled evaluation, where nothing moves, 0 differences in 1.5 M deep looks. */
int  mogp_model_flow_diag(mogp_model* m, unsigned* out8);

/* The fused factorisation + inversion behind mogp_exact_eval(MOGP_EVAL_GRAD) (reference gpr/model.py:242-246 and the O(N^3) solves of its
 * autograd backward, :291) runs as a static graph of 128 x 128 tile products inside ONE resident kernel (csrc/flow.hip).  This call returns
 * that graph for a matrix of nb tile rows as numbers -- no device work, callable without a GPU -- so that tests can replay it on the CPU
 * (dependencies sufficient, no deadlock, no data race, result = the inverse).  *count = rows; out (cap >= 24 * count, or NULL to ask for
 * the count) gets 24 numbers per row:
 *   tile task:    [queue, key, A buffer, A tile row, A tile col, B buffer, row, col, C buffer, row, col, var, k blocks of 16, ndep,
 *                  dep counter x 4, needed value x 4, counter bumped x 2 (-1: none)]
 *                 buffers 0 Schur matrix, 1 panels L, 2 running product Wt, 3 W = L^-1, 4 inverse; var bits 0-1: 0 = C (+)= a A B^T (both
 *                 k-contiguous), 1 = a A B (B k-major), 2 = a A^T B (both k-major); bit 2: overwrite (beta = 0); bit 3: a = -1
 *   launches of the private stream, in stream order (one at a time), three per outer block:
 *     chain kernel          [-1, key, block, first tile k0, tiles nk, 0 .., counter bumped, by how much]                 (columns 22, 23)
 *     mini-panel            [-2, key, block, k0, nk, first row tile k1, row tiles na, .., ndep, dep counter x 4, needed value x 4,
 *                            first counter bumped (one per row tile, consecutive), by how much each]   L[k1.., K] = A[k1.., K] W_KK^T
 *     next-diagonal update  [-3, key, block, k0, nk, k1, na, .., ndep, dep counter, .., needed value, .., -1, 0]          A[k1.., k1..] -= P P^T */
int  mogp_flow_plan(int nb, int64_t* out, int64_t cap, int64_t* count);
/* The same for the prediction's schedule: factorisation + forward substitution X L^T = T of rhs_nt tile rows of right-hand sides (no inverse);
 * the right-hand sides T are buffer 2, the solution X buffer 4 in the task rows (reference seam: Exact.predict_f, gpr/model.py:455-483). */
int  mogp_flow_plan_rhs(int nb, int rhs_nt, int64_t* out, int64_t cap, int64_t* count);
/* Time stamps (100 MHz device wall clock) of the last mogp_exact_eval that ran as dataflow with MOGP_FLOW_TRACE=1 in the environment:
 * 6 numbers per tile task in the row order of mogp_flow_plan's tile tasks (the workgroup starts looking for work, has taken the task, its k loop
 * starts, ends, the tile is stored and its counters are bumped, XCC << 16 | workgroup), then 4 per chain kernel
 * (launched, wait over, end, 0).  *count = 0 when there is no trace.  Measurement only (tools/flow_trace.py -> profiles/). */
int  mogp_flow_trace(mogp_model* m, int64_t* out, int64_t cap, int64_t* count);

/* copy device-resident matrices of the last eval back (tests / CholeskyException payload):
 * which: 0 = W = L^-1, inverse of the lower Cholesky factor of Kj, in CHANNEL-SORTED row order (after any eval), 1 = Kj^-1 (after MOGP_EVAL_GRAD),
 *        2 = alpha (N).  Output in the caller's original row order, full N x N (symmetrised / lower-filled). */
int  mogp_model_fetch(mogp_model* m, int which, double* out);

/* The pseudo-input sparse GP of Snelson & Ghahramani (FITC) on the model's data: reference gpr/model.py:516-541
 * (Snelson.log_marginal_likelihood) + the autograd backward through it, and :543-576 (Snelson.predict_f).
 * Z: M x (1 + D) inducing inputs in kernel format (channel id first); noise_var[C] = sigma_c^2; kff_diag[C] / kss_diag[C] = K_diag per channel.
 * Gradient outputs as for mogp_titsias_eval (moments of the adjoints of Kuu and Kuf, d/dZ, tr dp/dKuu for the jitter term), and
 * hsum[C] = sum over the points of a channel of dp/dKff_nn = dp/dsigma_n^2.
 * Term rows of width 2 + 5 D (an envelope on the input midpoint, MOHSM: reference gpr/multioutput.py:340-395 -- any kernel under any inference):
 * the kernel diagonal follows the points, so kff_diag has N entries and kss_diag S entries (per point, caller's order) and hsum receives the
 * N per-point values dp/dKff_nn (caller's order); the moments have 2 + 5 D entries per term.  Not on the data-parallel entry points. */
int  mogp_snelson_eval(mogp_model* m, int64_t M, const double* Z, const double* noise_var, double jitter, const double* kff_diag, int flags,
                       double* lml, double* mom_uu, double* mom_uf, double* gZ, double* trGA, double* hsum, double* jitter_abs, int64_t* info);
int  mogp_snelson_predict(mogp_model* m, int64_t M, const double* Z, const double* noise_var, double jitter, const double* kff_diag,
                          const double* kss_diag, int64_t S, const double* Xs, double* mu, double* var, int64_t* info);
/* The same two calls DATA-PARALLEL (cf. mogp_titsias_eval_sharded below): each rank's handle holds its own shard of the training points;
 * v G v^T and (v diag h) v^T (M x M), v G y, sum log g, y^T G y, N, the (Z, X) moments with their share of d/dZ and hsum are all-reduced
 * inside the library; every rank returns the full model's value and gradient. */
int  mogp_snelson_eval_sharded(mogp_model* m, int64_t M, const double* Z, const double* noise_var, double jitter, const double* kff_diag,
                               int flags, double* lml, double* mom_uu, double* mom_uf, double* gZ, double* trGA, double* hsum,
                               double* jitter_abs, int64_t* info);
int  mogp_snelson_predict_sharded(mogp_model* m, int64_t M, const double* Z, const double* noise_var, double jitter, const double* kff_diag,
                                  const double* kss_diag, int64_t S, const double* Xs, double* mu, double* var, int64_t* info);

/* The variational GP of Hensman et al. (whitened; reference gpr/model.py:767-886), the O(N M^2) algebra around ANY likelihood, in two calls
 * per evaluation.  forward: mean and variance of q(f) at the training inputs (S = 0; the state for the backward call is kept in the handle)
 * or at S test inputs -- SparseHensman._predict_f (:851-868); q_mu[M], q_sqrt[M x M] (its lower triangle is used) in the order of Z's rows;
 * dense != 0: the non-sparse model at its own inputs (:834-840; M = N, Z = the data points).  The caller evaluates its likelihood's
 * expectation E(mu, var) and hands e = dE/dmu, f = dE/dvar (per training point, caller order) to backward, which returns the moments of
 * the adjoints of Kuu / Kuf (as mogp_titsias_eval), d/dZ, tr dE/dKuu (jitter term), dE/dq_mu[M] and dE/dS[M x M] (take its lower triangle). */
int  mogp_svgp_forward(mogp_model* m, int64_t M, const double* Z, const double* q_mu, const double* q_sqrt, double jitter,
                       const double* kff_diag, int dense, int64_t S, const double* Xs, const double* kss_diag,
                       double* mu, double* var, double* jitter_abs, int64_t* info);
int  mogp_svgp_backward(mogp_model* m, const double* e, const double* f, double* mom_uu, double* mom_uf, double* gZ, double* trGA,
                        double* g_qmu, double* g_qsqrt);
/* The sparse model DATA-PARALLEL (cf. mogp_titsias_eval_sharded): each rank's handle holds its own shard of the training points, the forward
 * call is the plain one (mu / var of the LOCAL points; the caller evaluates its likelihood on them and all-reduces the expectation and the
 * likelihood parameters' gradients, O(1) numbers); this backward call all-reduces what sums over points -- the two M x M products, v e, the
 * (Z, X) moments and their share of d/dZ -- and returns the FULL model's gradients on every rank.  Not for the dense model (M = N). */
int  mogp_svgp_backward_sharded(mogp_model* m, const double* e, const double* f, double* mom_uu, double* mom_uf, double* gZ, double* trGA,
                                double* g_qmu, double* g_qsqrt);

/* The variational Gaussian approximation of Opper & Archambeau at the data points (reference gpr/model.py:578-668, OpperArchambeau):
 * q(f) = N(K nu, (K^-1 + diag(lambda^2))^-1), one nu and one lambda > 0 per data point (caller order, like y).  Like the Hensman
 * pair, the likelihood sits between two calls.
 *   forward:   mu[N] = K nu, var[N] = diag of the covariance of q(f), *kl = nu^T K nu + log det B + tr B^-1 - N with
 *              B = Lambda K Lambda + I (the reference's `kl`: ELBO = E(mu, var) - kl / 2).  No jitter, as in the reference (:613).
 *   backward:  e = dE/dmu, f = dE/dvar per point -> the gradient of E - kl / 2: moments[P][T][2+3D] of the adjoint of K (contract with
 *              the term table's derivatives like mogp_exact_eval's), g_nu[N], g_lambda[N].  Consumes the forward call's state.
 *   predict:   OpperArchambeau.predict_f (:640-668): mu = K_sf nu, var = K_ss - K_sf (K + diag(1 / lambda^2))^-1 K_fs (full: S x S). */
int  mogp_oa_forward(mogp_model* m, const double* q_nu, const double* q_lambda, double* mu, double* var, double* kl, int64_t* info);
int  mogp_oa_backward(mogp_model* m, const double* e, const double* f, double* moments, double* g_nu, double* g_lambda);
int  mogp_oa_predict(mogp_model* m, const double* q_nu, const double* q_lambda, const double* kss_diag, int64_t S, const double* Xs,
                     int full, double* mu, double* var, int64_t* info);

/* Full predictive covariance of the sparse models (Titsias.predict_f / SparseHensman._predict_f with full=True, reference
 * gpr/model.py:758-760, 870-872): K_ss - a^T a + b^T b for the S test points of the LAST mogp_titsias_predict(_sharded) or
 * mogp_svgp_forward(S > 0) call on this handle (a = L^-1 K_us and b stay on the device), cov[S][S] in the caller's order. */
int  mogp_sparse_predict_cov(mogp_model* m, int64_t S, double* cov);

/* Host-side pair algebra of the MOSM kernel in native code (no device work): the cross-spectral term table of every channel pair
 * (reference gpr/multioutput.py:178-204) and the reverse-mode gradient autograd takes through it.
 * w (C,Q), mu / v / th (C,Q,D), ph (C,Q) are the CONSTRAINED weight, mean, variance, delay, phase; table is [C][C][Q][2+3D];
 * gtable (same shape) is zero for channel pairs i < j and already carries the double count of the off-diagonal pairs. */
int  mogp_mosm_terms(int C, int Q, int D, const double* w, const double* mu, const double* v, const double* th, const double* ph,
                     double twopi, double phase_scale, double* table);
int  mogp_mosm_terms_backward(int C, int Q, int D, const double* w, const double* mu, const double* v, const double* th, const double* ph,
                              double twopi, double phase_scale, const double* gtable, double* gw, double* gmu, double* gv, double* gth,
                              double* gph);

#ifdef __cplusplus
}
#endif
#endif /* MOGP_HIP_H */
