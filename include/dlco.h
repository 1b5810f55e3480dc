/*
 * dlco.h — C ABI of libdlco.so: the MI355X (gfx950) implementation of opencv-dlco's
 * projection-learning hot path (pj-learn).
 *
 * The reference (cbalint13/opencv-dlco) has no plugin or FFI interface: the path is a
 * monolithic main() (src/pj-learn.cpp:57-600) plus the free functions declared in
 * include/trainer.hpp:43-64.  Each entry point below names the reference code it
 * replaces; a maintainer's pj-learn main() would call them in the order shown in
 * INTEGRATION.md.  Plain pointers and sizes only; every buffer is caller-owned and
 * copied by the library unless a function says otherwise.  All functions return
 * DLCO_OK (0) or a negative error code; dlco_last_error() gives the message.  A context
 * is driven by one host thread; distinct contexts are independent.
 *
 * There is no CPU fallback: dlco_ctx_create fails when no gfx950 device is usable.
 */
#ifndef DLCO_H
#define DLCO_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DLCO_OK            0
#define DLCO_ERR_INVALID  -2   /* bad argument / wrong call order          */
#define DLCO_ERR_HIP      -3   /* HIP runtime error (message has details)  */
#define DLCO_ERR_NODEVICE -4   /* no usable gfx950 device                  */
#define DLCO_ERR_NOCONV   -5   /* eigen tracker did not reach its tolerance */
#define DLCO_ERR_COMM     -6   /* the host's collective callback failed      */

typedef struct dlco_ctx dlco_ctx;

/* Hyper-parameters and shapes.  Defaults (dlco_cfg_default) are the reference's
 * hard-coded values, src/pj-learn.cpp:89-96,225. */
typedef struct dlco_cfg {
    int32_t  F;          /* FeatDim: columns of "Distance"                 (:179) */
    int32_t  N;          /* nDists: rows of "Distance"/"Label"             (:178) */
    int32_t  B;          /* szBatch per class, GLOBAL over all ranks       (:93)  */
    float    mu;         /* trace-norm weight                              (:89)  */
    float    gamma;      /* RDA step parameter                             (:90)  */
    uint64_t seed;       /* sampler RNG seed                               (:225) */
    int32_t  device;     /* HIP device ordinal                             (:267) */
    int32_t  rank;       /* data-parallel rank: owns batch slots [rank*B/world, (rank+1)*B/world) */
    int32_t  world;      /* number of ranks (1 = the reference's single device)           */
    float    eig_tol;    /* subspace tracker tolerance on the weighted c*|residual| / max e (default 2e-4; the first
                          * update after a (re)start converges to a quarter of it).  Measured error of A+ against
                          * ssyevr at the default: 1.5e-5 (rank 64) / 3.6e-5 (rank 128) of its largest entry at
                          * F = 8192, i.e. inside SURVEY 8(d)'s 1e-4 gate; 5e-5 costs 6 % / 40 % of the step rate */
    int32_t  eig_guard;  /* guard vectors kept beyond the positive eigenspace (default 32)  */
    int32_t  eig_max_iter; /* filter+Rayleigh-Ritz iterations per step before giving up      */
                          /* The tracker block holds at most min(F, max(1024, 2*B + 2*eig_guard)) rows (positive
                           * rank + guards); a positive eigenspace that does not fit is reported as a
                           * non-converged step, never truncated silently.                              */
    int32_t  shard;      /* world > 1 only.  0: every rank keeps the whole dual average, the F x F
                          * partial gradients are all-reduced (dlco_step_begin/grad/finish).
                          * 1: the dual average is sharded by columns, rank g owns columns
                          * [g*F/world, (g+1)*F/world); no F x F exchange, the step (dlco_step) calls
                          * the all-gather registered with dlco_set_allgather.  F % (128*world) == 0
                          * selects the fused kernels.                                            */
    int32_t  strict_conv; /* 1: dlco_step returns DLCO_ERR_NOCONV when the tracker misses eig_tol (the step is
                          * still applied).  0 (default): the miss is counted (dlco_counters out[2],
                          * dlco_log_entry.nonconv) and the run goes on with the approximate W.            */
    int32_t  grad_bf16;  /* 1: BASELINE configs[4], "bf16 MFMA + fp32 accumulate": every GEMM whose operand is the resident
                          * Distance matrix multiplies on the bf16 matrix cores with fp32 accumulation, operands rounded
                          * to bf16 once when the MFMA fragments are formed - the gradient SYRK (Q1), the projection of
                          * the batch (P1), of the validation rows (T1) and of all N rows in the statistics pass (S2).
                          * Gather, weights, dual average, squared sums, the PSD projection stay fp32.  Not the
                          * reference's arithmetic: the result is gated on the FPR@95 band, not on the fp32
                          * tolerances.  Needs F % 128 == 0.  0 (default): fp32 results (fp32 MFMA, or split-bf16
                          * MFMA carrying all 24 mantissa bits where a pass would otherwise be matrix-bound).      */
    int32_t  reserved[5];
} dlco_cfg;

void dlco_cfg_default(dlco_cfg *cfg);

const char *dlco_version(void);
/* Message of the last error on this thread (ctx may be NULL for create failures). */
const char *dlco_last_error(const dlco_ctx *ctx);

/* Allocates device state; replaces cuda::setDevice(0) and the Mat::zeros block,
 * src/pj-learn.cpp:259-291. */
int dlco_ctx_create(dlco_ctx **out, const dlco_cfg *cfg);
void dlco_ctx_destroy(dlco_ctx *ctx);
/* "Found GPU: <name>" line of the reference log, src/pj-learn.cpp:259-263. */
int dlco_device_name(const dlco_ctx *ctx, char *buf, size_t cap, int *cc_major, int *cc_minor);

/* Uploads Distance [N,F] row-major f32 and Label [N] u8 (1 = match, 0 = non-match) to HBM,
 * builds IdxPos/IdxNeg, shuffles and splits them.  Replaces src/pj-learn.cpp:214-256,277-281. */
int dlco_set_data(dlco_ctx *ctx, const float *dists_host, const uint8_t *labels_host);
/* Same, but `dists_dev` [N,F] (row stride F) already lives in device memory.  When F is a multiple of the
 * device width's granularity (dlco_device_width(ctx) == F) it is adopted, not copied, and must stay valid
 * for the life of the context; otherwise it is copied once into the padded resident layout. */
int dlco_set_data_device(dlco_ctx *ctx, const float *dists_dev, const uint8_t *labels_host);
/* FeatDim is whatever the input file says (src/pj-learn.cpp:176-179: 480, 544, 608 in the reference's own
 * runs).  On the device every row is kept at dlco_device_width(ctx) floats: F rounded up to whole 128-column
 * tiles (to 128*world when the dual average is column-sharded), the extra columns zero.  Zero columns add
 * zero rows and columns to the dual average; their eigenvalue of A = -c (dfAvg + mu I) is -c mu <= 0 and is
 * never kept (src/pj-learn.cpp:452,480-484), so W, A, dfAvg, distances and statistics restricted to the
 * first F columns are those of the unpadded computation.  Every host-side argument and result of this
 * header has the caller's width F; only DLCO_BUF_* device buffers show the padded width. */
int dlco_device_width(const dlco_ctx *ctx);
/* The resident matrix (row or pair mode, labels, pair index) of another context on the same device with the
 * same F and N: shared, not copied (`src` must outlive ctx).  What bench.py's second trainer uses. */
int dlco_set_data_shared(dlco_ctx *ctx, dlco_ctx *src);
/* Pair mode: instead of the N x F "Distance" matrix that comp-uprjdists materialises
 * (Dist = Desc1 - Desc2 per pair, src/comp-uprjdists.cpp:308-327; 16 GB for 500k x 8192)
 * the caller uploads the P per-patch descriptors desc_host [P,F] once and the pair table
 * pairs_host [N,4] = (patchID1, 3DpointID1, patchID2, 3DpointID2) of the reference's
 * "Indices" dataset.  Row i of the training matrix is desc[pairs[i][0]] - desc[pairs[i][2]],
 * formed in fp32 inside the kernels (the same single rounding as the reference's Mat
 * subtraction); Label[i] = (pairs[i][1] == pairs[i][3]), src/comp-uprjdists.cpp:268-272.
 * Every result is bit-identical to dlco_set_data on the pre-differenced matrix. */
int dlco_set_pairs(dlco_ctx *ctx, const float *desc_host, int32_t P, const int32_t *pairs_host);
/* same with the descriptor table already in device memory (e.g. written by dlco_desc_compute_device);
 * the caller keeps it alive for the life of the context */
int dlco_set_pairs_device(dlco_ctx *ctx, const float *desc_dev, int32_t P, const int32_t *pairs_host);
/* Bench helper (no reference counterpart: the Brown/Winder sets are not redistributable):
 * fills the context's Distance matrix in HBM with d = U^T z + noise*eps clipped to [-1,1],
 * z ~ N(0, (sigma * s_i)^2 I_k), s_i = exp(scale_jitter * g_i) a per-row log-normal scale (0 = none),
 * label = 1 for even rows; U is [k,F] on the host. */
int dlco_synth_data(dlco_ctx *ctx, const float *U_host, int32_t k, uint64_t seed,
                    float sigma_pos, float sigma_neg, float noise, float scale_jitter);
/* Copies `n` rows [row0, row0+n) of the device-resident Distance matrix back to the host. */
int dlco_get_rows(dlco_ctx *ctx, int32_t row0, int32_t n, float *out_host);

/* Pair indexing (bit-exact contract): shuffled index vectors and the 80/20 split,
 * src/pj-learn.cpp:214-237.  pos/neg may be NULL to query sizes only. */
int dlco_get_index(const dlco_ctx *ctx, int32_t *pos, int32_t *n_pos, int32_t *n_pos_trn,
                   int32_t *neg, int32_t *n_neg, int32_t *n_neg_trn);

/* One iteration of the training loop, src/pj-learn.cpp:305-490 (sampling, distances,
 * violation counts, gradient, dual average, PSD projection).  dlco_steps runs n of them
 * without host synchronisation beyond what the eigen tracker needs. */
int dlco_step(dlco_ctx *ctx);
int dlco_steps(dlco_ctx *ctx, int32_t n);

/* The same iteration split at its two exchange points, for data-parallel runs (one
 * process per GPU; the caller moves the buffers with RCCL between the phases):
 *   dlco_step_begin  : sample the global batch, project the rank's own slots;
 *                      fills the rank's slice of the distance buffer
 *   -- all-gather DLCO_BUF_DIST (2*B floats) --
 *   dlco_step_grad   : violation counts over the global batch, the rank's partial
 *                      gradient P^T diag(rho) P - N^T diag(kappa) N into DLCO_BUF_GRAD
 *   -- sum all-reduce DLCO_BUF_GRAD (F*F floats) --
 *   dlco_step_finish : dual average, PSD projection (replicated on every rank)
 * With world == 1 the three calls in sequence equal dlco_step. */
int dlco_step_begin(dlco_ctx *ctx);
int dlco_step_grad(dlco_ctx *ctx);
int dlco_step_finish(dlco_ctx *ctx);

#define DLCO_BUF_DIST   1   /* f32 [world][2*B/world]: rank g's slice holds the distances of its
                               B/world positive slots, then of its B/world negative slots   */
/* (device buffers: F below is the padded device width Fd = dlco_device_width(ctx), not cfg.F) */
#define DLCO_BUF_GRAD   2   /* f32 [Fd*Fd]: this rank's dLoss partial                  */
#define DLCO_BUF_DFAVG  3   /* f32 [Fd*Fd]: running dual average.  A single-rank context with Fd <= 8192 keeps only the
                               nt (nt + 1) / 2 tiles on or above the diagonal (nt = Fd / 128): contiguous row-major
                               128 x 128 blocks, tile (I, J), I <= J, at block index nt I - I (I - 1) / 2 + J - I
                               (dlco_get_dfavg / dlco_set_state convert)                                           */
#define DLCO_BUF_W      4   /* f32 [r*Fd]: current projection                          */
#define DLCO_BUF_GATHER 5   /* f32 [world][rows*F/world]: column slabs of a tracker product
                               (sharded contexts only)                                   */
#define DLCO_BUF_DATA   6   /* f32 [N*Fd] (row mode) or [P*Fd] (pair mode): the resident Distance /
                               descriptor matrix (dlco_set_data_shared shares it with a second context) */
/* Device pointer and byte size of an exchange buffer (valid until ctx is destroyed). */
int dlco_dev_buffer(dlco_ctx *ctx, int32_t which, void **dev_ptr, size_t *bytes);
/* Makes the context use caller-owned device memory for an exchange buffer (DLCO_BUF_DIST,
 * DLCO_BUF_GRAD or DLCO_BUF_GATHER), e.g. a tensor the caller's RCCL communicator operates
 * on.  The memory must outlive the context. */
int dlco_bind_buffer(dlco_ctx *ctx, int32_t which, void *dev_ptr, size_t bytes);

/* Sharded contexts (cfg.shard = 1, world > 1): the collective the step needs, supplied by the
 * host.  fn(user, which, bytes_per_rank) must all-gather, IN PLACE, the first
 * world*bytes_per_rank bytes of exchange buffer `which` (DLCO_BUF_DIST or DLCO_BUF_GATHER)
 * viewed as [world][bytes_per_rank] — the rank's own chunk is already filled — ordered after
 * the work queued on dlco_stream() and visible to work queued there afterwards (enqueue the
 * RCCL call on that stream, or synchronise around it).  Return 0 on success.  dlco_step calls
 * it once for the 2B distances and once per product of the eigen tracker with the dual
 * average (about six times a step, 4*rows*F/world bytes per rank each); every rank makes the
 * same sequence of calls.  The reference has no counterpart (single device). */
typedef int (*dlco_allgather_fn)(void *user, int32_t which, size_t bytes_per_rank);
int dlco_set_allgather(dlco_ctx *ctx, dlco_allgather_fn fn, void *user);
/* Alternative to the callback: the library performs the all-gathers itself with RCCL
 * (ncclAllGather over xGMI, in place, on its own stream; librccl is reached through dlopen:
 * the copy already loaded in the process, else `rccl_path`, else the system one).  Rank 0 obtains
 * a 128-byte ncclUniqueId with dlco_comm_unique_id, the host hands the same bytes to every rank
 * (MPI, torch.distributed, a file ...), and every rank calls dlco_comm_init (a collective:
 * ncclCommInitRank with cfg.rank / cfg.world).  A communicator, once created, takes precedence
 * over the callback; dlco_comm_destroy (or dlco_ctx_destroy) releases it.
 * A REPLICATED context (cfg.shard = 0, world > 1) with a communicator runs the literal exchange of BASELINE configs[3]
 * inside dlco_step: all-gather of the 2B distances, then ncclAllReduce (sum, f32) of the F x F partial gradients, then
 * the dual average and PSD projection on every rank (= dlco_step_begin / grad / finish with the library moving the
 * buffers).  The reference has no counterpart: it is single-device (src/pj-learn.cpp:267). */
int dlco_comm_unique_id(void *out_id, size_t cap, const char *rccl_path);
int dlco_comm_init(dlco_ctx *ctx, const void *id, size_t id_bytes, const char *rccl_path);
/* Fallback transport for ranks that are processes of one node when RCCL cannot be used (no librccl, or
 * several ranks sharing one GPU, which RCCL refuses): the all-gathers go through the POSIX shared-memory
 * segment `shm_name` (device -> host -> device, barriers on counters inside the segment).  Every rank
 * calls it (a collective); the segment must not exist beforehand or be all zeros, and is unlinked once
 * every rank has attached.  Correct and deterministic but host-staged: a test / fallback path, an RCCL
 * communicator takes precedence. */
int dlco_comm_init_host(dlco_ctx *ctx, const char *shm_name);
int dlco_comm_destroy(dlco_ctx *ctx);
/* The HIP stream the context launches on (as void*), so callers can order collectives. */
int dlco_stream(dlco_ctx *ctx, void **stream);
int dlco_sync(dlco_ctx *ctx);

/* Row ids (into Distance) of the last sampled batch and their squared distances
 * (src/pj-learn.cpp:311-314, 346-363); rho/kappa = violation counts (:373-376).
 * Valid once a step has run (DLCO_ERR_INVALID before that). */
int dlco_get_batch(const dlco_ctx *ctx, int32_t *pos_rows, int32_t *neg_rows,
                   float *pos_dist, float *neg_dist, int32_t *rho, int32_t *kappa);

/* State access.  Rank follows the reference: rows of W with a strictly positive
 * eigenvalue; when there are none, *r = F and W is F*F zeros (src/pj-learn.cpp:481-490). */
int dlco_get_t(const dlco_ctx *ctx, uint32_t *t);
int dlco_get_W(dlco_ctx *ctx, float *W_host /* cap F*F */, int32_t *r);
int dlco_get_A(dlco_ctx *ctx, float *A_host /* F*F: PSD-projected A = W^T W */);
int dlco_get_dfavg(dlco_ctx *ctx, float *dfavg_host /* F*F */);
/* Teacher forcing: overwrite t, dfAvg and W (W may be NULL with r = 0 for the zero start).
 * The eigen tracker restarts from the given W's row space. */
int dlco_set_state(dlco_ctx *ctx, uint32_t t, const float *dfavg_host, const float *W_host, int32_t r);

/* Validation objective, src/pj-learn.cpp:501-527: loss_val = hinge sum over the
 * validation positives x negatives / nPosVal / nNegVal, regul = mu * trace(A). */
int dlco_validate(dlco_ctx *ctx, float *loss_val, float *regul, int32_t *rank);

/* ComputePJStats (src/misc.cpp:266-333) on all N rows.  W_host == NULL uses the
 * context's current W. */
int dlco_stats(dlco_ctx *ctx, const float *W_host, int32_t r, int32_t *dim, float *fpr95, double *auc);

/* ---- single operators, exposed for parity tests and for eval-fpr95 ------------------- */
/* P1+P2: out[i] = || W x_{row_ids[i]} ||^2 over rows of the resident Distance matrix. */
int dlco_project_sqdist(dlco_ctx *ctx, const int32_t *row_ids_host, int32_t n,
                        const float *W_host, int32_t r, float *out_host);
/* V1: rho_i = #{j : pd_i + 1 > nd_j}, kappa_j = #{i : pd_i + 1 > nd_j}. */
int dlco_viol_counts(dlco_ctx *ctx, const float *pd_host, const float *nd_host, int32_t B,
                     int32_t *rho_host, int32_t *kappa_host);
/* Q1+U1: dfavg_out = beta*dfavg_in + alpha*(P^T diag(rho) P - N^T diag(kappa) N) with rows
 * taken from the resident Distance matrix (fused SYRK + dual average; the step uses alpha = 1/(B*B*(t+1)),
 * beta = t/(t+1), src/pj-learn.cpp:422). */
int dlco_grad_rda(dlco_ctx *ctx, const int32_t *pos_rows_host, const int32_t *neg_rows_host,
                  const int32_t *rho_host, const int32_t *kappa_host, int32_t B,
                  float alpha, float beta, const float *dfavg_in_host, float *dfavg_out_host);
/* E1+E2: PSD projection of A = -(sqrt(t+1)/gamma)(dfavg + mu I); returns W (rows ascending
 * in eigenvalue, as LAPACK orders them) and optionally A+. */
int dlco_psd_project(dlco_ctx *ctx, const float *dfavg_host, uint32_t t,
                     float *W_host, int32_t *r, float *A_host /* may be NULL */);
/* Building block of E1 (no single reference line: the reference calls LAPACKE_ssyevr, :440):
 * out[rows,F] = X[rows,F] * G[F,F] for a symmetric G, rows <= 128.  mode 0 = fp32 MFMA (k-ordered
 * fmaf chain), mode 1 = two-way split-bf16 MFMA with fp32 accumulation (the Chebyshev filter;
 * relative error ~1e-5), mode 2 = three-way split-bf16 MFMA (the Rayleigh-Ritz product when
 * F % 512 == 0; error at the level of fp32 rounding, ~1e-7).  Modes 3 / 4 (F == 8192 only) are modes 1 / 2 on the packed
 * upper-tile form of G that a single-rank trainer keeps its dual average in: only the upper triangle of G_host is read,
 * and every 128 x 128 tile is fetched from HBM once per product.  rows <= 160 for modes 1 / 3. */
int dlco_sym_product(dlco_ctx *ctx, const float *X_host, int32_t rows, const float *G_host, int32_t mode,
                     float *out_host);
/* H1: sum_i sum_j max(pos_i + 1 - neg_j, 0)  (src/kernelop-opencv.cu:49-80). */
int dlco_hinge_sum(dlco_ctx *ctx, const float *pos_host, int32_t n_pos,
                   const float *neg_host, int32_t n_neg, double *out);
/* S3+S4 on host arrays. */
int dlco_roc_stats(dlco_ctx *ctx, const float *dist_host, const uint8_t *labels_host, int32_t n,
                   float *fpr95, double *auc);

/* ---- model selection and logging, src/pj-learn.cpp:492-587 ---------------------------- */
typedef struct dlco_log_entry {
    uint32_t t;
    int32_t  is_best;      /* 1: "Best:" line + "Stat:" line, 0: "Step:" line */
    int32_t  saved;        /* 1: "[saved]" */
    float    loss_val, regul, obj, obj_best;
    int32_t  rank, rank_best;
    int32_t  dim;
    double   auc, auc_best;
    float    fpr95, fpr95_best;
    double   vtime;        /* seconds spent in this call (Vtime) */
    int32_t  nonconv;      /* training steps since the previous dlco_log_step whose eigen tracker stopped
                              short of eig_tol (no reference counterpart: LAPACKE_ssyevr is direct)      */
    int32_t  reserved;
} dlco_log_entry;
/* Runs the LogStep block once: validation, best-objective test, stats and save rule.
 * Keeps W_Best / W_Save / A_Save inside the context. */
int dlco_log_step(dlco_ctx *ctx, dlco_log_entry *out);
/* W_Save [r,F] and A_Save [F,F] (src/pj-learn.cpp:592-597); *r = 0 when nothing was saved. */
int dlco_get_saved(dlco_ctx *ctx, float *W_host, int32_t *r, float *A_host);

/* ---- measurement ------------------------------------------------------------------------ */
/* on = 1: the launches of the kernel groups "grad_syrk", "eig_product", "jacobi", "project", "rank_update" are bracketed by HIP events on
 * the context's stream; dlco_profile_read returns launches and their summed duration.  on = 2: the gradient SYRK only
 * (two records per step; all four groups cost ~3 % of the step in event records, measured).  on = 0: off. */
int dlco_profile_enable(dlco_ctx *ctx, int32_t on);
int dlco_profile_read(dlco_ctx *ctx, const char *kernel, int64_t *launches, double *total_ms);
/* Counters since creation: out[0] = training steps run, out[1] = sum over those steps of the
 * rows that entered this rank's gradient SYRK (rows with a non-zero violation count),
 * out[2] = steps whose eigen tracker stopped at its iteration cap, out[3] = calls of the multi-workgroup
 * Jacobi kernel that gave up at its bounded grid barrier (the m x m problem was then solved again on one
 * workgroup: a slow event, never a wrong result), out[4] = tracker updates whose first filter term came from the
 * step's own rank update (dfAvg_{t+1} = beta dfAvg_t + alpha X_a^T diag(w) X_a, src/pj-learn.cpp:367-422) instead
 * of a pass over the dual average, out[5] = with DLCO_RANK_UPDATE_CHECK=1 in the environment, 1e9 x the largest
 * relative deviation of such a term from the product it replaces (developer aid), out[6] = tracker passes that ran
 * with the converged top of the block locked out of the filter (start-up transient), out[7] = rows locked in them. */
int dlco_counters(const dlco_ctx *ctx, int64_t out[8]);
/* Tracker statistics since creation: filter/RR iterations, H-products (in rows), restarts. */
int dlco_eig_stats(const dlco_ctx *ctx, int64_t *iters, int64_t *product_rows, int64_t *jacobi_sweeps,
                   int32_t *block_rows);

/* ======================================================================================================
 * pr-learn — the pooling-region stage that precedes pj-learn.
 * Replaces the loop and the LogStep block of src/pr-learn.cpp:229-434 and ComputePRStats
 * (src/misc.cpp:171-264).  L1-regularised dual averaging on the weight vector w [F] of the F candidate
 * pooling regions: one (positive, negative) row pair per iteration, strictly sequential, so a window of
 * iterations is ONE kernel launch on one workgroup (w and dfAvg in registers).  The arithmetic follows
 * the reference operation for operation (single-thread order of its OpenMP loop), see
 * opencv-dlco_amd/csrc/kernels_pr.hip.
 * ====================================================================================================== */
typedef struct dlco_pr_ctx dlco_pr_ctx;
const char *dlco_pr_last_error(const dlco_pr_ctx *ctx);
/* F = FeatDim (5120 in the reference's runs; a multiple of 4, at most 8192), N rows; defaults of the
 * reference: mu 0.025, gamma 0.10, seed 2215 (src/pr-learn.cpp:78-82,241). */
int dlco_pr_create(dlco_pr_ctx **out, int32_t F, int32_t N, float mu, float gamma, uint64_t seed, int32_t device);
void dlco_pr_destroy(dlco_pr_ctx *ctx);
int dlco_pr_device_name(const dlco_pr_ctx *ctx, char *buf, size_t cap, int *cc_major, int *cc_minor);
/* Distance [N,F] f32 + Label [N] u8; index build, randShuffle and 80/20 split, src/pr-learn.cpp:229-253. */
int dlco_pr_set_data(dlco_pr_ctx *ctx, const float *dists_host, const uint8_t *labels_host);
int dlco_pr_get_index(const dlco_pr_ctx *ctx, int32_t *n_pos, int32_t *n_pos_trn, int32_t *n_neg, int32_t *n_neg_trn);
/* n iterations of src/pr-learn.cpp:302-329 (sample, subtract, dot product, dual average, soft threshold). */
int dlco_pr_steps(dlco_pr_ctx *ctx, uint32_t n);
int dlco_pr_get_state(dlco_pr_ctx *ctx, uint32_t *t, float *w_host /* F */, float *dfavg_host /* F */);
int dlco_pr_set_state(dlco_pr_ctx *ctx, uint32_t t, const float *w_host, const float *dfavg_host);
/* Validation objective of the current w, src/pr-learn.cpp:340-361: hinge loss over validation positives x
 * negatives / (nPosVal * nNegVal), regul = mu * sum|w|, nnz = countNonZero(w). */
int dlco_pr_validate(dlco_pr_ctx *ctx, float *loss_val, float *regul, int32_t *nnz);
/* ComputePRStats (src/misc.cpp:171-264) for w_host (NULL: the current w) on all N rows.  PRParams is the
 * [pr_rows >= 8*F, pr_cols] table of the filter file; nPR / Dim / nzDim as the reference counts them; when
 * max_dim != -1 and Dim > max_dim the function returns before the ROC pass (fpr95 / auc untouched). */
int dlco_pr_stats(dlco_pr_ctx *ctx, const float *w_host, const float *prparams_host, int32_t pr_rows, int32_t pr_cols,
                  int32_t nchannels, int32_t max_dim, int32_t *nPR, int32_t *dim, int32_t *nzdim, float *fpr95, double *auc);

/* ===========================================================================================
 * Descriptor generation — comp-uprjdists: the producer of pj-learn's input.
 * Replaces get_desc (src/vgg-desc.cpp:41-152), SelectPRFilters (src/misc.cpp:78-168) and the
 * per-pair loop of src/comp-uprjdists.cpp:298-349.  Descriptors are computed once per PATCH
 * (the reference recomputes both patches of every pair) and can stay in HBM for the trainer's
 * pair mode (dlco_set_pairs_device).
 * =========================================================================================== */
typedef struct dlco_desc_ctx dlco_desc_ctx;

/* InitSigma / nAngleBins / bNorm of src/comp-uprjdists.cpp:64-68 (1.4, 8, true); nAngleBins must be 8 */
int  dlco_desc_create(dlco_desc_ctx **out, float init_sigma, int32_t n_angle_bins, int32_t norm, int32_t device);
void dlco_desc_destroy(dlco_desc_ctx *ctx);
const char *dlco_desc_last_error(const dlco_desc_ctx *ctx);
/* SelectPRFilters: rows of pr_filters [8*wcols, cols] with w > 0 that are not all zero, unique,
 * ascending.  out [nsel, cols] may be NULL to query *nsel_out.  Host logic, no device needed. */
int  dlco_desc_select_filters(const float *pr_filters, int32_t rows, int32_t cols, const float *w, int32_t wcols,
                              float *out, int32_t *nsel_out);
/* sPRFilters [nsel, 4096], columns in the reference's order (pixel x*64+y of the transposed patch) */
int  dlco_desc_set_filters(dlco_desc_ctx *ctx, const float *filters_host, int32_t nsel);
int32_t dlco_desc_size(const dlco_desc_ctx *ctx);                         /* nsel * 8 */
/* get_desc of one 64x64 u8 patch: PatchTrans [4096][8] in the reference's layout */
int  dlco_desc_transform(dlco_desc_ctx *ctx, const uint8_t *patch_host, float *patch_trans_host);
/* Desc [n, nsel*8] = min(sPRFilters * get_desc(patch), 1), one row per patch [n,64,64] u8 */
int  dlco_desc_compute(dlco_desc_ctx *ctx, const uint8_t *patches_host, int64_t n, float *desc_host);
int  dlco_desc_compute_device(dlco_desc_ctx *ctx, const uint8_t *patches_host, int64_t n, float *desc_dev, int64_t ld);
/* Distance [n_pairs, nsel*8] and Label [n_pairs] (may be NULL) for pairs [n_pairs,4] =
 * (patchID1, 3DpointID1, patchID2, 3DpointID2): the two datasets comp-uprjdists writes */
int  dlco_desc_pair_dists(dlco_desc_ctx *ctx, const uint8_t *patches_host, int64_t n_patches, const int32_t *pairs_host,
                          int64_t n_pairs, float *dist_host, uint8_t *label_host);
/* comp-fulldists (src/comp-fulldists.cpp:285-369), the producer of pr-learn's input: with ALL pooling-region
 * filters set (8 rows per region), Distance [n_pairs, rows/8] = per region, the sum over its 8 rows and 8 bins of
 * (Desc2 - Desc1)^2 (cuda::subtract / pow / reduce there), and Label (may be NULL) */
int  dlco_desc_full_dists(dlco_desc_ctx *ctx, const uint8_t *patches_host, int64_t n_patches, const int32_t *pairs_host,
                          int64_t n_pairs, float *dist_host, uint8_t *label_host);
/* Streaming forms of the two calls above: the rows are handed to `sink` at most chunk_rows pairs at a time, in row
 * order (row0 = first pair of the chunk; dist [rows, cols] and label [rows] are library-owned staging buffers, valid
 * during the call only), so a tool can write each chunk as a hyperslab and check it, as the reference's loops do
 * (src/comp-uprjdists.cpp:298-349, src/comp-fulldists.cpp:300-369), without ever holding the whole Distance matrix.
 * A non-zero return of sink aborts the call with DLCO_ERR_INVALID. */
typedef int (*dlco_desc_sink_fn)(void *user, int64_t row0, int64_t rows, const float *dist, const uint8_t *label);
int  dlco_desc_pair_dists_stream(dlco_desc_ctx *ctx, const uint8_t *patches_host, int64_t n_patches, const int32_t *pairs_host,
                                 int64_t n_pairs, int64_t chunk_rows, dlco_desc_sink_fn sink, void *user);
int  dlco_desc_full_dists_stream(dlco_desc_ctx *ctx, const uint8_t *patches_host, int64_t n_patches, const int32_t *pairs_host,
                                 int64_t n_pairs, int64_t chunk_rows, dlco_desc_sink_fn sink, void *user);
double dlco_desc_last_kernel_ms(const dlco_desc_ctx *ctx);               /* HIP-event time of the last compute call */

#ifdef __cplusplus
}
#endif
#endif
