/*
 * dlco_ref.h — CPU ORACLE for the pj-learn hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This is a plain-C restatement of the algorithm in the reference
 * (cbalint13/opencv-dlco):
 *     src/pj-learn.cpp:214-256   index build, shuffle, 80/20 split
 *     src/pj-learn.cpp:305-490   one RDA training step
 *     src/pj-learn.cpp:492-587   validation, model selection, save rule
 *     src/kernelop-opencv.cu:49-66  hinge sum (SubtractVectorsByRows)
 *     src/misc.cpp:266-333       ComputePJStats (FPR@95, AUC)
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library.  Nothing under opencv-dlco_amd/ links, imports or calls
 * it; the product path fails loudly when its HIP library is missing.
 *
 * PIN STATUS
 *   - E2 conventions (eigenvalue order, W = sqrt(e)*v rows, A = W^T W,
 *     Regul = mu*trace(A)) are pinned against the reference's committed
 *     result files workspace/pj-learn/ *.h5 and logs (tests/golden/).
 *   - The arithmetic that lives in un-vendored third-party code (OpenCV >= 3.1
 *     cv::RNG / randShuffle / addWeighted / reduce / sortIdx / contourArea,
 *     OpenBLAS sgemm / ssyevr; version unpinned in the reference's CMake)
 *     is restated from the published algorithms.  OpenCV cannot be built or
 *     run in this environment and the reference has no tests or golden
 *     vectors for it, so for those pieces (R1-R3 pair indexing, S3/S4 ROC):
 *     PARITY UNPINNED against a live OpenCV; "bit-exact" is defined against
 *     this restatement.
 *   - The reference binary itself is unbuildable here (needs OpenCV core/hdf/
 *     cudaarithm, CUDA toolkit, OpenBLAS headers): there is no oracle/_ref.
 */
#ifndef DLCO_REF_H
#define DLCO_REF_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- BLAS/LAPACK back end ------------------------------------------------
 * The reference links OpenBLAS (CMakeLists.txt:97-101).  When a path to an
 * OpenBLAS shared object is given (e.g. scipy's bundled libscipy_openblas),
 * cblas_sgemm and LAPACKE_ssyevr are resolved from it (symbol prefixes ""
 * and "scipy_" are tried).  Otherwise built-in loops and a cyclic Jacobi
 * eigensolver (double precision) are used.  Returns 1 when OpenBLAS is in
 * use, 0 for the built-in kernels.                                          */
int  dlco_ref_load_blas(const char *path);
int  dlco_ref_blas_kind(void);           /* 1 = OpenBLAS, 0 = built-in */
void dlco_ref_set_threads(int n);

/* ---- R1..R3: cv::RNG (multiply-with-carry) and friends [OpenCV-src] ------ */
uint32_t dlco_ref_rng_next(uint64_t *state);
int      dlco_ref_rng_uniform(uint64_t *state, int a, int b);
void     dlco_ref_rand_shuffle_i32(int32_t *arr, uint32_t n, uint64_t *state);

/* src/pj-learn.cpp:214-227.  pos/neg must hold N entries each.  Shuffles with
 * OpenCV's thread-default RNG (state 0xFFFFFFFF), positives first.          */
void dlco_ref_build_index(const uint8_t *labels, int N,
                          int32_t *pos, int *n_pos, int32_t *neg, int *n_neg);
/* src/pj-learn.cpp:234-235: size_t(n * 0.80f) with float multiply.          */
size_t dlco_ref_split(size_t n);
/* src/pj-learn.cpp:310-314: interleaved iPos/iNeg draws for k < B.          */
void dlco_ref_sample(uint64_t *state, unsigned n_pos_trn, unsigned n_neg_trn,
                     int B, int32_t *ipos, int32_t *ineg);

/* ---- P1/P2: dist_k = sum_rho (W . x_k)_rho^2  (src/pj-learn.cpp:336-347) - */
void dlco_ref_project_sqdist(const float *W, int r, int F,
                             const float *X, int n, float *dist);
/* same, rows addressed through ids into a [N,F] matrix                      */
void dlco_ref_project_sqdist_ids(const float *W, int r, int F,
                                 const float *D, const int32_t *ids, int n,
                                 float *dist);

/* ---- V1: rho_i = #{j: pd_i + 1.0f > nd_j}, kappa_j = #{i: ...} ----------- */
void dlco_ref_viol_counts(const float *pd, const float *nd, int B,
                          int32_t *rho, int32_t *kappa);

/* ---- Q1: gradient, reference loop order (src/pj-learn.cpp:367-418) ------- */
void dlco_ref_grad_reforder(const float *P, const float *Ng,
                            const float *pd, const float *nd,
                            int B, int F, float *dloss);
/* reformulated: P^T diag(rho) P - N^T diag(kappa) N, fp32 (sgemm)           */
void dlco_ref_grad_reform(const float *P, const float *Ng,
                          const int32_t *rho, const int32_t *kappa,
                          int B, int F, float *dloss);
/* same in double accumulation: ground truth for tolerance checks            */
void dlco_ref_grad_reform_f64(const float *P, const float *Ng,
                              const int32_t *rho, const int32_t *kappa,
                              int B, int F, double *dloss);

/* ---- U1/U2 (src/pj-learn.cpp:422-432) ------------------------------------ */
void dlco_ref_rda_update(float *dfavg, const float *dloss,
                         unsigned t, unsigned B, int F);
void dlco_ref_dual_to_primal(const float *dfavg, float mu, float gamma,
                             unsigned t, int F, float *A);

/* ---- E1/E2 (src/pj-learn.cpp:434-490) ------------------------------------
 * A: in = symmetric matrix, out = its PSD projection Evec*diag(e+)*Evec^T.
 * W: capacity F*F; receives r rows sqrt(e_k)*v_k^T for e_k != 0, ascending.
 * If no eigenvalue is positive, W is set to F*F zeros and *r = F (reference
 * quirk, src/pj-learn.cpp:489-490).  evals (may be NULL) receives all F
 * eigenvalues ascending.  Returns 0 on success.                            */
int dlco_ref_psd_project(float *A, int F, float *W, int *r, float *evals);
/* The same ssyevr call and the same W (src/pj-learn.cpp:434-469,481-490) without the F^3
 * back-multiplication A = Evec*Bmul (:472-478); A is not modified.  Used by the F = 8192
 * parity tests, which form A+ = W^T W themselves.                                          */
int dlco_ref_psd_factor(const float *A, int F, float *W, int *r, float *evals);

/* ---- H1 (src/kernelop-opencv.cu:49-66 + src/pj-learn.cpp:520) ------------
 * per-row sequential fp32 sum of max(pos_i + 1 - neg_j, 0); rows summed in
 * double (cv::cuda::sum accumulates in double) [OpenCV-src].               */
double dlco_ref_hinge_sum(const float *pos, int n_pos,
                          const float *neg, int n_neg);
/* H2 trace with double accumulator (cv::trace)                              */
double dlco_ref_trace(const float *A, int F);

/* ---- S1..S4 (src/misc.cpp:266-333) ---------------------------------------
 * dist[N] are the squared distances of ALL rows, labels[N] in {0,1,other}.
 * Sort is ascending by (dist, index): the reference's std::sort leaves ties
 * unordered; the restatement fixes them by index.                          */
int  dlco_ref_nonzero_rows(const float *W, int r, int F, float *nzW);
void dlco_ref_roc_stats(const float *dist, const uint8_t *labels, int N,
                        float *fpr95, double *auc);

/* ---- whole trainer -------------------------------------------------------- */
typedef struct dlco_ref_ctx dlco_ref_ctx;

/* dists [N,F] row-major and labels [N] are borrowed (must outlive ctx).     */
dlco_ref_ctx *dlco_ref_create(const float *dists, const uint8_t *labels,
                              int N, int F, int B, float mu, float gamma);
void dlco_ref_destroy(dlco_ref_ctx *c);
/* grad_order: 0 = reference loop order, 1 = reformulated                    */
void dlco_ref_set_grad_order(dlco_ref_ctx *c, int order);
/* one iteration of the loop at src/pj-learn.cpp:305-490 (uses and then
 * increments the context's t).  Returns 0 on success.                      */
int  dlco_ref_step(dlco_ref_ctx *c);
/* wall seconds since creation: [0] sampling + P1/P2, [1] Q1, [2] U1/U2, [3] E1/E2 */
void dlco_ref_get_timers(const dlco_ref_ctx *c, double out[4]);
/* teacher forcing hooks                                                     */
void dlco_ref_get_batch_ids(const dlco_ref_ctx *c, int32_t *pos_rows, int32_t *neg_rows);
void dlco_ref_get_batch_dists(const dlco_ref_ctx *c, float *pd, float *nd);
void dlco_ref_get_state(const dlco_ref_ctx *c, unsigned *t, int *r,
                        float *W, float *A, float *dfavg, float *dloss);
void dlco_ref_set_state(dlco_ref_ctx *c, unsigned t, const float *dfavg,
                        const float *W, int r);
void dlco_ref_get_index(const dlco_ref_ctx *c, int32_t *pos, int *n_pos, int *n_pos_trn,
                        int32_t *neg, int *n_neg, int *n_neg_trn);
/* src/pj-learn.cpp:501-527: validation loss and regulariser                 */
void dlco_ref_validate(const dlco_ref_ctx *c, float *loss_val, float *regul);
/* src/misc.cpp:266-333 on the ctx's data and current W                      */
void dlco_ref_stats(const dlco_ref_ctx *c, int *dim, float *fpr95, double *auc);

/* ---- pr-learn: src/pr-learn.cpp:229-434, src/misc.cpp:171-264 ---------------
 * L1-regularised dual averaging on the pooling-region weight vector w [F]: one (positive, negative)
 * row pair per iteration, in the reference's single-thread order (see dlco_ref.c for the OpenCV
 * semantics that are restated, parity unpinned like the PJ stage).                            */
typedef struct dlco_ref_pr dlco_ref_pr;
dlco_ref_pr *dlco_ref_pr_create(const float *dists, const uint8_t *labels, int N, int F, float mu, float gamma);
void dlco_ref_pr_destroy(dlco_ref_pr *c);
void dlco_ref_pr_step(dlco_ref_pr *c);
void dlco_ref_pr_steps(dlco_ref_pr *c, unsigned n);
void dlco_ref_pr_get(const dlco_ref_pr *c, unsigned *t, float *w, float *dfavg, int32_t *last_pos, int32_t *last_neg, float *last_f);
void dlco_ref_pr_set(dlco_ref_pr *c, unsigned t, const float *w, const float *dfavg);
void dlco_ref_pr_validate(const dlco_ref_pr *c, float *loss_val, float *regul, int *nnz);
void dlco_ref_pr_stats(const float *prparams, int pr_cols, const float *dists, const uint8_t *labels, int N, int F,
                       const float *w, int nchannels, int max_dim, int *nPR, int *Dim, int *nzDim, float *fpr95, double *auc);

/* ---- descriptor generation: src/vgg-desc.cpp:41-152, src/comp-uprjdists.cpp:298-349 ----
 * get_desc on one 64 x 64 u8 patch -> PatchTrans [4096][nAngleBins]; and the pooled descriptor
 * min(sPRFilters * PatchTrans, 1) [nsel*8] that comp-uprjdists differences per pair.                */
void dlco_ref_get_desc(const uint8_t *patch, int nAngleBins, float InitSigma, int bNorm, float *PatchTrans);
void dlco_ref_patch_descriptor(const uint8_t *patch, const float *sPR, int nsel, float *desc);

#ifdef __cplusplus
}
#endif
#endif
