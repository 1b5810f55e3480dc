/*
 * dlco_ref.c — CPU ORACLE for the pj-learn hot path.  TEST INFRASTRUCTURE ONLY.
 * See dlco_ref.h for scope, citations and pin status ("parity unpinned" for the
 * pieces that restate un-vendored OpenCV/OpenBLAS internals).
 *
 * Every function cites the reference lines it follows (paths relative to the
 * reference checkout, cbalint13/opencv-dlco).
 */
#define _GNU_SOURCE
#include "dlco_ref.h"

#include <dlfcn.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------- */
/* BLAS / LAPACK back end                                                    */
/* ------------------------------------------------------------------------- */
enum { kRowMajor = 101, kNoTrans = 111, kTrans = 112 };

typedef void (*sgemm_fn)(int order, int ta, int tb, int M, int N, int K,
                         float alpha, const float *A, int lda,
                         const float *B, int ldb, float beta, float *C, int ldc);
typedef int (*ssyevr_fn)(int layout, char jobz, char range, char uplo, int n,
                         float *a, int lda, float vl, float vu, int il, int iu,
                         float abstol, int *m, float *w, float *z, int ldz,
                         int *isuppz);
typedef void (*setthr_fn)(int);

static sgemm_fn  g_sgemm  = NULL;
static ssyevr_fn g_ssyevr = NULL;
static setthr_fn g_setthr = NULL;
static void     *g_blas_handle = NULL;

static void *sym2(void *h, const char *a, const char *b)
{
    void *p = dlsym(h, a);
    return p ? p : dlsym(h, b);
}

int dlco_ref_load_blas(const char *path)
{
    if (!path || !*path) return dlco_ref_blas_kind();
    void *h = dlopen(path, RTLD_NOW | RTLD_LOCAL);
    if (!h) return 0;
    sgemm_fn  sg = (sgemm_fn) sym2(h, "cblas_sgemm", "scipy_cblas_sgemm");
    ssyevr_fn sy = (ssyevr_fn)sym2(h, "LAPACKE_ssyevr", "scipy_LAPACKE_ssyevr");
    if (!sg || !sy) { dlclose(h); return 0; }
    g_blas_handle = h; g_sgemm = sg; g_ssyevr = sy;
    g_setthr = (setthr_fn)sym2(h, "openblas_set_num_threads", "scipy_openblas_set_num_threads");
    return 1;
}

int dlco_ref_blas_kind(void) { return g_sgemm && g_ssyevr ? 1 : 0; }

void dlco_ref_set_threads(int n)
{
    if (n < 1) n = 1;
    if (g_setthr) g_setthr(n);
#ifdef _OPENMP
    omp_set_num_threads(n);
#endif
}

/* row-major sgemm, C = alpha*op(A)*op(B) + beta*C, k-ordered fp32 sums       */
static void naive_sgemm(int ta, int tb, int M, int N, int K, float alpha,
                        const float *A, int lda, const float *B, int ldb,
                        float beta, float *C, int ldc)
{
#pragma omp parallel for schedule(static)
    for (int i = 0; i < M; i++) {
        for (int j = 0; j < N; j++) {
            float s = 0.0f;
            for (int k = 0; k < K; k++) {
                float a = (ta == kTrans) ? A[(size_t)k * lda + i] : A[(size_t)i * lda + k];
                float b = (tb == kTrans) ? B[(size_t)j * ldb + k] : B[(size_t)k * ldb + j];
                s += a * b;
            }
            float c = (beta == 0.0f) ? 0.0f : beta * C[(size_t)i * ldc + j];
            C[(size_t)i * ldc + j] = alpha * s + c;
        }
    }
}

static void ref_sgemm(int ta, int tb, int M, int N, int K, float alpha,
                      const float *A, int lda, const float *B, int ldb,
                      float beta, float *C, int ldc)
{
    if (M <= 0 || N <= 0) return;
    if (g_sgemm) g_sgemm(kRowMajor, ta, tb, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc);
    else naive_sgemm(ta, tb, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc);
}

/* cyclic Jacobi in double on a symmetric matrix; eigenvalues ascending,
 * Z[i*n + j] = component i of eigenvector j (the layout LAPACKE's ROW_MAJOR
 * ssyevr returns).  Built-in stand-in used only when OpenBLAS is not loaded. */
static int jacobi_eig(const float *Ain, int n, float *w, float *Z)
{
    double *a = (double *)malloc(sizeof(double) * n * n);
    double *v = (double *)malloc(sizeof(double) * n * n);
    if (!a || !v) { free(a); free(v); return -1; }
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) {
            /* 'U': only the upper triangle is referenced */
            a[(size_t)i * n + j] = (i <= j) ? Ain[(size_t)i * n + j] : Ain[(size_t)j * n + i];
            v[(size_t)i * n + j] = (i == j) ? 1.0 : 0.0;
        }
    for (int sweep = 0; sweep < 60; sweep++) {
        double off = 0.0, diag = 0.0;
        for (int i = 0; i < n; i++) {
            diag += a[(size_t)i * n + i] * a[(size_t)i * n + i];
            for (int j = i + 1; j < n; j++) off += a[(size_t)i * n + j] * a[(size_t)i * n + j];
        }
        if (off <= 1e-30 * (diag + off) || off == 0.0) break;
        for (int p = 0; p < n - 1; p++)
            for (int q = p + 1; q < n; q++) {
                double apq = a[(size_t)p * n + q];
                if (apq == 0.0) continue;
                double app = a[(size_t)p * n + p], aqq = a[(size_t)q * n + q];
                double theta = (aqq - app) / (2.0 * apq);
                double tt = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                double c = 1.0 / sqrt(tt * tt + 1.0), s = tt * c;
                for (int k = 0; k < n; k++) {
                    double akp = a[(size_t)k * n + p], akq = a[(size_t)k * n + q];
                    a[(size_t)k * n + p] = c * akp - s * akq;
                    a[(size_t)k * n + q] = s * akp + c * akq;
                }
                for (int k = 0; k < n; k++) {
                    double apk = a[(size_t)p * n + k], aqk = a[(size_t)q * n + k];
                    a[(size_t)p * n + k] = c * apk - s * aqk;
                    a[(size_t)q * n + k] = s * apk + c * aqk;
                }
                for (int k = 0; k < n; k++) {
                    double vkp = v[(size_t)k * n + p], vkq = v[(size_t)k * n + q];
                    v[(size_t)k * n + p] = c * vkp - s * vkq;
                    v[(size_t)k * n + q] = s * vkp + c * vkq;
                }
            }
    }
    /* sort ascending */
    int *ord = (int *)malloc(sizeof(int) * n);
    for (int i = 0; i < n; i++) ord[i] = i;
    for (int i = 1; i < n; i++) {
        int o = ord[i]; double key = a[(size_t)o * n + o]; int j = i - 1;
        while (j >= 0 && a[(size_t)ord[j] * n + ord[j]] > key) { ord[j + 1] = ord[j]; j--; }
        ord[j + 1] = o;
    }
    for (int j = 0; j < n; j++) {
        w[j] = (float)a[(size_t)ord[j] * n + ord[j]];
        for (int i = 0; i < n; i++) Z[(size_t)i * n + j] = (float)v[(size_t)i * n + ord[j]];
    }
    free(ord); free(a); free(v);
    return 0;
}

/* ------------------------------------------------------------------------- */
/* R1..R3: cv::RNG restated [OpenCV-src: core/operations.hpp, rand.cpp]      */
/* ------------------------------------------------------------------------- */
#define CV_RNG_COEFF 4164903690U

/* RNG::next(): state = (uint64)(unsigned)state*CV_RNG_COEFF + (unsigned)(state>>32) */
uint32_t dlco_ref_rng_next(uint64_t *state)
{
    uint64_t s = *state;
    s = (uint64_t)(uint32_t)s * CV_RNG_COEFF + (uint32_t)(s >> 32);
    *state = s;
    return (uint32_t)s;
}

/* RNG::uniform(int a, int b): a == b ? a : (int)(next() % (b - a) + a)       */
int dlco_ref_rng_uniform(uint64_t *state, int a, int b)
{
    if (a == b) return a;
    return (int)(dlco_ref_rng_next(state) % (uint32_t)(b - a) + (uint32_t)a);
}

/* randShuffle_<T> on a continuous array: for i<sz: j = (unsigned)rng % sz;
 * swap(arr[j], arr[i])  (iterFactor unused for this branch)                 */
void dlco_ref_rand_shuffle_i32(int32_t *arr, uint32_t n, uint64_t *state)
{
    for (uint32_t i = 0; i < n; i++) {
        uint32_t j = dlco_ref_rng_next(state) % n;
        int32_t tmp = arr[j]; arr[j] = arr[i]; arr[i] = tmp;
    }
}

/* src/pj-learn.cpp:214-227.  NB the reference declares RNG rng(2215) but calls
 * randShuffle() without it, so the shuffles draw from theRNG() whose default
 * state is 0xFFFFFFFF; positives are shuffled first, negatives continue the
 * same stream.                                                              */
void dlco_ref_build_index(const uint8_t *labels, int N,
                          int32_t *pos, int *n_pos, int32_t *neg, int *n_neg)
{
    int np = 0, nn = 0;
    for (int i = 0; i < N; i++) {
        if (labels[i] == 1) pos[np++] = i;
        if (labels[i] == 0) neg[nn++] = i;
    }
    uint64_t the_rng = 0xffffffffULL;
    if (np) dlco_ref_rand_shuffle_i32(pos, (uint32_t)np, &the_rng);
    if (nn) dlco_ref_rand_shuffle_i32(neg, (uint32_t)nn, &the_rng);
    *n_pos = np; *n_neg = nn;
}

/* src/pj-learn.cpp:96,234-235: size_t nTrn = size() * nDiv with float nDiv   */
size_t dlco_ref_split(size_t n)
{
    const float nDiv = 0.80f;
    volatile float prod = (float)n * nDiv;
    return (size_t)prod;
}

/* src/pj-learn.cpp:310-314                                                   */
void dlco_ref_sample(uint64_t *state, unsigned n_pos_trn, unsigned n_neg_trn,
                     int B, int32_t *ipos, int32_t *ineg)
{
    for (int k = 0; k < B; k++) {
        ipos[k] = dlco_ref_rng_uniform(state, 0, (int)n_pos_trn);
        ineg[k] = dlco_ref_rng_uniform(state, 0, (int)n_neg_trn);
    }
}

/* ------------------------------------------------------------------------- */
/* P1/P2                                                                     */
/* ------------------------------------------------------------------------- */
/* src/pj-learn.cpp:336-347: Proj = W*X^T (sgemm NoTrans,Trans), pow(.,2),
 * reduce(dim 0, SUM) which adds the rows one after another in fp32.         */
void dlco_ref_project_sqdist(const float *W, int r, int F,
                             const float *X, int n, float *dist)
{
    for (int k = 0; k < n; k++) dist[k] = 0.0f;
    if (r <= 0 || n <= 0) return;
    float *proj = (float *)malloc(sizeof(float) * (size_t)r * n);
    ref_sgemm(kNoTrans, kTrans, r, n, F, 1.0f, W, F, X, F, 0.0f, proj, n);
    for (int q = 0; q < r; q++)
        for (int k = 0; k < n; k++) {
            float p = proj[(size_t)q * n + k];
            dist[k] += p * p;
        }
    free(proj);
}

void dlco_ref_project_sqdist_ids(const float *W, int r, int F,
                                 const float *D, const int32_t *ids, int n,
                                 float *dist)
{
    const int chunk = 4096;
    float *X = (float *)malloc(sizeof(float) * (size_t)chunk * F);
    for (int s = 0; s < n; s += chunk) {
        int m = n - s < chunk ? n - s : chunk;
        for (int k = 0; k < m; k++)
            memcpy(X + (size_t)k * F, D + (size_t)ids[s + k] * F, sizeof(float) * F);
        dlco_ref_project_sqdist(W, r, F, X, m, dist + s);
    }
    free(X);
}

/* ------------------------------------------------------------------------- */
/* V1: src/pj-learn.cpp:373-376 — strict '>' on (PosDist[i] + 1.0f)          */
/* ------------------------------------------------------------------------- */
void dlco_ref_viol_counts(const float *pd, const float *nd, int B,
                          int32_t *rho, int32_t *kappa)
{
    for (int j = 0; j < B; j++) kappa[j] = 0;
    for (int i = 0; i < B; i++) {
        float thr = pd[i] + 1.0f;
        int c = 0;
        for (int j = 0; j < B; j++)
            if (thr > nd[j]) { c++; kappa[j]++; }
        rho[i] = c;
    }
}

/* ------------------------------------------------------------------------- */
/* Q1                                                                        */
/* ------------------------------------------------------------------------- */
/* src/pj-learn.cpp:367-418.  Per positive: gather violating negatives,
 * mul = Ncur^T Ncur (sgemm), mul = nViol*p^T p - mul (sgemm alpha=nViol,
 * beta=-1), dLoss += mul.  The reference adds under `omp critical` in thread
 * arrival order; the restatement adds in ascending iPos.                    */
void dlco_ref_grad_reforder(const float *P, const float *Ng,
                            const float *pd, const float *nd,
                            int B, int F, float *dloss)
{
    size_t FF = (size_t)F * F;
    memset(dloss, 0, sizeof(float) * FF);
    float *mul  = (float *)malloc(sizeof(float) * FF);
    float *ncur = (float *)malloc(sizeof(float) * (size_t)B * F);
    for (int i = 0; i < B; i++) {
        float thr = pd[i] + 1.0f;
        int nviol = 0;
        for (int j = 0; j < B; j++)
            if (thr > nd[j]) {
                memcpy(ncur + (size_t)nviol * F, Ng + (size_t)j * F, sizeof(float) * F);
                nviol++;
            }
        if (nviol == 0) continue;
        ref_sgemm(kTrans, kNoTrans, F, F, nviol, 1.0f, ncur, F, ncur, F, 0.0f, mul, F);
        const float *p = P + (size_t)i * F;
        ref_sgemm(kTrans, kNoTrans, F, F, 1, (float)nviol, p, F, p, F, -1.0f, mul, F);
#pragma omp parallel for schedule(static)
        for (size_t e = 0; e < FF; e++) dloss[e] += mul[e];
    }
    free(mul); free(ncur);
}

/* P^T diag(rho) P - N^T diag(kappa) N as one sgemm over the stacked, scaled
 * batch (rows with zero weight dropped).                                    */
void dlco_ref_grad_reform(const float *P, const float *Ng,
                          const int32_t *rho, const int32_t *kappa,
                          int B, int F, float *dloss)
{
    float *L = (float *)malloc(sizeof(float) * (size_t)2 * B * F);
    float *R = (float *)malloc(sizeof(float) * (size_t)2 * B * F);
    int K = 0;
    for (int i = 0; i < B; i++) if (rho[i]) {
        const float *x = P + (size_t)i * F; float s = (float)rho[i];
        for (int f = 0; f < F; f++) { L[(size_t)K * F + f] = s * x[f]; R[(size_t)K * F + f] = x[f]; }
        K++;
    }
    for (int j = 0; j < B; j++) if (kappa[j]) {
        const float *x = Ng + (size_t)j * F; float s = -(float)kappa[j];
        for (int f = 0; f < F; f++) { L[(size_t)K * F + f] = s * x[f]; R[(size_t)K * F + f] = x[f]; }
        K++;
    }
    if (K == 0) memset(dloss, 0, sizeof(float) * (size_t)F * F);
    else ref_sgemm(kTrans, kNoTrans, F, F, K, 1.0f, L, F, R, F, 0.0f, dloss, F);
    free(L); free(R);
}

void dlco_ref_grad_reform_f64(const float *P, const float *Ng,
                              const int32_t *rho, const int32_t *kappa,
                              int B, int F, double *dloss)
{
#pragma omp parallel for schedule(static)
    for (int a = 0; a < F; a++) {
        double *row = dloss + (size_t)a * F;
        for (int b = 0; b < F; b++) row[b] = 0.0;
        for (int i = 0; i < B; i++) if (rho[i]) {
            const float *x = P + (size_t)i * F; double s = (double)rho[i] * x[a];
            for (int b = 0; b < F; b++) row[b] += s * x[b];
        }
        for (int j = 0; j < B; j++) if (kappa[j]) {
            const float *x = Ng + (size_t)j * F; double s = (double)kappa[j] * x[a];
            for (int b = 0; b < F; b++) row[b] -= s * x[b];
        }
    }
}

/* ------------------------------------------------------------------------- */
/* U1/U2                                                                     */
/* ------------------------------------------------------------------------- */
/* src/pj-learn.cpp:422: addWeighted(dfAvg, (double)t/(t+1), dLoss,
 * 1.0f/(szBatch*szBatch*(t+1)), 0, dfAvg).  The reference's beta denominator is a 32-bit
 * unsigned product.  cv::addWeighted on CV_32F evaluates
 * src1*alpha + src2*beta + gamma with float scalars [OpenCV-src].           */
void dlco_ref_rda_update(float *dfavg, const float *dloss,
                         unsigned t, unsigned B, int F)
{
    double alpha_d = (double)t / (t + 1);
    /* 64-bit product: the same bits as the reference's 32-bit one wherever that does not
     * wrap (B = 200: t < 107374); the GLOBAL batch of a multi-GPU run would wrap it */
    float  beta_f  = 1.0f / (float)((unsigned long long)B * (unsigned long long)B * ((unsigned long long)t + 1ull));
    float  alpha   = (float)alpha_d, beta = beta_f;
    size_t FF = (size_t)F * F;
#pragma omp parallel for schedule(static)
    for (size_t e = 0; e < FF; e++)
        dfavg[e] = dfavg[e] * alpha + dloss[e] * beta;
}

/* src/pj-learn.cpp:426-432: A = dfAvg + mu*I; A = A.mul(-sqrt(t+1)/gamma)
 * (scalar applied as float on a CV_32F Mat [OpenCV-src]); A = 0.5f*(A + A^T) */
void dlco_ref_dual_to_primal(const float *dfavg, float mu, float gamma,
                             unsigned t, int F, float *A)
{
    double div = -sqrt((double)t + 1.0f) / (double)gamma;
    float  fdiv = (float)div;
    size_t FF = (size_t)F * F;
    float *tmp = (float *)malloc(sizeof(float) * FF);
#pragma omp parallel for schedule(static)
    for (int i = 0; i < F; i++)
        for (int j = 0; j < F; j++) {
            float v = dfavg[(size_t)i * F + j] + (i == j ? mu : 0.0f);
            tmp[(size_t)i * F + j] = v * fdiv;
        }
#pragma omp parallel for schedule(static)
    for (int i = 0; i < F; i++)
        for (int j = 0; j < F; j++)
            A[(size_t)i * F + j] = 0.5f * (tmp[(size_t)i * F + j] + tmp[(size_t)j * F + i]);
    free(tmp);
}

/* ------------------------------------------------------------------------- */
/* E1/E2: src/pj-learn.cpp:434-490                                           */
/* ------------------------------------------------------------------------- */
static int psd_project_impl(float *A, int F, float *W, int *r, float *evals, int want_A)
{
    size_t FF = (size_t)F * F;
    float *Eval = (float *)malloc(sizeof(float) * F);
    float *Evec = (float *)malloc(sizeof(float) * FF);
    float *Bmul = (float *)malloc(sizeof(float) * (want_A ? FF : 1));
    float *sqB  = (float *)malloc(sizeof(float) * FF);
    int rc = 0;
    if (g_ssyevr) {
        /* LAPACKE_ssyevr(ROW_MAJOR,'V','A','U',n,A,n,0,0,0,0,abstol=0,&m,w,z,ldz=n,isuppz) */
        int m = 0;
        int *isuppz = (int *)malloc(sizeof(int) * 2 * (size_t)F);
        float *Acopy = (float *)malloc(sizeof(float) * FF);
        memcpy(Acopy, A, sizeof(float) * FF);
        rc = g_ssyevr(kRowMajor, 'V', 'A', 'U', F, Acopy, F, 0.0f, 0.0f, 0, 0, 0.0f,
                      &m, Eval, Evec, F, isuppz);
        free(Acopy); free(isuppz);
    } else {
        rc = jacobi_eig(A, F, Eval, Evec);
    }
    if (rc != 0) { free(Eval); free(Evec); free(Bmul); free(sqB); return rc; }
    if (evals) memcpy(evals, Eval, sizeof(float) * F);

    /* V = Evec^T; diagDPos = max(Eval,0); Bmul = e*V; sqBmul = sqrt(e)*V     */
#pragma omp parallel for schedule(static)
    for (int q = 0; q < F; q++) {
        float e   = Eval[q] > 0.0f ? Eval[q] : 0.0f;
        float sqe = sqrtf(e);
        for (int c = 0; c < F; c++) {
            float v = Evec[(size_t)c * F + q];
            if (want_A) Bmul[(size_t)q * F + c] = e * v;
            sqB [(size_t)q * F + c] = sqe * v;
        }
    }
    /* A = Evec * Bmul (src/pj-learn.cpp:472-478)                            */
    if (want_A) ref_sgemm(kNoTrans, kNoTrans, F, F, F, 1.0f, Evec, F, Bmul, F, 0.0f, A, F);
    /* W = rows with diagDPos != 0 (src/pj-learn.cpp:481-484)                */
    int rows = 0;
    for (int q = 0; q < F; q++) {
        float e = Eval[q] > 0.0f ? Eval[q] : 0.0f;
        if (e != 0.0f) { memcpy(W + (size_t)rows * F, sqB + (size_t)q * F, sizeof(float) * F); rows++; }
    }
    if (rows == 0) { memset(W, 0, sizeof(float) * FF); rows = F; }   /* :489-490 */
    *r = rows;
    free(Eval); free(Evec); free(Bmul); free(sqB);
    return 0;
}

int dlco_ref_psd_project(float *A, int F, float *W, int *r, float *evals)
{
    return psd_project_impl(A, F, W, r, evals, 1);
}

/* Same eigendecomposition and the same W (src/pj-learn.cpp:434-469,481-490) without the
 * F^3 back-multiplication A = Evec*Bmul (:472-478); A is left untouched.  For parity tests at
 * F = 8192, where A+ is formed as W^T W by the caller.                                      */
int dlco_ref_psd_factor(const float *A, int F, float *W, int *r, float *evals)
{
    return psd_project_impl((float *)A, F, W, r, evals, 0);
}

/* ------------------------------------------------------------------------- */
/* H1/H2                                                                     */
/* ------------------------------------------------------------------------- */
/* src/kernelop-opencv.cu:55-65: tsum += (rsum > 0) ? rsum : 0 with
 * rsum = src1[idx] + 1 - src2[i] evaluated left to right in fp32;
 * src/pj-learn.cpp:520: cuda::sum over the per-row results (double).         */
double dlco_ref_hinge_sum(const float *pos, int n_pos, const float *neg, int n_neg)
{
    double total = 0.0;
#pragma omp parallel for schedule(static) reduction(+:total)
    for (int i = 0; i < n_pos; i++) {
        float tsum = 0.0f;
        float base = pos[i] + 1.0f;          /* (src1 + 1) first, then - src2 */
        for (int j = 0; j < n_neg; j++) {
            float rsum = base - neg[j];
            tsum += (rsum > 0.0f) ? rsum : 0.0f;
        }
        total += (double)tsum;
    }
    return total;
}

double dlco_ref_trace(const float *A, int F)
{
    double s = 0.0;
    for (int i = 0; i < F; i++) s += (double)A[(size_t)i * F + i];
    return s;
}

/* ------------------------------------------------------------------------- */
/* S1..S4: src/misc.cpp:266-333                                              */
/* ------------------------------------------------------------------------- */
int dlco_ref_nonzero_rows(const float *W, int r, int F, float *nzW)
{
    int rows = 0;
    for (int i = 0; i < r; i++) {
        int nz = 0;
        for (int f = 0; f < F; f++) if (W[(size_t)i * F + f] != 0.0f) { nz = 1; break; }
        if (nz) { if (nzW) memcpy(nzW + (size_t)rows * F, W + (size_t)i * F, sizeof(float) * F); rows++; }
    }
    return rows;
}

typedef struct { float d; int32_t i; } dist_idx;
static int cmp_dist_idx(const void *a, const void *b)
{
    const dist_idx *x = (const dist_idx *)a, *y = (const dist_idx *)b;
    if (x->d < y->d) return -1;
    if (x->d > y->d) return 1;
    return (x->i > y->i) - (x->i < y->i);
}

void dlco_ref_roc_stats(const float *dist, const uint8_t *labels, int N,
                        float *fpr95, double *auc)
{
    dist_idx *ord = (dist_idx *)malloc(sizeof(dist_idx) * (size_t)N);
    float *TPR = (float *)malloc(sizeof(float) * (size_t)N);
    float *FPR = (float *)malloc(sizeof(float) * (size_t)N);
    for (int i = 0; i < N; i++) { ord[i].d = dist[i]; ord[i].i = i; }
    qsort(ord, (size_t)N, sizeof(dist_idx), cmp_dist_idx);     /* sortIdx ascending */

    float tplast = 0.0f, fplast = 0.0f;                        /* float counters (:302) */
    for (int i = 0; i < N; i++) {
        int idx = ord[i].i;
        if (labels[idx] == 1) tplast++;
        if (labels[idx] == 0) fplast++;
        TPR[i] = tplast; FPR[i] = fplast;
    }
    /* TPR /= tplast: Mat /= double is convertTo(a,-1,1./s); the 32f->32f scale
     * path multiplies by (float)(1./s) [OpenCV-src]                          */
    float ts = (float)(1.0 / (double)tplast), fs = (float)(1.0 / (double)fplast);
    for (int i = 0; i < N; i++) { TPR[i] = TPR[i] * ts; FPR[i] = FPR[i] * fs; }

    float f95 = -1.0f;
    for (int i = 0; i < N; i++)
        if (f95 == -1.0f && TPR[i] >= 0.95f) f95 = FPR[i];
    *fpr95 = f95;

    /* contourArea of {(FPR_i,TPR_i)} + (1,0): 0.5*|sum prev.x*p.y - prev.y*p.x|
     * in double, prev starting at the last point [OpenCV-src shapedescr.cpp] */
    double a00 = 0.0;
    float px = 1.0f, py = 0.0f;
    for (int i = 0; i <= N; i++) {
        float x = (i < N) ? FPR[i] : 1.0f;
        float y = (i < N) ? TPR[i] : 0.0f;
        a00 += (double)px * y - (double)py * x;
        px = x; py = y;
    }
    *auc = fabs(a00 * 0.5);
    free(ord); free(TPR); free(FPR);
}

/* ------------------------------------------------------------------------- */
/* Whole trainer                                                             */
/* ------------------------------------------------------------------------- */
struct dlco_ref_ctx {
    const float   *dists;
    const uint8_t *labels;
    int N, F, B;
    float mu, gamma;
    int32_t *idx_pos, *idx_neg;
    int n_pos, n_neg, n_pos_trn, n_neg_trn;
    uint64_t rng;              /* RNG rng(2215), src/pj-learn.cpp:225 */
    unsigned t;
    int r;                     /* rows of W                          */
    float *W, *A, *dloss, *dfavg;
    float *Pb, *Nb;            /* DescDiff{Pos,Neg}Batch             */
    float *pd, *nd;
    int32_t *rows_pos, *rows_neg;
    int grad_order;
    double tm[4];              /* seconds: sample+project, gradient, RDA+dual->primal, PSD projection */
};

dlco_ref_ctx *dlco_ref_create(const float *dists, const uint8_t *labels,
                              int N, int F, int B, float mu, float gamma)
{
    dlco_ref_ctx *c = (dlco_ref_ctx *)calloc(1, sizeof(*c));
    size_t FF = (size_t)F * F;
    c->dists = dists; c->labels = labels; c->N = N; c->F = F; c->B = B;
    c->mu = mu; c->gamma = gamma;
    c->idx_pos = (int32_t *)malloc(sizeof(int32_t) * (size_t)(N > 0 ? N : 1));
    c->idx_neg = (int32_t *)malloc(sizeof(int32_t) * (size_t)(N > 0 ? N : 1));
    dlco_ref_build_index(labels, N, c->idx_pos, &c->n_pos, c->idx_neg, &c->n_neg);
    c->n_pos_trn = (int)dlco_ref_split((size_t)c->n_pos);
    c->n_neg_trn = (int)dlco_ref_split((size_t)c->n_neg);
    c->rng = 2215;
    c->t = 0;
    /* src/pj-learn.cpp:287-290: W = A = dLoss = dfAvg = zeros(F,F)           */
    c->W = (float *)calloc(FF, sizeof(float)); c->r = F;
    c->A = (float *)calloc(FF, sizeof(float));
    c->dloss = (float *)calloc(FF, sizeof(float));
    c->dfavg = (float *)calloc(FF, sizeof(float));
    c->Pb = (float *)malloc(sizeof(float) * (size_t)B * F);
    c->Nb = (float *)malloc(sizeof(float) * (size_t)B * F);
    c->pd = (float *)calloc(B, sizeof(float));
    c->nd = (float *)calloc(B, sizeof(float));
    c->rows_pos = (int32_t *)calloc(B, sizeof(int32_t));
    c->rows_neg = (int32_t *)calloc(B, sizeof(int32_t));
    c->grad_order = 0;
    return c;
}

void dlco_ref_destroy(dlco_ref_ctx *c)
{
    if (!c) return;
    free(c->idx_pos); free(c->idx_neg); free(c->W); free(c->A); free(c->dloss);
    free(c->dfavg); free(c->Pb); free(c->Nb); free(c->pd); free(c->nd);
    free(c->rows_pos); free(c->rows_neg); free(c);
}

void dlco_ref_set_grad_order(dlco_ref_ctx *c, int order) { c->grad_order = order; }

int dlco_ref_step(dlco_ref_ctx *c)
{
    const int B = c->B, F = c->F;
    double t0 = omp_get_wtime(), t1;
    /* sample (src/pj-learn.cpp:310-329)                                     */
    for (int k = 0; k < B; k++) {
        int ip = dlco_ref_rng_uniform(&c->rng, 0, c->n_pos_trn);
        int in = dlco_ref_rng_uniform(&c->rng, 0, c->n_neg_trn);
        c->rows_pos[k] = c->idx_pos[ip];
        c->rows_neg[k] = c->idx_neg[in];
        memcpy(c->Pb + (size_t)k * F, c->dists + (size_t)c->rows_pos[k] * F, sizeof(float) * F);
        memcpy(c->Nb + (size_t)k * F, c->dists + (size_t)c->rows_neg[k] * F, sizeof(float) * F);
    }
    /* distances (:332-365)                                                  */
    dlco_ref_project_sqdist(c->W, c->r, F, c->Pb, B, c->pd);
    dlco_ref_project_sqdist(c->W, c->r, F, c->Nb, B, c->nd);
    t1 = omp_get_wtime(); c->tm[0] += t1 - t0; t0 = t1;
    /* gradient (:367-418)                                                   */
    if (c->grad_order == 0) {
        dlco_ref_grad_reforder(c->Pb, c->Nb, c->pd, c->nd, B, F, c->dloss);
    } else {
        int32_t *rho = (int32_t *)malloc(sizeof(int32_t) * B), *kap = (int32_t *)malloc(sizeof(int32_t) * B);
        dlco_ref_viol_counts(c->pd, c->nd, B, rho, kap);
        dlco_ref_grad_reform(c->Pb, c->Nb, rho, kap, B, F, c->dloss);
        free(rho); free(kap);
    }
    t1 = omp_get_wtime(); c->tm[1] += t1 - t0; t0 = t1;
    /* RDA average, dual -> primal, PSD projection (:420-490)                */
    dlco_ref_rda_update(c->dfavg, c->dloss, c->t, (unsigned)B, F);
    dlco_ref_dual_to_primal(c->dfavg, c->mu, c->gamma, c->t, F, c->A);
    t1 = omp_get_wtime(); c->tm[2] += t1 - t0; t0 = t1;
    int rc = dlco_ref_psd_project(c->A, F, c->W, &c->r, NULL);
    t1 = omp_get_wtime(); c->tm[3] += t1 - t0;
    c->t++;
    return rc;
}

/* wall seconds spent since creation in: [0] sampling + P1/P2, [1] Q1 gradient, [2] U1/U2,
 * [3] E1/E2 (ssyevr + back-multiplication)                                                  */
void dlco_ref_get_timers(const dlco_ref_ctx *c, double out[4])
{
    for (int i = 0; i < 4; i++) out[i] = c->tm[i];
}

void dlco_ref_get_batch_ids(const dlco_ref_ctx *c, int32_t *pos_rows, int32_t *neg_rows)
{
    memcpy(pos_rows, c->rows_pos, sizeof(int32_t) * c->B);
    memcpy(neg_rows, c->rows_neg, sizeof(int32_t) * c->B);
}

void dlco_ref_get_batch_dists(const dlco_ref_ctx *c, float *pd, float *nd)
{
    memcpy(pd, c->pd, sizeof(float) * c->B);
    memcpy(nd, c->nd, sizeof(float) * c->B);
}

void dlco_ref_get_state(const dlco_ref_ctx *c, unsigned *t, int *r,
                        float *W, float *A, float *dfavg, float *dloss)
{
    size_t FF = (size_t)c->F * c->F;
    if (t) *t = c->t;
    if (r) *r = c->r;
    if (W) memcpy(W, c->W, sizeof(float) * (size_t)c->r * c->F);
    if (A) memcpy(A, c->A, sizeof(float) * FF);
    if (dfavg) memcpy(dfavg, c->dfavg, sizeof(float) * FF);
    if (dloss) memcpy(dloss, c->dloss, sizeof(float) * FF);
}

void dlco_ref_set_state(dlco_ref_ctx *c, unsigned t, const float *dfavg,
                        const float *W, int r)
{
    size_t FF = (size_t)c->F * c->F;
    c->t = t;
    if (dfavg) memcpy(c->dfavg, dfavg, sizeof(float) * FF);
    if (W) { memcpy(c->W, W, sizeof(float) * (size_t)r * c->F); c->r = r; }
}

void dlco_ref_get_index(const dlco_ref_ctx *c, int32_t *pos, int *n_pos, int *n_pos_trn,
                        int32_t *neg, int *n_neg, int *n_neg_trn)
{
    if (pos) memcpy(pos, c->idx_pos, sizeof(int32_t) * c->n_pos);
    if (neg) memcpy(neg, c->idx_neg, sizeof(int32_t) * c->n_neg);
    if (n_pos) *n_pos = c->n_pos;
    if (n_neg) *n_neg = c->n_neg;
    if (n_pos_trn) *n_pos_trn = c->n_pos_trn;
    if (n_neg_trn) *n_neg_trn = c->n_neg_trn;
}

/* src/pj-learn.cpp:501-527                                                   */
void dlco_ref_validate(const dlco_ref_ctx *c, float *loss_val, float *regul)
{
    int npv = c->n_pos - c->n_pos_trn, nnv = c->n_neg - c->n_neg_trn;
    float *pdv = (float *)malloc(sizeof(float) * (size_t)(npv > 0 ? npv : 1));
    float *ndv = (float *)malloc(sizeof(float) * (size_t)(nnv > 0 ? nnv : 1));
    dlco_ref_project_sqdist_ids(c->W, c->r, c->F, c->dists, c->idx_pos + c->n_pos_trn, npv, pdv);
    dlco_ref_project_sqdist_ids(c->W, c->r, c->F, c->dists, c->idx_neg + c->n_neg_trn, nnv, ndv);
    float Loss = (float)dlco_ref_hinge_sum(pdv, npv, ndv, nnv);      /* :520 */
    *loss_val = Loss / (float)npv / (float)nnv;                      /* :524 */
    *regul = (float)((double)c->mu * dlco_ref_trace(c->A, c->F));     /* :527: float mu * double trace */
    free(pdv); free(ndv);
}

void dlco_ref_stats(const dlco_ref_ctx *c, int *dim, float *fpr95, double *auc)
{
    float *nzW = (float *)malloc(sizeof(float) * (size_t)c->r * c->F);
    int d = dlco_ref_nonzero_rows(c->W, c->r, c->F, nzW);
    *dim = d;
    float *dist = (float *)malloc(sizeof(float) * (size_t)c->N);
    int32_t *ids = (int32_t *)malloc(sizeof(int32_t) * (size_t)c->N);
    for (int i = 0; i < c->N; i++) ids[i] = i;
    dlco_ref_project_sqdist_ids(nzW, d, c->F, c->dists, ids, c->N, dist);
    dlco_ref_roc_stats(dist, c->labels, c->N, fpr95, auc);
    free(nzW); free(dist); free(ids);
}


/* ========================================================================= */
/* pr-learn: L1-regularised RDA on the pooling-region weight  */
/* vector w, src/pr-learn.cpp:229-434, and ComputePRStats, src/misc.cpp:171-264 */
/* ========================================================================= */
struct dlco_ref_pr {
    const float   *dists;
    const uint8_t *labels;
    int N, F;
    float mu, gamma;
    int32_t *idx_pos, *idx_neg;
    int n_pos, n_neg, n_pos_trn, n_neg_trn;
    uint64_t rng;              /* RNG rng(2215), src/pr-learn.cpp:241 */
    unsigned t;
    float *w, *dfavg, *diff;
    int32_t last_pos, last_neg;
    float last_f;
};

dlco_ref_pr *dlco_ref_pr_create(const float *dists, const uint8_t *labels, int N, int F, float mu, float gamma)
{
    dlco_ref_pr *c = (dlco_ref_pr *)calloc(1, sizeof(*c));
    c->dists = dists; c->labels = labels; c->N = N; c->F = F; c->mu = mu; c->gamma = gamma;
    c->idx_pos = (int32_t *)malloc(sizeof(int32_t) * (size_t)(N > 0 ? N : 1));
    c->idx_neg = (int32_t *)malloc(sizeof(int32_t) * (size_t)(N > 0 ? N : 1));
    /* :229-243: the same index build and randShuffle-on-theRNG() pattern as pj-learn */
    dlco_ref_build_index(labels, N, c->idx_pos, &c->n_pos, c->idx_neg, &c->n_neg);
    c->n_pos_trn = (int)dlco_ref_split((size_t)c->n_pos);       /* :251-252 */
    c->n_neg_trn = (int)dlco_ref_split((size_t)c->n_neg);
    c->rng = 2215;
    c->w = (float *)calloc((size_t)F, sizeof(float));           /* :195-196 */
    c->dfavg = (float *)calloc((size_t)F, sizeof(float));
    c->diff = (float *)calloc((size_t)F, sizeof(float));
    return c;
}

void dlco_ref_pr_destroy(dlco_ref_pr *c)
{
    if (!c) return;
    free(c->idx_pos); free(c->idx_neg); free(c->w); free(c->dfavg); free(c->diff); free(c);
}

/* One iteration of the loop body at src/pr-learn.cpp:302-329, single-threaded order (with more
 * than one OpenMP thread the reference draws its samples in thread-arrival order, :306-310, so only
 * the one-thread order is reproducible).  OpenCV semantics restated [OpenCV-src]:
 *   gemm(w, FeatDiff, .., GEMM_2_T) on CV_32F accumulates in double and rounds once to float;
 *   `dfAvg = t * dfAvg / (t + 1)` is a MatExpr scaled by alpha = t * (1.0 / (t + 1)) (double),
 *     applied by convertTo as float(x * (float)alpha);
 *   scaleAdd(FeatDiff, 1/(t+1), dfAvg, dfAvg) = FeatDiff * (float)(1.0/(t+1)) + dfAvg in float
 *     (taken unfused; an FMA build of OpenCV may differ in the last bit);
 *   `w = -sqrt(t+1)/gamma * (dfAvg + mu)` = dfAvg * (float)a + (float)(mu * a), a in double, unfused;
 *   max(w, 0).                                                                              */
void dlco_ref_pr_step(dlco_ref_pr *c)
{
    const int F = c->F;
    const unsigned t = c->t;
    int ip = dlco_ref_rng_uniform(&c->rng, 0, c->n_pos_trn);
    int in = dlco_ref_rng_uniform(&c->rng, 0, c->n_neg_trn);
    const float *xp = c->dists + (size_t)c->idx_pos[ip] * F, *xn = c->dists + (size_t)c->idx_neg[in] * F;
    c->last_pos = c->idx_pos[ip]; c->last_neg = c->idx_neg[in];
    double acc = 0.0;
    for (int k = 0; k < F; k++) {
        float d = xp[k] - xn[k];                               /* subtract(), :312-314 */
        c->diff[k] = d;
        acc += (double)c->w[k] * (double)d;                    /* gemm, :319 */
    }
    const float f = (float)acc;
    c->last_f = f;
    const float sa = (float)((double)t * (1.0 / ((double)t + 1.0)));         /* :322 */
    for (int k = 0; k < F; k++) c->dfavg[k] = c->dfavg[k] * sa;
    if (f > -1.0f) {                                                             /* :324-325 */
        const float al = (float)(1.0 / ((double)t + 1.0));
        for (int k = 0; k < F; k++) {
            volatile float prod = c->diff[k] * al;
            c->dfavg[k] = prod + c->dfavg[k];
        }
    }
    const double a = -sqrt((double)t + 1.0) / (double)c->gamma;                  /* :328 */
    const float fa = (float)a, fb = (float)((double)c->mu * a);
    for (int k = 0; k < F; k++) {
        volatile float prod = c->dfavg[k] * fa;
        float v = prod + fb;
        c->w[k] = v > 0.0f ? v : 0.0f;                                            /* :329 */
    }
    c->t = t + 1;
}

void dlco_ref_pr_steps(dlco_ref_pr *c, unsigned n) { for (unsigned i = 0; i < n; i++) dlco_ref_pr_step(c); }

void dlco_ref_pr_get(const dlco_ref_pr *c, unsigned *t, float *w, float *dfavg, int32_t *last_pos, int32_t *last_neg, float *last_f)
{
    if (t) *t = c->t;
    if (w) memcpy(w, c->w, sizeof(float) * (size_t)c->F);
    if (dfavg) memcpy(dfavg, c->dfavg, sizeof(float) * (size_t)c->F);
    if (last_pos) *last_pos = c->last_pos;
    if (last_neg) *last_neg = c->last_neg;
    if (last_f) *last_f = c->last_f;
}

void dlco_ref_pr_set(dlco_ref_pr *c, unsigned t, const float *w, const float *dfavg)
{
    c->t = t;
    if (w) memcpy(c->w, w, sizeof(float) * (size_t)c->F);
    if (dfavg) memcpy(c->dfavg, dfavg, sizeof(float) * (size_t)c->F);
}

/* dist[i] = w . x_i (float accumulation in index order: cuda::gemm / cuBLAS order is unspecified,
 * src/pr-learn.cpp:343-344; CPU gemm of ComputePRStats accumulates in double, src/misc.cpp:226) */
static void pr_gemv(const float *w, const float *D, const int32_t *ids, int n, int F, float *out, int in_double)
{
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n; i++) {
        const float *x = D + (size_t)(ids ? ids[i] : i) * F;
        if (in_double) {
            double s = 0.0;
            for (int k = 0; k < F; k++) s += (double)w[k] * (double)x[k];
            out[i] = (float)s;
        } else {
            float s = 0.0f;
            for (int k = 0; k < F; k++) s += w[k] * x[k];
            out[i] = s;
        }
    }
}

/* src/pr-learn.cpp:340-361: validation loss, regulariser mu * sum|w|, NNZ */
void dlco_ref_pr_validate(const dlco_ref_pr *c, float *loss_val, float *regul, int *nnz)
{
    int npv = c->n_pos - c->n_pos_trn, nnv = c->n_neg - c->n_neg_trn;
    float *pdv = (float *)malloc(sizeof(float) * (size_t)(npv > 0 ? npv : 1));
    float *ndv = (float *)malloc(sizeof(float) * (size_t)(nnv > 0 ? nnv : 1));
    pr_gemv(c->w, c->dists, c->idx_pos + c->n_pos_trn, npv, c->F, pdv, 0);
    pr_gemv(c->w, c->dists, c->idx_neg + c->n_neg_trn, nnv, c->F, ndv, 0);
    float Loss = (float)dlco_ref_hinge_sum(pdv, npv, ndv, nnv);                  /* :348-351 */
    *loss_val = Loss / (float)((size_t)npv * (size_t)nnv);                       /* :355 */
    double s = 0.0; int nz = 0;
    for (int k = 0; k < c->F; k++) { s += fabs((double)c->w[k]); nz += c->w[k] != 0.0f; }
    *regul = (float)((double)c->mu * s);                                         /* :358 */
    if (nnz) *nnz = nz;
    free(pdv); free(ndv);
}

/* ComputePRStats, src/misc.cpp:171-264.  PRParams [8*F, pr_cols]; returns Dim > MaxDim early like
 * the reference when max_dim != -1 (fpr95/auc untouched).                                     */
void dlco_ref_pr_stats(const float *prparams, int pr_cols, const float *dists, const uint8_t *labels, int N, int F,
                       const float *w, int nchannels, int max_dim, int *nPR, int *Dim, int *nzDim, float *fpr95, double *auc)
{
    /* rows of PRParams selected by w > 0 that have a non-zero entry (:183-193) */
    int cap = 8 * F, nsel = 0;
    const float **sel = (const float **)malloc(sizeof(float *) * (size_t)cap);
    for (int i = 0; i < F; i++)
        for (int j = 0; j < 8; j++) {
            const float *row = prparams + (size_t)(i * 8 + j) * pr_cols;
            if (!(w[i] > 0.0f)) continue;
            int any = 0;
            for (int k = 0; k < pr_cols; k++) any |= row[k] != 0.0f;
            if (any) sel[nsel++] = row;
        }
    /* rows that have an identical twin elsewhere (:196-213); nPR = nzDim - dup/2 */
    int dup = 0;
    for (int i = 0; i < nsel; i++) {
        int inside = 0;
        for (int j = 0; j < nsel && !inside; j++) {
            if (i == j) continue;
            int same = 1;
            for (int k = 0; k < pr_cols && same; k++) same = sel[i][k] == sel[j][k];
            inside = same;
        }
        dup += inside;
    }
    free(sel);
    *nzDim = nsel;
    *nPR = nsel - dup / 2;
    *Dim = *nPR * nchannels;
    if (max_dim != -1 && *Dim > max_dim) return;
    float *pd = (float *)malloc(sizeof(float) * (size_t)N);
    pr_gemv(w, dists, NULL, N, F, pd, 1);                                        /* :226 */
    dlco_ref_roc_stats(pd, labels, N, fpr95, auc);                               /* :227-263, same sweep as the PJ stage */
    free(pd);
}


/* ========================================================================= */
/* Descriptor generation: get_desc, src/vgg-desc.cpp:41-152,  */
/* and the per-pair loop of comp-uprjdists, src/comp-uprjdists.cpp:298-349     */
/* ========================================================================= */
/* OpenCV primitives restated [OpenCV-src] (parity unpinned against a live build):
 *   GaussianBlur(Size(0,0), sigma) on CV_32F: ksize = cvRound(sigma*4*2 + 1) | 1; kernel
 *     cf[i] = (float)exp(-0.5/sigma^2 * x^2), normalised by the float sum's reciprocal (double);
 *     separable, float intermediate, BORDER_REPLICATE; the row pass adds the taps left to right,
 *     the column pass in symmetric form k0*c + sum_k k_k*(up_k + down_k);
 *   filter2D with [-1 0 1]: right neighbour minus left neighbour, BORDER_REPLICATE;
 *   magnitude: sqrtf(x*x + y*y), products and sum in float, not fused;
 *   `GAngle / AngleStep - 0.5f` and `GMag /= s`: MatExpr scalings by (float)(1.0 / s) in float;
 *   cv::sort ascending; mquantiles(alphap = betap = 0.5) as written at :113-130.                  */
#define DESC_P 64                 /* patch edge */
#define DESC_NPIX (DESC_P * DESC_P)

static int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
static int cmp_float_asc(const void *a, const void *b)
{
    float x = *(const float *)a, y = *(const float *)b;
    return (x > y) - (x < y);
}

/* PatchTrans [4096][nAngleBins] (row p = x*64 + y: the reference transposes before filling, :136-150) */
void dlco_ref_get_desc(const uint8_t *patch, int nAngleBins, float InitSigma, int bNorm, float *PatchTrans)
{
    enum { P = DESC_P, NP = DESC_NPIX };
    static const double kPi = 3.1415926535897932384626433832795;        /* CV_PI */
    float *img = (float *)malloc(sizeof(float) * NP), *tmp = (float *)malloc(sizeof(float) * NP);
    float *mag = (float *)malloc(sizeof(float) * NP), *ratio = (float *)malloc(sizeof(float) * NP);
    for (int i = 0; i < NP; i++) img[i] = (float)patch[i];                /* convertTo CV_32F, :44 */
    /* ---- GaussianBlur, :46 ---- */
    int ks = (int)lrint((double)InitSigma * 4.0 * 2.0 + 1.0) | 1;
    if (ks > 63) ks = 63;
    float cf[64];
    {
        double scale2x = -0.5 / ((double)InitSigma * (double)InitSigma), sum = 0.0;
        for (int i = 0; i < ks; i++) {
            double x = i - (ks - 1) * 0.5;
            cf[i] = (float)exp(scale2x * x * x);
            sum += cf[i];
        }
        sum = 1.0 / sum;
        for (int i = 0; i < ks; i++) cf[i] = (float)(cf[i] * sum);
    }
    const int r = ks / 2;
    for (int y = 0; y < P; y++)
        for (int x = 0; x < P; x++) {
            float s = cf[0] * img[y * P + clampi(x - r, 0, P - 1)];
            for (int k = 1; k < ks; k++) { volatile float pr = cf[k] * img[y * P + clampi(x - r + k, 0, P - 1)]; s += pr; }
            tmp[y * P + x] = s;
        }
    for (int y = 0; y < P; y++)
        for (int x = 0; x < P; x++) {
            float s = cf[r] * tmp[y * P + x];
            for (int k = 1; k <= r; k++) {
                volatile float pair = tmp[clampi(y - k, 0, P - 1) * P + x] + tmp[clampi(y + k, 0, P - 1) * P + x];
                volatile float pr = cf[r + k] * pair;
                s += pr;
            }
            img[y * P + x] = s;
        }
    /* ---- gradient, magnitude, orientation ratio, :48-70 ---- */
    const float AngleStep = (float)(2.0f * kPi / (double)(float)nAngleBins);
    const float inv_step = (float)(1.0 / (double)AngleStep);
    for (int y = 0; y < P; y++)
        for (int x = 0; x < P; x++) {
            float ix = img[y * P + clampi(x + 1, 0, P - 1)] - img[y * P + clampi(x - 1, 0, P - 1)];
            float iy = img[clampi(y + 1, 0, P - 1) * P + x] - img[clampi(y - 1, 0, P - 1) * P + x];
            volatile float xx = ix * ix, yy = iy * iy;
            mag[y * P + x] = sqrtf(xx + yy);
            /* atan2 on floats rounded to float: taken from the double routine, whose rounding to float is the
             * correctly rounded single result (glibc 2.35's atan2f itself is only accurate to < 1 ulp)       */
            float at = (float)atan2((double)iy, (double)ix);
            float ang = (float)((double)at + kPi);
            volatile float sc = ang * inv_step;
            ratio[y * P + x] = sc - 0.5f;
        }
    /* ---- quantile normalisation, :106-133 ---- */
    if (bNorm) {
        memcpy(tmp, mag, sizeof(float) * NP);
        qsort(tmp, NP, sizeof(float), cmp_float_asc);
        const int n = NP;
        float aleph = (float)n * 0.8f + 0.5f;
        int k = (int)floorf(aleph);
        if (k >= n - 1) k = n - 1;
        if (k <= 1) k = 1;
        float gamma = aleph - (float)k;
        if (gamma >= 1.0f) gamma = 1.0f;
        if (gamma <= 0.0f) gamma = 0.0f;
        volatile float t1 = (1.0f - gamma) * tmp[k - 1], t2 = gamma * tmp[k];
        float T = t1 + t2;
        if (T != 0.0f) {
            const float sc = (float)(1.0 / (double)(T / (float)nAngleBins));
            for (int i = 0; i < NP; i++) mag[i] = mag[i] * sc;
        }
    }
    /* ---- soft assignment into the transposed layout, :72-104,136-150 ---- */
    memset(PatchTrans, 0, sizeof(float) * (size_t)NP * nAngleBins);
    for (int y = 0; y < P; y++)
        for (int x = 0; x < P; x++) {
            const float rt = ratio[y * P + x];
            const float off = rt - floorf(rt);
            int b1 = (ceilf(rt - 1.0f) == -1.0f) ? nAngleBins - 1 : (int)(unsigned char)ceilf(rt - 1.0f);
            int b2 = (b1 + 1 > nAngleBins - 1) ? 0 : b1 + 1;
            const int p = x * P + y;                           /* .t() */
            const float m = mag[y * P + x];
            if (b1 >= 0 && b1 < nAngleBins) PatchTrans[(size_t)p * nAngleBins + b1] = (1.0f - off) * m;
            if (b2 >= 0 && b2 < nAngleBins) PatchTrans[(size_t)p * nAngleBins + b2] = off * m;
        }
    free(img); free(tmp); free(mag); free(ratio);
}

/* Desc = min(sPRFilters [nsel,4096] * PatchTrans [4096,8], 1), row-major [nsel*8]
 * (src/comp-uprjdists.cpp:320-325; cv::gemm accumulates in double for CV_32F)                    */
void dlco_ref_patch_descriptor(const uint8_t *patch, const float *sPR, int nsel, float *desc)
{
    enum { NB = 8 };
    float *pt = (float *)malloc(sizeof(float) * (size_t)DESC_NPIX * NB);
    dlco_ref_get_desc(patch, NB, 1.4f, 1, pt);
#pragma omp parallel for schedule(static)
    for (int i = 0; i < nsel; i++) {
        double acc[NB] = {0};
        const float *f = sPR + (size_t)i * DESC_NPIX;
        for (int p = 0; p < DESC_NPIX; p++) {
            if (f[p] == 0.0f) continue;
            for (int b = 0; b < NB; b++) acc[b] += (double)f[p] * (double)pt[(size_t)p * NB + b];
        }
        for (int b = 0; b < NB; b++) { float v = (float)acc[b]; desc[(size_t)i * NB + b] = v < 1.0f ? v : 1.0f; }
    }
    free(pt);
}
