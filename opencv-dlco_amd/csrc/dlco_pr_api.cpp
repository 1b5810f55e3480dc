// dlco_pr_api.cpp — C ABI of the pooling-region stage (pr-learn): see the dlco_pr_*
// block of include/dlco.h.  Host orchestration only; the iterations, the validation products and the
// statistics run in HIP kernels (kernels_pr.hip, kernels_step.hip, kernels_stats.hip).
#include "../../include/dlco.h"

#include "dlco_internal.hpp"
#include "pair_index.hpp"

#include <algorithm>
#include <cmath>
#include <cstring>

using namespace dlco;

namespace dlco {
bool pr_steps(const float *D, long ld, const int32_t *pos_rows, const int32_t *neg_rows, unsigned n, unsigned t0, float mu, float gamma,
              int F, float *w_io, float *df_io, float *last_f, hipStream_t s);
void pr_gemv(const float *D, long ld, const int32_t *ids, int n, const float *w, int F, float *out, bool in_double, hipStream_t s);
}

struct dlco_pr_ctx {
    int F = 0, N = 0, device = 0;
    float mu = 0.025f, gamma = 0.10f;
    hipStream_t stream = nullptr;
    std::string err;
    DevBuf<float> dists, w, df, vdist, hrows, wtmp;
    DevBuf<uint8_t> labels_dev;
    DevBuf<int32_t> val_pos_ids, val_neg_ids, win_pos, win_neg;
    DevBuf<double> dscal;
    std::vector<uint8_t> labels;
    std::vector<int32_t> h_pos, h_neg;
    PairIndex idx;
    CvRng rng{2215};
    uint32_t t = 0;
    bool have_data = false;
    RocWork *roc = nullptr;
    size_t win_cap = 0;
};

static thread_local std::string g_pr_error;

namespace {

template <typename Fn>
int guarded(dlco_pr_ctx *c, Fn &&fn)
{
    try {
        fn();
        return DLCO_OK;
    } catch (const Error &e) {
        g_pr_error = e.what();
        if (c) c->err = e.what();
        return e.code;
    } catch (const std::exception &e) {
        g_pr_error = e.what();
        if (c) c->err = e.what();
        return DLCO_ERR_INVALID;
    }
}

void sync(dlco_pr_ctx *c) { DLCO_HIP(hipStreamSynchronize(c->stream)); }
void h2d(dlco_pr_ctx *c, void *dst, const void *src, size_t bytes)
{
    DLCO_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stream));
    sync(c);
}
void d2h(dlco_pr_ctx *c, void *dst, const void *src, size_t bytes)
{
    DLCO_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream));
    sync(c);
}

}  // namespace

extern "C" {

const char *dlco_pr_last_error(const dlco_pr_ctx *c) { return c ? c->err.c_str() : g_pr_error.c_str(); }

int dlco_pr_create(dlco_pr_ctx **out, int32_t F, int32_t N, float mu, float gamma, uint64_t seed, int32_t device)
{
    if (!out) return DLCO_ERR_INVALID;
    *out = nullptr;
    dlco_pr_ctx *c = nullptr;
    const int rc = guarded(nullptr, [&] {
        DLCO_CHECK(F >= 4 && F % 4 == 0 && F <= 8192, DLCO_ERR_INVALID, "pr-learn: F must be a multiple of 4, at most 8192");
        DLCO_CHECK(N >= 2 && gamma > 0.f, DLCO_ERR_INVALID, "pr-learn: N >= 2 and gamma > 0 required");
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device >= ndev)
            throw Error(DLCO_ERR_NODEVICE, "no usable HIP device (this library has no CPU fallback)");
        DLCO_HIP(hipSetDevice(device));
        hipDeviceProp_t prop;
        DLCO_HIP(hipGetDeviceProperties(&prop, device));
        if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
            throw Error(DLCO_ERR_NODEVICE, std::string("device is ") + prop.gcnArchName + ", this build targets gfx950 only");
        c = new dlco_pr_ctx();
        c->F = F; c->N = N; c->mu = mu; c->gamma = gamma; c->device = device;
        c->rng = CvRng(seed);
        DLCO_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
        c->w.alloc(F); c->df.alloc(F); c->wtmp.alloc(F);
        c->w.zero(c->stream); c->df.zero(c->stream);              // src/pr-learn.cpp:195-196
        c->dscal.alloc(4);
        sync(c);
    });
    if (rc != DLCO_OK) { delete c; return rc; }
    *out = c;
    return DLCO_OK;
}

void dlco_pr_destroy(dlco_pr_ctx *c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->roc) roc_work_destroy(c->roc);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

// Uploads Distance [N,F] and Label [N]; index build, shuffle and 80/20 split as src/pr-learn.cpp:229-253
int dlco_pr_set_data(dlco_pr_ctx *c, const float *dists_host, const uint8_t *labels_host)
{
    if (!c || !dists_host || !labels_host) return DLCO_ERR_INVALID;
    return guarded(c, [&] {
        DLCO_HIP(hipSetDevice(c->device));
        c->dists.alloc((size_t)c->N * c->F);
        h2d(c, c->dists.p, dists_host, (size_t)c->N * c->F * sizeof(float));
        c->labels.assign(labels_host, labels_host + c->N);
        c->labels_dev.alloc(c->N);
        h2d(c, c->labels_dev.p, c->labels.data(), c->N);
        c->idx.build(c->labels.data(), c->N);
        const int npv = (int)c->idx.pos.size() - c->idx.n_pos_trn, nnv = (int)c->idx.neg.size() - c->idx.n_neg_trn;
        c->val_pos_ids.alloc(std::max(npv, 1)); c->val_neg_ids.alloc(std::max(nnv, 1));
        if (npv > 0) h2d(c, c->val_pos_ids.p, c->idx.pos.data() + c->idx.n_pos_trn, (size_t)npv * sizeof(int32_t));
        if (nnv > 0) h2d(c, c->val_neg_ids.p, c->idx.neg.data() + c->idx.n_neg_trn, (size_t)nnv * sizeof(int32_t));
        c->vdist.alloc((size_t)c->N + 16);
        c->hrows.alloc((size_t)std::max(npv, 1));
        if (c->roc) roc_work_destroy(c->roc);
        c->roc = roc_work_create(c->N);
        c->have_data = true;
    });
}

int dlco_pr_get_index(const dlco_pr_ctx *c, int32_t *n_pos, int32_t *n_pos_trn, int32_t *n_neg, int32_t *n_neg_trn)
{
    if (!c || !c->have_data) return DLCO_ERR_INVALID;
    if (n_pos) *n_pos = (int32_t)c->idx.pos.size();
    if (n_neg) *n_neg = (int32_t)c->idx.neg.size();
    if (n_pos_trn) *n_pos_trn = c->idx.n_pos_trn;
    if (n_neg_trn) *n_neg_trn = c->idx.n_neg_trn;
    return DLCO_OK;
}

// n iterations of the loop body, src/pr-learn.cpp:302-329 (one launch; the sample list of the window is
// drawn on the host from the reference's generator, iPos then iNeg per iteration)
int dlco_pr_steps(dlco_pr_ctx *c, uint32_t n)
{
    if (!c) return DLCO_ERR_INVALID;
    return guarded(c, [&] {
        DLCO_HIP(hipSetDevice(c->device));
        DLCO_CHECK(c->have_data, DLCO_ERR_INVALID, "dlco_pr_steps: no data set");
        DLCO_CHECK(c->idx.n_pos_trn > 0 && c->idx.n_neg_trn > 0, DLCO_ERR_INVALID, "dlco_pr_steps: empty training split");
        if (n == 0) return;
        if (n > c->win_cap) {
            sync(c);
            c->win_cap = std::max<size_t>(n, 1024);
            c->win_pos.alloc(c->win_cap); c->win_neg.alloc(c->win_cap);
        }
        c->h_pos.resize(n); c->h_neg.resize(n);
        for (uint32_t s = 0; s < n; s++) {
            const int ip = c->rng.uniform(0, c->idx.n_pos_trn);
            const int in = c->rng.uniform(0, c->idx.n_neg_trn);
            c->h_pos[s] = c->idx.pos[ip];
            c->h_neg[s] = c->idx.neg[in];
        }
        h2d(c, c->win_pos.p, c->h_pos.data(), (size_t)n * sizeof(int32_t));
        h2d(c, c->win_neg.p, c->h_neg.data(), (size_t)n * sizeof(int32_t));
        const bool ok = pr_steps(c->dists.p, c->F, c->win_pos.p, c->win_neg.p, n, c->t, c->mu, c->gamma, c->F, c->w.p, c->df.p, nullptr,
                                 c->stream);
        DLCO_CHECK(ok, DLCO_ERR_INVALID, "dlco_pr_steps: shape not supported");
        c->t += n;
    });
}

int dlco_pr_get_state(dlco_pr_ctx *c, uint32_t *t, float *w_host, float *dfavg_host)
{
    if (!c) return DLCO_ERR_INVALID;
    return guarded(c, [&] {
        if (t) *t = c->t;
        if (w_host) d2h(c, w_host, c->w.p, (size_t)c->F * sizeof(float));
        if (dfavg_host) d2h(c, dfavg_host, c->df.p, (size_t)c->F * sizeof(float));
    });
}

int dlco_pr_set_state(dlco_pr_ctx *c, uint32_t t, const float *w_host, const float *dfavg_host)
{
    if (!c) return DLCO_ERR_INVALID;
    return guarded(c, [&] {
        c->t = t;
        if (w_host) h2d(c, c->w.p, w_host, (size_t)c->F * sizeof(float));
        if (dfavg_host) h2d(c, c->df.p, dfavg_host, (size_t)c->F * sizeof(float));
    });
}

// src/pr-learn.cpp:340-361: validation objective on the current w
int dlco_pr_validate(dlco_pr_ctx *c, float *loss_val, float *regul, int32_t *nnz)
{
    if (!c || !loss_val || !regul) return DLCO_ERR_INVALID;
    return guarded(c, [&] {
        DLCO_HIP(hipSetDevice(c->device));
        DLCO_CHECK(c->have_data, DLCO_ERR_INVALID, "dlco_pr_validate: no data set");
        const int npv = (int)c->idx.pos.size() - c->idx.n_pos_trn, nnv = (int)c->idx.neg.size() - c->idx.n_neg_trn;
        DLCO_CHECK(npv > 0 && nnv > 0, DLCO_ERR_INVALID, "dlco_pr_validate: empty validation split");
        float *pdv = c->vdist.p, *ndv = c->vdist.p + npv;
        pr_gemv(c->dists.p, c->F, c->val_pos_ids.p, npv, c->w.p, c->F, pdv, false, c->stream);     // :343
        pr_gemv(c->dists.p, c->F, c->val_neg_ids.p, nnv, c->w.p, c->F, ndv, false, c->stream);     // :344
        hinge_rows(pdv, npv, ndv, nnv, c->hrows.p, c->stream);                                      // :348
        sum_f32_to_f64(c->hrows.p, npv, c->dscal.p, c->stream);                                     // :351
        double total = 0.0;
        d2h(c, &total, c->dscal.p, sizeof(double));
        const float Loss = (float)total;
        *loss_val = Loss / (float)((size_t)npv * (size_t)nnv);                                      // :355
        std::vector<float> w(c->F);
        d2h(c, w.data(), c->w.p, (size_t)c->F * sizeof(float));
        double s = 0.0;
        int nz = 0;
        for (float v : w) { s += std::fabs((double)v); nz += v != 0.0f; }
        *regul = (float)((double)c->mu * s);                                                        // :358
        if (nnz) *nnz = nz;
    });
}

// ComputePRStats (src/misc.cpp:171-264): pooling-region counts on the host, distances + ROC on the device
int dlco_pr_stats(dlco_pr_ctx *c, const float *w_host, const float *prparams_host, int32_t pr_rows, int32_t pr_cols, int32_t nchannels,
                  int32_t max_dim, int32_t *nPR, int32_t *dim, int32_t *nzdim, float *fpr95, double *auc)
{
    if (!c || !prparams_host || !nPR || !dim || !nzdim || !fpr95 || !auc) return DLCO_ERR_INVALID;
    return guarded(c, [&] {
        DLCO_HIP(hipSetDevice(c->device));
        DLCO_CHECK(c->have_data, DLCO_ERR_INVALID, "dlco_pr_stats: no data set");
        DLCO_CHECK(pr_rows >= 8 * c->F && pr_cols >= 1, DLCO_ERR_INVALID, "dlco_pr_stats: PRParams needs 8 rows per weight");
        std::vector<float> w(c->F);
        if (w_host) std::memcpy(w.data(), w_host, (size_t)c->F * sizeof(float));
        else d2h(c, w.data(), c->w.p, (size_t)c->F * sizeof(float));
        std::vector<const float *> sel;                                     // :183-193
        for (int i = 0; i < c->F; i++)
            for (int j = 0; j < 8; j++) {
                const float *row = prparams_host + (size_t)(i * 8 + j) * pr_cols;
                if (!(w[i] > 0.0f)) continue;
                bool any = false;
                for (int k = 0; k < pr_cols && !any; k++) any = row[k] != 0.0f;
                if (any) sel.push_back(row);
            }
        int dup = 0;                                                         // :196-213
        for (size_t i = 0; i < sel.size(); i++) {
            bool inside = false;
            for (size_t j = 0; j < sel.size() && !inside; j++) {
                if (i == j) continue;
                bool same = true;
                for (int k = 0; k < pr_cols && same; k++) same = sel[i][k] == sel[j][k];
                inside = same;
            }
            dup += inside ? 1 : 0;
        }
        *nzdim = (int32_t)sel.size();
        *nPR = *nzdim - dup / 2;
        *dim = *nPR * nchannels;
        if (max_dim != -1 && *dim > max_dim) return;                        // :219-221
        const float *wd = c->w.p;
        if (w_host) { h2d(c, c->wtmp.p, w.data(), (size_t)c->F * sizeof(float)); wd = c->wtmp.p; }
        pr_gemv(c->dists.p, c->F, nullptr, c->N, wd, c->F, c->vdist.p, true, c->stream);             // :226
        roc_stats(c->roc, c->vdist.p, c->labels_dev.p, c->N, fpr95, auc, c->stream);                 // :227-263
    });
}

int dlco_pr_device_name(const dlco_pr_ctx *c, char *buf, size_t cap, int *cc_major, int *cc_minor)
{
    if (!c || !buf || cap == 0) return DLCO_ERR_INVALID;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, c->device) != hipSuccess) return DLCO_ERR_HIP;
    std::snprintf(buf, cap, "%s (%s)", prop.name, prop.gcnArchName);
    if (cc_major) *cc_major = prop.major;
    if (cc_minor) *cc_minor = prop.minor;
    return DLCO_OK;
}

}  // extern "C"
