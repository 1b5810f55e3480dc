// dlco_api.cpp — C ABI of libdlco.so (see include/dlco.h).  Host orchestration of the
// pj-learn step on one MI355X; every arithmetic stage runs in a HIP kernel of this library.
#include "../../include/dlco.h"

#include "dlco_internal.hpp"
#include "eig_tracker.hpp"
#include "pair_index.hpp"
#include "rccl_comm.hpp"
#include "host_comm.hpp"

#include <algorithm>
#include <cfloat>
#include <chrono>
#include <cmath>
#include <cstring>

using namespace dlco;

static thread_local std::string g_last_error;

struct dlco_ctx {
    dlco_cfg cfg{};
    // F is the width every device buffer and kernel works at: the caller's FeatDim (Fu, src/pj-learn.cpp:176-179) rounded up
    // to the 128-column tile of the fused kernels, the extra columns zero.  Zero columns add zero rows / columns to the dual
    // average, whose eigenvalue in A = -c (dfAvg + mu I) is -c mu <= 0: never kept (src/pj-learn.cpp:452,480-484), so every
    // result restricted to the first Fu columns is that of the unpadded computation; the ABI converts at its boundary.
    int F = 0, Fu = 0, N = 0, B = 0, Bl = 0, lo = 0;
    hipStream_t stream = nullptr;
    std::string err;
    hipDeviceProp_t prop{};

    // data
    DevBuf<float> dists_own;
    const float *dists = nullptr;
    std::vector<uint8_t> labels;
    DevBuf<uint8_t> labels_dev;
    PairIndex idx;
    DevBuf<int32_t> val_pos_ids, val_neg_ids;
    bool have_data = false;
    // pair mode (dlco_set_pairs): `dists` holds the P per-patch descriptors and row i of the
    // training matrix is dists[pair_a[i]] - dists[pair_b[i]], formed inside the kernels
    bool pair_mode = false;
    int P = 0;
    DevBuf<int32_t> pair_a, pair_b, tr_a, tr_b;
    size_t tr_cap = 0;

    // training state
    CvRng rng{2215};
    uint32_t t = 0;
    int r = 0;                       // rows of W (0 = the reference's all-zero F x F W)
    double traceA = 0.0;
    DevBuf<float> dfavg, grad, W;
    bool packed = false;             // dfavg holds the packed upper tiles (syrk_packed_floats(F) floats), not the F x F matrix
    int w_cap = 0;
    std::vector<int32_t> h_pos_rows, h_neg_rows;
    int32_t *pin_ids = nullptr;      // pinned staging of the sampled row ids: [2 slots][2B], no sync after the upload
    uint32_t upload_ctr = 0;
    // sampled row ids on the device, one allocation [pos: B | neg: B | own pos slots: Bl | own neg slots: Bl]
    // filled by a single upload per step
    struct IdView { int32_t *p = nullptr; };
    DevBuf<int32_t> ids_all;
    IdView pos_rows, neg_rows, local_ids;
    const float *pd_cur = nullptr, *nd_cur = nullptr;    // distance vectors of the current step (see gather_dists)
    DevBuf<int32_t> rho, kappa, act_ids, seed_ids, act_slot;
    // the step's first filter term from its own rank update (kernels_rankupd.hip): the batch projection also covers the
    // tracker's guard rows, and what the tracker needs of the step is handed over after the gradient
    bool rank_update = false;
    int planes_mode = 0;             // syrk_planes_mode(): which planes the gradient leaves for the rank update
    const float *ru_proj = nullptr;  // this step's projection [ru_rows][2 Bl] of the batch on the rows of W (guards included)
    int ru_rows = 0;
    DevBuf<float> act_w, seed_w, dist_x, pd, nd, proj_slab, vproj, vdist, hrows;
    DevBuf<char> pplane[3];          // split-bf16 planes of W for the fused many-row projection (kernels_project.hip)
    float *xdist = nullptr, *xgrad = nullptr;   // exchange buffers (own allocations unless bound by the caller)
    DevBuf<int> k_active;
    DevBuf<double> dscal;
    size_t proj_slab_floats = 0;
    int phase = 0;                   // 0 idle, 1 after begin, 2 after grad
    EigTracker *eig = nullptr;
    DevBuf<char> syrk_planes;                 // the gradient's split row planes (kernels_syrk.hip): one workspace per context
    RocWork *roc = nullptr;
    int64_t nonconv_steps = 0, steps_run = 0, active_rows_sum = 0;
    int32_t nonconv_window = 0;      // non-converged steps since the last dlco_log_step

    // model selection (src/pj-learn.cpp:229-232)
    double auc_best = 0.0;
    float obj_best = FLT_MAX, fpr95_best = FLT_MAX;
    int r_best = 0;
    // W_Save stays in HBM; A_Save = W_Save^T W_Save is only formed when the caller asks for it (dlco_get_saved):
    // cloning an F x F matrix to the host at every "[saved]" line cost more than the statistics pass itself
    DevBuf<float> W_save_dev;
    int r_save = 0;                  // rows the caller sees (F for the reference's all-zero W)
    int r_save_dev = 0;              // rows held in W_save_dev (0 for the all-zero W)

    // HIP-event timers (gradient SYRK, tracker products, ...)
    Profiler prof;

    // column-sharded dual average (cfg.shard, world > 1): see dlco_set_allgather
    bool shard = false;
    ShardComm comm;
    DevBuf<float> gather_own;
    dlco_allgather_fn ag_fn = nullptr;
    void *ag_user = nullptr;
    RcclComm *rccl = nullptr;        // direct RCCL communicator (dlco_comm_init); takes precedence over the callback
    HostComm *hostcomm = nullptr;    // host-staged shared-memory fallback (dlco_comm_init_host)
};

namespace {

constexpr int VCHUNK = 65536;

template <typename Fn>
int guarded(const dlco_ctx *ctx, Fn &&fn)
{
    try {
        fn();
        return DLCO_OK;
    } catch (const Error &e) {
        g_last_error = e.what();
        if (ctx) const_cast<dlco_ctx *>(ctx)->err = e.what();
        return e.code;
    } catch (const std::exception &e) {
        g_last_error = e.what();
        if (ctx) const_cast<dlco_ctx *>(ctx)->err = e.what();
        return DLCO_ERR_INVALID;
    }
}

void sync(dlco_ctx *c) { DLCO_HIP(hipStreamSynchronize(c->stream)); }

// in-place all-gather of one of the exchange buffers through the host's collective (RCCL)
void allgather(dlco_ctx *c, int32_t buffer_id, size_t bytes_per_rank)
{
    if (c->rccl) {                                            // RCCL over xGMI, on the library's own stream
        void *buf = buffer_id == DLCO_BUF_DIST ? static_cast<void *>(c->xdist) : static_cast<void *>(c->comm.gather);
        c->rccl->allgather_inplace(buf, bytes_per_rank, c->stream);
        return;
    }
    if (c->hostcomm) {                                        // fallback: through host shared memory
        void *buf = buffer_id == DLCO_BUF_DIST ? static_cast<void *>(c->xdist) : static_cast<void *>(c->comm.gather);
        c->hostcomm->allgather_inplace(buf, bytes_per_rank, c->stream);
        return;
    }
    DLCO_CHECK(c->ag_fn != nullptr, DLCO_ERR_INVALID, "sharded step: no all-gather callback (dlco_set_allgather)");
    const int rc = c->ag_fn(c->ag_user, buffer_id, bytes_per_rank);
    if (rc != 0) throw Error(DLCO_ERR_COMM, "all-gather callback failed with code " + std::to_string(rc));
}

// Replicated dual average on several ranks (cfg.shard = 0, BASELINE configs[3] as worded): sum all-reduce of the
// F x F partial gradients through the library's communicator
void allreduce_grad(dlco_ctx *c)
{
    const size_t FF = (size_t)c->F * c->F;
    if (c->rccl) { c->rccl->allreduce_sum_f32(c->xgrad, FF, c->stream); return; }   // ncclAllReduce over xGMI
    if (c->hostcomm) { c->hostcomm->allreduce_sum_f32(c->xgrad, FF, c->stream); return; }
    throw Error(DLCO_ERR_INVALID, "dlco_step: world > 1 without cfg.shard needs a communicator (dlco_comm_init / dlco_comm_init_host) "
                                  "or the begin/grad/finish protocol");
}

void h2d(dlco_ctx *c, void *dst, const void *src, size_t bytes)
{
    DLCO_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stream));
    sync(c);
}
void d2h(dlco_ctx *c, void *dst, const void *src, size_t bytes)
{
    DLCO_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream));
    sync(c);
}

// rows of the caller's width Fu <-> device rows of the padded width F.  An upload leaves the pad columns alone: the
// destination was zeroed (or is zeroed here with `zero_pad`) before.
void h2d_rows(dlco_ctx *c, float *dst_dev, const float *src_host, size_t rows, bool zero_pad = false)
{
    if (rows == 0) return;
    if (c->Fu == c->F) { h2d(c, dst_dev, src_host, rows * c->F * sizeof(float)); return; }
    if (zero_pad) DLCO_HIP(hipMemsetAsync(dst_dev, 0, rows * c->F * sizeof(float), c->stream));
    DLCO_HIP(hipMemcpy2DAsync(dst_dev, (size_t)c->F * sizeof(float), src_host, (size_t)c->Fu * sizeof(float),
                              (size_t)c->Fu * sizeof(float), rows, hipMemcpyHostToDevice, c->stream));
    sync(c);
}
void d2h_rows(dlco_ctx *c, float *dst_host, const float *src_dev, size_t rows)
{
    if (rows == 0) return;
    if (c->Fu == c->F) { d2h(c, dst_host, src_dev, rows * c->F * sizeof(float)); return; }
    DLCO_HIP(hipMemcpy2DAsync(dst_host, (size_t)c->Fu * sizeof(float), src_dev, (size_t)c->F * sizeof(float),
                              (size_t)c->Fu * sizeof(float), rows, hipMemcpyDeviceToHost, c->stream));
    sync(c);
}
// an Fu x Fu matrix of the caller <-> the leading block of an F x F device matrix (pad rows and columns zero)
void h2d_square(dlco_ctx *c, float *dst_dev, const float *src_host)
{
    if (c->Fu != c->F) DLCO_HIP(hipMemsetAsync(dst_dev, 0, (size_t)c->F * c->F * sizeof(float), c->stream));
    h2d_rows(c, dst_dev, src_host, (size_t)c->Fu);
}
void d2h_square(dlco_ctx *c, float *dst_host, const float *src_dev) { d2h_rows(c, dst_host, src_dev, (size_t)c->Fu); }

// A resident matrix handed over on the device with the caller's row stride Fu: used in place when no padding is needed,
// else copied once into an own padded allocation.
const float *adopt_device_rows(dlco_ctx *c, const float *src_dev, size_t rows)
{
    if (c->Fu == c->F) return src_dev;
    c->dists_own.alloc(rows * c->F);
    DLCO_HIP(hipMemsetAsync(c->dists_own.p, 0, rows * c->F * sizeof(float), c->stream));
    DLCO_HIP(hipMemcpy2DAsync(c->dists_own.p, (size_t)c->F * sizeof(float), src_dev, (size_t)c->Fu * sizeof(float),
                              (size_t)c->Fu * sizeof(float), rows, hipMemcpyDeviceToDevice, c->stream));
    sync(c);
    return c->dists_own.p;
}

void finish_data(dlco_ctx *c, const uint8_t *labels_host)
{
    c->labels.assign(labels_host, labels_host + c->N);
    c->labels_dev.alloc(c->N);
    h2d(c, c->labels_dev.p, c->labels.data(), c->N);
    c->idx.build(c->labels.data(), c->N);
    const int npv = (int)c->idx.pos.size() - c->idx.n_pos_trn, nnv = (int)c->idx.neg.size() - c->idx.n_neg_trn;
    c->val_pos_ids.alloc(std::max(npv, 1));
    c->val_neg_ids.alloc(std::max(nnv, 1));
    if (npv > 0) h2d(c, c->val_pos_ids.p, c->idx.pos.data() + c->idx.n_pos_trn, (size_t)npv * sizeof(int32_t));
    if (nnv > 0) h2d(c, c->val_neg_ids.p, c->idx.neg.data() + c->idx.n_neg_trn, (size_t)nnv * sizeof(int32_t));
    c->vdist.alloc((size_t)c->N + 16);
    c->hrows.alloc((size_t)std::max(npv, 1));
    if (c->roc) roc_work_destroy(c->roc);
    c->roc = roc_work_create(c->N);
    c->have_data = true;
}

// Where the kernels find training rows `ids_dev[0..n)` (or rows base..base+n when ids_dev is
// NULL).  Row mode: the ids themselves.  Pair mode: two descriptor-row lists whose difference is
// the training row (src/comp-uprjdists.cpp:327); a listed selection is translated on the stream
// into the scratch lists, which stay valid until the next call.
struct RowRef { const int32_t *a, *b; };

RowRef rows_of(dlco_ctx *c, const int32_t *ids_dev, int base, int n)
{
    if (!c->pair_mode) return {ids_dev, nullptr};
    if (!ids_dev) return {c->pair_a.p + base, c->pair_b.p + base};
    if ((size_t)n > c->tr_cap) {
        sync(c);
        c->tr_cap = (size_t)std::max(n, 4096);
        c->tr_a.alloc(c->tr_cap); c->tr_b.alloc(c->tr_cap);
    }
    translate_ids(ids_dev, 0, n, c->pair_a.p, c->pair_b.p, c->tr_a.p, c->tr_b.p, c->stream);
    return {c->tr_a.p, c->tr_b.p};
}

// dist[i] = |W x_{row(i)}|^2 for many rows (validation / statistics): chunked GEMM, fused square-sum
void project_many(dlco_ctx *c, const int32_t *ids_dev, int row0, int n, const float *Wd, int r, float *out_dev)
{
    if (n <= 0) return;
    if (r <= 0) { fill_f32(out_dev, 0.f, (size_t)n, c->stream); return; }
    const RowRef rr = rows_of(c, ids_dev, row0, n);
    // fused project + square + column sum, the rows of D streamed once and no r x n buffer (W up to 384 rows: four
    // passes of 96; taller W - the start-up transient - takes the GEMM path below)
    if (r <= 384 && c->F % 64 == 0) {
        for (auto &pl : c->pplane) pl.alloc(project_rows_plane_bytes(c->F));
        if (project_rows_sqdist(Wd, c->F, r, c->dists, c->F, c->F, rr.a, rr.b, row0, n, out_dev, c->pplane[0].p, c->pplane[1].p,
                                c->pplane[2].p, c->cfg.grad_bf16 != 0, c->stream))
            return;
    }
    c->vproj.alloc((size_t)r * VCHUNK);
    for (int c0 = 0; c0 < n; c0 += VCHUNK) {
        const int nc = std::min(VCHUNK, n - c0);
        GemmArgs g;
        g.M = r; g.N = nc; g.K = c->F;
        g.A.p = Wd; g.A.ld = c->F; g.A.kmajor = false;
        g.B.ld = c->F; g.B.kmajor = false;
        if (rr.a) { g.B.p = c->dists; g.B.row_ids = rr.a + c0; g.B.row_ids2 = rr.b ? rr.b + c0 : nullptr; }
        else g.B.p = c->dists + (size_t)(row0 + c0) * c->F;
        g.C = c->vproj.p; g.ldc = VCHUNK;
        gemm_f32(g, c->stream);
        sqdist_from_proj(c->vproj.p, 1, r, nc, VCHUNK, out_dev + c0, c->stream);
    }
}

// few rows (the training batch): split-K slabs summed in order, then squared
// r_ext > r: W holds r_ext rows, the tracker's guard rows behind its r own; all are projected (one GEMM), the distances
// sum the first r, and the whole projection stays in place for the tracker (c->ru_proj)
void project_few(dlco_ctx *c, const int32_t *ids_dev, int n, const float *Wd, int r, float *out_dev, int r_ext = 0)
{
    c->ru_proj = nullptr; c->ru_rows = 0;
    if (n <= 0) return;
    if (r <= 0) { fill_f32(out_dev, 0.f, (size_t)n, c->stream); return; }
    const int rp = std::max(r, r_ext);
    if (c->cfg.grad_bf16 && rp <= 96 && c->F % 64 == 0) {
        // BASELINE configs[4] variant: the batch projection on the bf16 matrix cores too (operands rounded to bf16 once,
        // fp32 accumulation), K split over enough slices to fill the chip, slices summed in order
        const int blocks = (n + 127) / 128;
        int ks = 1;
        while (ks * 2 * blocks <= 256 && (c->F / 64) % (ks * 2) == 0) ks *= 2;
        const size_t need = (size_t)(ks + 1) * rp * n;
        if (need > c->proj_slab_floats) { sync(c); c->proj_slab.alloc(need); c->proj_slab_floats = need; }
        float *reduced = rp > r ? c->proj_slab.p + (size_t)ks * rp * n : nullptr;   // [rp][n] behind the slabs: what the rank update reads
        for (auto &pl : c->pplane) pl.alloc(project_rows_plane_bytes(c->F));
        const RowRef rr = rows_of(c, ids_dev, 0, n);
        c->prof.begin(PROF_PROJECT);
        const bool ok = project_rows_slab(Wd, c->F, rp, c->dists, c->F, c->F, rr.a, rr.b, n, ks, c->proj_slab.p, n, c->pplane[0].p,
                                          c->pplane[1].p, c->pplane[2].p, true, c->stream);
        if (ok) {
            if (ks > 2) sqdist_from_proj(c->proj_slab.p, ks, r, n, n, out_dev, c->stream, rp, reduced);
            else { sqdist_from_proj(c->proj_slab.p, ks, r, n, n, out_dev, c->stream); reduced = nullptr; }   // (never with F >= 256)
        }
        c->prof.end(PROF_PROJECT);
        if (ok) {
            if (reduced) { c->ru_proj = reduced; c->ru_rows = rp; }
            return;
        }
    }
    const int bm = rp <= 128 ? 64 : 128;     // (a 96-row block: two 64-row tiles or one of 128 measured the same, 25 us;
    const int bn = (n <= 64 || bm == 64) ? 64 : 128;                                                         //  64-column tiles as well: 19 us)
    const long tiles = (long)((rp + bm - 1) / bm) * ((n + bn - 1) / bn);
    long split = std::max(1L, std::min((512 + tiles - 1) / tiles, (long)c->F / 128));
    const size_t need = (size_t)(split + 1) * rp * n;
    if (need > c->proj_slab_floats) { sync(c); c->proj_slab.alloc(need); c->proj_slab_floats = need; }
    float *proj = c->proj_slab.p + (size_t)split * rp * n;      // reduced [rp][n] projection behind the slabs
    GemmArgs g;
    g.M = rp; g.N = n; g.K = c->F;
    g.A.p = Wd; g.A.ld = c->F; g.A.kmajor = false;
    const RowRef rr = rows_of(c, ids_dev, 0, n);
    g.B.p = c->dists; g.B.ld = c->F; g.B.kmajor = false; g.B.row_ids = rr.a; g.B.row_ids2 = rr.b;
    g.C = proj; g.ldc = n;
    g.split_k = (int)split; g.slab = c->proj_slab.p;
    g.small_m_tiles = bm == 64;
    g.small_n_tiles = bn == 64;
    c->prof.begin(PROF_PROJECT);
    gemm_f32(g, c->stream);                                      // slabs summed in slice order: deterministic
    sqdist_from_proj(proj, 1, r, n, n, out_dev, c->stream);
    c->prof.end(PROF_PROJECT);
    if (rp > r) { c->ru_proj = proj; c->ru_rows = rp; }
}

// dst = beta*dst_in + alpha * X^T diag(w) X over the active rows (upper triangle computed, mirrored)
void grad_syrk(dlco_ctx *c, const int32_t *ids, const float *w, const int *k_dev, int kmax, float alpha, float beta,
               float *dst, bool packed = false, const RankCoeffJob *coeff_job = nullptr)
{
    const int kpad = (kmax + 31) & ~31;                 // the lists are zero padded up to here
    const RowRef rr = rows_of(c, ids, 0, kpad);
    // fused, symmetric, branch-free kernel when the shape allows it (F a multiple of 128)
    if (c->F % 128 == 0 && (reinterpret_cast<uintptr_t>(c->dists) & 15) == 0 &&
        (!c->shard || (c->comm.c0 % 128 == 0 && c->comm.cw % 128 == 0))) {
        c->prof.begin(PROF_GRAD_SYRK);
        const size_t pbytes = syrk_planes_bytes(kpad, c->F);
        if (pbytes > c->syrk_planes.n) { sync(c); c->syrk_planes.alloc(pbytes); }
        const bool done = syrk_rda_f32(c->dists, c->F, rr.a, rr.b, w, k_dev, kpad, c->F, alpha, beta, dst, c->F, c->stream,
                                       c->shard ? c->comm.c0 : 0, c->shard ? c->comm.cw : 0, c->cfg.grad_bf16 != 0, packed,
                                       c->syrk_planes.p, coeff_job);
        c->prof.end(PROF_GRAD_SYRK);
        DLCO_CHECK(done, DLCO_ERR_INVALID, "grad_syrk: fused kernel rejected an eligible shape");
        return;
    }
    DLCO_CHECK(!packed, DLCO_ERR_INVALID, "grad_syrk: the packed layout needs the fused kernel");
    GemmArgs g;
    g.M = c->F; g.N = c->F; g.K = kmax;
    g.A.p = c->dists; g.A.ld = c->F; g.A.kmajor = true; g.A.row_ids = rr.a; g.A.row_ids2 = rr.b; g.A.row_scale = w;
    g.B.p = c->dists; g.B.ld = c->F; g.B.kmajor = true; g.B.row_ids = rr.a; g.B.row_ids2 = rr.b;
    g.C = dst; g.ldc = c->F;
    g.alpha = alpha; g.beta = beta;
    g.k_dev = k_dev;
    g.upper_only = !c->shard;
    if (c->shard) { g.N = c->comm.cw; g.B.p = c->dists + c->comm.c0; g.C = dst + c->comm.c0; }   // the rank's column slab
    c->prof.begin(PROF_GRAD_SYRK);
    gemm_f32(g, c->stream);
    c->prof.end(PROF_GRAD_SYRK);
}

// Global distance vectors pd[B], nd[B] for the violation counts.  With one rank the exchange buffer
// already is [pd | nd]; with several it is [world][pd slice | nd slice] and is regrouped.
void gather_dists(dlco_ctx *c, const float **pd, const float **nd)
{
    const int Bl = c->Bl, world = c->cfg.world;
    if (world == 1) { *pd = c->xdist; *nd = c->xdist + Bl; return; }
    for (int g = 0; g < world; g++) {
        DLCO_HIP(hipMemcpyAsync(c->pd.p + (size_t)g * Bl, c->xdist + (size_t)g * 2 * Bl, Bl * sizeof(float),
                                hipMemcpyDeviceToDevice, c->stream));
        DLCO_HIP(hipMemcpyAsync(c->nd.p + (size_t)g * Bl, c->xdist + (size_t)g * 2 * Bl + Bl, Bl * sizeof(float),
                                hipMemcpyDeviceToDevice, c->stream));
    }
    *pd = c->pd.p; *nd = c->nd.p;
}

void rda_coeffs(const dlco_ctx *c, float *alpha, float *beta)
{
    dlco::rda_coeffs((uint32_t)c->B, c->t, alpha, beta);   // pair_index.hpp (64-bit denominator)
}

void step_begin(dlco_ctx *c)
{
    DLCO_CHECK(c->have_data, DLCO_ERR_INVALID, "dlco_step: no data set");
    DLCO_CHECK(c->phase == 0, DLCO_ERR_INVALID, "dlco_step_begin: previous step not finished");
    DLCO_CHECK(c->idx.n_pos_trn > 0 && c->idx.n_neg_trn > 0, DLCO_ERR_INVALID, "dlco_step: empty training split");
    const int B = c->B, Bl = c->Bl;
    // R3 (src/pj-learn.cpp:310-314): interleaved draws, the same list on every rank
    for (int k = 0; k < B; k++) {
        const int ip = c->rng.uniform(0, c->idx.n_pos_trn);
        const int in = c->rng.uniform(0, c->idx.n_neg_trn);
        c->h_pos_rows[k] = c->idx.pos[ip];
        c->h_neg_rows[k] = c->idx.neg[in];
    }
    // the upload reads a pinned slot that is not rewritten before the step after next (every step
    // synchronises with the stream at least once in the tracker), so no sync is needed here
    const size_t n_ids = (size_t)2 * B + 2 * Bl;
    int32_t *slot = c->pin_ids + (size_t)(c->upload_ctr++ & 1u) * n_ids;
    std::memcpy(slot, c->h_pos_rows.data(), B * sizeof(int32_t));
    std::memcpy(slot + B, c->h_neg_rows.data(), B * sizeof(int32_t));
    std::memcpy(slot + 2 * B, c->h_pos_rows.data() + c->lo, Bl * sizeof(int32_t));
    std::memcpy(slot + 2 * B + Bl, c->h_neg_rows.data() + c->lo, Bl * sizeof(int32_t));
    // (letting the kernels read the pinned slot directly instead of this copy was measured slower: 0.852 against 0.838 ms per step)
    DLCO_HIP(hipMemcpyAsync(c->ids_all.p, slot, n_ids * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
    // P1+P2 on this rank's slots -> its slice of the exchange buffer
    // (the tracker's guard rows sit behind the r rows of W: projected along, see step_grad)
    const int r_ext = (c->rank_update && c->r > 0 && c->eig->ext_rows() > c->r) ? c->eig->ext_rows() : 0;
    project_few(c, c->local_ids.p, 2 * Bl, c->W.p, c->r, c->xdist + (size_t)c->cfg.rank * 2 * Bl, r_ext);
    c->phase = 1;
}

void step_grad(dlco_ctx *c)
{
    DLCO_CHECK(c->phase == 1, DLCO_ERR_INVALID, "dlco_step_grad: call dlco_step_begin first");
    DLCO_CHECK(!c->shard, DLCO_ERR_INVALID, "dlco_step_grad: a sharded context steps with dlco_step");
    const int B = c->B, Bl = c->Bl, world = c->cfg.world;
    (void)world;
    gather_dists(c, &c->pd_cur, &c->nd_cur);
    viol_counts_active_rows(c->pd_cur, c->nd_cur, B, c->rho.p, c->kappa.p, c->pos_rows.p, c->neg_rows.p, c->lo, c->lo + Bl,
                            c->act_ids.p, c->act_w.p, c->k_active.p, c->stream, c->act_slot.p);
    float alpha, beta;
    rda_coeffs(c, &alpha, &beta);
    // (the coefficient fragments of the tracker's rank-update first term ride in the gradient's row-split launch)
    RankCoeffJob job;
    // (the bf16-once variant's row kernel has no such column: its fragments take a launch of their own)
    const bool have_job = c->rank_update && c->planes_mode == 3 && c->ru_proj && world == 1 && c->eig->rank_coeff_job(&job, (2 * Bl + 31) & ~31);
    if (have_job) { job.proj = c->ru_proj; job.ldp = 2 * Bl; job.slot = c->act_slot.p; }
    if (world == 1) grad_syrk(c, c->act_ids.p, c->act_w.p, c->k_active.p, 2 * Bl, alpha, beta, c->dfavg.p, c->packed, have_job ? &job : nullptr);
    else grad_syrk(c, c->act_ids.p, c->act_w.p, c->k_active.p, 2 * Bl, 1.0f, 0.0f, c->xgrad);
    if (c->rank_update && c->ru_proj && c->syrk_planes.p) {
        // dfAvg <- beta dfAvg + alpha X_a^T diag(w) X_a has just been applied: the tracker may form its first filter
        // term from Y = Q H_t, the projections of the batch on its rows and the gradient's planes of X_a
        RankUpdate ru;
        ru.proj = c->ru_proj; ru.ldp = 2 * Bl; ru.rows = c->ru_rows;
        ru.slot = c->act_slot.p; ru.w = c->act_w.p; ru.k_dev = c->k_active.p; ru.kmax = (2 * Bl + 31) & ~31;
        ru.planes = c->syrk_planes.p; ru.alpha = alpha; ru.beta = beta;
        ru.coeff_ready = have_job && job.m == c->ru_rows;
        ru.planes_mode = c->planes_mode;
        c->eig->offer_rank_update(ru);
    }
    c->phase = 2;
}

void step_finish(dlco_ctx *c)
{
    DLCO_CHECK(c->phase == 2, DLCO_ERR_INVALID, "dlco_step_finish: call dlco_step_grad first");
    if (c->cfg.world > 1 && !c->shard) {
        float alpha, beta;
        rda_coeffs(c, &alpha, &beta);
        axpby_inplace(c->dfavg.p, c->xgrad, beta, alpha, (size_t)c->F * c->F, c->stream);
    }
    // E1/E2.  Cold tracker: the range of dfAvg after the first step is spanned by the batch rows,
    // so they seed the block (global batch: every rank holds the full row lists and counts).
    // The positive eigen-directions of H = -dfAvg come from the negative pairs' outer products,
    // so the negatives are listed first; at most 512 rows are used (with the reference's B = 200
    // that is every row and the first Rayleigh-Ritz is exact; larger global batches start from
    // a subspace and iterate).
    if (c->eig->block_rows() == 0) {
        build_active_rows(c->neg_rows.p, c->pos_rows.p, c->kappa.p, c->rho.p, c->B, 0, c->B, c->seed_ids.p, c->seed_w.p,
                          c->k_active.p + 1, c->stream);
        int k = 0;
        d2h(c, &k, c->k_active.p + 1, sizeof(int));
        if (k > 0) {
            const int ks = std::min(k, 512);
            const RowRef rr = rows_of(c, c->seed_ids.p, 0, ks);
            c->eig->seed_rows(c->dists, c->F, rr.a, ks, rr.b);
        }
    }
    const float cscale = (float)(std::sqrt((double)c->t + 1.0) / (double)c->cfg.gamma);
    bool conv = true;
    // rows that entered this rank's SYRK this step: the count rides in the tracker's own read-back of its Ritz block
    c->r = c->eig->update(c->dfavg.p, c->cfg.mu, cscale, c->W.p, &c->traceA, &conv);
    if (!conv) { c->nonconv_steps++; c->nonconv_window++; }
    c->active_rows_sum += c->eig->readback_extra();      // as of the last pass of this update: this step's count
    c->steps_run++;
    c->t++;
    c->phase = 0;
    if (!conv && c->cfg.strict_conv)
        throw Error(DLCO_ERR_NOCONV, "dlco_step: the eigen tracker missed its tolerance at t = " + std::to_string(c->t - 1) +
                                     " (the step was applied; W is approximate)");
}

// One step with the dual average sharded by columns (cfg.shard): the rank projects its batch
// slots, all-gathers the 2B distances, computes its column slab of the gradient over the WHOLE
// global batch fused with the dual-average update (no F x F exchange), and runs the replicated
// tracker whose products with dfAvg all-gather their column slabs.
void step_sharded(dlco_ctx *c)
{
    step_begin(c);
    allgather(c, DLCO_BUF_DIST, (size_t)2 * c->Bl * sizeof(float));
    const int B = c->B;
    gather_dists(c, &c->pd_cur, &c->nd_cur);
    viol_counts_active_rows(c->pd_cur, c->nd_cur, B, c->rho.p, c->kappa.p, c->pos_rows.p, c->neg_rows.p, 0, B, c->act_ids.p,
                            c->act_w.p, c->k_active.p, c->stream);
    float alpha, beta;
    rda_coeffs(c, &alpha, &beta);
    grad_syrk(c, c->act_ids.p, c->act_w.p, c->k_active.p, 2 * B, alpha, beta, c->dfavg.p);
    c->phase = 2;
    step_finish(c);
}

void get_W_host(dlco_ctx *c, float *W_host, int32_t *r)
{
    if (c->r > 0) {
        if (W_host) d2h_rows(c, W_host, c->W.p, (size_t)c->r);
        if (r) *r = c->r;
    } else {
        if (W_host) std::memset(W_host, 0, (size_t)c->Fu * c->Fu * sizeof(float));   // src/pj-learn.cpp:489-490
        if (r) *r = c->Fu;
    }
}

// A = W^T W into the scratch `grad` buffer (device)
void build_A(dlco_ctx *c, const float *Wd, int r)
{
    if (r <= 0) { fill_f32(c->grad.p, 0.f, (size_t)c->F * c->F, c->stream); return; }
    GemmArgs g;
    g.M = c->F; g.N = c->F; g.K = r;
    g.A.p = Wd; g.A.ld = c->F; g.A.kmajor = true;
    g.B.p = Wd; g.B.ld = c->F; g.B.kmajor = true;
    g.C = c->grad.p; g.ldc = c->F;
    g.upper_only = true;
    gemm_f32(g, c->stream);
}

void validate(dlco_ctx *c, float *loss_val, float *regul, int32_t *rank)
{
    DLCO_CHECK(c->have_data, DLCO_ERR_INVALID, "dlco_validate: no data set");
    const int npv = (int)c->idx.pos.size() - c->idx.n_pos_trn, nnv = (int)c->idx.neg.size() - c->idx.n_neg_trn;
    DLCO_CHECK(npv > 0 && nnv > 0, DLCO_ERR_INVALID, "dlco_validate: empty validation split");
    float *pdv = c->vdist.p, *ndv = c->vdist.p + npv;
    project_many(c, c->val_pos_ids.p, 0, npv, c->W.p, c->r, pdv);
    project_many(c, c->val_neg_ids.p, 0, nnv, c->W.p, c->r, ndv);
    hinge_rows(pdv, npv, ndv, nnv, c->hrows.p, c->stream);
    sum_f32_to_f64(c->hrows.p, npv, c->dscal.p, c->stream);
    double total = 0.0;
    d2h(c, &total, c->dscal.p, sizeof(double));
    const float Loss = (float)total;                               // src/pj-learn.cpp:520
    *loss_val = Loss / (float)npv / (float)nnv;                    // :524
    *regul = (float)((double)c->cfg.mu * c->traceA);               // :527, trace(A) = sum of kept eigenvalues
    if (rank) *rank = c->r > 0 ? c->r : c->Fu;
}

void stats(dlco_ctx *c, const float *Wd, int r, int32_t *dim, float *fpr95, double *auc)
{
    DLCO_CHECK(c->have_data, DLCO_ERR_INVALID, "dlco_stats: no data set");
    project_many(c, nullptr, 0, c->N, Wd, r, c->vdist.p);
    roc_stats(c->roc, c->vdist.p, c->labels_dev.p, c->N, fpr95, auc, c->stream);
    if (dim) *dim = r;
}

}  // namespace

// ---------------------------------------------------------------------------------------------
extern "C" {

void dlco_cfg_default(dlco_cfg *cfg)
{
    std::memset(cfg, 0, sizeof(*cfg));
    cfg->B = 200; cfg->mu = 0.001f; cfg->gamma = 0.5f; cfg->seed = 2215;
    cfg->device = 0; cfg->rank = 0; cfg->world = 1;
    cfg->eig_tol = 2e-4f; cfg->eig_guard = 32; cfg->eig_max_iter = 40;
}

const char *dlco_version(void) { return "dlco-mi355x 0.1 (gfx950)"; }

const char *dlco_last_error(const dlco_ctx *ctx) { return ctx ? ctx->err.c_str() : g_last_error.c_str(); }

int dlco_ctx_create(dlco_ctx **out, const dlco_cfg *cfg)
{
    if (!out || !cfg) { g_last_error = "dlco_ctx_create: null argument"; return DLCO_ERR_INVALID; }
    *out = nullptr;
    dlco_ctx *c = nullptr;
    int rc = guarded(nullptr, [&] {
        DLCO_CHECK(cfg->F >= 1 && cfg->F <= (1 << 20), DLCO_ERR_INVALID, "F must be positive");
        DLCO_CHECK(cfg->N >= 2 && cfg->B >= 1, DLCO_ERR_INVALID, "N >= 2 and B >= 1 required");
        DLCO_CHECK(cfg->world >= 1 && cfg->rank >= 0 && cfg->rank < cfg->world, DLCO_ERR_INVALID, "bad rank/world");
        DLCO_CHECK(cfg->B % cfg->world == 0, DLCO_ERR_INVALID, "B must be divisible by world");
        DLCO_CHECK(cfg->gamma > 0.f, DLCO_ERR_INVALID, "gamma must be positive");
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || cfg->device >= ndev)
            throw Error(DLCO_ERR_NODEVICE, "no usable HIP device (this library has no CPU fallback)");
        DLCO_HIP(hipSetDevice(cfg->device));
        c = new dlco_ctx();
        c->cfg = *cfg;
        if (c->cfg.eig_tol <= 0.f) c->cfg.eig_tol = 2e-4f;
        if (const char *e = std::getenv("DLCO_EIG_TOL")) { const float v = (float)std::atof(e); if (v > 0.f) c->cfg.eig_tol = v; }   // developer override
        if (c->cfg.eig_guard <= 0) c->cfg.eig_guard = 32;
        if (c->cfg.eig_max_iter <= 0) c->cfg.eig_max_iter = 40;
        DLCO_HIP(hipGetDeviceProperties(&c->prop, cfg->device));
        if (std::strncmp(c->prop.gcnArchName, "gfx950", 6) != 0)
            throw Error(DLCO_ERR_NODEVICE, std::string("device is ") + c->prop.gcnArchName + ", this build targets gfx950 only");
        c->Fu = cfg->F; c->N = cfg->N; c->B = cfg->B;
        c->Bl = cfg->B / cfg->world; c->lo = cfg->rank * c->Bl;
        // column-sharded dual average (world > 1, or forced for single-rank testing of the path)
        c->shard = cfg->shard != 0 && (cfg->world > 1 || std::getenv("DLCO_FORCE_SHARD") != nullptr);
        // device width: whole 128-column tiles (whole tiles per rank when the columns are sharded)
        {
            const int gran = 128 * (c->shard ? cfg->world : 1);
            c->F = (c->Fu + gran - 1) / gran * gran;
        }
        c->rng = CvRng(cfg->seed);
        DLCO_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
        c->prof.s = c->stream;
        const size_t FF = (size_t)c->F * c->F;
        // One rank, whole dual average, F = 8192: dfAvg is kept as its packed upper 128 x 128 tiles (136 MB instead of
        // 268 MB) - the SYRK stores no mirror and the tracker's products fetch every tile once (kernels_bf16x2.hip).
        c->packed = cfg->world == 1 && cfg->shard == 0 && EigTracker::packed_supported(c->F);
        c->dfavg.alloc(c->packed ? syrk_packed_floats(c->F) : FF); c->dfavg.zero(c->stream);
        c->grad.alloc(FF);
        // block capacity (positive rank + guards).  At t = 0 (W = 0, every pair violates) the positive
        // eigenspace of -dfAvg has up to B dimensions (the negative rows), so the capacity follows the
        // global batch; a block that fills up nevertheless is reported as non-converged.
        const int max_rows = std::max(1024, 2 * cfg->B + 2 * c->cfg.eig_guard);
        // the m x m solver of the Rayleigh-Ritz step (jacobi_eigh) takes at most 4096 rows: a global batch whose
        // block could outgrow that is refused here, not in the middle of a run
        DLCO_CHECK(std::min(c->Fu, max_rows) <= 4096, DLCO_ERR_INVALID,
                   "global batch too large for the eigen tracker: 2*B + 2*eig_guard must not exceed 4096 when F > 4096");
        c->eig = new EigTracker(c->F, max_rows, c->cfg.eig_guard, c->cfg.eig_tol, c->cfg.eig_max_iter, c->stream, c->Fu);
        c->eig->set_profiler(&c->prof);
        c->eig->set_packed(c->packed);
        c->w_cap = std::min(c->Fu, std::max(max_rows, 2 * c->cfg.eig_guard + 32));
        c->W.alloc((size_t)c->w_cap * c->F);
        const int B = c->B;
        c->h_pos_rows.assign(B, 0); c->h_neg_rows.assign(B, 0);
        DLCO_HIP(hipHostMalloc((void **)&c->pin_ids, (size_t)2 * (2 * B + 2 * c->Bl) * sizeof(int32_t)));
        c->ids_all.alloc((size_t)2 * B + 2 * c->Bl);
        c->pos_rows.p = c->ids_all.p; c->neg_rows.p = c->ids_all.p + B; c->local_ids.p = c->ids_all.p + 2 * B;
        c->rho.alloc(B); c->kappa.alloc(B);
        const int kcap = (2 * B + 31) & ~31;                      // row lists are zero padded to whole K tiles
        c->act_ids.alloc(kcap); c->act_w.alloc(kcap); c->seed_ids.alloc(kcap); c->seed_w.alloc(kcap); c->act_slot.alloc(kcap);
        // one rank, packed dual average, the gradient's default arithmetic (its split planes are what the shortcut reads)
        c->planes_mode = syrk_planes_mode(cfg->grad_bf16 != 0);
        c->rank_update = c->packed && cfg->world == 1 && c->planes_mode != 0 && std::getenv("DLCO_NO_RANK_UPDATE") == nullptr;
        c->eig->set_emit_guards(c->rank_update);
        c->dist_x.alloc(2 * B); c->pd.alloc(B); c->nd.alloc(B);
        c->xdist = c->dist_x.p; c->xgrad = c->grad.p;
        c->k_active.alloc(4);
        c->k_active.zero(c->stream);                              // read back with every Ritz block, also before the first step
        c->eig->set_readback_extra(c->k_active.p);
        // a growing block takes half of its new rows from the END of the step's active list (the negatives: the positive
        // eigen-directions of H = -dfAvg come from their outer products).  Every rank must add the same rows, so only
        // where the list is the global one: a single rank, or the sharded layout.
        if (c->cfg.world == 1 || c->shard) {
            c->eig->set_growth_rows([c](float *dst, int want) -> int {
                // only inside a training step (phase 2: the active list on the device is this step's); an operator call
                // (dlco_psd_project) or a context without data grows from random rows alone
                if (c->phase != 2 || !c->have_data) return 0;
                const int k = c->eig->readback_extra();          // this step's active row count (rode in with the Ritz block)
                if (k <= 0 || k > 2 * c->B) return 0;
                const int cnt = std::min(want, k);
                if (cnt <= 0) return 0;
                const RowRef rr = rows_of(c, c->act_ids.p + (k - cnt), 0, cnt);
                scale_rows(dst, c->F, c->dists, c->F, nullptr, rr.a, cnt, c->F, c->stream, rr.b);
                return cnt;
            });
        }
        c->dscal.alloc(4);
        if (c->shard) {
            c->comm.world = cfg->world; c->comm.rank = cfg->rank;
            c->comm.cw = c->F / cfg->world; c->comm.c0 = cfg->rank * c->comm.cw;
            c->comm.gather_floats = (size_t)c->w_cap * c->F;
            c->gather_own.alloc(c->comm.gather_floats);
            c->comm.gather = c->gather_own.p;
            c->comm.allgather = [c](size_t bytes) { allgather(c, DLCO_BUF_GATHER, bytes); };
            c->eig->set_shard(&c->comm);
        }
        sync(c);
    });
    if (rc != DLCO_OK) { delete c; return rc; }
    *out = c;
    return DLCO_OK;
}

void dlco_ctx_destroy(dlco_ctx *c)
{
    if (!c) return;
    (void)hipSetDevice(c->cfg.device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    delete c->eig;
    delete c->rccl;
    delete c->hostcomm;
    if (c->roc) roc_work_destroy(c->roc);
    if (c->pin_ids) (void)hipHostFree(c->pin_ids);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

int dlco_device_name(const dlco_ctx *c, char *buf, size_t cap, int *cc_major, int *cc_minor)
{
    if (!c || !buf || cap == 0) return DLCO_ERR_INVALID;
    std::snprintf(buf, cap, "%s (%s)", c->prop.name, c->prop.gcnArchName);
    if (cc_major) *cc_major = c->prop.major;
    if (cc_minor) *cc_minor = c->prop.minor;
    return DLCO_OK;
}

int dlco_set_data(dlco_ctx *c, const float *dists_host, const uint8_t *labels_host)
{
    if (!c || !dists_host || !labels_host) return DLCO_ERR_INVALID;
    return guarded(c, [&] {
        DLCO_HIP(hipSetDevice(c->cfg.device));
        c->dists_own.alloc((size_t)c->N * c->F);
        h2d_rows(c, c->dists_own.p, dists_host, (size_t)c->N, true);
        c->dists = c->dists_own.p;
        c->pair_mode = false;
        finish_data(c, labels_host);
    });
}

// Pair mode: the caller hands over the P per-patch descriptors and the [N,4] pair table
// (patchID1, 3DpointID1, patchID2, 3DpointID2: src/comp-uprjdists.cpp:268-269,308-314) instead of
// the N x F matrix of differences that comp-uprjdists writes to the "Distance" dataset.
static int set_pairs(dlco_ctx *c, const float *desc, bool desc_on_device, int32_t P, const int32_t *pairs_host);

int dlco_set_pairs(dlco_ctx *c, const float *desc_host, int32_t P, const int32_t *pairs_host)
{
    return set_pairs(c, desc_host, false, P, pairs_host);
}

int dlco_set_pairs_device(dlco_ctx *c, const float *desc_dev, int32_t P, const int32_t *pairs_host)
{
    return set_pairs(c, desc_dev, true, P, pairs_host);
}

static int set_pairs(dlco_ctx *c, const float *desc_host, bool desc_on_device, int32_t P, const int32_t *pairs_host)
{
    if (!c || !desc_host || !pairs_host || P < 1) return DLCO_ERR_INVALID;
    return guarded(c, [&] {
        DLCO_HIP(hipSetDevice(c->cfg.device));
        std::vector<int32_t> a(c->N), b(c->N);
        std::vector<uint8_t> lab(c->N);
        for (int i = 0; i < c->N; i++) {
            const int32_t *q = pairs_host + (size_t)i * 4;
            DLCO_CHECK(q[0] >= 0 && q[0] < P && q[2] >= 0 && q[2] < P, DLCO_ERR_INVALID, "dlco_set_pairs: patch id out of range");
            a[i] = q[0]; b[i] = q[2];
            lab[i] = (q[1] == q[3]) ? 1 : 0;                       // src/comp-uprjdists.cpp:268-272
        }
        if (desc_on_device) {
            c->dists = adopt_device_rows(c, desc_host, (size_t)P);
        } else {
            c->dists_own.alloc((size_t)P * c->F);
            h2d_rows(c, c->dists_own.p, desc_host, (size_t)P, true);
            c->dists = c->dists_own.p;
        }
        c->pair_a.alloc(c->N); c->pair_b.alloc(c->N);
        h2d(c, c->pair_a.p, a.data(), (size_t)c->N * sizeof(int32_t));
        h2d(c, c->pair_b.p, b.data(), (size_t)c->N * sizeof(int32_t));
        c->pair_mode = true; c->P = P;
        finish_data(c, lab.data());
    });
}

int dlco_set_data_device(dlco_ctx *c, const float *dists_dev, const uint8_t *labels_host)
{
    if (!c || !dists_dev || !labels_host) return DLCO_ERR_INVALID;
    return guarded(c, [&] {
        DLCO_HIP(hipSetDevice(c->cfg.device));
        c->dists = adopt_device_rows(c, dists_dev, (size_t)c->N);
        c->pair_mode = false;
        finish_data(c, labels_host);
    });
}

int dlco_device_width(const dlco_ctx *c) { return c ? c->F : DLCO_ERR_INVALID; }

int dlco_set_data_shared(dlco_ctx *c, dlco_ctx *src)
{
    if (!c || !src || c == src) return DLCO_ERR_INVALID;
    return guarded(c, [&] {
        DLCO_HIP(hipSetDevice(c->cfg.device));
        DLCO_CHECK(src->have_data, DLCO_ERR_INVALID, "dlco_set_data_shared: the source context holds no data");
        DLCO_CHECK(src->cfg.device == c->cfg.device && src->Fu == c->Fu && src->F == c->F && src->N == c->N, DLCO_ERR_INVALID,
                   "dlco_set_data_shared: device, F (and its padded width) and N must match");
        c->dists = src->dists;
        c->pair_mode = src->pair_mode;
        c->P = src->P;
        if (src->pair_mode) {                                     // the pair table is small: an own copy
            c->pair_a.alloc(c->N); c->pair_b.alloc(c->N);
            DLCO_HIP(hipMemcpyAsync(c->pair_a.p, src->pair_a.p, (size_t)c->N * sizeof(int32_t), hipMemcpyDeviceToDevice, c->stream));
            DLCO_HIP(hipMemcpyAsync(c->pair_b.p, src->pair_b.p, (size_t)c->N * sizeof(int32_t), hipMemcpyDeviceToDevice, c->stream));
            sync(c);
        }
        finish_data(c, src->labels.data());
    });
}

int dlco_synth_data(dlco_ctx *c, const float *U_host, int32_t k, uint64_t seed, float sigma_pos, float sigma_neg,
                    float noise, float scale_jitter)
{
    if (!c || !U_host || k < 1 || k > 4096) return DLCO_ERR_INVALID;
    return guarded(c, [&] {
        DLCO_HIP(hipSetDevice(c->cfg.device));
        c->dists_own.alloc((size_t)c->N * c->F);
        if (c->Fu != c->F) c->dists_own.zero(c->stream);
        DevBuf<float> U;
        U.alloc((size_t)k * c->Fu);
        h2d(c, U.p, U_host, (size_t)k * c->Fu * sizeof(float));
        synth_rows(c->dists_own.p, c->N, c->Fu, c->F, U.p, k, seed, sigma_pos, sigma_neg, noise, scale_jitter, c->stream);
        sync(c);
        c->dists = c->dists_own.p;
        c->pair_mode = false;
        std::vector<uint8_t> lab(c->N);
        for (int i = 0; i < c->N; i++) lab[i] = (i % 2 == 0) ? 1 : 0;
        finish_data(c, lab.data());
    });
}

int dlco_get_rows(dlco_ctx *c, int32_t row0, int32_t n, float *out_host)
{
    if (!c || !out_host) return DLCO_ERR_INVALID;
    return guarded(c, [&] {
        DLCO_CHECK(c->have_data && row0 >= 0 && n >= 0 && row0 + n <= c->N, DLCO_ERR_INVALID, "dlco_get_rows: range");
        if (c->pair_mode) {                                // materialise the differences of the asked rows
            DevBuf<float> tmp;
            tmp.alloc((size_t)std::max(n, 1) * c->F);
            scale_rows(tmp.p, c->F, c->dists, c->F, nullptr, c->pair_a.p + row0, n, c->F, c->stream, c->pair_b.p + row0);
            d2h_rows(c, out_host, tmp.p, (size_t)n);
            return;
        }
        d2h_rows(c, out_host, c->dists + (size_t)row0 * c->F, (size_t)n);
    });
}

int dlco_get_index(const dlco_ctx *c, int32_t *pos, int32_t *n_pos, int32_t *n_pos_trn, int32_t *neg, int32_t *n_neg,
                   int32_t *n_neg_trn)
{
    if (!c || !c->have_data) return DLCO_ERR_INVALID;
    if (pos && !c->idx.pos.empty()) std::memcpy(pos, c->idx.pos.data(), c->idx.pos.size() * sizeof(int32_t));
    if (neg && !c->idx.neg.empty()) std::memcpy(neg, c->idx.neg.data(), c->idx.neg.size() * sizeof(int32_t));
    if (n_pos) *n_pos = (int32_t)c->idx.pos.size();
    if (n_neg) *n_neg = (int32_t)c->idx.neg.size();
    if (n_pos_trn) *n_pos_trn = c->idx.n_pos_trn;
    if (n_neg_trn) *n_neg_trn = c->idx.n_neg_trn;
    return DLCO_OK;
}

int dlco_step_begin(dlco_ctx *c) { return c ? guarded(c, [&] { DLCO_HIP(hipSetDevice(c->cfg.device)); step_begin(c); }) : DLCO_ERR_INVALID; }
int dlco_step_grad(dlco_ctx *c) { return c ? guarded(c, [&] { DLCO_HIP(hipSetDevice(c->cfg.device)); step_grad(c); }) : DLCO_ERR_INVALID; }
int dlco_step_finish(dlco_ctx *c) { return c ? guarded(c, [&] { DLCO_HIP(hipSetDevice(c->cfg.device)); step_finish(c); }) : DLCO_ERR_INVALID; }

int dlco_step(dlco_ctx *c)
{
    if (!c) return DLCO_ERR_INVALID;
    return guarded(c, [&] {
        DLCO_HIP(hipSetDevice(c->cfg.device));
        if (c->shard) { step_sharded(c); return; }
        if (c->cfg.world == 1) { step_begin(c); step_grad(c); step_finish(c); return; }
        // replicated dual average, exchanges issued by the library: all-gather of the 2B distances, then the sum
        // all-reduce of the F x F partial gradients that BASELINE configs[3] names, then the replicated update
        DLCO_CHECK(c->rccl || c->hostcomm, DLCO_ERR_INVALID,
                   "dlco_step: world > 1 needs cfg.shard, or a communicator (dlco_comm_init / dlco_comm_init_host), or the "
                   "begin/grad/finish protocol");
        step_begin(c);
        allgather(c, DLCO_BUF_DIST, (size_t)2 * c->Bl * sizeof(float));
        step_grad(c);
        allreduce_grad(c);
        step_finish(c);
    });
}

int dlco_steps(dlco_ctx *c, int32_t n)
{
    for (int i = 0; i < n; i++) {
        const int rc = dlco_step(c);
        if (rc != DLCO_OK) return rc;
    }
    return DLCO_OK;
}

int dlco_dev_buffer(dlco_ctx *c, int32_t which, void **dev_ptr, size_t *bytes)
{
    if (!c || !dev_ptr || !bytes) return DLCO_ERR_INVALID;
    const size_t FF = (size_t)c->F * c->F * sizeof(float);
    switch (which) {
    case DLCO_BUF_DIST: *dev_ptr = c->xdist; *bytes = (size_t)2 * c->B * sizeof(float); return DLCO_OK;
    case DLCO_BUF_GRAD: *dev_ptr = c->xgrad; *bytes = FF; return DLCO_OK;
    case DLCO_BUF_DFAVG:                                          // (packed contexts: the packed upper tiles, see dlco.h)
        *dev_ptr = c->dfavg.p; *bytes = c->packed ? syrk_packed_floats(c->F) * sizeof(float) : FF; return DLCO_OK;
    case DLCO_BUF_W: *dev_ptr = c->W.p; *bytes = (size_t)c->r * c->F * sizeof(float); return DLCO_OK;
    case DLCO_BUF_DATA:
        if (!c->have_data) return DLCO_ERR_INVALID;
        *dev_ptr = const_cast<float *>(c->dists);
        *bytes = (size_t)(c->pair_mode ? c->P : c->N) * c->F * sizeof(float);
        return DLCO_OK;
    case DLCO_BUF_GATHER:
        if (!c->shard) return DLCO_ERR_INVALID;
        *dev_ptr = c->comm.gather; *bytes = c->comm.gather_floats * sizeof(float); return DLCO_OK;
    default: return DLCO_ERR_INVALID;
    }
}

int dlco_bind_buffer(dlco_ctx *c, int32_t which, void *dev_ptr, size_t bytes)
{
    if (!c || !dev_ptr) return DLCO_ERR_INVALID;
    return guarded(c, [&] {
        DLCO_CHECK(c->phase == 0, DLCO_ERR_INVALID, "dlco_bind_buffer: step in flight");
        if (which == DLCO_BUF_DIST) {
            DLCO_CHECK(bytes >= (size_t)2 * c->B * sizeof(float), DLCO_ERR_INVALID, "dlco_bind_buffer: DIST buffer too small");
            c->xdist = static_cast<float *>(dev_ptr);
        } else if (which == DLCO_BUF_GRAD) {
            DLCO_CHECK(bytes >= (size_t)c->F * c->F * sizeof(float), DLCO_ERR_INVALID, "dlco_bind_buffer: GRAD buffer too small");
            c->xgrad = static_cast<float *>(dev_ptr);
        } else if (which == DLCO_BUF_GATHER) {
            DLCO_CHECK(c->shard, DLCO_ERR_INVALID, "dlco_bind_buffer: GATHER exists only in a sharded context");
            DLCO_CHECK(bytes >= c->comm.gather_floats * sizeof(float), DLCO_ERR_INVALID, "dlco_bind_buffer: GATHER buffer too small");
            c->comm.gather = static_cast<float *>(dev_ptr);
        } else {
            throw Error(DLCO_ERR_INVALID, "dlco_bind_buffer: only DLCO_BUF_DIST, DLCO_BUF_GRAD and DLCO_BUF_GATHER can be bound");
        }
    });
}

int dlco_stream(dlco_ctx *c, void **stream)
{
    if (!c || !stream) return DLCO_ERR_INVALID;
    *stream = (void *)c->stream;
    return DLCO_OK;
}

int dlco_set_allgather(dlco_ctx *c, dlco_allgather_fn fn, void *user)
{
    if (!c) return DLCO_ERR_INVALID;
    c->ag_fn = fn;
    c->ag_user = user;
    return DLCO_OK;
}

int dlco_comm_unique_id(void *out_id, size_t cap, const char *rccl_path)
{
    if (!out_id || cap < 128) return DLCO_ERR_INVALID;
    return guarded(nullptr, [&] { RcclComm::unique_id(out_id, rccl_path); });
}

int dlco_comm_init(dlco_ctx *c, const void *id, size_t id_bytes, const char *rccl_path)
{
    if (!c || !id || id_bytes < 128) return DLCO_ERR_INVALID;
    return guarded(c, [&] {
        DLCO_CHECK(c->cfg.world > 1 || c->shard, DLCO_ERR_INVALID, "dlco_comm_init: a single-rank context has nothing to exchange");
        DLCO_CHECK(c->phase == 0, DLCO_ERR_INVALID, "dlco_comm_init: step in flight");
        DLCO_HIP(hipSetDevice(c->cfg.device));
        delete c->rccl;
        c->rccl = nullptr;
        c->rccl = new RcclComm(id, c->cfg.rank, c->cfg.world, rccl_path);
    });
}

int dlco_comm_init_host(dlco_ctx *c, const char *shm_name)
{
    if (!c || !shm_name || !*shm_name) return DLCO_ERR_INVALID;
    return guarded(c, [&] {
        DLCO_CHECK(c->cfg.world > 1 || c->shard, DLCO_ERR_INVALID, "dlco_comm_init_host: a single-rank context has nothing to exchange");
        DLCO_CHECK(c->phase == 0, DLCO_ERR_INVALID, "dlco_comm_init_host: step in flight");
        DLCO_HIP(hipSetDevice(c->cfg.device));
        delete c->hostcomm;
        c->hostcomm = nullptr;
        // sharded: a slot holds a rank's column slab of a tracker product; replicated: a piece (<= 32 MB) of the F x F gradient
        const size_t slab = c->shard ? c->comm.gather_floats * sizeof(float) / c->cfg.world
                                     : std::min((size_t)c->F * c->F * sizeof(float), (size_t)32 << 20);
        const size_t slot = std::max((size_t)2 * c->Bl * sizeof(float), slab);
        c->hostcomm = new HostComm(shm_name, c->cfg.rank, c->cfg.world, slot);
    });
}

int dlco_comm_destroy(dlco_ctx *c)
{
    if (!c) return DLCO_ERR_INVALID;
    return guarded(c, [&] { sync(c); delete c->rccl; c->rccl = nullptr; delete c->hostcomm; c->hostcomm = nullptr; });
}

int dlco_sync(dlco_ctx *c) { return c ? guarded(c, [&] { sync(c); }) : DLCO_ERR_INVALID; }

int dlco_get_batch(const dlco_ctx *cc, int32_t *pos_rows, int32_t *neg_rows, float *pos_dist, float *neg_dist,
                   int32_t *rho, int32_t *kappa)
{
    dlco_ctx *c = const_cast<dlco_ctx *>(cc);
    if (!c) return DLCO_ERR_INVALID;
    return guarded(c, [&] {
        const size_t B = c->B;
        if (pos_rows) std::memcpy(pos_rows, c->h_pos_rows.data(), B * sizeof(int32_t));
        if (neg_rows) std::memcpy(neg_rows, c->h_neg_rows.data(), B * sizeof(int32_t));
        DLCO_CHECK(c->pd_cur && c->nd_cur, DLCO_ERR_INVALID, "dlco_get_batch: no step has run");
        if (pos_dist) d2h(c, pos_dist, c->pd_cur, B * sizeof(float));
        if (neg_dist) d2h(c, neg_dist, c->nd_cur, B * sizeof(float));
        if (rho) d2h(c, rho, c->rho.p, B * sizeof(int32_t));
        if (kappa) d2h(c, kappa, c->kappa.p, B * sizeof(int32_t));
    });
}

int dlco_get_t(const dlco_ctx *c, uint32_t *t)
{
    if (!c || !t) return DLCO_ERR_INVALID;
    *t = c->t;
    return DLCO_OK;
}

int dlco_get_W(dlco_ctx *c, float *W_host, int32_t *r) { return c ? guarded(c, [&] { get_W_host(c, W_host, r); }) : DLCO_ERR_INVALID; }

int dlco_get_A(dlco_ctx *c, float *A_host)
{
    if (!c || !A_host) return DLCO_ERR_INVALID;
    return guarded(c, [&] {
        DLCO_CHECK(c->phase == 0, DLCO_ERR_INVALID, "dlco_get_A: step in flight");
        build_A(c, c->W.p, c->r);
        d2h_square(c, A_host, c->grad.p);
    });
}

int dlco_get_dfavg(dlco_ctx *c, float *out)
{
    if (!c || !out) return DLCO_ERR_INVALID;
    return guarded(c, [&] {
        if (c->packed) {                                         // unpack into the F x F scratch first
            DLCO_CHECK(c->phase == 0, DLCO_ERR_INVALID, "dlco_get_dfavg: step in flight");
            syrk_unpack_upper(c->dfavg.p, c->F, c->grad.p, c->F, c->stream);
            d2h_square(c, out, c->grad.p);
            return;
        }
        d2h_square(c, out, c->dfavg.p);
    });
}

int dlco_set_state(dlco_ctx *c, uint32_t t, const float *dfavg_host, const float *W_host, int32_t r)
{
    if (!c) return DLCO_ERR_INVALID;
    return guarded(c, [&] {
        DLCO_CHECK(c->phase == 0, DLCO_ERR_INVALID, "dlco_set_state: step in flight");
        DLCO_CHECK(r >= 0 && r <= c->w_cap, DLCO_ERR_INVALID, "dlco_set_state: r out of range");
        c->t = t;
        if (dfavg_host && c->packed) {                            // the upper tiles are all that is kept
            h2d_square(c, c->grad.p, dfavg_host);
            syrk_mirror_upper(c->grad.p, c->F, c->F, c->stream);   // (diagonal tiles are stored whole: make them exactly symmetric)
            syrk_pack_upper(c->grad.p, c->F, c->F, c->dfavg.p, c->stream);
        } else if (dfavg_host) {
            h2d_square(c, c->dfavg.p, dfavg_host);
            if (!c->shard) syrk_mirror_upper(c->dfavg.p, c->F, c->F, c->stream);
        }
        c->eig->reset();
        if (W_host && r > 0) {
            h2d_rows(c, c->W.p, W_host, (size_t)r, true);
            c->r = r;
            c->eig->seed_rows(c->W.p, c->F, nullptr, r);
            // trace(A) of the regulariser (src/pj-learn.cpp:527) for this W: trace(W^T W) = |W|_F^2
            double tr = 0.0;
            for (size_t i = 0; i < (size_t)r * c->Fu; i++) tr += (double)W_host[i] * (double)W_host[i];
            c->traceA = tr;
        } else {
            c->r = 0;
            c->traceA = 0.0;
        }
    });
}

int dlco_validate(dlco_ctx *c, float *loss_val, float *regul, int32_t *rank)
{
    if (!c || !loss_val || !regul) return DLCO_ERR_INVALID;
    return guarded(c, [&] { DLCO_HIP(hipSetDevice(c->cfg.device)); validate(c, loss_val, regul, rank); });
}

int dlco_stats(dlco_ctx *c, const float *W_host, int32_t r, int32_t *dim, float *fpr95, double *auc)
{
    if (!c || !fpr95 || !auc) return DLCO_ERR_INVALID;
    return guarded(c, [&] {
        DLCO_HIP(hipSetDevice(c->cfg.device));
        if (!W_host) { stats(c, c->W.p, c->r, dim, fpr95, auc); return; }
        // S1 (src/misc.cpp:269-277): keep the rows of W that have a non-zero entry
        std::vector<float> nz;
        int rows = 0;
        for (int i = 0; i < r; i++) {
            const float *row = W_host + (size_t)i * c->Fu;
            bool any = false;
            for (int f = 0; f < c->Fu && !any; f++) any = row[f] != 0.0f;
            if (any) { nz.insert(nz.end(), row, row + c->Fu); rows++; }
        }
        DevBuf<float> Wd;
        Wd.alloc((size_t)std::max(rows, 1) * c->F);
        if (rows) h2d_rows(c, Wd.p, nz.data(), (size_t)rows, true);
        stats(c, Wd.p, rows, dim, fpr95, auc);
    });
}

int dlco_project_sqdist(dlco_ctx *c, const int32_t *row_ids_host, int32_t n, const float *W_host, int32_t r,
                        float *out_host)
{
    if (!c || !row_ids_host || !out_host || n < 0 || r < 0) return DLCO_ERR_INVALID;
    return guarded(c, [&] {
        DLCO_HIP(hipSetDevice(c->cfg.device));
        DLCO_CHECK(c->have_data, DLCO_ERR_INVALID, "dlco_project_sqdist: no data set");
        if (n == 0) return;
        for (int i = 0; i < n; i++)
            DLCO_CHECK(row_ids_host[i] >= 0 && row_ids_host[i] < c->N, DLCO_ERR_INVALID, "dlco_project_sqdist: row id out of range");
        DevBuf<int32_t> ids; ids.alloc(n);
        DevBuf<float> Wd, out; Wd.alloc((size_t)std::max(r, 1) * c->F); out.alloc(n);
        h2d(c, ids.p, row_ids_host, (size_t)n * sizeof(int32_t));
        if (r) h2d_rows(c, Wd.p, W_host, (size_t)r, true);
        if (n <= 4096) project_few(c, ids.p, n, Wd.p, r, out.p);
        else project_many(c, ids.p, 0, n, Wd.p, r, out.p);
        d2h(c, out_host, out.p, (size_t)n * sizeof(float));
    });
}

int dlco_viol_counts(dlco_ctx *c, const float *pd_host, const float *nd_host, int32_t B, int32_t *rho_host,
                     int32_t *kappa_host)
{
    if (!c || !pd_host || !nd_host || !rho_host || !kappa_host || B < 0) return DLCO_ERR_INVALID;
    return guarded(c, [&] {
        DLCO_HIP(hipSetDevice(c->cfg.device));
        if (B == 0) return;
        DevBuf<float> pd, nd; DevBuf<int32_t> rho, kap;
        pd.alloc(B); nd.alloc(B); rho.alloc(B); kap.alloc(B);
        h2d(c, pd.p, pd_host, (size_t)B * sizeof(float));
        h2d(c, nd.p, nd_host, (size_t)B * sizeof(float));
        viol_counts(pd.p, nd.p, B, rho.p, kap.p, c->stream);
        d2h(c, rho_host, rho.p, (size_t)B * sizeof(int32_t));
        d2h(c, kappa_host, kap.p, (size_t)B * sizeof(int32_t));
    });
}

int dlco_grad_rda(dlco_ctx *c, const int32_t *pos_rows_host, const int32_t *neg_rows_host, const int32_t *rho_host,
                  const int32_t *kappa_host, int32_t B, float alpha, float beta, const float *dfavg_in_host,
                  float *dfavg_out_host)
{
    if (!c || !pos_rows_host || !neg_rows_host || !rho_host || !kappa_host || !dfavg_out_host || B < 1) return DLCO_ERR_INVALID;
    return guarded(c, [&] {
        DLCO_HIP(hipSetDevice(c->cfg.device));
        DLCO_CHECK(c->have_data && c->phase == 0, DLCO_ERR_INVALID, "dlco_grad_rda: no data / step in flight");
        for (int i = 0; i < B; i++)
            DLCO_CHECK(pos_rows_host[i] >= 0 && pos_rows_host[i] < c->N && neg_rows_host[i] >= 0 && neg_rows_host[i] < c->N,
                       DLCO_ERR_INVALID, "dlco_grad_rda: row id out of range");
        DevBuf<int32_t> pr, nr, rho, kap, ids; DevBuf<float> w; DevBuf<int> k;
        pr.alloc(B); nr.alloc(B); rho.alloc(B); kap.alloc(B); ids.alloc((2 * B + 31) & ~31); w.alloc((2 * B + 31) & ~31); k.alloc(1);
        h2d(c, pr.p, pos_rows_host, (size_t)B * sizeof(int32_t));
        h2d(c, nr.p, neg_rows_host, (size_t)B * sizeof(int32_t));
        h2d(c, rho.p, rho_host, (size_t)B * sizeof(int32_t));
        h2d(c, kap.p, kappa_host, (size_t)B * sizeof(int32_t));
        const size_t FF = (size_t)c->F * c->F;
        if (dfavg_in_host) {
            h2d_square(c, c->grad.p, dfavg_in_host);
            if (!c->shard) syrk_mirror_upper(c->grad.p, c->F, c->F, c->stream);
        } else fill_f32(c->grad.p, 0.f, FF, c->stream);
        build_active_rows(pr.p, nr.p, rho.p, kap.p, B, 0, B, ids.p, w.p, k.p, c->stream);
        if (c->packed) {                                         // the trainer's own kernel and layout: pack, update, unpack
            DevBuf<float> pk;
            pk.alloc(syrk_packed_floats(c->F));
            syrk_pack_upper(c->grad.p, c->F, c->F, pk.p, c->stream);
            grad_syrk(c, ids.p, w.p, k.p, 2 * B, alpha, beta, pk.p, true);
            syrk_unpack_upper(pk.p, c->F, c->grad.p, c->F, c->stream);
            d2h_square(c, dfavg_out_host, c->grad.p);
            return;
        }
        grad_syrk(c, ids.p, w.p, k.p, 2 * B, alpha, beta, c->grad.p);
        d2h_square(c, dfavg_out_host, c->grad.p);
    });
}

int dlco_psd_project(dlco_ctx *c, const float *dfavg_host, uint32_t t, float *W_host, int32_t *r, float *A_host)
{
    if (!c || !dfavg_host || !r) return DLCO_ERR_INVALID;
    return guarded(c, [&] {
        DLCO_HIP(hipSetDevice(c->cfg.device));
        DLCO_CHECK(c->phase == 0, DLCO_ERR_INVALID, "dlco_psd_project: step in flight");
        const size_t FF = (size_t)c->F * c->F;
        DevBuf<float> G, Gp;
        G.alloc(FF);
        h2d_square(c, G.p, dfavg_host);
        if (c->packed) {                                         // the tracker of a packed context reads packed tiles
            Gp.alloc(syrk_packed_floats(c->F));
            syrk_mirror_upper(G.p, c->F, c->F, c->stream);
            syrk_pack_upper(G.p, c->F, c->F, Gp.p, c->stream);
        }
        c->eig->reset();
        const float cscale = (float)(std::sqrt((double)t + 1.0) / (double)c->cfg.gamma);
        double tr = 0.0;
        bool conv = true;
        DevBuf<float> Wd;
        Wd.alloc((size_t)c->w_cap * c->F);
        const int rr = c->eig->update(c->packed ? Gp.p : G.p, c->cfg.mu, cscale, Wd.p, &tr, &conv);
        c->eig->reset();
        if (rr > 0) {
            if (W_host) d2h_rows(c, W_host, Wd.p, (size_t)rr);
            *r = rr;
        } else {
            if (W_host) std::memset(W_host, 0, (size_t)c->Fu * c->Fu * sizeof(float));
            *r = c->Fu;
        }
        if (A_host) {
            build_A(c, Wd.p, rr);
            d2h_square(c, A_host, c->grad.p);
        }
        if (!conv) throw Error(DLCO_ERR_NOCONV, "dlco_psd_project: tracker did not reach its tolerance");
    });
}

int dlco_sym_product(dlco_ctx *c, const float *X_host, int32_t rows, const float *G_host, int32_t mode, float *out_host)
{
    if (!c || !X_host || !G_host || !out_host || rows < 1 || rows > ((mode == 1 || mode == 3) ? 160 : 128)) return DLCO_ERR_INVALID;
    return guarded(c, [&] {
        DLCO_HIP(hipSetDevice(c->cfg.device));
        const int F = c->F, pad = ((rows + 31) / 32) * 32;
        DevBuf<float> X, G, out;
        X.alloc((size_t)pad * F); G.alloc((size_t)F * F); out.alloc((size_t)pad * F);
        X.zero(c->stream);
        h2d_rows(c, X.p, X_host, (size_t)rows);
        h2d_square(c, G.p, G_host);
        bool ok;
        if (mode == 3 || mode == 4) {                            // packed upper tiles, every tile fetched once
            DevBuf<char> hi, lo, lo2;
            hi.alloc(bf16x2_plane_bytes(rows, F)); lo.alloc(bf16x2_plane_bytes(rows, F));
            if (mode == 4) lo2.alloc(bf16x2_plane_bytes(rows, F));
            DevBuf<float> slab, Gp;
            slab.alloc((size_t)4 * pad * F);
            Gp.alloc(syrk_packed_floats(F));
            syrk_mirror_upper(G.p, F, F, c->stream);              // diagonal tiles are stored whole: take their lower half from the upper
            syrk_pack_upper(G.p, F, F, Gp.p, c->stream);
            ok = skinny_product_sym(X.p, F, rows, Gp.p, F, 1.0f, out.p, F, nullptr, 0.f, nullptr, 0.f, hi.p, lo.p, slab.p, c->stream,
                                    mode == 4 ? lo2.p : nullptr);
            sync(c);
        } else if (mode == 1 || mode == 2) {
            DevBuf<char> hi, lo, lo2;
            hi.alloc(bf16x2_plane_bytes(rows, F)); lo.alloc(bf16x2_plane_bytes(rows, F));
            if (mode == 2) lo2.alloc(bf16x2_plane_bytes(rows, F));
            DevBuf<float> slab;
            slab.alloc(bf16x2_slab_floats(rows, F));
            ok = skinny_product_bf16x2(X.p, F, rows, G.p, F, F, F, 1.0f, out.p, F, nullptr, 0.f, nullptr, 0.f, hi.p, lo.p, slab.p,
                                       c->stream, 0, mode == 2 ? lo2.p : nullptr);
            sync(c);
        } else {
            ok = skinny_product_f32(X.p, F, rows, pad, G.p, F, F, F, 1.0f, out.p, F, nullptr, 0.f, nullptr, 0.f, c->stream);
        }
        DLCO_CHECK(ok, DLCO_ERR_INVALID, "dlco_sym_product: shape not supported by this mode");
        d2h_rows(c, out_host, out.p, (size_t)rows);
    });
}

int dlco_hinge_sum(dlco_ctx *c, const float *pos_host, int32_t n_pos, const float *neg_host, int32_t n_neg, double *out)
{
    if (!c || !out || n_pos < 0 || n_neg < 0) return DLCO_ERR_INVALID;
    return guarded(c, [&] {
        DLCO_HIP(hipSetDevice(c->cfg.device));
        *out = 0.0;
        if (n_pos == 0) return;
        DevBuf<float> p, n, rows;
        p.alloc(n_pos); n.alloc(std::max(n_neg, 1)); rows.alloc(n_pos);
        h2d(c, p.p, pos_host, (size_t)n_pos * sizeof(float));
        if (n_neg) h2d(c, n.p, neg_host, (size_t)n_neg * sizeof(float));
        hinge_rows(p.p, n_pos, n.p, n_neg, rows.p, c->stream);
        sum_f32_to_f64(rows.p, n_pos, c->dscal.p, c->stream);
        d2h(c, out, c->dscal.p, sizeof(double));
    });
}

int dlco_roc_stats(dlco_ctx *c, const float *dist_host, const uint8_t *labels_host, int32_t n, float *fpr95, double *auc)
{
    if (!c || !dist_host || !labels_host || !fpr95 || !auc || n < 1) return DLCO_ERR_INVALID;
    return guarded(c, [&] {
        DLCO_HIP(hipSetDevice(c->cfg.device));
        DevBuf<float> d; DevBuf<uint8_t> l;
        d.alloc(n); l.alloc(n);
        h2d(c, d.p, dist_host, (size_t)n * sizeof(float));
        h2d(c, l.p, labels_host, (size_t)n);
        RocWork *w = roc_work_create(n);
        try { roc_stats(w, d.p, l.p, n, fpr95, auc, c->stream); } catch (...) { roc_work_destroy(w); throw; }
        roc_work_destroy(w);
    });
}

int dlco_log_step(dlco_ctx *c, dlco_log_entry *out)
{
    if (!c || !out) return DLCO_ERR_INVALID;
    return guarded(c, [&] {
        DLCO_HIP(hipSetDevice(c->cfg.device));
        DLCO_CHECK(c->phase == 0, DLCO_ERR_INVALID, "dlco_log_step: step in flight");
        const auto t0 = std::chrono::steady_clock::now();
        std::memset(out, 0, sizeof(*out));
        // the reference prints the loop variable t of the iteration that just ran (src/pj-learn.cpp:538)
        out->t = c->t - 1;
        float loss = 0.f, regul = 0.f;
        int32_t rank = 0;
        validate(c, &loss, &regul, &rank);
        const auto t1 = std::chrono::steady_clock::now();
        out->loss_val = loss; out->regul = regul; out->obj = loss + regul; out->rank = rank;
        if ((loss + regul) < c->obj_best) {                        // :532
            c->obj_best = loss + regul;
            c->r_best = rank;
            out->is_best = 1;
            int32_t dim = 0; float f95 = 0.f; double auc = 0.0;
            stats(c, c->W.p, c->r, &dim, &f95, &auc);                // :551
            out->dim = dim; out->auc = auc; out->fpr95 = f95;
            if (c->auc_best <= auc && c->fpr95_best >= f95) {       // :558-559
                c->auc_best = auc; c->fpr95_best = f95;
                // W_Save = W.clone(), A_Save = A.clone() (:561-562): the clone is a device copy of the rows of W
                c->r_save_dev = c->r;
                c->r_save = c->r > 0 ? c->r : c->Fu;                 // no positive eigenvalue: W = zeros(F, F), :489-490
                if (c->r > 0) {
                    c->W_save_dev.alloc((size_t)c->w_cap * c->F);
                    DLCO_HIP(hipMemcpyAsync(c->W_save_dev.p, c->W.p, (size_t)c->r * c->F * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
                }
                out->saved = 1;
            }
        }
        out->obj_best = c->obj_best; out->rank_best = c->r_best;
        out->auc_best = c->auc_best; out->fpr95_best = c->fpr95_best;
        out->vtime = std::chrono::duration<double>(t1 - t0).count();
        out->nonconv = c->nonconv_window;
        c->nonconv_window = 0;
    });
}

int dlco_get_saved(dlco_ctx *c, float *W_host, int32_t *r, float *A_host)
{
    if (!c || !r) return DLCO_ERR_INVALID;
    *r = c->r_save;
    if (c->r_save <= 0 || (!W_host && !A_host)) return DLCO_OK;
    return guarded(c, [&] {
        DLCO_HIP(hipSetDevice(c->cfg.device));
        DLCO_CHECK(c->phase == 0, DLCO_ERR_INVALID, "dlco_get_saved: step in flight");
        if (W_host) {
            if (c->r_save_dev > 0) d2h_rows(c, W_host, c->W_save_dev.p, (size_t)c->r_save_dev);
            else std::memset(W_host, 0, (size_t)c->Fu * c->Fu * sizeof(float));
        }
        if (A_host) {
            build_A(c, c->W_save_dev.p, c->r_save_dev);              // A+ = W^T W (:472-478 builds the same matrix as Evec * Bmul)
            d2h_square(c, A_host, c->grad.p);
        }
    });
}

int dlco_profile_enable(dlco_ctx *c, int32_t on)
{
    if (!c) return DLCO_ERR_INVALID;
    return guarded(c, [&] {
        c->prof.reset();
        c->prof.on = on != 0;
        c->prof.mask = on == 2 ? (1u << PROF_GRAD_SYRK) : ~0u;
    });
}

int dlco_profile_read(dlco_ctx *c, const char *kernel, int64_t *launches, double *total_ms)
{
    if (!c || !launches || !total_ms) return DLCO_ERR_INVALID;
    return guarded(c, [&] {
        int slot = PROF_GRAD_SYRK;
        if (kernel && std::strcmp(kernel, "eig_product") == 0) slot = PROF_EIG_PRODUCT;
        else if (kernel && std::strcmp(kernel, "jacobi") == 0) slot = PROF_JACOBI;
        else if (kernel && std::strcmp(kernel, "project") == 0) slot = PROF_PROJECT;
        else if (kernel && std::strcmp(kernel, "rank_update") == 0) slot = PROF_RANK_UPDATE;
        else if (kernel && std::strcmp(kernel, "grad_syrk") != 0) throw Error(DLCO_ERR_INVALID, "dlco_profile_read: unknown kernel group");
        c->prof.drain(slot);
        *launches = c->prof.rec[slot].n;
        *total_ms = c->prof.rec[slot].ms;
    });
}

int dlco_counters(const dlco_ctx *c, int64_t out[8])
{
    if (!c || !out) return DLCO_ERR_INVALID;
    for (int i = 0; i < 8; i++) out[i] = 0;
    out[0] = c->steps_run; out[1] = c->active_rows_sum; out[2] = c->nonconv_steps;
    out[3] = c->eig->stats().jacobi_barrier_timeouts;
    out[4] = c->eig->stats().rank_update_passes;
    out[5] = (int64_t)(c->eig->stats().rank_update_check * 1e9);
    out[6] = c->eig->stats().locked_passes;
    out[7] = c->eig->stats().locked_rows;
    return DLCO_OK;
}

int dlco_eig_stats(const dlco_ctx *c, int64_t *iters, int64_t *product_rows, int64_t *jacobi_sweeps, int32_t *block_rows)
{
    if (!c) return DLCO_ERR_INVALID;
    const EigStats &s = c->eig->stats();
    if (iters) *iters = s.iters;
    if (product_rows) *product_rows = s.product_rows;
    if (jacobi_sweeps) *jacobi_sweeps = s.jacobi_sweeps;
    if (block_rows) *block_rows = c->eig->block_rows();
    return DLCO_OK;
}

}  // extern "C"
