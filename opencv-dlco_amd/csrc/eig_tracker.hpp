// eig_tracker.hpp — positive-eigenspace tracker for the PSD projection of the pj-learn
// step (E1/E2, src/pj-learn.cpp:434-490).
//
// The reference eigendecomposes the full F x F matrix A = -(sqrt(t+1)/gamma)(dfAvg + mu I)
// with LAPACKE_ssyevr every step and keeps only the eigenpairs with a positive eigenvalue.
// Here only those pairs are computed: with H = -dfAvg they are the eigenpairs of H above
// mu.  A block of m = r + guard Ritz vectors is carried from step to step and refreshed by
// Chebyshev-filtered subspace iteration (filter damps [lambda_min(H), lowest Ritz value of
// the block]) followed by one Rayleigh-Ritz step; all tall products are fp32 MFMA GEMMs,
// the m x m problems run on one workgroup (one-sided Jacobi).  The block grows when the
// positive eigenspace does; when it reaches F the method is the dense eigensolver.
#pragma once

#include <cstdlib>
#include "dlco_internal.hpp"

#include <functional>
#include <initializer_list>

namespace dlco {

struct EigStats {
    int64_t iters = 0, product_rows = 0, jacobi_sweeps = 0, updates = 0, nonconverged = 0, cheap_passes = 0;
    int64_t jacobi_barrier_timeouts = 0;   // multi-workgroup Jacobi calls that gave up at their grid barrier and were redone on one workgroup
    int64_t locked_passes = 0, locked_rows = 0;   // filter passes that ran with the converged top of the block locked, rows locked in them
    int64_t rank_update_passes = 0;        // first filter terms formed from the step's rank update instead of a pass over the matrix
    double rank_update_check = 0.0;        // DLCO_RANK_UPDATE_CHECK=1: largest |shortcut - product| / max|product| seen (developer aid)
};

// What the trainer knows about the matrix of the NEXT update(): G = beta * G_prev + alpha * X_a^T diag(w) X_a, G_prev the
// matrix of the last update(), X_a the active rows of the step (kernels_rankupd.hip).
struct RankUpdate {
    const float *proj = nullptr;     // [rows][ldp]: projections of the batch slots on the rows of W as the last update() emitted them (guard rows included)
    long ldp = 0;
    int rows = 0;
    const int32_t *slot = nullptr;   // slot[k]: column of proj that belongs to active row k
    const float *w = nullptr;        // weights of the active rows
    const int *k_dev = nullptr;      // their count (device)
    int kmax = 0;                    // capacity of the lists (multiple of 32, zero padded)
    const void *planes = nullptr;    // syrk_split_rows_kernel's planes of the active rows
    float alpha = 0.f, beta = 1.f;
    int planes_mode = 3;             // syrk_planes_mode(): 3 = three-way split planes, 1 = bf16-once planes
    bool coeff_ready = false;        // the coefficient fragments were written by the gradient's row split (rank_coeff_job)
};

class EigTracker {
public:
    // live_cols: the columns [live_cols, F) of every row of G are zero padding (dlco_api.cpp pads FeatDim to whole
    // tiles): the block never leaves the span of the first live_cols coordinates, and its size is bounded by it
    EigTracker(int F, int max_rows, int guard, float tol, int max_iter, hipStream_t stream, int live_cols = 0);
    ~EigTracker();

    // Forget the subspace (next update seeds from `seed_rows` or random vectors).
    void reset();
    // Seed the block with `n` rows gathered from a device matrix (ids may be NULL = rows 0..n-1;
    // with ids2 the seed row is src[ids] - src[ids2], the pair mode's descriptor difference).
    void seed_rows(const float *src, long ld, const int32_t *ids_dev, int n, const int32_t *ids2_dev = nullptr);

    // Track the eigenpairs of H = -G above mu.  cscale = sqrt(t+1)/gamma.
    // Writes W [r][F] (rows sqrt(cscale*(theta-mu)) * q, ascending like LAPACK) and returns r.
    // *trace receives sum of cscale*(theta-mu) (trace of the projected A).
    int update(const float *G, float mu, float cscale, float *W, double *trace, bool *converged);

    int block_rows() const { return m_; }
    void set_profiler(Profiler *p) { prof_ = p; }
    // G is sharded by columns over the ranks: every product with G computes the rank's column slab
    // and all-gathers the rest; everything else of the tracker is replicated (and deterministic).
    void set_shard(const ShardComm *sc) { shard_ = sc; }
    // Every G handed to update() is the PACKED upper-tile form (syrk_packed_floats(F) floats, kernels_syrk.hip): the
    // products take the symmetric kernel that fetches each tile once.  Only valid where packed_supported() holds.
    // One device int the caller wants read back with every Ritz block (the step's active row count): it rides in the
    // block's read-back instead of a copy of its own.  readback_extra() is its value as of the last update().
    void set_readback_extra(const int *dev) { extra_dev_ = dev; }
    // Where a growing block takes new directions from: the callback writes up to `want` rows (row stride F, on the
    // tracker's stream) and returns how many it wrote; what it does not provide is filled with random vectors.  The
    // trainer passes rows of the step's batch: the directions that enter the dual average come from their span.
    void set_growth_rows(std::function<int(float *dst, int want)> cb) { grow_cb_ = std::move(cb); }
    int readback_extra() const { return extra_val_; }
    // update() also writes the block's guard rows behind the nw rows of W (the caller's W holds block_rows() rows then) and
    // keeps the scale of every row: the step's own projection of its batch on these rows is what offer_rank_update() hands back
    void set_emit_guards(bool on) { emit_guards_ = on; }
    int ext_rows() const { return wext_rows_; }          // rows of W the last update() wrote, guard rows included (0: none)
    // consumed by the next update() only; ignored when the block is not the one the projection was made with
    void offer_rank_update(const RankUpdate &ru) { ru_ = ru; ru_offered_ = true; }
    // The part of the coefficient job the tracker knows (rows, scales, output buffer) for a row list of capacity kmax;
    // false when the block is not one the shortcut takes.  The caller adds proj / slot and lets the job ride in its gradient.
    bool rank_coeff_job(RankCoeffJob *job, int kmax);
    void set_packed(bool on) { packed_ = on; }
    bool packed() const { return packed_; }
    static bool packed_supported(int F) { return F % 128 == 0 && F / 128 <= 64 && std::getenv("DLCO_FP32_FILTER") == nullptr && std::getenv("DLCO_FP32_RR") == nullptr && std::getenv("DLCO_NO_PACKED") == nullptr; }
    const EigStats &stats() const { return st_; }
    float last_crit() const { return last_crit_; }

private:
    void product(const float *X, int rows, const float *G, float alpha, float *out, const float *E1, float b1,
                 const float *E2, float b2, bool approx);
    void gram(const float *X, const float *Y, int rows, float *T);
    void gram_rect(const float *X, int xrows, const float *Y, int yrows, float *T);
    void project_out(float *Wp, int np, const float *Q, int kept);
    void rotate(const float *C, long ldc, int k_in, int k_out, const float *X, float *out, const float *X2 = nullptr, float *out2 = nullptr);
    int orthonormalize(float *Z, int rows, float *scratch, const std::vector<int> &panel_ends, int skip = 0);
    int drop_dead_rows();
    void refresh_lower_bound(const float *G, int iters, float theta_top);
    void append_random(float *Q, int have, int add);
    float next_uniform();
    float *pick(std::initializer_list<const float *> busy) const;

    int F_, live_, cap_, guard_, max_iter_;
    float tol_;
    std::function<int(float *, int)> grow_cb_;
    const int *extra_dev_ = nullptr;
    int extra_val_ = 0;
    bool packed_ = false;
    bool emit_guards_ = false;
    int wext_rows_ = 0, wext_nw_ = 0;   // W as the last update() left it: rows in all, Ritz rows among them
    RankUpdate ru_;
    bool ru_offered_ = false;
    bool ru_check_ = std::getenv("DLCO_RANK_UPDATE_CHECK") != nullptr;
    DevBuf<float> wscale_;
    DevBuf<char> coeff_;
    const float *chain_planes_of_ = nullptr;   // the matrix whose two-way planes sit in plane_hi_/plane_lo_ (emitted by a reduction)
    int chain_rows_ = 0;
    bool chain_next_ = false;        // the product being issued feeds the next one of a filter chain
    bool cold_ = true;               // no update since construction / reset(): the next one converges to tol_ / 4
    hipStream_t s_;
    int m_ = 0;                      // rows currently in the block
    bool have_theta_ = false;
    bool theta_dev_valid_ = false;   // evals_ on the device matches h_theta_ row for row
    float lo_bound_ = 0.f;           // lower bound of H's spectrum
    bool have_lo_ = false;
    int steps_since_lo_ = 0;
    uint64_t rng_ = 0x243F6A8885A308D3ULL;
    float last_crit_ = 0.f;
    int deg0_ = 4;                   // filter degree of the first pass of a step (adapted)
    int last_deg_ = 0;               // degree actually used by the last filter (after the amplification cap)
    bool y_ok_ = false;              // Y_ = Q_ * H row for row for the matrix of the current update
    bool cheap_pass_ = std::getenv("DLCO_NO_CHEAP_PASS") == nullptr;
    float cheap_margin_ = 1.6f;         // see update(): how narrow a miss must be for a pass without products
    bool lock_ = std::getenv("DLCO_NO_LOCKING") == nullptr;           // see update(): converged top pairs leave the filter
    float tol_pass1_ = std::getenv("DLCO_EIG_TOL_PASS1") ? (float)std::atof(std::getenv("DLCO_EIG_TOL_PASS1")) : 0.85f;  // ... and of the first
    float tol_pass2_ = 0.5f;            // see update(): tolerance factor of the passes after the first
    double panel_amp_ = 1e5;         // largest filter-amplification ratio inside one orthonormalisation panel
    bool debug_ = std::getenv("DLCO_EIG_DEBUG") != nullptr;
    bool guard_stop_ = std::getenv("DLCO_JACOBI_ALL_PAIRS") == nullptr;       // see update(): guard-guard pairs do not prolong Jacobi
    bool weighted_crit_ = std::getenv("DLCO_EIG_UNIFORM_CRIT") == nullptr;   // see update(): residuals weighted by their share in A

    DevBuf<float> buf_[6];           // Ritz vectors, their H-products and temporaries, each cap x F
    float *all_[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    float *Q_ = nullptr, *Y_ = nullptr;
    DevBuf<float> Tm_, Vm_, Cw_, jwork_, slab_, pv_, pw_, scale_;
    // Ritz values, residual norms and the Jacobi sweep count sit in one device block
    // [evals: cap+8 | res: cap+8 | sweeps (int)] so that one copy brings them to the host
    struct FView { float *p = nullptr; };
    DevBuf<float> ritz_block_;
    FView evals_, res_;
    int *sweeps_dev_ = nullptr;
    DevBuf<int32_t> srcrow_;
    bool bf16_filter_ = false;       // filter products as split-bf16 MFMA (kernels_bf16x2.hip)
    int ks_ = 4;                     // K slices of such a product over a full matrix (a divisor of F / 128)
    DevBuf<char> plane_hi_, plane_lo_, plane_lo2_;
    DevBuf<int> dead_;               // dead-row flags of the panel being factored
    size_t slab_floats_ = 0;
    std::vector<float> h_theta_, h_res_, h_tmp_, h_sc_;
    std::vector<int32_t> h_sr_;
    float *pin_ = nullptr;           // pinned host staging
    unsigned publish_seq_ = 0;
    bool poll_readback_ = std::getenv("DLCO_SYNC_READBACK") == nullptr;
    size_t pin_floats_ = 0;
    EigStats st_;
    Profiler *prof_ = nullptr;
    const ShardComm *shard_ = nullptr;
};

}  // namespace dlco
