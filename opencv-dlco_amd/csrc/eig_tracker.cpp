// eig_tracker.cpp — host orchestration of the positive-eigenspace tracker (see the header).
#include "eig_tracker.hpp"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstring>
#include <initializer_list>

namespace dlco {

namespace {
inline int ceil_div(long a, long b) { return (int)((a + b - 1) / b); }
}  // namespace

EigTracker::EigTracker(int F, int max_rows, int guard, float tol, int max_iter, hipStream_t stream, int live_cols)
    : F_(F), live_(live_cols > 0 ? std::min(live_cols, F) : F), guard_(guard), max_iter_(max_iter), tol_(tol), s_(stream)
{
    cap_ = std::min(live_, std::max(max_rows, 2 * guard + 32));
    if (guard_ > cap_ / 2) guard_ = std::max(1, cap_ / 2);
    for (int i = 0; i < 6; i++) { buf_[i].alloc((size_t)cap_ * F_); all_[i] = buf_[i].p; }
    Q_ = all_[0]; Y_ = all_[1];
    Tm_.alloc((size_t)cap_ * cap_);
    Vm_.alloc((size_t)cap_ * cap_);
    Cw_.alloc((size_t)cap_ * cap_);
    ritz_block_.alloc((size_t)2 * (cap_ + 8) + 8);
    evals_.p = ritz_block_.p;
    res_.p = ritz_block_.p + (cap_ + 8);
    sweeps_dev_ = reinterpret_cast<int *>(ritz_block_.p + 2 * (cap_ + 8));
    DLCO_HIP(hipMemsetAsync(sweeps_dev_, 0, 8 * sizeof(int), s_));   // [0] Jacobi sweeps, [4] ticket of the residual / publish kernel
    scale_.alloc(cap_ + 8);
    wscale_.alloc(cap_ + 8);
    srcrow_.alloc(cap_ + 8);
    jwork_.alloc(jacobi_work_floats(cap_));
    dead_.alloc(CHOL_INV_MAX_N);
    if (const char *e = std::getenv("DLCO_PANEL_AMP")) panel_amp_ = std::max(1.0, std::atof(e));
    // filter products run on the bf16 matrix cores with split operands unless DLCO_FP32_FILTER is set
    bf16_filter_ = (F_ % 128 == 0) && std::getenv("DLCO_FP32_FILTER") == nullptr;
    // K slices of a product over a full (unpacked) matrix: a divisor of the tile count, four where it divides
    {
        const int nt = F_ / 128;
        ks_ = 1;
        for (int d : {4, 2, 8, 7, 6, 5, 3})
            if (nt > 0 && nt % d == 0) { ks_ = d; break; }
    }
    if (bf16_filter_) {
        plane_hi_.alloc(bf16x2_plane_bytes(std::min(cap_, 160), F_));
        plane_lo_.alloc(bf16x2_plane_bytes(std::min(cap_, 160), F_));
        // the Rayleigh-Ritz product takes the three-way split (fp32-level accuracy) unless DLCO_FP32_RR is set
        if (std::getenv("DLCO_FP32_RR") == nullptr) plane_lo2_.alloc(bf16x2_plane_bytes(std::min(cap_, 128), F_));
    }
    pv_.alloc(F_);
    pw_.alloc(F_);
    slab_floats_ = std::max((size_t)4 * cap_ * F_, (size_t)8 << 20);
    slab_floats_ = std::min(slab_floats_, (size_t)64 << 20);
    slab_floats_ = std::max(slab_floats_, (size_t)cap_ * F_);
    slab_.alloc(slab_floats_);
    pin_floats_ = (size_t)3 * cap_ + 64;                  // >= the Ritz block (2*(cap+8)+1 floats); the last 16 hold the poll flag, the 16 before them the caller's extra word
    // coherent (fine-grained) on purpose: the host polls a sequence number that a kernel writes with a
    // system-scope release; with HIP_HOST_COHERENT=0 a default allocation would only be seen at a sync
    DLCO_HIP(hipHostMalloc((void **)&pin_, pin_floats_ * sizeof(float), hipHostMallocCoherent));
    std::memset(pin_, 0, pin_floats_ * sizeof(float));
    h_theta_.assign(cap_, 0.f);
    h_res_.assign(cap_, 0.f);
}

EigTracker::~EigTracker()
{
    if (pin_) (void)hipHostFree(pin_);
}

float *EigTracker::pick(std::initializer_list<const float *> busy) const
{
    for (int i = 0; i < 6; i++) {
        bool used = false;
        for (const float *b : busy) used = used || (b == all_[i]);
        if (!used) return all_[i];
    }
    throw Error(-2, "eig tracker: no free work buffer");
}

void EigTracker::reset()
{
    m_ = 0;
    have_theta_ = false;
    have_lo_ = false;
    steps_since_lo_ = 0;
    cold_ = true;
    wext_rows_ = 0;
    ru_offered_ = false;
}

void EigTracker::seed_rows(const float *src, long ld, const int32_t *ids_dev, int n, const int32_t *ids2_dev)
{
    const int k = std::min(n, cap_);
    have_theta_ = false;
    wext_rows_ = 0;
    ru_offered_ = false;
    if (k <= 0) { m_ = 0; return; }
    scale_rows(Q_, F_, src, ld, nullptr, ids_dev, k, F_, s_, ids2_dev);
    m_ = k;
}

bool EigTracker::rank_coeff_job(RankCoeffJob *job, int kmax)
{
    if (wext_rows_ <= 0 || wext_rows_ != m_ || m_ > 160 || !packed_ || !bf16_filter_ || shard_ || kmax % 32 != 0) return false;
    coeff_.alloc(rank_coeff_bytes(cap_ < 160 ? cap_ : 160, kmax));
    job->nw = wext_nw_; job->m = m_; job->MT = (m_ + 31) / 32;
    job->wscale = wscale_.p; job->frag = coeff_.p;
    return true;
}

float EigTracker::next_uniform()
{
    rng_ ^= rng_ << 13; rng_ ^= rng_ >> 7; rng_ ^= rng_ << 17;
    return (float)((double)(rng_ >> 11) * (2.0 / 9007199254740992.0) - 1.0);
}

void EigTracker::append_random(float *Q, int have, int add)
{
    if (add <= 0) return;
    y_ok_ = false;
    h_tmp_.assign((size_t)add * F_, 0.f);
    for (int i = 0; i < add; i++)
        for (int j = 0; j < live_; j++) h_tmp_[(size_t)i * F_ + j] = next_uniform();   // (zero in the pad columns)
    DLCO_HIP(hipMemcpyAsync(Q + (size_t)have * F_, h_tmp_.data(), h_tmp_.size() * sizeof(float), hipMemcpyHostToDevice, s_));
    DLCO_HIP(hipStreamSynchronize(s_));
}

// out[rows][F] = alpha * X*G + b1*E1 + b2*E2     (G symmetric F x F)
void EigTracker::product(const float *X, int rows, const float *G, float alpha, float *out, const float *E1, float b1,
                         const float *E2, float b2, bool approx)
{
    // the split-bf16 kernels take up to 128 rows: a taller block goes through them in row chunks
    // (one HBM-bound pass over G per chunk, still cheaper than the generic fp32 GEMM)
    // (the row-streaming kernel takes 160 rows with the two-way split - a block of rank ~128 plus its guards
    // in ONE pass over G - and 96 with the three-way split; a sharded rank's slab kernel 128)
    const int one_pass = (approx && !shard_ && bf16_filter_) ? 160 : (packed_ ? 96 : 128);   // (symmetric three-way kernel: 96 rows)
    if (rows > one_pass && bf16_filter_ && (packed_ || F_ >= 256) && (approx || plane_lo2_.p)) {
        chain_next_ = false;                                     // row chunks share the plane buffers: no carried planes
        const int step = approx ? (rows <= 2 * one_pass ? (rows / 2 + 31) / 32 * 32 : one_pass) : 96;
        for (int r0 = 0; r0 < rows; r0 += step) {
            const int nr = std::min(step, rows - r0);
            const size_t o = (size_t)r0 * F_;
            product(X + o, nr, G, alpha, out + o, E1 ? E1 + o : nullptr, b1, E2 ? E2 + o : nullptr, b2, approx);
        }
        return;
    }
    st_.product_rows += rows;
    if (packed_) {
        // G = packed upper tiles: the symmetric kernel (every tile fetched once); two-way split for the filter, three-way
        // for the exact products (rows <= 96 there: the chunking above)
        if (prof_) prof_->begin(PROF_EIG_PRODUCT);
        // inside a filter chain the planes of X are those the previous product's reduction emitted (chain_next_)
        const bool ready = approx && chain_planes_of_ == X && chain_rows_ == rows;
        const bool emit = approx && chain_next_;
        const bool ok = skinny_product_sym(X, F_, rows, G, F_, alpha, out, F_, E1, b1, E2, b2, plane_hi_.p, plane_lo_.p, slab_.p, s_,
                                           approx ? nullptr : plane_lo2_.p, ready, emit);
        chain_planes_of_ = emit ? out : nullptr;                 // (a product without `emit` has overwritten the planes with X's)
        chain_rows_ = rows;
        if (prof_) prof_->end(PROF_EIG_PRODUCT);
        DLCO_CHECK(ok, -2, "eig tracker: the symmetric product rejected a shape of the packed layout");
        return;
    }
    if (shard_) {
        // the rank's column slab out[:, c0:c0+cw] = alpha * X * G[:, c0:c0+cw] + ..., then all-gather
        const int c0 = shard_->c0, cw = shard_->cw, world = shard_->world;
        DLCO_CHECK((size_t)world * rows * cw <= shard_->gather_floats, -2, "eig tracker: exchange buffer too small");
        const float *E1s = E1 ? E1 + c0 : nullptr, *E2s = E2 ? E2 + c0 : nullptr;
        if (prof_) prof_->begin(PROF_EIG_PRODUCT);
        bool done = false;
        if (bf16_filter_ && rows <= 128 && (approx || plane_lo2_.p)) {
            int ks = ks_;                                            // keep ~256 workgroups in flight
            while (ks < 4 * world && F_ % (128 * ks * 2) == 0 && bf16x2_slab_floats(rows, cw, ks * 2) <= slab_floats_) ks *= 2;
            done = skinny_product_bf16x2(X, F_, rows, G + c0, F_, cw, F_, alpha, out + c0, F_, E1s, b1, E2s, b2, plane_hi_.p,
                                         plane_lo_.p, slab_.p, s_, ks, approx ? nullptr : plane_lo2_.p);
        }
        if (!done) {
            GemmArgs g;
            g.M = rows; g.N = cw; g.K = F_;
            g.A.p = X; g.A.ld = F_; g.A.kmajor = false;
            g.B.p = G + c0; g.B.ld = F_; g.B.kmajor = true;
            g.C = out + c0; g.ldc = F_;
            g.alpha = alpha; g.E1 = E1s; g.b1 = b1; g.E2 = E2s; g.b2 = b2;
            const long tiles = (long)ceil_div(rows, rows <= 64 ? 64 : 128) * ceil_div(cw, 128);
            long split = std::max(1L, std::min((long)ceil_div(512, tiles), (long)F_ / 256));
            split = std::min(split, (long)(slab_floats_ / ((size_t)rows * cw)));
            g.split_k = (int)std::max(1L, split);
            g.slab = slab_.p;
            gemm_f32(g, s_);
        }
        if (prof_) prof_->end(PROF_EIG_PRODUCT);
        pack_cols(shard_->gather + (size_t)shard_->rank * rows * cw, out, F_, c0, cw, rows, s_);
        shard_->allgather((size_t)rows * cw * sizeof(float));
        unpack_cols(out, F_, shard_->gather, cw, rows, world, s_);
        return;
    }
    if (bf16_filter_ && rows <= one_pass && F_ >= 256 && (approx || plane_lo2_.p)) {
        // filter products: two-way split (~1e-5); exact products (Rayleigh-Ritz): three-way split (~1e-7)
        if (prof_) prof_->begin(PROF_EIG_PRODUCT);
        const bool ok = skinny_product_bf16x2(X, F_, rows, G, F_, F_, F_, alpha, out, F_, E1, b1, E2, b2, plane_hi_.p,
                                              plane_lo_.p, slab_.p, s_, ks_, approx ? nullptr : plane_lo2_.p);
        if (prof_) prof_->end(PROF_EIG_PRODUCT);
        if (ok) return;
    }
    if (rows <= 128 && F_ >= 256) {
        if (prof_) prof_->begin(PROF_EIG_PRODUCT);
        skinny_product_f32(X, F_, rows, cap_, G, F_, F_, F_, alpha, out, F_, E1, b1, E2, b2, s_);
        if (prof_) prof_->end(PROF_EIG_PRODUCT);
        return;
    }
    GemmArgs g;
    g.M = rows; g.N = F_; g.K = F_;
    g.A.p = X; g.A.ld = F_; g.A.kmajor = false;
    g.B.p = G; g.B.ld = F_; g.B.kmajor = true;
    g.C = out; g.ldc = F_;
    g.alpha = alpha; g.E1 = E1; g.b1 = b1; g.E2 = E2; g.b2 = b2;
    const int bm = rows <= 64 ? 64 : 128;
    const long tiles = (long)ceil_div(rows, bm) * ceil_div(F_, 128);
    long split = std::max(1L, std::min((long)ceil_div(1024, tiles), (long)F_ / 256));
    split = std::min(split, (long)(slab_floats_ / ((size_t)rows * F_)));
    g.split_k = (int)std::max(1L, split);
    g.slab = slab_.p;
    if (prof_) prof_->begin(PROF_EIG_PRODUCT);
    gemm_f32(g, s_);
    if (prof_) prof_->end(PROF_EIG_PRODUCT);
}

// T[rows][rows] (ld cap_) = X * Y^T
void EigTracker::gram(const float *X, const float *Y, int rows, float *T)
{
    GemmArgs g;
    g.M = rows; g.N = rows; g.K = F_;
    g.A.p = X; g.A.ld = F_; g.A.kmajor = false;
    g.B.p = Y; g.B.ld = F_; g.B.kmajor = false;
    g.C = T; g.ldc = cap_;
    const int bt = rows <= 64 ? 64 : 128;
    const long tiles = (long)ceil_div(rows, bt) * ceil_div(rows, bt);
    // (at most 64 K slices, none shorter than 64: 64 slices of 128 at F = 8192 - with 128 slices of 64 the launch is no faster
    // and the ordered reduce reads twice as much, 593 -> 608 k on c2 - and ten slices of 64 at the reference's F = 640, where five
    // of 128 measured 2.5 % slower)
    const long kslice = std::max(64L, (long)F_ / 64 / 64 * 64);
    long split = std::max(1L, std::min((long)ceil_div(512, tiles), (long)F_ / kslice));
    split = std::min(split, (long)(slab_floats_ / ((size_t)rows * rows)));
    g.split_k = (int)std::max(1L, split);
    g.slab = slab_.p;
    gemm_f32(g, s_);
}

// out[k_out][F] = C X with C [k_out][ldc] holding the coefficient vectors in its rows (jacobi_eigh's layout)
void EigTracker::rotate(const float *C, long ldc, int k_in, int k_out, const float *X, float *out, const float *X2, float *out2)
{
    GemmArgs g;
    g.M = k_out; g.N = F_; g.K = k_in;
    g.A.p = C; g.A.ld = ldc; g.A.kmajor = false;
    g.B.p = X; g.B.ld = F_; g.B.kmajor = true;
    g.C = out; g.ldc = F_;
    g.B2 = X2; g.C2 = out2;                                          // the Ritz step rotates Q and Y = Q H with one launch
    gemm_f32(g, s_);
}

// T[xrows][yrows] (ld cap_) = X * Y^T
void EigTracker::gram_rect(const float *X, int xrows, const float *Y, int yrows, float *T)
{
    GemmArgs g;
    g.M = xrows; g.N = yrows; g.K = F_;
    g.A.p = X; g.A.ld = F_; g.A.kmajor = false;
    g.B.p = Y; g.B.ld = F_; g.B.kmajor = false;
    g.C = T; g.ldc = cap_;
    const long tiles = (long)ceil_div(xrows, xrows <= 64 ? 64 : 128) * ceil_div(yrows, yrows <= 64 ? 64 : 128);
    const long kslice = std::max(64L, (long)F_ / 64 / 64 * 64);          // as in gram()
    long split = std::max(1L, std::min((long)ceil_div(512, tiles), (long)F_ / kslice));
    split = std::min(split, (long)(slab_floats_ / ((size_t)xrows * yrows)));
    g.split_k = (int)std::max(1L, split);
    g.slab = slab_.p;
    gemm_f32(g, s_);
}

// Wp[np][F] -= (Wp Q^T) Q for the `kept` orthonormal rows of Q
void EigTracker::project_out(float *Wp, int np, const float *Q, int kept)
{
    gram_rect(Wp, np, Q, kept, Cw_.p);
    GemmArgs g;
    g.M = np; g.N = F_; g.K = kept;
    g.A.p = Cw_.p; g.A.ld = cap_; g.A.kmajor = false;
    g.B.p = Q; g.B.ld = F_; g.B.kmajor = true;
    g.C = Wp; g.ldc = F_;
    g.alpha = -1.0f; g.beta = 1.0f;
    gemm_f32(g, s_);
}

// Orthonormalise the rows of Z in place (`scratch` is a second [rows][F] buffer); returns rows.
// Block Gram-Schmidt over panels of at most 128 rows: a panel is projected (twice) against the
// rows already accepted and renormalised, then CholQR2: Gram matrix -> L^-1 on one workgroup
// -> rows = L^-1 * rows as a GEMM, twice (Z -> scratch -> Z).  The rows arrive ordered by Ritz
// value, i.e. by how strongly the filter amplified them; the caller cuts the panels so that
// rows of different provenance or of very different amplification never share one.  Nothing
// here synchronises with the host and no row is dropped (dependent rows become zero rows).
int EigTracker::orthonormalize(float *Z, int rows, float *scratch, const std::vector<int> &panel_ends, int skip)
{
    // One panel (the steady state): Cholesky QR is invariant under a scaling of the rows - the Gram entries keep their
    // relative accuracy, the dead-row rule compares a pivot with the row's own diagonal entry - so the rows go in as the
    // filter left them.  With several panels the rows are normalised first: the Gram-Schmidt step between panels zeroes
    // what is left of a row after projection against a threshold that assumes unit rows.
    const bool one_panel = skip == 0 && rows <= CHOL_INV_MAX_N && (panel_ends.empty() || panel_ends[0] >= rows);
    if (!one_panel) row_normalize(Z + (size_t)skip * F_, F_, rows - skip, F_, s_);
    size_t pi = 0;
    // (the first `skip` rows are orthonormal as they are - locked Ritz vectors - and only serve as the span the panels
    // behind them are projected against)
    for (int p0 = skip; p0 < rows;) {
        while (pi < panel_ends.size() && panel_ends[pi] <= p0) pi++;
        int pend = pi < panel_ends.size() ? std::min(rows, panel_ends[pi]) : rows;
        if (pend - p0 > CHOL_INV_MAX_N) pend = p0 + CHOL_INV_MAX_N;
        const int np = pend - p0;
        float *Wp = Z + (size_t)p0 * F_, *Sp = scratch + (size_t)p0 * F_;
        if (p0 > 0) {
            for (int pass = 0; pass < 2; pass++) {
                project_out(Wp, np, Z, p0);
                // after the first projection, rows that lie (to fp32 accuracy) inside the span above are zeroed
                row_normalize(Wp, F_, np, F_, s_, pass == 0 ? 2e-5f : 0.f);
            }
        }
        for (int pass = 0; pass < 2; pass++) {
            const float *src = pass == 0 ? Wp : Sp;
            float *dst = pass == 0 ? Sp : Wp;
            gram(src, src, np, Tm_.p);
            chol_inverse128(Tm_.p, cap_, np, 1e-5f, Vm_.p, cap_, dead_.p, s_);
            GemmArgs g;                                           // dst = Linv * src
            g.M = np; g.N = F_; g.K = np;
            g.A.p = Vm_.p; g.A.ld = cap_; g.A.kmajor = false;
            g.B.p = src; g.B.ld = F_; g.B.kmajor = true;
            g.C = dst; g.ldc = F_;
            gemm_f32(g, s_);
        }
        p0 = pend;
    }
    return rows;
}

// Remove rows that the orthonormalisation zeroed (they surface as exact-zero Ritz pairs).
int EigTracker::drop_dead_rows()
{
    std::vector<int32_t> live;
    for (int i = 0; i < m_; i++)
        if (!(h_theta_[i] == 0.f && h_res_[i] == 0.f)) live.push_back(i);
    const int k = (int)live.size();
    if (k == m_ || k == 0) return m_;
    DLCO_HIP(hipMemcpyAsync(srcrow_.p, live.data(), (size_t)k * sizeof(int32_t), hipMemcpyHostToDevice, s_));
    DLCO_HIP(hipStreamSynchronize(s_));
    float *Qn = pick({Q_, Y_});
    scale_rows(Qn, F_, Q_, F_, nullptr, srcrow_.p, k, F_, s_);
    float *Yn = pick({Q_, Y_, Qn});
    scale_rows(Yn, F_, Y_, F_, nullptr, srcrow_.p, k, F_, s_);
    Q_ = Qn; Y_ = Yn;
    for (int j = 0; j < k; j++) { h_theta_[j] = h_theta_[live[j]]; h_res_[j] = h_res_[live[j]]; }
    m_ = k;
    return k;
}

// Power iteration for lambda_max(G); the lower end of H = -G is -lambda_max(G).
void EigTracker::refresh_lower_bound(const float *G, int iters, float theta_top)
{
    static const float kZero = 0.f;
    const bool cold = !have_lo_;
    if (cold) {
        std::vector<float> v(F_, 0.f);
        for (int i = 0; i < live_; i++) v[i] = next_uniform();
        DLCO_HIP(hipMemcpyAsync(pv_.p, v.data(), F_ * sizeof(float), hipMemcpyHostToDevice, s_));
        DLCO_HIP(hipStreamSynchronize(s_));
        row_normalize(pv_.p, F_, 1, F_, s_);
    }
    DLCO_HIP(hipMemcpyAsync(scale_.p, &kZero, sizeof(float), hipMemcpyHostToDevice, s_));   // theta = 0: plain norm
    auto run = [&](float shift, int n) -> float {
        float *v = pv_.p, *w = pw_.p;
        for (int i = 0; i < n; i++) {
            if (shard_ || packed_) product(v, 1, G, 1.0f, w, nullptr, 0.f, nullptr, 0.f, false);   // v^T G = (G v)^T, G symmetric
            else symv(G, F_, F_, v, w, s_);
            if (shift != 0.f) axpby_inplace(w, v, 1.0f, shift, F_, s_);
            if (i == n - 1) residual_norms(v, w, F_, scale_.p, 1, F_, res_.p, s_);   // |w|
            row_normalize(w, F_, 1, F_, s_);
            std::swap(v, w);
        }
        if (v != pv_.p) DLCO_HIP(hipMemcpyAsync(pv_.p, v, F_ * sizeof(float), hipMemcpyDeviceToDevice, s_));
        DLCO_HIP(hipMemcpyAsync(pin_, res_.p, sizeof(float), hipMemcpyDeviceToHost, s_));
        DLCO_HIP(hipStreamSynchronize(s_));
        return pin_[0];
    };
    // |G v| <= spectral radius; the shift must dominate -lambda_min(G) = top eigenvalue of H
    const float rho = run(0.f, cold ? 12 : 2);
    const float shift = std::max(cold ? 2.0f * rho : 1.05f * rho, 1.1f * std::max(theta_top, 0.f)) + 1e-30f;
    const float top = run(shift, iters) - shift;        // <= lambda_max(G), tight once converged
    const float lam_max = std::max(top, 0.f);
    lo_bound_ = -(1.3f * lam_max + 0.05f * std::max(rho, theta_top)) - 1e-12f;
    have_lo_ = true;
    steps_since_lo_ = 0;
}

int EigTracker::update(const float *G, float mu, float cscale, float *W, double *trace, bool *converged)
{
    st_.updates++;
    // The caller may have told what changed: G = beta G_prev + alpha X_a^T diag(w) X_a (offer_rank_update).  With Y_ = Q_ H_prev
    // still there from the last Rayleigh-Ritz step and the projections of X_a on exactly these rows, the first filter
    // term needs no pass over G (kernels_rankupd.hip).
    const bool ru_ok = ru_offered_ && y_ok_ && have_theta_ && packed_ && bf16_filter_ && !shard_ && wext_rows_ == m_ &&
                       ru_.rows == m_ && m_ >= 1 && m_ <= 160 && m_ < live_ && ru_.proj && ru_.planes;
    ru_offered_ = false;
    const int ru_nw = wext_nw_;
    wext_rows_ = 0;                                                  // W is rewritten below
    y_ok_ = false;                                                   // G changed since the last Rayleigh-Ritz step
    if (m_ == 0) {
        m_ = std::min(cap_, 2 * guard_ + 32);
        append_random(Q_, 0, m_);
        have_theta_ = false;
    }
    float theta_top = have_theta_ ? h_theta_[0] : 0.f;
    float block_min = have_theta_ ? std::min(h_theta_[m_ - 1], mu) : mu;
    steps_since_lo_++;
    const int period = st_.updates < 20 ? 1 : (st_.updates < 200 ? 5 : (st_.updates < 400 ? 20 : 50));   // lambda_max(G) drifts as 1/t
    if (!have_lo_) refresh_lower_bound(G, 30, theta_top);
    else if (steps_since_lo_ >= period) refresh_lower_bound(G, 6, theta_top);

    // A block that has just been (re)started - the first step of a run, a teacher-forced state - has no accuracy of
    // earlier steps to lean on: in the steady state a step inherits Ritz vectors that already met the tolerance and the
    // measured error of A+ sits 5-10x below it (1.5e-5 / 3.6e-5 at eig_tol = 2e-4 on the two full-width workloads),
    // while a cold block stops right AT the tolerance (1-2e-4).  The cold update therefore converges to a quarter of
    // it; it happens once per run.
    const float tol = cold_ ? 0.25f * tol_ : tol_;
    cold_ = false;
    bool conv = false, cheap_done = false;
    bool w_emitted = false;          // the last pass's residual kernel has written W (all rows of that pass's block, scales too)
    int nw = 0, it = 0;
    int n_ritz = have_theta_ ? m_ : 0;        // leading rows that are Ritz vectors with a known theta
    std::vector<int> panel_ends;
    for (; it < max_iter_; it++) {
        float *Z = Q_;
        panel_ends.clear();
        int lock_rows = 0;
        // The filter is only applied to a block of Ritz vectors (theta known): the amplification of
        // each row is then predictable and the panels of the orthonormalisation can follow it.
        if (m_ < live_ && n_ritz > 0) {
            // ---- Chebyshev filter of degree d damping [a, b] of H = -G ---------------------------
            const float a = lo_bound_;
            float b = std::min(block_min, mu);
            const float minw = 1e-3f * (std::fabs(a) + std::fabs(mu)) + 1e-20f;
            if (b < a + minw) b = a + minw;
            const float c0 = 0.5f * (a + b), e0 = 0.5f * (b - a);
            int d = std::min(12, deg0_ + 2 * it);
            const float xmax = std::max(1.5f, (std::max(theta_top, mu) - c0) / e0);
            // keep the top/guard amplification ratio T_d(xmax) below ~1e5: the guard rows survive the
            // cancellation against the amplified rows with ~1e5 * 6e-8 relative noise
            const int dcap = (int)(12.2f / std::acosh(xmax));
            // ---- locking (passes after the first): while theta_top / mu is in the hundreds (the start-up transient) the
            // cap above holds the filter at degree 2-4, and the pairs just above mu - hundreds of them, a dense spectrum -
            // gain a factor ~1.6 per pass: ten passes per step.  The pairs at the top converge in the first pass or two
            // (their amplification is the largest).  Once they have - residuals of the CURRENT matrix, a quarter of the
            // tolerance - they are locked: not filtered, and projected out of every term of the recurrence that runs on
            // the rows below them, so that the cap is set by the largest UNLOCKED Ritz value and the degree can rise.
            // (The locked rows stay in the Rayleigh-Ritz step and in the convergence test.)
            int n_lock = 0, dcap_a = dcap;
            if (lock_ && it >= 1 && n_ritz == m_ && nw > 0 && m_ > 32 && dcap < d + 4) {
                const float emax_l = cscale * (h_theta_[0] - mu);
                int np = 0;
                while (np < nw && np < m_ - 16 && cscale * h_res_[np] <= 0.25f * tol * emax_l) np++;
                if (np >= 8) {
                    const float xa = std::max(1.5f, (std::max(h_theta_[np], mu) - c0) / e0);
                    const int cap_a = (int)(12.2f / std::acosh(xa));
                    if (cap_a >= dcap + 2) { n_lock = np; dcap_a = cap_a; }
                }
            }
            if (n_lock > 0) d = std::min(16, d + 4);
            d = std::max(2, std::min(d, dcap_a));
            // A pass that missed the tolerance only narrowly is followed by a degree-1 pass that costs
            // no product at all: Y = Q H is still there from the Rayleigh-Ritz step, and
            // (H - c0) Q / e0 = (Y - c0 Q) / e0 already damps everything below the block by the
            // factor the marginal case needs.
            // (It gains a factor 0.5-0.9 on the criterion, measured, and what it must reach is the tolerance of the passes
            // after the first: only a miss within 1.6x of THAT is worth a pass without products.)
            const bool cheap = it >= 1 && !cheap_done && y_ok_ && cheap_pass_ && last_crit_ <= cheap_margin_ * tol_pass2_ * tol;
            if (cheap) {
                cheap_done = true;                                   // once per step: if it is not enough, filter properly
                d = 1;
                n_lock = 0;
                axpby_inplace(Y_, Q_, 1.0f / e0, -c0 / e0, (size_t)m_ * F_, s_);
                Z = Y_;
                y_ok_ = false;
                st_.cheap_passes++;
            } else if (n_lock > 0) {
                // the recurrence on the rows below the locked ones, every term cleaned of the locked span (its top
                // components would otherwise grow by ~2 x_top per product against ~2 x_j for the rows' own); no carried
                // planes here: the projection changes a term after its product's reduction has split it
                const int na = m_ - n_lock;
                const size_t off = (size_t)n_lock * F_;
                y_ok_ = false;
                const float *prev = Q_;
                float *cur = pick({Q_, Y_});
                chain_planes_of_ = nullptr;
                chain_next_ = false;
                product(Q_ + off, na, G, -1.0f / e0, cur + off, Q_ + off, -c0 / e0, nullptr, 0.f, true);
                project_out(cur + off, na, Q_, n_lock);
                for (int k = 2; k <= d; k++) {
                    float *nxt = pick({Q_, prev, cur});
                    product(cur + off, na, G, -2.0f / e0, nxt + off, cur + off, -2.0f * c0 / e0, prev + off, -1.0f, true);
                    project_out(nxt + off, na, Q_, n_lock);
                    prev = cur; cur = nxt;
                }
                DLCO_HIP(hipMemcpyAsync(cur, Q_, off * sizeof(float), hipMemcpyDeviceToDevice, s_));
                Z = cur;
                st_.locked_passes++;
                st_.locked_rows += n_lock;
            } else {
                const float *prev = Q_;
                float *cur = pick({Q_, Y_});
                chain_planes_of_ = nullptr;
                chain_next_ = d >= 2;                            // the result is the X of the next product: emit its planes
                bool first_done = false;
                if (it == 0 && ru_ok && n_ritz == m_) {
                    // (H - c0) Q / e0 = (beta / e0) Y - (alpha / e0) C_w X_a - (c0 / e0) Q, H = -G
                    coeff_.alloc(rank_coeff_bytes(cap_ < 160 ? cap_ : 160, ru_.kmax));
                    if (prof_) prof_->begin(PROF_RANK_UPDATE);
                    first_done = rank_first_term(Y_, Q_, F_, m_, F_, ru_.beta / e0, -c0 / e0, -ru_.alpha / e0, cur, ru_.proj, ru_.ldp,
                                                 ru_nw, wscale_.p, ru_.slot, ru_.w, ru_.k_dev, ru_.kmax, ru_.planes, coeff_.p,
                                                 chain_next_ ? plane_hi_.p : nullptr, plane_lo_.p, s_, ru_.coeff_ready, ru_.planes_mode);
                    if (prof_) prof_->end(PROF_RANK_UPDATE);
                    if (first_done) {
                        st_.rank_update_passes++;
                        chain_planes_of_ = chain_next_ ? cur : nullptr;
                        chain_rows_ = m_;
                        if (ru_check_) {
                            float *ref = pick({Q_, Y_, cur});
                            const float *keep = chain_planes_of_;
                            const bool keep_next = chain_next_;
                            chain_next_ = false; chain_planes_of_ = nullptr;
                            std::vector<char> ph(bf16x2_plane_bytes(m_, F_)), pl(ph.size());   // the check must not disturb the planes
                            DLCO_HIP(hipMemcpyAsync(ph.data(), plane_hi_.p, ph.size(), hipMemcpyDeviceToHost, s_));
                            DLCO_HIP(hipMemcpyAsync(pl.data(), plane_lo_.p, pl.size(), hipMemcpyDeviceToHost, s_));
                            product(Q_, m_, G, -1.0f / e0, ref, Q_, -c0 / e0, nullptr, 0.f, true);
                            std::vector<float> a((size_t)m_ * F_), b(a.size());
                            DLCO_HIP(hipMemcpyAsync(a.data(), cur, a.size() * sizeof(float), hipMemcpyDeviceToHost, s_));
                            DLCO_HIP(hipMemcpyAsync(b.data(), ref, b.size() * sizeof(float), hipMemcpyDeviceToHost, s_));
                            DLCO_HIP(hipMemcpyAsync(plane_hi_.p, ph.data(), ph.size(), hipMemcpyHostToDevice, s_));
                            DLCO_HIP(hipMemcpyAsync(plane_lo_.p, pl.data(), pl.size(), hipMemcpyHostToDevice, s_));
                            DLCO_HIP(hipStreamSynchronize(s_));
                            double dmax = 0.0, rmax = 0.0;
                            for (size_t e = 0; e < a.size(); e++) { dmax = std::max(dmax, (double)std::fabs(a[e] - b[e])); rmax = std::max(rmax, (double)std::fabs(b[e])); }
                            st_.rank_update_check = std::max(st_.rank_update_check, rmax > 0.0 ? dmax / rmax : 0.0);
                            st_.product_rows -= m_;
                            chain_next_ = keep_next; chain_planes_of_ = keep;
                        }
                    }
                }
                if (!first_done) product(Q_, m_, G, -1.0f / e0, cur, Q_, -c0 / e0, nullptr, 0.f, true);
                for (int k = 2; k <= d; k++) {
                    float *nxt = pick({prev, cur});
                    chain_next_ = k < d;
                    product(cur, m_, G, -2.0f / e0, nxt, cur, -2.0f * c0 / e0, prev, -1.0f, true);
                    prev = cur; cur = nxt;
                }
                chain_next_ = false;
                chain_planes_of_ = nullptr;
                Z = cur;
            }
            last_deg_ = d;
            // panels: at most 128 rows, predicted amplification T_d(x_j) within panel_amp_ inside a
            // panel.  A Ritz row carries the directions above it only at the level of its own
            // residual, so after the filter it is at worst (residual level x amplification ratio)
            // parallel to them: CholQR2 resolves that as long as the product stays below ~1e2.
            int start = n_lock;
            double amp0 = 0.0;
            if (n_lock > 0) panel_ends.push_back(n_lock);
            lock_rows = n_lock;
            for (int j = n_lock; j < n_ritz; j++) {
                const double x = std::max(1.0, (double)(h_theta_[j] - c0) / e0);
                const double amp = std::cosh((double)d * std::acosh(x));
                if (j == start) amp0 = amp;
                else if (j - start >= CHOL_INV_MAX_N || amp0 > panel_amp_ * amp) { panel_ends.push_back(j); start = j; amp0 = amp; }
            }
            if (n_ritz < m_) panel_ends.push_back(n_ritz);
        }
        for (int j = n_ritz + 64; j < m_; j += 64) panel_ends.push_back(j);   // rows without a Ritz value: panels of 64
        panel_ends.push_back(m_);
        // ---- orthonormalise, Rayleigh-Ritz -------------------------------------------------------
        float *Qo = Z;
        orthonormalize(Z, m_, pick({Z}), panel_ends, lock_rows);     // in place
        float *Yb = pick({Qo});
        product(Qo, m_, G, -1.0f, Yb, nullptr, 0.f, nullptr, 0.f, false);     // Yb = Qo * H, exact fp32
        gram(Yb, Qo, m_, Tm_.p);
        if (prof_) prof_->begin(PROF_JACOBI);
        // Guard directions only have to span: pairs of two columns below the Ritz value of the guard at a
        // quarter of the guard depth (known from the last pass; the leading guards above it still count,
        // they decide guards_ok) are rotated but do not prolong the sweeps.  What this may leave unfinished
        // shows up in the residuals below, which decide convergence.
        float lam_cut = -3.0e38f;
        if (guard_stop_ && have_theta_ && n_ritz == m_) {
            int nwp = 0;
            while (nwp < m_ && h_theta_[nwp] > mu) nwp++;
            const int ig = nwp + std::max(2, guard_ / 4);
            if (ig < m_) lam_cut = h_theta_[ig];
        }
        float *Qn = pick({Qo, Yb});
        float *Yn = pick({Qo, Yb, Qn});
        w_emitted = false;
        const size_t blk = (size_t)2 * (cap_ + 8) + 1;               // evals | res | sweeps in one copy
        // (second attempt: only after the multi-workgroup Jacobi gave up at its grid barrier - T, Qo and Yb are untouched
        // by a failed attempt, the m x m problem is then solved again by the single-workgroup kernel)
        for (int attempt = 0;; attempt++) {
        jacobi_eigh(Tm_.p, cap_, m_, evals_.p, Vm_.p, cap_, jwork_.p, sweeps_dev_, s_, lam_cut, attempt > 0);
        if (prof_) prof_->end(PROF_JACOBI);
        rotate(Vm_.p, cap_, m_, m_, Qo, Qn, Yb, Yn);
        Q_ = Qn; Y_ = Yn;
        y_ok_ = true;                                                // Y_ = Q_ H for the current H
        if (!poll_readback_) residual_norms(Q_, Y_, F_, evals_.p, m_, F_, res_.p, s_);
        if (poll_readback_) {
            // the block is written into pinned memory by a kernel that raises a sequence number last; polling it
            // costs a few microseconds where copy + hipStreamSynchronize cost tens (one read-back per pass)
            unsigned *flag = reinterpret_cast<unsigned *>(pin_ + pin_floats_ - 16);
            const unsigned seq = ++publish_seq_;
            int *extra_host = reinterpret_cast<int *>(pin_ + pin_floats_ - 32);
            // (the residual kernel's last workgroup publishes: one launch)
            // (the kernel also writes this pass's rows of W, guard rows and scales included: if the pass is the update's last,
            // W is already where it belongs)
            ResidualEmit em;
            em.W = W; em.ldw = F_; em.mu = mu; em.cscale = cscale; em.wscale = wscale_.p; em.guards = emit_guards_;
            residual_norms_publish(Q_, Y_, F_, evals_.p, m_, F_, res_.p, reinterpret_cast<unsigned *>(sweeps_dev_ + 4), ritz_block_.p, pin_,
                                   (int)blk, flag, seq, s_, extra_dev_, extra_host, &em, &w_emitted);
            const auto t0 = std::chrono::steady_clock::now();
            bool seen = false;
            for (long spins = 0;; spins++) {
                if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) == seq) { seen = true; break; }
                if ((spins & 1023) == 1023 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(200)) break;
            }
            if (!seen) {                                             // slow box or a faulted kernel: let the runtime tell
                DLCO_HIP(hipStreamSynchronize(s_));
                DLCO_CHECK(__atomic_load_n(flag, __ATOMIC_ACQUIRE) == seq, -3, "eig tracker: read-back did not arrive");
            }
        } else {
            DLCO_HIP(hipMemcpyAsync(pin_, ritz_block_.p, blk * sizeof(float), hipMemcpyDeviceToHost, s_));
            if (extra_dev_) DLCO_HIP(hipMemcpyAsync(pin_ + pin_floats_ - 32, extra_dev_, sizeof(int), hipMemcpyDeviceToHost, s_));
            DLCO_HIP(hipStreamSynchronize(s_));
        }
        if (extra_dev_) extra_val_ = *reinterpret_cast<int *>(pin_ + pin_floats_ - 32);
        const int sweeps = *reinterpret_cast<int *>(pin_ + 2 * (cap_ + 8));
        if (sweeps >= 0) { st_.jacobi_sweeps += sweeps; break; }
        // the kernel wrote neither Ritz values nor V: nothing read back in this attempt may be used
        st_.jacobi_barrier_timeouts++;
        DLCO_CHECK(attempt == 0, -3, "eig tracker: the Jacobi kernel reported a failure twice (grid barrier of jacobi_mw_kernel, then the single-workgroup kernel)");
        if (prof_) prof_->begin(PROF_JACOBI);
        }
        std::memcpy(h_theta_.data(), pin_, (size_t)m_ * sizeof(float));
        std::memcpy(h_res_.data(), pin_ + (cap_ + 8), (size_t)m_ * sizeof(float));
        have_theta_ = true;
        {
            const int before = m_;
            theta_dev_valid_ = (drop_dead_rows() == before);    // a compaction leaves evals_ on the device stale
        }
        theta_top = h_theta_[0];
        block_min = std::min(h_theta_[m_ - 1], mu);

        n_ritz = m_;
        nw = 0;
        while (nw < m_ && h_theta_[nw] > mu) nw++;
        // What has to be small is the error of A = c * (G - mu)_+, not every Ritz pair by itself.  The
        // Ritz pairs of the block are exact eigenpairs of G + E with E = -sum_i (r_i q_i^T + q_i r_i^T),
        // r_i orthogonal to the block, and the derivative of the matrix function (x - mu)_+ damps the
        // component of E that couples a pair at theta_i > mu with a direction at lambda < mu by the
        // divided difference (theta_i - mu) / (theta_i - lambda) <= (theta_i - mu) / (theta_i - b), b
        // the top of the spectrum outside the converged part of the block (taken at the middle guard):
        // a pair just above the threshold carries almost no weight in A and its residual counts with
        // that weight.  Pairs well above mu keep the weight 1 of the plain residual test.
        float crit = 0.f;
        const float bsep = (weighted_crit_ && nw < m_) ? std::min(mu, h_theta_[nw + (m_ - nw) / 2]) : mu;
        for (int i = 0; i < nw; i++) {
            float wgt = 1.f;
            const float den = h_theta_[i] - bsep;
            if (weighted_crit_ && den > 0.f) wgt = std::min(1.f, (h_theta_[i] - mu + h_res_[i]) / den);
            crit = std::max(crit, cscale * h_res_[i] * wgt);
        }
        const float emax = nw > 0 ? cscale * (h_theta_[0] - mu) : 0.f;
        last_crit_ = emax > 0.f ? crit / emax : 0.f;
        // the leading guards must be resolved too, or an eigenvalue just above mu can hide in them: each
        // needs theta + |r| < mu (no eigenvalue of that pair's interval reaches mu), or an interval that
        // reaches above mu by less than the tolerance (what could hide there weighs less than that in A),
        // or a small residual
        bool guards_ok = true;
        const int ng = std::min(m_, nw + std::max(2, guard_ / 4));
        const float gtol = tol * std::max(emax, cscale * 1e-3f * std::fabs(mu));
        for (int i = nw; i < ng; i++) {
            const float excess = h_theta_[i] + h_res_[i] - mu;
            const bool below = excess < 0.f || (weighted_crit_ && cscale * excess <= gtol);
            const bool small = cscale * h_res_[i] <= gtol;
            guards_ok = guards_ok && (below || small);
        }
        if (m_ >= live_) conv = true;                                // dense: Rayleigh-Ritz is exact
        else if (nw == 0) conv = guards_ok || it >= 6;
        // A step that converges in its first pass inherits Ritz vectors that already met the tolerance, and the measured
        // error of A+ then sits 3-10x below it; one that needed more passes (start-up transient, a growing block) has
        // nothing to lean on and stops right at the tolerance (free runs against ssyevr step by step: up to 1.03e-4 of
        // the 1e-4 gate at eig_tol = 2e-4).  Passes after the first therefore converge to half of it.
        // (And a first pass that stops right AT the tolerance was measured at 0.46 of it in A+ - 9.2e-5 against the 1e-4 gate, in a
        // small free run; at full width the ratio is 0.25-0.29, tools/full_width_crit_vs_error.py - so the first pass is held to
        // 0.85 of it: worst free-run step 7.4e-5, and the rank ~128 workload pays 1.12 instead of 1.11 passes per step.)
        else conv = crit <= (it == 0 ? tol_pass1_ * tol : tol_pass2_ * tol) * emax && guards_ok;
        if (debug_)
            std::fprintf(stderr, "[eig] upd %ld it %d deg %d lock %d m %d nw %d theta[%.5g .. %.5g] mu %.5g lo %.5g crit/emax %.3g guards_ok %d conv %d\n",
                         (long)st_.updates, it, last_deg_, lock_rows, m_, nw, h_theta_[0], h_theta_[m_ - 1], mu, lo_bound_, last_crit_,
                         (int)guards_ok, (int)conv);
        if (h_theta_[m_ - 1] < lo_bound_) lo_bound_ = h_theta_[m_ - 1] - 0.5f * std::fabs(h_theta_[m_ - 1]) - 1e-12f;
        // ---- grow the block when the positive eigenspace reaches into the guard ------------------
        if (m_ < live_ && m_ < cap_ && nw > m_ - std::max(2, guard_ / 2)) {
            const int add = std::min(cap_ - m_, std::max(guard_, nw / 4));
            // half of the new rows from the caller's source (rows of the step's batch), the rest random: the directions
            // that have just risen above mu lie mostly in the span of recent batch rows, and a block that starts there
            // needs fewer passes than one that starts from noise; the random half keeps the block able to find the rest
            int got = 0;
            if (grow_cb_) got = std::max(0, std::min(add / 2, grow_cb_(Q_ + (size_t)m_ * F_, add / 2)));
            append_random(Q_, m_ + got, add - got);
            for (int i = m_; i < m_ + add; i++) { h_theta_[i] = block_min; h_res_[i] = 0.f; }
            m_ += add;
            conv = false;
            continue;
        }
        // a full block that cannot grow while the positive eigenspace still reaches into its guards
        // cannot certify the rank: the reference keeps EVERY positive eigen-direction
        // (src/pj-learn.cpp:480-484), so this is reported, never silently truncated
        if (m_ < live_ && m_ >= cap_ && nw > m_ - std::max(2, guard_ / 2)) { conv = false; it++; break; }
        if (conv) { it++; break; }
    }
    st_.iters += it;
    // starting degree for the next step: the cheapest one that has been converging in a single
    // filter + Rayleigh-Ritz pass (each extra pass costs a Jacobi and an orthonormalisation)
    if (it >= 2) deg0_ = std::min(10, deg0_ + 2);
    else if (conv && last_crit_ < 0.15f * tol_) deg0_ = std::max(2, deg0_ - 1);
    if (!conv) st_.nonconverged++;
    if (converged) *converged = conv;

    // ---- trim the block, emit W (ascending eigenvalue order, like LAPACK's) ----------------------
    {
        // the products pad the block to a multiple of 32 rows: prefer a block that fills its
        // padding exactly, as long as at least half of the guard vectors stay
        int keep = nw + guard_;
        const int r32 = (keep / 32) * 32;
        if (r32 >= nw + std::max(4, guard_ / 2)) keep = r32;
        else if (keep <= 96) {
            // the m x m solvers work on PAIRS of eight-column blocks (kernels_jacobi.hip): 81-96 rows cost eleven block rounds
            // per sweep, 65-80 rows nine.  A small block - where that chain is half the step - that can drop to a multiple of
            // 16 rows and keep three quarters of its guards does (the reference's own shape: rank 55 + 32 guards = 87 rows
            // -> 80; Jacobi 0.117 -> 0.100 ms, 1.005 -> 1.015 passes per step, +3.5 % on ref544).  Not above 96 rows: at rank
            // ~115 the same rule (152 -> 144 rows, 24 guards) cost more passes than it saved rounds (1.16 -> 1.24 per step).
            const int r16 = (keep / 16) * 16;
            if (r16 >= nw + std::max(4, (3 * guard_) / 4)) keep = r16;
        }
        m_ = std::max(1, std::min(m_, keep));
    }
    double tr = 0.0;
    if (nw > 0) {
        for (int i = 0; i < nw; i++) tr += (double)(cscale * (h_theta_[i] - mu));
        if (theta_dev_valid_) {
            // the Ritz values are still on the device in the order of the rows of Q: scale and reverse there
            const bool ext = emit_guards_ && y_ok_ && m_ > nw;
            if (!w_emitted) emit_w_rows(W, F_, Q_, F_, evals_.p, nw, mu, cscale, F_, s_, ext ? m_ : 0, ext ? wscale_.p : nullptr);
            if (ext) { wext_rows_ = m_; wext_nw_ = nw; }
        } else {
            h_sc_.resize(nw);
            h_sr_.resize(nw);
            for (int j = 0; j < nw; j++) {
                const int i = nw - 1 - j;
                h_sc_[j] = std::sqrt(cscale * (h_theta_[i] - mu));
                h_sr_[j] = i;
            }
            DLCO_HIP(hipMemcpyAsync(scale_.p, h_sc_.data(), nw * sizeof(float), hipMemcpyHostToDevice, s_));
            DLCO_HIP(hipMemcpyAsync(srcrow_.p, h_sr_.data(), nw * sizeof(int32_t), hipMemcpyHostToDevice, s_));
            DLCO_HIP(hipStreamSynchronize(s_));
            scale_rows(W, F_, Q_, F_, scale_.p, srcrow_.p, nw, F_, s_);
        }
    }
    if (trace) *trace = tr;
    return nw;
}

}  // namespace dlco
