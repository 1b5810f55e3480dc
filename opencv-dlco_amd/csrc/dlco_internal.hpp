// dlco_internal.hpp — shared declarations for libdlco.so (gfx950 only).
// Host side C++17, device side HIP.  No torch types, no OpenCV, no CUDA shims.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <mutex>
#include <utility>
#include <stdexcept>
#include <string>
#include <vector>

namespace dlco {

// ---------------------------------------------------------------------------
// error handling: exceptions inside the library, converted to codes at the ABI
// ---------------------------------------------------------------------------
struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string &m) : std::runtime_error(m), code(c) {}
};

#define DLCO_HIP(expr)                                                             \
    do {                                                                           \
        hipError_t e__ = (expr);                                                   \
        if (e__ != hipSuccess)                                                     \
            throw ::dlco::Error(-3, std::string(#expr) + ": " + hipGetErrorString(e__)); \
    } while (0)

#define DLCO_CHECK(cond, code, msg)                                                \
    do {                                                                           \
        if (!(cond)) throw ::dlco::Error((code), std::string(msg));                \
    } while (0)

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per (kernel, device): the attribute is per device, and contexts may
// be created on several devices and from several threads (cfg.device)
inline void ensure_dynamic_lds(const void *func, int bytes)
{
    static std::mutex mu;
    static std::vector<std::pair<const void *, int>> done;
    int dev = 0;
    DLCO_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(mu);
    for (const auto &d : done)
        if (d.first == func && d.second == dev) return;
    DLCO_HIP(hipFuncSetAttribute(func, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    done.emplace_back(func, dev);
}

// simple owning device buffer
template <typename T>
struct DevBuf {
    T *p = nullptr;
    size_t n = 0;
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    ~DevBuf() { release(); }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        n = 0;
    }
    void alloc(size_t count) {
        if (count <= n && p) return;
        release();
        if (count == 0) count = 1;
        DLCO_HIP(hipMalloc((void **)&p, count * sizeof(T)));
        n = count;
    }
    void zero(hipStream_t s) { if (p) DLCO_HIP(hipMemsetAsync(p, 0, n * sizeof(T), s)); }
};

// ---------------------------------------------------------------------------
// HIP-event timers on the launch stream, one slot per named kernel group
// ---------------------------------------------------------------------------
enum ProfSlot { PROF_GRAD_SYRK = 0, PROF_EIG_PRODUCT = 1, PROF_JACOBI = 2, PROF_PROJECT = 3, PROF_RANK_UPDATE = 4, PROF_SLOTS = 5 };

struct Profiler {
    bool on = false;
    unsigned mask = ~0u;          // slots that record (bit per ProfSlot): every record costs a few microseconds of queue time
    hipStream_t s = nullptr;
    struct Rec { std::vector<hipEvent_t> ev; size_t used = 0; double ms = 0.0; int64_t n = 0; };
    Rec rec[PROF_SLOTS];
    ~Profiler() { for (auto &r : rec) for (hipEvent_t e : r.ev) (void)hipEventDestroy(e); }
    void mark(int slot) {
        if (!on || !((mask >> slot) & 1u)) return;
        Rec &r = rec[slot];
        // timing-only events: without the system-scope fence a default event carries (hipEventDisableSystemFence), a
        // record is a time stamp in the queue instead of a cache flush - the default form cost ~6 us of idle GPU per record
        if (r.used == r.ev.size()) { hipEvent_t e; DLCO_HIP(hipEventCreateWithFlags(&e, hipEventDisableSystemFence)); r.ev.push_back(e); }
        DLCO_HIP(hipEventRecord(r.ev[r.used++], s));
        if (r.used >= 8192 && (r.used % 2) == 0) drain(slot);
    }
    void begin(int slot) { mark(slot); }
    void end(int slot) { mark(slot); }
    void drain(int slot) {
        Rec &r = rec[slot];
        if (r.used == 0) return;
        DLCO_HIP(hipStreamSynchronize(s));
        for (size_t i = 0; i + 1 < r.used; i += 2) {
            float ms = 0.f;
            DLCO_HIP(hipEventElapsedTime(&ms, r.ev[i], r.ev[i + 1]));
            r.ms += ms; r.n++;
        }
        r.used = 0;
    }
    void drain_all() { for (int i = 0; i < PROF_SLOTS; i++) drain(i); }
    void reset() { drain_all(); for (auto &r : rec) { r.ms = 0.0; r.n = 0; } }
};

// ---------------------------------------------------------------------------
// generic fp32 MFMA GEMM (kernels_gemm.hip)
//   C[M,N] = alpha * A(M,K) * B(K,N)  (+ epilogue)
// Operand addressing:
//   A.kmajor : element (i,k) at A.p[row(k)*A.ld + i]   (memory rows run along K)
//   else     : element (i,k) at A.p[row(i)*A.ld + k]   (memory rows run along M)
//   B.kmajor : element (k,j) at B.p[row(k)*B.ld + j]
//   else     : element (k,j) at B.p[row(j)*B.ld + k]
//   row(x) = row_ids ? row_ids[x] : x ; k-major rows may carry a per-row scale.
// Epilogue:  C = alpha*acc + beta*C_in + b1*E1 + b2*E2   (E1/E2 share C's ld)
// ---------------------------------------------------------------------------
struct GemmOperand {
    const float *p = nullptr;
    long ld = 0;
    bool kmajor = false;
    const int32_t *row_ids = nullptr;
    const int32_t *row_ids2 = nullptr;  // optional: the operand row is p[row_ids[x]] - p[row_ids2[x]] (pair mode)
    const float *row_scale = nullptr;   // only for k-major operands
};

struct GemmArgs {
    int M = 0, N = 0, K = 0;
    GemmOperand A, B;
    float *C = nullptr;
    long ldc = 0;
    float alpha = 1.0f, beta = 0.0f;
    const float *E1 = nullptr;
    float b1 = 0.0f;
    const float *E2 = nullptr;
    float b2 = 0.0f;
    const int *k_dev = nullptr;         // optional device-resident K (<= K)
    int split_k = 1;                    // >1: partial slabs + ordered reduce
    float *slab = nullptr;              // workspace >= split_k*M*N floats when split_k > 1
    bool upper_only = false;            // M==N: compute tiles with j-block >= i-block, mirror the rest
    bool raw_slab = false;              // keep the [split][M][N] partials in `slab`, skip the reduce (C unused)
    int *split_out = nullptr;           // receives the number of K slices actually used
    const float *B2 = nullptr;          // twin product in the same launch: C2 = alpha * A * B2 (+ the same epilogue terms), B2 laid
    float *C2 = nullptr;                // out like B; needs split_k == 1
    bool small_m_tiles = false;         // 64-row tiles whatever M ...
    bool small_n_tiles = false;         // ... and 64-column tiles whatever N: the 96 x 400 batch projection fills 64 x 64 tiles to 67 %, 64 x 128 ones to 58 % (-6 us)
};

void gemm_f32(const GemmArgs &a, hipStream_t s);
// Same product on the bf16 matrix cores with split operands (x = hi + lo, three MFMAs per term,
// fp32 accumulation; ~1e-5 relative error): kernels_bf16x2.hip.  plane_hi/lo are workspaces of
// bf16x2_plane_bytes(M, K) bytes each.  Returns false for unsupported shapes.
size_t bf16x2_plane_bytes(int M, int K);
size_t bf16x2_slab_floats(int M, int N, int ksplit = 0);
// plane_lo2 != nullptr selects the three-way split (x = hi + mid + lo, six MFMAs per term): fp32-level accuracy.
// ksplit: number of K slices (grid y, default 4); K must be a multiple of 128 * ksplit.  A column slab of G
// (N < K) takes ksplit = 4 * K / N so that the launch still fills the chip.
bool skinny_product_bf16x2(const float *X, long ldx, int M, const float *G, long ldg, int N, int K, float alpha, float *C,
                           long ldc, const float *E1, float b1, const float *E2, float b2, void *plane_hi, void *plane_lo,
                           float *slab, hipStream_t s, int ksplit = 0, void *plane_lo2 = nullptr, bool g_tiled = false);
// P1 + P2 for many rows without the r x N projection (kernels_project.hip): dist[n] = |W x_n|^2, x_n = row ids[n] (or
// row0 + n) of D, minus row ids2[n] in pair mode.  Three-way split-bf16 MFMA (fp32-level accuracy) or, with bf16 = true,
// operands rounded to bf16 once (BASELINE configs[4]).  planes: workspaces of project_rows_plane_bytes(F) bytes each.
size_t project_rows_plane_bytes(int K);
bool project_rows_sqdist(const float *W, long ldw, int r, const float *D, long ldd, int F, const int32_t *ids, const int32_t *ids2,
                         long row0, int nrows, float *dist, void *plane_hi, void *plane_lo, void *plane_lo2, bool bf16, hipStream_t s);
// raw partial projections of a few rows, K split over ksplit slices: slab [ksplit][r][ldn] (r <= 96)
bool project_rows_slab(const float *W, long ldw, int r, const float *D, long ldd, int F, const int32_t *ids, const int32_t *ids2,
                       int nrows, int ksplit, float *slab, long ldn, void *plane_hi, void *plane_lo, void *plane_lo2, bool bf16,
                       hipStream_t s);
// The coefficient fragments of the rank-update first filter term (kernels_rankupd.hip), as a job that rides in the
// gradient's row-split launch (its extra column of blocks): frag = nullptr means no job.
struct RankCoeffJob {
    const float *proj = nullptr;     // [rows of W incl. guard rows][ldp] projections of the batch slots
    long ldp = 0;
    int nw = 0, m = 0, MT = 0;       // Ritz rows among them, rows in all, 32-row tiles
    const float *wscale = nullptr;   // scale row i of the block went into W with
    const int32_t *slot = nullptr;   // slot[k]: column of proj of active row k
    void *frag = nullptr;            // out: rank_coeff_bytes(m, kmax) bytes
};
// Fused gradient SYRK + dual average (kernels_syrk.hip): C = beta*C + alpha * sum_k w_k x_k x_k^T over the
// rows ids[0 .. *k_dev) of D, upper tiles computed and mirrored.  ids/w hold kmax entries (multiple of 32,
// zero padded beyond *k_dev).  Returns false when the shape is not supported (F % 128 != 0).
// ids2 != nullptr (pair mode): row k is D[ids[k]] - D[ids2[k]], formed on the fly.
// slab_cols > 0 (dual average sharded over GPUs): only the columns [slab_col0, slab_col0 + slab_cols)
// of C are computed and stored (all rows, no mirror); both must be multiples of 128.
// lower triangle := upper triangle: the fused SYRK relies on an exactly symmetric dual average (it keeps
// it so itself); a matrix that comes from outside (dlco_set_state, dlco_grad_rda) is symmetrised first
void syrk_mirror_upper(float *C, long ldc, int F, hipStream_t s);
// packed = true: C holds the packed upper tiles (syrk_packed_floats(F) floats, ldc ignored): tile (bi, bj), bi <= bj, is a
// contiguous row-major 128 x 128 block at tile index bi*nt - bi*(bi-1)/2 + (bj - bi); no mirrored store.
// planes_ws: >= syrk_planes_bytes(kmax, F) bytes of device memory for the split (or bf16-rounded) row planes of the launch;
// nullptr: a per-device workspace inside the library (single-stream callers only).
size_t syrk_planes_bytes(int kmax, int F);
bool syrk_rda_f32(const float *D, long ldd, const int32_t *ids, const int32_t *ids2, const float *w, const int *k_dev,
                  int kmax, int F, float alpha, float beta, float *C, long ldc, hipStream_t s, int slab_col0 = 0,
                  int slab_cols = 0, bool bf16 = false, bool packed = false, void *planes_ws = nullptr,
                  const RankCoeffJob *coeff_job = nullptr);
// which planes syrk_rda_f32(bf16) writes into planes_ws: rank_first_term() reads operand 1 of them
int syrk_planes_mode(bool bf16);     // 3: three-way split planes (default arithmetic), 1: bf16-once planes (cfg.grad_bf16), 0: no planes (DLCO_SYRK_FP32)
// The first Chebyshev term of a step's filter from the step's own rank update, without a pass over the matrix
// (kernels_rankupd.hip): out = ay*Y + aq*Q + ac * C_w X_a, C_w[i][k] = w[k] * proj[row(i)][slot[k]] / wscale[i] with
// row(i) = nw-1-i for i < nw (W's ascending order), i for the guard rows behind; X_a = operand 1 of `planes`; also emits
// the two-way planes of `out` (split_x_kernel's order) when plane_hi/lo are given.  coeff_ws: rank_coeff_bytes(m, kmax).
size_t rank_coeff_bytes(int m, int kmax);
bool rank_first_term(const float *Y, const float *Q, long ld, int m, int F, float ay, float aq, float ac, float *out,
                     const float *proj, long ldp, int nw, const float *wscale, const int32_t *slot, const float *w,
                     const int *k_dev, int kmax, const void *planes, void *coeff_ws, void *plane_hi, void *plane_lo, hipStream_t s,
                     bool coeff_ready = false, int planes_mode = 3);   // coeff_ws already holds the fragments (they rode in the gradient's row split)
size_t syrk_packed_floats(int F);
void syrk_pack_upper(const float *C, long ldc, int F, float *packed, hipStream_t s);      // upper tiles of a full matrix -> packed
void syrk_unpack_upper(const float *packed, int F, float *C, long ldc, hipStream_t s);    // packed -> full symmetric matrix
// out[M][F] = alpha * X[M][F] * G + b1*E1 + b2*E2 for a symmetric G given as its packed upper tiles, every tile fetched
// from HBM once (kernels_bf16x2.hip, skinny_sym_kernel).  F == 8192 only (64 x 64 tiles on 8 XCDs x 32 CUs); M <= 160
// (two-way split) / 96 (three-way, plane_lo2 != nullptr).  slab: >= 4*M*F floats.  Returns false for other shapes.
bool skinny_product_sym(const float *X, long ldx, int M, const float *Gpacked, int F, float alpha, float *C, long ldc,
                        const float *E1, float b1, const float *E2, float b2, void *plane_hi, void *plane_lo, float *slab,
                        hipStream_t s, void *plane_lo2 = nullptr, bool planes_ready = false, bool emit_planes = false);
// planes_ready: plane_hi / plane_lo already hold the two-way planes of X (skip the split); emit_planes: the reduction also
// writes the two-way planes of the RESULT into plane_hi / plane_lo (the next product of a Chebyshev chain takes it as X)
// C[M][N] = alpha * X[M][K] * G[K][N] + b1*E1 + b2*E2 for M <= 128 (single launch, K split over
// the waves of a workgroup, deterministic).  Returns false when the shape is not supported.
bool skinny_product_f32(const float *X, long ldx, int M, int x_rows_alloc, const float *G, long ldg, int N, int K,
                        float alpha, float *C, long ldc, const float *E1, float b1, const float *E2, float b2, hipStream_t s);
size_t gemm_slab_floats(int M, int N, int split_k);
void splitk_reduce_f32(const float *slab, int split, int M, int N, float *C, long ldc, float alpha, float beta,
                       const float *E1, float b1, const float *E2, float b2, hipStream_t s);

// ---------------------------------------------------------------------------
// step kernels (kernels_step.hip)
// ---------------------------------------------------------------------------
// dist[j] = sum_q (sum_z slab[z][q][j])^2 for a [split][r][n] projection slab
void sqdist_from_proj(const float *proj, int split, int r, int n, long ld, float *dist, hipStream_t s, int r_all = 0,
                      float *reduced = nullptr);   // r_all > r (slab form): the slabs hold r_all rows, all reduced into `reduced`, the first r summed
// rho/kappa + signed weights + compact active list for one batch
//   pd, nd: [B] distances; out rho[B], kappa[B]; weights[2B] (rho_i for positives, -kappa_j for negatives)
void viol_counts(const float *pd, const float *nd, int B, int32_t *rho, int32_t *kappa, hipStream_t s);
// viol_counts + build_active_rows in one launch
void viol_counts_active_rows(const float *pd, const float *nd, int B, int32_t *rho, int32_t *kappa, const int32_t *pos_rows,
                             const int32_t *neg_rows, int slot_lo, int slot_hi, int32_t *ids, float *w, int *k_active, hipStream_t s,
                             int32_t *slots = nullptr);   // slots[k]: position of active row k in the rank's [pos | neg] slot list
// build the stacked weighted row list of the SYRK: ids[2B] = (pos rows, neg rows), w[2B] = (rho, -kappa);
// rows with zero weight are dropped; *k_active receives the count. [lo,hi) selects the slots owned by a rank.
void build_active_rows(const int32_t *pos_rows, const int32_t *neg_rows, const int32_t *rho, const int32_t *kappa,
                       int B, int slot_lo, int slot_hi, int32_t *ids, float *w, int *k_active, hipStream_t s);
// per-row sequential hinge sum (src/kernelop-opencv.cu:49-66), rows then summed in double
void hinge_rows(const float *pos, int n_pos, const float *neg, int n_neg, float *row_sums, hipStream_t s);
void sum_f32_to_f64(const float *x, int n, double *out, hipStream_t s);
void trace_f64(const float *A, int F, long ld, double *out, hipStream_t s);
void axpby_inplace(float *y, const float *x, float a, float b, size_t n, hipStream_t s);  // y = a*y + b*x
void scale_rows(float *dst, long ldd, const float *src, long lds, const float *scale, const int32_t *src_rows,
                int rows, int cols, hipStream_t s, const int32_t *src_rows2 = nullptr);    // dst[i] = scale[i]*(src[src_rows[i]] - src[src_rows2[i]])
// out_a[i] = pa[ids ? ids[i] : base + i], out_b likewise: row ids -> (patch, patch) ids of the pair table
// W[j][:] = sqrt(cscale * (theta[nw-1-j] - mu)) * Q[nw-1-j][:] for j < nw: the kept Ritz pairs in ascending
// eigenvalue order, scaled like the reference's W = sqrt(e) * v^T (src/pj-learn.cpp:480-487)
// m_ext > nw: the rows nw..m_ext-1 of Q (the guard rows) are copied unscaled behind the nw rows of W, and wscale[i]
// receives the factor row i of Q went out with (1 for guards) - what the next step's rank update divides by
void emit_w_rows(float *W, long ldw, const float *Q, long ldq, const float *theta, int nw, float mu, float cscale, int F,
                 hipStream_t s, int m_ext = 0, float *wscale = nullptr);
void translate_ids(const int32_t *ids, int base, int n, const int32_t *pa, const int32_t *pb, int32_t *out_a, int32_t *out_b,
                   hipStream_t s);
void fill_f32(float *p, float v, size_t n, hipStream_t s);

// Column-sharded dual average (one slab of F/world columns per rank): what the tracker and the
// step need to know about the decomposition and how they reach the other ranks.  `allgather`
// performs an in-place all-gather of `gather` viewed as [world][bytes_per_rank] (the rank's own
// chunk is filled by the caller), stream-ordered after the work already queued on the library
// stream; it throws on failure.
struct ShardComm {
    int world = 1, rank = 0;
    int c0 = 0, cw = 0;                  // this rank's columns [c0, c0 + cw)
    float *gather = nullptr;             // exchange buffer, >= gather_floats
    size_t gather_floats = 0;
    std::function<void(size_t bytes_per_rank)> allgather;
};
void pack_cols(float *dst, const float *src, long ld, int c0, int cw, int rows, hipStream_t s);
void unpack_cols(float *dst, long ld, const float *src, int cw, int rows, int world, hipStream_t s);

// synthetic stand-in for a *-unproj.h5 generated in HBM (bench): d = U^T z + eps, clipped
// (rows of F values at row stride ld >= F; columns beyond F are left alone)
void synth_rows(float *D, int N, int F, long ld, const float *U, int k, uint64_t seed, float sig_pos, float sig_neg,
                float noise, float jitter, hipStream_t s);

// ---------------------------------------------------------------------------
// small dense eigen problems (kernels_jacobi.hip, kernels_eig.hip)
// ---------------------------------------------------------------------------
// One-sided Jacobi on a symmetric n x n matrix T (ld = ldt; symmetric up to rounding: for n <= 160 the rows of T are
// taken as the columns of the work image without averaging).  Output: evals[n] descending, V[n][ldv] with ROW j =
// eigenvector j.  work: >= jacobi_work_floats(n) floats.  n <= 4096.
// lam_cut: a pair of columns whose eigenvalue estimates are BOTH below it is still rotated but does not
// keep the sweeps going (the tracker's guard vectors: only their span matters, and the caller checks the
// residuals of the pairs it keeps); the default counts every pair.
// *sweeps_out < 0: the multi-workgroup kernel (160 < n <= 2048) gave up at its bounded grid barrier (its workgroups were
// not all resident at once); evals / V are then NOT written and the caller repeats the call with single_workgroup = true.
void jacobi_eigh(const float *T, long ldt, int n, float *evals, float *V, long ldv, float *work, int *sweeps_out,
                 hipStream_t s, float lam_cut = -3.0e38f, bool single_workgroup = false);
size_t jacobi_work_floats(int n);
// CholQR building block for panels of <= 160 rows: factors the panel's Gram matrix M = L L^T
// (only its lower triangle is read) and returns Linv = L^-1 [n][ldl] (lower triangular, upper
// part zero), so that the orthonormal rows are the plain product Linv * Z.  A row whose pivot
// falls below rel_thresh * M_jj lies in the span of the rows before it: dead[j] = 1 and row j of
// Linv is zero.
constexpr int CHOL_INV_MAX_N = 160;
void chol_inverse128(const float *M, long ldm, int n, float rel_thresh, float *Linv, long ldl, int *dead, hipStream_t s);
// row norms of (Y - theta_i X) and of X
// n floats -> pinned host memory, then *flag_host = seq with system-scope release (the host polls flag_host)
void publish_block(const float *src, float *dst_host, int n, unsigned *flag_host, unsigned seq, hipStream_t s,
                   const int *extra_dev = nullptr, int *extra_host = nullptr);   // extra: one device int copied along
// What the residual kernel may emit on the way (see residual_publish_kernel): the rows of W of the pass it closes
struct ResidualEmit {
    float *W = nullptr;
    long ldw = 0;
    float mu = 0.f, cscale = 0.f;
    float *wscale = nullptr;         // scale row i of the block went out with (1 for the rows behind the nw kept ones)
    bool guards = false;             // also write the rows with theta <= mu, unscaled and in place
};
void residual_norms_publish(const float *X, const float *Y, long ld, const float *theta, int m, int F, float *res, unsigned *ticket,
                            const float *src, float *dst_host, int n, unsigned *flag_host, unsigned seq, hipStream_t s,
                            const int *extra_dev = nullptr, int *extra_host = nullptr, const ResidualEmit *emit = nullptr,
                            bool *emitted = nullptr);
void residual_norms(const float *X, const float *Y, long ld, const float *theta, int m, int F, float *res,
                    hipStream_t s);
void row_normalize(float *X, long ld, int m, int F, hipStream_t s, float min_norm = 0.f);
// y = H x style GEMV on a symmetric matrix (memory-bound), used by the spectral-bound estimator
void symv(const float *H, long ld, int F, const float *x, float *y, hipStream_t s);

// ---------------------------------------------------------------------------
// ROC statistics on device (kernels_stats.hip) — src/misc.cpp:297-332
// ---------------------------------------------------------------------------
struct RocWork;
RocWork *roc_work_create(int n_max);
void roc_work_destroy(RocWork *w);
void roc_stats(RocWork *w, const float *dist_dev, const uint8_t *labels_dev, int n, float *fpr95, double *auc,
               hipStream_t s);

}  // namespace dlco
