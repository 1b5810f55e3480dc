// kernels_step.hip — the small (non-GEMM) kernels of one pj-learn step and of the
// validation pass, for gfx950.  References are to the reference checkout
// (cbalint13/opencv-dlco).
#include "dlco_internal.hpp"

#include <algorithm>

namespace dlco {

namespace {

__device__ __forceinline__ double wave_sum_d(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// P2 (src/pj-learn.cpp:346-347): dist_j = sum_q proj[q][j]^2, rows added one after another in fp32
__global__ void sqdist_kernel(const float *proj, int split, int r, int n, long ld, float *dist)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const long plane = (long)r * ld;
    float d = 0.f;
    int q = 0;
    if (split == 1) {
        // eight independent loads in flight per trip; the squares are still added row after row
        for (; q + 8 <= r; q += 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) v[u] = proj[(long)(q + u) * ld + j];
#pragma unroll
            for (int u = 0; u < 8; u++) d += v[u] * v[u];
        }
    }
    for (; q < r; q++) {
        float p = 0.f;
        for (int z = 0; z < split; z++) p += proj[z * plane + (long)q * ld + j];
        d += p * p;
    }
    dist[j] = d;
}

// The same for a projection that arrives as `split` K-slice slabs (the bf16 variant's batch projection: 32 slices): one
// thread per column would add split * r values one after the other (1 ms at 32 x 61).  A workgroup takes 64 columns; eight
// groups of 64 threads sum the slices (in slice order) of eight rows q at a time into LDS, then the first 64 threads add
// the squares row after row - the same two orders as above.
// r_all > r: the slabs hold r_all rows (the tracker's guard rows behind the r rows of W); all of them are reduced and written
// to `reduced` [r_all][ld] (what the rank update of kernels_rankupd.hip reads), the distances sum the first r.
__global__ __launch_bounds__(512) void sqdist_split_kernel(const float *proj, int split, int r, int n, long ld, float *dist, int r_all,
                                                           float *reduced)
{
    __shared__ float p[96][64];
    const int jl = threadIdx.x & 63, qg = threadIdx.x >> 6;
    const int j = blockIdx.x * 64 + jl;
    const long plane = (long)r_all * ld;
    float d = 0.f;
    for (int q0 = 0; q0 < r_all; q0 += 96) {
        const int qn = min(96, r_all - q0);
        for (int q = qg; q < qn; q += 8) {
            float acc = 0.f;
            if (j < n) {
                const float *src = proj + (long)(q0 + q) * ld + j;
                int z = 0;
                for (; z + 8 <= split; z += 8) {
                    float v[8];
#pragma unroll
                    for (int u = 0; u < 8; u++) v[u] = src[(z + u) * plane];
#pragma unroll
                    for (int u = 0; u < 8; u++) acc += v[u];
                }
                for (; z < split; z++) acc += src[z * plane];
            }
            p[q][jl] = acc;
            if (reduced && j < n) reduced[(long)(q0 + q) * ld + j] = acc;
        }
        __syncthreads();
        if (qg == 0)
            for (int q = 0; q < min(qn, r - q0); q++) d += p[q][jl] * p[q][jl];
        __syncthreads();
    }
    if (qg == 0 && j < n) dist[j] = d;
}

// V1 (src/pj-learn.cpp:373-376): strict (pd_i + 1.0f) > nd_j
__device__ __forceinline__ void viol_body(float *sh, const float *pd, const float *nd, int B, int32_t *rho, int32_t *kappa)
{
    float *spd = sh, *snd = sh + B;
    for (int i = threadIdx.x; i < B; i += blockDim.x) { spd[i] = pd[i] + 1.0f; snd[i] = nd[i]; }
    __syncthreads();
    for (int i = threadIdx.x; i < B; i += blockDim.x) {
        const float thr = spd[i];
        int c = 0;
        for (int j = 0; j < B; j++) c += (thr > snd[j]) ? 1 : 0;
        rho[i] = c;
        const float me = snd[i];
        int k = 0;
        for (int j = 0; j < B; j++) k += (spd[j] > me) ? 1 : 0;
        kappa[i] = k;
    }
}
__global__ void viol_kernel(const float *pd, const float *nd, int B, int32_t *rho, int32_t *kappa)
{
    extern __shared__ float sh[];
    viol_body(sh, pd, nd, B, rho, kappa);
}

// stacked, weighted, compacted row list of the gradient SYRK (slots [lo,hi) of each class)
__device__ __forceinline__ void active_rows_body(const int32_t *pos_rows, const int32_t *neg_rows, const int32_t *rho,
                                                 const int32_t *kappa, int B, int lo, int hi, int32_t *ids, float *w, int *k_active,
                                                 int32_t *slots = nullptr)
{
    // ordered stream compaction of the 2*(hi-lo) candidates (positives first) by one workgroup:
    // chunk-wise ballot + prefix over the waves keeps the output order equal to the slot order
    __shared__ int wave_cnt[16];
    __shared__ int base;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
    const int n_own = max(0, min(hi, B) - lo), total = 2 * n_own;
    if (tid == 0) base = 0;
    __syncthreads();
    for (int c0 = 0; c0 < total; c0 += blockDim.x) {
        const int e = c0 + tid;
        int id = 0;
        float wt = 0.f;
        bool keep = false;
        if (e < total) {
            if (e < n_own) { const int i = lo + e; keep = rho[i] != 0; id = pos_rows[i]; wt = (float)rho[i]; }
            else { const int j = lo + e - n_own; keep = kappa[j] != 0; id = neg_rows[j]; wt = -(float)kappa[j]; }
        }
        const unsigned long long m = __ballot(keep);
        if (lane == 0) wave_cnt[wave] = __popcll(m);
        __syncthreads();
        int off = base;
        for (int v = 0; v < wave; v++) off += wave_cnt[v];
        if (keep) {
            const int pos = off + __popcll(m & ((1ull << lane) - 1ull));
            ids[pos] = id;
            w[pos] = wt;
            if (slots) slots[pos] = e;
        }
        __syncthreads();
        if (tid == 0) { int s = 0; for (int v = 0; v < nw; v++) s += wave_cnt[v]; base += s; }
        __syncthreads();
    }
    const int k = base;
    if (tid == 0) *k_active = k;
    // zero padding up to the next multiple of 32 entries: the SYRK consumes whole K tiles
    for (int z = k + tid; z < ((2 * B + 31) & ~31); z += blockDim.x) { ids[z] = 0; w[z] = 0.f; }
}
__global__ void active_rows_kernel(const int32_t *pos_rows, const int32_t *neg_rows, const int32_t *rho,
                                   const int32_t *kappa, int B, int lo, int hi, int32_t *ids, float *w, int *k_active)
{
    active_rows_body(pos_rows, neg_rows, rho, kappa, B, lo, hi, ids, w, k_active);
}
// V1 and the row list in one launch (the step's own sequence: one workgroup does both anyway)
__global__ void viol_active_kernel(const float *pd, const float *nd, int B, int32_t *rho, int32_t *kappa, const int32_t *pos_rows,
                                   const int32_t *neg_rows, int lo, int hi, int32_t *ids, float *w, int *k_active, int32_t *slots)
{
    extern __shared__ float sh[];
    viol_body(sh, pd, nd, B, rho, kappa);
    __threadfence_block();
    __syncthreads();
    active_rows_body(pos_rows, neg_rows, rho, kappa, B, lo, hi, ids, w, k_active, slots);
}

// H1 (src/kernelop-opencv.cu:49-66): one thread per positive row, the inner sum runs over the
// negatives in index order in fp32 exactly as the reference kernel does:
//   rsum = src1[idx] + 1 - src2[i]; tsum += (rsum > 0) ? rsum : 0
__global__ __launch_bounds__(256) void hinge_rows_kernel(const float *pos, int n_pos, const float *neg, int n_neg,
                                                         float *row_sums)
{
    __shared__ float sn[2048];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const float base = (i < n_pos) ? pos[i] + 1.0f : 0.f;
    float tsum = 0.f;
    for (int j0 = 0; j0 < n_neg; j0 += 2048) {
        const int cnt = min(2048, n_neg - j0);
        __syncthreads();
        for (int t = threadIdx.x; t < cnt; t += blockDim.x) sn[t] = neg[j0 + t];
        __syncthreads();
        for (int t = 0; t < cnt; t++) {
            const float rsum = base - sn[t];
            tsum += (rsum > 0.f) ? rsum : 0.f;
        }
    }
    if (i < n_pos) row_sums[i] = tsum;
}

__global__ __launch_bounds__(1024) void sum_f64_kernel(const float *x, int n, double *out)
{
    __shared__ double part[16];
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) s += (double)x[i];
    s = wave_sum_d(s);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int w = 0; w < (int)(blockDim.x >> 6); w++) t += part[w];
        *out = t;
    }
}

__global__ __launch_bounds__(1024) void trace_kernel(const float *A, int F, long ld, double *out)
{
    __shared__ double part[16];
    double s = 0.0;
    for (int i = threadIdx.x; i < F; i += blockDim.x) s += (double)A[(long)i * ld + i];
    s = wave_sum_d(s);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int w = 0; w < (int)(blockDim.x >> 6); w++) t += part[w];
        *out = t;
    }
}

__global__ void axpby_kernel(float *y, const float *x, float a, float b, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        y[i] = a * y[i] + b * x[i];
}

__global__ void fill_kernel(float *p, float v, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}

__global__ void scale_rows_kernel(float *dst, long ldd, const float *src, long lds, const float *scale,
                                  const int32_t *src_rows, const int32_t *src_rows2, int rows, int cols)
{
    const int i = blockIdx.x;
    if (i >= rows) return;
    const long sr = src_rows ? (long)src_rows[i] : (long)i;
    const float sc = scale ? scale[i] : 1.0f;
    if (src_rows2) {      // pair mode: the row is a difference of two descriptor rows
        const long sr2 = (long)src_rows2[i];
        for (int c = threadIdx.x; c < cols; c += blockDim.x) dst[(long)i * ldd + c] = sc * (src[sr * lds + c] - src[sr2 * lds + c]);
    } else {
        for (int c = threadIdx.x; c < cols; c += blockDim.x) dst[(long)i * ldd + c] = sc * src[sr * lds + c];
    }
}

__global__ void emit_w_kernel(float *W, long ldw, const float *Q, long ldq, const float *theta, int nw, float mu,
                              float cscale, int F, int m_ext, float *wscale)
{
    const int j = blockIdx.x;
    if (j >= max(nw, m_ext)) return;
    const int i = j < nw ? nw - 1 - j : j;                              // guard rows keep their place behind the nw rows of W
    const float sc = j < nw ? sqrtf(cscale * (theta[i] - mu)) : 1.0f;
    if (wscale && threadIdx.x == 0) wscale[i] = sc;
    for (int c = threadIdx.x * 4; c < F; c += blockDim.x * 4) {
        float4 v = *reinterpret_cast<const float4 *>(Q + (long)i * ldq + c);
        v.x *= sc; v.y *= sc; v.z *= sc; v.w *= sc;
        *reinterpret_cast<float4 *>(W + (long)j * ldw + c) = v;
    }
}

__global__ void translate_ids_kernel(const int32_t *ids, int base, int n, const int32_t *pa, const int32_t *pb,
                                     int32_t *out_a, int32_t *out_b)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int r = ids ? ids[i] : base + i;
    out_a[i] = pa[r];
    out_b[i] = pb[r];
}

// column slabs of a [rows][ld] block <-> the [world][rows][cw] exchange layout of the all-gather
__global__ void pack_cols_kernel(float *dst, const float *src, long ld, int c0, int cw, int rows)
{
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;           // float4 index
    const int q = cw >> 2;
    if (e >= (long)rows * q) return;
    const int i = (int)(e / q), c = (int)(e % q) * 4;
    *reinterpret_cast<float4 *>(dst + (long)i * cw + c) = *reinterpret_cast<const float4 *>(src + (long)i * ld + c0 + c);
}

__global__ void unpack_cols_kernel(float *dst, long ld, const float *src, int cw, int rows, int world)
{
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int q = cw >> 2;
    if (e >= (long)world * rows * q) return;
    const int c = (int)(e % q) * 4;
    const long t = e / q;
    const int i = (int)(t % rows), g = (int)(t / rows);
    *reinterpret_cast<float4 *>(dst + (long)i * ld + (long)g * cw + c) = *reinterpret_cast<const float4 *>(src + ((long)g * rows + i) * cw + c);
}

// ---- synthetic data ---------------------------------------------------------------------------
__device__ __forceinline__ uint64_t splitmix(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ULL;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL;
    return x ^ (x >> 31);
}
__device__ __forceinline__ float gauss(uint64_t key)
{
    const uint64_t h = splitmix(key);
    const float u1 = ((float)(uint32_t)(h >> 40) + 1.0f) * (1.0f / 16777217.0f);   // (0,1)
    const float u2 = (float)(uint32_t)(h & 0xFFFFFF) * (1.0f / 16777216.0f);
    return sqrtf(-2.0f * __logf(u1)) * __cosf(6.2831853f * u2);
}

// row i: label 1 when i is even; d = U^T z + noise*eps, z ~ N(0, (sigma*s_i)^2 I_k), clipped to [-1, 1].
// s_i = exp(jitter * g_i) is a per-row log-normal scale: it makes the two classes overlap the way real
// match / non-match distances do (FPR@95 of a few per cent instead of perfectly separable rows).
__global__ __launch_bounds__(256) void synth_kernel(float *D, int N, int F, long ld, const float *U, int k, uint64_t seed,
                                                    float sig_pos, float sig_neg, float noise, float jitter)
{
    extern __shared__ float z[];
    const int i = blockIdx.x;
    if (i >= N) return;
    float sig = (i % 2 == 0) ? sig_pos : sig_neg;
    if (jitter != 0.f) sig *= __expf(jitter * gauss(seed * 0x100000001B3ULL + ((uint64_t)i << 20) + 0x7000000000000000ULL));
    for (int q = threadIdx.x; q < k; q += blockDim.x)
        z[q] = sig * gauss(seed * 0x100000001B3ULL + ((uint64_t)i << 20) + (uint64_t)q + 0x5000000000000000ULL);
    __syncthreads();
    for (int f = threadIdx.x; f < F; f += blockDim.x) {
        float v = 0.f;
        for (int q = 0; q < k; q++) v += z[q] * U[(long)q * F + f];
        v += noise * gauss(seed * 0x100000001B3ULL + ((uint64_t)i << 20) + (uint64_t)f);
        v = fminf(1.0f, fmaxf(-1.0f, v));
        D[(long)i * ld + f] = v;
    }
}

}  // namespace

void sqdist_from_proj(const float *proj, int split, int r, int n, long ld, float *dist, hipStream_t s, int r_all, float *reduced)
{
    if (n <= 0) return;
    DLCO_CHECK(r_all <= r || split > 2, -2, "sqdist_from_proj: extra rows need the slab form");
    if (split > 2) hipLaunchKernelGGL(sqdist_split_kernel, dim3((n + 63) / 64), dim3(512), 0, s, proj, split, r, n, ld, dist, std::max(r, r_all), reduced);
    else hipLaunchKernelGGL(sqdist_kernel, dim3((n + 255) / 256), dim3(256), 0, s, proj, split, r, n, ld, dist);
    DLCO_HIP(hipGetLastError());
}

void viol_counts(const float *pd, const float *nd, int B, int32_t *rho, int32_t *kappa, hipStream_t s)
{
    if (B <= 0) return;
    DLCO_CHECK(B <= 8192, -2, "viol_counts: batch too large for one workgroup's LDS");
    hipLaunchKernelGGL(viol_kernel, dim3(1), dim3(256), 2 * B * sizeof(float), s, pd, nd, B, rho, kappa);
    DLCO_HIP(hipGetLastError());
}

void viol_counts_active_rows(const float *pd, const float *nd, int B, int32_t *rho, int32_t *kappa, const int32_t *pos_rows,
                             const int32_t *neg_rows, int slot_lo, int slot_hi, int32_t *ids, float *w, int *k_active, hipStream_t s,
                             int32_t *slots)
{
    if (B <= 0) return;
    DLCO_CHECK(B <= 8192, -2, "viol_counts: batch too large for one workgroup's LDS");
    hipLaunchKernelGGL(viol_active_kernel, dim3(1), dim3(256), 2 * B * sizeof(float), s, pd, nd, B, rho, kappa, pos_rows, neg_rows, slot_lo,
                       slot_hi, ids, w, k_active, slots);
    DLCO_HIP(hipGetLastError());
}

void build_active_rows(const int32_t *pos_rows, const int32_t *neg_rows, const int32_t *rho, const int32_t *kappa,
                       int B, int slot_lo, int slot_hi, int32_t *ids, float *w, int *k_active, hipStream_t s)
{
    hipLaunchKernelGGL(active_rows_kernel, dim3(1), dim3(256), 0, s, pos_rows, neg_rows, rho, kappa, B, slot_lo, slot_hi,
                       ids, w, k_active);
    DLCO_HIP(hipGetLastError());
}

void hinge_rows(const float *pos, int n_pos, const float *neg, int n_neg, float *row_sums, hipStream_t s)
{
    if (n_pos <= 0) return;
    hipLaunchKernelGGL(hinge_rows_kernel, dim3((n_pos + 255) / 256), dim3(256), 0, s, pos, n_pos, neg, n_neg, row_sums);
    DLCO_HIP(hipGetLastError());
}

void sum_f32_to_f64(const float *x, int n, double *out, hipStream_t s)
{
    hipLaunchKernelGGL(sum_f64_kernel, dim3(1), dim3(1024), 0, s, x, n, out);
    DLCO_HIP(hipGetLastError());
}

void trace_f64(const float *A, int F, long ld, double *out, hipStream_t s)
{
    hipLaunchKernelGGL(trace_kernel, dim3(1), dim3(1024), 0, s, A, F, ld, out);
    DLCO_HIP(hipGetLastError());
}

void axpby_inplace(float *y, const float *x, float a, float b, size_t n, hipStream_t s)
{
    if (n == 0) return;
    size_t blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(axpby_kernel, dim3((unsigned)blocks), dim3(256), 0, s, y, x, a, b, n);
    DLCO_HIP(hipGetLastError());
}

void fill_f32(float *p, float v, size_t n, hipStream_t s)
{
    if (n == 0) return;
    size_t blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(fill_kernel, dim3((unsigned)blocks), dim3(256), 0, s, p, v, n);
    DLCO_HIP(hipGetLastError());
}

void scale_rows(float *dst, long ldd, const float *src, long lds, const float *scale, const int32_t *src_rows,
                int rows, int cols, hipStream_t s, const int32_t *src_rows2)
{
    if (rows <= 0) return;
    hipLaunchKernelGGL(scale_rows_kernel, dim3(rows), dim3(256), 0, s, dst, ldd, src, lds, scale, src_rows, src_rows2, rows,
                       cols);
    DLCO_HIP(hipGetLastError());
}

void pack_cols(float *dst, const float *src, long ld, int c0, int cw, int rows, hipStream_t s)
{
    if (rows <= 0) return;
    const long n4 = (long)rows * (cw / 4);
    hipLaunchKernelGGL(pack_cols_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, dst, src, ld, c0, cw, rows);
    DLCO_HIP(hipGetLastError());
}

void unpack_cols(float *dst, long ld, const float *src, int cw, int rows, int world, hipStream_t s)
{
    if (rows <= 0) return;
    const long n4 = (long)world * rows * (cw / 4);
    hipLaunchKernelGGL(unpack_cols_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, dst, ld, src, cw, rows, world);
    DLCO_HIP(hipGetLastError());
}

void emit_w_rows(float *W, long ldw, const float *Q, long ldq, const float *theta, int nw, float mu, float cscale, int F,
                 hipStream_t s, int m_ext, float *wscale)
{
    const int rows = std::max(nw, m_ext);
    if (rows <= 0) return;
    hipLaunchKernelGGL(emit_w_kernel, dim3(rows), dim3(256), 0, s, W, ldw, Q, ldq, theta, nw, mu, cscale, F, m_ext, wscale);
    DLCO_HIP(hipGetLastError());
}

void translate_ids(const int32_t *ids, int base, int n, const int32_t *pa, const int32_t *pb, int32_t *out_a, int32_t *out_b,
                   hipStream_t s)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(translate_ids_kernel, dim3((n + 255) / 256), dim3(256), 0, s, ids, base, n, pa, pb, out_a, out_b);
    DLCO_HIP(hipGetLastError());
}

void synth_rows(float *D, int N, int F, long ld, const float *U, int k, uint64_t seed, float sig_pos, float sig_neg,
                float noise, float jitter, hipStream_t s)
{
    hipLaunchKernelGGL(synth_kernel, dim3(N), dim3(256), k * sizeof(float), s, D, N, F, ld, U, k, seed, sig_pos, sig_neg,
                       noise, jitter);
    DLCO_HIP(hipGetLastError());
}

}  // namespace dlco
