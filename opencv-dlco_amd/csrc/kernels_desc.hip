// kernels_desc.hip — descriptor generation on gfx950: the work of comp-uprjdists
// (src/comp-uprjdists.cpp:298-349) that produces the matrix pj-learn trains on.
//
// The reference computes, for every training PAIR, get_desc() of both patches, two products
// sPRFilters [nsel,4096] x PatchTrans [4096,8] and their clamped difference.  A patch takes part in
// many pairs, so here the work is per PATCH and the table of descriptors stays in HBM (16 GB for
// 500k patches x 8192 floats; the trainer's pair mode then forms Desc1 - Desc2 inside its kernels):
//
//   desc_transform_kernel   one workgroup per patch: blur, gradient, orientation soft-assignment and
//                           the 0.8-quantile normalisation of get_desc (src/vgg-desc.cpp:41-152), all
//                           in LDS; the quantile comes from a radix select, not a sort
//   desc_pool_kernel        Desc = min(PatchTrans^T-major x Filters^T, 1) for a chunk of patches as
//                           ONE product on the f64 matrix cores: cv::gemm accumulates this product in
//                           double (the descriptor entries are sums of 4096 terms clamped at 1), so
//                           the f64 MFMA keeps the results within one double rounding of the oracle's
//   desc_pair_diff_kernel   Dist = Desc[p1] - Desc[p2], Label = (id1 == id2) for callers that want
//                           the reference's pre-differenced matrix
//
// Layouts: PatchTrans of a chunk is [patch][bin][y*64+x] (each bin a plane, pixel-natural order);
// the filters are uploaded with their columns permuted from the reference's transposed pixel order
// (x*64+y, src/vgg-desc.cpp:136-150) to y*64+x, so both operands of the product are K-contiguous.
#include "dlco_internal.hpp"

namespace dlco {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef double f64x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int DP = 64;                 // patch edge
constexpr int DNP = DP * DP;
constexpr int DNB = 8;                 // orientation bins (nAngleBins of the reference's callers)
constexpr int DT = 256;                // threads of the transform kernel
constexpr int DPX = DNP / DT;          // pixels per thread

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

__global__ __launch_bounds__(DT) void desc_transform_kernel(const uint8_t *patches, int n_patches, const float *cf_g, int ks,
                                                            float inv_step, int norm, int qk, float qgamma, float *PT)
{
#pragma clang fp contract(off)
    __shared__ float A[DNP], B[DNP];
    __shared__ float cf[64];
    __shared__ unsigned hist[256];
    __shared__ unsigned wsum[4];
    __shared__ unsigned sel[4];          // 0: prefix bits, 1: remaining rank, 2: count <= v1, 3: min bits above v1
    const int tid = threadIdx.x;
    const long patch = blockIdx.x;
    if (patch >= n_patches) return;
    if (tid < 64) cf[tid] = tid < ks ? cf_g[tid] : 0.f;
    {
        // 16 bytes per thread: pixels 16*tid .. 16*tid+15
        const uint4 raw = reinterpret_cast<const uint4 *>(patches + patch * DNP)[tid];
        const unsigned wds[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
        for (int q = 0; q < 4; q++)
#pragma unroll
            for (int e = 0; e < 4; e++) A[16 * tid + 4 * q + e] = (float)((wds[q] >> (8 * e)) & 255u);   // convertTo CV_32F, :44
    }
    __syncthreads();
    const int r = ks / 2;
    const int x = tid & 63, y0 = tid >> 6;
    // ---- GaussianBlur, :46: rows left to right, columns in symmetric form ----
#pragma unroll 4
    for (int u = 0; u < DPX; u++) {
        const int y = y0 + 4 * u;
        float s = cf[0] * A[y * DP + clampi(x - r, 0, DP - 1)];
        for (int k = 1; k < ks; k++) { const float p = cf[k] * A[y * DP + clampi(x - r + k, 0, DP - 1)]; s = s + p; }
        B[y * DP + x] = s;
    }
    __syncthreads();
#pragma unroll 4
    for (int u = 0; u < DPX; u++) {
        const int y = y0 + 4 * u;
        float s = cf[r] * B[y * DP + x];
        for (int k = 1; k <= r; k++) {
            const float pr2 = B[clampi(y - k, 0, DP - 1) * DP + x] + B[clampi(y + k, 0, DP - 1) * DP + x];
            const float p = cf[r + k] * pr2;
            s = s + p;
        }
        A[y * DP + x] = s;
    }
    __syncthreads();
    // ---- gradient, magnitude, orientation ratio, :48-70 ----
    float mag[DPX], ratio[DPX];
    const double kPi = 3.1415926535897932384626433832795;
#pragma unroll
    for (int u = 0; u < DPX; u++) {
        const int y = y0 + 4 * u;
        const float ix = A[y * DP + clampi(x + 1, 0, DP - 1)] - A[y * DP + clampi(x - 1, 0, DP - 1)];
        const float iy = A[clampi(y + 1, 0, DP - 1) * DP + x] - A[clampi(y - 1, 0, DP - 1) * DP + x];
        const float xx = ix * ix, yy = iy * iy;
        mag[u] = sqrtf(xx + yy);
        const float at = (float)atan2((double)iy, (double)ix);        // atan2 rounded to float
        const float ang = (float)((double)at + kPi);
        const float sc = ang * inv_step;
        ratio[u] = sc - 0.5f;
    }
    // ---- 0.8 quantile of the magnitudes (:106-133): order statistics qk-1 and qk by radix select ----
    float scale = 1.0f;
    if (norm) {
        __syncthreads();
#pragma unroll
        for (int u = 0; u < DPX; u++) B[tid + DT * u] = mag[u];
        if (tid == 0) { sel[0] = 0u; sel[1] = (unsigned)(qk - 1); sel[2] = 0u; sel[3] = 0xffffffffu; }
        unsigned mask = 0u;
        for (int shift = 24; shift >= 0; shift -= 8) {
            hist[tid] = 0u;
            __syncthreads();
            const unsigned prefix = sel[0], want = sel[1];
#pragma unroll
            for (int u = 0; u < DPX; u++) {
                const unsigned v = __float_as_uint(B[tid + DT * u]);
                if ((v & mask) == prefix) atomicAdd(&hist[(v >> shift) & 255u], 1u);
            }
            __syncthreads();
            // inclusive scan of the 256 bins: one bin per thread
            const unsigned h = hist[tid];
            unsigned inc = h;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const unsigned t = __shfl_up(inc, o, 64);
                if ((tid & 63) >= o) inc += t;
            }
            if ((tid & 63) == 63) wsum[tid >> 6] = inc;
            __syncthreads();
            unsigned base = 0u;
            for (int w = 0; w < (tid >> 6); w++) base += wsum[w];
            inc += base;
            const unsigned exc = inc - h;
            __syncthreads();                                         // every thread has read sel[] and wsum[]
            if (h > 0u && exc <= want && want < inc) { sel[0] = prefix | ((unsigned)tid << shift); sel[1] = want - exc; }
            mask |= 255u << shift;
            __syncthreads();
        }
        const unsigned v1b = sel[0];
        unsigned cnt = 0u, mn = 0xffffffffu;
#pragma unroll
        for (int u = 0; u < DPX; u++) {
            const unsigned v = __float_as_uint(B[tid + DT * u]);
            if (v <= v1b) cnt++;
            else mn = v < mn ? v : mn;
        }
        atomicAdd(&sel[2], cnt);
        atomicMin(&sel[3], mn);
        __syncthreads();
        const float v1 = __uint_as_float(v1b);
        const float v2 = sel[2] >= (unsigned)(qk + 1) ? v1 : __uint_as_float(sel[3]);
        const float t1 = (1.0f - qgamma) * v1, t2 = qgamma * v2;
        const float T = t1 + t2;
        if (T != 0.0f) scale = (float)(1.0 / (double)(T / (float)DNB));     // GMag /= (T / nAngleBins)
        else norm = 0;
    }
    // ---- soft assignment, one plane per bin ----
    float *out = PT + patch * (long)(DNB * DNP);
#pragma unroll
    for (int u = 0; u < DPX; u++) {
        const int i = (y0 + 4 * u) * DP + x;
        const float rt = ratio[u];
        const float off = rt - floorf(rt);
        const float c1 = ceilf(rt - 1.0f);
        const int b1 = (c1 == -1.0f) ? DNB - 1 : (int)c1;
        const int b2 = (b1 + 1 > DNB - 1) ? 0 : b1 + 1;
        const float m = norm ? mag[u] * scale : mag[u];
        const float w1 = (1.0f - off) * m, w2 = off * m;
#pragma unroll
        for (int b = 0; b < DNB; b++) out[b * DNP + i] = (b == b1) ? w1 : ((b == b2) ? w2 : 0.f);
    }
}

// ---- pooled descriptors: C[n][f] = sum_k PT[n][k] * Fl[f][k], both K-contiguous (K = 4096) --------------
// 128 x 128 tile per workgroup, 8 waves of 64 x 32, v_mfma_f64_16x16x4: lane l supplies A[l%16][kq] and
// B[kq][l%16] with kq = l/16 and holds C[l/16 + 4r][l%16], r = 0..3.  Within a 16-wide k chunk lane group g reads the
// four consecutive k = 4g..4g+3 of its row as one 16-byte LDS read and MFMA step s uses element s on both
// operands: the k order inside a chunk is permuted the same way for A and B, which a sum does not see
// beyond the rounding of the double accumulation.
constexpr int GT = 128;                // tile edge (both sides)
constexpr int GK = 32;                 // k per staged chunk
constexpr int GP = GK + 4;             // LDS pitch in floats
constexpr int GTH = 512;               // eight waves (two per SIMD): a single wave per SIMD does not keep the matrix pipe full

__global__ __launch_bounds__(GTH, 4) void desc_pool_kernel(const float *PT, const float *Fl, int n_rows, int F8pad, int nsel, float *desc,
                                                        long desc_ld, int tiles_f)
{
    __shared__ float As[2][GT * GP], Bs[2][GT * GP];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // workgroups w, w+8, ... run on one XCD: give them the same PatchTrans tile and different filter tiles
    const int w = blockIdx.x;
    const int xcd = w & 7, slot = w >> 3;
    const int tn = (slot / tiles_f) * 8 + xcd, tf = slot % tiles_f;
    if ((long)tn * GT >= n_rows) return;
    const float *Ag = PT + (long)tn * GT * DNP;
    const float *Bg = Fl + (long)tf * GT * DNP;
    const int lrow = tid >> 3, lseg = tid & 7;          // loader: rows lrow + 64*u, 16-byte segment lseg
    f32x4 ra[2], rb[2];
    auto gload = [&](int kc) {
#pragma unroll
        for (int u = 0; u < 2; u++) {
            ra[u] = *reinterpret_cast<const f32x4 *>(Ag + (long)(lrow + 64 * u) * DNP + kc * GK + 4 * lseg);
            rb[u] = *reinterpret_cast<const f32x4 *>(Bg + (long)(lrow + 64 * u) * DNP + kc * GK + 4 * lseg);
        }
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int u = 0; u < 2; u++) {
            *reinterpret_cast<f32x4 *>(&As[buf][(lrow + 64 * u) * GP + 4 * lseg]) = ra[u];
            *reinterpret_cast<f32x4 *>(&Bs[buf][(lrow + 64 * u) * GP + 4 * lseg]) = rb[u];
        }
    };
    const int wm = (wave >> 2) * 64, wf = (wave & 3) * 32;       // a wave: 64 PatchTrans rows x 32 filters
    const int li = lane & 15, lg = lane >> 4;
    // the f64 MFMA leaves rows lg, lg+4, lg+8, lg+12 of its 16 x 16 result in a lane's four registers: feed
    // it the PatchTrans rows in the order 0,4,8,12,1,5,.. so that those are four CONSECUTIVE rows (bins)
    const int lrowA = 4 * (li & 3) + (li >> 2);
    f64x4 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 2; j++) acc[i][j] = f64x4{0.0, 0.0, 0.0, 0.0};
    constexpr int NCH = DNP / GK;
    gload(0);
    lstore(0);
    __syncthreads();
    for (int kc = 0; kc < NCH; kc++) {
        const int buf = kc & 1;
        if (kc + 1 < NCH) gload(kc + 1);
#pragma unroll
        for (int c = 0; c < GK / 16; c++) {
            f32x4 fa[4], fb[2];
#pragma unroll
            for (int i = 0; i < 4; i++) fa[i] = *reinterpret_cast<const f32x4 *>(&As[buf][(wm + 16 * i + lrowA) * GP + 16 * c + 4 * lg]);
#pragma unroll
            for (int j = 0; j < 2; j++) fb[j] = *reinterpret_cast<const f32x4 *>(&Bs[buf][(wf + 16 * j + li) * GP + 16 * c + 4 * lg]);
#pragma unroll
            for (int s = 0; s < 4; s++)
#pragma unroll
                for (int i = 0; i < 4; i++)
#pragma unroll
                    for (int j = 0; j < 2; j++)
                        acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64((double)fa[i][s], (double)fb[j][s], acc[i][j], 0, 0, 0);
        }
        if (kc + 1 < NCH) lstore(buf ^ 1);
        __syncthreads();
    }
    // ---- crop at 1 (:324-325) and store: row n = patch*8 + bin, column f -> desc[patch][f*8 + bin] ----
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int n0 = tn * GT + wm + 16 * i + 4 * lg;          // four consecutive bins of one patch
        const long prow = n0 >> 3;
        const int b0 = n0 & 7;
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const int f = tf * GT + wf + 16 * j + li;
            if (f < nsel && n0 < n_rows) {
                f32x4 v;
#pragma unroll
                for (int rr = 0; rr < 4; rr++) { const float q = (float)acc[i][j][rr]; v[rr] = q < 1.0f ? q : 1.0f; }
                *reinterpret_cast<f32x4 *>(desc + prow * desc_ld + (long)f * 8 + b0) = v;
            }
        }
    }
    (void)F8pad;
}

__global__ __launch_bounds__(256) void desc_pair_diff_kernel(const float *desc, long ld, int F, const int32_t *pairs, long n_pairs,
                                                             float *dist, uint8_t *label)
{
    const long pr = blockIdx.x;
    if (pr >= n_pairs) return;
    const int32_t *q = pairs + pr * 4;
    const f32x4 *a = reinterpret_cast<const f32x4 *>(desc + (long)q[0] * ld);
    const f32x4 *b = reinterpret_cast<const f32x4 *>(desc + (long)q[2] * ld);
    f32x4 *o = reinterpret_cast<f32x4 *>(dist + pr * (long)F);
    for (int c = threadIdx.x; c < F / 4; c += 256) o[c] = a[c] - b[c];                   // :327
    if (threadIdx.x == 0 && label) label[pr] = (q[1] == q[3]) ? 1 : 0;                    // :268-272
}

// comp-fulldists (src/comp-fulldists.cpp:318-343): per pair and pooling region g (8 ring rows x 8 bins = 64
// consecutive descriptor entries), dist[g] = sum (Desc2 - Desc1)^2.  16 lanes per region: a 16-byte piece each,
// squared and summed in the reference's grouping (the 8 bins of a row first, then the 8 rows).
__global__ __launch_bounds__(256) void desc_full_dist_kernel(const float *desc, long ld, int n_groups, int n_pairs, float *dist)
{
    const long pr = blockIdx.y;
    const int g = blockIdx.x * 16 + (threadIdx.x >> 4), l = threadIdx.x & 15;
    if (pr >= n_pairs || g >= n_groups) return;
    const f32x4 a = *reinterpret_cast<const f32x4 *>(desc + pr * ld + (long)g * 64 + 4 * l);
    const f32x4 b = *reinterpret_cast<const f32x4 *>(desc + (pr + n_pairs) * ld + (long)g * 64 + 4 * l);
    const f32x4 d = b - a;
    float s = ((d[0] * d[0] + d[1] * d[1]) + d[2] * d[2]) + d[3] * d[3];      // half a row: 4 of its 8 bins
    s += __shfl_xor(s, 1, 64);                                                 // the row
    s += __shfl_xor(s, 2, 64);
    s += __shfl_xor(s, 4, 64);
    s += __shfl_xor(s, 8, 64);                                                 // the 8 rows
    if (l == 0) dist[pr * n_groups + g] = s;
}

__global__ __launch_bounds__(256) void desc_permute_filters_kernel(const float *src, int nsel, float *dst)
{
    // dst[f][y*64+x] = src[f][x*64+y]
    __shared__ float t[DP][DP + 1];
    const int f = blockIdx.x;
    if (f >= nsel) return;
    for (int i = threadIdx.x; i < DNP; i += 256) t[i >> 6][i & 63] = src[(long)f * DNP + i];
    __syncthreads();
    for (int i = threadIdx.x; i < DNP; i += 256) dst[(long)f * DNP + i] = t[i & 63][i >> 6];
}

}  // namespace

void desc_permute_filters(const float *src, int nsel, float *dst, hipStream_t s)
{
    if (nsel <= 0) return;
    hipLaunchKernelGGL(desc_permute_filters_kernel, dim3(nsel), dim3(256), 0, s, src, nsel, dst);
    DLCO_HIP(hipGetLastError());
}

void desc_transform(const uint8_t *patches, int n_patches, const float *cf, int ks, float inv_step, bool norm, int qk, float qgamma,
                    float *PT, hipStream_t s)
{
    if (n_patches <= 0) return;
    hipLaunchKernelGGL(desc_transform_kernel, dim3(n_patches), dim3(DT), 0, s, patches, n_patches, cf, ks, inv_step, norm ? 1 : 0, qk,
                       qgamma, PT);
    DLCO_HIP(hipGetLastError());
}

// PT: [n_patches_pad*8][4096] with n_patches_pad a multiple of 16 (rows beyond n_patches*8 are read but
// never stored); Fl: [nsel_pad][4096] with nsel_pad a multiple of 128 (rows beyond nsel zero).
void desc_pool(const float *PT, int n_patches, const float *Fl, int nsel, int nsel_pad, float *desc, long desc_ld, hipStream_t s)
{
    if (n_patches <= 0 || nsel <= 0) return;
    const int n_rows = n_patches * DNB;
    const int tiles_n = (n_rows + GT - 1) / GT, tiles_f = nsel_pad / GT;
    const int groups = (tiles_n + 7) / 8;
    hipLaunchKernelGGL(desc_pool_kernel, dim3(groups * tiles_f * 8), dim3(GTH), 0, s, PT, Fl, n_rows, nsel_pad * 8, nsel, desc, desc_ld,
                       tiles_f);
    DLCO_HIP(hipGetLastError());
}

// desc: [2 * n_pairs][ld], rows [0, n_pairs) the first patch of every pair, rows [n_pairs, 2 n_pairs) the second
void desc_full_dist(const float *desc, long ld, int n_groups, int n_pairs, float *dist, hipStream_t s)
{
    if (n_pairs <= 0 || n_groups <= 0) return;
    hipLaunchKernelGGL(desc_full_dist_kernel, dim3((n_groups + 15) / 16, n_pairs), dim3(256), 0, s, desc, ld, n_groups, n_pairs, dist);
    DLCO_HIP(hipGetLastError());
}

void desc_pair_diff(const float *desc, long ld, int F, const int32_t *pairs, long n_pairs, float *dist, uint8_t *label, hipStream_t s)
{
    if (n_pairs <= 0) return;
    hipLaunchKernelGGL(desc_pair_diff_kernel, dim3((unsigned)n_pairs), dim3(256), 0, s, desc, ld, F, pairs, n_pairs, dist, label);
    DLCO_HIP(hipGetLastError());
}

}  // namespace dlco
