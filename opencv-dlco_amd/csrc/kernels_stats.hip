// kernels_stats.hip — ComputePJStats' ranking and ROC sweep on the device
// (src/misc.cpp:297-332): sort the N squared distances ascending, sweep TP/FP counts,
// FPR at the first TPR >= 0.95f, AUC by the shoelace formula of cv::contourArea.
//
// The sort is rocPRIM's stable LSD radix sort on (distance, row id) pairs, so ties keep
// ascending row order (the reference's std::sort leaves ties unordered).  Everything else
// is hand-written: the sweep is an integer scan, the area a double reduction.
#include "dlco_internal.hpp"

#include <cstring>
#include <string.h>
#include <rocprim/rocprim.hpp>

namespace dlco {

struct RocWork {
    int n_max = 0;
    DevBuf<float> keys_out;
    DevBuf<int32_t> idx_in, idx_out;
    DevBuf<int32_t> tp, fp;          // flags then inclusive scans
    DevBuf<char> tmp;
    size_t tmp_bytes = 0;
    DevBuf<double> area_part;
    DevBuf<int32_t> first_part;
    DevBuf<float> out_f;             // [0] fpr95
    DevBuf<double> out_d;            // [0] auc
};

namespace {

__global__ void iota_kernel(int32_t *p, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = i;
}

__global__ void flags_kernel(const int32_t *order, const uint8_t *labels, int n, int32_t *tp, int32_t *fp)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint8_t l = labels[order[i]];
    tp[i] = (l == 1) ? 1 : 0;       // src/misc.cpp:308-309 (labels other than 0/1 count for neither)
    fp[i] = (l == 0) ? 1 : 0;
}

// TPR_i = float(tp_i) * float(1/tplast) — Mat /= scalar multiplies by the reciprocal in fp32 [OpenCV-src]
__device__ __forceinline__ void rates(const int32_t *tp, const int32_t *fp, int i, int n, float ts, float fs, float &x,
                                      float &y)
{
    if (i >= n) { x = 1.0f; y = 0.0f; return; }       // closing point (1,0), src/misc.cpp:329-330
    x = (float)fp[i] * fs;
    y = (float)tp[i] * ts;
}

__global__ __launch_bounds__(256) void roc_partial_kernel(const int32_t *tp, const int32_t *fp, int n,
                                                          double *area_part, int32_t *first_part)
{
    __shared__ double sa[4];
    __shared__ int sf[4];
    const float tplast = (float)tp[n - 1], fplast = (float)fp[n - 1];
    const float ts = (float)(1.0 / (double)tplast), fs = (float)(1.0 / (double)fplast);
    double a = 0.0;
    int first = 0x7fffffff;
    // contour points 0..n (n+1 points); term i uses prev = point i-1 (point n for i = 0)
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i <= n; i += gridDim.x * blockDim.x) {
        float x, y, px, py;
        rates(tp, fp, i, n, ts, fs, x, y);
        rates(tp, fp, i == 0 ? n : i - 1, n, ts, fs, px, py);
        a += (double)px * (double)y - (double)py * (double)x;
        if (i < n && y >= 0.95f) first = min(first, i);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        a += __shfl_xor(a, o, 64);
        first = min(first, __shfl_xor(first, o, 64));
    }
    if ((threadIdx.x & 63) == 0) { sa[threadIdx.x >> 6] = a; sf[threadIdx.x >> 6] = first; }
    __syncthreads();
    if (threadIdx.x == 0) {
        area_part[blockIdx.x] = sa[0] + sa[1] + sa[2] + sa[3];
        first_part[blockIdx.x] = min(min(sf[0], sf[1]), min(sf[2], sf[3]));
    }
}

__global__ void roc_final_kernel(const double *area_part, const int32_t *first_part, int nblk, const int32_t *tp,
                                 const int32_t *fp, int n, float *fpr95, double *auc)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double a = 0.0;
    int first = 0x7fffffff;
    for (int b = 0; b < nblk; b++) { a += area_part[b]; first = min(first, first_part[b]); }
    *auc = fabs(a * 0.5);
    if (first == 0x7fffffff) { *fpr95 = -1.0f; return; }
    const float fs = (float)(1.0 / (double)(float)fp[n - 1]);
    *fpr95 = (float)fp[first] * fs;
}

constexpr int ROC_BLOCKS = 512;

}  // namespace

RocWork *roc_work_create(int n_max)
{
    RocWork *w = new RocWork();
    w->n_max = n_max;
    w->keys_out.alloc(n_max);
    w->idx_in.alloc(n_max);
    w->idx_out.alloc(n_max);
    w->tp.alloc(n_max);
    w->fp.alloc(n_max);
    size_t b1 = 0, b2 = 0;
    DLCO_HIP(rocprim::radix_sort_pairs(nullptr, b1, (const float *)nullptr, (float *)nullptr, (const int32_t *)nullptr,
                                       (int32_t *)nullptr, (size_t)n_max, 0, 32, (hipStream_t)0));
    DLCO_HIP(rocprim::inclusive_scan(nullptr, b2, (int32_t *)nullptr, (int32_t *)nullptr, (size_t)n_max,
                                     rocprim::plus<int32_t>(), (hipStream_t)0));
    w->tmp_bytes = b1 > b2 ? b1 : b2;
    w->tmp.alloc(w->tmp_bytes + 256);
    w->area_part.alloc(ROC_BLOCKS);
    w->first_part.alloc(ROC_BLOCKS);
    w->out_f.alloc(4);
    w->out_d.alloc(4);
    return w;
}

void roc_work_destroy(RocWork *w) { delete w; }

void roc_stats(RocWork *w, const float *dist_dev, const uint8_t *labels_dev, int n, float *fpr95, double *auc,
               hipStream_t s)
{
    DLCO_CHECK(n >= 1 && n <= w->n_max, -2, "roc_stats: n out of range");
    const int nb = (n + 255) / 256;
    hipLaunchKernelGGL(iota_kernel, dim3(nb), dim3(256), 0, s, w->idx_in.p, n);
    size_t bytes = w->tmp_bytes;
    DLCO_HIP(rocprim::radix_sort_pairs((void *)w->tmp.p, bytes, dist_dev, w->keys_out.p, (const int32_t *)w->idx_in.p,
                                       w->idx_out.p, (size_t)n, 0, 32, s));
    hipLaunchKernelGGL(flags_kernel, dim3(nb), dim3(256), 0, s, (const int32_t *)w->idx_out.p, labels_dev, n, w->tp.p,
                       w->fp.p);
    bytes = w->tmp_bytes;
    DLCO_HIP(rocprim::inclusive_scan((void *)w->tmp.p, bytes, w->tp.p, w->tp.p, (size_t)n, rocprim::plus<int32_t>(), s));
    bytes = w->tmp_bytes;
    DLCO_HIP(rocprim::inclusive_scan((void *)w->tmp.p, bytes, w->fp.p, w->fp.p, (size_t)n, rocprim::plus<int32_t>(), s));
    hipLaunchKernelGGL(roc_partial_kernel, dim3(ROC_BLOCKS), dim3(256), 0, s, (const int32_t *)w->tp.p,
                       (const int32_t *)w->fp.p, n, w->area_part.p, w->first_part.p);
    hipLaunchKernelGGL(roc_final_kernel, dim3(1), dim3(64), 0, s, (const double *)w->area_part.p,
                       (const int32_t *)w->first_part.p, ROC_BLOCKS, (const int32_t *)w->tp.p, (const int32_t *)w->fp.p,
                       n, w->out_f.p, w->out_d.p);
    DLCO_HIP(hipGetLastError());
    DLCO_HIP(hipMemcpyAsync(fpr95, w->out_f.p, sizeof(float), hipMemcpyDeviceToHost, s));
    DLCO_HIP(hipMemcpyAsync(auc, w->out_d.p, sizeof(double), hipMemcpyDeviceToHost, s));
    DLCO_HIP(hipStreamSynchronize(s));
}

}  // namespace dlco
