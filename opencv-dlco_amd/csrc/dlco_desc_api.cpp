// dlco_desc_api.cpp — C ABI of descriptor generation (comp-uprjdists): see the
// dlco_desc_* block of include/dlco.h.  Host orchestration and the filter selection of
// SelectPRFilters (src/misc.cpp:78-168, pure host logic); the patch transform, the pooling product and
// the pair differences run in HIP kernels (kernels_desc.hip).
#include "../../include/dlco.h"

#include "dlco_internal.hpp"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <numeric>

using namespace dlco;

namespace dlco {
void desc_permute_filters(const float *src, int nsel, float *dst, hipStream_t s);
void desc_transform(const uint8_t *patches, int n_patches, const float *cf, int ks, float inv_step, bool norm, int qk, float qgamma,
                    float *PT, hipStream_t s);
void desc_pool(const float *PT, int n_patches, const float *Fl, int nsel, int nsel_pad, float *desc, long desc_ld, hipStream_t s);
void desc_pair_diff(const float *desc, long ld, int F, const int32_t *pairs, long n_pairs, float *dist, uint8_t *label, hipStream_t s);
void desc_full_dist(const float *desc, long ld, int n_groups, int n_pairs, float *dist, hipStream_t s);
}

namespace {
constexpr int kPix = 64 * 64;
constexpr int kBins = 8;
constexpr int kChunk = 2048;            // patches per launch pair: 256 MB of PatchTrans
}

struct dlco_desc_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    std::string err;
    float sigma = 1.4f;
    bool norm = true;
    int ks = 13, qk = 1;
    float inv_step = 0.f, qgamma = 0.f;
    int nsel = 0, nsel_pad = 0;
    DevBuf<float> cf, filt, pt, desc_chunk;
    DevBuf<uint8_t> patches;
    double last_ms = 0.0;
};

static thread_local std::string g_desc_error;

namespace {

template <typename Fn>
int guarded(dlco_desc_ctx *c, Fn &&fn)
{
    try {
        fn();
        return DLCO_OK;
    } catch (const Error &e) {
        g_desc_error = e.what();
        if (c) c->err = e.what();
        return e.code;
    } catch (const std::exception &e) {
        g_desc_error = e.what();
        if (c) c->err = e.what();
        return DLCO_ERR_INVALID;
    }
}

void sync(dlco_desc_ctx *c) { DLCO_HIP(hipStreamSynchronize(c->stream)); }

// descriptors of patches [n,64,64] (host) into dst: device memory when dst_dev, else host; row pitch ld floats
void compute(dlco_desc_ctx *c, const uint8_t *patches_host, int64_t n, float *dst, bool dst_dev, long ld)
{
    DLCO_CHECK(c->nsel > 0, DLCO_ERR_INVALID, "dlco_desc: set the filters first");
    const int F = c->nsel * kBins;
    DLCO_CHECK(ld >= F, DLCO_ERR_INVALID, "dlco_desc: row pitch below the descriptor size");
    DLCO_HIP(hipSetDevice(c->device));
    c->patches.alloc((size_t)kChunk * kPix);
    c->pt.alloc((size_t)kChunk * kBins * kPix);
    if (!dst_dev) c->desc_chunk.alloc((size_t)kChunk * F);
    hipEvent_t e0, e1;
    DLCO_HIP(hipEventCreate(&e0));
    DLCO_HIP(hipEventCreate(&e1));
    double ms_total = 0.0;
    for (int64_t p0 = 0; p0 < n; p0 += kChunk) {
        const int cnt = (int)std::min<int64_t>(kChunk, n - p0);
        DLCO_HIP(hipMemcpyAsync(c->patches.p, patches_host + p0 * kPix, (size_t)cnt * kPix, hipMemcpyHostToDevice, c->stream));
        DLCO_HIP(hipEventRecord(e0, c->stream));
        desc_transform(c->patches.p, cnt, c->cf.p, c->ks, c->inv_step, c->norm, c->qk, c->qgamma, c->pt.p, c->stream);
        float *out = dst_dev ? dst + p0 * ld : c->desc_chunk.p;
        const long old = dst_dev ? ld : F;
        desc_pool(c->pt.p, cnt, c->filt.p, c->nsel, c->nsel_pad, out, old, c->stream);
        DLCO_HIP(hipEventRecord(e1, c->stream));
        if (!dst_dev)
            DLCO_HIP(hipMemcpy2DAsync(dst + p0 * ld, (size_t)ld * sizeof(float), c->desc_chunk.p, (size_t)F * sizeof(float),
                                      (size_t)F * sizeof(float), (size_t)cnt, hipMemcpyDeviceToHost, c->stream));
        sync(c);
        float ms = 0.f;
        DLCO_HIP(hipEventElapsedTime(&ms, e0, e1));
        ms_total += ms;
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    c->last_ms = ms_total;
}

}  // namespace

extern "C" {

const char *dlco_desc_last_error(const dlco_desc_ctx *c) { return c ? c->err.c_str() : g_desc_error.c_str(); }

int dlco_desc_create(dlco_desc_ctx **out, float init_sigma, int32_t n_angle_bins, int32_t norm, int32_t device)
{
    if (!out) return DLCO_ERR_INVALID;
    *out = nullptr;
    dlco_desc_ctx *c = nullptr;
    const int rc = guarded(nullptr, [&] {
        DLCO_CHECK(n_angle_bins == kBins, DLCO_ERR_INVALID, "dlco_desc: nAngleBins must be 8 (the callers' value, src/comp-uprjdists.cpp:66)");
        DLCO_CHECK(init_sigma > 0.f, DLCO_ERR_INVALID, "dlco_desc: InitSigma must be positive");
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device >= ndev)
            throw Error(DLCO_ERR_NODEVICE, "no usable HIP device (this library has no CPU fallback)");
        DLCO_HIP(hipSetDevice(device));
        hipDeviceProp_t prop;
        DLCO_HIP(hipGetDeviceProperties(&prop, device));
        if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
            throw Error(DLCO_ERR_NODEVICE, std::string("device is ") + prop.gcnArchName + ", this build targets gfx950 only");
        c = new dlco_desc_ctx();
        c->device = device; c->sigma = init_sigma; c->norm = norm != 0;
        DLCO_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
        // Gaussian taps exactly as cv::getGaussianKernel(ksize, sigma, CV_32F) forms them [OpenCV-src]
        int ks = (int)std::lrint((double)init_sigma * 4.0 * 2.0 + 1.0) | 1;
        DLCO_CHECK(ks <= 63, DLCO_ERR_INVALID, "dlco_desc: InitSigma too large for a 64 x 64 patch (kernel above 63 taps)");
        float cf[64] = {0};
        {
            const double scale2x = -0.5 / ((double)init_sigma * (double)init_sigma);
            double sum = 0.0;
            for (int i = 0; i < ks; i++) {
                const double x = i - (ks - 1) * 0.5;
                cf[i] = (float)std::exp(scale2x * x * x);
                sum += cf[i];
            }
            sum = 1.0 / sum;
            for (int i = 0; i < ks; i++) cf[i] = (float)(cf[i] * sum);
        }
        c->ks = ks;
        c->cf.alloc(64);
        DLCO_HIP(hipMemcpyAsync(c->cf.p, cf, sizeof(cf), hipMemcpyHostToDevice, c->stream));
        const double kPi = 3.1415926535897932384626433832795;
        const float step = (float)(2.0f * kPi / (double)(float)n_angle_bins);      // src/vgg-desc.cpp:71
        c->inv_step = (float)(1.0 / (double)step);
        // mquantiles constants for n = 4096, q = 0.8 (src/vgg-desc.cpp:113-126)
        const int n = kPix;
        const float aleph = (float)n * 0.8f + 0.5f;
        int k = (int)std::floor(aleph);
        if (k >= n - 1) k = n - 1;
        if (k <= 1) k = 1;
        float g = aleph - (float)k;
        if (g >= 1.0f) g = 1.0f;
        if (g <= 0.0f) g = 0.0f;
        c->qk = k; c->qgamma = g;
        sync(c);
    });
    if (rc != DLCO_OK) { delete c; return rc; }
    *out = c;
    return DLCO_OK;
}

void dlco_desc_destroy(dlco_desc_ctx *c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) { (void)hipStreamSynchronize(c->stream); (void)hipStreamDestroy(c->stream); }
    delete c;
}

// SelectPRFilters (src/misc.cpp:78-168): rows of PRFilters [8*wcols, cols] whose w entry is positive and
// that are not all zero, duplicates removed, sorted ascending lexicographically.  out may be NULL to
// query the count.  Pure host logic.
int dlco_desc_select_filters(const float *pr_filters, int32_t rows, int32_t cols, const float *w, int32_t wcols, float *out,
                             int32_t *nsel_out)
{
    if (!pr_filters || !w || !nsel_out || rows != wcols * 8 || cols < 1) return DLCO_ERR_INVALID;
    std::vector<int> keep;
    for (int i = 0; i < wcols; i++)
        for (int j = 0; j < 8; j++) {
            const int row = i * 8 + j;
            if (!(w[i] > 0.0f)) continue;
            const float *f = pr_filters + (size_t)row * cols;
            bool any = false;
            for (int k = 0; k < cols && !any; k++) any = f[k] != 0.0f;
            if (any) keep.push_back(row);
        }
    auto less = [&](int a, int b) {
        const float *x = pr_filters + (size_t)a * cols, *y = pr_filters + (size_t)b * cols;
        for (int k = 0; k < cols; k++) {
            if (x[k] == y[k]) continue;
            return x[k] < y[k];
        }
        return false;
    };
    auto same = [&](int a, int b) {
        const float *x = pr_filters + (size_t)a * cols, *y = pr_filters + (size_t)b * cols;
        for (int k = 0; k < cols; k++)
            if (!(x[k] == y[k])) return false;
        return true;
    };
    std::stable_sort(keep.begin(), keep.end(), less);
    keep.erase(std::unique(keep.begin(), keep.end(), same), keep.end());
    *nsel_out = (int32_t)keep.size();
    if (out)
        for (size_t i = 0; i < keep.size(); i++) std::memcpy(out + i * cols, pr_filters + (size_t)keep[i] * cols, (size_t)cols * sizeof(float));
    return DLCO_OK;
}

// sPRFilters [nsel, 4096] in the reference's column order (pixel x*64+y, the transposed patch)
int dlco_desc_set_filters(dlco_desc_ctx *c, const float *filters_host, int32_t nsel)
{
    if (!c || !filters_host || nsel < 1) return DLCO_ERR_INVALID;
    return guarded(c, [&] {
        DLCO_CHECK(nsel <= 65536, DLCO_ERR_INVALID, "dlco_desc: too many filters");
        DLCO_HIP(hipSetDevice(c->device));
        c->nsel = nsel;
        c->nsel_pad = (nsel + 127) / 128 * 128;
        DevBuf<float> raw;
        raw.alloc((size_t)nsel * kPix);
        DLCO_HIP(hipMemcpyAsync(raw.p, filters_host, (size_t)nsel * kPix * sizeof(float), hipMemcpyHostToDevice, c->stream));
        c->filt.alloc((size_t)c->nsel_pad * kPix);
        c->filt.zero(c->stream);
        desc_permute_filters(raw.p, nsel, c->filt.p, c->stream);
        sync(c);
    });
}

int32_t dlco_desc_size(const dlco_desc_ctx *c) { return c ? c->nsel * kBins : 0; }

// get_desc of ONE patch in the reference's layout: PatchTrans [4096][8], row x*64+y (src/vgg-desc.cpp:41-152)
int dlco_desc_transform(dlco_desc_ctx *c, const uint8_t *patch_host, float *patch_trans_host)
{
    if (!c || !patch_host || !patch_trans_host) return DLCO_ERR_INVALID;
    return guarded(c, [&] {
        DLCO_HIP(hipSetDevice(c->device));
        c->patches.alloc((size_t)kChunk * kPix);
        c->pt.alloc((size_t)kChunk * kBins * kPix);
        DLCO_HIP(hipMemcpyAsync(c->patches.p, patch_host, kPix, hipMemcpyHostToDevice, c->stream));
        desc_transform(c->patches.p, 1, c->cf.p, c->ks, c->inv_step, c->norm, c->qk, c->qgamma, c->pt.p, c->stream);
        std::vector<float> planes((size_t)kBins * kPix);
        DLCO_HIP(hipMemcpyAsync(planes.data(), c->pt.p, planes.size() * sizeof(float), hipMemcpyDeviceToHost, c->stream));
        sync(c);
        for (int y = 0; y < 64; y++)
            for (int x = 0; x < 64; x++)
                for (int b = 0; b < kBins; b++) patch_trans_host[(size_t)(x * 64 + y) * kBins + b] = planes[(size_t)b * kPix + y * 64 + x];
    });
}

// Desc [n, nsel*8] = min(sPRFilters * get_desc(patch), 1) per patch, row-major like Mat Desc1 flattened
int dlco_desc_compute(dlco_desc_ctx *c, const uint8_t *patches_host, int64_t n, float *desc_host)
{
    if (!c || !patches_host || !desc_host || n < 1) return DLCO_ERR_INVALID;
    return guarded(c, [&] { compute(c, patches_host, n, desc_host, false, (long)c->nsel * kBins); });
}

int dlco_desc_compute_device(dlco_desc_ctx *c, const uint8_t *patches_host, int64_t n, float *desc_dev, int64_t ld)
{
    if (!c || !patches_host || !desc_dev || n < 1) return DLCO_ERR_INVALID;
    return guarded(c, [&] { compute(c, patches_host, n, desc_dev, true, (long)ld); });
}

// The output of comp-uprjdists: Distance [n_pairs, nsel*8] and Label [n_pairs] for pairs [n_pairs,4] =
// (patchID1, 3DpointID1, patchID2, 3DpointID2).  Descriptors are computed once per patch.
namespace {
// sink of the non-streaming entry points: rows land in the caller's whole-matrix buffers
struct CopySink { float *dist; uint8_t *label; int64_t cols; };
int copy_sink(void *user, int64_t row0, int64_t rows, const float *dist, const uint8_t *label)
{
    CopySink *k = static_cast<CopySink *>(user);
    std::memcpy(k->dist + row0 * k->cols, dist, (size_t)rows * k->cols * sizeof(float));
    if (k->label) std::memcpy(k->label + row0, label, (size_t)rows);
    return 0;
}
void check_pairs(const int32_t *pairs_host, int64_t n_pairs, int64_t n_patches)
{
    for (int64_t i = 0; i < n_pairs; i++) {
        const int32_t *q = pairs_host + i * 4;
        DLCO_CHECK(q[0] >= 0 && q[0] < n_patches && q[2] >= 0 && q[2] < n_patches, DLCO_ERR_INVALID, "dlco_desc: patch id out of range");
    }
}
struct PinnedStage {
    float *dist = nullptr; uint8_t *label = nullptr;
    PinnedStage(size_t floats, size_t labels)
    {
        DLCO_HIP(hipHostMalloc((void **)&dist, std::max<size_t>(floats, 1) * sizeof(float)));
        if (hipHostMalloc((void **)&label, std::max<size_t>(labels, 1)) != hipSuccess) { (void)hipHostFree(dist); throw Error(DLCO_ERR_HIP, "hipHostMalloc failed"); }
    }
    ~PinnedStage() { (void)hipHostFree(dist); (void)hipHostFree(label); }
};
}  // namespace

// comp-uprjdists' Distance / Label rows, handed to `sink` chunk_rows pairs at a time in row order: the reference
// writes a 128-row hyperslab per chunk and checks it (src/comp-uprjdists.cpp:298-349), so its host memory stays at one
// chunk; a whole-matrix buffer (16 GB at 500k x 8192) is only needed by callers that ask for one.
int dlco_desc_pair_dists_stream(dlco_desc_ctx *c, const uint8_t *patches_host, int64_t n_patches, const int32_t *pairs_host,
                                int64_t n_pairs, int64_t chunk_rows, dlco_desc_sink_fn sink, void *user)
{
    if (!c || !patches_host || !pairs_host || !sink || n_patches < 1 || n_pairs < 1 || chunk_rows < 1) return DLCO_ERR_INVALID;
    return guarded(c, [&] {
        const int F = c->nsel * kBins;
        check_pairs(pairs_host, n_pairs, n_patches);
        DLCO_HIP(hipSetDevice(c->device));
        DevBuf<float> table, dist;
        table.alloc((size_t)n_patches * F);
        compute(c, patches_host, n_patches, table.p, true, F);
        const int64_t pchunk = std::max<int64_t>(1, std::min<int64_t>(std::min(n_pairs, chunk_rows), ((int64_t)1 << 28) / F));
        DevBuf<int32_t> pairs;
        DevBuf<uint8_t> lab;
        pairs.alloc((size_t)pchunk * 4);
        lab.alloc((size_t)pchunk);
        dist.alloc((size_t)pchunk * F);
        PinnedStage st((size_t)pchunk * F, (size_t)pchunk);
        for (int64_t p0 = 0; p0 < n_pairs; p0 += pchunk) {
            const int64_t cnt = std::min(pchunk, n_pairs - p0);
            DLCO_HIP(hipMemcpyAsync(pairs.p, pairs_host + p0 * 4, (size_t)cnt * 4 * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
            desc_pair_diff(table.p, F, F, pairs.p, cnt, dist.p, lab.p, c->stream);
            DLCO_HIP(hipMemcpyAsync(st.dist, dist.p, (size_t)cnt * F * sizeof(float), hipMemcpyDeviceToHost, c->stream));
            DLCO_HIP(hipMemcpyAsync(st.label, lab.p, (size_t)cnt, hipMemcpyDeviceToHost, c->stream));
            sync(c);
            const int rc = sink(user, p0, cnt, st.dist, st.label);
            if (rc != 0) throw Error(DLCO_ERR_INVALID, "dlco_desc: the row sink failed with code " + std::to_string(rc));
        }
    });
}

int dlco_desc_pair_dists(dlco_desc_ctx *c, const uint8_t *patches_host, int64_t n_patches, const int32_t *pairs_host, int64_t n_pairs,
                         float *dist_host, uint8_t *label_host)
{
    if (!c || !dist_host) return DLCO_ERR_INVALID;
    CopySink k{dist_host, label_host, (int64_t)c->nsel * kBins};
    return dlco_desc_pair_dists_stream(c, patches_host, n_patches, pairs_host, n_pairs, (int64_t)1 << 40, copy_sink, &k);
}

// The output of comp-fulldists (src/comp-fulldists.cpp:285-369), the input of pr-learn: with ALL pooling-region
// filters set (rows = 8 * n_regions, one row per ring replica), Distance [n_pairs, n_regions] =
// sum over a region's 8 rows and 8 bins of (Desc2 - Desc1)^2, and Label.  A patch's full descriptor is
// rows * 8 floats (1.3 MB at 5120 regions), so the work goes by chunks of pairs: both patches of a chunk are
// transformed and pooled, reduced per pair, and dropped; `sink` receives the rows of every chunk in order.
int dlco_desc_full_dists_stream(dlco_desc_ctx *c, const uint8_t *patches_host, int64_t n_patches, const int32_t *pairs_host,
                                int64_t n_pairs, int64_t chunk_rows, dlco_desc_sink_fn sink, void *user)
{
    if (!c || !patches_host || !pairs_host || !sink || n_patches < 1 || n_pairs < 1 || chunk_rows < 1) return DLCO_ERR_INVALID;
    return guarded(c, [&] {
        DLCO_CHECK(c->nsel > 0 && c->nsel % 8 == 0, DLCO_ERR_INVALID, "dlco_desc_full_dists: the filter bank must hold 8 rows per pooling region");
        const int F = c->nsel * kBins, groups = c->nsel / 8;
        check_pairs(pairs_host, n_pairs, n_patches);
        DLCO_HIP(hipSetDevice(c->device));
        // pairs per chunk: 2 * chunk descriptors of F floats within ~2 GB, and within one transform chunk
        const int64_t pchunk = std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(kChunk / 2, chunk_rows), ((int64_t)1 << 29) / ((int64_t)2 * F)));
        DevBuf<float> desc, dist;
        desc.alloc((size_t)2 * pchunk * F);
        dist.alloc((size_t)pchunk * groups);
        std::vector<uint8_t> stage((size_t)2 * pchunk * kPix);
        PinnedStage st((size_t)pchunk * groups, (size_t)pchunk);
        double ms_total = 0.0;
        for (int64_t p0 = 0; p0 < n_pairs; p0 += pchunk) {
            const int64_t cnt = std::min(pchunk, n_pairs - p0);
            for (int64_t i = 0; i < cnt; i++) {
                const int32_t *q = pairs_host + (p0 + i) * 4;
                std::memcpy(stage.data() + (size_t)i * kPix, patches_host + (size_t)q[0] * kPix, kPix);
                std::memcpy(stage.data() + (size_t)(cnt + i) * kPix, patches_host + (size_t)q[2] * kPix, kPix);
                st.label[i] = (q[1] == q[3]) ? 1 : 0;                              // :268-272 of both tools
            }
            compute(c, stage.data(), 2 * cnt, desc.p, true, F);
            ms_total += c->last_ms;
            desc_full_dist(desc.p, F, groups, (int)cnt, dist.p, c->stream);
            DLCO_HIP(hipMemcpyAsync(st.dist, dist.p, (size_t)cnt * groups * sizeof(float), hipMemcpyDeviceToHost, c->stream));
            sync(c);
            const int rc = sink(user, p0, cnt, st.dist, st.label);
            if (rc != 0) throw Error(DLCO_ERR_INVALID, "dlco_desc: the row sink failed with code " + std::to_string(rc));
        }
        c->last_ms = ms_total;
    });
}

int dlco_desc_full_dists(dlco_desc_ctx *c, const uint8_t *patches_host, int64_t n_patches, const int32_t *pairs_host, int64_t n_pairs,
                         float *dist_host, uint8_t *label_host)
{
    if (!c || !dist_host) return DLCO_ERR_INVALID;
    CopySink k{dist_host, label_host, (int64_t)c->nsel / 8};
    return dlco_desc_full_dists_stream(c, patches_host, n_patches, pairs_host, n_pairs, (int64_t)1 << 40, copy_sink, &k);
}

// milliseconds the transform + pooling kernels of the last dlco_desc_compute* call took (HIP events)
double dlco_desc_last_kernel_ms(const dlco_desc_ctx *c) { return c ? c->last_ms : 0.0; }

}  // extern "C"
