// kernels_pr.hip — pr-learn: the pooling-region stage of the pipeline on gfx950.
//
// The reference's loop (src/pr-learn.cpp:302-329) is a strictly sequential stochastic method: one
// (positive, negative) row pair per iteration, a 5120-term dot product whose SIGN decides the update,
// and three element-wise vector updates; iteration t+1 needs the w of iteration t.  There is nothing to
// batch, so the whole window of iterations between two log steps (100 000 in the reference) runs inside
// ONE launch of one workgroup: w and dfAvg live in registers (20 floats per thread at F = 5120), the two
// rows of the next iteration are prefetched while the current one is reduced, and an iteration costs one
// workgroup barrier.  Arithmetic follows the reference operation for operation (double accumulation of
// the dot product like cv::gemm, the float scale factors of the MatExpr / scaleAdd calls, products and
// sums NOT fused), so the trajectory is the oracle's, bit for bit, as long as no dot product lands
// within one double-rounding of a float boundary.
#include "dlco_internal.hpp"

namespace dlco {

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int PT = 256;           // threads of the sequential kernel
constexpr int PW = PT / 64;

__device__ __forceinline__ double wave_sum_f64(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// CH = 16-byte chunks of w per thread (F <= 1024 * CH)
template <int CH>
__global__ __launch_bounds__(PT) void pr_steps_kernel(const float *D, long ld, const int32_t *pos_rows, const int32_t *neg_rows,
                                                      unsigned n, unsigned t0, float mu, float gamma, int F, float *w_io,
                                                      float *df_io, float *last_f)
{
#pragma clang fp contract(off)
    __shared__ double part[2][PW];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nch = F >> 2;
    f32x4 w[CH], df[CH], xp[CH], xn[CH];
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < CH; u++) {
        const int c = tid + PT * u;
        w[u] = c < nch ? reinterpret_cast<const f32x4 *>(w_io)[c] : z4;
        df[u] = c < nch ? reinterpret_cast<const f32x4 *>(df_io)[c] : z4;
    }
    auto fetch = [&](unsigned s) {
        const f32x4 *rp = reinterpret_cast<const f32x4 *>(D + (long)pos_rows[s] * ld);
        const f32x4 *rn = reinterpret_cast<const f32x4 *>(D + (long)neg_rows[s] * ld);
#pragma unroll
        for (int u = 0; u < CH; u++) {
            const int c = tid + PT * u;
            xp[u] = c < nch ? rp[c] : z4;
            xn[u] = c < nch ? rn[c] : z4;
        }
    };
    if (n > 0) fetch(0);
    float f = 0.f;
    for (unsigned s = 0; s < n; s++) {
        const unsigned t = t0 + s;
        f32x4 d[CH];
        double acc = 0.0;
#pragma unroll
        for (int u = 0; u < CH; u++) {
            d[u] = xp[u] - xn[u];                              // subtract(), :312-314
#pragma unroll
            for (int e = 0; e < 4; e++) acc += (double)w[u][e] * (double)d[u][e];   // gemm in double, :319
        }
        if (s + 1 < n) fetch(s + 1);                           // the next iteration's rows travel meanwhile
        acc = wave_sum_f64(acc);
        if (lane == 0) part[s & 1][wave] = acc;
        __syncthreads();
        double tot = 0.0;
#pragma unroll
        for (int q = 0; q < PW; q++) tot += part[s & 1][q];
        f = (float)tot;
        const float sa = (float)((double)t * (1.0 / ((double)t + 1.0)));            // :322
        const float al = (float)(1.0 / ((double)t + 1.0));                           // :325
        const double a = -sqrt((double)t + 1.0) / (double)gamma;                     // :328
        const float fa = (float)a, fb = (float)((double)mu * a);
        const bool upd = f > -1.0f;                                                  // :324
#pragma unroll
        for (int u = 0; u < CH; u++)
#pragma unroll
            for (int e = 0; e < 4; e++) {
                float v = df[u][e] * sa;
                if (upd) { const float p = d[u][e] * al; v = p + v; }
                df[u][e] = v;
                const float q = v * fa;
                const float r = q + fb;
                w[u][e] = r > 0.0f ? r : 0.0f;                                       // :329
            }
    }
#pragma unroll
    for (int u = 0; u < CH; u++) {
        const int c = tid + PT * u;
        if (c < nch) {
            reinterpret_cast<f32x4 *>(w_io)[c] = w[u];
            reinterpret_cast<f32x4 *>(df_io)[c] = df[u];
        }
    }
    if (tid == 0 && last_f) *last_f = f;
}

// out[i] = w . D[row(i)]; one wave per row.  DOUBLE = accumulate in double (cv::gemm on the CPU,
// src/misc.cpp:226), else in float (cuda::gemm, src/pr-learn.cpp:343-344: order unspecified).
template <bool DOUBLE>
__global__ __launch_bounds__(256) void pr_gemv_kernel(const float *D, long ld, const int32_t *ids, int n, const float *w, int F, float *out)
{
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= n) return;
    const f32x4 *x = reinterpret_cast<const f32x4 *>(D + (long)(ids ? ids[row] : row) * ld);
    const f32x4 *w4 = reinterpret_cast<const f32x4 *>(w);
    if (DOUBLE) {
        double s = 0.0;
        for (int c = lane; c < F / 4; c += 64) {
            const f32x4 a = x[c], b = w4[c];
            s += (double)a[0] * b[0] + (double)a[1] * b[1] + (double)a[2] * b[2] + (double)a[3] * b[3];
        }
        s = wave_sum_f64(s);
        if (lane == 0) out[row] = (float)s;
    } else {
        float s = 0.f;
        for (int c = lane; c < F / 4; c += 64) {
            const f32x4 a = x[c], b = w4[c];
            s += a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3];
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        if (lane == 0) out[row] = s;
    }
}

}  // namespace

bool pr_steps(const float *D, long ld, const int32_t *pos_rows, const int32_t *neg_rows, unsigned n, unsigned t0, float mu, float gamma,
              int F, float *w_io, float *df_io, float *last_f, hipStream_t s)
{
    if (F % 4 != 0 || F > 1024 * 8 || ld % 4 != 0) return false;
    const int ch = (F / 4 + PT - 1) / PT;
#define DLCO_PR_LAUNCH(C) hipLaunchKernelGGL(pr_steps_kernel<C>, dim3(1), dim3(PT), 0, s, D, ld, pos_rows, neg_rows, n, t0, mu, gamma, F, w_io, df_io, last_f)
    if (ch <= 1) DLCO_PR_LAUNCH(1);
    else if (ch <= 2) DLCO_PR_LAUNCH(2);
    else if (ch <= 4) DLCO_PR_LAUNCH(4);
    else if (ch <= 5) DLCO_PR_LAUNCH(5);
    else DLCO_PR_LAUNCH(8);
#undef DLCO_PR_LAUNCH
    DLCO_HIP(hipGetLastError());
    return true;
}

void pr_gemv(const float *D, long ld, const int32_t *ids, int n, const float *w, int F, float *out, bool in_double, hipStream_t s)
{
    if (n <= 0) return;
    if (in_double) hipLaunchKernelGGL(pr_gemv_kernel<true>, dim3((n + 3) / 4), dim3(256), 0, s, D, ld, ids, n, w, F, out);
    else hipLaunchKernelGGL(pr_gemv_kernel<false>, dim3((n + 3) / 4), dim3(256), 0, s, D, ld, ids, n, w, F, out);
    DLCO_HIP(hipGetLastError());
}

}  // namespace dlco
