// kernels_jacobi.hip — symmetric eigendecomposition of the small Rayleigh-Ritz matrices of the
// subspace tracker (n = block rows, tens to a few hundred) on ONE workgroup, for gfx950.
//
// One-sided (Hestenes) Jacobi on G = T + sigma*I, sigma from a Gershgorin bound so that G is
// positive definite: the method orthogonalises the columns of G by plane rotations; at
// convergence column j equals lambda'_j * v_j, so the eigenvectors are the normalised columns
// and no separate rotation accumulator is kept (half the LDS traffic and footprint).
//   * a column pair is owned by a 16-lane DPP row; its three dot products are reduced with
//     four DPP adds (no LDS permutes); 64 pairs are in flight per round;
//   * columns are stored padded to a multiple of 4 and moved with 16-byte LDS accesses;
//   * columns live in LDS up to n = 192, in an L2-resident workspace beyond.
#include "dlco_internal.hpp"

namespace dlco {

namespace {

constexpr int JT = 1024;
constexpr int JW = JT / 64;
constexpr int JACOBI_LDS_MAX_N = 192;

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int CTRL>
__device__ __forceinline__ float dpp_f(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float row16_sum(float v)
{
    v += dpp_f<0xB1>(v);     // quad_perm [1,0,3,2]
    v += dpp_f<0x4E>(v);     // quad_perm [2,3,0,1]
    v += dpp_f<0x141>(v);    // row_half_mirror
    v += dpp_f<0x140>(v);    // row_mirror
    return v;
}
__device__ __forceinline__ float wsum(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wmax(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// G: column-major, column j at G + j*ldc (ldc multiple of 4, entries [n, ldc) are zero)
__device__ void jacobi_body(float *G, int ldc, const float *T, long ldt, int n, float *evals, float *Vout, long ldv,
                            float *scratch /* >= 4n floats, global */, int *sweeps_out, float *red /* LDS, JW+4 */)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    // ---- shift: sigma = 1.01 * max_i sum_j |T_ij| + tiny  (Gershgorin) ------------------------
    float rmax = 0.f;
    for (int i = wave; i < n; i += JW) {
        float s = 0.f;
        for (int j = lane; j < n; j += 64) s += fabsf(T[(long)i * ldt + j]);
        s = wsum(s);
        rmax = fmaxf(rmax, s);
    }
    if (lane == 0) red[wave] = rmax;
    __syncthreads();
    float sigma = 0.f;
    for (int w = 0; w < JW; w++) sigma = fmaxf(sigma, red[w]);
    sigma = 1.01f * sigma + 1e-30f;
    __syncthreads();

    // ---- init: G = sym(T) + sigma I, zero padding ---------------------------------------------------
    for (int e = tid; e < n * ldc; e += JT) {
        const int j = e / ldc, i = e % ldc;
        float v = 0.f;
        if (i < n) v = 0.5f * (T[(long)i * ldt + j] + T[(long)j * ldt + i]) + (i == j ? sigma : 0.f);
        G[e] = v;
    }
    __syncthreads();

    const int ne = n + (n & 1);           // even player count; index n (if present) is a bye
    const int half = ne / 2;
    const float tol = 3e-6f;
    const int grp = tid >> 4, sub = tid & 15;
    int sweep = 0;
    for (; sweep < 40; sweep++) {
        float off_max = 0.f;
        for (int r = 0; r < ne - 1; r++) {
            for (int k = grp; k < half; k += JT / 16) {
                int p, q;
                if (k == 0) { p = ne - 1; q = r; }
                else { p = (r + k) % (ne - 1); q = (r - k + (ne - 1)) % (ne - 1); }
                const bool live = (p < n && q < n);       // uniform over the 16-lane row
                if (p > q) { const int t = p; p = q; q = t; }
                f32x4 *gp = reinterpret_cast<f32x4 *>(G + (long)(live ? p : 0) * ldc);
                f32x4 *gq = reinterpret_cast<f32x4 *>(G + (long)(live ? q : 0) * ldc);
                const int nch = ldc >> 2;
                float a = 0.f, b = 0.f, c = 0.f;
                if (nch <= 32) {
                    // n <= 128: a lane owns at most two 16-byte chunks of each column and keeps them
                    // in registers between the dot products and the rotation (one LDS read, one write)
                    const bool h0 = live && sub < nch, h1 = live && sub + 16 < nch;
                    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
                    const f32x4 x0 = h0 ? gp[sub] : z4, y0 = h0 ? gq[sub] : z4;
                    const f32x4 x1 = h1 ? gp[sub + 16] : z4, y1 = h1 ? gq[sub + 16] : z4;
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        a += x0[e] * x0[e] + x1[e] * x1[e];
                        b += y0[e] * y0[e] + y1[e] * y1[e];
                        c += x0[e] * y0[e] + x1[e] * y1[e];
                    }
                    a = row16_sum(a); b = row16_sum(b); c = row16_sum(c);
                    const float denom = sqrtf(a * b);
                    const float off = (live && denom > 0.f) ? fabsf(c) / denom : 0.f;
                    off_max = fmaxf(off_max, off);
                    if (off > tol) {
                        const float zeta = (b - a) / (2.f * c);
                        const float t = copysignf(1.f, zeta) / (fabsf(zeta) + sqrtf(1.f + zeta * zeta));
                        const float cs = 1.f / sqrtf(1.f + t * t), sn = cs * t;
                        if (h0) { gp[sub] = cs * x0 - sn * y0; gq[sub] = sn * x0 + cs * y0; }
                        if (h1) { gp[sub + 16] = cs * x1 - sn * y1; gq[sub + 16] = sn * x1 + cs * y1; }
                    }
                    continue;
                }
                if (live)
                    for (int ch = sub; ch < nch; ch += 16) {
                        const f32x4 x = gp[ch], y = gq[ch];
                        a += x[0] * x[0] + x[1] * x[1] + x[2] * x[2] + x[3] * x[3];
                        b += y[0] * y[0] + y[1] * y[1] + y[2] * y[2] + y[3] * y[3];
                        c += x[0] * y[0] + x[1] * y[1] + x[2] * y[2] + x[3] * y[3];
                    }
                a = row16_sum(a); b = row16_sum(b); c = row16_sum(c);
                const float denom = sqrtf(a * b);
                const float off = (live && denom > 0.f) ? fabsf(c) / denom : 0.f;
                off_max = fmaxf(off_max, off);
                if (off > tol) {
                    const float zeta = (b - a) / (2.f * c);
                    const float t = copysignf(1.f, zeta) / (fabsf(zeta) + sqrtf(1.f + zeta * zeta));
                    const float cs = 1.f / sqrtf(1.f + t * t), sn = cs * t;
                    for (int ch = sub; ch < nch; ch += 16) {
                        const f32x4 x = gp[ch], y = gq[ch];
                        gp[ch] = cs * x - sn * y;
                        gq[ch] = sn * x + cs * y;
                    }
                }
            }
            __syncthreads();
        }
        off_max = wmax(off_max);
        if (lane == 0) red[wave] = off_max;
        __syncthreads();
        float m = 0.f;
        for (int w = 0; w < JW; w++) m = fmaxf(m, red[w]);
        __syncthreads();
        // every pair whose cosine exceeded tol was rotated to exact orthogonality in this sweep and
        // the later rotations of the sweep disturb it only to second order: a sweep that started
        // below 1e-4 ends below ~n*1e-8, so no separate verification sweep is needed
        if (m <= 1e-4f) { sweep++; break; }
    }
    if (tid == 0 && sweeps_out) *sweeps_out = sweep;

    // ---- eigenvalues lambda_j = |g_j| - sigma, eigenvectors v_j = g_j / |g_j| -----------------------
    float *lam = scratch;                 // [n]
    float *inv = scratch + n;             // [n]
    int *rank = reinterpret_cast<int *>(scratch + 2 * n);
    for (int j = wave; j < n; j += JW) {
        float d = 0.f;
        for (int i = lane; i < n; i += 64) { const float v = G[(long)j * ldc + i]; d += v * v; }
        d = wsum(d);
        if (lane == 0) { const float nrm = sqrtf(d); lam[j] = nrm - sigma; inv[j] = nrm > 0.f ? 1.f / nrm : 0.f; }
    }
    __syncthreads();
    for (int j = tid; j < n; j += JT) {
        const float me = lam[j];
        int rk = 0;
        for (int k = 0; k < n; k++) {
            const float o = lam[k];
            rk += (o > me || (o == me && k < j)) ? 1 : 0;
        }
        rank[j] = rk;
        evals[rk] = me;
    }
    __syncthreads();
    for (int e = tid; e < n * n; e += JT) {
        const int j = e / n, i = e % n;
        Vout[(long)i * ldv + rank[j]] = G[(long)j * ldc + i] * inv[j];
    }
}

__global__ __launch_bounds__(JT) void jacobi_lds_kernel(const float *T, long ldt, int n, int ldc, float *evals, float *Vout,
                                                        long ldv, float *scratch, int *sweeps_out)
{
    extern __shared__ __attribute__((aligned(16))) float sh[];
    float *G = sh, *red = sh + (size_t)n * ldc;
    jacobi_body(G, ldc, T, ldt, n, evals, Vout, ldv, scratch, sweeps_out, red);
}

__global__ __launch_bounds__(JT) void jacobi_gmem_kernel(const float *T, long ldt, int n, int ldc, float *evals, float *Vout,
                                                         long ldv, float *work, int *sweeps_out)
{
    __shared__ float red[JW + 4];
    float *G = work, *scratch = work + (size_t)n * ldc;
    jacobi_body(G, ldc, T, ldt, n, evals, Vout, ldv, scratch, sweeps_out, red);
}

inline int col_stride(int n) { return (n + 3) & ~3; }

}  // namespace

size_t jacobi_work_floats(int n) { return (size_t)n * col_stride(n) + 4 * (size_t)n + 64; }

void jacobi_eigh(const float *T, long ldt, int n, float *evals, float *V, long ldv, float *work, int *sweeps_out,
                 hipStream_t s)
{
    DLCO_CHECK(n >= 1 && n <= 4096, -2, "jacobi_eigh: n out of range");
    const int ldc = col_stride(n);
    if (n <= JACOBI_LDS_MAX_N) {
        const size_t lds = ((size_t)n * ldc + JW + 4) * sizeof(float);
        static bool attr_set = false;
        if (!attr_set) {
            DLCO_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(jacobi_lds_kernel),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024));
            attr_set = true;
        }
        hipLaunchKernelGGL(jacobi_lds_kernel, dim3(1), dim3(JT), lds, s, T, ldt, n, ldc, evals, V, ldv, work, sweeps_out);
    } else {
        hipLaunchKernelGGL(jacobi_gmem_kernel, dim3(1), dim3(JT), 0, s, T, ldt, n, ldc, evals, V, ldv, work, sweeps_out);
    }
    DLCO_HIP(hipGetLastError());
}

}  // namespace dlco
