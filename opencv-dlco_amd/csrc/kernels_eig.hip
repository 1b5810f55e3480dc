// kernels_eig.hip — small dense symmetric eigenproblems and the vector kernels of the
// subspace tracker (E1/E2 of the pj-learn step, src/pj-learn.cpp:434-490), for gfx950.
//
// jacobi_eigh: one-sided (Hestenes) Jacobi on ONE workgroup of 16 waves.  The input is
// shifted to be positive definite (T + sigma*I, sigma from a Gershgorin bound) so that the
// singular vectors the one-sided method finds are the eigenvectors; each wave owns one
// column pair at a time, columns live in LDS (n <= 128) or in an L2-resident workspace.
#include "dlco_internal.hpp"

namespace dlco {

namespace {

constexpr int JT = 1024;          // threads of the Jacobi workgroup
constexpr int JW = JT / 64;       // waves
constexpr int JACOBI_LDS_MAX_N = 128;

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

// G, V: column-major n x n (column j at [j*n, j*n+n)).
__device__ void jacobi_body(float *G, float *V, const float *T, long ldt, int n, float *evals, float *Vout, long ldv,
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

    // ---- init: G = T + sigma I (column j = row j by symmetry; symmetrised on the fly), V = I ----
    for (int e = tid; e < n * n; e += JT) {
        const int j = e / n, i = e % n;
        const float t = 0.5f * (T[(long)i * ldt + j] + T[(long)j * ldt + i]);
        G[e] = t + (i == j ? sigma : 0.f);
        V[e] = (i == j) ? 1.f : 0.f;
    }
    __syncthreads();

    const int ne = n + (n & 1);           // even player count; index n (if present) is a bye
    const int half = ne / 2;
    const float tol = 3e-6f;
    int sweep = 0;
    for (; sweep < 40; sweep++) {
        float off_max = 0.f;
        for (int r = 0; r < ne - 1; r++) {
            for (int k = wave; k < half; k += JW) {
                int p, q;
                if (k == 0) { p = ne - 1; q = r; }
                else { p = (r + k) % (ne - 1); q = (r - k + (ne - 1)) % (ne - 1); }
                if (p >= n || q >= n) continue;
                if (p > q) { const int t = p; p = q; q = t; }
                float *gp = G + (long)p * n, *gq = G + (long)q * n;
                float a = 0.f, b = 0.f, c = 0.f;
                for (int i = lane; i < n; i += 64) {
                    const float x = gp[i], y = gq[i];
                    a += x * x; b += y * y; c += x * y;
                }
                a = wsum(a); b = wsum(b); c = wsum(c);
                const float denom = sqrtf(a * b);
                const float off = denom > 0.f ? fabsf(c) / denom : 0.f;
                off_max = fmaxf(off_max, off);
                if (off > tol) {
                    const float zeta = (b - a) / (2.f * c);
                    const float t = copysignf(1.f, zeta) / (fabsf(zeta) + sqrtf(1.f + zeta * zeta));
                    const float cs = 1.f / sqrtf(1.f + t * t), sn = cs * t;
                    float *vp = V + (long)p * n, *vq = V + (long)q * n;
                    for (int i = lane; i < n; i += 64) {
                        const float x = gp[i], y = gq[i];
                        gp[i] = cs * x - sn * y;
                        gq[i] = sn * x + cs * y;
                        const float u = vp[i], w = vq[i];
                        vp[i] = cs * u - sn * w;
                        vq[i] = sn * u + cs * w;
                    }
                }
            }
            __syncthreads();
        }
        if (lane == 0) red[wave] = off_max;
        __syncthreads();
        float m = 0.f;
        for (int w = 0; w < JW; w++) m = fmaxf(m, red[w]);
        __syncthreads();
        if (m <= tol) { sweep++; break; }
    }
    if (tid == 0 && sweeps_out) *sweeps_out = sweep;

    // ---- eigenvalues: lambda_j = v_j . g_j - sigma  (g_j = (T + sigma I) v_j) --------------------
    float *lam = scratch;                 // [n]
    int *rank = reinterpret_cast<int *>(scratch + n);
    for (int j = wave; j < n; j += JW) {
        float d = 0.f, vv = 0.f;
        for (int i = lane; i < n; i += 64) { d += V[(long)j * n + i] * G[(long)j * n + i]; vv += V[(long)j * n + i] * V[(long)j * n + i]; }
        d = wsum(d); vv = wsum(vv);
        if (lane == 0) lam[j] = d / vv - sigma;
    }
    __syncthreads();
    // ---- sort descending (rank by counting; ties by index) and write out ---------------------------
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
        Vout[(long)i * ldv + rank[j]] = V[e];
    }
}

__global__ __launch_bounds__(JT) void jacobi_lds_kernel(const float *T, long ldt, int n, float *evals, float *Vout,
                                                        long ldv, float *scratch, int *sweeps_out)
{
    extern __shared__ __attribute__((aligned(16))) float sh[];
    float *G = sh, *V = sh + n * n, *red = sh + 2 * n * n;
    jacobi_body(G, V, T, ldt, n, evals, Vout, ldv, scratch, sweeps_out, red);
}

__global__ __launch_bounds__(JT) void jacobi_gmem_kernel(const float *T, long ldt, int n, float *evals, float *Vout,
                                                         long ldv, float *work, int *sweeps_out)
{
    __shared__ float red[JW + 4];
    float *G = work, *V = work + (long)n * n, *scratch = work + 2L * n * n;
    jacobi_body(G, V, T, ldt, n, evals, Vout, ldv, scratch, sweeps_out, red);
}

__global__ __launch_bounds__(256) void residual_kernel(const float *X, const float *Y, long ld, const float *theta,
                                                       int m, int F, float *res)
{
    __shared__ float part[4];
    const int i = blockIdx.x;
    if (i >= m) return;
    const float th = theta[i];
    float s = 0.f;
    for (int f = threadIdx.x; f < F; f += blockDim.x) {
        const float d = Y[(long)i * ld + f] - th * X[(long)i * ld + f];
        s += d * d;
    }
    s = wsum(s);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) res[i] = sqrtf(part[0] + part[1] + part[2] + part[3]);
}

// rows whose norm is below `min_norm` are zeroed (dependent directions), the rest scaled to unit norm
__global__ __launch_bounds__(256) void row_normalize_kernel(float *X, long ld, int m, int F, float min_norm)
{
    __shared__ float part[4];
    __shared__ float inv;
    const int i = blockIdx.x;
    if (i >= m) return;
    float s = 0.f;
    for (int f = threadIdx.x; f < F; f += blockDim.x) { const float v = X[(long)i * ld + f]; s += v * v; }
    s = wsum(s);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) { const float n2 = part[0] + part[1] + part[2] + part[3]; inv = (n2 > 0.f && n2 >= min_norm * min_norm) ? 1.f / sqrtf(n2) : 0.f; }
    __syncthreads();
    const float sc = inv;
    for (int f = threadIdx.x; f < F; f += blockDim.x) X[(long)i * ld + f] *= sc;
}

__global__ __launch_bounds__(256) void symv_kernel(const float *H, long ld, int F, const float *x, float *y)
{
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= F) return;
    float s = 0.f;
    for (int f = lane; f < F; f += 64) s += H[(long)row * ld + f] * x[f];
    s = wsum(s);
    if (lane == 0) y[row] = s;
}

__global__ __launch_bounds__(256) void whitener_kernel(const float *evals, const float *V, long ldv, int n,
                                                       float rel_thresh, float *Cw, long ldcw, int *k_out)
{
    __shared__ int kk;
    if (threadIdx.x == 0) {
        const float lmax = evals[0];
        int k = 0;
        while (k < n && evals[k] > rel_thresh * lmax && evals[k] > 0.f) k++;
        kk = k;
        *k_out = k;
    }
    __syncthreads();
    const int k = kk;
    for (int e = threadIdx.x; e < n * n; e += blockDim.x) {
        const int i = e / n, j = e % n;
        Cw[(long)i * ldcw + j] = (j < k) ? V[(long)i * ldv + j] * rsqrtf(evals[j]) : 0.f;
    }
}

}  // namespace

size_t jacobi_work_floats(int n) { return 2 * (size_t)n * n + 4 * (size_t)n + 64; }

void jacobi_eigh(const float *T, long ldt, int n, float *evals, float *V, long ldv, float *work, int *sweeps_out,
                 hipStream_t s)
{
    DLCO_CHECK(n >= 1 && n <= 2048, -2, "jacobi_eigh: n out of range");
    if (n <= JACOBI_LDS_MAX_N) {
        const size_t lds = (2 * (size_t)n * n + JW + 4) * sizeof(float);
        static bool attr_set = false;
        if (!attr_set) {
            DLCO_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(jacobi_lds_kernel),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024));
            attr_set = true;
        }
        hipLaunchKernelGGL(jacobi_lds_kernel, dim3(1), dim3(JT), lds, s, T, ldt, n, evals, V, ldv, work, sweeps_out);
    } else {
        hipLaunchKernelGGL(jacobi_gmem_kernel, dim3(1), dim3(JT), 0, s, T, ldt, n, evals, V, ldv, work, sweeps_out);
    }
    DLCO_HIP(hipGetLastError());
}

void residual_norms(const float *X, const float *Y, long ld, const float *theta, int m, int F, float *res,
                    hipStream_t s)
{
    if (m <= 0) return;
    hipLaunchKernelGGL(residual_kernel, dim3(m), dim3(256), 0, s, X, Y, ld, theta, m, F, res);
    DLCO_HIP(hipGetLastError());
}

void row_normalize(float *X, long ld, int m, int F, hipStream_t s, float min_norm)
{
    if (m <= 0) return;
    hipLaunchKernelGGL(row_normalize_kernel, dim3(m), dim3(256), 0, s, X, ld, m, F, min_norm);
    DLCO_HIP(hipGetLastError());
}

void symv(const float *H, long ld, int F, const float *x, float *y, hipStream_t s)
{
    hipLaunchKernelGGL(symv_kernel, dim3((F + 3) / 4), dim3(256), 0, s, H, ld, F, x, y);
    DLCO_HIP(hipGetLastError());
}

void build_whitener(const float *evals, const float *V, long ldv, int n, float rel_thresh, float *Cw, long ldcw,
                    int *k_out_dev, hipStream_t s)
{
    hipLaunchKernelGGL(whitener_kernel, dim3(1), dim3(256), 0, s, evals, V, ldv, n, rel_thresh, Cw, ldcw, k_out_dev);
    DLCO_HIP(hipGetLastError());
}

}  // namespace dlco
