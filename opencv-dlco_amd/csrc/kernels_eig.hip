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
// sum over a 16-lane DPP row, result in every lane of the row
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
__device__ __forceinline__ float wmax(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
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

// Left-looking Cholesky of a Gram matrix M (n <= 64) on one wave: lane i owns row i.
// A row whose pivot drops below rel_thresh * M_jj lies (to fp32 accuracy) in the span of the
// rows before it: it is marked dead (L_jj = 1, rest of the column 0) and later zeroed.
__device__ __forceinline__ float lane_bcast(float v, int src_lane)
{
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src_lane));
}

// Lane i keeps row i of M and of L in registers (both loops fully unrolled, so every register
// index is static); L_jk of the pivot row reaches the other lanes through v_readlane.
__global__ __launch_bounds__(64) void chol64_kernel(const float *M, long ldm, int n, float rel_thresh, float *L, long ldl,
                                                   int *dead)
{
    const int i = threadIdx.x;
    float Mr[64], Lr[64];
#pragma unroll
    for (int k = 0; k < 64; k++) {
        Mr[k] = (i < n && k <= i && k < n) ? M[(long)i * ldm + k] : 0.f;     // lower triangle of the Gram matrix
        Lr[k] = 0.f;
    }
    float diag = 1.f;
#pragma unroll
    for (int k = 0; k < 64; k++) diag = (k == i) ? Mr[k] : diag;
#pragma unroll
    for (int j = 0; j < 64; j++) {
        if (j < n) {                                                           // wave-uniform
            float s = (i >= j) ? Mr[j] : 0.f;
#pragma unroll
            for (int k = 0; k < j; k++) s -= Lr[k] * lane_bcast(Lr[k], j);
            const float d = lane_bcast(s, j);
            const float mjj = lane_bcast(diag, j);
            const bool is_dead = !(d > rel_thresh * mjj) || !(mjj > 0.f);
            float v;
            if (is_dead) v = (i == j) ? 1.f : 0.f;
            else v = (i == j) ? sqrtf(d) : s * rsqrtf(d);
            Lr[j] = (i >= j && i < n) ? v : 0.f;
            if (i == j) dead[j] = is_dead ? 1 : 0;
        }
    }
    if (i < n) {
#pragma unroll
        for (int k = 0; k < 64; k++)
            if (k <= i) L[(long)i * ldl + k] = Lr[k];
    }
}

// Q = L^-1 Z for a panel of m <= 64 rows (forward substitution); thread f owns column f, so the
// whole solve is independent per column and coalesced.  Dead rows become zero.  In place is fine.
__global__ __launch_bounds__(256) void trsm64_kernel(const float *L, long ldl, const int *dead, int m, const float *Z,
                                                    float *Q, long ld, int F)
{
    // the solved rows of this column stay in registers (loops fully unrolled: static indices);
    // the coefficients L_ij are wave-uniform and come through the scalar cache
    const int f = blockIdx.x * 256 + threadIdx.x;
    if (f >= F) return;
    float q[64];
#pragma unroll
    for (int i = 0; i < 64; i++) {
        if (i < m) {                                                          // wave-uniform
            float acc = Z[(long)i * ld + f];
#pragma unroll
            for (int j = 0; j < i; j++) acc -= L[(long)i * ldl + j] * q[j];
            q[i] = dead[i] ? 0.f : acc / L[(long)i * ldl + i];
            Q[(long)i * ld + f] = q[i];
        } else {
            q[i] = 0.f;
        }
    }
}

}  // namespace

void chol_factor64(const float *M, long ldm, int n, float rel_thresh, float *L, long ldl, int *dead, hipStream_t s)
{
    DLCO_CHECK(n >= 1 && n <= 64, -2, "chol_factor64: n out of range");
    hipLaunchKernelGGL(chol64_kernel, dim3(1), dim3(64), 0, s, M, ldm, n, rel_thresh, L, ldl, dead);
    DLCO_HIP(hipGetLastError());
}

void trsm_rows64(const float *L, long ldl, const int *dead, int m, const float *Z, float *Q, long ld, int F, hipStream_t s)
{
    DLCO_CHECK(m >= 1 && m <= 64, -2, "trsm_rows64: m out of range");
    hipLaunchKernelGGL(trsm64_kernel, dim3((F + 255) / 256), dim3(256), 0, s, L, ldl, dead, m, Z, Q, ld, F);
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
