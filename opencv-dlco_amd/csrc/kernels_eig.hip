// kernels_eig.hip — vector kernels and the CholQR building block of the subspace tracker
// (E1/E2 of the pj-learn step, src/pj-learn.cpp:434-490), for gfx950.  The Rayleigh-Ritz
// eigensolver itself is in kernels_jacobi.hip.
#include "dlco_internal.hpp"

namespace dlco {

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float wsum(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
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
    if ((F & 3) == 0 && (ld & 3) == 0) {                  // 16-byte accesses (rows are 16-byte aligned)
        const f32x4 *y4 = reinterpret_cast<const f32x4 *>(Y + (long)i * ld), *x4 = reinterpret_cast<const f32x4 *>(X + (long)i * ld);
        for (int f = threadIdx.x; f < F / 4; f += blockDim.x) {
            const f32x4 d = y4[f] - th * x4[f];
            s += d[0] * d[0] + d[1] * d[1] + d[2] * d[2] + d[3] * d[3];
        }
    } else {
        for (int f = threadIdx.x; f < F; f += blockDim.x) {
            const float d = Y[(long)i * ld + f] - th * X[(long)i * ld + f];
            s += d * d;
        }
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
    const bool vec = (F & 3) == 0 && (ld & 3) == 0;       // 16-byte accesses (rows are 16-byte aligned)
    f32x4 *x4 = reinterpret_cast<f32x4 *>(X + (long)i * ld);
    if (vec) {
        for (int f = threadIdx.x; f < F / 4; f += blockDim.x) { const f32x4 v = x4[f]; s += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3]; }
    } else {
        for (int f = threadIdx.x; f < F; f += blockDim.x) { const float v = X[(long)i * ld + f]; s += v * v; }
    }
    s = wsum(s);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) { const float n2 = part[0] + part[1] + part[2] + part[3]; inv = (n2 > 0.f && n2 >= min_norm * min_norm) ? 1.f / sqrtf(n2) : 0.f; }
    __syncthreads();
    const float sc = inv;
    if (vec) {
        for (int f = threadIdx.x; f < F / 4; f += blockDim.x) x4[f] = x4[f] * sc;
    } else {
        for (int f = threadIdx.x; f < F; f += blockDim.x) X[(long)i * ld + f] *= sc;
    }
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

// CholQR building block: Gram matrix M (n <= 128, lower triangle read) -> Linv = L^-1 with
// M = L L^T, on one workgroup of 16 waves.
//
// Right-looking elimination without pivoting on the augmented matrix [M | I]: step j subtracts
// (A_ij / A_jj) * row j from every row i > j.  The left half turns into the Schur complements
// (pivot d_j = A_jj at step j), the right half into U^-1 of M = U D U^T, and L^-1 = D^-1/2 U^-1.
// Thread (i, g) keeps 32 consecutive entries of row i of the 256-wide augmented matrix in
// registers (g < 4: columns of M, g >= 4: columns of I); a wave is 64 rows of one column group,
// so groups that a step cannot touch (M columns <= j, U^-1 columns > j) and rows <= j retire as
// whole waves.  Pivot row and pivot column of the next step travel through double-buffered
// LDS vectors: one barrier per step.  A row whose pivot drops below rel_thresh * M_jj lies (to
// fp32 accuracy) in the span of the rows before it: it is marked dead, eliminated from nothing,
// and its row of Linv is zero.
constexpr int CI_T = 1024;

__global__ __launch_bounds__(CI_T) void chol_inv_kernel(const float *M, long ldm, int n, float rel_thresh, float *Linv,
                                                       long ldl, int *dead)
{
    __shared__ __attribute__((aligned(16))) float rowbuf[2][256];
    __shared__ float colbuf[2][128];
    __shared__ float diag0[128], dpiv[128];
    __shared__ int deadf[128];
    const int t = threadIdx.x, i = t & 127, g = t >> 7;
    const bool is_x = g >= 4;
    const int c0 = (g & 3) * 32;
    float v[32];
#pragma unroll
    for (int c = 0; c < 32; c++) {
        const int k = c0 + c;
        if (is_x) v[c] = (i == k) ? 1.f : 0.f;
        else if (i < n && k < n) v[c] = M[(long)max(i, k) * ldm + min(i, k)];
        else v[c] = (i == k) ? 1.f : 0.f;
    }
    if (t < 128) diag0[t] = t < n ? M[(long)t * ldm + t] : 1.f;
    if (i == 0) {
#pragma unroll
        for (int c = 0; c < 32; c++) rowbuf[0][g * 32 + c] = v[c];
    }
    if (g == 0) colbuf[0][i] = v[0];
    // Fast path (second pass of CholQR2): M = I + E with |E|_F <= 1e-3.  Then M^-1/2 = I - E/2 up
    // to 3/8 |E|^2 <= 4e-7, the symmetric orthogonaliser replaces the triangular one and the n
    // sequential elimination steps are skipped.  (Padding entries hold the identity: E = 0 there.)
    {
        float e2 = 0.f;
        if (!is_x) {
#pragma unroll
            for (int c = 0; c < 32; c++) { const float e = v[c] - ((i == c0 + c) ? 1.f : 0.f); e2 += e * e; }
        }
        e2 = wsum(e2);
        if ((t & 63) == 0) dpiv[t >> 6] = e2;
    }
    __syncthreads();
    {
        float tot = 0.f;
        for (int w = 0; w < CI_T / 64; w++) tot += dpiv[w];
        __syncthreads();                                     // dpiv is reused by the elimination below
        if (tot <= 1e-6f) {                                   // workgroup-uniform (false for NaN)
            if (!is_x && i < n) {
#pragma unroll
                for (int c = 0; c < 32; c++) {
                    const int k = c0 + c;
                    if (k < n) Linv[(long)i * ldl + k] = ((i == k) ? 1.5f : 0.f) - 0.5f * v[c];
                }
            }
            if (t < n) dead[t] = 0;
            return;
        }
    }

    // the step loop stays rolled: the body is executed once per step, so an unrolled copy of it
    // would run entirely out of cold instruction-cache lines
#pragma unroll 1
    for (int j = 0; j < n; j++) {
        const int buf = j & 1;
        const float piv = rowbuf[buf][j];
        const float d0 = diag0[j];
        const bool is_dead = !(piv > rel_thresh * d0) || !(d0 > 0.f);
        if (t == 0) { dpiv[j] = piv; deadf[j] = is_dead ? 1 : 0; }
        const bool grp_active = is_x ? (c0 <= j) : (c0 + 31 > j);            // wave-uniform
        if (!is_dead && grp_active && i > j && i < n) {
            const float f = colbuf[buf][i] / piv;
            const f32x4 *rp = reinterpret_cast<const f32x4 *>(&rowbuf[buf][g * 32]);
#pragma unroll
            for (int q = 0; q < 8; q++) {
                const f32x4 r = rp[q];
                v[4 * q + 0] -= f * r[0];
                v[4 * q + 1] -= f * r[1];
                v[4 * q + 2] -= f * r[2];
                v[4 * q + 3] -= f * r[3];
            }
        }
        if (j + 1 < n) {                                                      // publish pivot row / column of step j+1
            if (i == j + 1) {
                f32x4 *wp = reinterpret_cast<f32x4 *>(&rowbuf[buf ^ 1][g * 32]);
#pragma unroll
                for (int q = 0; q < 8; q++) {
                    f32x4 r = {v[4 * q + 0], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]};
                    wp[q] = r;
                }
            }
            if (g == ((j + 1) >> 5)) {                                        // the two waves that own column j+1 of M
                const int cn = (j + 1) & 31;
                float cv = v[0];
#pragma unroll
                for (int c = 1; c < 32; c++) cv = (c == cn) ? v[c] : cv;     // register select, no dynamic indexing
                colbuf[buf ^ 1][i] = cv;
            }
        }
        __syncthreads();
    }
    // Linv = D^-1/2 U^-1 (lower triangular); dead rows and the padding are zero
    if (is_x && i < n) {
        const bool dd = deadf[i] != 0;
        const float sc = dd ? 0.f : rsqrtf(dpiv[i]);
#pragma unroll
        for (int c = 0; c < 32; c++) {
            const int k = c0 + c;
            if (k < n) Linv[(long)i * ldl + k] = (k <= i) ? v[c] * sc : 0.f;
        }
    }
    if (t < n) dead[t] = deadf[t];
}

}  // namespace

void chol_inverse128(const float *M, long ldm, int n, float rel_thresh, float *Linv, long ldl, int *dead, hipStream_t s)
{
    DLCO_CHECK(n >= 1 && n <= CHOL_INV_MAX_N, -2, "chol_inverse128: n out of range");
    hipLaunchKernelGGL(chol_inv_kernel, dim3(1), dim3(CI_T), 0, s, M, ldm, n, rel_thresh, Linv, ldl, dead);
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

}  // namespace dlco
