// kernels_jacobi.hip — symmetric eigendecomposition of the small Rayleigh-Ritz matrices of the
// subspace tracker (n = block rows, tens to a few hundred) on ONE workgroup, for gfx950.
//
// One-sided (Hestenes) Jacobi on G = T + sigma*I, sigma from a Gershgorin bound so that G is
// positive definite: the method orthogonalises the columns of G by plane rotations; at
// convergence column j equals lambda'_j * v_j, so the eigenvectors are the normalised columns
// and no separate rotation accumulator is kept (half the LDS traffic and footprint).
// The kernel is bound by VALU issue on its single CU, so the round is kept short:
//   * a column pair is owned by 8 lanes (half a DPP row): 64 pairs per pass on 8 waves, the
//     scalar rotation arithmetic is replicated 8 times instead of 64;
//   * only the cross product x.y is reduced per pair (three DPP adds); the squared norms live
//     in an LDS vector and follow the rotations analytically (|x'|^2 = |x|^2 - t x.y,
//     |y'|^2 = |y|^2 + t x.y), refreshed exactly at the start of every sweep;
//   * v_rcp / v_rsq / v_sqrt (1 ulp) instead of the IEEE division and square-root sequences:
//     a rotation only has to be orthogonal to rounding, its angle is re-measured every sweep;
//   * columns are padded to a multiple of 4 and moved with 16-byte LDS accesses; up to
//     n = 128 a lane keeps its chunks in registers between the dot product and the rotation;
//   * columns live in LDS up to n = 192, in an L2-resident workspace beyond.
#include "dlco_internal.hpp"

namespace dlco {

namespace {

constexpr int JT = 512;
constexpr int JW = JT / 64;
constexpr int LP = 8;                 // lanes per column pair
constexpr int NG = JT / LP;           // pairs in flight
constexpr int JACOBI_LDS_MAX_N = 192;

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int CTRL>
__device__ __forceinline__ float dpp_f(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float row8_sum(float v)
{
    v += dpp_f<0xB1>(v);     // quad_perm [1,0,3,2]
    v += dpp_f<0x4E>(v);     // quad_perm [2,3,0,1]
    v += dpp_f<0x141>(v);    // row_half_mirror: lane i <-> 7 - i inside each group of 8
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
__device__ __forceinline__ float dot4(const f32x4 &x, const f32x4 &y) { return x[0] * y[0] + x[1] * y[1] + x[2] * y[2] + x[3] * y[3]; }

// rotation (cs, sn) that orthogonalises two columns with squared norms a, b and cross product c
__device__ __forceinline__ void rotation(float a, float b, float c, float &t, float &cs, float &sn)
{
    const float zeta = (b - a) * __builtin_amdgcn_rcpf(2.f * c);
    t = copysignf(__builtin_amdgcn_rcpf(fabsf(zeta) + __builtin_amdgcn_sqrtf(1.f + zeta * zeta)), zeta);
    cs = __builtin_amdgcn_rsqf(1.f + t * t);
    sn = cs * t;
}

// G: column-major, column j at G + j*ldc (ldc multiple of 4, entries [n, ldc) are zero)
__device__ void jacobi_body(float *G, int ldc, const float *T, long ldt, int n, float *evals, float *Vout, long ldv,
                            float *scratch /* >= 4n floats, global */, int *sweeps_out, float *red /* LDS, JW+4 */,
                            float *nrm /* n floats, LDS or global */, float stop_cos)
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
    const int grp = tid / LP, sub = tid % LP;
    const int nch = ldc >> 2;
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    int sweep = 0;
    for (; sweep < 40; sweep++) {
        // exact squared column norms at the start of the sweep
        for (int j = grp; j < n; j += NG) {
            const f32x4 *gj = reinterpret_cast<const f32x4 *>(G + (long)j * ldc);
            float s = 0.f;
            for (int ch = sub; ch < nch; ch += LP) { const f32x4 x = gj[ch]; s += dot4(x, x); }
            s = row8_sum(s);
            if (sub == 0) nrm[j] = s;
        }
        __syncthreads();
        float off_max = 0.f;
        for (int r = 0; r < ne - 1; r++) {
            for (int k = grp; k < half; k += NG) {
                int p, q;                                     // round-robin tournament, round r
                if (k == 0) { p = ne - 1; q = r; }
                else {
                    p = r + k; if (p >= ne - 1) p -= ne - 1;
                    q = r - k; if (q < 0) q += ne - 1;
                }
                const bool live = (p < n && q < n);           // uniform over the 8 lanes of the pair
                if (p > q) { const int t = p; p = q; q = t; }
                if (!live) { p = 0; q = 0; }
                f32x4 *gp = reinterpret_cast<f32x4 *>(G + (long)p * ldc);
                f32x4 *gq = reinterpret_cast<f32x4 *>(G + (long)q * ldc);
                const float a = nrm[p], b = nrm[q];
                const float ab = a * b;
                if (nch <= 4 * LP) {
                    // n <= 128: a lane owns at most four 16-byte chunks of each column and keeps them
                    // in registers between the dot product and the rotation (one LDS read, one write)
                    const bool h0 = live && sub < nch, h1 = live && sub + LP < nch, h2 = live && sub + 2 * LP < nch,
                               h3 = live && sub + 3 * LP < nch;
                    // the loads are unconditional (clamped chunk index) so that all eight are in flight
                    // together; chunks a lane does not own are replaced by zeros afterwards
                    const int k0 = min(sub, nch - 1), k1 = min(sub + LP, nch - 1), k2 = min(sub + 2 * LP, nch - 1),
                              k3 = min(sub + 3 * LP, nch - 1);
                    f32x4 x0 = gp[k0], y0 = gq[k0], x1 = gp[k1], y1 = gq[k1];
                    f32x4 x2 = gp[k2], y2 = gq[k2], x3 = gp[k3], y3 = gq[k3];
                    x0 = h0 ? x0 : z4; y0 = h0 ? y0 : z4; x1 = h1 ? x1 : z4; y1 = h1 ? y1 : z4;
                    x2 = h2 ? x2 : z4; y2 = h2 ? y2 : z4; x3 = h3 ? x3 : z4; y3 = h3 ? y3 : z4;
                    const float c = row8_sum((dot4(x0, y0) + dot4(x1, y1)) + (dot4(x2, y2) + dot4(x3, y3)));
                    const float off = (live && ab > 0.f) ? fabsf(c) * __builtin_amdgcn_rsqf(ab) : 0.f;
                    off_max = fmaxf(off_max, off);
                    if (off > tol) {
                        float t, cs, sn;
                        rotation(a, b, c, t, cs, sn);
                        if (h0) { gp[sub] = cs * x0 - sn * y0; gq[sub] = sn * x0 + cs * y0; }
                        if (h1) { gp[sub + LP] = cs * x1 - sn * y1; gq[sub + LP] = sn * x1 + cs * y1; }
                        if (h2) { gp[sub + 2 * LP] = cs * x2 - sn * y2; gq[sub + 2 * LP] = sn * x2 + cs * y2; }
                        if (h3) { gp[sub + 3 * LP] = cs * x3 - sn * y3; gq[sub + 3 * LP] = sn * x3 + cs * y3; }
                        if (sub == 0) { nrm[p] = a - t * c; nrm[q] = b + t * c; }
                    }
                    continue;
                }
                float c = 0.f;
                if (live)
                    for (int ch = sub; ch < nch; ch += LP) c += dot4(gp[ch], gq[ch]);
                c = row8_sum(c);
                const float off = (live && ab > 0.f) ? fabsf(c) * __builtin_amdgcn_rsqf(ab) : 0.f;
                off_max = fmaxf(off_max, off);
                if (off > tol) {
                    float t, cs, sn;
                    rotation(a, b, c, t, cs, sn);
                    for (int ch = sub; ch < nch; ch += LP) {
                        const f32x4 x = gp[ch], y = gq[ch];
                        gp[ch] = cs * x - sn * y;
                        gq[ch] = sn * x + cs * y;
                    }
                    if (sub == 0) { nrm[p] = a - t * c; nrm[q] = b + t * c; }
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
        // every pair whose cosine exceeded tol was rotated to orthogonality in this sweep and the
        // later rotations of the sweep disturb it only to second order: a sweep whose largest
        // cosine was m ends below ~n*m^2, so no separate verification sweep is run (the tracker
        // checks the residuals of the Ritz pairs it keeps anyway)
        if (m <= stop_cos) { sweep++; break; }
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
        if (lane == 0) { const float nr = sqrtf(d); lam[j] = nr - sigma; inv[j] = nr > 0.f ? 1.f / nr : 0.f; }
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
                                                        long ldv, float *scratch, int *sweeps_out, float stop_cos)
{
    extern __shared__ __attribute__((aligned(16))) float sh[];
    float *G = sh, *red = sh + (size_t)n * ldc, *nrm = red + JW + 4;
    jacobi_body(G, ldc, T, ldt, n, evals, Vout, ldv, scratch, sweeps_out, red, nrm, stop_cos);
}

__global__ __launch_bounds__(JT) void jacobi_gmem_kernel(const float *T, long ldt, int n, int ldc, float *evals, float *Vout,
                                                         long ldv, float *work, int *sweeps_out, float stop_cos)
{
    __shared__ float red[JW + 4];
    float *G = work, *scratch = work + (size_t)n * ldc;
    jacobi_body(G, ldc, T, ldt, n, evals, Vout, ldv, scratch, sweeps_out, red, scratch + 3 * (size_t)n, stop_cos);
}

inline int col_stride(int n) { return (n + 3) & ~3; }

}  // namespace

size_t jacobi_work_floats(int n) { return (size_t)n * col_stride(n) + 4 * (size_t)n + 64; }

void jacobi_eigh(const float *T, long ldt, int n, float *evals, float *V, long ldv, float *work, int *sweeps_out,
                 hipStream_t s)
{
    DLCO_CHECK(n >= 1 && n <= 4096, -2, "jacobi_eigh: n out of range");
    const int ldc = col_stride(n);
    // largest column cosine of a sweep below which the sweep is the last one (see the kernel)
    static const float stop_cos = std::getenv("DLCO_JACOBI_STOP") ? (float)std::atof(std::getenv("DLCO_JACOBI_STOP")) : 1e-3f;
    if (n <= JACOBI_LDS_MAX_N) {
        const size_t lds = ((size_t)n * ldc + JW + 4 + n + 4) * sizeof(float);
        static bool attr_set = false;
        if (!attr_set) {
            DLCO_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(jacobi_lds_kernel),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024));
            attr_set = true;
        }
        hipLaunchKernelGGL(jacobi_lds_kernel, dim3(1), dim3(JT), lds, s, T, ldt, n, ldc, evals, V, ldv, work, sweeps_out, stop_cos);
    } else {
        hipLaunchKernelGGL(jacobi_gmem_kernel, dim3(1), dim3(JT), 0, s, T, ldt, n, ldc, evals, V, ldv, work, sweeps_out, stop_cos);
    }
    DLCO_HIP(hipGetLastError());
}

}  // namespace dlco
