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

#include <algorithm>

namespace dlco {

namespace {

constexpr int LP = 8;                 // lanes per column pair
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
// JT threads: JT / 8 pairs in flight (512: up to n = 128 in one pass per round; 1024: up to n = 256)
template <int JT, int CH>
__device__ void jacobi_body(float *G, int ldc, const float *T, long ldt, int n, float *evals, float *Vout, long ldv,
                            float *scratch /* >= 4n floats, global */, int *sweeps_out, float *red /* LDS, JW+4 */,
                            float *nrm /* n floats, LDS or global */, float stop_cos, float lam_cut)
{
    constexpr int JW = JT / 64, NG = JT / LP;
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
    // columns whose eigenvalue estimate |g| - sigma is below lam_cut are guard directions of the caller: a pair of
    // two such columns is rotated like any other but does not keep the sweeps going (squared norms are compared)
    const float cut2 = lam_cut + sigma > 0.f ? (lam_cut + sigma) * (lam_cut + sigma) : 0.f;
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
                if (nch <= CH * LP) {
                    // a lane owns at most CH 16-byte chunks of each column (n <= 128: four, n <= 192: six) and keeps
                    // them in registers between the dot product and the rotation (one LDS read, one write).  The
                    // loads are unconditional (clamped chunk index) so that all of them are in flight together;
                    // chunks a lane does not own are replaced by zeros afterwards
                    f32x4 x[CH], y[CH];
                    bool h[CH];
#pragma unroll
                    for (int e = 0; e < CH; e++) {
                        const int ke = min(sub + e * LP, nch - 1);
                        x[e] = gp[ke];
                        y[e] = gq[ke];
                        h[e] = live && sub + e * LP < nch;
                    }
                    float cp = 0.f;
#pragma unroll
                    for (int e = 0; e < CH; e++) {
                        x[e] = h[e] ? x[e] : z4;
                        y[e] = h[e] ? y[e] : z4;
                    }
#pragma unroll
                    for (int e = 0; e < CH; e += 2) cp += dot4(x[e], y[e]) + dot4(x[e + 1], y[e + 1]);
                    const float c = row8_sum(cp);
                    const float off = (live && ab > 0.f) ? fabsf(c) * __builtin_amdgcn_rsqf(ab) : 0.f;
                    off_max = fmaxf(off_max, fmaxf(a, b) > cut2 ? off : 0.f);
                    if (off > tol) {
                        float t, cs, sn;
                        rotation(a, b, c, t, cs, sn);
#pragma unroll
                        for (int e = 0; e < CH; e++)
                            if (h[e]) { gp[sub + e * LP] = cs * x[e] - sn * y[e]; gq[sub + e * LP] = sn * x[e] + cs * y[e]; }
                        if (sub == 0) { nrm[p] = a - t * c; nrm[q] = b + t * c; }
                    }
                    continue;
                }
                float c = 0.f;
                if (live)
                    for (int ch = sub; ch < nch; ch += LP) c += dot4(gp[ch], gq[ch]);
                c = row8_sum(c);
                const float off = (live && ab > 0.f) ? fabsf(c) * __builtin_amdgcn_rsqf(ab) : 0.f;
                off_max = fmaxf(off_max, fmaxf(a, b) > cut2 ? off : 0.f);
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

template <int JT>
__global__ __launch_bounds__(JT) void jacobi_lds_kernel(const float *T, long ldt, int n, int ldc, float *evals, float *Vout,
                                                        long ldv, float *scratch, int *sweeps_out, float stop_cos, float lam_cut)
{
    extern __shared__ __attribute__((aligned(16))) float sh[];
    float *G = sh, *red = sh + (size_t)n * ldc, *nrm = red + JT / 64 + 4;
    jacobi_body<JT, (JT > 512 ? 6 : 4)>(G, ldc, T, ldt, n, evals, Vout, ldv, scratch, sweeps_out, red, nrm, stop_cos, lam_cut);
}

template <int JT>
__global__ __launch_bounds__(JT) void jacobi_gmem_kernel(const float *T, long ldt, int n, int ldc, float *evals, float *Vout,
                                                         long ldv, float *work, int *sweeps_out, float stop_cos, float lam_cut)
{
    __shared__ float red[JT / 64 + 4];
    float *G = work, *scratch = work + (size_t)n * ldc;
    jacobi_body<JT, 4>(G, ldc, T, ldt, n, evals, Vout, ldv, scratch, sweeps_out, red, scratch + 3 * (size_t)n, stop_cos, lam_cut);
}

// ---- n <= 128: one column pair per 16 lanes ------------------------------------------------------
// Same method and the same arithmetic per rotation as jacobi_body, organised for the length of a
// round on one CU (the eigen tracker calls this once per step with n ~ 96 and every round is a
// workgroup barrier): 16 lanes per pair on up to 16 waves, so a lane carries n/16 entries of each
// column instead of n/8 and the whole round's vector work issues in a third of the slots; the
// round-robin schedule is a table in LDS (one 2-byte read instead of the modulo arithmetic);
// 8-byte LDS accesses at a stride of 16 lanes, no ownership masks (columns are padded with zeros
// to a whole number of chunks, which rotations keep zero); one barrier per round.
constexpr int J16_MAX_N = 128;
constexpr int J16_LP = 16;
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float row16_sum(float v)
{
    v += dpp_f<0x128>(v);    // row_ror:8
    v += dpp_f<0x124>(v);    // row_ror:4
    v += dpp_f<0x122>(v);    // row_ror:2
    v += dpp_f<0x121>(v);    // row_ror:1
    return v;
}

inline int j16_chunks(int n) { return (n + 31) / 32; }                       // 8-byte chunks per lane and column
inline int j16_ldc(int n) { const int e = j16_chunks(n); return 32 * e + ((e & 1) ? 0 : 32); }   // = 32 mod 64: two pairs of a
                                                                             // 32-lane group mostly hit different banks
inline int j16_threads(int n) { const int half = (n + 1) / 2; return ((half * J16_LP + 63) / 64) * 64; }
inline size_t j16_lds_bytes(int n)
{
    const int ne = n + (n & 1);
    return ((size_t)n * j16_ldc(n) + 136 + 24) * sizeof(float) + (size_t)(ne - 1) * (ne / 2) * sizeof(uint16_t) + 16;
}

template <int E>
__global__ __launch_bounds__(1024) void jacobi16_kernel(const float *T, long ldt, int n, int ldc, float *evals, float *Vout,
                                                        long ldv, float *scratch, int *sweeps_out, float stop_cos, float lam_cut)
{
    extern __shared__ __attribute__((aligned(16))) float sh[];
    float *G = sh;                                   // [n][ldc], column j at G + j*ldc
    float *nrm = G + (size_t)n * ldc;                // [136]
    float *red = nrm + 136;                          // [24]
    uint16_t *sched = reinterpret_cast<uint16_t *>(red + 24);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nthr = blockDim.x, nw = nthr >> 6;

    // ---- shift: sigma = 1.01 * max_i sum_j |T_ij| + tiny  (Gershgorin) ------------------------
    float rmax = 0.f;
    for (int i = wave; i < n; i += nw) {
        float s = 0.f;
        for (int j = lane; j < n; j += 64) s += fabsf(T[(long)i * ldt + j]);
        s = wsum(s);
        rmax = fmaxf(rmax, s);
    }
    if (lane == 0) red[wave] = rmax;
    // ---- the schedule: round r, slot k -> (p, q), p < q; 0xFFFF = bye ---------------------------------
    const int ne = n + (n & 1), half = ne / 2;
    for (int e = tid; e < (ne - 1) * half; e += nthr) {
        const int r = e / half, k = e % half;
        int p, q;
        if (k == 0) { p = ne - 1; q = r; }
        else {
            p = r + k; if (p >= ne - 1) p -= ne - 1;
            q = r - k; if (q < 0) q += ne - 1;
        }
        if (p > q) { const int t = p; p = q; q = t; }
        sched[e] = (p < n && q < n) ? (uint16_t)(p | (q << 8)) : (uint16_t)0xFFFF;
    }
    __syncthreads();
    float sigma = 0.f;
    for (int w = 0; w < nw; w++) sigma = fmaxf(sigma, red[w]);
    sigma = 1.01f * sigma + 1e-30f;
    // columns whose eigenvalue estimate |g| - sigma is below lam_cut are guard directions of the caller: a pair of
    // two such columns is rotated like any other but does not keep the sweeps going (squared norms are compared)
    const float cut2 = lam_cut + sigma > 0.f ? (lam_cut + sigma) * (lam_cut + sigma) : 0.f;

    // ---- init: G = sym(T) + sigma I, zero padding ---------------------------------------------------
    for (int e = tid; e < n * ldc; e += nthr) {
        const int j = e / ldc, i = e % ldc;
        float v = 0.f;
        if (i < n) v = 0.5f * (T[(long)i * ldt + j] + T[(long)j * ldt + i]) + (i == j ? sigma : 0.f);
        G[e] = v;
    }
    __syncthreads();

    const float tol = 3e-6f;
    const int grp = tid / J16_LP, sub = tid % J16_LP;
    const int ngrp = nthr / J16_LP;
    const bool has_slot = grp < half;
    int sweep = 0;
    for (; sweep < 40; sweep++) {
        // exact squared column norms at the start of the sweep
        for (int j = grp; j < n; j += ngrp) {
            const f32x2 *gj = reinterpret_cast<const f32x2 *>(G + (long)j * ldc) + sub;
            float s = 0.f;
#pragma unroll
            for (int e = 0; e < E; e++) { const f32x2 x = gj[16 * e]; s += x[0] * x[0] + x[1] * x[1]; }
            s = row16_sum(s);
            if (sub == 0) nrm[j] = s;
        }
        __syncthreads();
        float off_max = 0.f;
        for (int r = 0; r < ne - 1; r++) {
            const unsigned pq = has_slot ? sched[r * half + grp] : 0xFFFFu;
            const bool live = pq != 0xFFFFu;                 // uniform over the 16 lanes of the pair
            const int p = live ? (int)(pq & 0xFF) : 0, q = live ? (int)(pq >> 8) : 0;
            f32x2 *gp = reinterpret_cast<f32x2 *>(G + (long)p * ldc) + sub;
            f32x2 *gq = reinterpret_cast<f32x2 *>(G + (long)q * ldc) + sub;
            const float a = nrm[p], b = nrm[q];
            f32x2 x[E], y[E];
#pragma unroll
            for (int e = 0; e < E; e++) { x[e] = gp[16 * e]; y[e] = gq[16 * e]; }
            float c = 0.f;
#pragma unroll
            for (int e = 0; e < E; e++) c += x[e][0] * y[e][0] + x[e][1] * y[e][1];
            c = row16_sum(c);
            const float ab = a * b;
            const float off = (live && ab > 0.f) ? fabsf(c) * __builtin_amdgcn_rsqf(ab) : 0.f;
            off_max = fmaxf(off_max, fmaxf(a, b) > cut2 ? off : 0.f);
            if (off > tol) {
                float t, cs, sn;
                rotation(a, b, c, t, cs, sn);
#pragma unroll
                for (int e = 0; e < E; e++) {
                    gp[16 * e] = cs * x[e] - sn * y[e];
                    gq[16 * e] = sn * x[e] + cs * y[e];
                }
                if (sub == 0) { nrm[p] = a - t * c; nrm[q] = b + t * c; }
            }
            __syncthreads();
        }
        off_max = wmax(off_max);
        if (lane == 0) red[wave] = off_max;
        __syncthreads();
        float m = 0.f;
        for (int w = 0; w < nw; w++) m = fmaxf(m, red[w]);
        __syncthreads();
        if (m <= stop_cos) { sweep++; break; }               // see jacobi_body for why no verification sweep follows
    }
    if (tid == 0 && sweeps_out) *sweeps_out = sweep;

    // ---- eigenvalues lambda_j = |g_j| - sigma, eigenvectors v_j = g_j / |g_j|, sorted descending ----
    float *lam = scratch, *inv = scratch + n;
    int *rank = reinterpret_cast<int *>(scratch + 2 * n);
    for (int j = wave; j < n; j += nw) {
        float d = 0.f;
        for (int i = lane; i < n; i += 64) { const float v = G[(long)j * ldc + i]; d += v * v; }
        d = wsum(d);
        if (lane == 0) { const float nr = sqrtf(d); lam[j] = nr - sigma; inv[j] = nr > 0.f ? 1.f / nr : 0.f; }
    }
    __syncthreads();
    for (int j = tid; j < n; j += nthr) {
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
    for (int e = tid; e < n * n; e += nthr) {
        const int j = e / n, i = e % n;
        Vout[(long)i * ldv + rank[j]] = G[(long)j * ldc + i] * inv[j];
    }
}

// ---- n <= 128, columns travel: every address of a round is static ------------------------------------
// The round-robin tournament as a fixed permutation of SEATS: pair k always works on seat k (two column
// slots, "top" and "bottom") and always sends its two rotated columns to the same two destination slots
// (top_k -> top_k+1, bottom_k -> bottom_k-1, with top_0 fixed and the two turn-arounds), so that after
// ne-1 rounds every column has met every other one and is back where it started.  Two LDS images are
// used in turn (read one, write the other): a round is
//     barrier -> 6 static 8-byte reads per column -> dot product (16-lane DPP sum) -> rotation ->
//     6 static 8-byte writes per column
// with no schedule lookup, no address arithmetic and no separate norm vector (a column's squared norm
// rides in its slot, after the padding).  Same rotation arithmetic and stopping rule as jacobi_body.
inline int jseat_ldc(int n) { return 32 * ((n + 31) / 32) + 16; }    // 16-byte aligned, consecutive seats 32 banks apart
inline size_t jseat_lds_bytes(int n) { const int ne = n + (n & 1); return ((size_t)2 * ne * jseat_ldc(n) + 32) * sizeof(float); }

template <int E>
__global__ __launch_bounds__(1024) void jacobi_seat_kernel(const float *T, long ldt, int n, float *evals, float *Vout, long ldv,
                                                           float *scratch, int *sweeps_out, float stop_cos, float lam_cut)
{
    constexpr int LDC = 32 * E + 16;
    extern __shared__ __attribute__((aligned(16))) float sh[];
    const int ne = n + (n & 1), h = ne / 2;
    float *buf0 = sh, *buf1 = sh + (size_t)ne * LDC, *red = buf1 + (size_t)ne * LDC;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nthr = blockDim.x, nw = nthr >> 6;

    // ---- shift: sigma = 1.01 * max_i sum_j |T_ij| + tiny  (Gershgorin) ------------------------
    float rmax = 0.f;
    for (int i = wave; i < n; i += nw) {
        float s = 0.f;
        for (int j = lane; j < n; j += 64) s += fabsf(T[(long)i * ldt + j]);
        s = wsum(s);
        rmax = fmaxf(rmax, s);
    }
    if (lane == 0) red[wave] = rmax;
    __syncthreads();
    float sigma = 0.f;
    for (int w = 0; w < nw; w++) sigma = fmaxf(sigma, red[w]);
    sigma = 1.01f * sigma + 1e-30f;
    // columns whose eigenvalue estimate |g| - sigma is below lam_cut are guard directions of the caller: a pair of
    // two such columns is rotated like any other but does not keep the sweeps going (squared norms are compared)
    const float cut2 = lam_cut + sigma > 0.f ? (lam_cut + sigma) * (lam_cut + sigma) : 0.f;
    // ---- column j of G = sym(T) + sigma I starts in slot j; slot n (odd n) is a zero column (the bye) ---
    for (int e = tid; e < ne * LDC; e += nthr) {
        const int j = e / LDC, i = e % LDC;
        float v = 0.f;
        if (i < n && j < n) v = 0.5f * (T[(long)i * ldt + j] + T[(long)j * ldt + i]) + (i == j ? sigma : 0.f);
        buf0[e] = v;
    }
    __syncthreads();

    const float tol = 3e-6f;
    const int grp = tid / J16_LP, sub = tid % J16_LP;
    const bool seated = grp < h;
    // destination slots of this seat's two columns (static)
    int dtop = 2 * grp, dbot = 2 * grp + 1;
    if (h > 1) {
        dtop = grp == 0 ? 0 : (grp < h - 1 ? 2 * (grp + 1) : 2 * (h - 1) + 1);
        dbot = grp == 0 ? 2 : 2 * (grp - 1) + 1;
    }
    const int src_off = (seated ? 2 * grp : 0) * LDC + 2 * sub;
    const int dtop_off = (seated ? dtop : 0) * LDC + 2 * sub, dbot_off = (seated ? dbot : 0) * LDC + 2 * sub;
    int it = 0, sweep = 0;
    for (; sweep < 40; sweep++) {
        float off_max = 0.f;
        for (int r = 0; r < ne - 1; r++, it++) {
            const float *cur = (it & 1) ? buf1 : buf0;
            float *nxt = (it & 1) ? buf0 : buf1;
            if (seated) {
                const float *st = cur + src_off, *sb = st + LDC;
                f32x2 x[E], y[E];
#pragma unroll
                for (int e = 0; e < E; e++) {
                    x[e] = *reinterpret_cast<const f32x2 *>(st + 32 * e);
                    y[e] = *reinterpret_cast<const f32x2 *>(sb + 32 * e);
                }
                float a = st[32 * E - 2 * sub], b = sb[32 * E - 2 * sub];      // the norms ride behind the padding
                float c = 0.f;
#pragma unroll
                for (int e = 0; e < E; e++) c += x[e][0] * y[e][0] + x[e][1] * y[e][1];
                c = row16_sum(c);
                if (r == 0) {                                        // exact squared norms at the start of a sweep
                    float sa = 0.f, sb2 = 0.f;
#pragma unroll
                    for (int e = 0; e < E; e++) { sa += x[e][0] * x[e][0] + x[e][1] * x[e][1]; sb2 += y[e][0] * y[e][0] + y[e][1] * y[e][1]; }
                    a = row16_sum(sa);
                    b = row16_sum(sb2);
                }
                const float ab = a * b;
                const float off = ab > 0.f ? fabsf(c) * __builtin_amdgcn_rsqf(ab) : 0.f;
                off_max = fmaxf(off_max, fmaxf(a, b) > cut2 ? off : 0.f);
                float cs = 1.f, sn = 0.f, tc = 0.f;
                if (off > tol) {
                    float t;
                    rotation(a, b, c, t, cs, sn);
                    tc = t * c;
                }
                float *wt = nxt + dtop_off, *wb = nxt + dbot_off;
#pragma unroll
                for (int e = 0; e < E; e++) {
                    *reinterpret_cast<f32x2 *>(wt + 32 * e) = cs * x[e] - sn * y[e];
                    *reinterpret_cast<f32x2 *>(wb + 32 * e) = sn * x[e] + cs * y[e];
                }
                if (sub == 0) { wt[32 * E] = a - tc; wb[32 * E] = b + tc; }
            }
            __syncthreads();
        }
        off_max = wmax(off_max);
        if (lane == 0) red[wave] = off_max;
        __syncthreads();
        float m = 0.f;
        for (int w = 0; w < nw; w++) m = fmaxf(m, red[w]);
        __syncthreads();
        if (m <= stop_cos) { sweep++; break; }               // see jacobi_body for why no verification sweep follows
    }
    if (tid == 0 && sweeps_out) *sweeps_out = sweep;

    // ---- eigenvalues lambda = |g| - sigma, eigenvectors g / |g|, sorted descending; the bye has |g| = 0 ----
    const float *G = (it & 1) ? buf1 : buf0;
    float *lam = scratch, *inv = scratch + ne;
    int *rank = reinterpret_cast<int *>(scratch + 2 * ne);
    for (int j = wave; j < ne; j += nw) {
        float d = 0.f;
        for (int i = lane; i < n; i += 64) { const float v = G[(long)j * LDC + i]; d += v * v; }
        d = wsum(d);
        if (lane == 0) { const float nr = sqrtf(d); lam[j] = nr > 0.f ? nr - sigma : -3.0e38f; inv[j] = nr > 0.f ? 1.f / nr : 0.f; }
    }
    __syncthreads();
    for (int j = tid; j < ne; j += nthr) {
        const float me = lam[j];
        int rk = 0;
        for (int k = 0; k < ne; k++) {
            const float o = lam[k];
            rk += (o > me || (o == me && k < j)) ? 1 : 0;
        }
        rank[j] = rk;
        if (rk < n) evals[rk] = me;
    }
    __syncthreads();
    for (int e = tid; e < ne * n; e += nthr) {
        const int j = e / n, i = e % n;
        if (rank[j] < n) Vout[(long)i * ldv + rank[j]] = G[(long)j * LDC + i] * inv[j];
    }
}

// ---- the same seat scheme with 8 lanes per pair (16-byte chunks) -------------------------------------------
// PING = true : two LDS images, one barrier per round (n <= 128), half the waves of the 16-lane form;
// PING = false: ONE image, read -> barrier -> write -> barrier, for 128 < n <= 190 where two images of the
//               matrix do not fit 160 KB (the rank ~128 workload: blocks of 148 - 160 rows).
inline int jseat8_threads(int n) { const int half = (n + 1) / 2; return ((half * 8 + 63) / 64) * 64; }
inline size_t jseat8_lds_bytes(int n, bool ping) { const int ne = n + (n & 1); return ((size_t)(ping ? 2 : 1) * ne * jseat_ldc(n) + 32) * sizeof(float); }

template <int E, bool PING>
__global__ __launch_bounds__(1024) void jacobi_seat8_kernel(const float *T, long ldt, int n, float *evals, float *Vout, long ldv,
                                                            float *scratch, int *sweeps_out, float stop_cos, float lam_cut)
{
    constexpr int LDC = 32 * E + 16;
    extern __shared__ __attribute__((aligned(16))) float sh[];
    const int ne = n + (n & 1), h = ne / 2;
    float *buf0 = sh, *buf1 = PING ? sh + (size_t)ne * LDC : sh, *red = sh + (size_t)(PING ? 2 : 1) * ne * LDC;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nthr = blockDim.x, nw = nthr >> 6;

    float rmax = 0.f;                                          // Gershgorin shift, as in jacobi_seat_kernel
    for (int i = wave; i < n; i += nw) {
        float s = 0.f;
        for (int j = lane; j < n; j += 64) s += fabsf(T[(long)i * ldt + j]);
        s = wsum(s);
        rmax = fmaxf(rmax, s);
    }
    if (lane == 0) red[wave] = rmax;
    __syncthreads();
    float sigma = 0.f;
    for (int w = 0; w < nw; w++) sigma = fmaxf(sigma, red[w]);
    sigma = 1.01f * sigma + 1e-30f;
    const float cut2 = lam_cut + sigma > 0.f ? (lam_cut + sigma) * (lam_cut + sigma) : 0.f;
    for (int e = tid; e < ne * LDC; e += nthr) {
        const int j = e / LDC, i = e % LDC;
        float v = 0.f;
        if (i < n && j < n) v = 0.5f * (T[(long)i * ldt + j] + T[(long)j * ldt + i]) + (i == j ? sigma : 0.f);
        buf0[e] = v;
    }
    __syncthreads();

    const float tol = 3e-6f;
    const int grp = tid >> 3, sub = tid & 7;
    const bool seated = grp < h;
    int dtop = 2 * grp, dbot = 2 * grp + 1;
    if (h > 1) {
        dtop = grp == 0 ? 0 : (grp < h - 1 ? 2 * (grp + 1) : 2 * (h - 1) + 1);
        dbot = grp == 0 ? 2 : 2 * (grp - 1) + 1;
    }
    const int src_off = (seated ? 2 * grp : 0) * LDC + 4 * sub;
    const int dtop_off = (seated ? dtop : 0) * LDC + 4 * sub, dbot_off = (seated ? dbot : 0) * LDC + 4 * sub;
    int it = 0, sweep = 0;
    for (; sweep < 40; sweep++) {
        float off_max = 0.f;
        for (int r = 0; r < ne - 1; r++, it++) {
            const float *cur = (PING && (it & 1)) ? buf1 : buf0;
            float *nxt = (PING && !(it & 1)) ? buf1 : buf0;
            f32x4 x[E], y[E];
            float cs = 1.f, sn = 0.f, na = 0.f, nb = 0.f;
            if (seated) {
                const float *st = cur + src_off, *sb = st + LDC;
#pragma unroll
                for (int e = 0; e < E; e++) {
                    x[e] = *reinterpret_cast<const f32x4 *>(st + 32 * e);
                    y[e] = *reinterpret_cast<const f32x4 *>(sb + 32 * e);
                }
                float a = st[32 * E - 4 * sub], b = sb[32 * E - 4 * sub];      // the norms ride behind the padding
                float c = 0.f;
#pragma unroll
                for (int e = 0; e < E; e++) c += dot4(x[e], y[e]);
                c = row8_sum(c);
                if (r == 0) {                                        // exact squared norms at the start of a sweep
                    float sa = 0.f, sb2 = 0.f;
#pragma unroll
                    for (int e = 0; e < E; e++) { sa += dot4(x[e], x[e]); sb2 += dot4(y[e], y[e]); }
                    a = row8_sum(sa);
                    b = row8_sum(sb2);
                }
                const float ab = a * b;
                const float off = ab > 0.f ? fabsf(c) * __builtin_amdgcn_rsqf(ab) : 0.f;
                off_max = fmaxf(off_max, fmaxf(a, b) > cut2 ? off : 0.f);
                float tc = 0.f;
                if (off > tol) {
                    float t;
                    rotation(a, b, c, t, cs, sn);
                    tc = t * c;
                }
                na = a - tc; nb = b + tc;
            }
            if (!PING) __syncthreads();                              // every seat has read the image before any writes it
            if (seated) {
                float *wt = nxt + dtop_off, *wb = nxt + dbot_off;
#pragma unroll
                for (int e = 0; e < E; e++) {
                    *reinterpret_cast<f32x4 *>(wt + 32 * e) = cs * x[e] - sn * y[e];
                    *reinterpret_cast<f32x4 *>(wb + 32 * e) = sn * x[e] + cs * y[e];
                }
                if (sub == 0) { wt[32 * E] = na; wb[32 * E] = nb; }
            }
            __syncthreads();
        }
        off_max = wmax(off_max);
        if (lane == 0) red[wave] = off_max;
        __syncthreads();
        float m = 0.f;
        for (int w = 0; w < nw; w++) m = fmaxf(m, red[w]);
        __syncthreads();
        if (m <= stop_cos) { sweep++; break; }
    }
    if (tid == 0 && sweeps_out) *sweeps_out = sweep;

    const float *G = (PING && (it & 1)) ? buf1 : buf0;
    float *lam = scratch, *inv = scratch + ne;
    int *rank = reinterpret_cast<int *>(scratch + 2 * ne);
    for (int j = wave; j < ne; j += nw) {
        float d = 0.f;
        for (int i = lane; i < n; i += 64) { const float v = G[(long)j * LDC + i]; d += v * v; }
        d = wsum(d);
        if (lane == 0) { const float nr = sqrtf(d); lam[j] = nr > 0.f ? nr - sigma : -3.0e38f; inv[j] = nr > 0.f ? 1.f / nr : 0.f; }
    }
    __syncthreads();
    for (int j = tid; j < ne; j += nthr) {
        const float me = lam[j];
        int rk = 0;
        for (int k = 0; k < ne; k++) {
            const float o = lam[k];
            rk += (o > me || (o == me && k < j)) ? 1 : 0;
        }
        rank[j] = rk;
        if (rk < n) evals[rk] = me;
    }
    __syncthreads();
    for (int e = tid; e < ne * n; e += nthr) {
        const int j = e / n, i = e % n;
        if (rank[j] < n) Vout[(long)i * ldv + rank[j]] = G[(long)j * LDC + i] * inv[j];
    }
}

// ---- n <= 160: blocks of eight columns, one WAVE per pair of blocks ------------------------------------------------
// The seat kernels above spend a workgroup barrier and a full instruction stream on every one of the n - 1 rounds of a
// sweep, and the round is bound by vector-instruction issue (12 - 16 waves on four SIMDs).  Here the tournament is
// played between BLOCKS of eight columns: a wave takes two blocks and rotates all 64 cross pairs without leaving its
// own part of the matrix - eight lanes per pair, the eight columns of block A stay in registers for the whole
// meeting, the columns of block B pass through them from LDS in eight inner rounds (pair (A_g, B_(g+k) mod 8) in round
// k) - so a sweep has nb - 1 barriers (nb = n / 8 blocks) instead of n - 1, and the LDS traffic of a rotation is one
// column in and out instead of two.  The 28 pairs inside each block are rotated once per sweep (seven wave-private
// rounds through LDS) before the blocks start to meet.  Blocks are paired by the circle method, computed from the
// round number: columns never move in LDS, a single image suffices (160 columns: 100 KB).  Same rotation arithmetic,
// tracked squared norms (refreshed exactly every sweep) and stopping rule as the kernels above.
// Column j lives at G + j * 32 E; lane l of a pair's eight owns the 16-byte chunks l + 8 e (e < E).  E is odd, so
// consecutive columns start 32 banks apart and the eight-column reads of a wave are conflict-free.
constexpr int JBLK_MAX_N = 160;
inline int jblk_chunks(int n) { return n <= 32 ? 1 : (n <= 96 ? 3 : 5); }
inline int jblk_waves(int n) { const int nb = (n + 7) / 8; return (nb + (nb & 1)) / 2; }
inline size_t jblk_lds_bytes(int n) { const int ncol = 16 * jblk_waves(n); return ((size_t)ncol * 32 * jblk_chunks(n) + ncol + 32) * sizeof(float); }

template <int E>
__global__ __launch_bounds__(640) void jacobi_blk_kernel(const float *T, long ldt, int n, float *evals, float *Vout, long ldv,
                                                         float *scratch, int *sweeps_out, float stop_cos, float lam_cut)
{
    constexpr int LDC = 32 * E;
    extern __shared__ __attribute__((aligned(16))) float sh[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nthr = blockDim.x, nw = nthr >> 6;               // nw = block pairs = (number of blocks, made even) / 2
    const int nbe = 2 * nw, ncol = 8 * nbe, m = nbe - 1;
    float *G = sh, *nrm = sh + (size_t)ncol * LDC, *red = nrm + ncol;

    // ---- shift: sigma = 1.01 * max_i sum_j |T_ij| + tiny  (Gershgorin) ------------------------
    float rmax = 0.f;
    for (int i = wave; i < n; i += nw) {
        float s = 0.f;
        for (int j = lane; j < n; j += 64) s += fabsf(T[(long)i * ldt + j]);
        s = wsum(s);
        rmax = fmaxf(rmax, s);
    }
    if (lane == 0) red[wave] = rmax;
    __syncthreads();
    float sigma = 0.f;
    for (int w = 0; w < nw; w++) sigma = fmaxf(sigma, red[w]);
    sigma = 1.01f * sigma + 1e-30f;
    const float cut2 = lam_cut + sigma > 0.f ? (lam_cut + sigma) * (lam_cut + sigma) : 0.f;
    // ---- column j of G = sym(T) + sigma I; columns n .. ncol-1 and rows n .. LDC-1 are zero (a zero column is never rotated)
    for (int e = tid; e < ncol * LDC; e += nthr) {
        const int j = e / LDC, i = e % LDC;
        float v = 0.f;
        if (i < n && j < n) v = 0.5f * (T[(long)i * ldt + j] + T[(long)j * ldt + i]) + (i == j ? sigma : 0.f);
        G[e] = v;
    }
    __syncthreads();

    const float tol = 3e-6f;
    const int g = lane >> 3, l = lane & 7;
    float off_max = 0.f;
    // one rotation of the pair (x: squared norm a, y: squared norm b); returns t * (x . y) for the tracked norms
    auto rotate = [&](f32x4 (&x)[E], f32x4 (&y)[E], float a, float b) -> float {
        float c = 0.f;
#pragma unroll
        for (int e = 0; e < E; e++) c += dot4(x[e], y[e]);
        c = row8_sum(c);
        const float ab = a * b;
        const float off = ab > 0.f ? fabsf(c) * __builtin_amdgcn_rsqf(ab) : 0.f;
        off_max = fmaxf(off_max, fmaxf(a, b) > cut2 ? off : 0.f);
        float cs = 1.f, sn = 0.f, tc = 0.f;
        if (off > tol) {
            float t;
            rotation(a, b, c, t, cs, sn);
            tc = t * c;
        }
#pragma unroll
        for (int e = 0; e < E; e++) {
            const f32x4 xn = cs * x[e] - sn * y[e];
            y[e] = sn * x[e] + cs * y[e];
            x[e] = xn;
        }
        return tc;
    };
    auto load_col = [&](int col, f32x4 (&x)[E]) {
        const float *p = G + (size_t)col * LDC + 4 * l;
#pragma unroll
        for (int e = 0; e < E; e++) x[e] = *reinterpret_cast<const f32x4 *>(p + 32 * e);
    };
    auto store_col = [&](int col, const f32x4 (&x)[E]) {
        float *p = G + (size_t)col * LDC + 4 * l;
#pragma unroll
        for (int e = 0; e < E; e++) *reinterpret_cast<f32x4 *>(p + 32 * e) = x[e];
    };

    int sweep = 0;
    for (; sweep < 40; sweep++) {
        off_max = 0.f;
        f32x4 x[E], y[E];
        // exact squared norms of this wave's sixteen columns at the start of the sweep
#pragma unroll
        for (int q = 0; q < 2; q++) {
            const int col = 16 * wave + 8 * q + g;
            load_col(col, x);
            float s = 0.f;
#pragma unroll
            for (int e = 0; e < E; e++) s += dot4(x[e], x[e]);
            s = row8_sum(s);
            if (l == 0) nrm[col] = s;
        }
        // ---- the pairs inside blocks 2 wave (lane groups 0-3) and 2 wave + 1 (groups 4-7): seven wave-private rounds ---
        {
            const int base = 16 * wave + 8 * (g >> 2), k = g & 3;
            for (int r = 0; r < 7; r++) {
                const int pa = k == 0 ? 7 : (r + k) % 7, pb = k == 0 ? r : (r - k + 7) % 7;
                const int ca = base + pa, cb = base + pb;
                load_col(ca, x);
                load_col(cb, y);
                const float a = nrm[ca], b = nrm[cb];
                const float tc = rotate(x, y, a, b);
                store_col(ca, x);
                store_col(cb, y);
                if (l == 0) { nrm[ca] = a - tc; nrm[cb] = b + tc; }
            }
        }
        __syncthreads();
        // ---- the blocks meet: round r pairs block m with block r, and (r + k) mod m with (r - k) mod m ---------------
        for (int r = 0; r < m; r++) {
            const int ba = wave == 0 ? m : (r + wave) % m, bb = wave == 0 ? r : (r - wave + m) % m;
            const int ca = 8 * ba + g;
            load_col(ca, x);
            float a = nrm[ca];
#pragma unroll 1
            for (int kk = 0; kk < 8; kk++) {
                const int cb = 8 * bb + ((g + kk) & 7);
                load_col(cb, y);
                const float b = nrm[cb];
                const float tc = rotate(x, y, a, b);
                store_col(cb, y);
                a -= tc;
                if (l == 0) nrm[cb] = b + tc;
            }
            store_col(ca, x);
            if (l == 0) nrm[ca] = a;
            __syncthreads();
        }
        off_max = wmax(off_max);
        if (lane == 0) red[wave] = off_max;
        __syncthreads();
        float mx = 0.f;
        for (int w = 0; w < nw; w++) mx = fmaxf(mx, red[w]);
        __syncthreads();
        if (mx <= stop_cos) { sweep++; break; }              // see jacobi_body for why no verification sweep follows
    }
    if (tid == 0 && sweeps_out) *sweeps_out = sweep;

    // ---- eigenvalues lambda = |g| - sigma, eigenvectors g / |g|, sorted descending ------------------------------------
    float *lam = scratch, *inv = scratch + n;
    int *rank = reinterpret_cast<int *>(scratch + 2 * n);
    for (int j = wave; j < n; j += nw) {
        float d = 0.f;
        for (int i = lane; i < n; i += 64) { const float v = G[(size_t)j * LDC + i]; d += v * v; }
        d = wsum(d);
        if (lane == 0) { const float nr = sqrtf(d); lam[j] = nr - sigma; inv[j] = nr > 0.f ? 1.f / nr : 0.f; }
    }
    __syncthreads();
    for (int j = tid; j < n; j += nthr) {
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
    for (int e = tid; e < n * n; e += nthr) {
        const int j = e / n, i = e % n;
        Vout[(long)i * ldv + rank[j]] = G[(size_t)j * LDC + i] * inv[j];
    }
}

template <int E>
void launch_jacobi_blk(const float *T, long ldt, int n, float *evals, float *V, long ldv, float *work, int *sweeps_out,
                       float stop_cos, float lam_cut, hipStream_t s)
{
    static bool attr = false;
    if (!attr) {
        DLCO_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(jacobi_blk_kernel<E>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     160 * 1024 - 1024));
        attr = true;
    }
    hipLaunchKernelGGL(jacobi_blk_kernel<E>, dim3(1), dim3(64 * jblk_waves(n)), jblk_lds_bytes(n), s, T, ldt, n, evals, V, ldv, work,
                       sweeps_out, stop_cos, lam_cut);
}

// ---- 160 < n <= 2048: the same blocks, one workgroup (one wave) per block pair, MANY CUs ------------------------------
// Blocks of several hundred rows only occur in the start-up transient of a run (from W = 0 the rank overshoots to 300 -
// 900 before it settles), but there the one-workgroup kernel in global memory cost 6 ms a call, two thirds of the
// transient's 3.7 s.  The block tournament needs no shared LDS image: the matrix G = sym(T) + sigma I stays in global
// memory (L2 / Infinity Cache resident), a workgroup of ONE wave takes the two blocks of its pair into its own LDS, rotates
// the 64 cross pairs there (eight wave-private inner rounds, both columns of a pair read from and written to LDS) and writes
// the sixteen columns back; between two rounds of the tournament all workgroups meet at a grid barrier.  n / 16 workgroups on
// as many CUs, n / 8 - 1 barriers per sweep.
//   * Hand-off through global memory between CUs on different XCDs: every storing wave drains its stores, lane 0 issues an
//     agent-scope release, then arrives at a monotonic counter; after the poll an agent-scope acquire, then plain loads.
//   * The grid is at most 128 one-wave workgroups: always co-resident.  Every spin is bounded; a barrier that does not
//     complete sets *sweeps_out = -1 and every workgroup leaves.
struct JmwDev {
    const float *T; long ldt; int n;
    float *evals, *Vout; long ldv;
    float *G;                     // [ncol][ldc] work image of the columns, ldc = n rounded up to 32
    float *lam;                   // [ncol] + [ncol] (inverse norms) + [ncol] ranks
    unsigned *sync;               // [0] barrier counter, [1] timeout flag, [2 + sweep] max cosine of a sweep, [50] Gershgorin bound (float bits); zeroed by the host
    int *sweeps_out;
    float stop_cos, lam_cut;
    int ldc, nbe;
};

__device__ __forceinline__ bool jmw_grid_sync(unsigned *sync, unsigned nwg, unsigned &target)
{
    // one wave per workgroup: lane 0 speaks for it
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this wave's stores have left
    bool ok = true;
    if ((threadIdx.x & 63) == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        target += nwg;
        __hip_atomic_fetch_add(sync, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        while (__hip_atomic_load(sync, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(2);
            if (++spins > (1u << 22) || __hip_atomic_load(sync + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
                __hip_atomic_store(sync + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ok = false;
                break;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    return __shfl(ok ? 1 : 0, 0, 64) != 0;
}

__global__ __launch_bounds__(64) void jacobi_mw_kernel(JmwDev g)
{
    extern __shared__ __attribute__((aligned(16))) float sh[];
    const int n = g.n, ldc = g.ldc, nch = ldc >> 5;            // 32-float (4 floats x 8 lanes) pieces per column
    const int lane = threadIdx.x, grp = lane >> 3, l = lane & 7;
    const int k = blockIdx.x, nwg = gridDim.x, nbe = g.nbe, m = nbe - 1, ncol = 8 * nbe;
    float *colA = sh, *colB = sh + 8 * (size_t)ldc;             // the pair's two blocks, column c of a block at c * ldc
    float *nA = sh + 16 * (size_t)ldc, *nB = nA + 8;            // tracked squared norms of the sixteen columns
    unsigned target = 0;

    // ---- sigma = 1.01 * max_i sum_j |T_ij| + tiny (Gershgorin): rows dealt to the workgroups, maximum through a word --------
    {
        float rmax = 0.f;
        for (int i = k; i < n; i += nwg) {
            float sacc = 0.f;
            for (int j = lane; j < n; j += 64) sacc += fabsf(g.T[(long)i * g.ldt + j]);
            rmax = fmaxf(rmax, wsum(sacc));
        }
        if (lane == 0) atomicMax(g.sync + 50, __float_as_uint(rmax));   // non-negative floats order like their bits
    }
    if (!jmw_grid_sync(g.sync, nwg, target)) { if (k == 0 && lane == 0) *g.sweeps_out = -1; return; }
    const float sigma = 1.01f * __uint_as_float(__hip_atomic_load(g.sync + 50, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) + 1e-30f;
    const float cut2 = g.lam_cut + sigma > 0.f ? (g.lam_cut + sigma) * (g.lam_cut + sigma) : 0.f;
    // ---- G = sym(T) + sigma I, zero padded: this workgroup writes the sixteen columns of blocks 2k, 2k+1 ------------------
    for (int c = 0; c < 16; c++) {
        const int j = 16 * k + c;
        for (int i = lane; i < ldc; i += 64) {
            float v = 0.f;
            if (i < n && j < n) v = 0.5f * (g.T[(long)i * g.ldt + j] + g.T[(long)j * g.ldt + i]) + (i == j ? sigma : 0.f);
            g.G[(size_t)j * ldc + i] = v;
        }
    }
    if (!jmw_grid_sync(g.sync, nwg, target)) { if (k == 0 && lane == 0) *g.sweeps_out = -1; return; }

    const float tol = 3e-6f;
    // block (8 columns) global <-> LDS, with the exact squared norm of every column on the way in
    auto load_block = [&](int b, float *dst, float *nrm) {
        for (int c = 0; c < 8; c++) {
            const float *src = g.G + (size_t)(8 * b + c) * ldc;
            float sq = 0.f;
            for (int i = lane * 4; i < ldc; i += 256) {
                const f32x4 v = *reinterpret_cast<const f32x4 *>(src + i);
                *reinterpret_cast<f32x4 *>(dst + (size_t)c * ldc + i) = v;
                sq += dot4(v, v);
            }
            sq = wsum(sq);
            if (lane == 0) nrm[c] = sq;
        }
    };
    auto store_block = [&](int b, const float *src) {
        for (int c = 0; c < 8; c++) {
            float *dst = g.G + (size_t)(8 * b + c) * ldc;
            for (int i = lane * 4; i < ldc; i += 256) *reinterpret_cast<f32x4 *>(dst + i) = *reinterpret_cast<const f32x4 *>(src + (size_t)c * ldc + i);
        }
    };
    float off_max = 0.f;
    // one rotation of the pair (px: squared norm *na, py: *nb), both columns in LDS; lane group of 8 lanes, lane l owns pieces l + 8 e
    auto rotate_pair = [&](float *px, float *py, float *na, float *nb) {
        float c = 0.f;
        for (int e = 0; e < nch; e++) {
            const int o = 4 * (l + 8 * e);
            c += dot4(*reinterpret_cast<const f32x4 *>(px + o), *reinterpret_cast<const f32x4 *>(py + o));
        }
        c = row8_sum(c);
        const float a = *na, b = *nb;
        const float ab = a * b;
        const float off = ab > 0.f ? fabsf(c) * __builtin_amdgcn_rsqf(ab) : 0.f;
        off_max = fmaxf(off_max, fmaxf(a, b) > cut2 ? off : 0.f);
        if (off > tol) {
            float t, cs, sn;
            rotation(a, b, c, t, cs, sn);
            for (int e = 0; e < nch; e++) {
                const int o = 4 * (l + 8 * e);
                const f32x4 x = *reinterpret_cast<const f32x4 *>(px + o), y = *reinterpret_cast<const f32x4 *>(py + o);
                *reinterpret_cast<f32x4 *>(px + o) = cs * x - sn * y;
                *reinterpret_cast<f32x4 *>(py + o) = sn * x + cs * y;
            }
            if (l == 0) { *na = a - t * c; *nb = b + t * c; }
        }
    };

    int sweep = 0;
    bool alive = true;
    for (; sweep < 40 && alive; sweep++) {
        off_max = 0.f;
        // ---- pairs inside blocks 2k (lane groups 0-3) and 2k+1 (groups 4-7): seven wave-private rounds -------------------
        load_block(2 * k, colA, nA);
        load_block(2 * k + 1, colB, nB);
        {
            float *base = grp < 4 ? colA : colB, *nb_ = grp < 4 ? nA : nB;
            const int kk = grp & 3;
            for (int r = 0; r < 7; r++) {
                const int pa = kk == 0 ? 7 : (r + kk) % 7, pb = kk == 0 ? r : (r - kk + 7) % 7;
                rotate_pair(base + (size_t)pa * ldc, base + (size_t)pb * ldc, nb_ + pa, nb_ + pb);
            }
        }
        store_block(2 * k, colA);
        store_block(2 * k + 1, colB);
        if (!jmw_grid_sync(g.sync, nwg, target)) { alive = false; break; }
        // ---- the blocks meet: round r pairs block m with block r, and (r + k) mod m with (r - k) mod m -------------------
        for (int r = 0; r < m; r++) {
            const int ba = k == 0 ? m : (r + k) % m, bb = k == 0 ? r : (r - k + m) % m;
            load_block(ba, colA, nA);
            load_block(bb, colB, nB);
            for (int kk = 0; kk < 8; kk++) {
                const int cb = (grp + kk) & 7;
                rotate_pair(colA + (size_t)grp * ldc, colB + (size_t)cb * ldc, nA + grp, nB + cb);
            }
            store_block(ba, colA);
            store_block(bb, colB);
            if (!jmw_grid_sync(g.sync, nwg, target)) { alive = false; break; }
        }
        if (!alive) break;
        // ---- the sweep's largest cosine, over all workgroups --------------------------------------------------------------
        off_max = wmax(off_max);
        if (lane == 0) atomicMax(g.sync + 2 + sweep, __float_as_uint(off_max));      // non-negative floats order like their bits
        if (!jmw_grid_sync(g.sync, nwg, target)) { alive = false; break; }
        const float mx = __uint_as_float(__hip_atomic_load(g.sync + 2 + sweep, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        if (mx <= g.stop_cos) { sweep++; break; }
    }
    if (!alive) { if (k == 0 && lane == 0) *g.sweeps_out = -1; return; }
    if (k == 0 && lane == 0 && g.sweeps_out) *g.sweeps_out = sweep;

    // ---- eigenvalues lambda = |g| - sigma, eigenvectors g / |g|, sorted descending ------------------------------------------
    float *lam = g.lam, *inv = g.lam + ncol;
    for (int c = 0; c < 16; c++) {
        const int j = 16 * k + c;
        if (j >= n) break;
        float d = 0.f;
        for (int i = lane; i < n; i += 64) { const float v = g.G[(size_t)j * ldc + i]; d += v * v; }
        d = wsum(d);
        if (lane == 0) { const float nr = sqrtf(d); lam[j] = nr - sigma; inv[j] = nr > 0.f ? 1.f / nr : 0.f; }
    }
    if (!jmw_grid_sync(g.sync, nwg, target)) { if (k == 0 && lane == 0) *g.sweeps_out = -1; return; }
    for (int c = 0; c < 16; c++) {
        const int j = 16 * k + c;
        if (j >= n) break;
        const float me = lam[j];
        int rk = 0;
        for (int q = lane; q < n; q += 64) {
            const float o = lam[q];
            rk += (o > me || (o == me && q < j)) ? 1 : 0;
        }
        rk = (int)wsum((float)rk);                              // n <= 2048: exact in fp32
        if (lane == 0) g.evals[rk] = me;
        const float sc = inv[j];
        for (int i = lane; i < n; i += 64) g.Vout[(long)i * g.ldv + rk] = g.G[(size_t)j * ldc + i] * sc;
    }
}

constexpr int JMW_MAX_N = 2048;
inline int jmw_ldc(int n) { return (n + 31) & ~31; }
inline int jmw_nbe(int n) { const int nb = (n + 7) / 8; return nb + (nb & 1); }
inline size_t jmw_work_floats(int n) { const size_t ncol = 8 * (size_t)jmw_nbe(n); return ncol * jmw_ldc(n) + 3 * ncol + 64; }

template <int E, bool PING>
void launch_jacobi_seat8(const float *T, long ldt, int n, float *evals, float *V, long ldv, float *work, int *sweeps_out,
                         float stop_cos, float lam_cut, hipStream_t s)
{
    static bool attr = false;
    if (!attr) {
        DLCO_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(jacobi_seat8_kernel<E, PING>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     160 * 1024 - 1024));
        attr = true;
    }
    hipLaunchKernelGGL((jacobi_seat8_kernel<E, PING>), dim3(1), dim3(jseat8_threads(n)), jseat8_lds_bytes(n, PING), s, T, ldt, n, evals, V, ldv,
                       work, sweeps_out, stop_cos, lam_cut);
}

template <int E>
void launch_jacobi_seat(const float *T, long ldt, int n, float *evals, float *V, long ldv, float *work, int *sweeps_out,
                        float stop_cos, float lam_cut, hipStream_t s)
{
    static bool attr = false;
    if (!attr) {
        DLCO_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(jacobi_seat_kernel<E>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     160 * 1024 - 1024));
        attr = true;
    }
    hipLaunchKernelGGL(jacobi_seat_kernel<E>, dim3(1), dim3(j16_threads(n)), jseat_lds_bytes(n), s, T, ldt, n, evals, V, ldv, work,
                       sweeps_out, stop_cos, lam_cut);
}

inline int col_stride(int n) { return (n + 3) & ~3; }

}  // namespace

size_t jacobi_work_floats(int n) { return std::max((size_t)n * col_stride(n) + 4 * (size_t)n + 64, n <= JMW_MAX_N ? jmw_work_floats(n) : (size_t)0); }

void jacobi_eigh(const float *T, long ldt, int n, float *evals, float *V, long ldv, float *work, int *sweeps_out,
                 hipStream_t s, float lam_cut)
{
    DLCO_CHECK(n >= 1 && n <= 4096, -2, "jacobi_eigh: n out of range");
    const int ldc = col_stride(n);
    // largest column cosine of a sweep below which the sweep is the last one (see the kernel)
    static const float stop_cos = std::getenv("DLCO_JACOBI_STOP") ? (float)std::atof(std::getenv("DLCO_JACOBI_STOP")) : 1e-3f;
    static const bool use_v1 = std::getenv("DLCO_JACOBI_V1") != nullptr;
    static const bool use_v2 = std::getenv("DLCO_JACOBI_V2") != nullptr;
    static const bool use_lp8 = std::getenv("DLCO_JACOBI_LP8") != nullptr;         // 8-lane seats for n <= 128 as well
    static const bool no_wide_seat = std::getenv("DLCO_JACOBI_NO_WIDE_SEAT") != nullptr;
    static const bool no_blk = std::getenv("DLCO_JACOBI_SEAT") != nullptr;    // A/B switch: the seat kernels of round 2
    if (n >= 9 && n <= JBLK_MAX_N && !no_blk && !use_v1 && !use_v2 && !use_lp8) {
        const int e = jblk_chunks(n);
        if (e == 1) launch_jacobi_blk<1>(T, ldt, n, evals, V, ldv, work, sweeps_out, stop_cos, lam_cut, s);
        else if (e == 3) launch_jacobi_blk<3>(T, ldt, n, evals, V, ldv, work, sweeps_out, stop_cos, lam_cut, s);
        else launch_jacobi_blk<5>(T, ldt, n, evals, V, ldv, work, sweeps_out, stop_cos, lam_cut, s);
    } else if (n > J16_MAX_N && n <= 190 && !use_v1 && !no_wide_seat && jseat8_lds_bytes(n, false) <= 160 * 1024 - 1024) {
        if (j16_chunks(n) == 5) launch_jacobi_seat8<5, false>(T, ldt, n, evals, V, ldv, work, sweeps_out, stop_cos, lam_cut, s);
        else launch_jacobi_seat8<6, false>(T, ldt, n, evals, V, ldv, work, sweeps_out, stop_cos, lam_cut, s);
    } else if (n <= J16_MAX_N && use_lp8 && !use_v1 && !use_v2) {
        const int e = j16_chunks(n);
        if (e == 1) launch_jacobi_seat8<1, true>(T, ldt, n, evals, V, ldv, work, sweeps_out, stop_cos, lam_cut, s);
        else if (e == 2) launch_jacobi_seat8<2, true>(T, ldt, n, evals, V, ldv, work, sweeps_out, stop_cos, lam_cut, s);
        else if (e == 3) launch_jacobi_seat8<3, true>(T, ldt, n, evals, V, ldv, work, sweeps_out, stop_cos, lam_cut, s);
        else launch_jacobi_seat8<4, true>(T, ldt, n, evals, V, ldv, work, sweeps_out, stop_cos, lam_cut, s);
    } else if (n <= J16_MAX_N && !use_v1 && !use_v2) {
        const int e = j16_chunks(n);
        if (e == 1) launch_jacobi_seat<1>(T, ldt, n, evals, V, ldv, work, sweeps_out, stop_cos, lam_cut, s);
        else if (e == 2) launch_jacobi_seat<2>(T, ldt, n, evals, V, ldv, work, sweeps_out, stop_cos, lam_cut, s);
        else if (e == 3) launch_jacobi_seat<3>(T, ldt, n, evals, V, ldv, work, sweeps_out, stop_cos, lam_cut, s);
        else launch_jacobi_seat<4>(T, ldt, n, evals, V, ldv, work, sweeps_out, stop_cos, lam_cut, s);
    } else if (n <= J16_MAX_N && !use_v1) {
        static bool attr16 = false;
        if (!attr16) {
            DLCO_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(jacobi16_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024));
            DLCO_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(jacobi16_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024));
            DLCO_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(jacobi16_kernel<3>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024));
            DLCO_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(jacobi16_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024));
            attr16 = true;
        }
        const int e = j16_chunks(n), ldc16 = j16_ldc(n);
        const dim3 grid(1), block(j16_threads(n));
        const size_t lds16 = j16_lds_bytes(n);
        if (e == 1) hipLaunchKernelGGL(jacobi16_kernel<1>, grid, block, lds16, s, T, ldt, n, ldc16, evals, V, ldv, work, sweeps_out, stop_cos, lam_cut);
        else if (e == 2) hipLaunchKernelGGL(jacobi16_kernel<2>, grid, block, lds16, s, T, ldt, n, ldc16, evals, V, ldv, work, sweeps_out, stop_cos, lam_cut);
        else if (e == 3) hipLaunchKernelGGL(jacobi16_kernel<3>, grid, block, lds16, s, T, ldt, n, ldc16, evals, V, ldv, work, sweeps_out, stop_cos, lam_cut);
        else hipLaunchKernelGGL(jacobi16_kernel<4>, grid, block, lds16, s, T, ldt, n, ldc16, evals, V, ldv, work, sweeps_out, stop_cos, lam_cut);
    } else if (n > JBLK_MAX_N && n <= JMW_MAX_N && !no_blk && !use_v1 && std::getenv("DLCO_JACOBI_NO_MW") == nullptr) {
        // many CUs, one wave per block pair; the grid barrier's words sit behind the work image and are cleared per call
        JmwDev g;
        g.T = T; g.ldt = ldt; g.n = n; g.evals = evals; g.Vout = V; g.ldv = ldv;
        g.ldc = jmw_ldc(n); g.nbe = jmw_nbe(n);
        const size_t ncol = 8 * (size_t)g.nbe;
        g.G = work; g.lam = work + ncol * g.ldc;
        static unsigned *sync_words = nullptr;                  // 64 words: counter, flag, per-sweep maxima
        if (!sync_words) DLCO_HIP(hipMalloc((void **)&sync_words, 64 * sizeof(unsigned)));
        DLCO_HIP(hipMemsetAsync(sync_words, 0, 64 * sizeof(unsigned), s));
        g.sync = sync_words; g.sweeps_out = sweeps_out; g.stop_cos = stop_cos; g.lam_cut = lam_cut;
        const size_t lds = ((size_t)16 * g.ldc + 32) * sizeof(float);
        static bool attr_mw = false;
        if (!attr_mw) {
            DLCO_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(jacobi_mw_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024));
            attr_mw = true;
        }
        hipLaunchKernelGGL(jacobi_mw_kernel, dim3(g.nbe / 2), dim3(64), lds, s, g);
    } else if (n <= JACOBI_LDS_MAX_N) {
        // 8 lanes per pair: 512 threads cover 64 pairs per pass, 1024 threads 128 (n > 128: one pass per round)
        const bool wide = n > 128;
        const size_t lds = ((size_t)n * ldc + (wide ? 16 : 8) + 4 + n + 4) * sizeof(float);
        static bool attr_set = false;
        if (!attr_set) {
            DLCO_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(jacobi_lds_kernel<512>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024));
            DLCO_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(jacobi_lds_kernel<1024>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024));
            attr_set = true;
        }
        if (wide) hipLaunchKernelGGL(jacobi_lds_kernel<1024>, dim3(1), dim3(1024), lds, s, T, ldt, n, ldc, evals, V, ldv, work, sweeps_out, stop_cos, lam_cut);
        else hipLaunchKernelGGL(jacobi_lds_kernel<512>, dim3(1), dim3(512), lds, s, T, ldt, n, ldc, evals, V, ldv, work, sweeps_out, stop_cos, lam_cut);
    } else if (n <= 256) {
        hipLaunchKernelGGL(jacobi_gmem_kernel<1024>, dim3(1), dim3(1024), 0, s, T, ldt, n, ldc, evals, V, ldv, work, sweeps_out, stop_cos, lam_cut);
    } else {
        hipLaunchKernelGGL(jacobi_gmem_kernel<512>, dim3(1), dim3(512), 0, s, T, ldt, n, ldc, evals, V, ldv, work, sweeps_out, stop_cos, lam_cut);
    }
    DLCO_HIP(hipGetLastError());
}

}  // namespace dlco
