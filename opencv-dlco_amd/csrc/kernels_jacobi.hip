// kernels_jacobi.hip — symmetric eigendecomposition of the small Rayleigh-Ritz matrices of the
// subspace tracker (n = block rows, tens to a few hundred), for gfx950.
//
// One-sided (Hestenes) Jacobi on G = T + sigma*I, sigma from a Gershgorin bound so that G is
// positive definite: the method orthogonalises the columns of G by plane rotations; at
// convergence column j equals lambda'_j * v_j, so the eigenvectors are the normalised columns
// and no separate rotation accumulator is kept (half the LDS traffic and footprint).
//   * a column pair is owned by 8 lanes (half a DPP row); the scalar rotation arithmetic is
//     replicated 8 times instead of 64;
//   * only the cross product x.y is reduced per pair (three DPP adds); the squared norms are
//     tracked and follow the rotations analytically (|x'|^2 = |x|^2 - t x.y,
//     |y'|^2 = |y|^2 + t x.y), refreshed exactly at the start of every sweep;
//   * v_rcp / v_rsq / v_sqrt (1 ulp) instead of the IEEE division and square-root sequences:
//     a rotation only has to be orthogonal to rounding, its angle is re-measured every sweep.
// Three kernels: n <= 160 one workgroup with the matrix in LDS (the steady state of a run),
// 160 < n <= 2048 one wave per block pair on many CUs (the start-up transient), beyond that one
// workgroup on an L2-resident image.  All of them return the eigenvectors as the ROWS of V.
#include "dlco_internal.hpp"

#include <cstdio>
#include <vector>

#include <algorithm>

namespace dlco {

namespace {

constexpr int LP = 8;                 // lanes per column pair

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
        Vout[(long)rank[j] * ldv + i] = G[(long)j * ldc + i] * inv[j];
    }
}

template <int JT>
__global__ __launch_bounds__(JT) void jacobi_gmem_kernel(const float *T, long ldt, int n, int ldc, float *evals, float *Vout,
                                                         long ldv, float *work, int *sweeps_out, float stop_cos, float lam_cut)
{
    __shared__ float red[JT / 64 + 4];
    float *G = work, *scratch = work + (size_t)n * ldc;
    jacobi_body<JT, 4>(G, ldc, T, ldt, n, evals, Vout, ldv, scratch, sweeps_out, red, scratch + 3 * (size_t)n, stop_cos, lam_cut);
}


// ---- n <= 160: blocks of eight columns, one WAVE per pair of blocks, the matrix in LDS ------------------------------
// The tournament is played between BLOCKS of eight columns: a wave takes two blocks and rotates all 64 cross pairs
// without leaving its own part of the matrix - eight lanes per pair, the eight columns of block A stay in registers for
// the whole meeting, the columns of block B pass through them from LDS in eight inner rounds (pair (A_g, B_(g+k) mod 8)
// in round k) - so a sweep has nb - 1 barriers (nb = n / 8 blocks) instead of n - 1, and the LDS traffic of a rotation
// is one column in and out instead of two.  The 28 pairs inside each block are rotated once per sweep (seven
// wave-private rounds through LDS) before the blocks start to meet.  Blocks are paired by the circle method, computed
// from the round number: columns never move in LDS, a single image suffices (160 columns: 100 KB).
// Column j lives at G + j * 32 E; lane l of a pair's eight owns the 16-byte chunks l + 8 e (e < E).  E is odd, so
// consecutive columns start 32 banks apart and the eight-column reads of a wave are conflict-free.
//
// Rotations in scaled ("fast Givens") form.  A column is kept as  true column = s_j * stored column : the rotation
// x' = cs (x - t y), y' = cs (y + t x)  becomes
//     stored x' = x - (t s_y / s_x) y ,   stored y' = y + (t cs^2 s_x / s_y) x' ,   s_x' = cs s_x ,  s_y' = s_y / cs
// (done in place, x first: t1 t2 = -t^2 moves the factor 1 + t^2 of the second line into the scale, so no copy of the
// old x is needed) - one packed fma per pair of elements and column instead of a multiply and an fma; the scales ride
// with the tracked squared norms (of the TRUE columns) in an eight-byte record per column.  1/sqrt 2 <= cs <= 1: a
// column that arrives from LDS with its scale outside [2^-20, 2^20] is multiplied out before it is used (a scale moves
// by at most 2^+-4 between two such checks), and every sweep starts by multiplying all of them out (with the exact
// norms).  The columns are held as float2 pairs so that the dot product and the updates are written directly in the
// packed instructions the hardware has (v_pk_mul / v_pk_fma).  A rotation step in which none of the wave's eight pairs
// is above the rotation threshold touches nothing.
//
// What surrounds the sweeps costs as much as a sweep if done naively (26 - 59 us of a 90 - 160 us call were measured
// for it): T is read ONCE, row by row with coalesced 16-byte loads all issued before the first is used (row j becomes
// column j: T is symmetric up to the rounding of the product that made it, and a one-sided method needs no more - the
// antisymmetric part perturbs the result like any other rounding error of that size), the Gershgorin bound comes from
// the LDS image, the eigenvalue ranks are counted from LDS, and the eigenvectors leave as ROWS of V, every store a
// contiguous run.
constexpr int JBLK_MAX_N = 160;
inline int jblk_chunks(int n) { return n <= 32 ? 1 : (n <= 96 ? 3 : 5); }
inline int jblk_waves(int n) { const int nb = (n + 7) / 8; return (nb + (nb & 1)) / 2; }
inline size_t jblk_lds_bytes(int n) { const int ncol = 16 * jblk_waves(n); return ((size_t)ncol * 32 * jblk_chunks(n) + 5 * ncol + 32) * sizeof(float); }

typedef float f32x2 __attribute__((ext_vector_type(2)));

// TR (developer aid, DLCO_JACOBI_TRACE=file): lane 0 of wave 1 stamps s_memtime at four points of every inner rotation step
template <int E, bool TR = false>
__global__ __launch_bounds__(640) void jacobi_blk_kernel(const float *T, long ldt, int n, float *evals, float *Vout, long ldv,
                                                         int *sweeps_out, float stop_cos, float lam_cut, float rot_tol,
                                                         unsigned long long *trace = nullptr)
{
    int tri = 0;
    bool tron = false;                                         // stamps only while the blocks meet (four per inner step)
    auto stamp = [&]() {
        if (TR && tron && threadIdx.x == 64 && tri < 4096) trace[tri++] = __builtin_amdgcn_s_memtime();
    };
    constexpr int LDC = 32 * E, P = 2 * E;
    extern __shared__ __attribute__((aligned(16))) float sh[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nthr = blockDim.x, nw = nthr >> 6;               // nw = block pairs = (number of blocks, made even) / 2
    const int nbe = 2 * nw, ncol = 8 * nbe, m = nbe - 1;
    float *G = sh;
    f32x2 *rec = reinterpret_cast<f32x2 *>(sh + (size_t)ncol * LDC);      // {squared norm of the true column, scale}
    float *lam = reinterpret_cast<float *>(rec + ncol), *inv = lam + ncol;
    int *perm = reinterpret_cast<int *>(inv + ncol);
    float *red = reinterpret_cast<float *>(perm + ncol);

    // ---- row j of T -> column j of the image (zero padded), all loads of a wave in flight together ------------------
    {
        constexpr int RMAX = 16;                                  // a wave owns ncol / nw = 16 columns
        const bool vec = (ldt & 3) == 0 && (reinterpret_cast<size_t>(T) & 15) == 0;
        const int nq = LDC / 4;                                   // 16-byte chunks of a column: <= 40 lanes busy
        f32x4 v[RMAX];
#pragma unroll
        for (int q = 0; q < RMAX; q++) {
            const int j = wave + nw * q;
            v[q] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (j < n && lane < nq) {
                const int i = 4 * lane;
                const float *p = T + (long)j * ldt + i;
                if (vec && i + 3 < n) v[q] = *reinterpret_cast<const f32x4 *>(p);
                else {
                    if (i < n) v[q][0] = p[0];
                    if (i + 1 < n) v[q][1] = p[1];
                    if (i + 2 < n) v[q][2] = p[2];
                    if (i + 3 < n) v[q][3] = p[3];
                }
            }
        }
        float rmax = 0.f;
#pragma unroll
        for (int q = 0; q < RMAX; q++) {
            const int j = wave + nw * q;
            if (lane < nq) *reinterpret_cast<f32x4 *>(G + (size_t)j * LDC + 4 * lane) = v[q];
            rmax = fmaxf(rmax, wsum(fabsf(v[q][0]) + fabsf(v[q][1]) + fabsf(v[q][2]) + fabsf(v[q][3])));
        }
        if (lane == 0) red[wave] = rmax;
    }
    for (int j = tid; j < ncol; j += nthr) rec[j] = f32x2{0.f, 1.f};
    __syncthreads();
    // ---- shift: sigma = 1.01 * max_j sum_i |T_ji| + tiny  (Gershgorin), added to the diagonal ---------------------------
    float sigma = 0.f;
    for (int w = 0; w < nw; w++) sigma = fmaxf(sigma, red[w]);
    sigma = 1.01f * sigma + 1e-30f;
    const float cut2 = lam_cut + sigma > 0.f ? (lam_cut + sigma) * (lam_cut + sigma) : 0.f;
    for (int j = tid; j < n; j += nthr) G[(size_t)j * LDC + j] += sigma;
    __syncthreads();

    const float tol = rot_tol, small = 9.5367431640625e-7f, big = 1048576.f;     // 2^-20, 2^20
    const int g = lane >> 3, l = lane & 7;
    float off_max = 0.f;
    auto load_col = [&](int col, f32x2 (&x)[P]) {
        const float *p = G + (size_t)col * LDC + 4 * l;
#pragma unroll
        for (int e = 0; e < E; e++) {
            const f32x4 v = *reinterpret_cast<const f32x4 *>(p + 32 * e);
            x[2 * e] = f32x2{v[0], v[1]};
            x[2 * e + 1] = f32x2{v[2], v[3]};
        }
    };
    auto store_col = [&](int col, const f32x2 (&x)[P]) {
        float *p = G + (size_t)col * LDC + 4 * l;
#pragma unroll
        for (int e = 0; e < E; e++) *reinterpret_cast<f32x4 *>(p + 32 * e) = f32x4{x[2 * e][0], x[2 * e][1], x[2 * e + 1][0], x[2 * e + 1][1]};
    };
    auto multiply_out = [&](f32x2 (&x)[P], float &s) {
#pragma unroll
        for (int p = 0; p < P; p++) x[p] *= s;
        s = 1.f;
    };
    // one rotation of the pair (x: true squared norm a, scale sa, 1 / sa; y: b, sb, 1 / sb); false when no pair of the wave
    // rotates (nothing changed).  In place, x first:  x' = x - (t sb / sa) y  (scale cs sa),  y' = y + (t cs^2 sa / sb) x'
    // (scale sb / cs).  The step is a dependent chain (the y column has just come from another lane group through LDS and
    // goes to the next one), so what does not depend on the dot product - the reciprocals of the scales, 1 / sqrt(a b) - is
    // computed beside the LDS loads, and the tangent comes from  t = sign(d) 2c / (|d| + sqrt(d^2 + 4 c^2)), d = b - a:
    // two transcendentals in a row instead of three; x' needs t only, y' waits for 1 / (1 + t^2).
    auto rotate = [&](f32x2 (&x)[P], f32x2 (&y)[P], float &a, float &sa, float &rsa, float &b, float &sb, float rsb) -> bool {
        const float ab = a * b;
        float rab = ab > 0.f ? __builtin_amdgcn_rsqf(ab) : 0.f;
        float ssab = sa * sb, r1 = sb * rsa, r2 = sa * rsb, d = b - a;
        asm volatile("" : "+v"(rab), "+v"(r1), "+v"(r2));           // evaluated here, beside the loads, not after the branch below
        f32x2 acc0 = x[0] * y[0], acc1 = x[1] * y[1];
#pragma unroll
        for (int p = 2; p < P; p += 2) { acc0 += x[p] * y[p]; acc1 += x[p + 1] * y[p + 1]; }
        acc0 += acc1;
        const float c = row8_sum(acc0[0] + acc0[1]) * ssab;
        if (TR) { asm volatile("" :: "v"(c)); stamp(); }
        const float off = fabsf(c) * rab;
        off_max = fmaxf(off_max, fmaxf(a, b) > cut2 ? off : 0.f);
        const bool rot = off > tol;
        if (__builtin_amdgcn_ballot_w64(rot) == 0ull) return false;
        float t = 0.f;
        if (rot) {
            const float c2 = 2.f * c;
            const float den = fabsf(d) + __builtin_amdgcn_sqrtf(d * d + c2 * c2);
            t = __uint_as_float(__float_as_uint(c2 * __builtin_amdgcn_rcpf(den)) ^ (__float_as_uint(d) & 0x80000000u));
        }
        const float t1 = -t * r1;
#pragma unroll
        for (int p = 0; p < P; p++) x[p] += t1 * y[p];
        const float w = 1.f + t * t;
        const float cs2 = __builtin_amdgcn_rcpf(w), cs = __builtin_amdgcn_rsqf(w);
        const float t2 = t * cs2 * r2;
#pragma unroll
        for (int p = 0; p < P; p++) y[p] += t2 * x[p];
        const float tc = t * c;
        a -= tc;
        b += tc;
        sa *= cs;
        rsa *= w * cs;
        sb *= w * cs;
        return true;
    };

    int sweep = 0;
    for (; sweep < 40; sweep++) {
        off_max = 0.f;
        f32x2 x[P], y[P];
        // this wave's sixteen columns: scales multiplied out, exact squared norms
#pragma unroll
        for (int q = 0; q < 2; q++) {
            const int col = 16 * wave + 8 * q + g;
            load_col(col, x);
            float sc = rec[col][1];
            if (sc != 1.f) { multiply_out(x, sc); store_col(col, x); }
            f32x2 acc = x[0] * x[0];
#pragma unroll
            for (int p = 1; p < P; p++) acc += x[p] * x[p];
            const float s = row8_sum(acc[0] + acc[1]);
            if (l == 0) rec[col] = f32x2{s, 1.f};
        }
        // ---- the pairs inside blocks 2 wave (lane groups 0-3) and 2 wave + 1 (groups 4-7): seven wave-private rounds ---
        {
            const int base = 16 * wave + 8 * (g >> 2), k = g & 3;
            for (int r = 0; r < 7; r++) {
                const int pa = k == 0 ? 7 : (r + k) % 7, pb = k == 0 ? r : (r - k + 7) % 7;
                const int ca = base + pa, cb = base + pb;
                load_col(ca, x);
                load_col(cb, y);
                const f32x2 ra = rec[ca], rb = rec[cb];
                float a = ra[0], sa = ra[1], b = rb[0], sb = rb[1];
                if (sa < small || sa > big) multiply_out(x, sa);
                if (sb < small || sb > big) multiply_out(y, sb);
                float rsa = __builtin_amdgcn_rcpf(sa);
                if (rotate(x, y, a, sa, rsa, b, sb, __builtin_amdgcn_rcpf(sb))) {
                    store_col(ca, x);
                    store_col(cb, y);
                    if (l == 0) { rec[ca] = f32x2{a, sa}; rec[cb] = f32x2{b, sb}; }
                }
            }
        }
        __syncthreads();
        // ---- the blocks meet: round r pairs block m with block r, and (r + k) mod m with (r - k) mod m ---------------
        tron = true;
        for (int r = 0; r < m; r++) {
            const int ba = wave == 0 ? m : (r + wave) % m, bb = wave == 0 ? r : (r - wave + m) % m;
            const int ca = 8 * ba + g;
            load_col(ca, x);
            const f32x2 ra = rec[ca];
            float a = ra[0], sa = ra[1];
            if (sa < small || sa > big) multiply_out(x, sa);
            float rsa = __builtin_amdgcn_rcpf(sa);
            bool touched = false;
#pragma unroll 1
            for (int kk = 0; kk < 8; kk++) {
                stamp();
                const int cb = 8 * bb + ((g + kk) & 7);
                load_col(cb, y);
                const f32x2 rb = rec[cb];
                float b = rb[0], sb = rb[1];
                if (TR) { __builtin_amdgcn_s_waitcnt(0xc07f); stamp(); }          // lgkmcnt(0): the partner column is in
                // (a wave-uniform branch: written as a plain `if`, hipcc if-converted the rare rescaling into six packed
                // multiplies and twelve selects executed in EVERY step of this issue-bound chain)
                if (__builtin_amdgcn_ballot_w64(sb < small || sb > big) != 0ull) {
                    asm volatile("" ::: "memory");                                 // (keeps the branch a branch)
                    if (sb < small || sb > big) multiply_out(y, sb);
                }
                if (rotate(x, y, a, sa, rsa, b, sb, __builtin_amdgcn_rcpf(sb))) {
                    if (TR) { asm volatile("" :: "v"(y[0][0])); stamp(); }
                    store_col(cb, y);
                    if (l == 0) rec[cb] = f32x2{b, sb};
                    touched = true;
                } else if (TR) stamp();
            }
            if (touched) {
                store_col(ca, x);
                if (l == 0) rec[ca] = f32x2{a, sa};
            }
            __syncthreads();
        }
        tron = false;
        off_max = wmax(off_max);
        if (lane == 0) red[wave] = off_max;
        __syncthreads();
        float mx = 0.f;
        for (int w = 0; w < nw; w++) mx = fmaxf(mx, red[w]);
        __syncthreads();
        if (mx <= stop_cos) { sweep++; break; }              // see jacobi_body for why no verification sweep follows
    }
    if (tid == 0 && sweeps_out) *sweeps_out = sweep;

    // ---- eigenvalues lambda = s |g| - sigma, eigenvectors g / |g| (the rows of V), sorted descending -------------------
    for (int j = wave; j < n; j += nw) {
        float d = 0.f;
        for (int i = lane; i < LDC; i += 64) { const float v = G[(size_t)j * LDC + i]; d += v * v; }
        d = wsum(d);
        if (lane == 0) { const float nr = sqrtf(d); lam[j] = nr * rec[j][1] - sigma; inv[j] = nr > 0.f ? 1.f / nr : 0.f; }
    }
    __syncthreads();
    for (int j = tid; j < n; j += nthr) {
        const float me = lam[j];
        int rk = 0;
        for (int k = 0; k < n; k++) {
            const float o = lam[k];
            rk += (o > me || (o == me && k < j)) ? 1 : 0;
        }
        perm[rk] = j;
        evals[rk] = me;
    }
    __syncthreads();
    for (int c = wave; c < n; c += nw) {
        const int j = perm[c];
        const float sc = inv[j];
        for (int i = lane; i < n; i += 64) Vout[(long)c * ldv + i] = G[(size_t)j * LDC + i] * sc;
    }
}

template <int E>
void launch_jacobi_blk(const float *T, long ldt, int n, float *evals, float *V, long ldv, int *sweeps_out, float stop_cos, float lam_cut,
                       hipStream_t s)
{
    ensure_dynamic_lds(reinterpret_cast<const void *>(jacobi_blk_kernel<E>), 160 * 1024 - 1024);
    // column cosine below which a pair is not rotated
    static const float rot_tol = std::getenv("DLCO_JACOBI_ROT") ? (float)std::atof(std::getenv("DLCO_JACOBI_ROT")) : 3e-6f;
    static const char *trace_path = std::getenv("DLCO_JACOBI_TRACE");
    static int calls = 0;
    if (trace_path && E == 3 && ++calls == 40) {                // one call in the steady state of a bench run
        unsigned long long *buf = nullptr;
        DLCO_HIP(hipMalloc((void **)&buf, 4096 * 8));
        DLCO_HIP(hipMemsetAsync(buf, 0, 4096 * 8, s));
        ensure_dynamic_lds(reinterpret_cast<const void *>(jacobi_blk_kernel<E, true>), 160 * 1024 - 1024);
        hipLaunchKernelGGL((jacobi_blk_kernel<E, true>), dim3(1), dim3(64 * jblk_waves(n)), jblk_lds_bytes(n), s, T, ldt, n, evals, V, ldv,
                           sweeps_out, stop_cos, lam_cut, rot_tol, buf);
        DLCO_HIP(hipStreamSynchronize(s));
        std::vector<unsigned long long> h(4096);
        DLCO_HIP(hipMemcpy(h.data(), buf, 4096 * 8, hipMemcpyDeviceToHost));
        (void)hipFree(buf);
        if (FILE *f = std::fopen(trace_path, "w")) {
            std::fprintf(f, "# n = %d; per inner rotation step of wave 1: top, partner column in, dot product reduced, updates done (s_memtime)\n", n);
            for (int i = 0; i + 3 < 4096 && h[i]; i += 4) std::fprintf(f, "%llu %llu %llu %llu\n", h[i], h[i + 1], h[i + 2], h[i + 3]);
            std::fclose(f);
        }
        return;
    }
    hipLaunchKernelGGL(jacobi_blk_kernel<E>, dim3(1), dim3(64 * jblk_waves(n)), jblk_lds_bytes(n), s, T, ldt, n, evals, V, ldv,
                       sweeps_out, stop_cos, lam_cut, rot_tol);
}

// ---- 160 < n <= 2048: the same blocks, one workgroup per block pair, MANY CUs ----------------------------------------
// Blocks of several hundred rows only occur in the start-up transient of a run (from W = 0 the rank overshoots to 300 -
// 900 before it settles), but there they are most of the time: 45 % of the first hundred steps went into this solver
// when a workgroup was ONE wave that fetched its sixteen columns one after the other (a dependent L2 round trip each)
// and rotated 400-row columns with eight lanes.  The block tournament needs no shared LDS image: the matrix G = T +
// sigma I stays in global memory (L2 / Infinity Cache resident), the workgroup of a block pair brings the two blocks
// into its own LDS - by LDS-DMA, every piece of all sixteen columns in flight at once - rotates the 64 cross pairs
// there (eight inner rounds of eight pairs; the rows of a pair are dealt to the four waves, 32 lanes per pair, the
// partial cross products meet in LDS) and writes the sixteen columns back; between two rounds of the tournament all
// workgroups meet at a grid barrier.  n / 16 workgroups on as many CUs, n / 8 - 1 barriers per sweep.
//   * Hand-off through global memory between CUs on different XCDs: every storing wave drains its stores and issues an
//     agent-scope release, the workgroup meets, thread 0 arrives at a monotonic counter and polls it; after the poll
//     every wave issues an agent-scope acquire, then plain loads.
//   * The grid is at most 128 workgroups of 256 threads: always co-resident.  Every spin is bounded; a barrier that
//     does not complete sets *sweeps_out = -1 and every workgroup leaves.
// T is taken as symmetric up to rounding (row j becomes column j), as in the one-workgroup kernel above.
struct JmwDev {
    const float *T; long ldt; int n;
    float *evals, *Vout; long ldv;
    float *G;                     // [ncol][ldc] work image of the columns, ldc = n rounded up to 32
    float *lam;                   // [ncol] + [ncol] (inverse norms)
    unsigned *sync;               // [0] barrier counter, [1] timeout flag, [2 + sweep] max cosine of a sweep, [50] Gershgorin bound (float bits); zeroed by the host
    int *sweeps_out;
    float stop_cos, lam_cut;
    int ldc, nbe;
};

constexpr int JMW_T = 256;

// all threads of the workgroup call it; false: the barrier timed out somewhere
__device__ __forceinline__ bool jmw_grid_sync(unsigned *sync, unsigned nwg, unsigned &target, int *flag_lds)
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");          // this wave's stores are visible to the other CUs ...
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                            // ... and so are the other waves' of this workgroup
    target += nwg;
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(sync, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        int ok = 1;
        while (__hip_atomic_load(sync, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(2);
            if (++spins > (1u << 22) || __hip_atomic_load(sync + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
                __hip_atomic_store(sync + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ok = 0;
                break;
            }
        }
        *flag_lds = ok;
    }
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    return *flag_lds != 0;
}

__global__ __launch_bounds__(JMW_T) void jacobi_mw_kernel(JmwDev g)
{
    extern __shared__ __attribute__((aligned(16))) float sh[];
    const int n = g.n, ldc = g.ldc, nq = ldc >> 2;             // 16-byte chunks per column
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = lane >> 3, l = lane & 7;
    const int q0 = l + 8 * wave;                               // this thread's chunks of a pair's columns: q0, q0 + 32, ...
    const int k = blockIdx.x, nwg = gridDim.x, nbe = g.nbe, m = nbe - 1, ncol = 8 * nbe;
    float *cols = sh;                                           // sixteen columns: block A 0-7, block B 8-15, column c at c * ldc
    float *nrm = sh + 16 * (size_t)ldc;                         // [16] tracked squared norms
    float *red = nrm + 16;                                      // [2][8][4] partial cross products of a step
    float *wred = red + 64;                                     // [4] per-wave maxima
    int *flag = reinterpret_cast<int *>(wred + 4);
    unsigned target = 0;

    // ---- sigma = 1.01 * max_i sum_j |T_ij| + tiny (Gershgorin): rows dealt to the waves of all workgroups ------------------
    {
        float rmax = 0.f;
        for (int i = 4 * k + wave; i < n; i += 4 * nwg) {
            float sacc = 0.f;
            for (int j = lane; j < n; j += 64) sacc += fabsf(g.T[(long)i * g.ldt + j]);
            rmax = fmaxf(rmax, wsum(sacc));
        }
        if (lane == 0) atomicMax(g.sync + 50, __float_as_uint(rmax));   // non-negative floats order like their bits
    }
    if (!jmw_grid_sync(g.sync, nwg, target, flag)) { if (k == 0 && tid == 0) *g.sweeps_out = -1; return; }
    const float sigma = 1.01f * __uint_as_float(__hip_atomic_load(g.sync + 50, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) + 1e-30f;
    const float cut2 = g.lam_cut + sigma > 0.f ? (g.lam_cut + sigma) * (g.lam_cut + sigma) : 0.f;
    // ---- G = T + sigma I, zero padded: this workgroup writes the sixteen columns of blocks 2k, 2k+1 -----------------------
    for (int c = wave; c < 16; c += 4) {
        const int j = 16 * k + c;
        for (int i = lane; i < ldc; i += 64) {
            float v = 0.f;
            if (i < n && j < n) v = g.T[(long)j * g.ldt + i] + (i == j ? sigma : 0.f);
            g.G[(size_t)j * ldc + i] = v;
        }
    }
    if (!jmw_grid_sync(g.sync, nwg, target, flag)) { if (k == 0 && tid == 0) *g.sweeps_out = -1; return; }

    const float tol = 3e-6f;
    // two blocks (sixteen columns) global -> LDS, every piece in flight at once; then the exact squared norms
    auto load_blocks = [&](int ba, int bb) {
        for (int c = wave; c < 16; c += 4) {
            const float *src = g.G + (size_t)(8 * (c < 8 ? ba : bb) + (c & 7)) * ldc;
            float *dst = cols + (size_t)c * ldc;
            for (int p0 = 0; p0 < ldc; p0 += 256) {
                if (p0 + 4 * lane < ldc)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + p0 + 4 * lane),
                                                     (__attribute__((address_space(3))) void *)(dst + p0), 16, 0, 0);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        for (int c = wave; c < 16; c += 4) {
            const f32x4 *col = reinterpret_cast<const f32x4 *>(cols + (size_t)c * ldc);
            float sq = 0.f;
            for (int q = lane; q < nq; q += 64) { const f32x4 v = col[q]; sq += dot4(v, v); }
            sq = wsum(sq);
            if (lane == 0) nrm[c] = sq;
        }
        __syncthreads();
    };
    auto store_blocks = [&](int ba, int bb) {
        for (int c = wave; c < 16; c += 4) {
            f32x4 *dst = reinterpret_cast<f32x4 *>(g.G + (size_t)(8 * (c < 8 ? ba : bb) + (c & 7)) * ldc);
            const f32x4 *col = reinterpret_cast<const f32x4 *>(cols + (size_t)c * ldc);
            for (int q = lane; q < nq; q += 64) dst[q] = col[q];
        }
    };
    float off_max = 0.f;
    int parity = 0;
    // one rotation step: the pair (columns cx, cy of the sixteen) of this thread's lane group; 32 lanes (8 per wave) per pair
    auto rotate_step = [&](int cx, int cy) {
        f32x4 *px = reinterpret_cast<f32x4 *>(cols + (size_t)cx * ldc), *py = reinterpret_cast<f32x4 *>(cols + (size_t)cy * ldc);
        float c = 0.f;
        for (int q = q0; q < nq; q += 32) c += dot4(px[q], py[q]);
        c = row8_sum(c);
        float *rp = red + (parity * 8 + grp) * 4;
        if (l == 0) rp[wave] = c;
        __syncthreads();
        c = (rp[0] + rp[1]) + (rp[2] + rp[3]);
        const float a = nrm[cx], b = nrm[cy];
        const float ab = a * b;
        const float off = ab > 0.f ? fabsf(c) * __builtin_amdgcn_rsqf(ab) : 0.f;
        off_max = fmaxf(off_max, fmaxf(a, b) > cut2 ? off : 0.f);
        float t = 0.f;
        if (off > tol) {
            float cs, sn;
            rotation(a, b, c, t, cs, sn);
            for (int q = q0; q < nq; q += 32) {
                const f32x4 x = px[q], y = py[q];
                px[q] = cs * x - sn * y;
                py[q] = sn * x + cs * y;
            }
        }
        __syncthreads();                                        // everybody has read the norms of this step
        if (wave == 0 && l == 0 && off > tol) { nrm[cx] = a - t * c; nrm[cy] = b + t * c; }
        parity ^= 1;
    };

    int sweep = 0;
    bool alive = true;
    for (; sweep < 40 && alive; sweep++) {
        off_max = 0.f;
        // ---- pairs inside blocks 2k (lane groups 0-3) and 2k+1 (groups 4-7): seven rounds ---------------------------------
        load_blocks(2 * k, 2 * k + 1);
        {
            const int base = grp < 4 ? 0 : 8, kk = grp & 3;
            for (int r = 0; r < 7; r++) {
                const int pa = kk == 0 ? 7 : (r + kk) % 7, pb = kk == 0 ? r : (r - kk + 7) % 7;
                rotate_step(base + pa, base + pb);
            }
        }
        __syncthreads();
        store_blocks(2 * k, 2 * k + 1);
        if (!jmw_grid_sync(g.sync, nwg, target, flag)) { alive = false; break; }
        // ---- the blocks meet: round r pairs block m with block r, and (r + k) mod m with (r - k) mod m -------------------
        for (int r = 0; r < m; r++) {
            const int ba = k == 0 ? m : (r + k) % m, bb = k == 0 ? r : (r - k + m) % m;
            load_blocks(ba, bb);
            for (int kk = 0; kk < 8; kk++) rotate_step(grp, 8 + ((grp + kk) & 7));
            __syncthreads();
            store_blocks(ba, bb);
            if (!jmw_grid_sync(g.sync, nwg, target, flag)) { alive = false; break; }
        }
        if (!alive) break;
        // ---- the sweep's largest cosine, over all workgroups --------------------------------------------------------------
        off_max = wmax(off_max);
        if (lane == 0) atomicMax(g.sync + 2 + sweep, __float_as_uint(off_max));      // non-negative floats order like their bits
        if (!jmw_grid_sync(g.sync, nwg, target, flag)) { alive = false; break; }
        const float mx = __uint_as_float(__hip_atomic_load(g.sync + 2 + sweep, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        if (mx <= g.stop_cos) { sweep++; break; }
    }
    if (!alive) { if (k == 0 && tid == 0) *g.sweeps_out = -1; return; }
    if (k == 0 && tid == 0 && g.sweeps_out) *g.sweeps_out = sweep;

    // ---- eigenvalues lambda = |g| - sigma, eigenvectors g / |g| (the rows of V), sorted descending ---------------------------
    float *lam = g.lam, *inv = g.lam + ncol;
    for (int c = wave; c < 16; c += 4) {
        const int j = 16 * k + c;
        if (j >= n) break;
        float d = 0.f;
        for (int i = lane; i < n; i += 64) { const float v = g.G[(size_t)j * ldc + i]; d += v * v; }
        d = wsum(d);
        if (lane == 0) { const float nr = sqrtf(d); lam[j] = nr - sigma; inv[j] = nr > 0.f ? 1.f / nr : 0.f; }
    }
    if (!jmw_grid_sync(g.sync, nwg, target, flag)) { if (k == 0 && tid == 0) *g.sweeps_out = -1; return; }
    for (int c = wave; c < 16; c += 4) {
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
        for (int i = lane; i < n; i += 64) g.Vout[(long)rk * g.ldv + i] = g.G[(size_t)j * ldc + i] * sc;
    }
}

constexpr int JMW_MAX_N = 2048;
inline int jmw_ldc(int n) { return (n + 31) & ~31; }
inline int jmw_nbe(int n) { const int nb = (n + 7) / 8; return nb + (nb & 1); }
inline size_t jmw_work_floats(int n) { const size_t ncol = 8 * (size_t)jmw_nbe(n); return ncol * jmw_ldc(n) + 3 * ncol + 64; }

inline int col_stride(int n) { return (n + 3) & ~3; }

}  // namespace

size_t jacobi_work_floats(int n) { return std::max((size_t)n * col_stride(n) + 4 * (size_t)n + 64, n <= JMW_MAX_N ? jmw_work_floats(n) : (size_t)0); }

// can the grid of the multi-workgroup kernel be resident all at once on this device?  (its barrier needs that; the
// bounded wait catches what this cannot know: another stream's kernels holding the CUs)
static bool jmw_fits(int nwg, size_t lds)
{
    struct Entry { int dev; size_t lds; long slots; };
    static std::vector<Entry> cache;
    static std::mutex mu;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return false;
    std::lock_guard<std::mutex> lock(mu);
    for (const Entry &e : cache)
        if (e.dev == dev && e.lds == lds) return e.slots >= nwg;
    int per_cu = 0;
    hipDeviceProp_t prop;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void *>(jacobi_mw_kernel), JMW_T, lds) != hipSuccess) return false;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return false;
    cache.push_back({dev, lds, (long)per_cu * prop.multiProcessorCount});
    return cache.back().slots >= nwg;
}

void jacobi_eigh(const float *T, long ldt, int n, float *evals, float *V, long ldv, float *work, int *sweeps_out,
                 hipStream_t s, float lam_cut, bool single_workgroup)
{
    DLCO_CHECK(n >= 1 && n <= 4096, -2, "jacobi_eigh: n out of range");
    // largest column cosine of a sweep below which the sweep is the last one (see jacobi_body)
    static const float stop_cos = std::getenv("DLCO_JACOBI_STOP") ? (float)std::atof(std::getenv("DLCO_JACOBI_STOP")) : 1e-3f;
    if (n <= JBLK_MAX_N) {
        const int e = jblk_chunks(n);
        if (e == 1) launch_jacobi_blk<1>(T, ldt, n, evals, V, ldv, sweeps_out, stop_cos, lam_cut, s);
        else if (e == 3) launch_jacobi_blk<3>(T, ldt, n, evals, V, ldv, sweeps_out, stop_cos, lam_cut, s);
        else launch_jacobi_blk<5>(T, ldt, n, evals, V, ldv, sweeps_out, stop_cos, lam_cut, s);
    } else if (n <= JMW_MAX_N && !single_workgroup &&
               (ensure_dynamic_lds(reinterpret_cast<const void *>(jacobi_mw_kernel), 160 * 1024 - 1024),
                jmw_fits(jmw_nbe(n) / 2, ((size_t)16 * jmw_ldc(n) + 128) * sizeof(float)))) {
        // many CUs, one workgroup per block pair; the grid barrier's words sit behind the work image and are cleared per call
        JmwDev g;
        g.T = T; g.ldt = ldt; g.n = n; g.evals = evals; g.Vout = V; g.ldv = ldv;
        g.ldc = jmw_ldc(n); g.nbe = jmw_nbe(n);
        const size_t ncol = 8 * (size_t)g.nbe;
        g.G = work; g.lam = work + ncol * g.ldc;
        // 64 words behind the image and the two vectors (counter, flag, per-sweep maxima), cleared per call: they belong to
        // the caller's workspace, so two trackers on two streams never share a barrier
        unsigned *sync_words = reinterpret_cast<unsigned *>(work + ncol * g.ldc + 2 * ncol);
        DLCO_HIP(hipMemsetAsync(sync_words, 0, 64 * sizeof(unsigned), s));
        // test hook: raise the barrier's give-up flag before the launch, every workgroup then leaves at its first grid barrier
        static const bool force_timeout = std::getenv("DLCO_TEST_JMW_TIMEOUT") != nullptr;
        if (force_timeout) DLCO_HIP(hipMemsetAsync(sync_words + 1, 1, 1, s));
        g.sync = sync_words; g.sweeps_out = sweeps_out; g.stop_cos = stop_cos; g.lam_cut = lam_cut;
        const size_t lds = ((size_t)16 * g.ldc + 128) * sizeof(float);
        hipLaunchKernelGGL(jacobi_mw_kernel, dim3(g.nbe / 2), dim3(JMW_T), lds, s, g);
    } else {
        hipLaunchKernelGGL(jacobi_gmem_kernel<512>, dim3(1), dim3(512), 0, s, T, ldt, n, col_stride(n), evals, V, ldv, work, sweeps_out, stop_cos, lam_cut);
    }
    DLCO_HIP(hipGetLastError());
}

}  // namespace dlco
