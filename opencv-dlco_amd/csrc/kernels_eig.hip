// kernels_eig.hip — vector kernels and the CholQR building block of the subspace tracker
// (E1/E2 of the pj-learn step, src/pj-learn.cpp:434-490), for gfx950.  The Rayleigh-Ritz
// eigensolver itself is in kernels_jacobi.hip.
#include "dlco_internal.hpp"

#include <type_traits>

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

// CholQR building block: Gram matrix M (n <= 160, lower triangle read) -> Linv = L^-1 with M = L L^T, on one workgroup.
//
// Right-looking elimination without pivoting on the augmented matrix [M | I]: step j subtracts (A_ij / A_jj) * row j
// from every row i > j.  The left half turns into the Schur complements (pivot d_j = A_jj at step j), the right half into
// U^-1 of M = U D U^T, and L^-1 = D^-1/2 U^-1.  A row whose pivot drops below rel_thresh * M_jj lies (to fp32 accuracy) in
// the span of the rows before it: it is marked dead, eliminated from nothing, and its row of Linv is zero.  (Round 1's
// kernel did this one pivot per workgroup barrier - ~1.2 us each, 114 us at n = 96; it left the library in round 4.)
// ---- blocked: 32 pivots per workgroup barrier instead of one --------------------------------------------
// The elimination of [M | I] proceeds in blocks of 32 pivots:
//   1. the 32 x 32 diagonal block goes to LDS and ONE wave eliminates it in registers (lane i = row i of
//      [D | I]; the pivot row reaches the other lanes through v_readlane, no barrier inside the 32
//      steps), which yields the pivots d, the dead flags and the unit lower triangular T = L1^-1 of
//      the block;
//   2. every thread then applies the block's 32 steps at once to its entries: the block's rows become
//      P = T * rows, the multipliers of the rows below are F = (A[:,K] T^T) / d (the Schur complement
//      is symmetric, so U_KK^-1 = T^T D^-1), and the trailing update is the rank-32 product F * P.
// The matrix stays in registers (thread (ty, tx) holds rows ty + 32 r, columns tx + 32 c of [M | I]); LDS only
// carries the block's rows, columns, T, P and F.  Four barriers per block of 32 pivots.  The multipliers are
// a[j] * rcp(d) instead of a[j] / d (CholQR2's second pass absorbs that).  Fast path for M = I + E with
// |E|_F^2 <= 1e-6 (the second pass of CholQR2): M^-1/2 = I - E/2, no elimination.
// NBM = blocks of 32 rows the kernel is built for: 4 (n <= 128) or 5 (n <= 160: the block of the rank ~128 workload in ONE
// orthonormalisation panel instead of two panels with a Gram-Schmidt step between them; 50 registers of matrix per thread).
constexpr int C2_T = 1024;
__host__ __device__ constexpr int c2_pld(int nbm) { return 64 * nbm + 4; }            // row stride of the P images (floats)
__host__ __device__ constexpr size_t c2_lds_floats(int nbm)
{
    return (size_t)2 * 32 * 33 + 2 * 32 * c2_pld(nbm) + 2 * 32 * nbm * 33 + 32 * nbm * 3 + 64 + 2 * nbm * nbm * 64;
}

__device__ __forceinline__ float readlane_f(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }

template <int NBM>
__global__ __launch_bounds__(C2_T) void chol_inv2_kernel(const float *M, long ldm, int n, float rel_thresh, float *Linv, long ldl,
                                                         int *dead)
{
    extern __shared__ __attribute__((aligned(16))) float c2[];
    float (*Dblk)[33] = reinterpret_cast<float (*)[33]>(c2);
    float (*Tblk)[33] = reinterpret_cast<float (*)[33]>(c2 + 32 * 33);
    constexpr int C2_PLD = c2_pld(NBM), NR = 32 * NBM;         // NR: rows / columns the registers cover
    float (*Praw)[C2_PLD] = reinterpret_cast<float (*)[C2_PLD]>(c2 + 2 * 32 * 33);
    float (*Pbuf)[C2_PLD] = reinterpret_cast<float (*)[C2_PLD]>(c2 + 2 * 32 * 33 + 32 * C2_PLD);
    float (*Acol)[33] = reinterpret_cast<float (*)[33]>(c2 + 2 * 32 * 33 + 2 * 32 * C2_PLD);
    float (*Fbuf)[33] = reinterpret_cast<float (*)[33]>(c2 + 2 * 32 * 33 + 2 * 32 * C2_PLD + NR * 33);
    float *diag0 = c2 + 2 * 32 * 33 + 2 * 32 * C2_PLD + 2 * NR * 33;
    float *dpiv = diag0 + NR;
    int *deadf = reinterpret_cast<int *>(dpiv + NR);
    float *dinv = dpiv + 2 * NR;                               // [32] of the current block, then scratch [16]
    float *stash = dinv + 64;                                  // [2 NBM^2][64]: the eliminating wave's own entries
    const int t = threadIdx.x, tx = t & 31, ty = t >> 5;
    const int nb = (n + 31) >> 5;

    // reg[r][c]: row 32r + ty; c < NBM: column 32c + tx of M, c >= NBM: column 32(c-NBM) + tx of the identity
    float reg[NBM][2 * NBM];
#pragma unroll
    for (int r = 0; r < NBM; r++)
#pragma unroll
        for (int c = 0; c < NBM; c++) {
            const int i = 32 * r + ty, k = 32 * c + tx;
            reg[r][c] = (i < n && k < n) ? M[(long)max(i, k) * ldm + min(i, k)] : ((i == k) ? 1.f : 0.f);
            reg[r][NBM + c] = (i == k) ? 1.f : 0.f;
        }
    if (t < NR) { diag0[t] = t < n ? M[(long)t * ldm + t] : 1.f; deadf[t] = 0; dpiv[t] = 1.f; }
    // Fast path (second pass of CholQR2): M = I + E with |E|_F^2 <= 1e-6 -> M^-1/2 = I - E/2 
    {
        float e2 = 0.f;
#pragma unroll
        for (int r = 0; r < NBM; r++)
#pragma unroll
            for (int c = 0; c < NBM; c++) { const float e = reg[r][c] - ((32 * r + ty == 32 * c + tx) ? 1.f : 0.f); e2 += e * e; }
        e2 = wsum(e2);
        if ((t & 63) == 0) dinv[32 + (t >> 6)] = e2;
    }
    __syncthreads();
    {
        float tot = 0.f;
        for (int w = 0; w < C2_T / 64; w++) tot += dinv[32 + w];
        if (tot <= 1e-6f) {                                   // workgroup-uniform (false for NaN)
#pragma unroll
            for (int r = 0; r < NBM; r++)
#pragma unroll
                for (int c = 0; c < NBM; c++) {
                    const int i = 32 * r + ty, k = 32 * c + tx;
                    if (i < n && k < n) Linv[(long)i * ldl + k] = ((i == k) ? 1.5f : 0.f) - 0.5f * reg[r][c];
                }
            if (t < n) dead[t] = 0;
            return;
        }
    }

    // one block of 32 pivots; KB is a compile-time constant so that every index into reg[][] is static
    auto block = [&](auto KBc) {
        constexpr int kb = decltype(KBc)::value;
        if (kb < nb) {                                         // workgroup-uniform
            // ---- 1. the diagonal block of the current Schur complement -------------------------------
            Dblk[ty][tx] = reg[kb][kb];
            __syncthreads();
            // ---- 2. one wave eliminates [D | I] in registers ----------------------------------------------
            if (t < 64) {
                // this wave's own matrix entries wait in LDS meanwhile: the elimination needs 64 registers
                // per lane and must not spill (a spilled register costs a trip to memory per pivot)
#pragma unroll
                for (int r = 0; r < NBM; r++)
#pragma unroll
                    for (int c = 0; c < 2 * NBM; c++) stash[(r * 2 * NBM + c) * 64 + t] = reg[r][c];
                const int i = t & 31;
                float a[32], x[32];
#pragma unroll
                for (int k = 0; k < 32; k++) { a[k] = Dblk[i][k]; x[k] = (i == k) ? 1.f : 0.f; }
                const float d0_mine = diag0[32 * kb + i];
#pragma unroll
                for (int j = 0; j < 32; j++) {
                    const float piv = readlane_f(a[j], j);
                    const float d0 = readlane_f(d0_mine, j);
                    const bool dd = !(piv > rel_thresh * d0) || !(d0 > 0.f);
                    const float rinv = dd ? 0.f : __builtin_amdgcn_rcpf(piv);
                    if (t == 0) { dpiv[32 * kb + j] = piv; deadf[32 * kb + j] = dd ? 1 : 0; dinv[j] = rinv; }
                    const float f = (i > j) ? a[j] * rinv : 0.f;
#pragma unroll
                    for (int k = j + 1; k < 32; k++) a[k] -= f * readlane_f(a[k], j);
#pragma unroll
                    for (int k = 0; k <= j; k++) x[k] -= f * readlane_f(x[k], j);
                }
                if (t < 32) {
#pragma unroll
                    for (int k = 0; k < 32; k++) Tblk[i][k] = x[k];
                }
#pragma unroll
                for (int r = 0; r < NBM; r++)
#pragma unroll
                    for (int c = 0; c < 2 * NBM; c++) reg[r][c] = stash[(r * 2 * NBM + c) * 64 + t];
            }
            // ---- 3a. the block's rows (columns still needed) and the block's columns of the rows below ---
#pragma unroll
            for (int c = 0; c < NBM; c++) {
                if (c > kb && c < nb) Praw[ty][32 * c + tx] = reg[kb][c];            // M part right of the block
                if (c <= kb) Praw[ty][NR + 32 * c + tx] = reg[kb][NBM + c];           // identity part up to the block
            }
#pragma unroll
            for (int r = 0; r < NBM; r++)
                if (r > kb && r < nb) Acol[32 * r + ty][tx] = reg[r][kb];
            __syncthreads();
            // ---- 3b. P = T * rows (this thread's row of the block), F = (A[:,K] T^T) / d --------------------
            {
                float pacc[2 * NBM];
#pragma unroll
                for (int c = 0; c < 2 * NBM; c++) pacc[c] = 0.f;
#pragma unroll 2
                for (int j = 0; j < 32; j++) {
                    const float tj = Tblk[ty][j];                     // zero for j > ty
#pragma unroll
                    for (int c = 0; c < NBM; c++) {
                        if (c > kb && c < nb) pacc[c] += tj * Praw[j][32 * c + tx];
                        if (c <= kb) pacc[NBM + c] += tj * Praw[j][NR + 32 * c + tx];
                    }
                }
#pragma unroll
                for (int c = 0; c < NBM; c++) {
                    if (c > kb && c < nb) { reg[kb][c] = pacc[c]; Pbuf[ty][32 * c + tx] = pacc[c]; }
                    if (c <= kb) { reg[kb][NBM + c] = pacc[NBM + c]; Pbuf[ty][NR + 32 * c + tx] = pacc[NBM + c]; }
                }
                const float di = dinv[tx];
#pragma unroll
                for (int r = 0; r < NBM; r++) {
                    if (r > kb && r < nb) {
                        float facc = 0.f;
#pragma unroll 4
                        for (int l = 0; l < 32; l++) facc += Tblk[tx][l] * Acol[32 * r + ty][l];
                        Fbuf[32 * r + ty][tx] = facc * di;
                    }
                }
            }
            __syncthreads();
            // ---- 4. trailing update of the rows below the block: rank-32 product F * P ---------------------
            if (kb + 1 < nb) {
#pragma unroll 2
                for (int j = 0; j < 32; j++) {
                    float fr[NBM], pc[2 * NBM];
#pragma unroll
                    for (int r = 0; r < NBM; r++) fr[r] = (r > kb && r < nb) ? Fbuf[32 * r + ty][j] : 0.f;
#pragma unroll
                    for (int c = 0; c < NBM; c++) {
                        pc[c] = (c > kb && c < nb) ? Pbuf[j][32 * c + tx] : 0.f;
                        pc[NBM + c] = (c <= kb) ? Pbuf[j][NR + 32 * c + tx] : 0.f;
                    }
#pragma unroll
                    for (int r = 0; r < NBM; r++) {
                        if (r > kb) {
#pragma unroll
                            for (int c = 0; c < NBM; c++) {
                                if (c > kb) reg[r][c] -= fr[r] * pc[c];
                                if (c <= kb) reg[r][NBM + c] -= fr[r] * pc[NBM + c];
                            }
                        }
                    }
                }
            }
        }
    };
    block(std::integral_constant<int, 0>{});
    block(std::integral_constant<int, 1>{});
    block(std::integral_constant<int, 2>{});
    if constexpr (NBM > 3) block(std::integral_constant<int, 3>{});
    if constexpr (NBM > 4) block(std::integral_constant<int, NBM - 1>{});
    __syncthreads();
    // Linv = D^-1/2 U^-1 (lower triangular); dead rows and the padding are zero
#pragma unroll
    for (int r = 0; r < NBM; r++) {
        const int i = 32 * r + ty;
        if (i < n) {
            const float sc = deadf[i] ? 0.f : rsqrtf(dpiv[i]);
#pragma unroll
            for (int c = 0; c < NBM; c++) {
                const int k = 32 * c + tx;
                if (k < n) Linv[(long)i * ldl + k] = (k <= i) ? reg[r][NBM + c] * sc : 0.f;
            }
        }
    }
    if (t < n) dead[t] = deadf[t];
}

}  // namespace

template <int NBM>
static void launch_chol_inv2(const float *M, long ldm, int n, float rel_thresh, float *Linv, long ldl, int *dead, hipStream_t s)
{
    ensure_dynamic_lds(reinterpret_cast<const void *>(chol_inv2_kernel<NBM>), (int)(c2_lds_floats(NBM) * sizeof(float)));
    hipLaunchKernelGGL(chol_inv2_kernel<NBM>, dim3(1), dim3(C2_T), c2_lds_floats(NBM) * sizeof(float), s, M, ldm, n, rel_thresh, Linv, ldl, dead);
}

void chol_inverse128(const float *M, long ldm, int n, float rel_thresh, float *Linv, long ldl, int *dead, hipStream_t s)
{
    DLCO_CHECK(n >= 1 && n <= CHOL_INV_MAX_N, -2, "chol_inverse128: n out of range");
    if (n <= 96) launch_chol_inv2<3>(M, ldm, n, rel_thresh, Linv, ldl, dead, s);      // (the steady-state block: 18 matrix registers per thread instead of 32: 44.2 -> 39.7 us at n = 96)
    else if (n <= 128) launch_chol_inv2<4>(M, ldm, n, rel_thresh, Linv, ldl, dead, s);
    else launch_chol_inv2<5>(M, ldm, n, rel_thresh, Linv, ldl, dead, s);
    DLCO_HIP(hipGetLastError());
}

// Copies n floats into pinned host memory and then raises a sequence number there (system-scope release): the
// host polls the number instead of paying a stream synchronisation for a read-back of a few hundred bytes.
__global__ __launch_bounds__(256) void publish_block_kernel(const float *src, float *dst_host, int n, unsigned *flag_host, unsigned seq,
                                                            const int *extra_dev, int *extra_host)
{
    for (int i = threadIdx.x; i < n; i += 256) dst_host[i] = src[i];
    if (extra_dev && threadIdx.x == 0) *extra_host = *extra_dev;      // one more word the caller wants back (no copy of its own)
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(flag_host, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// residual_kernel and publish_block_kernel in one launch: the workgroup that finishes LAST (a ticket counter, left at zero
// again) copies the block - which holds the residuals the other workgroups have just written, hence the agent-scope
// release / acquire pair around the ticket and the agent-scope loads - and raises the sequence number.
// em.W != nullptr: the workgroup of row i also writes that row of W - W[nw-1-i] = sqrt(cscale (theta_i - mu)) * x_i for the nw rows
// with theta > mu (ascending order and scaling of src/pj-learn.cpp:480-487; the arithmetic of emit_w_kernel, bit for bit), the
// rows behind them unscaled in place when em.guards - and the scale into em.wscale: the row is in flight anyway, and the pass that
// turns out to be the last one of an update has then emitted W without a launch of its own (an earlier pass's rows are simply
// written over).
__global__ __launch_bounds__(256) void residual_publish_kernel(const float *X, const float *Y, long ld, const float *theta, int m, int F,
                                                               float *res, unsigned *ticket, const float *src, float *dst_host, int n,
                                                               unsigned *flag_host, unsigned seq, const int *extra_dev, int *extra_host,
                                                               ResidualEmit em)
{
    __shared__ float part[4];
    __shared__ int last;
    __shared__ int cnt[4];
    const int i = blockIdx.x;
    const float th = theta[i];
    float s = 0.f;
    if (em.W && (F & 3) == 0 && (ld & 3) == 0 && (em.ldw & 3) == 0) {
        int c = 0;
        for (int k = threadIdx.x; k < m; k += blockDim.x) c += theta[k] > em.mu ? 1 : 0;
        c = (int)wsum((float)c);                                  // m <= 4096: exact in fp32
        if ((threadIdx.x & 63) == 0) cnt[threadIdx.x >> 6] = c;
        __syncthreads();
        const int nw = cnt[0] + cnt[1] + cnt[2] + cnt[3];
        const bool mine = i < nw || em.guards;
        const float sc = i < nw ? sqrtf(em.cscale * (th - em.mu)) : 1.0f;
        f32x4 *w4 = reinterpret_cast<f32x4 *>(em.W + (long)(i < nw ? nw - 1 - i : i) * em.ldw);
        if (em.wscale && threadIdx.x == 0) em.wscale[i] = sc;
        const f32x4 *y4 = reinterpret_cast<const f32x4 *>(Y + (long)i * ld), *x4 = reinterpret_cast<const f32x4 *>(X + (long)i * ld);
        for (int f = threadIdx.x; f < F / 4; f += blockDim.x) {
            const f32x4 x = x4[f];
            const f32x4 d = y4[f] - th * x;
            s += d[0] * d[0] + d[1] * d[1] + d[2] * d[2] + d[3] * d[3];
            if (mine) { f32x4 v = x; v[0] *= sc; v[1] *= sc; v[2] *= sc; v[3] *= sc; w4[f] = v; }
        }
    } else if ((F & 3) == 0 && (ld & 3) == 0) {
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
    if (threadIdx.x == 0) {
        __hip_atomic_store(res + i, sqrtf(part[0] + part[1] + part[2] + part[3]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        last = t == gridDim.x - 1 ? 1 : 0;
        if (last) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (!last) return;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    for (int k = threadIdx.x; k < n; k += 256) dst_host[k] = __hip_atomic_load(src + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (extra_dev && threadIdx.x == 0) *extra_host = __hip_atomic_load(extra_dev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(flag_host, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

void residual_norms_publish(const float *X, const float *Y, long ld, const float *theta, int m, int F, float *res, unsigned *ticket,
                            const float *src, float *dst_host, int n, unsigned *flag_host, unsigned seq, hipStream_t s,
                            const int *extra_dev, int *extra_host, const ResidualEmit *emit, bool *emitted)
{
    DLCO_CHECK(m > 0, -2, "residual_norms_publish: empty block");
    ResidualEmit em;
    if (emit && emit->W && (F & 3) == 0 && (ld & 3) == 0 && (emit->ldw & 3) == 0 && (reinterpret_cast<uintptr_t>(emit->W) & 15) == 0) em = *emit;
    if (emitted) *emitted = em.W != nullptr;
    hipLaunchKernelGGL(residual_publish_kernel, dim3(m), dim3(256), 0, s, X, Y, ld, theta, m, F, res, ticket, src, dst_host, n, flag_host,
                       seq, extra_dev, extra_host, em);
    DLCO_HIP(hipGetLastError());
}

void publish_block(const float *src, float *dst_host, int n, unsigned *flag_host, unsigned seq, hipStream_t s, const int *extra_dev,
                   int *extra_host)
{
    hipLaunchKernelGGL(publish_block_kernel, dim3(1), dim3(256), 0, s, src, dst_host, n, flag_host, seq, extra_dev, extra_host);
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
