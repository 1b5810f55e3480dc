// kernels_gemm.hip — generic fp32 GEMM on v_mfma_f32_32x32x2_f32 (gfx950).
//
// One templated kernel serves every GEMM-shaped piece of the pj-learn path that is
// not the fused SYRK (kernels_syrk.hip): projection W*X^T, subspace products Q*H,
// Gram / Rayleigh matrices, basis rotations.  64-wide wavefronts, four waves per
// workgroup in a 2x2 arrangement, each wave owning WM x WN tiles of 32x32, LDS tiles
// stored k-major so that both MFMA operand fragments are conflict-free ds_read_b32
// (lane l reads element (l&31) of row k = 2*kk + (l>>5)).  Exact fp32: the MFMA is a
// k-ordered fmaf chain (MI355X guide, "FP32-input MFMA").
#include "dlco_internal.hpp"

namespace dlco {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int BK = 16;
constexpr int PAD = 4;
constexpr int NT = 256;

struct OperandDev {
    const float *p;
    long ld;
    const int32_t *row_ids;
    const float *row_scale;
    int vec_ok;
    const int32_t *row_ids2;      // pair mode: operand row = p[row_ids[x]] - p[row_ids2[x]]
};

// up to 4 consecutive floats starting at src, elements at or beyond `limit - pos` are zero
__device__ __forceinline__ f32x4 load4(const float *src, int pos, int limit, int vec_ok)
{
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (vec_ok && pos + 3 < limit) {
        v = *reinterpret_cast<const f32x4 *>(src);
    } else {
        v[0] = src[0];
        if (pos + 1 < limit) v[1] = src[1];
        if (pos + 2 < limit) v[2] = src[2];
        if (pos + 3 < limit) v[3] = src[3];
    }
    return v;
}

struct GemmDev {
    int M, N, K;
    OperandDev A, B;
    float *C;
    long ldc;
    float alpha, beta, b1, b2;
    const float *E1, *E2;
    const int *k_dev;
    int k_chunk;      // K range per blockIdx.z
    float *slab;      // != nullptr: write raw partials to slab[z][M][N]
    int upper_only;
    const float *B2;  // != nullptr: a second product with the same A in the same launch, blockIdx.z == 1: C2 = A * B2 (no K split)
    float *C2;
};

// Stage one BK x BT tile of an operand into LDS (k-major image [BK][BT+PAD]).
// KM = memory rows run along K (contiguous along the M/N index), else along M/N.
template <int BT, bool KM>
__device__ __forceinline__ void load_tile(const OperandDev &op, int t0, int tmax, int k0, int kend,
                                          f32x4 (&regs)[BT / 64])
{
    const int tid = threadIdx.x;
#pragma unroll
    for (int u = 0; u < BT / 64; u++) {
        const int f = tid + NT * u;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (KM) {
            const int row = f / (BT / 4), c4 = f % (BT / 4);
            const int k = k0 + row, col = t0 + c4 * 4;
            if (k < kend && col < tmax) {
                const long mr = op.row_ids ? (long)op.row_ids[k] : (long)k;
                v = load4(op.p + mr * op.ld + col, col, tmax, op.vec_ok);
                if (op.row_ids2) v -= load4(op.p + (long)op.row_ids2[k] * op.ld + col, col, tmax, op.vec_ok);
                if (op.row_scale) {
                    const float sc = op.row_scale[k];
                    v *= sc;
                }
            }
        } else {
            const int row = f / (BK / 4), kq = f % (BK / 4);
            const int t = t0 + row, k = k0 + kq * 4;
            if (t < tmax && k < kend) {
                const long mr = op.row_ids ? (long)op.row_ids[t] : (long)t;
                v = load4(op.p + mr * op.ld + k, k, kend, op.vec_ok);
                if (op.row_ids2) v -= load4(op.p + (long)op.row_ids2[t] * op.ld + k, k, kend, op.vec_ok);
            }
        }
        regs[u] = v;
    }
}

template <int BT, bool KM>
__device__ __forceinline__ void store_tile(float (*lds)[BT + PAD], const f32x4 (&regs)[BT / 64])
{
    const int tid = threadIdx.x;
#pragma unroll
    for (int u = 0; u < BT / 64; u++) {
        const int f = tid + NT * u;
        if (KM) {
            const int row = f / (BT / 4), c4 = f % (BT / 4);
            *reinterpret_cast<f32x4 *>(&lds[row][c4 * 4]) = regs[u];
        } else {
            const int row = f / (BK / 4), kq = f % (BK / 4);
            lds[kq * 4 + 0][row] = regs[u][0];
            lds[kq * 4 + 1][row] = regs[u][1];
            lds[kq * 4 + 2][row] = regs[u][2];
            lds[kq * 4 + 3][row] = regs[u][3];
        }
    }
}

template <int WM, int WN, bool A_KM, bool B_KM>
__global__ __launch_bounds__(NT) void gemm_kernel(GemmDev g)
{
    constexpr int BM = 64 * WM, BN = 64 * WN;
    __shared__ __attribute__((aligned(16))) float As[2][BK][BM + PAD];
    __shared__ __attribute__((aligned(16))) float Bs[2][BK][BN + PAD];

    const int bi = blockIdx.y, bj = blockIdx.x;
    if (g.upper_only && bj * BN + BN <= bi * BM) return;     // tile entirely below the diagonal
    const int i0 = bi * BM, j0 = bj * BN;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 1, wn = wave & 1;

    int kz = blockIdx.z;
    if (g.B2) {
        if (kz == 1) { g.B.p = g.B2; g.C = g.C2; }
        kz = 0;
    }
    int kbeg = kz * g.k_chunk;
    int kend = min(g.K, kbeg + g.k_chunk);
    if (g.k_dev) kend = min(kend, *g.k_dev);

    f32x16 acc[WM][WN];
#pragma unroll
    for (int a = 0; a < WM; a++)
#pragma unroll
        for (int b = 0; b < WN; b++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[a][b][r] = 0.f;

    const int nk = kend > kbeg ? (kend - kbeg + BK - 1) / BK : 0;
    f32x4 ra[BM / 64], rb[BN / 64];
    if (nk > 0) {
        load_tile<BM, A_KM>(g.A, i0, g.M, kbeg, kend, ra);
        load_tile<BN, B_KM>(g.B, j0, g.N, kbeg, kend, rb);
        store_tile<BM, A_KM>(As[0], ra);
        store_tile<BN, B_KM>(Bs[0], rb);
    }
    __syncthreads();
    for (int kt = 0; kt < nk; kt++) {
        const int buf = kt & 1;
        if (kt + 1 < nk) {
            load_tile<BM, A_KM>(g.A, i0, g.M, kbeg + (kt + 1) * BK, kend, ra);
            load_tile<BN, B_KM>(g.B, j0, g.N, kbeg + (kt + 1) * BK, kend, rb);
        }
        const int lr = lane & 31, lk = lane >> 5;
#pragma unroll
        for (int kk = 0; kk < BK / 2; kk++) {
            float af[WM], bf[WN];
#pragma unroll
            for (int a = 0; a < WM; a++) af[a] = As[buf][2 * kk + lk][wm * 32 * WM + a * 32 + lr];
#pragma unroll
            for (int b = 0; b < WN; b++) bf[b] = Bs[buf][2 * kk + lk][wn * 32 * WN + b * 32 + lr];
#pragma unroll
            for (int a = 0; a < WM; a++)
#pragma unroll
                for (int b = 0; b < WN; b++)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a], bf[b], acc[a][b], 0, 0, 0);
        }
        if (kt + 1 < nk) {
            store_tile<BM, A_KM>(As[buf ^ 1], ra);
            store_tile<BN, B_KM>(Bs[buf ^ 1], rb);
        }
        __syncthreads();
    }

    // epilogue: C/D layout of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    const int lc = lane & 31, lh = lane >> 5;
#pragma unroll
    for (int a = 0; a < WM; a++)
#pragma unroll
        for (int b = 0; b < WN; b++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int i = i0 + wm * 32 * WM + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                const int j = j0 + wn * 32 * WN + b * 32 + lc;
                if (i >= g.M || j >= g.N) continue;
                if (g.slab) {
                    g.slab[((long)blockIdx.z * g.M + i) * g.N + j] = acc[a][b][r];
                    continue;
                }
                if (g.upper_only && j < i) continue;
                const long idx = (long)i * g.ldc + j;
                float o = g.alpha * acc[a][b][r];
                if (g.beta != 0.f) o += g.beta * g.C[idx];
                if (g.E1) o += g.b1 * g.E1[idx];
                if (g.E2) o += g.b2 * g.E2[idx];
                g.C[idx] = o;
                if (g.upper_only && j > i) g.C[(long)j * g.ldc + i] = o;
            }
}

__global__ void splitk_reduce_kernel(const float *slab, int split, int M, int N, float *C, long ldc, float alpha,
                                     float beta, const float *E1, float b1, const float *E2, float b2)
{
    // 16 lanes (one DPP row) per output element: lane z adds slices z, z+16, ... in order, the 16
    // partial sums are combined by a fixed butterfly, so the result does not depend on timing.
    const long total = (long)M * N;
    const int zl = threadIdx.x & 15;
    for (long e = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 4; e < total; e += ((long)gridDim.x * blockDim.x) >> 4) {
        const int i = (int)(e / N), j = (int)(e % N);
        float s = 0.f;
        for (int z = zl; z < split; z += 16) s += slab[(long)z * total + e];
        s += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(s), 0xB1, 0xF, 0xF, true));
        s += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(s), 0x4E, 0xF, 0xF, true));
        s += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(s), 0x141, 0xF, 0xF, true));
        s += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(s), 0x140, 0xF, 0xF, true));
        if (zl != 0) continue;
        const long idx = (long)i * ldc + j;
        float o = alpha * s;
        if (beta != 0.f) o += beta * C[idx];
        if (E1) o += b1 * E1[idx];
        if (E2) o += b2 * E2[idx];
        C[idx] = o;
    }
}

// Few slices over a large output (the tracker's split-bf16 products: 4-32 slices of m x F): one
// thread per four consecutive outputs, slices added in order, 16-byte accesses throughout.
__global__ __launch_bounds__(256) void splitk_reduce_wide_kernel(const float *slab, int split, int M, int N, float *C,
                                                                 long ldc, float alpha, float beta, const float *E1,
                                                                 float b1, const float *E2, float b2)
{
    const long total = (long)M * N, q = N >> 2;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < (long)M * q; e += (long)gridDim.x * blockDim.x) {
        const int i = (int)(e / q), j = (int)(e % q) * 4;
        const float *p = slab + (long)i * N + j;
        f32x4 s = *reinterpret_cast<const f32x4 *>(p);
        for (int z = 1; z < split; z++) s += *reinterpret_cast<const f32x4 *>(p + (long)z * total);
        const long idx = (long)i * ldc + j;
        f32x4 o = alpha * s;
        if (beta != 0.f) o += beta * *reinterpret_cast<const f32x4 *>(C + idx);
        if (E1) o += b1 * *reinterpret_cast<const f32x4 *>(E1 + idx);
        if (E2) o += b2 * *reinterpret_cast<const f32x4 *>(E2 + idx);
        *reinterpret_cast<f32x4 *>(C + idx) = o;
    }
}

// -----------------------------------------------------------------------------------------
// Skinny product of the eigen tracker: out[M][N] = alpha * X[M][K] * G[K][N] + b1*E1 + b2*E2
// with M <= 128 (MT = ceil(M/32) row tiles), G row-major.  One workgroup owns 32 output
// columns for ALL rows; its four waves split K four ways and never synchronise inside the
// K loop (each wave stages its own X tile through a private LDS region, G fragments go
// straight from global memory to registers: 32 lanes read one 128-byte row segment).
// The four partial accumulators are summed through LDS in wave order at the end, so the
// result is deterministic and no split-K slab or second launch is needed.
// -----------------------------------------------------------------------------------------
struct SkinnyDev {
    int M, N, K;
    const float *X;
    long ldx;
    const float *G;
    long ldg;
    float *C;
    long ldc;
    float alpha, b1, b2;
    const float *E1, *E2;
    int xvec_ok;
};

constexpr int SKW = 8;                           // waves per skinny workgroup (K split 8 ways)
constexpr int SKT = 64 * SKW;

// FAST: N % 32 == 0, K % (SKW*BK) == 0, X 16-byte aligned with at least 32*MT allocated rows:
// no bounds checks anywhere in the K loop (rows beyond M are computed and discarded).
template <int MT, bool FAST>
__global__ __launch_bounds__(SKT) void skinny_kernel(SkinnyDev g)
{
    constexpr int MP = 32 * MT;                  // padded rows
    constexpr int LDA = MP + PAD;
    __shared__ __attribute__((aligned(16))) float As[SKW][2][BK][LDA];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lc = lane & 31, lh = lane >> 5;
    const int n0 = blockIdx.x * 32;
    const int col = n0 + lc;
    const bool col_ok = col < g.N;

    // K range of this wave (multiple of BK)
    int kq = ((g.K + SKW - 1) / SKW + BK - 1) / BK * BK;
    const int kbeg = wave * kq;
    const int kend = min(g.K, kbeg + kq);
    const int nk = kend > kbeg ? (kend - kbeg + BK - 1) / BK : 0;

    f32x16 acc[MT];
#pragma unroll
    for (int a = 0; a < MT; a++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[a][r] = 0.f;

    // X tile loader: lane -> (row = lane/4 + 16u, k-quad = lane%4), float4 along k
    constexpr int AU = MP / 16;
    f32x4 ra[AU];
    float rb0[BK / 2], rb1[BK / 2];              // G fragments of the next two K tiles (HBM latency cover)
    auto load_a = [&](int k0) {
#pragma unroll
        for (int u = 0; u < AU; u++) {
            const int row = (lane >> 2) + 16 * u, k = k0 + (lane & 3) * 4;
            if (FAST) {
                ra[u] = *reinterpret_cast<const f32x4 *>(g.X + (long)row * g.ldx + k);
                continue;
            }
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (row < g.M && k < kend) {
                const float *src = g.X + (long)row * g.ldx + k;
                if (g.xvec_ok && k + 3 < kend) v = *reinterpret_cast<const f32x4 *>(src);
                else {
                    v[0] = src[0];
                    if (k + 1 < kend) v[1] = src[1];
                    if (k + 2 < kend) v[2] = src[2];
                    if (k + 3 < kend) v[3] = src[3];
                }
            }
            ra[u] = v;
        }
    };
    auto load_b = [&](int k0, float (&rb)[BK / 2]) {
#pragma unroll
        for (int kk = 0; kk < BK / 2; kk++) {
            const int k = k0 + 2 * kk + lh;
            if (FAST) rb[kk] = g.G[(long)k * g.ldg + col];
            else rb[kk] = (col_ok && k < kend) ? g.G[(long)k * g.ldg + col] : 0.f;
        }
    };
    auto store_a = [&](int buf) {
#pragma unroll
        for (int u = 0; u < AU; u++) {
            const int row = (lane >> 2) + 16 * u, kq4 = (lane & 3) * 4;
            As[wave][buf][kq4 + 0][row] = ra[u][0];
            As[wave][buf][kq4 + 1][row] = ra[u][1];
            As[wave][buf][kq4 + 2][row] = ra[u][2];
            As[wave][buf][kq4 + 3][row] = ra[u][3];
        }
    };

    if (nk > 0) {
        load_a(kbeg);
        load_b(kbeg, rb0);
        load_b(kbeg + BK, rb1);
        store_a(0);
    }
    // one K tile: consume the fragments in `bcur`, refill them for tile kt+2, stage X tile kt+1
    auto tile = [&](int kt, float (&bcur)[BK / 2]) {
        const int buf = kt & 1;
        float bf[BK / 2];
#pragma unroll
        for (int kk = 0; kk < BK / 2; kk++) bf[kk] = bcur[kk];
        // (clamping these prefetch indices to make the loop branch-free was measured 13 % SLOWER here)
        if (kt + 1 < nk) load_a(kbeg + (kt + 1) * BK);
        if (kt + 2 < nk) load_b(kbeg + (kt + 2) * BK, bcur);
#pragma unroll
        for (int kk = 0; kk < BK / 2; kk++) {
#pragma unroll
            for (int a = 0; a < MT; a++) {
                const float af = As[wave][buf][2 * kk + lh][a * 32 + lc];
                acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(af, bf[kk], acc[a], 0, 0, 0);
            }
        }
        if (kt + 1 < nk) store_a(buf ^ 1);       // private region: ordered by this wave's own LDS counter
    };
    for (int kt = 0; kt < nk; kt += 2) {
        tile(kt, rb0);
        if (kt + 1 < nk) tile(kt + 1, rb1);
    }

    // ---- combine the K-partials in wave order through LDS (reusing the staging space) ----------
    __syncthreads();
    float *red = &As[0][0][0][0];
    static_assert((SKW - 1) * MT * 16 * 64 <= SKW * 2 * BK * LDA, "reduction scratch does not fit");
    if (wave > 0) {
#pragma unroll
        for (int a = 0; a < MT; a++)
#pragma unroll
            for (int r = 0; r < 16; r++) red[(((wave - 1) * MT + a) * 16 + r) * 64 + lane] = acc[a][r];
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int a = 0; a < MT; a++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                float s = acc[a][r];
#pragma unroll
                for (int w = 0; w < SKW - 1; w++) s += red[((w * MT + a) * 16 + r) * 64 + lane];
                const int i = a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (i < g.M && col_ok) {
                    const long idx = (long)i * g.ldc + col;
                    float o = g.alpha * s;
                    if (g.E1) o += g.b1 * g.E1[idx];
                    if (g.E2) o += g.b2 * g.E2[idx];
                    g.C[idx] = o;
                }
            }
    }
}

template <int WM, int WN>
void launch(const GemmDev &g, bool akm, bool bkm, dim3 grid, hipStream_t s)
{
    if (akm && bkm) hipLaunchKernelGGL((gemm_kernel<WM, WN, true, true>), grid, dim3(NT), 0, s, g);
    else if (akm && !bkm) hipLaunchKernelGGL((gemm_kernel<WM, WN, true, false>), grid, dim3(NT), 0, s, g);
    else if (!akm && bkm) hipLaunchKernelGGL((gemm_kernel<WM, WN, false, true>), grid, dim3(NT), 0, s, g);
    else hipLaunchKernelGGL((gemm_kernel<WM, WN, false, false>), grid, dim3(NT), 0, s, g);
}

inline int vec_ok(const GemmOperand &o) { return (o.ld % 4 == 0) && ((reinterpret_cast<uintptr_t>(o.p) & 15) == 0); }

}  // namespace

size_t gemm_slab_floats(int M, int N, int split_k) { return split_k > 1 ? (size_t)split_k * M * N : 0; }

// C = alpha * sum_z slab[z] + beta*C + b1*E1 + b2*E2 over `split` [M][N] partial slabs (fixed order)
void splitk_reduce_f32(const float *slab, int split, int M, int N, float *C, long ldc, float alpha, float beta,
                       const float *E1, float b1, const float *E2, float b2, hipStream_t s)
{
    const long total = (long)M * N;
    auto al16 = [](const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    if (split <= 32 && total >= 65536 && N % 4 == 0 && ldc % 4 == 0 && al16(slab) && al16(C) && (!E1 || al16(E1)) &&
        (!E2 || al16(E2))) {
        long blocks = (total / 4 + 255) / 256;
        if (blocks > 4096) blocks = 4096;
        hipLaunchKernelGGL(splitk_reduce_wide_kernel, dim3((unsigned)blocks), dim3(256), 0, s, slab, split, M, N, C, ldc, alpha,
                           beta, E1, b1, E2, b2);
        DLCO_HIP(hipGetLastError());
        return;
    }
    long blocks = (total * 16 + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, s, slab, split, M, N, C, ldc, alpha, beta,
                       E1, b1, E2, b2);
    DLCO_HIP(hipGetLastError());
}

bool skinny_product_f32(const float *X, long ldx, int M, int x_rows_alloc, const float *G, long ldg, int N, int K,
                        float alpha, float *C, long ldc, const float *E1, float b1, const float *E2, float b2, hipStream_t s)
{
    if (M < 1 || M > 128 || N < 1) return false;
    SkinnyDev g;
    g.M = M; g.N = N; g.K = K;
    g.X = X; g.ldx = ldx; g.G = G; g.ldg = ldg; g.C = C; g.ldc = ldc;
    g.alpha = alpha; g.E1 = E1; g.b1 = b1; g.E2 = E2; g.b2 = b2;
    g.xvec_ok = (ldx % 4 == 0) && ((reinterpret_cast<uintptr_t>(X) & 15) == 0);
    const dim3 grid((N + 31) / 32), block(SKT);
    const int mt = (M + 31) / 32;
    const bool fast = g.xvec_ok && (N % 32 == 0) && (K % (SKW * BK * 2) == 0) && x_rows_alloc >= 32 * mt;
    if (fast) {
        if (mt == 1) hipLaunchKernelGGL((skinny_kernel<1, true>), grid, block, 0, s, g);
        else if (mt == 2) hipLaunchKernelGGL((skinny_kernel<2, true>), grid, block, 0, s, g);
        else if (mt == 3) hipLaunchKernelGGL((skinny_kernel<3, true>), grid, block, 0, s, g);
        else hipLaunchKernelGGL((skinny_kernel<4, true>), grid, block, 0, s, g);
    } else {
        if (mt == 1) hipLaunchKernelGGL((skinny_kernel<1, false>), grid, block, 0, s, g);
        else if (mt == 2) hipLaunchKernelGGL((skinny_kernel<2, false>), grid, block, 0, s, g);
        else if (mt == 3) hipLaunchKernelGGL((skinny_kernel<3, false>), grid, block, 0, s, g);
        else hipLaunchKernelGGL((skinny_kernel<4, false>), grid, block, 0, s, g);
    }
    DLCO_HIP(hipGetLastError());
    return true;
}

void gemm_f32(const GemmArgs &a, hipStream_t s)
{
    if (a.M <= 0 || a.N <= 0) return;
    DLCO_CHECK(a.A.p && a.B.p && (a.C || a.raw_slab), -2, "gemm_f32: null operand");
    DLCO_CHECK(!(a.upper_only && (a.M != a.N || a.split_k > 1)), -2, "gemm_f32: upper_only needs M==N, split_k==1");
    GemmDev g;
    g.M = a.M; g.N = a.N; g.K = a.K;
    g.A = {a.A.p, a.A.ld, a.A.row_ids, a.A.kmajor ? a.A.row_scale : nullptr, vec_ok(a.A), a.A.row_ids2};
    g.B = {a.B.p, a.B.ld, a.B.row_ids, a.B.kmajor ? a.B.row_scale : nullptr, vec_ok(a.B), a.B.row_ids2};
    g.C = a.C; g.ldc = a.ldc;
    g.alpha = a.alpha; g.beta = a.beta; g.E1 = a.E1; g.b1 = a.b1; g.E2 = a.E2; g.b2 = a.b2;
    g.k_dev = a.k_dev;
    g.upper_only = a.upper_only ? 1 : 0;
    int split = a.split_k < 1 ? 1 : a.split_k;
    int chunk = (a.K + split - 1) / split;
    chunk = ((chunk + BK - 1) / BK) * BK;
    if (chunk < BK) chunk = BK;
    split = a.K > 0 ? (a.K + chunk - 1) / chunk : 1;
    g.k_chunk = chunk;
    g.slab = nullptr;
    if (split > 1 || a.raw_slab) {
        DLCO_CHECK(a.slab != nullptr, -2, "gemm_f32: split_k > 1 needs a slab workspace");
        g.slab = a.slab;
    }
    if (a.split_out) *a.split_out = split;
    g.B2 = a.B2; g.C2 = a.C2;
    if (a.B2) {
        DLCO_CHECK(a.C2 && split == 1 && !a.raw_slab && !a.upper_only, -2, "gemm_f32: a twin product needs C2 and no K split");
        split = 2;                                                 // grid z = the two products
    }
    // tile choice: the small dimension decides; a launch that would leave most of the 256 CUs
    // without a workgroup takes 64-wide tiles instead (the tracker's m x F x m products)
    bool small_m = a.M <= 64 || a.small_m_tiles, small_n = a.N <= 64 || a.small_n_tiles;
    auto n_wg = [&](bool sm, bool sn) {
        return (long)((a.M + (sm ? 63 : 127)) / (sm ? 64 : 128)) * ((a.N + (sn ? 63 : 127)) / (sn ? 64 : 128)) * split;
    };
    if (!a.upper_only && std::getenv("DLCO_GEMM_BIG_TILES") == nullptr) {
        if (!small_m && n_wg(small_m, small_n) < 256) small_m = true;
        if (!small_n && n_wg(small_m, small_n) < 256) small_n = true;
    }
    if (small_m && small_n) {
        dim3 grid((a.N + 63) / 64, (a.M + 63) / 64, split);
        launch<1, 1>(g, a.A.kmajor, a.B.kmajor, grid, s);
    } else if (small_m) {
        dim3 grid((a.N + 127) / 128, (a.M + 63) / 64, split);
        launch<1, 2>(g, a.A.kmajor, a.B.kmajor, grid, s);
    } else if (small_n) {
        dim3 grid((a.N + 63) / 64, (a.M + 127) / 128, split);
        launch<2, 1>(g, a.A.kmajor, a.B.kmajor, grid, s);
    } else {
        dim3 grid((a.N + 127) / 128, (a.M + 127) / 128, split);
        launch<2, 2>(g, a.A.kmajor, a.B.kmajor, grid, s);
    }
    DLCO_HIP(hipGetLastError());
    if (split > 1 && !a.raw_slab && !a.B2)
        splitk_reduce_f32(a.slab, split, a.M, a.N, a.C, a.ldc, a.alpha, a.beta, a.E1, a.b1, a.E2, a.b2, s);
}

}  // namespace dlco
