// kernels_bf16x2.hip — tracker filter product on the bf16 matrix cores with split operands.
//
//     out[M][F] = alpha * X[M][F] * G[F][F] + b1*E1 + b2*E2          (G symmetric, fp32 in HBM)
//
// The Chebyshev filter of the subspace tracker only has to ENRICH the block (the Rayleigh-Ritz
// step that follows uses the exact fp32 product and the convergence test is on fp32
// residuals), so its products may carry ~1e-5 relative error.  Each fp32 operand is split
// into two bf16 values, x = hi + lo (16 mantissa bits kept), and the product is formed as
// hi*hi + hi*lo + lo*hi with fp32 accumulation on v_mfma_f32_32x32x16_bf16: three bf16 MFMAs
// (3/16 of the fp32-MFMA time) in place of one fp32 pass, which moves the product from
// MFMA-bound to HBM-bound (one read of G).
//   * X is split once per product into fragment-ordered bf16 planes (split_x_kernel), so the
//     A fragments are whole 1-KiB coalesced loads;
//   * G is split on the fly in registers: a lane loads the 8 k-values of its column
//     (32 lanes = one 128-byte row segment per instruction) and packs hi / lo;
//   * one workgroup = 128 output columns x all rows x one K slice; its 8 waves are 4 column
//     tiles x 2 K halves, combined in order through LDS and a slab reduce (deterministic).
// NS = 3 splits every operand three ways, x = hi + mid + lo (24 mantissa bits, i.e. all of an
// fp32), and keeps the six product terms down to 2^-16 (hh, hm, mh, mm, hl, lh): the dropped
// terms are at the level of one fp32 rounding.  Six bf16 MFMAs still cost 3/8 of the fp32
// MFMA pass, so the Rayleigh-Ritz product of the tracker runs at fp32 accuracy from one
// HBM-bound read of G as well.
#include "dlco_internal.hpp"

namespace dlco {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int W8 = 8;             // waves per workgroup
constexpr int T8 = 64 * W8;
constexpr int KS = 4;             // default K split (gridDim.y); the slices are summed in order by splitk_reduce_f32

// plane element ((k16 * MT + tile) * 64 + lane) * 8 + j  =  X[tile*32 + (lane&31)][k16*16 + 8*(lane>>5) + j]
__global__ __launch_bounds__(256) void split_x_kernel(const float *X, long ldx, int M, int MT, int K, bf16x8 *hi, bf16x8 *lo,
                                                      bf16x8 *lo2)
{
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long total = (long)(K / 16) * MT * 64;
    if (t >= total) return;
    const int lane = (int)(t & 63);
    const long q = t >> 6;
    const int tile = (int)(q % MT);
    const int k16 = (int)(q / MT);
    const int row = tile * 32 + (lane & 31), k = k16 * 16 + 8 * (lane >> 5);
    bf16x8 h, l, l2;
    if (row < M) {
        const f32x4 a = *reinterpret_cast<const f32x4 *>(X + (long)row * ldx + k);
        const f32x4 b = *reinterpret_cast<const f32x4 *>(X + (long)row * ldx + k + 4);
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const float x = j < 4 ? a[j & 3] : b[j & 3];
            h[j] = (__bf16)x;
            const float r1 = x - (float)h[j];
            l[j] = (__bf16)r1;
            l2[j] = (__bf16)(r1 - (float)l[j]);
        }
    } else {
#pragma unroll
        for (int j = 0; j < 8; j++) { h[j] = (__bf16)0.f; l[j] = (__bf16)0.f; l2[j] = (__bf16)0.f; }
    }
    hi[t] = h;
    lo[t] = l;
    if (lo2) lo2[t] = l2;
}

struct Bf2Dev {
    int M, N, K;
    const bf16x8 *xhi, *xlo, *xlo2;
    const float *G;
    long ldg;
    float *slab;                  // [KS][M][N] raw partial sums
};

template <int MT, int NS>
__global__ __launch_bounds__(T8) void skinny_bf16x2_kernel(Bf2Dev g)
{
    // Workgroup = 128 output columns x one K quarter (blockIdx.y); wave w owns column tile w&3 and
    // K half w>>2 of that quarter.  All four column tiles walk the same X fragments, so the X
    // planes are pulled through L2 once per 128 columns instead of once per 32 (the planes do not
    // stay L2-resident beside the streamed G, and the Infinity Cache rate was the bound).
    __shared__ float red[4 * MT * 16 * 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lc = lane & 31, lh = lane >> 5;
    const int ct = wave & 3, kh = wave >> 2;
    const int col = blockIdx.x * 128 + ct * 32 + lc;
    const int nsteps = g.K / 16;
    const int per = nsteps / ((int)gridDim.y * 2);        // host guarantees divisibility (and per % 4 == 0)
    const int s0 = (blockIdx.y * 2 + kh) * per, s1 = s0 + per;

    f32x16 acc[MT];
#pragma unroll
    for (int a = 0; a < MT; a++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[a][r] = 0.f;

    bf16x8 ah[MT], al[MT], ahn[MT], aln[MT];
    bf16x8 am[NS == 3 ? MT : 1], amn[NS == 3 ? MT : 1];        // third plane of X (NS = 3 only)
    // (Reading G[col][k] instead — a lane streaming its own row, legal because G is symmetric —
    // was measured 15 % slower: 32 distinct lines per wave instruction.)
    auto load_g = [&](int s, float (&gv)[8]) {
        const float *p = g.G + ((long)s * 16 + 8 * lh) * g.ldg + col;
#pragma unroll
        for (int j = 0; j < 8; j++) gv[j] = p[(long)j * g.ldg];
    };
    auto load_a = [&](int s, bf16x8 (&h)[MT], bf16x8 (&l)[MT], bf16x8 (&m2)[NS == 3 ? MT : 1]) {
#pragma unroll
        for (int t = 0; t < MT; t++) {
            const long idx = ((long)s * MT + t) * 64 + lane;
            h[t] = g.xhi[idx];
            l[t] = g.xlo[idx];
            if (NS == 3) m2[t] = g.xlo2[idx];
        }
    };
    auto compute = [&](const float (&gv)[8], const bf16x8 (&h)[MT], const bf16x8 (&l)[MT], const bf16x8 (&m2)[NS == 3 ? MT : 1]) {
        bf16x8 bh, bl, bl2;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            bh[j] = (__bf16)gv[j];
            const float r1 = gv[j] - (float)bh[j];
            bl[j] = (__bf16)r1;
            if (NS == 3) bl2[j] = (__bf16)(r1 - (float)bl[j]);
        }
#pragma unroll
        for (int t = 0; t < MT; t++) {
            if (NS == 3) {                                   // smallest terms first
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(h[t], bl2, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(m2[t], bh, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(l[t], bl, acc[t], 0, 0, 0);
            }
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(h[t], bl, acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(l[t], bh, acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(h[t], bh, acc[t], 0, 0, 0);
        }
    };

    // G fragments run four K steps ahead (HBM latency: ~64 KiB in flight per CU), X fragments two
    // The loop is branch-free (the step count per wave is a multiple of 4, and the prefetch index
    // is clamped at the tail instead of guarded): any branch around a load makes hipcc drain
    // vmcnt(0) at the join, which serialises the whole prefetch.
    float gq[4][8];
    const int last = s1 - 1;
#pragma unroll
    for (int q = 0; q < 4; q++) load_g(s0 + q, gq[q]);
    load_a(s0, ah, al, am);
    load_a(s0 + 1, ahn, aln, amn);
    for (int s = s0; s < s1; s += 4) {
        compute(gq[0], ah, al, am);
        load_g(min(s + 4, last), gq[0]);
        load_a(min(s + 2, last), ah, al, am);
        compute(gq[1], ahn, aln, amn);
        load_g(min(s + 5, last), gq[1]);
        load_a(min(s + 3, last), ahn, aln, amn);
        compute(gq[2], ah, al, am);
        load_g(min(s + 6, last), gq[2]);
        load_a(min(s + 4, last), ah, al, am);
        compute(gq[3], ahn, aln, amn);
        load_g(min(s + 7, last), gq[3]);
        load_a(min(s + 5, last), ahn, aln, amn);
    }

    // ---- the two K halves of a column tile meet in LDS; the raw quarter-sum goes to its slab --------
    if (kh == 1) {
#pragma unroll
        for (int a = 0; a < MT; a++)
#pragma unroll
            for (int r = 0; r < 16; r++) red[((ct * MT + a) * 16 + r) * 64 + lane] = acc[a][r];
    }
    __syncthreads();
    if (kh == 0) {
        float *slab = g.slab + (long)blockIdx.y * g.M * g.N;
#pragma unroll
        for (int a = 0; a < MT; a++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const float s = acc[a][r] + red[((ct * MT + a) * 16 + r) * 64 + lane];
                const int i = a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (i < g.M) slab[(long)i * g.N + col] = s;
            }
    }
}

}  // namespace

size_t bf16x2_plane_bytes(int M, int K) { return (size_t)((M + 31) / 32) * 32 * K * sizeof(__bf16); }
size_t bf16x2_slab_floats(int M, int N, int ksplit) { return (size_t)(ksplit > 0 ? ksplit : KS) * M * N; }

// Returns false when the shape is not supported (caller falls back to the fp32 kernel).
bool skinny_product_bf16x2(const float *X, long ldx, int M, const float *G, long ldg, int N, int K, float alpha, float *C,
                           long ldc, const float *E1, float b1, const float *E2, float b2, void *plane_hi, void *plane_lo,
                           float *slab, hipStream_t s, int ksplit, void *plane_lo2)
{
    const int mt = (M + 31) / 32;
    const int ks = ksplit > 0 ? ksplit : KS;
    if (M < 1 || mt > 4 || N % 128 != 0 || K % (16 * ks * 2 * 4) != 0) return false;   // 4 steps per loop trip
    if (ldx % 4 != 0 || (reinterpret_cast<uintptr_t>(X) & 15) != 0) return false;
    const long total = (long)(K / 16) * mt * 64;
    hipLaunchKernelGGL(split_x_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, X, ldx, M, mt, K,
                       static_cast<bf16x8 *>(plane_hi), static_cast<bf16x8 *>(plane_lo), static_cast<bf16x8 *>(plane_lo2));
    Bf2Dev g;
    g.M = M; g.N = N; g.K = K;
    g.xhi = static_cast<const bf16x8 *>(plane_hi); g.xlo = static_cast<const bf16x8 *>(plane_lo);
    g.xlo2 = static_cast<const bf16x8 *>(plane_lo2);
    g.G = G; g.ldg = ldg; g.slab = slab;
    const dim3 grid(N / 128, ks), block(T8);
    if (plane_lo2) {                                         // three-way split: fp32-level accuracy
        if (mt == 1) hipLaunchKernelGGL((skinny_bf16x2_kernel<1, 3>), grid, block, 0, s, g);
        else if (mt == 2) hipLaunchKernelGGL((skinny_bf16x2_kernel<2, 3>), grid, block, 0, s, g);
        else if (mt == 3) hipLaunchKernelGGL((skinny_bf16x2_kernel<3, 3>), grid, block, 0, s, g);
        else hipLaunchKernelGGL((skinny_bf16x2_kernel<4, 3>), grid, block, 0, s, g);
    } else {
        if (mt == 1) hipLaunchKernelGGL((skinny_bf16x2_kernel<1, 2>), grid, block, 0, s, g);
        else if (mt == 2) hipLaunchKernelGGL((skinny_bf16x2_kernel<2, 2>), grid, block, 0, s, g);
        else if (mt == 3) hipLaunchKernelGGL((skinny_bf16x2_kernel<3, 2>), grid, block, 0, s, g);
        else hipLaunchKernelGGL((skinny_bf16x2_kernel<4, 2>), grid, block, 0, s, g);
    }
    DLCO_HIP(hipGetLastError());
    splitk_reduce_f32(slab, ks, M, N, C, ldc, alpha, 0.f, E1, b1, E2, b2, s);
    return true;
}

}  // namespace dlco
