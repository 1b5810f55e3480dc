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

#include <type_traits>

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

// ---- symmetric G, whole matrix: stream ROWS of G ---------------------------------------------------
// out[:, j] = sum_k X[:, k] G[k][j] = sum_k X[:, k] G[j][k] when G is symmetric: the B operand of output
// column j is ROW j of G, which is contiguous in memory.  A workgroup owns 128 output columns = 128 rows
// of G and one K slice, and pulls that 128 x kslice panel through LDS in chunks of 128 x 128 floats
// (64 KiB), so every global load instruction covers two whole 512-byte row segments (the kernel above
// fetches 128-byte segments of 16 different rows with dword loads).
//   * Roles are split by wave, because a wave has ONE in-order counter for its global loads: the four
//     LOADER waves keep two chunks (128 KiB per CU) of G in flight in registers and copy the chunk
//     that has arrived into LDS; the eight COMPUTE waves (4 column tiles x 2 K halves of the chunk)
//     only ever wait for their own X fragments.  One barrier per chunk, two LDS images.
//   * LDS row stride 132 floats: the 32-byte fragment reads of 32 different rows are conflict-free.
//   * G is read once: non-temporal loads, so the stream does not evict the X planes from L2.
//   * Workgroups start at different chunks of their K slice and wrap around: in step, they would all
//     request the same offset inside a 32-KiB row and pile onto a few HBM channels.
// Same MFMA schedule per K step and the same slab reduction as above; the K order differs (rotated
// chunks, halves of a chunk instead of halves of the slice), so the two kernels agree to fp32
// rounding, not bit for bit.  Deterministic from run to run.
constexpr int RK_KC = 128;                 // chunk depth (floats)
constexpr int RK_RS = RK_KC + 4;           // LDS row stride (floats)
constexpr int RK_T = 768;                  // 8 compute + 4 loader waves
constexpr size_t RK_LDS_BYTES = (size_t)2 * 128 * RK_RS * sizeof(float);

template <int MT, int NS>
__global__ __launch_bounds__(RK_T) void skinny_rows_kernel(Bf2Dev g)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *buf0 = lds, *buf1 = lds + 128 * RK_RS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool loader = wave >= 8;
    const int j0 = blockIdx.x * 128;
    const int kslice = g.K / (int)gridDim.y;                  // host: a multiple of RK_KC
    const int kbeg = blockIdx.y * kslice;
    const int nchunks = kslice / RK_KC;
    const int rot = (int)(blockIdx.x % (unsigned)nchunks);
    auto pc = [&](int c) { const int v = c + rot; return v >= nchunks ? v - nchunks : v; };   // position -> chunk of the slice

    if (loader) {
        // thread -> 16-byte column lc4 of rows lrow + 8u (u < 16) of the chunk
        const int lt = tid - 512, lrow = lt >> 5, lc4 = lt & 31;
        const float *gsrc = g.G + (long)(j0 + lrow) * g.ldg + kbeg + 4 * lc4;
        f32x4 sa[16], sb[16];
        auto gload = [&](int c, f32x4 (&st)[16]) {
            const float *p = gsrc + (long)pc(c) * RK_KC;
#pragma unroll
            for (int u = 0; u < 16; u++) st[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(p + (long)(8 * u) * g.ldg));
        };
        auto gstore = [&](float *buf, const f32x4 (&st)[16]) {
#pragma unroll
            for (int u = 0; u < 16; u++) *reinterpret_cast<f32x4 *>(buf + (lrow + 8 * u) * RK_RS + 4 * lc4) = st[u];
        };
        // position c is multiplied during iteration c; it is stored at the start of iteration c - 1
        // (position 0: before the loop) and was requested two iterations before that
        gload(0, sa);
        gload(nchunks > 1 ? 1 : 0, sb);
        gstore(buf0, sa);
        gload(nchunks > 2 ? 2 : 0, sa);
        __syncthreads();
        int c = 0;
        // steady state, free of branches around the loads (after a conditional load the compiler can no
        // longer count which requests are older than the ones it waits for, and drains them all)
        for (; c + 4 < nchunks; c += 2) {
            gstore(buf1, sb);                                  // iteration c: position c+1 -> buf1, request c+3
            gload(c + 3, sb);
            __syncthreads();
            gstore(buf0, sa);                                  // iteration c+1: position c+2 -> buf0, request c+4
            gload(c + 4, sa);
            __syncthreads();
        }
        for (; c < nchunks; c += 2) {
            // iteration c (even): store position c+1 (in sb) into buf1, request position c+3 into sb
            if (c + 1 < nchunks) gstore(buf1, sb);
            if (c + 3 < nchunks) gload(c + 3, sb);
            __syncthreads();
            if (c + 1 >= nchunks) break;
            // iteration c+1 (odd): store position c+2 (in sa) into buf0, request position c+4 into sa
            if (c + 2 < nchunks) gstore(buf0, sa);
            if (c + 4 < nchunks) gload(c + 4, sa);
            __syncthreads();
        }
        __syncthreads();                                       // the compute waves' K-half exchange
        return;
    }

    const int lc = lane & 31, lh = lane >> 5;
    const int nt = wave & 3, kh = wave >> 2;
    f32x16 acc[MT];
#pragma unroll
    for (int a = 0; a < MT; a++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[a][r] = 0.f;

    auto load_a = [&](int s, bf16x8 (&h)[MT], bf16x8 (&l)[MT], bf16x8 (&m2)[NS == 3 ? MT : 1]) {
#pragma unroll
        for (int t = 0; t < MT; t++) {
            const long idx = ((long)s * MT + t) * 64 + lane;
            h[t] = g.xhi[idx];
            l[t] = g.xlo[idx];
            if (NS == 3) m2[t] = g.xlo2[idx];
        }
    };
    auto compute = [&](const float *buf, int q, const bf16x8 (&h)[MT], const bf16x8 (&l)[MT], const bf16x8 (&m2)[NS == 3 ? MT : 1]) {
        const float *p = buf + (nt * 32 + lc) * RK_RS + 16 * (4 * kh + q) + 8 * lh;
        const f32x4 g0 = *reinterpret_cast<const f32x4 *>(p), g1 = *reinterpret_cast<const f32x4 *>(p + 4);
        bf16x8 bh, bl, bl2;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const float v = j < 4 ? g0[j & 3] : g1[j & 3];
            bh[j] = (__bf16)v;
            const float r1 = v - (float)bh[j];
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

    bf16x8 ah[MT], al[MT], ahn[MT], aln[MT];
    bf16x8 am[NS == 3 ? MT : 1], amn[NS == 3 ? MT : 1];
    // K step (of 16) of this wave inside the chunk at position c: kbeg/16 + 8*pc(c) + 4kh + q, q = 0..3
    const int st0 = kbeg / 16 + 4 * kh;
    load_a(st0 + 8 * pc(0), ah, al, am);
    load_a(st0 + 8 * pc(0) + 1, ahn, aln, amn);
    __syncthreads();
    for (int c = 0; c < nchunks; c++) {
        const float *cur = (c & 1) ? buf1 : buf0;
        const int st = st0 + 8 * pc(c);
        const int stn = st0 + 8 * pc(c + 1 < nchunks ? c + 1 : c);   // first two steps of the next chunk (last: reloaded, unused)
        compute(cur, 0, ah, al, am);
        load_a(st + 2, ah, al, am);
        compute(cur, 1, ahn, aln, amn);
        load_a(st + 3, ahn, aln, amn);
        compute(cur, 2, ah, al, am);
        load_a(stn, ah, al, am);
        compute(cur, 3, ahn, aln, amn);
        load_a(stn + 1, ahn, aln, amn);
        __syncthreads();
    }

    // ---- the two K halves of a column tile meet in LDS; the raw slice sum goes to its slab ------------
    float *red = lds;                                         // 4*MT*16*64 floats <= one chunk buffer
    if (kh == 1) {
#pragma unroll
        for (int a = 0; a < MT; a++)
#pragma unroll
            for (int r = 0; r < 16; r++) red[((nt * MT + a) * 16 + r) * 64 + lane] = acc[a][r];
    }
    __syncthreads();
    if (kh == 0) {
        float *slab = g.slab + (long)blockIdx.y * g.M * g.N;
        const int col = j0 + nt * 32 + lc;
#pragma unroll
        for (int a = 0; a < MT; a++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const float s = acc[a][r] + red[((nt * MT + a) * 16 + r) * 64 + lane];
                const int i = a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (i < g.M) slab[(long)i * g.N + col] = s;
            }
    }
}

template <int MT, int NS>
void launch_rows(const Bf2Dev &g, dim3 grid, hipStream_t s)
{
    static bool attr = false;
    if (!attr) {
        DLCO_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(skinny_rows_kernel<MT, NS>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)RK_LDS_BYTES));
        attr = true;
    }
    hipLaunchKernelGGL((skinny_rows_kernel<MT, NS>), grid, dim3(RK_T), RK_LDS_BYTES, s, g);
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
    // whole symmetric matrix: stream its rows through LDS (skinny_rows_kernel)
    static const bool rows_ok = std::getenv("DLCO_PRODUCT_V1") == nullptr;
    if (rows_ok && N == K && K % (RK_KC * ks) == 0 && ldg % 4 == 0 && (reinterpret_cast<uintptr_t>(G) & 15) == 0) {
        if (plane_lo2) {
            if (mt == 1) launch_rows<1, 3>(g, grid, s);
            else if (mt == 2) launch_rows<2, 3>(g, grid, s);
            else if (mt == 3) launch_rows<3, 3>(g, grid, s);
            else launch_rows<4, 3>(g, grid, s);
        } else {
            if (mt == 1) launch_rows<1, 2>(g, grid, s);
            else if (mt == 2) launch_rows<2, 2>(g, grid, s);
            else if (mt == 3) launch_rows<3, 2>(g, grid, s);
            else launch_rows<4, 2>(g, grid, s);
        }
        DLCO_HIP(hipGetLastError());
        splitk_reduce_f32(slab, ks, M, N, C, ldc, alpha, 0.f, E1, b1, E2, b2, s);
        return true;
    }
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
