// kernels_project.hip — P1 + P2 for MANY rows of the resident matrix (validation T1, ComputePJStats S2):
//
//     dist[n] = sum_q ( sum_k W[q][k] * x_n[k] )^2          x_n = row ids[n] (or row0 + n) of D [N,F]
//
// The reference forms the r x N projection with cuda::gemm, squares it with cuda::pow and sums its columns with
// cuda::reduce (src/pj-learn.cpp:504-511, src/misc.cpp:286-290: "the 2.5G" of its README at N = 500 000).  Here the
// r x N matrix never exists: a workgroup owns 128 rows of D, streams them ONCE from HBM in 64-deep chunks through LDS
// (16-byte loads of whole 256-byte row segments), keeps its 128 x r projections in MFMA accumulators over the whole
// K = F loop, and ends with the column sums of their squares - one float per row leaves the CU.
//
// fp32 MFMA would make this matrix-bound (2 r F flops per row: 3.4 ms for 500 000 x 8192 at r = 64 against 2.6 ms of
// HBM time), so the products run on the bf16 matrix cores with both operands split three ways, x = hi + mid + lo
// (all 24 mantissa bits; six MFMAs per term, products down to 2^-16 kept: the dropped part is at the level of one
// fp32 rounding - the tracker's Rayleigh-Ritz product uses the same scheme, kernels_bf16x2.hip), which is 3/8 of the
// fp32 MFMA time and leaves the pass HBM-bound.  NS = 1 is the BASELINE configs[4] variant ("bf16 MFMA + fp32
// accumulate"): both operands rounded to bf16 once, one MFMA per term (cfg.grad_bf16).
//
// Structure (the row-streaming product kernel's, kernels_bf16x2.hip): four LOADER waves keep two chunks of the 128 D
// rows in flight in registers (pair mode: one chunk, both descriptor rows, subtracted on the way into LDS - the same
// single fp32 rounding as the reference's Desc1 - Desc2) and copy the chunk's slice of the W planes (fragment order,
// split once per call by split_x_kernel) beside it; eight COMPUTE waves (4 tiles of 32 D rows x 2 halves of the
// chunk's K steps) read LDS only.  Epilogue: the two K halves meet in LDS, the projections are laid out [q][n], and
// one thread per D row adds the squares row after row in fp32 - the order of cv::reduce.  W taller than 96 rows goes
// through in passes of 96: a later pass continues the running sums of the one before, so the order over q is kept.
// SLAB = true (the training batch in the bf16 variant, a few hundred rows): K is split over gridDim.y and the raw
// partial projections go to a [slice][M][ldn] slab that sqdist_from_proj sums in slice order.
#include "dlco_internal.hpp"

#include <algorithm>

namespace dlco {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// kernels_bf16x2.hip: plane element ((k16 * MT + tile) * 64 + lane) * 8 + j = X[tile*32 + (lane&31)][k16*16 + 8*(lane>>5) + j]
void split_planes_bf16(const float *X, long ldx, int M, int K, void *hi, void *lo, void *lo2, hipStream_t s);

namespace {

constexpr int PK_KC = 64;                  // chunk depth (floats)
constexpr int PK_RS = PK_KC + 4;           // LDS row stride of the D image (floats): conflict-free 32-byte fragment reads
constexpr int PK_T = 768;                  // 8 compute + 4 loader waves
constexpr int PK_ROWS = 128;               // D rows per workgroup
constexpr int PK_GBYTES = PK_ROWS * PK_RS * 4;
__host__ __device__ constexpr int pk_abytes(int mt, int ns) { return ns * mt * 4 * 64 * 16; }
// two images of each operand; the epilogue reuses the space for the K-half exchange (16 KB per row tile) and the [q][n] image (16 KB per tile)
inline size_t pk_lds_bytes(int mt, int ns) { return std::max((size_t)2 * (PK_GBYTES + pk_abytes(mt, ns)), (size_t)32768 * mt); }

struct ProjDev {
    int M, K, nrows;              // rows of W in this pass (<= 32 MT), F, rows of D to project
    const bf16x8 *xhi, *xlo, *xlo2;
    const float *D;
    long ldd;
    const int32_t *ids, *ids2;    // row list (nullptr: rows row0 + n); pair mode: row n = D[ids[n]] - D[ids2[n]]
    long row0;
    float *out;                   // fused: dist [nrows]; slab: [gridDim.y][M][ldn]
    long ldn;
    int accumulate;               // fused: continue the running sums already in out (a later pass over more rows of W)
};

template <int MT, int NS, bool PAIR, bool SLAB>
__global__ __launch_bounds__(PK_T) void project_rows_kernel(ProjDev g)
{
    extern __shared__ __attribute__((aligned(16))) char lds_raw[];
    constexpr int ABYTES = pk_abytes(MT, NS);
    constexpr int APLANE = MT * 256;                          // 16-byte entries of one plane's chunk slice
    constexpr int AENT = NS * APLANE;
    constexpr int NA = (AENT + 255) / 256;                     // A entries per loader thread and chunk
    constexpr int NSTAGE = PAIR ? 1 : 2;                       // chunks a loader keeps in flight in registers
    float *gb0 = reinterpret_cast<float *>(lds_raw), *gb1 = reinterpret_cast<float *>(lds_raw + PK_GBYTES);
    bf16x8 *ab0 = reinterpret_cast<bf16x8 *>(lds_raw + 2 * PK_GBYTES), *ab1 = reinterpret_cast<bf16x8 *>(lds_raw + 2 * PK_GBYTES + ABYTES);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j0 = blockIdx.x * PK_ROWS;
    const int kslice = g.K / (int)gridDim.y;                  // host: a multiple of 64
    const int kbeg = blockIdx.y * kslice;
    const int nchunks = kslice / PK_KC;
    const int st0 = kbeg / 16;

    if (wave >= 8) {
        // ---- loader ---------------------------------------------------------------------------------------------
        // thread -> 16-byte column gc4 of rows grow + 16 u (u < 8) of the chunk.  D can be far larger than a buffer
        // resource's 4 GB window (16.4 GB at 500 000 x 8192) and the rows may be a gathered list: per-row 64-bit
        // pointers, computed once (a workgroup's rows do not change over the K loop).
        const int lt = tid - 512, grow = lt >> 4, gc4 = lt & 15;
        const float *rp[8];
        const float *rq[PAIR ? 8 : 1];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            int j = j0 + grow + 16 * u;
            if (j >= g.nrows) j = g.nrows - 1;                 // rows past the end repeat the last one; never stored
            const long row = g.ids ? (long)g.ids[j] : g.row0 + j;
            rp[u] = g.D + row * g.ldd + kbeg + 4 * gc4;
            if (PAIR) rq[u] = g.D + (long)g.ids2[j] * g.ldd + kbeg + 4 * gc4;
        }
        const int plane_bytes = MT * 32 * g.K * 2;
        const __amdgpu_buffer_rsrc_t rs_hi = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16x8 *>(g.xhi), 0, plane_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_lo = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16x8 *>(NS >= 2 ? g.xlo : g.xhi), 0, plane_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_lo2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16x8 *>(NS == 3 ? g.xlo2 : g.xhi), 0, plane_bytes, 0x00020000);
        // D rows come from HBM and are requested three iterations ahead (two register sets); the W planes come from L2
        // and are requested two ahead (one set) - and BEFORE the D request of the same iteration, so that waiting for
        // them (a wave's loads retire in order) never waits for the youngest D chunk.
        struct GStage { f32x4 gq[8]; f32x4 gr[PAIR ? 8 : 1]; };
        struct AStage { bf16x8 aq[NA]; };
        GStage sa, sb;
        AStage aa;
        auto gload = [&](int c, GStage &st) {
#pragma unroll
            for (int u = 0; u < 8; u++) {
                st.gq[u] = *reinterpret_cast<const f32x4 *>(rp[u] + c * PK_KC);
                if (PAIR) st.gr[u] = *reinterpret_cast<const f32x4 *>(rq[u] + c * PK_KC);
            }
        };
        auto aload = [&](int c, AStage &st) {
            const int a0 = (st0 + 4 * c) * MT * 64 * 16;       // byte offset of the chunk's slice inside a plane
#pragma unroll
            for (int v = 0; v < NA; v++) {
                const __amdgpu_buffer_rsrc_t &ra = (v / MT == 0) ? rs_hi : ((v / MT == 1) ? rs_lo : rs_lo2);
                st.aq[v] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(ra, 16 * lt, a0 + (v % MT) * 256 * 16, 0));
            }
        };
        auto gstore = [&](int buf, const GStage &st, const AStage &at) {
            float *gb = buf ? gb1 : gb0;
            bf16x8 *ab = buf ? ab1 : ab0;
#pragma unroll
            for (int u = 0; u < 8; u++) {
                f32x4 v = st.gq[u];
                if (PAIR) v -= st.gr[u];                       // Dist = Desc1 - Desc2, src/comp-uprjdists.cpp:327
                *reinterpret_cast<f32x4 *>(gb + (grow + 16 * u) * PK_RS + 4 * gc4) = v;
            }
#pragma unroll
            for (int v = 0; v < NA; v++) ab[lt + 256 * v] = at.aq[v];
        };
        const int last = nchunks - 1;
        if (NSTAGE == 2) {
            // chunk p: multiplied in iteration p, copied to LDS image p & 1 at the start of iteration p - 1; its D rows are
            // requested at iteration p - 3 into register set p & 1, its W slice at iteration p - 2
            gload(0, sa);
            aload(0, aa);
            gload(min(1, last), sb);
            gstore(0, sa, aa);
            aload(min(1, last), aa);
            gload(min(2, last), sa);
            __syncthreads();
            auto full = [&](int buf, GStage &set, int cc) { gstore(buf, set, aa); aload(cc + 2, aa); gload(cc + 3, set); __syncthreads(); };
            auto part = [&](int buf, GStage &set, int cc) {
                if (cc >= nchunks) return;
                if (cc + 1 < nchunks) gstore(buf, set, aa);
                if (cc + 2 < nchunks) aload(cc + 2, aa);
                if (cc + 3 < nchunks) gload(cc + 3, set);
                __syncthreads();
            };
            int c = 0;
            for (; c + 4 < nchunks; c += 2) { full(1, sb, c); full(0, sa, c + 1); }      // steady state: no branch around a load
            part(1, sb, c); part(0, sa, c + 1); part(1, sb, c + 2); part(0, sa, c + 3);
        } else {
            // one register set: chunk p is copied to image p & 1 in iteration p - 1 and chunk p + 1 requested right after
            gload(0, sa);
            aload(0, aa);
            gstore(0, sa, aa);
            aload(min(1, last), aa);
            gload(min(1, last), sa);
            __syncthreads();
            for (int c = 0; c < nchunks; c++) {
                if (c + 1 < nchunks) gstore((c + 1) & 1, sa, aa);
                if (c + 2 < nchunks) { aload(c + 2, aa); gload(c + 2, sa); }
                __syncthreads();
            }
        }
        __syncthreads();                                       // the compute waves' K-half exchange
        if (!SLAB) __syncthreads();                            // ... and their [q][n] image of the projections
        return;
    }

    // ---- compute: wave = (32-row tile of D, half of the chunk's four K steps), operands from LDS only ---------------
    const int lc = lane & 31, lh = lane >> 5;
    const int nt = wave & 3, kh = wave >> 2;
    f32x16 acc[MT];
#pragma unroll
    for (int a = 0; a < MT; a++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[a][r] = 0.f;

    struct Frag { f32x4 g0, g1; bf16x8 a[NS][MT]; };
    auto fetch = [&](const float *gb, const bf16x8 *ab, int q, Frag &f) {
        const float *p = gb + (nt * 32 + lc) * PK_RS + 16 * q + 8 * lh;
        f.g0 = *reinterpret_cast<const f32x4 *>(p);
        f.g1 = *reinterpret_cast<const f32x4 *>(p + 4);
#pragma unroll
        for (int n = 0; n < NS; n++)
#pragma unroll
            for (int t = 0; t < MT; t++) f.a[n][t] = ab[n * APLANE + (q * MT + t) * 64 + lane];
    };
    auto multiply = [&](const Frag &f) {
        bf16x8 bh, bl, bl2;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const float v = j < 4 ? f.g0[j & 3] : f.g1[j & 3];
            bh[j] = (__bf16)v;
            if (NS >= 2) {
                const float r1 = v - (float)bh[j];
                bl[j] = (__bf16)r1;
                if (NS == 3) bl2[j] = (__bf16)(r1 - (float)bl[j]);
            }
        }
#pragma unroll
        for (int t = 0; t < MT; t++) {
            if (NS == 3) {                                   // smallest terms first
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[0][t], bl2, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[NS - 1][t], bh, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[NS >= 2 ? 1 : 0][t], bl, acc[t], 0, 0, 0);
            }
            if (NS >= 2) {
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[0][t], bl, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[NS >= 2 ? 1 : 0][t], bh, acc[t], 0, 0, 0);
            }
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[0][t], bh, acc[t], 0, 0, 0);
        }
    };
    Frag fa;
    __syncthreads();
    for (int c = 0; c < nchunks; c++) {
        const float *gb = (c & 1) ? gb1 : gb0;
        const bf16x8 *ab = (c & 1) ? ab1 : ab0;
        fetch(gb, ab, 2 * kh, fa);
        multiply(fa);
        fetch(gb, ab, 2 * kh + 1, fa);
        multiply(fa);
        __syncthreads();
    }

    // ---- the two K halves of a tile meet in LDS -------------------------------------------------------------------------
    float *red = reinterpret_cast<float *>(lds_raw);          // 4*MT*16*64 floats (<= 48 KB)
    float *P = red + 4 * MT * 16 * 64;                        // [32 MT][128] projections (<= 48 KB): both inside the images
    if (kh == 1) {
#pragma unroll
        for (int a = 0; a < MT; a++)
#pragma unroll
            for (int r = 0; r < 16; r++) red[((nt * MT + a) * 16 + r) * 64 + lane] = acc[a][r];
    }
    __syncthreads();
    if (kh == 0) {
#pragma unroll
        for (int a = 0; a < MT; a++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const float sum = acc[a][r] + red[((nt * MT + a) * 16 + r) * 64 + lane];
                const int q = a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;     // row of W; column nt*32 + lc = row of D
                if (SLAB) {
                    const int n = j0 + nt * 32 + lc;
                    if (q < g.M && n < g.nrows) g.out[((long)blockIdx.y * g.M + q) * g.ldn + n] = sum;
                } else {
                    P[q * PK_ROWS + nt * 32 + lc] = sum;
                }
            }
    }
    if (SLAB) return;
    __syncthreads();
    // P2: squares added row after row in fp32, like cv::reduce(dim 0, SUM) after cv::pow (src/pj-learn.cpp:346-347)
    if (tid < PK_ROWS && j0 + tid < g.nrows) {
        float d = g.accumulate ? g.out[j0 + tid] : 0.f;
        for (int q = 0; q < g.M; q++) {
            const float p = P[q * PK_ROWS + tid];
            d += p * p;
        }
        g.out[j0 + tid] = d;
    }
}

template <int MT, int NS, bool PAIR, bool SLAB>
void launch_project(const ProjDev &g, dim3 grid, hipStream_t s)
{
    const size_t lds = pk_lds_bytes(MT, NS);
    ensure_dynamic_lds(reinterpret_cast<const void *>(project_rows_kernel<MT, NS, PAIR, SLAB>), (int)lds);
    hipLaunchKernelGGL((project_rows_kernel<MT, NS, PAIR, SLAB>), grid, dim3(PK_T), lds, s, g);
}

template <int NS, bool PAIR, bool SLAB>
void launch_project_mt(int mt, const ProjDev &g, dim3 grid, hipStream_t s)
{
    if (mt == 1) launch_project<1, NS, PAIR, SLAB>(g, grid, s);
    else if (mt == 2) launch_project<2, NS, PAIR, SLAB>(g, grid, s);
    else launch_project<3, NS, PAIR, SLAB>(g, grid, s);
}

}  // namespace

size_t project_rows_plane_bytes(int K) { return (size_t)96 * K * sizeof(__bf16); }

// dist[n] = |W x_n|^2 for n < nrows.  planes: three workspaces of project_rows_plane_bytes(F) bytes.  bf16 = the
// configs[4] variant (operands rounded to bf16 once).  Returns false when the shape is not supported (F % 64).
bool project_rows_sqdist(const float *W, long ldw, int r, const float *D, long ldd, int F, const int32_t *ids, const int32_t *ids2,
                         long row0, int nrows, float *dist, void *plane_hi, void *plane_lo, void *plane_lo2, bool bf16, hipStream_t s)
{
    if (r < 1 || nrows < 1 || F % 64 != 0 || ldd % 4 != 0 || ldw % 4 != 0 || (reinterpret_cast<uintptr_t>(D) & 15) != 0 ||
        (reinterpret_cast<uintptr_t>(W) & 15) != 0 || (long)96 * F * 2 >= (1L << 31))
        return false;
    const dim3 grid((nrows + PK_ROWS - 1) / PK_ROWS, 1);
    for (int q0 = 0; q0 < r; q0 += 96) {
        const int m = std::min(96, r - q0), mt = (m + 31) / 32;
        split_planes_bf16(W + (long)q0 * ldw, ldw, m, F, plane_hi, plane_lo, bf16 ? nullptr : plane_lo2, s);
        ProjDev g;
        g.M = m; g.K = F; g.nrows = nrows;
        g.xhi = static_cast<const bf16x8 *>(plane_hi); g.xlo = static_cast<const bf16x8 *>(plane_lo); g.xlo2 = static_cast<const bf16x8 *>(plane_lo2);
        g.D = D; g.ldd = ldd; g.ids = ids; g.ids2 = ids2; g.row0 = row0;
        g.out = dist; g.ldn = 0; g.accumulate = q0 > 0 ? 1 : 0;
        if (bf16) {
            if (ids2) launch_project_mt<1, true, false>(mt, g, grid, s); else launch_project_mt<1, false, false>(mt, g, grid, s);
        } else {
            if (ids2) launch_project_mt<3, true, false>(mt, g, grid, s); else launch_project_mt<3, false, false>(mt, g, grid, s);
        }
        DLCO_HIP(hipGetLastError());
    }
    return true;
}

// Raw partial projections of a FEW rows (the training batch) with K split over `ksplit` slices:
// slab[z][q][n] = sum over slice z of W[q][k] x_n[k], q < r <= 96, n < nrows; sqdist_from_proj(slab, ksplit, r, nrows, ldn, ...)
// finishes P2.  Used by the bf16 variant of the step (the fp32 step keeps the exact fp32 MFMA kernel).
bool project_rows_slab(const float *W, long ldw, int r, const float *D, long ldd, int F, const int32_t *ids, const int32_t *ids2,
                       int nrows, int ksplit, float *slab, long ldn, void *plane_hi, void *plane_lo, void *plane_lo2, bool bf16,
                       hipStream_t s)
{
    if (r < 1 || r > 96 || nrows < 1 || ksplit < 1 || F % (64 * ksplit) != 0 || ldd % 4 != 0 || ldw % 4 != 0 || !ids ||
        (reinterpret_cast<uintptr_t>(D) & 15) != 0 || (reinterpret_cast<uintptr_t>(W) & 15) != 0 || (long)96 * F * 2 >= (1L << 31))
        return false;
    const int mt = (r + 31) / 32;
    split_planes_bf16(W, ldw, r, F, plane_hi, plane_lo, bf16 ? nullptr : plane_lo2, s);
    ProjDev g;
    g.M = r; g.K = F; g.nrows = nrows;
    g.xhi = static_cast<const bf16x8 *>(plane_hi); g.xlo = static_cast<const bf16x8 *>(plane_lo); g.xlo2 = static_cast<const bf16x8 *>(plane_lo2);
    g.D = D; g.ldd = ldd; g.ids = ids; g.ids2 = ids2; g.row0 = 0;
    g.out = slab; g.ldn = ldn; g.accumulate = 0;
    const dim3 grid((nrows + PK_ROWS - 1) / PK_ROWS, ksplit);
    if (bf16) {
        if (ids2) launch_project_mt<1, true, true>(mt, g, grid, s); else launch_project_mt<1, false, true>(mt, g, grid, s);
    } else {
        if (ids2) launch_project_mt<3, true, true>(mt, g, grid, s); else launch_project_mt<3, false, true>(mt, g, grid, s);
    }
    DLCO_HIP(hipGetLastError());
    return true;
}

}  // namespace dlco
