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

#include <algorithm>
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
    int tiled;                    // G is stored as contiguous 128 x 128 tiles, tile (I, J) at ((I * K/128) + J) * 16384 floats
    int xcd_map;                  // rows kernel: give the workgroups of an XCD one K slice (see the kernel)
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
// of G and one K slice and walks it in chunks of 64 k.
//
// What the kernel above waits for is not G but its X fragments: every wave fetches them itself, two steps
// ahead, from an L2 that the G stream keeps flushing (measured: the same pass with deeper or wider G
// prefetch got slower, 75 -> 87 -> 131 us, whatever the shape of the G accesses; non-temporal G loads
// keep the planes in L2 but are themselves 30 % slower).  So here BOTH operands of a chunk go through
// LDS, and the roles are split by wave (a wave has one in-order counter for its loads):
//   * four LOADER waves keep two chunks in flight in their registers - the 128 x 64 panel of G
//     (32 KiB) and the chunk's slice of the X planes (contiguous: the planes are stored in fragment
//     order) - and copy the chunk that has arrived into one of two LDS images;
//   * eight COMPUTE waves (4 column tiles x 2 halves of the chunk's K steps; two per SIMD, so that one
//     converts / reads LDS while the other's MFMAs run) multiply the image before it: B fragments are
//     32-byte reads of 32 different rows (row stride 68 floats: conflict-free), A fragments whole
//     1-KiB reads; they issue no global load at all.
// One barrier per chunk.  Same MFMA schedule per K step and the same slab reduction as above; a column's
// K order is plain ascending inside the slice.  Deterministic.  `tiled`: G stored as contiguous
// 128 x 128 tiles (a chunk is then half a tile).
constexpr int RK_KC = 64;                  // chunk depth (floats)
constexpr int RK_RS = RK_KC + 4;           // LDS row stride of the G image (floats)
constexpr int RK_T = 768;                  // 8 compute + 4 loader waves
constexpr int RK_GBYTES = 128 * RK_RS * 4; // one G image
__host__ __device__ constexpr int rk_abytes(int mt, int ns) { return ns * mt * 4 * 64 * 16; }     // one A image: ns planes x 4 steps x mt tiles x 1 KiB
inline size_t rk_lds_bytes(int mt, int ns) { return (size_t)2 * (RK_GBYTES + rk_abytes(mt, ns)); }

template <int MT, int NS>
__global__ __launch_bounds__(RK_T) void skinny_rows_kernel(Bf2Dev g)
{
    extern __shared__ __attribute__((aligned(16))) char lds_raw[];
    constexpr int ABYTES = rk_abytes(MT, NS);
    constexpr int APLANE = MT * 256;                          // 16-byte entries of one plane's chunk slice
    constexpr int AENT = NS * APLANE;                          // 16-byte entries of one A image
    constexpr int NA = (AENT + 255) / 256;                     // A entries per loader thread and chunk
    float *gb0 = reinterpret_cast<float *>(lds_raw), *gb1 = reinterpret_cast<float *>(lds_raw + RK_GBYTES);
    bf16x8 *ab0 = reinterpret_cast<bf16x8 *>(lds_raw + 2 * RK_GBYTES), *ab1 = reinterpret_cast<bf16x8 *>(lds_raw + 2 * RK_GBYTES + ABYTES);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // Workgroup -> (column tile bx, K slice by).  Workgroups are dealt round-robin to the 8 XCDs, each with
    // its own L2: all workgroups of one XCD are given the SAME K slice (or as few as possible), so that the
    // slice of the X planes they all re-read (planes / ks: ~1 MB) stays in that L2 beside the G stream; with
    // the plain mapping every XCD touches all of the planes (3 - 4.5 MB > its 4 MB L2) and the re-reads,
    // as many bytes as G itself, come from the Infinity Cache.
    int bx = blockIdx.x, by = blockIdx.y;
    {
        const int gx = gridDim.x, ks = gridDim.y;
        if (g.xcd_map && ks <= 8 && 8 % ks == 0 && (gx * ks) % 8 == 0 && gx % (8 / ks) == 0) {
            const int b = by * gx + bx, xcd = b & 7, w = b >> 3, per = 8 / ks;
            by = xcd / per;
            bx = w * per + xcd % per;
        }
    }
    const int j0 = bx * 128;
    const int kslice = g.K / (int)gridDim.y;                  // host: a multiple of 128
    const int kbeg = by * kslice;
    const int nchunks = kslice / RK_KC;
    const int st0 = kbeg / 16;                                // K step (of 16) of chunk c, step q: st0 + 4c + q

    if (wave >= 8) {
        // ---- loader -----------------------------------------------------------------------------------------
        // G: thread -> 16-byte column c4 of rows row + 16u (u < 8) of the chunk; buffer loads with one per-lane
        // byte offset, the row / chunk displacement is a scalar operand.
        const int lt = tid - 512, grow = lt >> 4, gc4 = lt & 15;
        const __amdgpu_buffer_rsrc_t rsrc =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(g.G), 0, (int)((long)g.K * g.ldg * (long)sizeof(float)), 0x00020000);
        const long pitch = g.tiled ? 128 : g.ldg;
        const long base = g.tiled ? ((long)bx * (g.K / 128) + kbeg / 128) * 16384 + (long)grow * 128 + 4 * gc4
                                  : (long)(j0 + grow) * g.ldg + kbeg + 4 * gc4;
        const int voff = (int)(base * (long)sizeof(float));
        const int row16 = (int)(16 * pitch * (long)sizeof(float));
        const int plane_bytes = MT * 32 * g.K * 2;
        const __amdgpu_buffer_rsrc_t rs_hi = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16x8 *>(g.xhi), 0, plane_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_lo = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16x8 *>(g.xlo), 0, plane_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_lo2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16x8 *>(NS == 3 ? g.xlo2 : g.xlo), 0, plane_bytes, 0x00020000);
        struct Stage { f32x4 gq[8]; bf16x8 aq[NA]; };
        Stage s0, s1;
        auto gload = [&](int c, Stage &st) {
            const int soff = g.tiled ? ((c >> 1) * 16384 + (c & 1) * 64) * (int)sizeof(float) : c * RK_KC * (int)sizeof(float);
#pragma unroll
            for (int u = 0; u < 8; u++) st.gq[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, soff + u * row16, 0));
            // the chunk's slice of every plane is MT * 256 consecutive 16-byte entries (fragment order): entry
            // lt + 256 v belongs to plane v / MT -> static plane per v, scalar offset, one per-lane offset (16 lt)
            const int a0 = (st0 + 4 * c) * MT * 64 * 16;       // byte offset of the slice inside a plane
#pragma unroll
            for (int v = 0; v < NA; v++) {
                const __amdgpu_buffer_rsrc_t &ra = (v / MT == 0) ? rs_hi : ((v / MT == 1) ? rs_lo : rs_lo2);
                st.aq[v] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(ra, 16 * lt, a0 + (v % MT) * 256 * 16, 0));
            }
        };
        auto gstore = [&](int buf, const Stage &st) {
            float *gb = buf ? gb1 : gb0;
            bf16x8 *ab = buf ? ab1 : ab0;
#pragma unroll
            for (int u = 0; u < 8; u++) *reinterpret_cast<f32x4 *>(gb + (grow + 16 * u) * RK_RS + 4 * gc4) = st.gq[u];
#pragma unroll
            for (int v = 0; v < NA; v++) ab[lt + 256 * v] = st.aq[v];
        };
        // Chunk p is multiplied during iteration p, copied to LDS image p & 1 at the start of iteration
        // p - 1 and requested at iteration p - 3 into register set p & 1 (chunks 0..2: before the loop).
        const int last = nchunks - 1;
        gload(0, s0);
        gload(min(1, last), s1);
        gstore(0, s0);
        gload(min(2, last), s0);
        __syncthreads();
        // iteration cc: copy chunk cc+1 (waiting in `set`) into image `buf`, request chunk cc+3 into the same registers
        auto full = [&](int buf, Stage &set, int cc) { gstore(buf, set); gload(cc + 3, set); __syncthreads(); };
        auto part = [&](int buf, Stage &set, int cc) {
            if (cc >= nchunks) return;
            if (cc + 1 < nchunks) gstore(buf, set);
            if (cc + 3 < nchunks) gload(cc + 3, set);
            __syncthreads();
        };
        int c = 0;
        // steady state: no branch around a load (after one the compiler can no longer tell which requests
        // are older than the ones it waits for, and drains them all)
        for (; c + 4 < nchunks; c += 2) { full(1, s1, c); full(0, s0, c + 1); }
        // tail (at most four iterations): the same sequence with guards
        part(1, s1, c); part(0, s0, c + 1); part(1, s1, c + 2); part(0, s0, c + 3);
        __syncthreads();                                       // the compute waves' K-half exchange
        return;
    }

    // ---- compute: wave = (32-column tile, half of the chunk's four K steps), operands from LDS only ----------
    const int lc = lane & 31, lh = lane >> 5;
    const int nt = wave & 3, kh = wave >> 2;
    f32x16 acc[MT];
#pragma unroll
    for (int a = 0; a < MT; a++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[a][r] = 0.f;

    // operands of one K step, read from the LDS images: issued one step ahead of the MFMAs that use them
    // (this wave is alone on its SIMD as far as LDS latency goes)
    struct Frag { f32x4 g0, g1; bf16x8 a[NS][MT]; };
    auto fetch = [&](const float *gb, const bf16x8 *ab, int q, Frag &f) {
        const float *p = gb + (nt * 32 + lc) * RK_RS + 16 * q + 8 * lh;
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
            const float r1 = v - (float)bh[j];
            bl[j] = (__bf16)r1;
            if (NS == 3) bl2[j] = (__bf16)(r1 - (float)bl[j]);
        }
#pragma unroll
        for (int t = 0; t < MT; t++) {
            if (NS == 3) {                                   // smallest terms first
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[0][t], bl2, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[NS - 1][t], bh, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[1][t], bl, acc[t], 0, 0, 0);
            }
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[0][t], bl, acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[1][t], bh, acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[0][t], bh, acc[t], 0, 0, 0);
        }
    };
    Frag fa;                                                  // (the SIMD's other compute wave covers the LDS latency)
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

    // ---- the two K halves of a column tile meet in LDS; the raw slice sum goes to its slab ------------
    float *red = reinterpret_cast<float *>(lds_raw);          // 4*MT*16*64 floats <= the two G images
    if (kh == 1) {
#pragma unroll
        for (int a = 0; a < MT; a++)
#pragma unroll
            for (int r = 0; r < 16; r++) red[((nt * MT + a) * 16 + r) * 64 + lane] = acc[a][r];
    }
    __syncthreads();
    if (kh == 0) {
        float *slab = g.slab + (long)by * g.M * g.N;
        const int col = j0 + nt * 32 + lc;
#pragma unroll
        for (int a = 0; a < MT; a++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const float sum = acc[a][r] + red[((nt * MT + a) * 16 + r) * 64 + lane];
                const int i = a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (i < g.M) slab[(long)i * g.N + col] = sum;
            }
    }
}

// The ordered sum of the K-slice slabs (splitk_reduce_f32's arithmetic: slices added in order, then alpha, E1, E2) that ALSO
// emits the two-way split planes of its result in fragment order: inside the Chebyshev recurrence the output of one filter
// product is the X of the next, so the next product needs no split_x_kernel launch of its own.  One thread per 8 consecutive
// columns of a row: 32-byte slab reads (a wave covers 2 KiB of a row), one 16-byte store per plane.
__global__ __launch_bounds__(256) void reduce_split_kernel(const float *slab, int split, int M, int MT, int N, float *C, long ldc, float alpha,
                                                           const float *E1, float b1, const float *E2, float b2, bf16x8 *hi, bf16x8 *lo)
{
    const long total = (long)M * N, q = N >> 3;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < (long)M * q; e += (long)gridDim.x * blockDim.x) {
        const int i = (int)(e / q), k8 = (int)(e % q), j = k8 * 8;
        const float *p = slab + (long)i * N + j;
        f32x4 s0 = *reinterpret_cast<const f32x4 *>(p), s1 = *reinterpret_cast<const f32x4 *>(p + 4);
        for (int z = 1; z < split; z++) {
            s0 += *reinterpret_cast<const f32x4 *>(p + (long)z * total);
            s1 += *reinterpret_cast<const f32x4 *>(p + (long)z * total + 4);
        }
        const long idx = (long)i * ldc + j;
        f32x4 o0 = alpha * s0, o1 = alpha * s1;
        if (E1) { o0 += b1 * *reinterpret_cast<const f32x4 *>(E1 + idx); o1 += b1 * *reinterpret_cast<const f32x4 *>(E1 + idx + 4); }
        if (E2) { o0 += b2 * *reinterpret_cast<const f32x4 *>(E2 + idx); o1 += b2 * *reinterpret_cast<const f32x4 *>(E2 + idx + 4); }
        *reinterpret_cast<f32x4 *>(C + idx) = o0;
        *reinterpret_cast<f32x4 *>(C + idx + 4) = o1;
        bf16x8 h, l;
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const float x = u < 4 ? o0[u & 3] : o1[u & 3];
            h[u] = (__bf16)x;
            l[u] = (__bf16)(x - (float)h[u]);
        }
        // plane entry ((k16 * MT + tile) * 64 + lane): row tile*32 + (lane & 31), columns k16*16 + 8*(lane >> 5) .. +7
        const long t = ((long)(k8 >> 1) * MT + (i >> 5)) * 64 + (i & 31) + 32 * (k8 & 1);
        hi[t] = h;
        lo[t] = l;
    }
}

// ---- symmetric G stored as its packed upper tiles: every tile leaves HBM once ------------------------------------------
// out[:, I] = sum_J X[:, J] G[J][I] over tile blocks; G[J][I] is the stored tile (I, J) read along its rows when I <= J
// ("direct", exactly the chunk of skinny_rows_kernel), or the stored tile (J, I) read along its COLUMNS when J < I
// ("transposed").  Every off-diagonal tile therefore has two users, output blocks I and J, and the Infinity Cache does
// not help (tools/mall_probe.cpp: a table that fits it streams at the HBM rate all the same) - the second read has to
// hit the XCD's own L2.  So the two users of a tile are put on the SAME XCD at the SAME time:
//   * the 64 tile rows are the vectors of Z_2^6; block I meets block I ^ d, d = 0..63.  The 64 differences are dealt to
//     four groups p by their two top bits h_p (01, 10, 11, 00); within group p the blocks are halved by a linear form
//     f_p that vanishes on the group's differences (bit 5, bit 4, bit 5 ^ bit 4, bit 5), so that I and I ^ d always
//     fall into the same half.  XCD x = 2 p + c takes the 32 blocks with f_p(I) = c, one workgroup each (one per CU),
//     and in step s = 0..15 workgroup I works on the tile it shares with I ^ (16 h_p + s): the partner workgroup is on
//     the same XCD and works on the same tile in the same step.  An XCD streams 16 tiles (1 MB) per step, each read
//     twice back to back, and touches the X planes of its own 32 blocks only (1.5 MB, re-read every step: L2-resident).
//   * every output block appears once per group: four partial slabs, summed in order by splitk_reduce_f32 - the same
//     reduction as before; a column's K order is fixed by the schedule.  Deterministic.
// Workgroups are dealt to XCDs round-robin by the hardware (block b -> XCD b % 8; observed, not guaranteed): a wrong guess
// costs the L2 hits (the kernel then moves the full 268 MB again), never correctness.
// Roles, LDS images, MFMA schedule and epilogue are those of skinny_rows_kernel; a transposed chunk is staged as
// [64 k][128 columns] (row stride 132) and its B fragments are eight 4-byte reads of 32 consecutive columns.
constexpr int SYM_TS = 132;                // LDS row stride of a transposed chunk image (floats)
static_assert(RK_KC * SYM_TS * 4 <= RK_GBYTES, "a transposed chunk image must fit the G image");

// Other widths (round 4): nt = K / 128 tile rows are embedded in Z_2^b, 2^b = the next power of two >= max(nt, 4); the same
// pairing runs on 4 * 2^b workgroups, of which those of a block I >= nt leave at once and the others skip partners
// J >= nt (a workgroup's list of real partners sits in LDS).  Below 64 tile rows there are fewer than 256 workgroups and
// the XCD placement is whatever the dispatcher does: the matrix is then at most 34 MB and the pass is short anyway.
struct SymDev {
    int M, K;
    int nt, lb;                   // tile rows K / 128, log2 of the padded tile count (2..6)
    const bf16x8 *xhi, *xlo, *xlo2;
    const float *G;               // packed upper tiles: tile (I, J), I <= J, at (I*nt - I*(I-1)/2 + J - I) * 16384 floats
    float *slab;                  // [4][M][K] raw partial sums
};

// FULL: the tile count is the power of two itself (K = 8192: 64 tile rows) - every partner is real and comes from bit
// arithmetic alone, no list in LDS inside the loop.
template <int MT, int NS, bool FULL>
__global__ __launch_bounds__(RK_T) void skinny_sym_kernel(SymDev g)
{
    extern __shared__ __attribute__((aligned(16))) char lds_raw[];
    constexpr int ABYTES = rk_abytes(MT, NS);
    constexpr int APLANE = MT * 256;
    constexpr int AENT = NS * APLANE;
    constexpr int NA = (AENT + 255) / 256;
    const int NT = g.nt, lb = g.lb;                            // tile rows; 2^lb = padded tile count
    float *gb0 = reinterpret_cast<float *>(lds_raw), *gb1 = reinterpret_cast<float *>(lds_raw + RK_GBYTES);
    bf16x8 *ab0 = reinterpret_cast<bf16x8 *>(lds_raw + 2 * RK_GBYTES), *ab1 = reinterpret_cast<bf16x8 *>(lds_raw + 2 * RK_GBYTES + ABYTES);
    __shared__ int jlist[16];                                  // the real partners of this workgroup, in schedule order
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // workgroup -> (group p, half c, block I): see above (bits 5, 4 there are bits lb-1, lb-2 here)
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int p = xcd >> 1, c = xcd & 1;
    int I;
    if (p == 0 || p == 3) I = (c << (lb - 1)) | slot;
    else {
        const int b5 = slot >> (lb - 2), b4 = p == 1 ? c : (b5 ^ c);
        I = (b5 << (lb - 1)) | (b4 << (lb - 2)) | (slot & ((1 << (lb - 2)) - 1));
    }
    if (I >= NT) return;                                       // a block of the padding: the whole workgroup leaves
    const int dbase = (p == 0 ? 1 : (p == 1 ? 2 : (p == 2 ? 3 : 0))) << (lb - 2);
    int nchunks;                                               // real partner tiles x 2 halves of 64 k
    if (FULL) nchunks = 2 << (lb - 2);
    else {
        const int nslots = 1 << (lb - 2);
        int cnt = 0;
        for (int sl = 0; sl < nslots; sl++) {
            const int J = I ^ (dbase | sl);
            if (J < NT) { if (tid == 0) jlist[cnt] = J; cnt++; }
        }
        nchunks = 2 * cnt;
        __syncthreads();
    }
    if (!FULL && nchunks == 0) {
        // no real partner in this group (only possible beside padding blocks): the group's slab still has to hold zeros
        // for this block's columns, the reduction adds all four
        float *slab = g.slab + (long)p * g.M * g.K;
        for (int e = tid; e < g.M * 128; e += RK_T) slab[(long)(e >> 7) * g.K + I * 128 + (e & 127)] = 0.f;
        return;
    }
    // chunk cc: partner block, orientation, byte offset of the stored tile
    auto partner = [&](int cc) { return FULL ? (I ^ (dbase | (cc >> 1))) : jlist[cc >> 1]; };
    auto tile_bytes = [&](int J) {
        const int a = min(I, J), b = max(I, J);
        return (a * NT - a * (a - 1) / 2 + (b - a)) * (128 * 128 * (int)sizeof(float));
    };

    if (wave >= 8) {
        // ---- loader -----------------------------------------------------------------------------------------
        const int lt = tid - 512;
        // direct chunk: thread -> 16-byte column gc4 of tile rows grow + 16 u; transposed: column c4 of tile rows half*64 + kr + 8 u
        const int voff_d = ((lt >> 4) * 128 + 4 * (lt & 15)) * (int)sizeof(float);
        const int voff_t = ((lt >> 5) * 128 + 4 * (lt & 31)) * (int)sizeof(float);
        const int loff_d = (lt >> 4) * RK_RS + 4 * (lt & 15), loff_t = (lt >> 5) * SYM_TS + 4 * (lt & 31);
        const __amdgpu_buffer_rsrc_t rsrc =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(g.G), 0, (NT * (NT + 1) / 2) * 128 * 128 * (int)sizeof(float), 0x00020000);
        const int plane_bytes = MT * 32 * g.K * 2;
        const __amdgpu_buffer_rsrc_t rs_hi = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16x8 *>(g.xhi), 0, plane_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_lo = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16x8 *>(g.xlo), 0, plane_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_lo2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16x8 *>(NS == 3 ? g.xlo2 : g.xlo), 0, plane_bytes, 0x00020000);
        struct Stage { f32x4 gq[8]; bf16x8 aq[NA]; };
        Stage s0, s1;
        // The loads of a chunk are issued in ONE fixed order (G then X planes, fenced against the scheduler): hipcc places
        // its s_waitcnt from a per-register model that is merged conservatively at the loop header, and when the prologue
        // issues a set's loads in another order than the loop body does, the merged state makes a store in the loop wait
        // for the YOUNGEST load in flight (vmcnt(0): the chunk requested one barrier ago) instead of its own (vmcnt(14+)) -
        // the prefetch distance collapses to one chunk (seen in the ISA of round 3's kernel).
        auto gload = [&](int cc, Stage &st) {
            const int J = __builtin_amdgcn_readfirstlane(partner(cc)), half = cc & 1;
            const bool direct = I <= J;
            const int voff = direct ? voff_d : voff_t;
            const int soff = tile_bytes(J) + (direct ? half * 64 : half * 64 * 128) * (int)sizeof(float);
            const int ustep = (direct ? 16 : 8) * 128 * (int)sizeof(float);
            const int a0 = (J * 8 + half * 4) * MT * 64 * 16;  // the X planes' slice for block J, this half: byte offset inside a plane
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < 8; u++) {
                st.gq[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, soff + u * ustep, 0));
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int v = 0; v < NA; v++) {
                const __amdgpu_buffer_rsrc_t &ra = (v / MT == 0) ? rs_hi : ((v / MT == 1) ? rs_lo : rs_lo2);
                st.aq[v] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(ra, 16 * lt, a0 + (v % MT) * 256 * 16, 0));
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        auto gstore = [&](int cc, const Stage &st) {                 // into image cc & 1
            float *gb = (cc & 1) ? gb1 : gb0;
            bf16x8 *ab = (cc & 1) ? ab1 : ab0;
            const bool direct = I <= partner(cc);
            const int loff = direct ? loff_d : loff_t, lstep = direct ? 16 * RK_RS : 8 * SYM_TS;
#pragma unroll
            for (int u = 0; u < 8; u++) *reinterpret_cast<f32x4 *>(gb + loff + u * lstep) = st.gq[u];
#pragma unroll
            for (int v = 0; v < NA; v++) ab[lt + 256 * v] = st.aq[v];
        };
        // chunk q: multiplied in iteration q, copied to image q & 1 at the start of iteration q - 1, requested at iteration
        // q - 3 into register set q & 1 (chunks 0..2 before the loop); no branch around a load in the steady state
        // (nchunks is even and >= 2 here; the prologue is unconditional - chunk 2 is clamped to the last one, a harmless
        // re-load when there are only two - because a branch around a load would again leave the loop header with a
        // merged, conservative wait state)
        gload(0, s0);
        gload(1, s1);
        gstore(0, s0);
        gload(min(2, nchunks - 1), s0);
        __syncthreads();
        int cc = 0;
        for (; cc + 4 < nchunks; cc += 2) {
            gstore(cc + 1, s1); gload(cc + 3, s1); __syncthreads();
            gstore(cc + 2, s0); gload(cc + 4, s0); __syncthreads();
        }
        // tail: iterations cc .. nchunks - 1 (cc = nchunks - 4 when nchunks >= 4, else 0), the same sequence with guards
        // (cc is even here and so is nchunks: iterations in pairs, each with its own register set - a set chosen at run time
        // would put both into scratch memory)
        for (; cc < nchunks; cc += 2) {
            if (cc + 1 < nchunks) gstore(cc + 1, s1);
            if (cc + 3 < nchunks) gload(cc + 3, s1);
            __syncthreads();
            if (cc + 2 < nchunks) gstore(cc + 2, s0);
            if (cc + 4 < nchunks) gload(cc + 4, s0);
            __syncthreads();
        }
        __syncthreads();                                           // the compute waves' K-half exchange
        return;
    }

    // ---- compute: wave = (32-column tile of block I, half of the chunk's four K steps), operands from LDS only ------
    const int lc = lane & 31, lh = lane >> 5;
    const int nt = wave & 3, kh = wave >> 2;
    f32x16 acc[MT];
#pragma unroll
    for (int a = 0; a < MT; a++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[a][r] = 0.f;

    struct Frag { f32x4 g0, g1; bf16x8 a[NS][MT]; };
    auto fetch = [&](const float *gb, const bf16x8 *ab, bool direct, int q, Frag &f) {
        if (direct) {
            const float *pp = gb + (nt * 32 + lc) * RK_RS + 16 * q + 8 * lh;
            f.g0 = *reinterpret_cast<const f32x4 *>(pp);
            f.g1 = *reinterpret_cast<const f32x4 *>(pp + 4);
        } else {
            const float *pp = gb + (16 * q + 8 * lh) * SYM_TS + nt * 32 + lc;
#pragma unroll
            for (int j = 0; j < 4; j++) { f.g0[j] = pp[j * SYM_TS]; f.g1[j] = pp[(4 + j) * SYM_TS]; }
        }
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
            const float r1 = v - (float)bh[j];
            bl[j] = (__bf16)r1;
            if (NS == 3) bl2[j] = (__bf16)(r1 - (float)bl[j]);
        }
#pragma unroll
        for (int t = 0; t < MT; t++) {
            if (NS == 3) {                                   // smallest terms first
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[0][t], bl2, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[NS - 1][t], bh, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[1][t], bl, acc[t], 0, 0, 0);
            }
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[0][t], bl, acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[1][t], bh, acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[0][t], bh, acc[t], 0, 0, 0);
        }
    };
    // (Two fragment sets per wave - the second K step's reads in flight under the first one's MFMAs - were measured and
    // changed nothing: 59.3 against 59.6 us per pass.  Ablation of this kernel at M = 96, two-way, per pass incl. 10 us of
    // split + reduce: 60 us as is, 61 us without the G loads, 44 us without the compute waves' work, 32 us with neither -
    // the loaders' LDS stores and the compute waves' LDS reads and conversions add up rather than overlap.)
    Frag fa;
    __syncthreads();
    for (int cc = 0; cc < nchunks; cc++) {
        const float *gb = (cc & 1) ? gb1 : gb0;
        const bf16x8 *ab = (cc & 1) ? ab1 : ab0;
        const bool direct = I <= partner(cc);
        fetch(gb, ab, direct, 2 * kh, fa);
        multiply(fa);
        fetch(gb, ab, direct, 2 * kh + 1, fa);
        multiply(fa);
        __syncthreads();
    }

    // ---- the two K halves of a column tile meet in LDS; the raw group sum goes to its slab ---------------------------
    float *red = reinterpret_cast<float *>(lds_raw);
    if (kh == 1) {
#pragma unroll
        for (int a = 0; a < MT; a++)
#pragma unroll
            for (int r = 0; r < 16; r++) red[((nt * MT + a) * 16 + r) * 64 + lane] = acc[a][r];
    }
    __syncthreads();
    if (kh == 0) {
        float *slab = g.slab + (long)p * g.M * g.K;
        const int col = I * 128 + nt * 32 + lc;
#pragma unroll
        for (int a = 0; a < MT; a++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const float sum = acc[a][r] + red[((nt * MT + a) * 16 + r) * 64 + lane];
                const int i = a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (i < g.M) slab[(long)i * g.K + col] = sum;
            }
    }
}

template <int MT, int NS>
void launch_sym(const SymDev &g, hipStream_t s)
{
    const size_t lds = rk_lds_bytes(MT, NS);
    if (g.nt == (1 << g.lb)) {
        ensure_dynamic_lds(reinterpret_cast<const void *>(skinny_sym_kernel<MT, NS, true>), (int)lds);
        hipLaunchKernelGGL((skinny_sym_kernel<MT, NS, true>), dim3(4u << g.lb), dim3(RK_T), lds, s, g);
    } else {
        ensure_dynamic_lds(reinterpret_cast<const void *>(skinny_sym_kernel<MT, NS, false>), (int)lds);
        hipLaunchKernelGGL((skinny_sym_kernel<MT, NS, false>), dim3(4u << g.lb), dim3(RK_T), lds, s, g);
    }
}

template <int MT, int NS>
void launch_rows(const Bf2Dev &g, dim3 grid, hipStream_t s)
{
    const size_t lds = rk_lds_bytes(MT, NS);
    ensure_dynamic_lds(reinterpret_cast<const void *>(skinny_rows_kernel<MT, NS>), (int)lds);
    hipLaunchKernelGGL((skinny_rows_kernel<MT, NS>), grid, dim3(RK_T), lds, s, g);
}

}  // namespace

// fragment-ordered bf16 planes of X [M][K] (M <= 160): hi, lo (and lo2 when given); see split_x_kernel
void split_planes_bf16(const float *X, long ldx, int M, int K, void *hi, void *lo, void *lo2, hipStream_t s)
{
    const int mt = (M + 31) / 32;
    const long total = (long)(K / 16) * mt * 64;
    hipLaunchKernelGGL(split_x_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, X, ldx, M, mt, K,
                       static_cast<bf16x8 *>(hi), static_cast<bf16x8 *>(lo), static_cast<bf16x8 *>(lo2));
    DLCO_HIP(hipGetLastError());
}

size_t bf16x2_plane_bytes(int M, int K) { return (size_t)((M + 31) / 32) * 32 * K * sizeof(__bf16); }
size_t bf16x2_slab_floats(int M, int N, int ksplit) { return (size_t)(ksplit > 0 ? ksplit : KS) * M * N; }

// Returns false when the shape is not supported (caller falls back to the fp32 kernel).
bool skinny_product_bf16x2(const float *X, long ldx, int M, const float *G, long ldg, int N, int K, float alpha, float *C,
                           long ldc, const float *E1, float b1, const float *E2, float b2, void *plane_hi, void *plane_lo,
                           float *slab, hipStream_t s, int ksplit, void *plane_lo2, bool g_tiled)
{
    const int mt = (M + 31) / 32;
    const int ks = ksplit > 0 ? ksplit : KS;
    if (M < 1 || mt > 5 || N % 128 != 0 || K % (16 * ks * 2 * 4) != 0) return false;   // 4 steps per loop trip
    if (ldx % 4 != 0 || (reinterpret_cast<uintptr_t>(X) & 15) != 0) return false;
    const long total = (long)(K / 16) * mt * 64;
    hipLaunchKernelGGL(split_x_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, X, ldx, M, mt, K,
                       static_cast<bf16x8 *>(plane_hi), static_cast<bf16x8 *>(plane_lo), static_cast<bf16x8 *>(plane_lo2));
    Bf2Dev g;
    g.M = M; g.N = N; g.K = K;
    g.xhi = static_cast<const bf16x8 *>(plane_hi); g.xlo = static_cast<const bf16x8 *>(plane_lo);
    g.xlo2 = static_cast<const bf16x8 *>(plane_lo2);
    g.G = G; g.ldg = ldg; g.slab = slab; g.tiled = g_tiled ? 1 : 0;
    static const int xcd_map = std::getenv("DLCO_NO_XCDMAP") == nullptr ? 1 : 0;
    g.xcd_map = xcd_map;
    const dim3 grid(N / 128, ks), block(T8);
    // whole symmetric matrix: stream its rows through LDS (skinny_rows_kernel)
    // (four row tiles with the three-way split do not fit the register budget of the split-role kernel)
    if (N == K && K % (128 * ks) == 0 && ldg % 4 == 0 && (reinterpret_cast<uintptr_t>(G) & 15) == 0 &&
        (long)K * ldg * 4 < (1L << 31) && !(mt >= 4 && plane_lo2)) {
        if (plane_lo2) {
            if (mt == 1) launch_rows<1, 3>(g, grid, s);
            else if (mt == 2) launch_rows<2, 3>(g, grid, s);
            else if (mt == 3) launch_rows<3, 3>(g, grid, s);
            else launch_rows<4, 3>(g, grid, s);
        } else {
            if (mt == 1) launch_rows<1, 2>(g, grid, s);
            else if (mt == 2) launch_rows<2, 2>(g, grid, s);
            else if (mt == 3) launch_rows<3, 2>(g, grid, s);
            else if (mt == 4) launch_rows<4, 2>(g, grid, s);
            else launch_rows<5, 2>(g, grid, s);               // 160 rows: blocks of rank ~128 + guards in ONE pass over G
        }
        DLCO_HIP(hipGetLastError());
        splitk_reduce_f32(slab, ks, M, N, C, ldc, alpha, 0.f, E1, b1, E2, b2, s);
        return true;
    }
    if (g_tiled || mt > 4) return false;                       // only the row-streaming kernel reads the tiled layout / takes five row tiles
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

// The same product for a symmetric G given as its packed upper tiles (see skinny_sym_kernel).
bool skinny_product_sym(const float *X, long ldx, int M, const float *Gpacked, int F, float alpha, float *C, long ldc,
                        const float *E1, float b1, const float *E2, float b2, void *plane_hi, void *plane_lo, float *slab,
                        hipStream_t s, void *plane_lo2, bool planes_ready, bool emit_planes)
{
    const int mt = (M + 31) / 32;
    const int nt = F / 128;
    if (F % 128 != 0 || nt < 1 || nt > 64 || M < 1 || mt > 5 || (mt >= 4 && plane_lo2)) return false;
    if (ldx % 4 != 0 || (reinterpret_cast<uintptr_t>(X) & 15) != 0 || (reinterpret_cast<uintptr_t>(Gpacked) & 15) != 0) return false;
    if ((planes_ready || emit_planes) && plane_lo2) return false;       // the carried planes are the two-way ones
    const long total = (long)(F / 16) * mt * 64;
    if (!planes_ready)
        hipLaunchKernelGGL(split_x_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, X, ldx, M, mt, F,
                           static_cast<bf16x8 *>(plane_hi), static_cast<bf16x8 *>(plane_lo), static_cast<bf16x8 *>(plane_lo2));
    SymDev g;
    g.M = M; g.K = F;
    g.nt = nt; g.lb = 2;
    while ((1 << g.lb) < nt) g.lb++;
    g.xhi = static_cast<const bf16x8 *>(plane_hi); g.xlo = static_cast<const bf16x8 *>(plane_lo); g.xlo2 = static_cast<const bf16x8 *>(plane_lo2);
    g.G = Gpacked; g.slab = slab;
    if (plane_lo2) {
        if (mt == 1) launch_sym<1, 3>(g, s);
        else if (mt == 2) launch_sym<2, 3>(g, s);
        else launch_sym<3, 3>(g, s);
    } else {
        if (mt == 1) launch_sym<1, 2>(g, s);
        else if (mt == 2) launch_sym<2, 2>(g, s);
        else if (mt == 3) launch_sym<3, 2>(g, s);
        else if (mt == 4) launch_sym<4, 2>(g, s);
        else launch_sym<5, 2>(g, s);
    }
    DLCO_HIP(hipGetLastError());
    if (emit_planes && ldc % 4 == 0 && (reinterpret_cast<uintptr_t>(C) & 15) == 0) {
        // the product kernel has read the planes of X: they now receive the planes of the result, rows up to 32 mt zeroed by
        // the split of the first product of the chain (rows >= M of a tile are never written here and stay zero)
        const long work = (long)M * (F / 8);
        hipLaunchKernelGGL(reduce_split_kernel, dim3((unsigned)std::min<long>((work + 255) / 256, 4096)), dim3(256), 0, s, slab, 4, M, mt, F, C, ldc,
                           alpha, E1, b1, E2, b2, static_cast<bf16x8 *>(plane_hi), static_cast<bf16x8 *>(plane_lo));
        DLCO_HIP(hipGetLastError());
    } else {
        splitk_reduce_f32(slab, 4, M, F, C, ldc, alpha, 0.f, E1, b1, E2, b2, s);
    }
    return true;
}

}  // namespace dlco
