// kernels_syrk.hip — the dominant kernel of the pj-learn step on gfx950:
//
//     dfAvg  <-  beta * dfAvg  +  alpha * sum_k  w_k * x_k x_k^T          (Q1 + U1)
//
// x_k = row ids[k] of the resident PR-descriptor-difference matrix D [N,F], w_k = rho_i for
// a positive row, -kappa_j for a negative one (src/pj-learn.cpp:367-422 in the reformulated
// order P^T diag(rho) P - N^T diag(kappa) N; rows with zero weight are not in the list).
//
// Design (MI355X): the output is symmetric, so only the 128x128 tiles on or above the
// diagonal are computed (nt(nt+1)/2 workgroups, dealt to the 8 XCDs in contiguous chunks of 8 x 8
// super-blocks so that neighbouring tiles share their row panels in one L2).  fp32 results come by
// default from the bf16 matrix cores: the active rows are split three ways ONCE (syrk_split_rows_kernel)
// and a pure matrix kernel fed by LDS-DMA multiplies the planes (syrk_planes_kernel, below).  The older
// kernel that gathers inside the tile loop (syrk_rda_kernel8) remains for v_mfma_f32_32x32x2_f32, a k-ordered
// fmaf chain (DLCO_SYRK_FP32=1); operands rounded to bf16 once (cfg.grad_bf16, BASELINE configs[4]) go through
// the same two new kernels (syrk_round_rows_kernel).  Epilogue everywhere: dual-average update in registers.
#include "dlco_internal.hpp"
#include "rank_coeff_dev.hpp"

#include <mutex>
#include <type_traits>
#include <vector>

namespace dlco {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

namespace {

constexpr int TB = 128;           // tile edge
constexpr int KB = 32;            // K depth per LDS stage
constexpr int LD = TB + 4;        // LDS row pitch of the k-major images
constexpr int TLD = TB + 4;       // pitch of the output re-layout buffer (16-byte aligned rows)

struct SyrkDev {
    const float *D;
    long ldd;
    const int32_t *ids;
    const int32_t *ids2;          // pair mode: row k = D[ids[k]] - D[ids2[k]] (src/comp-uprjdists.cpp:327), else nullptr
    const float *w;
    const int *k_dev;             // device-resident active row count
    int kmax;                     // capacity of ids/w (multiple of KB, zero padded)
    int F;
    float *C;
    long ldc;
    float alpha, beta;
    int nt;                       // tiles per edge
    int slab_t0, slab_nt;         // column-slab mode: tile columns [slab_t0, slab_t0 + slab_nt) only
    const int32_t *tile_map;      // tile number -> (bi << 16 | bj), see syrk_tile_map()
    unsigned long long *trace;    // developer aid (DLCO_SYRK_TRACE): 6 words per workgroup, see syrk_rda_f32
};

union SyrkLds {
    struct { float A[2][KB][LD]; float B[2][KB][LD]; } st;    // 67,584 B
    float T[TB][TLD];                                          // 66,048 B
};

// SLAB = false: the whole symmetric matrix (upper tiles computed, mirrored).
// SLAB = true : only the tile columns [slab_t0, slab_t0 + slab_nt) of the matrix, all tile rows,
//               no mirror — the rank's column slab of a dual average that is sharded over GPUs.
// BF16 = true : the products run on the bf16 matrix cores (v_mfma_f32_32x32x16_bf16, 16x the fp32 MFMA
//               rate) with fp32 accumulation: the operands w_k x_k and x_k are rounded to bf16 when the
//               fragments are read (BASELINE configs[4], "bf16 MFMA + fp32 accum"); everything else -
//               gather, weights, dual average, stores - stays fp32.  Opt-in (cfg.grad_bf16), gated by the
//               FPR@95 band in the tests: not the reference's arithmetic.
// ---------------------------------------------------------------------------------------------------
// Eight waves per workgroup.  A single wave per SIMD issues v_mfma_f32_32x32x2_f32 at HALF the pipe's
// rate (tools/overlap_probe.cpp: 4 workgroup-waves per CU need 292 us for what 8 do in 155 us), so a
// workgroup has eight waves - two per SIMD, each a 64 x 32 part of the tile - and saturates the pipe on
// its own; two such workgroups per CU then alternate freely between K loop and memory phases.  The old
// tile is requested inside the peeled last K block (no gather follows it, so it lands under that block's
// MFMAs; a wave has one in-order counter for its loads), with scalar row bases and one per-lane offset.
// (Rounds 1-2 measured seven other structures - four-wave workgroups, three / four workgroups per CU,
// persistent wave-specialised workgroups, start-up staggers - all bit-identical and none faster; the
// table is in DESIGN.md section 3, the code in the history.)
// ---------------------------------------------------------------------------------------------------
constexpr int NT8 = 512;

// PACKED = true: C holds only the tiles on or above the diagonal, each a contiguous row-major 128 x 128 block, tile
//               (bi, bj) at tile index bi*nt - bi*(bi-1)/2 + (bj - bi) (the layout the symmetric tracker product reads,
//               kernels_bf16x2.hip): no mirrored store at all - half the bytes written, and 136 MB instead of 268 MB at
//               F = 8192.  Diagonal tiles are stored whole and exactly symmetric.
template <bool PAIR, bool SLAB, int BF16, bool PACKED = false, int UNR = 16>
__global__ __launch_bounds__(NT8, 4) void syrk_rda_kernel8(SyrkDev g)
{
    __shared__ __attribute__((aligned(16))) SyrkLds lds;
    const unsigned long long tr0 = g.trace ? wall_clock64() : 0ull;

    const int ntiles = SLAB ? g.nt * g.slab_nt : g.nt * (g.nt + 1) / 2;
    const int nxcd = 8;
    const int bid = blockIdx.x;
    int t;
    {
        const int q = ntiles / nxcd, r = ntiles % nxcd, xcd = bid % nxcd, within = bid / nxcd;
        t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + within;
    }
    const int code = g.tile_map[t];
    const int bi = code >> 16, bj = code & 0xffff;
    const int i0 = bi * TB, j0 = bj * TB;
    // where this tile's old and new contents live: a window of the full matrix, or its own packed block
    float *const ctile = PACKED ? g.C + (long)(bi * g.nt - bi * (bi - 1) / 2 + (bj - bi)) * (TB * TB) : g.C + (long)i0 * g.ldc + j0;
    const long cld = PACKED ? TB : g.ldc;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3;                  // rows wm*64.., columns wn*32.. of the tile
    const int lr = lane & 31, lk = lane >> 5;

    constexpr int KD = KB;
    const int kact = min(*g.k_dev, g.kmax);
    const int nk = (kact + KD - 1) / KD;

    f32x16 acc[2];
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[a][r] = 0.f;

    // loader mapping (fp32 / bf16-once): 2 float4 per operand per thread; rows rbase, rbase + 16 of the K block
    const int c4 = tid & 31, rbase = tid >> 5;
    int32_t id_nx[2], id2_nx[2];
    float w_nx[2];
    f32x4 ra[2], rb[2];
    auto load_ids = [&](int kt) {
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const int k = kt * KB + rbase + 16 * u;
            id_nx[u] = g.ids[k];
            if (PAIR) id2_nx[u] = g.ids2[k];
            w_nx[u] = g.w[k];
        }
    };
    auto load_rows = [&]() {
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const float *row = g.D + (long)id_nx[u] * g.ldd;
            f32x4 xa = *reinterpret_cast<const f32x4 *>(row + i0 + c4 * 4);
            f32x4 xb = *reinterpret_cast<const f32x4 *>(row + j0 + c4 * 4);
            if (PAIR) {
                const float *row2 = g.D + (long)id2_nx[u] * g.ldd;
                xa -= *reinterpret_cast<const f32x4 *>(row2 + i0 + c4 * 4);
                xb -= *reinterpret_cast<const f32x4 *>(row2 + j0 + c4 * 4);
            }
            ra[u] = xa * w_nx[u];
            rb[u] = xb;
        }
    };
    auto store_rows = [&](int buf) {
#pragma unroll
        for (int u = 0; u < 2; u++) {
            *reinterpret_cast<f32x4 *>(&lds.st.A[buf][rbase + 16 * u][c4 * 4]) = ra[u];
            *reinterpret_cast<f32x4 *>(&lds.st.B[buf][rbase + 16 * u][c4 * 4]) = rb[u];
        }
    };

    if (nk > 0) load_ids(0);
    // old tile: requested inside the last K block, scalar row base + one lane offset
    float oldv[2][16];
    const bool use_old = (g.beta != 0.f);
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int old_lane = 4 * lk * (int)cld + lr;
    auto fetch_old = [&]() {
#pragma unroll
        for (int a = 0; a < 2; a++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const float *rowb = ctile + (long)((wave_u >> 2) * 64 + a * 32 + (r & 3) + 8 * (r >> 2)) * cld + (wave_u & 3) * 32;
                oldv[a][r] = use_old ? rowb[old_lane] : 0.f;
            }
    };
    if (nk == 0) fetch_old();
    if (nk > 0) {
        load_rows();
        if (nk > 1) load_ids(1);
        store_rows(0);
    }
    __syncthreads();
    const unsigned long long tr1 = g.trace ? wall_clock64() : 0ull;
    // one K block; the last one is peeled so that the old tile's 32 loads (and their addresses) exist only there
    auto kblock = [&](int kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) load_rows();
        if (kt + 2 < nk) load_ids(kt + 2);
        if (BF16 == 1) {
#pragma unroll
            for (int ks = 0; ks < KB / 16; ks++) {
                bf16x8 a0, a1, b0;
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const int k = 16 * ks + 8 * lk + j;
                    a0[j] = (__bf16)lds.st.A[buf][k][wm * 64 + lr];
                    a1[j] = (__bf16)lds.st.A[buf][k][wm * 64 + 32 + lr];
                    b0[j] = (__bf16)lds.st.B[buf][k][wn * 32 + lr];
                }
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc[1], 0, 0, 0);
            }
        } else {
#pragma unroll UNR
            for (int kk = 0; kk < KB / 2; kk++) {
                const float a0 = lds.st.A[buf][2 * kk + lk][wm * 64 + lr];
                const float a1 = lds.st.A[buf][2 * kk + lk][wm * 64 + 32 + lr];
                const float b0 = lds.st.B[buf][2 * kk + lk][wn * 32 + lr];
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1], 0, 0, 0);
            }
        }
        if (kt + 1 < nk) store_rows(buf ^ 1);
        __syncthreads();
    };
    for (int kt = 0; kt + 1 < nk; kt++) kblock(kt);
    if (nk > 0) {
        fetch_old();
        kblock(nk - 1);
    }

    const unsigned long long tr2 = g.trace ? wall_clock64() : 0ull;
    // ---- epilogue: dual average in registers, mirrored store straight from the accumulator layout, the tile
    // itself through an LDS re-layout into whole 512-byte rows ------------------------------------------
    const bool diag = (bi == bj);
    const int jl = wn * 32 + lr;
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int q = 0; q < 4; q++) {
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; e++) o[e] = g.alpha * acc[a][4 * q + e] + g.beta * oldv[a][4 * q + e];
            const int il0 = wm * 64 + a * 32 + 8 * q + 4 * lk;           // rows il0 .. il0+3
            if (!SLAB && !PACKED && !diag) *reinterpret_cast<f32x4 *>(&g.C[(long)(j0 + jl) * g.ldc + (i0 + il0)]) = o;
#pragma unroll
            for (int e = 0; e < 4; e++) lds.T[il0 + e][jl] = o[e];
        }
    __syncthreads();
    for (int f = tid; f < TB * (TB / 4); f += NT8) {
        const int il = f / (TB / 4), cc = (f % (TB / 4)) * 4;
        f32x4 v = *reinterpret_cast<const f32x4 *>(&lds.T[il][cc]);
        if (diag) {
#pragma unroll
            for (int e = 0; e < 4; e++)
                if (cc + e < il) v[e] = lds.T[cc + e][il];
        }
        *reinterpret_cast<f32x4 *>(&ctile[(long)il * cld + cc]) = v;
    }
    if (g.trace && tid == 0) {
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        unsigned long long *o = g.trace + (size_t)blockIdx.x * 6;
        o[0] = ((unsigned long long)xcc << 32) | hw; o[1] = tr0; o[2] = tr1; o[3] = tr2; o[4] = wall_clock64(); o[5] = (unsigned long long)t;
    }
}

// ---------------------------------------------------------------------------------------------------
// fp32 results from the bf16 matrix cores: split ONCE, then a pure matrix kernel
// ---------------------------------------------------------------------------------------------------
// Every value that enters the product is split three ways, x = hi + mid + lo (bf16 each: all 24 mantissa bits), and the
// K loop forms the six products down to 2^-16 (hh, hm, mh, mm, hl, lh; smallest first, fp32 accumulation): 6/16 of
// the fp32-MFMA time.  Round 3's first form did the gather and the split inside the tile kernel, one 16-row block ahead
// of the MFMAs: 2080 tiles each gathered and converted their own operands (every row panel 64 times over), the gather's
// latency had one short K block to hide in, and the matrix pipe was 27 % busy (profiles/r3_pmc_sq.json).  Now
//   * syrk_split_rows_kernel gathers the active rows once, forms w_k x_k and x_k (pair mode: of D[ids] - D[ids2]),
//     splits them and writes bf16 planes in the order the tile kernel's LDS image has: per (operand, 16-row K block,
//     128-column tile) one contiguous 12 KB image [plane][k / 8][column] of 16-byte entries (the eight consecutive k of
//     a column: one MFMA fragment of one lane).  30 MB at K = 305, F = 8192; read back from L2 / Infinity Cache.
//   * syrk_planes_kernel computes a 128 x 128 tile with four waves (64 x 64 each: 12 fragment reads feed 24 MFMAs per
//     K block).  The images go global -> LDS by LDS-DMA (global_load_lds_dwordx4: no registers, no ds_write), three
//     stages, two K blocks in flight while one is multiplied: counted vmcnt, one raw s_barrier per K block and nothing
//     else in the loop but ds_read_b128 and MFMA.  Two workgroups per CU (72 KB of LDS each).
// The arithmetic (w_k x_k rounded once in fp32, the split, the order of the six products and of the K blocks) is that
// of the first form: results are bit-identical to it.
constexpr int PL_KD = 16;                 // K rows per block
constexpr int PL_IMG = 3 * 2 * TB * 16;   // bytes of one (operand, K block, column tile) image: 12,288
constexpr int PL_STAGE = 2 * PL_IMG;      // A image + B image
constexpr int NTP = 256;

// (coeff.frag != nullptr: the launch has one more column of blocks, which writes the coefficient fragments of the
// tracker's rank-update first term for their K block - a by-product of the same row list, see kernels_rankupd.hip)
__global__ __launch_bounds__(256) void syrk_split_rows_kernel(const float *D, long ldd, const int32_t *ids, const int32_t *ids2,
                                                              const float *w, const int *k_dev, int kmax, int nt, char *planes,
                                                              RankCoeffJob coeff)
{
    const int ct = blockIdx.x, kb = blockIdx.y;
    const int kact = min(*k_dev, kmax);
    if (kb * PL_KD >= kact) return;
    if (ct == nt) {
        for (int tile = threadIdx.x >> 6; tile < coeff.MT; tile += 4) rank_coeff_block(coeff, w, kact, kb, tile, threadIdx.x & 63);
        return;
    }
    const int col = threadIdx.x & (TB - 1), lk = threadIdx.x >> 7;
    const int nkb = kmax / PL_KD;
    float v[8], wv[8];
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const int k = kb * PL_KD + 8 * lk + j;
        const float *row = D + (long)ids[k] * ldd + ct * TB + col;
        v[j] = row[0];
        if (ids2) v[j] -= D[(long)ids2[k] * ldd + ct * TB + col];      // Dist = Desc1 - Desc2, src/comp-uprjdists.cpp:327
        wv[j] = w[k];
    }
    bf16x8 e[2][3];
#pragma unroll
    for (int j = 0; j < 8; j++)
#pragma unroll
        for (int op = 0; op < 2; op++) {
            const float x = op == 0 ? v[j] * wv[j] : v[j];              // w_k x_k for the A operand
            const __bf16 h = (__bf16)x;
            const float r1 = x - (float)h;
            const __bf16 m = (__bf16)r1;
            e[op][0][j] = h; e[op][1][j] = m; e[op][2][j] = (__bf16)(r1 - (float)m);
        }
#pragma unroll
    for (int op = 0; op < 2; op++) {
        bf16x8 *img = reinterpret_cast<bf16x8 *>(planes + (((long)op * nkb + kb) * nt + ct) * PL_IMG);
#pragma unroll
        for (int pl = 0; pl < 3; pl++) img[(pl * 2 + lk) * TB + col] = e[op][pl];
    }
}

// The bf16-once arithmetic (cfg.grad_bf16, BASELINE configs[4]: operands rounded to bf16 once, fp32 accumulation) through the
// same two kernels: the image's three plane slots hold three consecutive 16-row sub-blocks of a 48-row K block (hi parts
// only), and the tile kernel multiplies slot s of A with slot s of B - three MFMAs per accumulator and K block instead of six.
__global__ __launch_bounds__(256) void syrk_round_rows_kernel(const float *D, long ldd, const int32_t *ids, const int32_t *ids2,
                                                              const float *w, const int *k_dev, int kmax, int nt, char *planes)
{
    const int ct = blockIdx.x, kb = blockIdx.y;
    const int kact = min(*k_dev, kmax);
    if (kb * 3 * PL_KD >= kact) return;
    const int col = threadIdx.x & (TB - 1), lk = threadIdx.x >> 7;
    const int nkb = kmax / PL_KD;                              // (the buffer is sized for 16-row blocks; a third of them are used)
#pragma unroll
    for (int sb = 0; sb < 3; sb++) {
        bf16x8 e[2];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int k = (kb * 3 + sb) * PL_KD + 8 * lk + j;
            float v = 0.f, wv = 0.f;
            if (k < kmax) {                                      // rows in [kact, kmax) are zero-weight padding of the list
                v = D[(long)ids[k] * ldd + ct * TB + col];
                if (ids2) v -= D[(long)ids2[k] * ldd + ct * TB + col];
                wv = w[k];
            }
            e[0][j] = (__bf16)(v * wv);
            e[1][j] = (__bf16)v;
        }
#pragma unroll
        for (int op = 0; op < 2; op++) {
            bf16x8 *img = reinterpret_cast<bf16x8 *>(planes + (((long)op * nkb + kb) * nt + ct) * PL_IMG);
            img[(sb * 2 + lk) * TB + col] = e[op];
        }
    }
}

template <bool SLAB, bool PACKED, int NST, bool ONCE = false>
__global__ __launch_bounds__(NTP, NST == 2 ? 3 : 2) void syrk_planes_kernel(SyrkDev g, const char *planes)
{
    constexpr int PL_NSTAGE = NST;
    constexpr int KD = ONCE ? 3 * PL_KD : PL_KD;               // K rows per image (see syrk_round_rows_kernel)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const unsigned long long tr0 = g.trace ? wall_clock64() : 0ull;
    const int ntiles = SLAB ? g.nt * g.slab_nt : g.nt * (g.nt + 1) / 2;
    const int nxcd = 8;
    const int bid = blockIdx.x;
    int t;
    {
        const int q = ntiles / nxcd, r = ntiles % nxcd, xcd = bid % nxcd, within = bid / nxcd;
        t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + within;
    }
    const int code = g.tile_map[t];
    const int bi = code >> 16, bj = code & 0xffff;
    const int i0 = bi * TB, j0 = bj * TB;
    float *const ctile = PACKED ? g.C + (long)(bi * g.nt - bi * (bi - 1) / 2 + (bj - bi)) * (TB * TB) : g.C + (long)i0 * g.ldc + j0;
    const long cld = PACKED ? TB : g.ldc;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;                  // rows wm*64.., columns wn*64.. of the tile
    const int lr = lane & 31, lk = lane >> 5;
    const int kact = min(*g.k_dev, g.kmax);
    const int nk = (kact + KD - 1) / KD;
    const int nkb = g.kmax / PL_KD;

    // this wave's six pieces (1 KB each) of a stage: waves 0, 1 bring the A image (w x of tile row bi), waves 2, 3 the B image
    const long kstep = (long)g.nt * PL_IMG;
    const char *src0 = planes + ((wave < 2 ? (long)bi : (long)nkb * g.nt + bj)) * PL_IMG + (wave & 1) * 6 * 1024 + lane * 16;
    auto issue = [&](int kb, int stage) {
        const char *src = src0 + kb * kstep;
        char *dst = smem + stage * PL_STAGE + wave * 6 * 1024;
#pragma unroll
        for (int p = 0; p < 6; p++)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + p * 1024),
                                             (__attribute__((address_space(3))) void *)(dst + p * 1024), 16, 0, 0);
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[a][b][r] = 0.f;

    // old tile: 64 values per lane (scalar row base + one lane offset), requested inside the last K block (no DMA follows it)
    float oldv[2][2][16];
    const bool use_old = (g.beta != 0.f);
    const int old_lane = 4 * lk * (int)cld + lr;
    auto fetch_quadrant = [&](int a, int b, float (&dst)[16]) {  // unconditional (the tile's memory exists); discarded below when beta = 0
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const float *rowb = ctile + (long)(wm * 64 + a * 32 + (r & 3) + 8 * (r >> 2)) * cld + wn * 64 + b * 32;
            dst[r] = __builtin_nontemporal_load(rowb + old_lane);          // streamed once: keep the L2 for the planes
        }
    };
    constexpr int DEPTH = NST - 1;                            // K blocks requested ahead of the one being multiplied
    if (nk > 0) issue(0, 0);
    if (DEPTH > 1 && nk > 1) issue(1, 1);
    const unsigned long long tr1 = g.trace ? wall_clock64() : 0ull;
    int stage = 0;
    // One K block.  LAST: the old tile is requested here; VM: how many younger memory operations than block kb's pieces
    // this wave has in flight at the top; ISSUE: block kb + DEPTH exists.
    auto iteration = [&](int kb, auto last_c, auto vm_c, auto issue_c) {
        constexpr int VM = decltype(vm_c)::value;
        constexpr bool LAST = decltype(last_c)::value, ISSUE = decltype(issue_c)::value;
        // my pieces of block kb have landed; after the barrier everybody's have, and everybody has finished reading the
        // stage that block kb + DEPTH is about to overwrite
        if constexpr (VM == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if constexpr (LAST) {
            fetch_quadrant(0, 0, oldv[0][0]); fetch_quadrant(0, 1, oldv[0][1]); fetch_quadrant(1, 0, oldv[1][0]); fetch_quadrant(1, 1, oldv[1][1]);
        }
        const bf16x8 *Ai = reinterpret_cast<const bf16x8 *>(smem + stage * PL_STAGE);
        const bf16x8 *Bi = reinterpret_cast<const bf16x8 *>(smem + stage * PL_STAGE + PL_IMG);
        bf16x8 fa[2][3], fb[2][3];
#pragma unroll
        for (int pl = 0; pl < 3; pl++)
#pragma unroll
            for (int x = 0; x < 2; x++) {
                fa[x][pl] = Ai[(pl * 2 + lk) * TB + wm * 64 + 32 * x + lr];
                fb[x][pl] = Bi[(pl * 2 + lk) * TB + wn * 64 + 32 * x + lr];
            }
        if constexpr (ISSUE) issue(kb + DEPTH, stage >= 1 ? stage - 1 : PL_NSTAGE - 1);
#pragma unroll
        for (int a = 0; a < 2; a++)
#pragma unroll
            for (int b = 0; b < 2; b++) {
                if constexpr (ONCE) {
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[a][0], fb[b][0], acc[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[a][1], fb[b][1], acc[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[a][2], fb[b][2], acc[a][b], 0, 0, 0);
                } else {
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[a][0], fb[b][2], acc[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[a][2], fb[b][0], acc[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[a][1], fb[b][1], acc[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[a][0], fb[b][1], acc[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[a][1], fb[b][0], acc[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[a][0], fb[b][0], acc[a][b], 0, 0, 0);
                }
            }
        // issue order inside the iteration: fragment reads first, then the six DMA pieces spread between the MFMAs (a piece
        // costs the wave ~60 cycles of issue; between two MFMAs that is time the matrix pipe spends on the MFMA before)
        if constexpr (ISSUE) {
            __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);
#pragma unroll
            for (int u = 0; u < 6; u++) {
                __builtin_amdgcn_sched_group_barrier(0x008, ONCE ? 1 : 3, 0);
                __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
            }
        }
        stage = stage + 1 < PL_NSTAGE ? stage + 1 : 0;
    };
    using std::integral_constant;
    typedef integral_constant<bool, true> Yes;
    typedef integral_constant<bool, false> No;
    typedef integral_constant<int, 6 * (DEPTH - 1)> Ahead;   // the pieces of the blocks between kb and kb + DEPTH
    typedef integral_constant<int, 0> None;
    if (nk > 0) {
        for (int kb = 0; kb + DEPTH < nk; kb++) iteration(kb, No{}, Ahead{}, Yes{});
        if (DEPTH > 1 && nk > 1) iteration(nk - 2, No{}, Ahead{}, No{});
        iteration(nk - 1, Yes{}, None{}, No{});
    } else {
        fetch_quadrant(0, 0, oldv[0][0]); fetch_quadrant(0, 1, oldv[0][1]); fetch_quadrant(1, 0, oldv[1][0]); fetch_quadrant(1, 1, oldv[1][1]);
    }
    const unsigned long long tr2 = g.trace ? wall_clock64() : 0ull;

    // ---- epilogue: dual average in registers, every store straight from the accumulator layout (a lane holds a column:
    // 32 lanes write 128 contiguous bytes of a row; the mirrored copy of the unpacked layout takes four rows of a column
    // as 16 bytes).  A diagonal tile is made exactly symmetric: what lies above the diagonal is stored twice, what lies
    // below it is not stored at all. ---------------------------------------------------------------------------------
    const bool diag = (bi == bj);
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++) {
            const int jl = wn * 64 + b * 32 + lr;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; e++) o[e] = g.alpha * acc[a][b][4 * q + e] + (use_old ? g.beta * oldv[a][b][4 * q + e] : 0.f);
                const int il0 = wm * 64 + a * 32 + 8 * q + 4 * lk;           // rows il0 .. il0+3
                if (diag) {
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        const int il = il0 + e;
                        if (il <= jl) __builtin_nontemporal_store(o[e], &ctile[(long)il * cld + jl]);
                        if (il < jl) __builtin_nontemporal_store(o[e], &ctile[(long)jl * cld + il]);
                    }
                } else {
                    if (!SLAB && !PACKED) *reinterpret_cast<f32x4 *>(&g.C[(long)(j0 + jl) * g.ldc + (i0 + il0)]) = o;
#pragma unroll
                    for (int e = 0; e < 4; e++) __builtin_nontemporal_store(o[e], &ctile[(long)(il0 + e) * cld + jl]);
                }
            }
        }
    if (g.trace && tid == 0) {
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        unsigned long long *o = g.trace + (size_t)blockIdx.x * 6;
        o[0] = ((unsigned long long)xcc << 32) | hw; o[1] = tr0; o[2] = tr1; o[3] = tr2; o[4] = wall_clock64(); o[5] = (unsigned long long)t;
    }
}

// lower triangle := upper triangle (32 x 32 blocks through LDS)
__global__ __launch_bounds__(256) void mirror_upper_kernel(float *C, long ldc, int F)
{
    __shared__ float t[32][33];
    const int bi = blockIdx.y, bj = blockIdx.x;
    if (bj < bi) return;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int r = ty; r < 32; r += 8) {
        const int i = bi * 32 + r, j = bj * 32 + tx;
        t[r][tx] = (i < F && j < F) ? C[(long)i * ldc + j] : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int i = bj * 32 + r, j = bi * 32 + tx;          // element (i, j) of the lower side = upper (j, i)
        if (i < F && j < F && i > j) C[(long)i * ldc + j] = t[tx][r];
    }
}

// full symmetric F x F matrix <-> its packed upper tiles (see PACKED above); one workgroup per tile
__global__ __launch_bounds__(256) void pack_tiles_kernel(const float *C, long ldc, int nt, float *P, int unpack_into_C)
{
    const int bi = blockIdx.y, bj = blockIdx.x;
    if (bj < bi) return;
    float *tile = P + (long)(bi * nt - bi * (bi - 1) / 2 + (bj - bi)) * (TB * TB);
    for (int f = threadIdx.x; f < TB * (TB / 4); f += 256) {
        const int il = f / (TB / 4), cc = (f % (TB / 4)) * 4;
        float *full = const_cast<float *>(C) + (long)(bi * TB + il) * ldc + bj * TB + cc;
        if (!unpack_into_C) { *reinterpret_cast<f32x4 *>(tile + il * TB + cc) = *reinterpret_cast<const f32x4 *>(full); continue; }
        const f32x4 v = *reinterpret_cast<const f32x4 *>(tile + il * TB + cc);
        *reinterpret_cast<f32x4 *>(full) = v;
        if (bi != bj) {
#pragma unroll
            for (int e = 0; e < 4; e++) const_cast<float *>(C)[(long)(bj * TB + cc + e) * ldc + bi * TB + il] = v[e];
        }
    }
}

}  // namespace

size_t syrk_packed_floats(int F) { const long nt = F / TB; return (size_t)(nt * (nt + 1) / 2) * TB * TB; }

void syrk_pack_upper(const float *C, long ldc, int F, float *packed, hipStream_t s)
{
    const int nt = F / TB;
    hipLaunchKernelGGL(pack_tiles_kernel, dim3(nt, nt), dim3(256), 0, s, C, ldc, nt, packed, 0);
    DLCO_HIP(hipGetLastError());
}

void syrk_unpack_upper(const float *packed, int F, float *C, long ldc, hipStream_t s)
{
    const int nt = F / TB;
    hipLaunchKernelGGL(pack_tiles_kernel, dim3(nt, nt), dim3(256), 0, s, C, ldc, nt, const_cast<float *>(packed), 1);
    DLCO_HIP(hipGetLastError());
}

void syrk_mirror_upper(float *C, long ldc, int F, hipStream_t s)
{
    const int nb = (F + 31) / 32;
    hipLaunchKernelGGL(mirror_upper_kernel, dim3(nb, nb), dim3(256), 0, s, C, ldc, F);
    DLCO_HIP(hipGetLastError());
}

static void syrk_dump_trace(const char *path, const unsigned long long *buf, int ntiles, hipStream_t s)
{
    DLCO_HIP(hipStreamSynchronize(s));
    std::vector<unsigned long long> h((size_t)6 * ntiles);
    DLCO_HIP(hipMemcpy(h.data(), buf, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    if (FILE *f = std::fopen(path, "w")) {
        std::fprintf(f, "# workgroup xcc hw_id t_start t_kloop t_epilogue t_end tile   (wall_clock64 ticks, 100 MHz)\n");
        for (int i = 0; i < ntiles; i++)
            std::fprintf(f, "%d %llu %llu %llu %llu %llu %llu %llu\n", i, h[6 * i] >> 32, h[6 * i] & 0xffffffffull, h[6 * i + 1], h[6 * i + 2], h[6 * i + 3],
                         h[6 * i + 4], h[6 * i + 5]);
        std::fclose(f);
    }
}

// Tile list of a launch: 8 x 8 super-blocks of 128 x 128 tiles in row-major order, tiles row-major inside one;
// the symmetric matrix keeps the tiles on or above the diagonal.  Built once per shape and kept on the device.
static const int32_t *syrk_tile_map(int nt, int slab_t0, int slab_nt, int *count)
{
    struct Entry { int nt, t0, snt, dev; int32_t *p; int n; };
    static std::vector<Entry> cache;
    static std::mutex mu;
    int dev = 0;
    DLCO_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(mu);
    for (const Entry &e : cache)
        if (e.nt == nt && e.t0 == slab_t0 && e.snt == slab_nt && e.dev == dev) { *count = e.n; return e.p; }
    constexpr int S = 8;
    std::vector<int32_t> m;
    const int c0 = slab_nt > 0 ? slab_t0 : 0, c1 = slab_nt > 0 ? slab_t0 + slab_nt : nt;
    for (int I = 0; I < nt; I += S)
        for (int J = c0; J < c1; J += S)
            for (int bi = I; bi < std::min(I + S, nt); bi++)
                for (int bj = J; bj < std::min(J + S, c1); bj++)
                    if (slab_nt > 0 || bi <= bj) m.push_back((bi << 16) | bj);
    Entry e{nt, slab_t0, slab_nt, dev, nullptr, (int)m.size()};
    DLCO_HIP(hipMalloc((void **)&e.p, m.size() * sizeof(int32_t)));
    DLCO_HIP(hipMemcpy(e.p, m.data(), m.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    cache.push_back(e);
    *count = e.n;
    return e.p;
}


int syrk_planes_mode(bool bf16) { return bf16 ? 1 : (std::getenv("DLCO_SYRK_FP32") == nullptr ? 3 : 0); }

size_t syrk_planes_bytes(int kmax, int F) { return (size_t)2 * (kmax / PL_KD) * (F / TB) * PL_IMG; }

// Workspace of the split planes for callers that bring none: one per device, grown on demand (2 operands x 6 bytes per staged value).
static char *syrk_planes_buffer(size_t bytes)
{
    struct Entry { int dev; char *p; size_t cap; };
    static std::vector<Entry> cache;
    static std::mutex mu;
    int dev = 0;
    DLCO_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(mu);
    for (Entry &e : cache)
        if (e.dev == dev) {
            if (e.cap < bytes) {
                DLCO_HIP(hipDeviceSynchronize());              // launches that still read the old buffer
                DLCO_HIP(hipFree(e.p));
                e.p = nullptr; e.cap = 0;
                DLCO_HIP(hipMalloc((void **)&e.p, bytes));
                e.cap = bytes;
            }
            return e.p;
        }
    Entry e{dev, nullptr, bytes};
    DLCO_HIP(hipMalloc((void **)&e.p, bytes));
    cache.push_back(e);
    return e.p;
}

bool syrk_rda_f32(const float *D, long ldd, const int32_t *ids, const int32_t *ids2, const float *w, const int *k_dev,
                  int kmax, int F, float alpha, float beta, float *C, long ldc, hipStream_t s, int slab_col0, int slab_cols, bool bf16,
                  bool packed, void *planes_ws, const RankCoeffJob *coeff_job)
{
    if (F % TB != 0 || kmax % KB != 0 || ldd % 4 != 0 || (reinterpret_cast<uintptr_t>(D) & 15) != 0) return false;
    const bool slab = slab_cols > 0;
    if (packed && slab) return false;
    if (slab && (slab_col0 % TB != 0 || slab_cols % TB != 0 || slab_col0 + slab_cols > F)) return false;
    SyrkDev g;
    g.D = D; g.ldd = ldd; g.ids = ids; g.ids2 = ids2; g.w = w; g.k_dev = k_dev; g.kmax = kmax; g.F = F;
    g.C = C; g.ldc = ldc; g.alpha = alpha; g.beta = beta; g.nt = F / TB;
    g.slab_t0 = slab ? slab_col0 / TB : 0; g.slab_nt = slab ? slab_cols / TB : 0;
    const int ntiles = slab ? g.nt * g.slab_nt : g.nt * (g.nt + 1) / 2;
    g.trace = nullptr;
    static const char *trace_path = std::getenv("DLCO_SYRK_TRACE");          // developer aid: phase time stamps of one launch
    static int trace_calls = 0;
    static unsigned long long *trace_buf = nullptr;
    const bool tracing = trace_path && !slab && !ids2 && !bf16 && ++trace_calls == 400;
    if (tracing) {
        DLCO_HIP(hipMalloc((void **)&trace_buf, (size_t)6 * 4096 * sizeof(unsigned long long)));
        g.trace = trace_buf;
    }
    {
        int n_map = 0;
        g.tile_map = syrk_tile_map(g.nt, g.slab_t0, g.slab_nt, &n_map);
        DLCO_CHECK(n_map == ntiles, -2, "syrk: tile map size");
    }
    // precision of the products: 3 (default) = three-way split bf16 on pre-split planes, fp32-level results at 3/8 of
    // the fp32 matrix time; 0 = fp32 MFMA, a k-ordered fmaf chain (DLCO_SYRK_FP32=1); 1 = operands rounded to bf16
    // once (cfg.grad_bf16, the configs[4] variant)
    static const bool exact_fp32 = std::getenv("DLCO_SYRK_FP32") != nullptr;
    const int prec = bf16 ? 1 : (exact_fp32 ? 0 : 3);
    if (prec == 3 || prec == 1) {
        // split (or round) once, then the matrix kernel (see syrk_planes_kernel)
        // the caller's workspace (a context owns one: launches of two contexts on two streams must not share planes), or
        // the per-device one for callers without (developer tools: one stream)
        char *planes = planes_ws ? static_cast<char *>(planes_ws) : syrk_planes_buffer(syrk_planes_bytes(kmax, F));
        const RankCoeffJob job = (coeff_job && prec == 3) ? *coeff_job : RankCoeffJob();
        if (prec == 3) hipLaunchKernelGGL(syrk_split_rows_kernel, dim3(g.nt + (job.frag ? 1 : 0), kmax / PL_KD), dim3(256), 0, s, D, ldd, ids, ids2, w, k_dev, kmax, g.nt, planes, job);
        else hipLaunchKernelGGL(syrk_round_rows_kernel, dim3(g.nt, (kmax + 3 * PL_KD - 1) / (3 * PL_KD)), dim3(256), 0, s, D, ldd, ids, ids2, w, k_dev, kmax, g.nt, planes);
        // three stages of 24 KB, two workgroups per CU.  (Two stages and three workgroups per CU - twelve waves - measured
        // the same: 0.133 against 0.132 ms per launch; so did eight waves of 64 x 32 per tile: 0.137 against 0.133.  What the launch loses is spread over the K loop, where two
        // workgroups sharing a CU reach 72 % of the matrix rate, the 5 us of a tile's 23 outside the loop - first
        // pieces, old tile, stores - and the last of its 4.06 rounds.)
        constexpr int NST = 3;
        ensure_dynamic_lds(reinterpret_cast<const void *>(syrk_planes_kernel<false, false, NST>), NST * PL_STAGE);
        ensure_dynamic_lds(reinterpret_cast<const void *>(syrk_planes_kernel<true, false, NST>), NST * PL_STAGE);
        ensure_dynamic_lds(reinterpret_cast<const void *>(syrk_planes_kernel<false, true, NST>), NST * PL_STAGE);
        ensure_dynamic_lds(reinterpret_cast<const void *>(syrk_planes_kernel<false, false, NST, true>), NST * PL_STAGE);
        ensure_dynamic_lds(reinterpret_cast<const void *>(syrk_planes_kernel<true, false, NST, true>), NST * PL_STAGE);
        ensure_dynamic_lds(reinterpret_cast<const void *>(syrk_planes_kernel<false, true, NST, true>), NST * PL_STAGE);
        if (prec == 1) {
            if (packed) hipLaunchKernelGGL((syrk_planes_kernel<false, true, NST, true>), dim3(ntiles), dim3(NTP), NST * PL_STAGE, s, g, planes);
            else if (slab) hipLaunchKernelGGL((syrk_planes_kernel<true, false, NST, true>), dim3(ntiles), dim3(NTP), NST * PL_STAGE, s, g, planes);
            else hipLaunchKernelGGL((syrk_planes_kernel<false, false, NST, true>), dim3(ntiles), dim3(NTP), NST * PL_STAGE, s, g, planes);
        } else if (packed) hipLaunchKernelGGL((syrk_planes_kernel<false, true, NST>), dim3(ntiles), dim3(NTP), NST * PL_STAGE, s, g, planes);
        else if (slab) hipLaunchKernelGGL((syrk_planes_kernel<true, false, NST>), dim3(ntiles), dim3(NTP), NST * PL_STAGE, s, g, planes);
        else hipLaunchKernelGGL((syrk_planes_kernel<false, false, NST>), dim3(ntiles), dim3(NTP), NST * PL_STAGE, s, g, planes);
        DLCO_HIP(hipGetLastError());
        if (tracing) syrk_dump_trace(trace_path, trace_buf, ntiles, s);
        return true;
    }
    // prec == 0: the gathering kernel with v_mfma_f32_32x32x2_f32
#define DLCO_SYRK_LAUNCH(P, S) hipLaunchKernelGGL((syrk_rda_kernel8<P, S, 0>), dim3(ntiles), dim3(NT8), 0, s, g)
#define DLCO_SYRK_LAUNCH_PK(P) hipLaunchKernelGGL((syrk_rda_kernel8<P, false, 0, true>), dim3(ntiles), dim3(NT8), 0, s, g)
    if (packed) { if (ids2) DLCO_SYRK_LAUNCH_PK(true); else DLCO_SYRK_LAUNCH_PK(false); }
    else if (slab) { if (ids2) DLCO_SYRK_LAUNCH(true, true); else DLCO_SYRK_LAUNCH(false, true); }
    else { if (ids2) DLCO_SYRK_LAUNCH(true, false); else DLCO_SYRK_LAUNCH(false, false); }
#undef DLCO_SYRK_LAUNCH_PK
#undef DLCO_SYRK_LAUNCH
    DLCO_HIP(hipGetLastError());
    if (tracing) syrk_dump_trace(trace_path, trace_buf, ntiles, s);
    return true;
}

}  // namespace dlco
