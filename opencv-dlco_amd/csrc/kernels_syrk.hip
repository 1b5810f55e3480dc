// kernels_syrk.hip — the dominant kernel of the pj-learn step on gfx950:
//
//     dfAvg  <-  beta * dfAvg  +  alpha * sum_k  w_k * x_k x_k^T          (Q1 + U1)
//
// x_k = row ids[k] of the resident PR-descriptor-difference matrix D [N,F], w_k = rho_i for
// a positive row, -kappa_j for a negative one (src/pj-learn.cpp:367-422 in the reformulated
// order P^T diag(rho) P - N^T diag(kappa) N; rows with zero weight are not in the list).
//
// Design (MI355X): the output is symmetric, so only the 128x128 tiles on or above the
// diagonal are computed (nt(nt+1)/2 workgroups, dealt to the 8 XCDs in contiguous chunks so
// that neighbouring tiles share their row panels in one L2) and each tile is also stored
// transposed.  Both MFMA operands are k-major images of gathered rows (lane l reads element
// l&31 of row 2kk + (l>>5): conflict-free ds_read_b32), staged through registers with the
// row ids one tile ahead of the data so that the gather's two dependent loads never sit in
// the same iteration.  The hot loop is branch-free; the row list is zero-padded to the tile
// depth by the kernel that builds it.  Epilogue: dual-average update in registers, direct
// store of the tile, transposed store through LDS as whole 512-byte rows.  Exact fp32
// (v_mfma_f32_32x32x2_f32 = k-ordered fmaf chain).
#include "dlco_internal.hpp"

#include <mutex>
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
constexpr int NTH = 256;

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
    int stagger_from, stagger_units;   // workgroups [from, 2*from) sleep units * nk * 4096 cycles at start
    const int32_t *tile_map;      // tile number -> (bi << 16 | bj), see syrk_tile_map()
    unsigned long long *trace;    // developer aid (DLCO_SYRK_TRACE): 6 words per workgroup, see syrk_rda_f32
    int old_early, noprio;        // experiment switches of syrk_rda_kernel (DLCO_SYRK_OLD_EARLY, DLCO_SYRK_PRIO)
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
template <bool PAIR, bool SLAB, bool BF16>
__global__ __launch_bounds__(NTH, 2) void syrk_rda_kernel(SyrkDev g)
{
    __shared__ __attribute__((aligned(16))) SyrkLds lds;
    const unsigned long long tr0 = g.trace ? wall_clock64() : 0ull;

    // ---- tile assignment: XCD-contiguous chunks over the tile list --------------------------------
    const int ntiles = SLAB ? g.nt * g.slab_nt : g.nt * (g.nt + 1) / 2;
    const int nxcd = 8;
    const int bid = blockIdx.x;
    int t;
    {
        const int q = ntiles / nxcd, r = ntiles % nxcd, xcd = bid % nxcd, within = bid / nxcd;
        t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + within;     // bijective remap
    }
    // the tile list walks 8 x 8 super-blocks of tiles (syrk_tile_map): the ~64 tiles an XCD runs at a time then
    // need 8 + 8 row panels of the gathered rows (2.5 MB at K = 305) instead of 1 + 64 (10 MB > its 4 MB L2)
    const int code = g.tile_map[t];
    const int bi = code >> 16, bj = code & 0xffff;
    const int i0 = bi * TB, j0 = bj * TB;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int lr = lane & 31, lk = lane >> 5;

    const int kact = min(*g.k_dev, g.kmax);
    const int nk = (kact + KB - 1) / KB;          // the list is zero padded up to a multiple of KB

    // Phase stagger.  A tile is an MFMA phase (K loop) followed by an HBM phase (512 KiB of dfAvg
    // stored, 64 KiB prefetched for the next).  All tiles are alike, so the two workgroups that
    // share a CU would run their K loops together (each at half the MFMA rate) and then store
    // together (HBM idle meanwhile, MFMA idle afterwards).  The second workgroup of every CU in
    // the first round starts about a quarter of a K loop late; the offset then persists from tile
    // to tile and one workgroup computes while the other streams (measured: -6 % launch time at
    // K = 224, neutral at K >= 1700; longer delays lose more at the start than they win).
    if (g.stagger_units > 0 && bid >= g.stagger_from && bid < 2 * g.stagger_from) {
        for (int q = 0; q < g.stagger_units * nk; q++) __builtin_amdgcn_s_sleep(64);
    }

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[a][b][r] = 0.f;

    // loader mapping: 4 float4 per operand per thread; f = tid + 256u -> row = f / 32, c4 = f % 32
    const int c4 = tid & 31, rbase = tid >> 5;    // rows rbase, rbase + 8, rbase + 16, rbase + 24
    int32_t id_nx[4], id2_nx[4];
    float w_nx[4];
    f32x4 ra[4], rb[4];

    auto load_ids = [&](int kt) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int k = kt * KB + rbase + 8 * u;
            id_nx[u] = g.ids[k];
            if (PAIR) id2_nx[u] = g.ids2[k];
            w_nx[u] = g.w[k];
        }
    };
    auto load_rows = [&]() {
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const float *row = g.D + (long)id_nx[u] * g.ldd;
            f32x4 xa = *reinterpret_cast<const f32x4 *>(row + i0 + c4 * 4);
            f32x4 xb = *reinterpret_cast<const f32x4 *>(row + j0 + c4 * 4);
            if (PAIR) {                                       // descriptor difference formed on the fly
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
        for (int u = 0; u < 4; u++) {
            *reinterpret_cast<f32x4 *>(&lds.st.A[buf][rbase + 8 * u][c4 * 4]) = ra[u];
            *reinterpret_cast<f32x4 *>(&lds.st.B[buf][rbase + 8 * u][c4 * 4]) = rb[u];
        }
    };

    if (nk > 0) load_ids(0);
    // The tile's previous contents (dual average term) are needed in the epilogue only.  A wave has one in-order
    // counter for its loads, so requested up front they would have to arrive before the first K block's gather
    // can be waited for (measured with DLCO_SYRK_TRACE: 15 us from the start of a workgroup to its first MFMA, during
    // which the CU's other workgroup runs its K loop alone - and a single wave per SIMD issues MFMAs at half the
    // pipe's rate).  They are requested inside the last K block instead: no gather follows them, and they land
    // under that block's MFMAs.  (g.old_early restores the up-front request.)
    float oldv[2][2][16];
    const bool use_old = (g.beta != 0.f);
    // (scalar row base + one per-lane offset: the 64 loads then need no address VGPRs, which matters here because
    // they are issued where accumulators, staging registers and fragments are all live)
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int old_lane = 4 * lk * (int)g.ldc + lr;
    auto fetch_old = [&]() {
#pragma unroll
        for (int a = 0; a < 2; a++)
#pragma unroll
            for (int b = 0; b < 2; b++)
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const float *rowb = g.C + (long)(i0 + (wave_u >> 1) * 64 + a * 32 + (r & 3) + 8 * (r >> 2)) * g.ldc + (j0 + (wave_u & 1) * 64 + b * 32);
                    oldv[a][b][r] = use_old ? rowb[old_lane] : 0.f;
                }
    };
    constexpr bool LATE = !BF16;      // the bf16 K loop is an eighth as long and its fragments need the registers
    if (!LATE || g.old_early || nk == 0) fetch_old();
    if (nk > 0) {
        load_rows();
        if (nk > 1) load_ids(1);
        store_rows(0);
    }
    __syncthreads();
    const unsigned long long tr1 = g.trace ? wall_clock64() : 0ull;
    for (int kt = 0; kt < nk; kt++) {
        const int buf = kt & 1;
        if (kt + 1 < nk) load_rows();                         // data of tile kt+1 (ids already here)
        if (kt + 2 < nk) load_ids(kt + 2);                    // ids of tile kt+2
        if (LATE && kt == nk - 1 && !g.old_early) fetch_old();
        if (!g.noprio) __builtin_amdgcn_s_setprio(1);
        if (BF16) {
            // fragment of the 32x32x16 bf16 MFMA: lane (lr, lk) holds k = 8 lk .. 8 lk + 7 of row lr; the
            // k-major fp32 image is read with the same conflict-free 4-byte accesses as below
#pragma unroll
            for (int ks = 0; ks < KB / 16; ks++) {
                bf16x8 a0, a1, b0, b1;
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const int k = 16 * ks + 8 * lk + j;
                    a0[j] = (__bf16)lds.st.A[buf][k][wm * 64 + lr];
                    a1[j] = (__bf16)lds.st.A[buf][k][wm * 64 + 32 + lr];
                    b0[j] = (__bf16)lds.st.B[buf][k][wn * 64 + lr];
                    b1[j] = (__bf16)lds.st.B[buf][k][wn * 64 + 32 + lr];
                }
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[1][1], 0, 0, 0);
            }
        } else
#pragma unroll
        for (int kk = 0; kk < KB / 2; kk++) {
            const float a0 = lds.st.A[buf][2 * kk + lk][wm * 64 + lr];
            const float a1 = lds.st.A[buf][2 * kk + lk][wm * 64 + 32 + lr];
            const float b0 = lds.st.B[buf][2 * kk + lk][wn * 64 + lr];
            const float b1 = lds.st.B[buf][2 * kk + lk][wn * 64 + 32 + lr];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
        __builtin_amdgcn_s_setprio(0);
        if (kt + 1 < nk) store_rows(buf ^ 1);
        __syncthreads();
    }

    const unsigned long long tr2 = g.trace ? wall_clock64() : 0ull;
    // ---- epilogue: out = beta*old + alpha*acc on the upper tile; mirror below the diagonal -------
    // All stores are 16 bytes per lane.  The transposed copy goes straight from the accumulator
    // registers: a lane holds 4 consecutive rows of one column (regs 4g..4g+3), which are 4
    // consecutive columns of the transposed row.  The tile itself is re-laid through LDS so
    // that 32 lanes write one whole 512-byte row.
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
                for (int e = 0; e < 4; e++) {
                    const int r = 4 * q + e;
                    o[e] = g.alpha * acc[a][b][r] + g.beta * oldv[a][b][r];
                }
                const int il0 = wm * 64 + a * 32 + 8 * q + 4 * lk;       // rows il0 .. il0+3
                if (!SLAB && !diag) *reinterpret_cast<f32x4 *>(&g.C[(long)(j0 + jl) * g.ldc + (i0 + il0)]) = o;
#pragma unroll
                for (int e = 0; e < 4; e++) lds.T[il0 + e][jl] = o[e];
            }
        }
    __syncthreads();
    for (int f = tid; f < TB * (TB / 4); f += NTH) {
        const int il = f / (TB / 4), c4 = (f % (TB / 4)) * 4;
        f32x4 v = *reinterpret_cast<const f32x4 *>(&lds.T[il][c4]);
        if (diag) {
            // exact symmetry on diagonal tiles: the lower triangle takes the upper triangle's values
#pragma unroll
            for (int e = 0; e < 4; e++)
                if (c4 + e < il) v[e] = lds.T[c4 + e][il];
        }
        *reinterpret_cast<f32x4 *>(&g.C[(long)(i0 + il) * g.ldc + (j0 + c4)]) = v;
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
// Eight-wave form (default).  A single wave per SIMD issues v_mfma_f32_32x32x2_f32 at HALF the pipe's
// rate (tools/overlap_probe.cpp: 4 workgroup-waves per CU need 292 us for what 8 do in 155 us), so with
// four-wave workgroups the matrix cores only run at full rate while BOTH workgroups of a CU are inside
// their K loops; whenever one of them gathers its first block or writes its tile out, the other computes
// at half speed (DLCO_SYRK_TRACE time stamps: 37 us K loops against 17 us of MFMA work).  Here a
// workgroup has eight waves - two per SIMD, each a 64 x 32 part of the tile - and saturates the pipe on
// its own; two such workgroups per CU then alternate freely between K loop and memory phases.
// Same tile order, K order, arithmetic and stores as syrk_rda_kernel.
// ---------------------------------------------------------------------------------------------------
constexpr int NT8 = 512;

template <bool PAIR, bool SLAB, bool BF16, int UNR = 16>
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

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3;                  // rows wm*64.., columns wn*32.. of the tile
    const int lr = lane & 31, lk = lane >> 5;

    const int kact = min(*g.k_dev, g.kmax);
    const int nk = (kact + KB - 1) / KB;

    f32x16 acc[2];
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[a][r] = 0.f;

    // loader mapping: 2 float4 per operand per thread; rows rbase, rbase + 16 of the K block
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
    // old tile: requested inside the last K block (see syrk_rda_kernel), scalar row base + one lane offset
    float oldv[2][16];
    const bool use_old = (g.beta != 0.f);
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int old_lane = 4 * lk * (int)g.ldc + lr;
    auto fetch_old = [&]() {
#pragma unroll
        for (int a = 0; a < 2; a++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const float *rowb = g.C + (long)(i0 + (wave_u >> 2) * 64 + a * 32 + (r & 3) + 8 * (r >> 2)) * g.ldc + (j0 + (wave_u & 3) * 32);
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
        if (BF16) {
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
    // ---- epilogue: as syrk_rda_kernel -----------------------------------------------------------
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
            if (!SLAB && !diag) *reinterpret_cast<f32x4 *>(&g.C[(long)(j0 + jl) * g.ldc + (i0 + il0)]) = o;
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
        *reinterpret_cast<f32x4 *>(&g.C[(long)(i0 + il) * g.ldc + (j0 + cc)]) = v;
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
// Lean form: the same tile-per-workgroup scheme cut down to FOUR workgroups per CU (16-deep K blocks:
// 34 KB of LDS; no prefetched old tile and no LDS re-layout of the output: <= 128 VGPRs), so that the
// MFMA phase of some workgroups always has the HBM phase of others beside it on the same CU.  Same
// arithmetic and K order; the epilogue stores straight from the accumulator layout (a lane's register r
// is row (r&3) + 8(r>>2) + 4(lane>>5), column lane&31: 32 lanes = 128 contiguous bytes of a row; four
// consecutive registers = 16 contiguous bytes of the mirrored row).
// ---------------------------------------------------------------------------------------------------
constexpr int KL = 16;            // K depth per LDS stage of the lean kernel

template <bool PAIR, bool SLAB, bool BF16>
__global__ __launch_bounds__(NTH, 4) void syrk_rda_lean_kernel(SyrkDev g)
{
    __shared__ __attribute__((aligned(16))) float sA[2][KL][LD];
    __shared__ __attribute__((aligned(16))) float sB[2][KL][LD];

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

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int lr = lane & 31, lk = lane >> 5;

    const int kact = min(*g.k_dev, g.kmax);
    const int nk = (kact + KL - 1) / KL;          // the list is zero padded up to a multiple of 32

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[a][b][r] = 0.f;

    // loader mapping: 2 float4 per operand per thread; rows rbase, rbase + 8 of the K block
    const int c4 = tid & 31, rbase = tid >> 5;
    int32_t id_nx[2], id2_nx[2];
    float w_nx[2];
    f32x4 ra[2], rb[2];
    auto load_ids = [&](int kt) {
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const int k = kt * KL + rbase + 8 * u;
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
            *reinterpret_cast<f32x4 *>(&sA[buf][rbase + 8 * u][c4 * 4]) = ra[u];
            *reinterpret_cast<f32x4 *>(&sB[buf][rbase + 8 * u][c4 * 4]) = rb[u];
        }
    };

    if (nk > 0) {
        load_ids(0);
        load_rows();
        if (nk > 1) load_ids(1);
        store_rows(0);
    }
    __syncthreads();
    for (int kt = 0; kt < nk; kt++) {
        const int buf = kt & 1;
        if (kt + 1 < nk) load_rows();
        if (kt + 2 < nk) load_ids(kt + 2);
        __builtin_amdgcn_s_setprio(1);
        if (BF16) {
            bf16x8 a0, a1, b0, b1;
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const int k = 8 * lk + j;
                a0[j] = (__bf16)sA[buf][k][wm * 64 + lr];
                a1[j] = (__bf16)sA[buf][k][wm * 64 + 32 + lr];
                b0[j] = (__bf16)sB[buf][k][wn * 64 + lr];
                b1[j] = (__bf16)sB[buf][k][wn * 64 + 32 + lr];
            }
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[1][1], 0, 0, 0);
        } else {
#pragma unroll
            for (int kk = 0; kk < KL / 2; kk++) {
                const float a0 = sA[buf][2 * kk + lk][wm * 64 + lr];
                const float a1 = sA[buf][2 * kk + lk][wm * 64 + 32 + lr];
                const float b0 = sB[buf][2 * kk + lk][wn * 64 + lr];
                const float b1 = sB[buf][2 * kk + lk][wn * 64 + 32 + lr];
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
            }
        }
        __builtin_amdgcn_s_setprio(0);
        if (kt + 1 < nk) store_rows(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue: out = beta*old + alpha*acc from registers; the other workgroups of the CU compute meanwhile ----
    const bool diag = !SLAB && (bi == bj), use_old = (g.beta != 0.f);
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++) {
            const int jl = wn * 64 + b * 32 + lr;
            float old[16];
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int il = wm * 64 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
                old[r] = use_old ? g.C[(long)(i0 + il) * g.ldc + (j0 + jl)] : 0.f;
            }
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int il0 = wm * 64 + a * 32 + 8 * q + 4 * lk;       // rows il0 .. il0+3
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; e++) o[e] = g.alpha * acc[a][b][4 * q + e] + g.beta * old[4 * q + e];
#pragma unroll
                for (int e = 0; e < 4; e++)
                    if (!diag || jl >= il0 + e) g.C[(long)(i0 + il0 + e) * g.ldc + (j0 + jl)] = o[e];
                if (!SLAB) {
                    float *mp = &g.C[(long)(j0 + jl) * g.ldc + (i0 + il0)];
                    if (!diag) *reinterpret_cast<f32x4 *>(mp) = o;
                    else {
#pragma unroll
                        for (int e = 0; e < 4; e++)
                            if (il0 + e < jl) mp[e] = o[e];                // the lower triangle mirrors the upper one
                    }
                }
            }
        }
}

// ---------------------------------------------------------------------------------------------------
// Three-workgroup form: the tile-per-workgroup kernel with 16-deep K blocks (34 KB of LDS) and the output
// re-layout done in two halves of 64 rows through the same 34 KB, so that THREE workgroups share a CU
// (<= 168 VGPRs): more K loops beside more epilogues at any time.  Same arithmetic, same stores.
// ---------------------------------------------------------------------------------------------------
union SyrkLds3 {
    struct { float A[2][KL][LD]; float B[2][KL][LD]; } st;    // 33,792 B
    float T[TB / 2][TLD];                                      // 33,792 B
};

template <bool PAIR, bool SLAB, bool BF16>
__global__ __launch_bounds__(NTH, 3) void syrk_rda_kernel3(SyrkDev g)
{
    __shared__ __attribute__((aligned(16))) SyrkLds3 lds;

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

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int lr = lane & 31, lk = lane >> 5;

    const int kact = min(*g.k_dev, g.kmax);
    const int nk = (kact + KL - 1) / KL;

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[a][b][r] = 0.f;

    const int c4 = tid & 31, rbase = tid >> 5;
    int32_t id_nx[2], id2_nx[2];
    float w_nx[2];
    f32x4 ra[2], rb[2];
    auto load_ids = [&](int kt) {
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const int k = kt * KL + rbase + 8 * u;
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
            *reinterpret_cast<f32x4 *>(&lds.st.A[buf][rbase + 8 * u][c4 * 4]) = ra[u];
            *reinterpret_cast<f32x4 *>(&lds.st.B[buf][rbase + 8 * u][c4 * 4]) = rb[u];
        }
    };

    if (nk > 0) load_ids(0);
    // The tile's previous contents are fetched in the epilogue, one 32 x 32 quadrant (16 values per lane) at a time:
    // with three workgroups per CU their latency is covered by the K loops of the other two, and 64 registers are saved.
    const bool use_old = (g.beta != 0.f);
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int old_lane = 4 * lk * (int)g.ldc + lr;
    auto fetch_old = [&](int a, int b, float (&o16)[16]) {
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const float *rowb = g.C + (long)(i0 + (wave_u >> 1) * 64 + a * 32 + (r & 3) + 8 * (r >> 2)) * g.ldc + (j0 + (wave_u & 1) * 64 + b * 32);
            o16[r] = use_old ? rowb[old_lane] : 0.f;
        }
    };
    if (nk > 0) {
        load_rows();
        if (nk > 1) load_ids(1);
        store_rows(0);
    }
    __syncthreads();
    for (int kt = 0; kt < nk; kt++) {
        const int buf = kt & 1;
        if (kt + 1 < nk) load_rows();
        if (kt + 2 < nk) load_ids(kt + 2);
        if (!g.noprio) __builtin_amdgcn_s_setprio(1);
        if (BF16) {
            bf16x8 a0, a1, b0, b1;
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const int k = 8 * lk + j;
                a0[j] = (__bf16)lds.st.A[buf][k][wm * 64 + lr];
                a1[j] = (__bf16)lds.st.A[buf][k][wm * 64 + 32 + lr];
                b0[j] = (__bf16)lds.st.B[buf][k][wn * 64 + lr];
                b1[j] = (__bf16)lds.st.B[buf][k][wn * 64 + 32 + lr];
            }
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[1][1], 0, 0, 0);
        } else {
#pragma unroll
            for (int kk = 0; kk < KL / 2; kk++) {
                const float a0 = lds.st.A[buf][2 * kk + lk][wm * 64 + lr];
                const float a1 = lds.st.A[buf][2 * kk + lk][wm * 64 + 32 + lr];
                const float b0 = lds.st.B[buf][2 * kk + lk][wn * 64 + lr];
                const float b1 = lds.st.B[buf][2 * kk + lk][wn * 64 + 32 + lr];
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
            }
        }
        __builtin_amdgcn_s_setprio(0);
        if (kt + 1 < nk) store_rows(buf ^ 1);
        __syncthreads();
    }

    const bool diag = !SLAB && (bi == bj);
    if (diag) {
        // diagonal tiles (64 of 2080): element-wise, straight from registers - the upper triangle as computed,
        // the strictly lower one as its mirror
#pragma unroll
        for (int a = 0; a < 2; a++)
#pragma unroll
            for (int b = 0; b < 2; b++) {
                const int jl = wn * 64 + b * 32 + lr;
                float o16[16];
                fetch_old(a, b, o16);
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const int il = wm * 64 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
                    const float o = g.alpha * acc[a][b][r] + g.beta * o16[r];
                    if (jl >= il) g.C[(long)(i0 + il) * g.ldc + (j0 + jl)] = o;
                    if (il < jl) g.C[(long)(j0 + jl) * g.ldc + (i0 + il)] = o;
                }
            }
        return;
    }
    // two halves of 64 rows: the waves of that half lay their quadrants out in T (and store the mirrored 16-byte
    // pieces directly), then all threads store whole 512-byte rows
    for (int hlf = 0; hlf < 2; hlf++) {
        if (wm == hlf) {
#pragma unroll
            for (int a = 0; a < 2; a++)
#pragma unroll
                for (int b = 0; b < 2; b++) {
                    const int jl = wn * 64 + b * 32 + lr;
                    float o16[16];
                    fetch_old(a, b, o16);
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        f32x4 o;
#pragma unroll
                        for (int e = 0; e < 4; e++) o[e] = g.alpha * acc[a][b][4 * q + e] + g.beta * o16[4 * q + e];
                        const int il0 = a * 32 + 8 * q + 4 * lk;             // row inside the half
                        if (!SLAB) *reinterpret_cast<f32x4 *>(&g.C[(long)(j0 + jl) * g.ldc + (i0 + hlf * 64 + il0)]) = o;
#pragma unroll
                        for (int e = 0; e < 4; e++) lds.T[il0 + e][jl] = o[e];
                    }
                }
        }
        __syncthreads();
        for (int f = tid; f < (TB / 2) * (TB / 4); f += NTH) {
            const int il = f / (TB / 4), cc = (f % (TB / 4)) * 4;
            *reinterpret_cast<f32x4 *>(&g.C[(long)(i0 + hlf * 64 + il) * g.ldc + (j0 + cc)]) = *reinterpret_cast<const f32x4 *>(&lds.T[il][cc]);
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------
// Streaming form (default for row mode).  In the kernel above a tile is an MFMA phase followed by an HBM
// phase, and because every tile is alike the whole chip runs the two phases in lock-step: the matrix
// cores idle while 512 KiB of dfAvg per tile move, HBM idles during the K loops (measured: launch time
// = MFMA time + HBM time, MFMA busy 46 %).  Here ONE persistent workgroup per CU walks its list of tiles
// and its twelve waves are specialised:
//   waves 0-7   (compute)  MFMA from LDS only, each a 64 x 32 part of the tile; at the end of a K loop
//                          they leave the raw accumulators in an LDS image T of the tile
//   waves 8, 9  (gather)   the K blocks of both operands into the two staging images, ALTERNATING: a wave
//                          requests a whole block (32 rows x 2 x 512 B) two K blocks before it is needed
//                          and parks it in registers meanwhile - across tile boundaries too
//   waves 10, 11 (update)  the dual-average update of the PREVIOUS tile from T, spread over the K loop of
//                          the current one in 8 batches of 16 rows, ALTERNATING: old dfAvg values are
//                          requested two batches ahead; out = alpha*T + beta*old; rows and mirrored pieces
// A wave has ONE in-order counter for its memory operations, so a wave that both requests and waits in
// every K block can never have more than one block's worth of latency in flight; alternating waves give
// every request two K blocks (~3.4 us) to complete, and no wave mixes the gather (on the MFMAs' critical
// path) with the dfAvg round trips.  The row-id and weight lists sit in LDS (no dependent global load in
// the gather).  Arithmetic, K order and the mirror rule on diagonal tiles are those of the kernel above.
// ---------------------------------------------------------------------------------------------------
constexpr int ST = 768;
constexpr int SCW = 8;            // compute waves
constexpr int SK_MAX = 2048;      // capacity of the row list held in LDS

struct SyrkLds2 {
    float A[2][KB][LD];
    float B[2][KB][LD];
    float T[TB][TLD];
    int32_t ids[SK_MAX];
    float w[SK_MAX];
};                                 // 151,552 B: one workgroup per CU

struct TileRef { int i0, j0; bool diag; };

// Dual-average update of the tile held in T, in row groups of 4 rows (32 per tile).
// role 0: whole rows (s = 0..127: row s>>5 of the group, 16-byte column piece s&31); role 1: the
// mirrored copy (s = column jl; the group's four rows are 16 contiguous bytes of the mirrored row).
__device__ __forceinline__ const float *syrk_piece(const SyrkDev &g, TileRef tr, int role, int s, int gq)
{
    // The mirrored piece takes its old values from the mirrored position - the very 16 bytes this thread
    // overwrites, so no other wave's store can get in between - which holds the same numbers: dfAvg is
    // exactly symmetric (this kernel keeps it so; an uploaded one is symmetrised by syrk_mirror_upper).
    return role == 0 ? &g.C[(long)(tr.i0 + 4 * gq + (s >> 5)) * g.ldc + tr.j0 + 4 * (s & 31)]
                     : &g.C[(long)(tr.j0 + s) * g.ldc + tr.i0 + 4 * gq];
}

__device__ __forceinline__ f32x4 syrk_value(const SyrkDev &g, const float (*T)[TLD], int role, int s, int gq, f32x4 old)
{
    f32x4 t4;
    if (role == 0) t4 = *reinterpret_cast<const f32x4 *>(&T[4 * gq + (s >> 5)][4 * (s & 31)]);
    else {
#pragma unroll
        for (int e = 0; e < 4; e++) t4[e] = T[4 * gq + e][s];
    }
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; e++) o[e] = fmaf(g.alpha, t4[e], g.beta * old[e]);
    return o;
}

// groups g0, g0+gs, .. < g1 with loads and stores back to back: diagonal tiles (whose lower triangle mirrors
// the upper one element by element), short K loops, and the last tile of a workgroup
template <bool SLAB>
__device__ __forceinline__ void syrk_drain(const SyrkDev &g, const float (*T)[TLD], TileRef tr, int role, int s, int g0, int g1, int gs,
                                           bool use_old)
{
    if (SLAB && role != 0) return;
    for (int gb = g0; gb < g1; gb += 2 * gs) {
        f32x4 old[2];
#pragma unroll
        for (int u = 0; u < 2; u++) {
            old[u] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (gb + u * gs < g1 && use_old) old[u] = *reinterpret_cast<const f32x4 *>(syrk_piece(g, tr, role, s, gb + u * gs));
        }
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const int gq = gb + u * gs;
            if (gq >= g1) continue;
            const f32x4 o = syrk_value(g, T, role, s, gq, old[u]);
            float *dst = const_cast<float *>(syrk_piece(g, tr, role, s, gq));
            if (SLAB || !tr.diag) *reinterpret_cast<f32x4 *>(dst) = o;
            else {
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    // role 0 keeps the upper triangle (column >= row), role 1 writes the strictly lower one
                    const bool keep = role == 0 ? (4 * (s & 31) + e >= 4 * gq + (s >> 5)) : (4 * gq + e < s);
                    if (keep) dst[e] = o[e];
                }
            }
        }
    }
}

// tile number -> (bi, bj) of the enumeration shared with syrk_rda_kernel
template <bool SLAB>
__device__ __forceinline__ void syrk_tile(const SyrkDev &g, int t, int &bi, int &bj)
{
    const int code = g.tile_map[t];
    bi = code >> 16;
    bj = code & 0xffff;
}

// The roles are separate loops over the same tile list that meet at the same barriers: their registers
// (accumulators; a parked K block; old dfAvg values) are then allocated as a union, not a sum.
// Barrier protocol per tile: one at the start (T of the previous tile and the first staging image
// complete), one per K block, one more when there is no K block at all, and one before the all-wave
// write-out of a diagonal tile; one before the final flush.
template <bool SLAB, bool BF16>
__global__ __launch_bounds__(ST) void syrk_rda_stream_kernel(SyrkDev g)
{
    __shared__ __attribute__((aligned(16))) SyrkLds2 lds;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    // tile list: the XCD's contiguous chunk of the tile enumeration, dealt round-robin to its workgroups
    const int ntiles = SLAB ? g.nt * g.slab_nt : g.nt * (g.nt + 1) / 2;
    const int nxcd = 8, per = gridDim.x / nxcd;
    const int xcd = blockIdx.x % nxcd, within = blockIdx.x / nxcd;
    const int q = ntiles / nxcd, r = ntiles % nxcd;
    const int start = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q, len = q + (xcd < r ? 1 : 0);
    const int my_tiles = len > within ? (len - within + per - 1) / per : 0;

    const int kact = min(*g.k_dev, g.kmax);
    const int nk = (kact + KB - 1) / KB;
    const bool use_old = (g.beta != 0.f);
    // the update of a tile is spread over the next tile's K loop in 8 batches of 4 row groups, starting with
    // the SECOND K block (the first batch's old values are requested at the tile boundary and get the first
    // block to arrive): one batch per K block when there are at least 9 of them (K > 256), else several,
    // and whatever is left in the last block
    const int bpi = nk > 1 ? (8 + nk - 2) / (nk - 1) : 8;
    const int fl_pair = wave >> 1, fl_s = tid & 127;           // all-wave write-outs: 6 pairs of waves
    bool have_prev = false;
    TileRef prev{0, 0, false};

    for (int k = tid; k < nk * KB; k += ST) {                  // the list is zero padded to a multiple of KB
        lds.ids[k] = g.ids[k];
        lds.w[k] = g.w[k];
    }
    __syncthreads();

    if (wave >= SCW + 2) {
        // ================================ update waves (10, 11) ========================
        const int uw = wave - (SCW + 2);
        constexpr int NR = SLAB ? 1 : 2;                       // roles: rows, and mirrored pieces of the symmetric matrix
        f32x4 old[NR][8];
#pragma unroll
        for (int ro = 0; ro < NR; ro++)
#pragma unroll
            for (int v = 0; v < 8; v++) old[ro][v] = f32x4{0.f, 0.f, 0.f, 0.f};
        // item v of a batch: row group 4*batch + (it >> 7), position it & 127, it = lane + 64 v
        auto fetch = [&](int batch) {
#pragma unroll
            for (int ro = 0; ro < NR; ro++)
#pragma unroll
                for (int v = 0; v < 8; v++) {
                    const int it = lane + 64 * v;
                    old[ro][v] = *reinterpret_cast<const f32x4 *>(syrk_piece(g, prev, ro, it & 127, 4 * batch + (it >> 7)));
                }
        };
        auto apply = [&](int batch) {
#pragma unroll
            for (int ro = 0; ro < NR; ro++)
#pragma unroll
                for (int v = 0; v < 8; v++) {
                    const int it = lane + 64 * v;
                    old[ro][v] = syrk_value(g, lds.T, ro, it & 127, 4 * batch + (it >> 7), old[ro][v]);
                }
#pragma unroll
            for (int ro = 0; ro < NR; ro++)
#pragma unroll
                for (int v = 0; v < 8; v++) {
                    const int it = lane + 64 * v;
                    *reinterpret_cast<f32x4 *>(const_cast<float *>(syrk_piece(g, prev, ro, it & 127, 4 * batch + (it >> 7)))) = old[ro][v];
                }
        };
        for (int ts = 0; ts < my_tiles; ts++) {
            int bi, bj;
            syrk_tile<SLAB>(g, start + within + ts * per, bi, bj);
            int nb = uw;                                       // this wave's next batch of the previous tile
            if (have_prev && use_old) fetch(nb);
            __syncthreads();
            for (int kt = 0; kt < nk; kt++) {
                if (have_prev) {
                    const int lim = kt == nk - 1 ? 8 : min(kt * bpi, 8);
                    while (nb < lim) {
                        apply(nb);
                        nb += 2;
                        if (nb < 8 && use_old) fetch(nb);
                    }
                }
                __syncthreads();
            }
            if (nk == 0) {                                     // no active row: the tiles only decay (out = beta * old)
                if (have_prev)
                    while (nb < 8) {
                        apply(nb);
                        nb += 2;
                        if (nb < 8 && use_old) fetch(nb);
                    }
                __syncthreads();
            }
            prev = TileRef{bi * TB, bj * TB, bi == bj};
            have_prev = true;
            if (!SLAB && bi == bj) {                           // see the compute side
                __syncthreads();
                syrk_drain<SLAB>(g, lds.T, prev, fl_pair & 1, fl_s, fl_pair >> 1, 32, 3, use_old);
                have_prev = false;
            }
        }
    } else if (wave >= SCW) {
        // ================================ gather waves (8, 9) ==========================
        // Entries e = (tile sequence number) * nk + K block, over all tiles of this workgroup; entry e is
        // multiplied from staging image e & 1 and belongs to gather wave e & 1, which deposits it during
        // entry e-1 and requests entry e+2 right after.
        const int gw = wave - SCW;
        const int c4 = lane & 31, rh = lane >> 5;              // rows rh + 2u of a K block, 16-byte piece c4
        const int n_ent = my_tiles * nk;
        f32x4 ga[16], gb[16];
        auto request = [&](int e) {
            const int ts = e / nk, blk = e - ts * nk;
            int bi, bj;
            syrk_tile<SLAB>(g, start + within + ts * per, bi, bj);
            const int i0 = bi * TB, j0 = bj * TB;
#pragma unroll
            for (int u = 0; u < 16; u++) {
                const float *row = g.D + (long)lds.ids[blk * KB + rh + 2 * u] * g.ldd + c4 * 4;
                ga[u] = *reinterpret_cast<const f32x4 *>(row + i0);
                gb[u] = *reinterpret_cast<const f32x4 *>(row + j0);
            }
        };
        auto deposit = [&](int e) {
            const int blk = e % nk, buf = e & 1;
#pragma unroll
            for (int u = 0; u < 16; u++) {
                const float wk = lds.w[blk * KB + rh + 2 * u];
                *reinterpret_cast<f32x4 *>(&lds.A[buf][rh + 2 * u][c4 * 4]) = ga[u] * wk;
                *reinterpret_cast<f32x4 *>(&lds.B[buf][rh + 2 * u][c4 * 4]) = gb[u];
            }
        };
        if (gw < n_ent) {
            request(gw);
            if (gw == 0) {
                deposit(0);
                if (2 < n_ent) request(2);
            }
        }
        for (int ts = 0; ts < my_tiles; ts++) {
            int bi, bj;
            syrk_tile<SLAB>(g, start + within + ts * per, bi, bj);
            __syncthreads();
            for (int kt = 0; kt < nk; kt++) {
                const int en = ts * nk + kt + 1;               // the entry multiplied after this one
                if ((en & 1) == gw && en < n_ent) {
                    deposit(en);
                    if (en + 2 < n_ent) request(en + 2);
                }
                __syncthreads();
            }
            if (nk == 0) __syncthreads();
            prev = TileRef{bi * TB, bj * TB, bi == bj};
            have_prev = true;
            if (!SLAB && bi == bj) {
                __syncthreads();
                syrk_drain<SLAB>(g, lds.T, prev, fl_pair & 1, fl_s, fl_pair >> 1, 32, 3, use_old);
                have_prev = false;
            }
        }
    } else {
        // ================================ compute waves ===============================
        const int lr = lane & 31, lk = lane >> 5;
        const int wm = (wave >> 2) & 1, wn = wave & 3;         // rows wm*64.., columns wn*32.. of the tile
        for (int ts = 0; ts < my_tiles; ts++) {
            int bi, bj;
            syrk_tile<SLAB>(g, start + within + ts * per, bi, bj);
            f32x16 acc[2];
#pragma unroll
            for (int a = 0; a < 2; a++)
#pragma unroll
                for (int e = 0; e < 16; e++) acc[a][e] = 0.f;
            __syncthreads();
            for (int kt = 0; kt < nk; kt++) {
                const int buf = (ts * nk + kt) & 1;
                __builtin_amdgcn_s_setprio(1);
                if (BF16) {
#pragma unroll
                    for (int ks = 0; ks < KB / 16; ks++) {
                        bf16x8 a0, a1, b0;
#pragma unroll
                        for (int j = 0; j < 8; j++) {
                            const int k = 16 * ks + 8 * lk + j;
                            a0[j] = (__bf16)lds.A[buf][k][wm * 64 + lr];
                            a1[j] = (__bf16)lds.A[buf][k][wm * 64 + 32 + lr];
                            b0[j] = (__bf16)lds.B[buf][k][wn * 32 + lr];
                        }
                        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[0], 0, 0, 0);
                        acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc[1], 0, 0, 0);
                    }
                } else {
#pragma unroll
                    for (int kk = 0; kk < KB / 2; kk++) {
                        const float a0 = lds.A[buf][2 * kk + lk][wm * 64 + lr];
                        const float a1 = lds.A[buf][2 * kk + lk][wm * 64 + 32 + lr];
                        const float b0 = lds.B[buf][2 * kk + lk][wn * 32 + lr];
                        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0], 0, 0, 0);
                        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1], 0, 0, 0);
                    }
                }
                __builtin_amdgcn_s_setprio(0);
                __syncthreads();
            }
            if (nk == 0) __syncthreads();
#pragma unroll
            for (int a = 0; a < 2; a++)
#pragma unroll
                for (int e = 0; e < 16; e++)
                    lds.T[wm * 64 + a * 32 + (e & 3) + 8 * (e >> 2) + 4 * lk][wn * 32 + lr] = acc[a][e];
            prev = TileRef{bi * TB, bj * TB, bi == bj};
            have_prev = true;
            if (!SLAB && bi == bj) {
                // a diagonal tile is written out at once by all twelve waves (element-wise mirror rule): it
                // is never the "previous tile" of the pipelined path
                __syncthreads();
                syrk_drain<SLAB>(g, lds.T, prev, fl_pair & 1, fl_s, fl_pair >> 1, 32, 3, use_old);
                have_prev = false;
            }
        }
    }
    __syncthreads();
    if (have_prev) {
        // the last tile: all twelve waves stream it out
        if (SLAB) syrk_drain<SLAB>(g, lds.T, prev, 0, fl_s, fl_pair, 32, 6, use_old);
        else syrk_drain<SLAB>(g, lds.T, prev, fl_pair & 1, fl_s, fl_pair >> 1, 32, 3, use_old);
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

}  // namespace

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

bool syrk_rda_f32(const float *D, long ldd, const int32_t *ids, const int32_t *ids2, const float *w, const int *k_dev,
                  int kmax, int F, float alpha, float beta, float *C, long ldc, hipStream_t s, int slab_col0, int slab_cols, bool bf16)
{
    if (F % TB != 0 || kmax % KB != 0 || ldd % 4 != 0 || (reinterpret_cast<uintptr_t>(D) & 15) != 0) return false;
    const bool slab = slab_cols > 0;
    if (slab && (slab_col0 % TB != 0 || slab_cols % TB != 0 || slab_col0 + slab_cols > F)) return false;
    SyrkDev g;
    g.D = D; g.ldd = ldd; g.ids = ids; g.ids2 = ids2; g.w = w; g.k_dev = k_dev; g.kmax = kmax; g.F = F;
    g.C = C; g.ldc = ldc; g.alpha = alpha; g.beta = beta; g.nt = F / TB;
    g.slab_t0 = slab ? slab_col0 / TB : 0; g.slab_nt = slab ? slab_cols / TB : 0;
    const int ntiles = slab ? g.nt * g.slab_nt : g.nt * (g.nt + 1) / 2;
    static const int old_early = std::getenv("DLCO_SYRK_OLD_EARLY") ? 1 : 0, noprio = std::getenv("DLCO_SYRK_PRIO") ? 0 : 1;
    g.old_early = old_early; g.noprio = noprio;
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
    {
        static int n_cu = 0, units = -1;
        if (n_cu == 0) {
            int dev = 0; hipDeviceProp_t prop;
            DLCO_HIP(hipGetDevice(&dev));
            DLCO_HIP(hipGetDeviceProperties(&prop, dev));
            n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
            units = std::getenv("DLCO_SYRK_STAGGER") ? std::atoi(std::getenv("DLCO_SYRK_STAGGER")) : 1;
        }
        g.stagger_from = n_cu;
        g.stagger_units = ntiles >= 4 * n_cu ? units : 0;      // only worth it over several rounds of tiles
    }
    // The persistent, wave-specialised kernel is opt-in (DLCO_SYRK_STREAM=1): measured on MI355X it is SLOWER than
    // the tile-per-workgroup kernel (0.39 ms against 0.30 ms at K = 303, F = 8192; 0.25 against 0.15 ms with bf16
    // MFMAs), see DESIGN.md section 3.
    static const bool use_v1 = std::getenv("DLCO_SYRK_STREAM") == nullptr;
    int n_wg = 256;
    {
        int dev = 0; hipDeviceProp_t prop;
        static int n_cu_s = 0;
        if (n_cu_s == 0) {
            DLCO_HIP(hipGetDevice(&dev));
            DLCO_HIP(hipGetDeviceProperties(&prop, dev));
            n_cu_s = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
        }
        n_wg = std::max(8, n_cu_s / 8 * 8);
        n_wg = std::min(n_wg, (ntiles + 7) / 8 * 8);
    }
    static const bool use_k8 = std::getenv("DLCO_SYRK_W4") == nullptr;         // DLCO_SYRK_W4=1: the four-wave kernel
    static const bool other = std::getenv("DLCO_SYRK_K3") || std::getenv("DLCO_SYRK_LEAN") || std::getenv("DLCO_SYRK_STREAM");
    if (use_k8 && !other) {
#define DLCO_SYRK_K8_LAUNCH(P, S, H) hipLaunchKernelGGL((syrk_rda_kernel8<P, S, H>), dim3(ntiles), dim3(NT8), 0, s, g)
        if (bf16) {
            if (slab) { if (ids2) DLCO_SYRK_K8_LAUNCH(true, true, true); else DLCO_SYRK_K8_LAUNCH(false, true, true); }
            else { if (ids2) DLCO_SYRK_K8_LAUNCH(true, false, true); else DLCO_SYRK_K8_LAUNCH(false, false, true); }
        } else {
            if (slab) { if (ids2) DLCO_SYRK_K8_LAUNCH(true, true, false); else DLCO_SYRK_K8_LAUNCH(false, true, false); }
            else { if (ids2) DLCO_SYRK_K8_LAUNCH(true, false, false); else DLCO_SYRK_K8_LAUNCH(false, false, false); }
        }
#undef DLCO_SYRK_K8_LAUNCH
        DLCO_HIP(hipGetLastError());
        if (tracing) syrk_dump_trace(trace_path, trace_buf, ntiles, s);
        return true;
    }
    static const bool use_k3 = std::getenv("DLCO_SYRK_K3") != nullptr;
    if (use_k3) {
#define DLCO_SYRK_K3_LAUNCH(P, S, H) hipLaunchKernelGGL((syrk_rda_kernel3<P, S, H>), dim3(ntiles), dim3(NTH), 0, s, g)
        if (bf16) {
            if (slab) { if (ids2) DLCO_SYRK_K3_LAUNCH(true, true, true); else DLCO_SYRK_K3_LAUNCH(false, true, true); }
            else { if (ids2) DLCO_SYRK_K3_LAUNCH(true, false, true); else DLCO_SYRK_K3_LAUNCH(false, false, true); }
        } else {
            if (slab) { if (ids2) DLCO_SYRK_K3_LAUNCH(true, true, false); else DLCO_SYRK_K3_LAUNCH(false, true, false); }
            else { if (ids2) DLCO_SYRK_K3_LAUNCH(true, false, false); else DLCO_SYRK_K3_LAUNCH(false, false, false); }
        }
#undef DLCO_SYRK_K3_LAUNCH
        DLCO_HIP(hipGetLastError());
        return true;
    }
    static const bool use_lean = std::getenv("DLCO_SYRK_LEAN") != nullptr;
    if (use_lean) {
#define DLCO_SYRK_LEAN_LAUNCH(P, S, H) hipLaunchKernelGGL((syrk_rda_lean_kernel<P, S, H>), dim3(ntiles), dim3(NTH), 0, s, g)
        if (bf16) {
            if (slab) { if (ids2) DLCO_SYRK_LEAN_LAUNCH(true, true, true); else DLCO_SYRK_LEAN_LAUNCH(false, true, true); }
            else { if (ids2) DLCO_SYRK_LEAN_LAUNCH(true, false, true); else DLCO_SYRK_LEAN_LAUNCH(false, false, true); }
        } else {
            if (slab) { if (ids2) DLCO_SYRK_LEAN_LAUNCH(true, true, false); else DLCO_SYRK_LEAN_LAUNCH(false, true, false); }
            else { if (ids2) DLCO_SYRK_LEAN_LAUNCH(true, false, false); else DLCO_SYRK_LEAN_LAUNCH(false, false, false); }
        }
#undef DLCO_SYRK_LEAN_LAUNCH
        DLCO_HIP(hipGetLastError());
        return true;
    }
    // pair mode (four loads per row and K block) and row lists beyond the LDS capacity stay with the tile-per-workgroup kernel
    const bool stream = !use_v1 && !ids2 && kmax <= SK_MAX;
    if (stream) {
        if (bf16) {
            if (slab) hipLaunchKernelGGL((syrk_rda_stream_kernel<true, true>), dim3(n_wg), dim3(ST), 0, s, g);
            else hipLaunchKernelGGL((syrk_rda_stream_kernel<false, true>), dim3(n_wg), dim3(ST), 0, s, g);
        } else {
            if (slab) hipLaunchKernelGGL((syrk_rda_stream_kernel<true, false>), dim3(n_wg), dim3(ST), 0, s, g);
            else hipLaunchKernelGGL((syrk_rda_stream_kernel<false, false>), dim3(n_wg), dim3(ST), 0, s, g);
        }
        DLCO_HIP(hipGetLastError());
        return true;
    }
#define DLCO_SYRK_LAUNCH(P, S, H) hipLaunchKernelGGL((syrk_rda_kernel<P, S, H>), dim3(ntiles), dim3(NTH), 0, s, g)
    if (bf16) {
        if (slab) { if (ids2) DLCO_SYRK_LAUNCH(true, true, true); else DLCO_SYRK_LAUNCH(false, true, true); }
        else { if (ids2) DLCO_SYRK_LAUNCH(true, false, true); else DLCO_SYRK_LAUNCH(false, false, true); }
    } else {
        if (slab) { if (ids2) DLCO_SYRK_LAUNCH(true, true, false); else DLCO_SYRK_LAUNCH(false, true, false); }
        else { if (ids2) DLCO_SYRK_LAUNCH(true, false, false); else DLCO_SYRK_LAUNCH(false, false, false); }
    }
#undef DLCO_SYRK_LAUNCH
    DLCO_HIP(hipGetLastError());
    if (tracing) syrk_dump_trace(trace_path, trace_buf, ntiles, s);
    return true;
}

}  // namespace dlco
