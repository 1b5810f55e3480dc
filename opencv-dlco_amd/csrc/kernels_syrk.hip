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
// store of the tile, transposed store through LDS as whole 512-byte rows.  fp32 results: by default from
// the bf16 matrix cores with both operands split three ways at staging time (see PRE in the kernel), or
// from v_mfma_f32_32x32x2_f32 (a k-ordered fmaf chain) with DLCO_SYRK_FP32=1.
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

typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

union SyrkLds {
    struct { float A[2][KB][LD]; float B[2][KB][LD]; } st;    // 67,584 B
    // three-way split mode: per stage and operand three bf16 planes (hi, mid, lo) of a 16-deep K block, entry
    // [k / 4][column] = the four consecutive k of that column (8 bytes): an MFMA fragment is two 8-byte reads
    struct { bf16x4 A[2][3][4][TB]; bf16x4 B[2][3][4][TB]; } sp;   // 49,152 B
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

    // BF16 == 3: fp32 results from the bf16 matrix cores.  Every staged value is split three ways, x = hi + mid + lo (all
    // 24 mantissa bits), ONCE, by the thread that gathers it, and stored as three bf16 planes; the K loop then forms the
    // six products down to 2^-16 (hh, hm, mh, mm, hl, lh; smallest first, fp32 accumulation) - 6/16 of the fp32-MFMA
    // time, no conversion work in front of the MFMAs, 16-deep K blocks (49 KB of LDS: still two workgroups per CU).
    constexpr bool PRE = BF16 == 3;
    constexpr int KD = PRE ? 16 : KB;
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
    // loader mapping (three-way split): wave -> (operand, group of four K rows), lane -> a pair of columns; the four
    // rows of a wave are wave-uniform, the lane loads 8 bytes of each
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const int p_op = wave >> 2, p_kq = wave & 3, p_c2 = lane;
    int32_t pid[4], pid2[4];
    float pw[4], pwl[4];                                      // weights of the rows whose ids / whose values are held
    f32x2 pv[4], pv2[4];
    auto load_ids = [&](int kt) {
        if (PRE) {
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int k = kt * KD + p_kq * 4 + u;
                pid[u] = g.ids[k];
                if (PAIR) pid2[u] = g.ids2[k];
                pw[u] = p_op == 0 ? g.w[k] : 1.0f;
            }
            return;
        }
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const int k = kt * KB + rbase + 16 * u;
            id_nx[u] = g.ids[k];
            if (PAIR) id2_nx[u] = g.ids2[k];
            w_nx[u] = g.w[k];
        }
    };
    auto load_rows = [&]() {
        if (PRE) {
            const int col = (p_op == 0 ? i0 : j0) + 2 * p_c2;
#pragma unroll
            for (int u = 0; u < 4; u++) {
                pv[u] = *reinterpret_cast<const f32x2 *>(g.D + (long)pid[u] * g.ldd + col);
                if (PAIR) pv2[u] = *reinterpret_cast<const f32x2 *>(g.D + (long)pid2[u] * g.ldd + col);
                pwl[u] = pw[u];                                  // (the ids and weights of the NEXT block are fetched before these values are stored)
            }
            return;
        }
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
        if (PRE) {
            bf16x4 e[3][2];                                      // [plane][column of the pair]: the four K rows of this wave
#pragma unroll
            for (int u = 0; u < 4; u++)
#pragma unroll
                for (int cc = 0; cc < 2; cc++) {
                    float v = pv[u][cc];
                    if (PAIR) v -= pv2[u][cc];                   // Dist = Desc1 - Desc2, src/comp-uprjdists.cpp:327
                    v *= pwl[u];                                 // w_k x_k for the A operand (w = 1 for B); same rounding as the fp32 path
                    const __bf16 h = (__bf16)v;
                    const float r1 = v - (float)h;
                    const __bf16 m = (__bf16)r1;
                    e[0][cc][u] = h; e[1][cc][u] = m; e[2][cc][u] = (__bf16)(r1 - (float)m);
                }
            bf16x4 *base = p_op == 0 ? &lds.sp.A[buf][0][0][0] : &lds.sp.B[buf][0][0][0];
#pragma unroll
            for (int pl = 0; pl < 3; pl++) {
                bf16x4 *dst = base + ((pl * 4 + p_kq) * TB + 2 * p_c2);
                dst[0] = e[pl][0];
                dst[1] = e[pl][1];
            }
            return;
        }
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
        if (PRE) {
            // fragment of lane (lr, lk): K rows 8 lk .. 8 lk + 7 of the block = entries [2 lk] and [2 lk + 1] of its column
            auto frag = [&](const bf16x4 *plane, int col) {
                const bf16x4 lo4 = plane[(2 * lk) * TB + col], hi4 = plane[(2 * lk + 1) * TB + col];
                bf16x8 f;
#pragma unroll
                for (int j = 0; j < 4; j++) { f[j] = lo4[j]; f[4 + j] = hi4[j]; }
                return f;
            };
            const int bcol = wn * 32 + lr;
            const bf16x8 bh = frag(&lds.sp.B[buf][0][0][0], bcol), bm = frag(&lds.sp.B[buf][1][0][0], bcol), bl = frag(&lds.sp.B[buf][2][0][0], bcol);
#pragma unroll
            for (int a = 0; a < 2; a++) {
                const int acol = wm * 64 + 32 * a + lr;
                const bf16x8 ah = frag(&lds.sp.A[buf][0][0][0], acol), am = frag(&lds.sp.A[buf][1][0][0], acol), al = frag(&lds.sp.A[buf][2][0][0], acol);
                acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[a], 0, 0, 0);
                acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[a], 0, 0, 0);
                acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, acc[a], 0, 0, 0);
                acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, acc[a], 0, 0, 0);
                acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, acc[a], 0, 0, 0);
                acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[a], 0, 0, 0);
            }
        } else if (BF16 == 1) {
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


bool syrk_rda_f32(const float *D, long ldd, const int32_t *ids, const int32_t *ids2, const float *w, const int *k_dev,
                  int kmax, int F, float alpha, float beta, float *C, long ldc, hipStream_t s, int slab_col0, int slab_cols, bool bf16,
                  bool packed)
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
#define DLCO_SYRK_LAUNCH(P, S, H) hipLaunchKernelGGL((syrk_rda_kernel8<P, S, H>), dim3(ntiles), dim3(NT8), 0, s, g)
#define DLCO_SYRK_LAUNCH_PK(P, H) hipLaunchKernelGGL((syrk_rda_kernel8<P, false, H, true>), dim3(ntiles), dim3(NT8), 0, s, g)
    // precision of the products: 3 (default) = three-way split bf16, fp32-level results at 3/8 of the fp32 matrix time
    // (0.189 against 0.219 ms per launch at K = 305: the launch is then spread over matrix time, conversions, LDS and its
    // HBM phase instead of waiting on the fp32 MFMA pipe); 0 = fp32 MFMA, a k-ordered fmaf chain (DLCO_SYRK_FP32=1);
    // 1 = operands rounded to bf16 once (cfg.grad_bf16, the configs[4] variant)
    static const bool exact_fp32 = std::getenv("DLCO_SYRK_FP32") != nullptr;
    const int prec = bf16 ? 1 : (exact_fp32 ? 0 : 3);
    if (packed) {
        if (prec == 1) { if (ids2) DLCO_SYRK_LAUNCH_PK(true, 1); else DLCO_SYRK_LAUNCH_PK(false, 1); }
        else if (prec == 3) { if (ids2) DLCO_SYRK_LAUNCH_PK(true, 3); else DLCO_SYRK_LAUNCH_PK(false, 3); }
        else { if (ids2) DLCO_SYRK_LAUNCH_PK(true, 0); else DLCO_SYRK_LAUNCH_PK(false, 0); }
    } else if (prec == 1) {
        if (slab) { if (ids2) DLCO_SYRK_LAUNCH(true, true, 1); else DLCO_SYRK_LAUNCH(false, true, 1); }
        else { if (ids2) DLCO_SYRK_LAUNCH(true, false, 1); else DLCO_SYRK_LAUNCH(false, false, 1); }
    } else if (prec == 3) {
        if (slab) { if (ids2) DLCO_SYRK_LAUNCH(true, true, 3); else DLCO_SYRK_LAUNCH(false, true, 3); }
        else { if (ids2) DLCO_SYRK_LAUNCH(true, false, 3); else DLCO_SYRK_LAUNCH(false, false, 3); }
    } else {
        if (slab) { if (ids2) DLCO_SYRK_LAUNCH(true, true, 0); else DLCO_SYRK_LAUNCH(false, true, 0); }
        else { if (ids2) DLCO_SYRK_LAUNCH(true, false, 0); else DLCO_SYRK_LAUNCH(false, false, 0); }
    }
#undef DLCO_SYRK_LAUNCH_PK
#undef DLCO_SYRK_LAUNCH
    DLCO_HIP(hipGetLastError());
    if (tracing) syrk_dump_trace(trace_path, trace_buf, ntiles, s);
    return true;
}

}  // namespace dlco
