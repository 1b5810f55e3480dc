// kernels_rankupd.hip — the first Chebyshev term of a step's filter WITHOUT a pass over the dual average.
//
// The tracker (eig_tracker.cpp) leaves every step with the Ritz block Q and Y = Q H_t, H_t = -dfAvg_t, both at fp32
// accuracy.  The step that follows changes the matrix by a rank <= 2B update whose factors are already on the device:
//
//     dfAvg_{t+1} = beta dfAvg_t + alpha sum_k w_k x_k x_k^T                     (Q1 + U1, src/pj-learn.cpp:367-422)
//     Q dfAvg_{t+1} = -beta Y + alpha (Q X_a^T) diag(w) X_a
//
// X_a = the active rows of the batch, whose three-way bf16 planes syrk_split_rows_kernel has just written for the
// gradient; Q X_a^T = the projections of the batch rows on the block, which the step's own P1 (`project_few`) computed at
// its start (the rows of W are the scaled Ritz rows; the guard rows ride along unscaled behind them).  The first filter
// term (H - c0) Q / e0 of the step's tracker update is therefore
//
//     Z1 = (beta / e0) Y - (alpha / e0) C_w X_a - (c0 / e0) Q ,     C_w[i][k] = w_k (q_i . x_k)
//
// a (m x K_a) x (K_a x F) product with K_a ~ 300 instead of a (m x F) x (F x F) pass over 136 MB of packed tiles:
// 46 us + split + reduce -> a few us (rank_first_term_kernel), and the result is MORE accurate than the two-way split
// product it replaces (Y is an fp32-level product; only the small update term carries the 2^-17 of a two-way split).
// The output also leaves as the two-way planes the next product of the Chebyshev chain reads (split_x_kernel's order).
//
// Only the filter is fed this way - it enriches a subspace; the Rayleigh-Ritz product, the residuals and the convergence
// test still use the matrix itself, so a stale Y could cost passes, never a wrong result.
#include "dlco_internal.hpp"
#include "rank_coeff_dev.hpp"

namespace dlco {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int RU_TB = 128;                    // column tile of the gradient's planes (kernels_syrk.hip: TB)
constexpr int RU_KD = 16;                     // rows per K block there (PL_KD)
constexpr int RU_IMG = 3 * 2 * RU_TB * 16;    // bytes of one (operand, K block, column tile) image (PL_IMG)

// Coefficient fragments (rank_coeff_dev.hpp) as a launch of their own: one block of 64 lanes per (K block, row tile).  The
// trainer's steps let them ride in the gradient's row-split launch instead (RankCoeffJob).
__global__ __launch_bounds__(64) void rank_coeff_kernel(RankCoeffJob job, const float *w, const int *k_dev, int kmax)
{
    const int kb = blockIdx.x, tile = blockIdx.y;
    const int kact = min(*k_dev, kmax);
    if (kb * RU_KD >= kact) return;
    rank_coeff_block(job, w, kact, kb, tile, threadIdx.x);
}

struct RankUpdDev {
    const float *Y, *Q;           // [m][ld]
    float *out;                   // [m][ld]
    long ld;
    const bf16x8 *frag;           // rank_coeff_kernel's output
    const char *planes;           // syrk_split_rows_kernel's planes (operand 1 = x is read)
    const int *k_dev;
    int kmax, nkb, nt;            // capacity of the row list, its K blocks (kmax / 16), column tiles F / 128
    int m, F;
    float ay, aq, ac;             // out = ay*Y + aq*Q + ac * C_w X_a
    bf16x8 *hi, *lo;              // two-way planes of `out` in split_x_kernel's order (may be null)
};

// One workgroup = 32 output columns x all rows.  The launch is pure latency (15 MB of planes over 256 workgroups, a few
// hundred MFMAs each), so the K blocks of the active rows are dealt to KS wave groups: wave = (row tile a, K group g) takes
// the blocks g, g + KS, ... - at most two trips of four blocks, every load of a trip in flight before its MFMAs - with two
// coefficient fragments (L2) and three plane fragments (L2 / Infinity Cache: the SYRK has just read them) per block and
// five MFMAs (ah*bh, ah*bm, al*bh, ah*bl, al*bm: everything down to 2^-24 of the two-way coefficient).  The KS partial
// tiles meet in LDS; one thread per plane entry (a row's eight consecutive columns) then forms ay*Y + aq*Q + ac*sum - its
// Y and Q values requested before the K loop - and stores the row segment and the two plane entries.
// MODE 3: the three-way planes of syrk_split_rows_kernel (one 12 KB image per 16-row block: hi, mid, lo).  MODE 1: the bf16-once
// planes of syrk_round_rows_kernel (cfg.grad_bf16: one image per 48-row block, its three slots = three 16-row sub-blocks of
// values rounded once) - two MFMAs per block; the update the gradient applied was formed from these rounded values, and the
// projections were rounded the same way, so the term is exact to the 2^-9 of that arithmetic in its (1/t-sized) update part.
template <int MT, int KS, int MODE>
__global__ __launch_bounds__(64 * MT * KS) void rank_first_term_kernel(RankUpdDev g)
{
    __shared__ float P[KS][MT * 32][36];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int tile = wave % MT, kg = wave / MT;
    const int f0 = blockIdx.x * 32;
    const int ct = f0 >> 7, cl = (f0 & 127) + (lane & 31), lk = lane >> 5;
    const int kact = min(*g.k_dev, g.kmax);
    const int nkb = (kact + RU_KD - 1) / RU_KD;
    // epilogue role: entry e = (row, eight columns c8 .. c8 + 7)
    const int e = threadIdx.x, erow = e >> 2, ec8 = (e & 3) * 8;
    const bool eact = e < MT * 32 * 4 && erow < g.m;
    f32x4 y0, y1, q0, q1;
    if (eact) {
        const long idx = (long)erow * g.ld + f0 + ec8;
        y0 = *reinterpret_cast<const f32x4 *>(g.Y + idx); y1 = *reinterpret_cast<const f32x4 *>(g.Y + idx + 4);
        q0 = *reinterpret_cast<const f32x4 *>(g.Q + idx); q1 = *reinterpret_cast<const f32x4 *>(g.Q + idx + 4);
    }
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; r++) acc[r] = 0.f;
    const bf16x8 *xb = reinterpret_cast<const bf16x8 *>(g.planes + ((long)g.nkb * g.nt + ct) * RU_IMG) + lk * RU_TB + cl;   // operand 1, K block 0
    const long xstride = (long)g.nt * (RU_IMG / 16);
    const bf16x8 *af = g.frag + (long)tile * 128 + lane;
    const long astride = (long)MT * 128;
    int kb = kg;
    for (; kb + 3 * KS < nkb; kb += 4 * KS) {
        bf16x8 ah[4], al[4], bh[4], bm[4], bl[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int k = kb + u * KS;
            ah[u] = af[k * astride]; al[u] = af[k * astride + 64];
            if (MODE == 3) { bh[u] = xb[k * xstride]; bm[u] = xb[k * xstride + 2 * RU_TB]; bl[u] = xb[k * xstride + 4 * RU_TB]; }
            else bh[u] = xb[(k / 3) * xstride + (k % 3) * 2 * RU_TB];
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            if (MODE == 3) {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[u], bm[u], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[u], bl[u], acc, 0, 0, 0);
            }
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[u], bh[u], acc, 0, 0, 0);
            if (MODE == 3) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[u], bm[u], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[u], bh[u], acc, 0, 0, 0);
        }
    }
    {
        // the last (partial) trip: up to three blocks, loads first
        bf16x8 ah[3], al[3], bh[3], bm[3], bl[3];
#pragma unroll
        for (int u = 0; u < 3; u++) {
            const int k = kb + u * KS;
            if (k < nkb) {
                ah[u] = af[k * astride]; al[u] = af[k * astride + 64];
                if (MODE == 3) { bh[u] = xb[k * xstride]; bm[u] = xb[k * xstride + 2 * RU_TB]; bl[u] = xb[k * xstride + 4 * RU_TB]; }
                else bh[u] = xb[(k / 3) * xstride + (k % 3) * 2 * RU_TB];
            }
        }
#pragma unroll
        for (int u = 0; u < 3; u++) {
            if (kb + u * KS < nkb) {
                if (MODE == 3) {
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[u], bm[u], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[u], bl[u], acc, 0, 0, 0);
                }
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[u], bh[u], acc, 0, 0, 0);
                if (MODE == 3) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[u], bm[u], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[u], bh[u], acc, 0, 0, 0);
            }
        }
    }
    // accumulator register r of a lane: row (r & 3) + 8 (r >> 2) + 4 (lane >> 5) of the wave's tile, column lane & 31
#pragma unroll
    for (int r = 0; r < 16; r++) P[kg][tile * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk][lane & 31] = acc[r];
    __syncthreads();
    if (e >= MT * 32 * 4) return;
    f32x4 s0 = *reinterpret_cast<const f32x4 *>(&P[0][erow][ec8]), s1 = *reinterpret_cast<const f32x4 *>(&P[0][erow][ec8 + 4]);
#pragma unroll
    for (int z = 1; z < KS; z++) {
        s0 += *reinterpret_cast<const f32x4 *>(&P[z][erow][ec8]);
        s1 += *reinterpret_cast<const f32x4 *>(&P[z][erow][ec8 + 4]);
    }
    bf16x8 h, l;
    if (eact) {
        const f32x4 o0 = g.ay * y0 + g.aq * q0 + g.ac * s0, o1 = g.ay * y1 + g.aq * q1 + g.ac * s1;
        const long idx = (long)erow * g.ld + f0 + ec8;
        *reinterpret_cast<f32x4 *>(g.out + idx) = o0;
        *reinterpret_cast<f32x4 *>(g.out + idx + 4) = o1;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const float x = j < 4 ? o0[j & 3] : o1[j & 3];
            h[j] = (__bf16)x;
            l[j] = (__bf16)(x - (float)h[j]);
        }
    } else {
#pragma unroll
        for (int j = 0; j < 8; j++) { h[j] = (__bf16)0.f; l[j] = (__bf16)0.f; }       // rows >= m of a tile: zero planes
    }
    if (g.hi) {
        // plane entry ((k16 * MT + tile) * 64 + pl): row tile*32 + (pl & 31), columns k16*16 + 8 (pl >> 5) .. + 7
        const int col = f0 + ec8;
        const long t = ((long)(col >> 4) * MT + (erow >> 5)) * 64 + (erow & 31) + 32 * ((col >> 3) & 1);
        g.hi[t] = h;
        g.lo[t] = l;
    }
}

template <int MT, int KS>
void launch_first_term(const RankUpdDev &g, int mode, hipStream_t s)
{
    if (mode == 3) hipLaunchKernelGGL((rank_first_term_kernel<MT, KS, 3>), dim3(g.F / 32), dim3(64 * MT * KS), 0, s, g);
    else hipLaunchKernelGGL((rank_first_term_kernel<MT, KS, 1>), dim3(g.F / 32), dim3(64 * MT * KS), 0, s, g);
}

}  // namespace

size_t rank_coeff_bytes(int m, int kmax) { return (size_t)(kmax / RU_KD) * ((m + 31) / 32) * 2 * 64 * sizeof(bf16x8); }

// out = ay*Y + aq*Q + ac * C_w X_a with C_w built from the step's projection (see the header comment); returns false for
// shapes the kernels do not take (the caller then forms the term with a product over the matrix).
bool rank_first_term(const float *Y, const float *Q, long ld, int m, int F, float ay, float aq, float ac, float *out,
                     const float *proj, long ldp, int nw, const float *wscale, const int32_t *slot, const float *w,
                     const int *k_dev, int kmax, const void *planes, void *coeff_ws, void *plane_hi, void *plane_lo, hipStream_t s,
                     bool coeff_ready, int planes_mode)
{
    if (planes_mode != 3 && planes_mode != 1) return false;
    const int mt = (m + 31) / 32;
    if (m < 1 || mt > 5 || F % 128 != 0 || kmax % 32 != 0 || kmax < 32) return false;
    if (ld % 4 != 0 || (reinterpret_cast<uintptr_t>(Y) & 15) != 0 || (reinterpret_cast<uintptr_t>(Q) & 15) != 0 ||
        (reinterpret_cast<uintptr_t>(out) & 15) != 0) return false;
    if (!coeff_ready) {
        RankCoeffJob job;
        job.proj = proj; job.ldp = ldp; job.nw = nw; job.m = m; job.MT = mt; job.wscale = wscale; job.slot = slot; job.frag = coeff_ws;
        hipLaunchKernelGGL(rank_coeff_kernel, dim3(kmax / RU_KD, mt), dim3(64), 0, s, job, w, k_dev, kmax);
    }
    RankUpdDev g;
    g.Y = Y; g.Q = Q; g.out = out; g.ld = ld;
    g.frag = static_cast<const bf16x8 *>(coeff_ws);
    g.planes = static_cast<const char *>(planes);
    g.k_dev = k_dev; g.kmax = kmax; g.nkb = kmax / RU_KD; g.nt = F / RU_TB;
    g.m = m; g.F = F; g.ay = ay; g.aq = aq; g.ac = ac;
    g.hi = static_cast<bf16x8 *>(plane_hi); g.lo = static_cast<bf16x8 *>(plane_lo);
    switch (mt) {
    case 1: launch_first_term<1, 4>(g, planes_mode, s); break;
    case 2: launch_first_term<2, 4>(g, planes_mode, s); break;
    case 3: launch_first_term<3, 4>(g, planes_mode, s); break;
    case 4: launch_first_term<4, 2>(g, planes_mode, s); break;
    default: launch_first_term<5, 2>(g, planes_mode, s); break;
    }
    DLCO_HIP(hipGetLastError());
    return true;
}

}  // namespace dlco
