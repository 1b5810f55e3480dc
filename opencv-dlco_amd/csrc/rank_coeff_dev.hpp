// rank_coeff_dev.hpp — device code shared by kernels_rankupd.hip and kernels_syrk.hip: the coefficient fragments of the
// rank-update first filter term (see kernels_rankupd.hip).  One call = one (K block, 32-row tile) of 64 lanes.
#pragma once

#include "dlco_internal.hpp"

namespace dlco {

typedef __bf16 rc_bf16x8 __attribute__((ext_vector_type(8)));

// Two-way split fragments in MFMA A-operand order: entry ((kb * MT + tile) * 2 + plane) * 64 + lane holds
// C_w[tile*32 + (lane & 31)][kb*16 + 8*(lane >> 5) .. + 7],  C_w[i][k] = w[k] * proj[row(i)][slot[k]] / wscale[i],
// row(i) = nw-1-i for a Ritz row i < nw (W is in ascending order, src/pj-learn.cpp:480-484), i for a guard row behind.
__device__ __forceinline__ void rank_coeff_block(const RankCoeffJob &job, const float *w, int kact, int kb, int tile, int lane)
{
    const int i = tile * 32 + (lane & 31), k0 = kb * 16 + 8 * (lane >> 5);
    rc_bf16x8 h, l;
#pragma unroll
    for (int j = 0; j < 8; j++) { h[j] = (__bf16)0.f; l[j] = (__bf16)0.f; }
    if (i < job.m) {
        const float *prow = job.proj + (long)(i < job.nw ? job.nw - 1 - i : i) * job.ldp;
        const float inv = 1.0f / job.wscale[i];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int k = k0 + j;
            float v = 0.f;
            if (k < kact) v = w[k] * (prow[job.slot[k]] * inv);
            h[j] = (__bf16)v;
            l[j] = (__bf16)(v - (float)h[j]);
        }
    }
    rc_bf16x8 *o = static_cast<rc_bf16x8 *>(job.frag) + ((long)(kb * job.MT + tile) * 2) * 64 + lane;
    o[0] = h;
    o[64] = l;
}

}  // namespace dlco
