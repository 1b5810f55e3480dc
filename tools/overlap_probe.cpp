// overlap_probe — can fp32 MFMA work and an HBM read-modify-write stream overlap on one MI355X CU?
// Three launches of the same grid (2 workgroups per CU): role M = a dependent-free chain of v_mfma_f32_32x32x2_f32,
// role S = out = 0.9 * in + 1 over a 268 MB matrix, and both roles interleaved (even / odd workgroups).
// Build: hipcc --offload-arch=gfx950 -O3 -o overlap_probe tools/overlap_probe.cpp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

// mode: 0 = all workgroups M, 1 = all S, 2 = even M / odd S (each role then has half the workgroups and does the whole job)
__global__ __launch_bounds__(256, 2) void probe(int mode, int mfma_iters, const float *in, float *out, long n4, float *sink)
{
    // modes 3 / 4: the halves of mode 2 on their own (the other half of the workgroups exits at once)
    if (mode == 3 && (blockIdx.x & 1)) return;
    if (mode == 4 && !(blockIdx.x & 1)) return;
    const bool split = mode >= 2;
    const int role = split ? (blockIdx.x & 1) : mode;           // 0 = M, 1 = S
    const int nrole = split ? gridDim.x / 2 : gridDim.x, irole = split ? blockIdx.x / 2 : blockIdx.x;
    if (role == 0) {
        f32x16 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
        const float x = (float)threadIdx.x, y = 1.0f;
        // total MFMA work is fixed: mfma_iters per workgroup slot of the full grid
        const int iters = mfma_iters * (int)gridDim.x / nrole;
        for (int i = 0; i < iters; i++) {
            a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a1, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a2, 0, 0, 0);
            a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a3, 0, 0, 0);
        }
        if (a0[0] + a1[1] + a2[2] + a3[3] == 12345.f) sink[0] = 1.f;
    } else {
        const f32x4 *i4 = reinterpret_cast<const f32x4 *>(in);
        f32x4 *o4 = reinterpret_cast<f32x4 *>(out);
        for (long p = (long)irole * 256 + threadIdx.x; p < n4; p += (long)nrole * 256) {
            f32x4 v = i4[p];
            o4[p] = v * 0.9f + 1.0f;
        }
    }
}

int main()
{
    const long F = 8192, n = F * F, n4 = n / 4;
    float *in, *out, *sink;
    CK(hipMalloc(&in, n * 4)); CK(hipMalloc(&out, n * 4)); CK(hipMalloc(&sink, 64));
    CK(hipMemset(in, 0, n * 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int grid = 512;
    // 4 MFMAs of 64 cycles per iteration and wave; one wave per SIMD and workgroup: pick iters for ~140 us of MFMA
    for (int iters : {650, 1300}) {
        float ms[5] = {0, 0, 0, 0, 0};
        for (int mode = 0; mode < 5; mode++) {
            for (int rep = 0; rep < 6; rep++) {
                CK(hipEventRecord(e0));
                hipLaunchKernelGGL(probe, dim3(grid), dim3(256), 0, 0, mode, iters, in, out, n4, sink);
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                float t; CK(hipEventElapsedTime(&t, e0, e1));
                if (rep >= 2) ms[mode] += t / 4;
            }
        }
        printf("mfma iters %d: all workgroups MFMA %.1f us | all stream %.1f us (%.2f TB/s rd+wr) | half MFMA alone %.1f us, half stream alone %.1f us, "
               "the two halves together %.1f us (sum %.1f, max %.1f)\n", iters, ms[0] * 1e3, ms[1] * 1e3, 2.0 * n * 4 / ms[1] / 1e9, ms[3] * 1e3, ms[4] * 1e3,
               ms[2] * 1e3, (ms[3] + ms[4]) * 1e3, (ms[3] > ms[4] ? ms[3] : ms[4]) * 1e3);
    }
    return 0;
}
