// mall_probe.cpp — developer micro-benchmark: does a dual average stored as its packed upper triangle
// (2080 tiles x 64 KiB = 136 MB at F = 8192) stay in the 256 MiB Infinity Cache between the passes of a
// training step, where the full symmetric matrix (268 MB) cannot?  For a table of S MB it times
//   (a) repeated read passes over the table (every byte once per pass),
//   (b) passes in which every byte is read TWICE by two different workgroups (the symmetric product: tile
//       (I, J) is the B operand of output block I and, transposed, of output block J),
//   (c) the same with 24 MB of unrelated traffic (write + read) between two passes (planes, slabs, blocks),
//   (d) a read-modify-write pass over the table (the SYRK epilogue) followed by a read pass.
//   hipcc --offload-arch=gfx950 -O2 -std=c++17 tools/mall_probe.cpp -o tools/mall_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));

// a workgroup reads `chunk` contiguous bytes (as 16-byte loads, 8 in flight per thread) of up to two places
__global__ __launch_bounds__(256) void read_pass(const f32x4 *T, long chunk_v, long nchunks, int twice, float *out)
{
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int rep = 0; rep <= twice; rep++) {
        long c = blockIdx.x;
        if (rep) c = (c * 7 + nchunks / 2 + 3) % nchunks;         // another workgroup's chunk, far away in time and place
        const f32x4 *p = T + c * chunk_v;
        for (long i = threadIdx.x; i < chunk_v; i += 256 * 8) {
            f32x4 v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) v[u] = (i + u * 256 < chunk_v) ? p[i + u * 256] : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int u = 0; u < 8; u++) acc += v[u];
        }
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 123.456f) out[blockIdx.x] = acc[0];
}

__global__ __launch_bounds__(256) void rmw_pass(f32x4 *T, long chunk_v, float a)
{
    f32x4 *p = T + (long)blockIdx.x * chunk_v;
    for (long i = threadIdx.x; i < chunk_v; i += 256 * 8) {
        f32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) v[u] = (i + u * 256 < chunk_v) ? p[i + u * 256] : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < 8; u++)
            if (i + u * 256 < chunk_v) p[i + u * 256] = v[u] * a + 1e-9f;
    }
}

int main()
{
    hipStream_t s;
    hipStreamCreate(&s);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    float *out, *other;
    hipMalloc(&out, 1 << 22);
    const size_t other_bytes = (size_t)24 << 20;
    hipMalloc(&other, other_bytes);
    hipMemset(other, 0, other_bytes);
    const long chunk = 64 << 10;                                   // one 128 x 128 fp32 tile
    const long chunk_v = chunk / 16;
    const int sizes_mb[] = {32, 64, 100, 136, 170, 200, 230, 268, 400};
    for (int mb : sizes_mb) {
        const long nchunks = ((long)mb << 20) / chunk;
        const size_t bytes = (size_t)nchunks * chunk;
        float *T;
        if (hipMalloc(&T, bytes) != hipSuccess) { std::printf("alloc %d MB failed\n", mb); return 1; }
        hipMemset(T, 0, bytes);
        auto timeit = [&](const char *what, double logical_bytes, auto &&body) {
            body();
            hipStreamSynchronize(s);
            hipEventRecord(a, s);
            const int reps = 20;
            for (int i = 0; i < reps; i++) body();
            hipEventRecord(b, s);
            hipEventSynchronize(b);
            float ms = 0.f;
            hipEventElapsedTime(&ms, a, b);
            ms /= reps;
            std::printf("%4d MB  %-58s %8.1f us  %6.2f TB/s logical\n", mb, what, ms * 1e3, logical_bytes / (ms * 1e-3) / 1e12);
        };
        const unsigned g = (unsigned)nchunks;
        timeit("read once per pass", (double)bytes, [&] { hipLaunchKernelGGL(read_pass, dim3(g), dim3(256), 0, s, (const f32x4 *)T, chunk_v, nchunks, 0, out); });
        timeit("every byte twice per pass (two workgroups)", 2.0 * bytes, [&] { hipLaunchKernelGGL(read_pass, dim3(g), dim3(256), 0, s, (const f32x4 *)T, chunk_v, nchunks, 1, out); });
        timeit("twice per pass + 24 MB rmw of other data between passes", 2.0 * bytes, [&] {
            hipLaunchKernelGGL(read_pass, dim3(g), dim3(256), 0, s, (const f32x4 *)T, chunk_v, nchunks, 1, out);
            hipLaunchKernelGGL(rmw_pass, dim3((unsigned)(other_bytes / chunk)), dim3(256), 0, s, (f32x4 *)other, chunk_v, 0.5f);
        });
        timeit("rmw pass over the table alone", 2.0 * bytes, [&] { hipLaunchKernelGGL(rmw_pass, dim3(g), dim3(256), 0, s, (f32x4 *)T, chunk_v, 0.999f); });
        timeit("rmw pass + 4 double-read passes (one training step)", 2.0 * bytes + 8.0 * bytes, [&] {
            hipLaunchKernelGGL(rmw_pass, dim3(g), dim3(256), 0, s, (f32x4 *)T, chunk_v, 0.999f);
            for (int q = 0; q < 4; q++) {
                hipLaunchKernelGGL(read_pass, dim3(g), dim3(256), 0, s, (const f32x4 *)T, chunk_v, nchunks, 1, out);
                hipLaunchKernelGGL(rmw_pass, dim3((unsigned)(other_bytes / chunk)), dim3(256), 0, s, (f32x4 *)other, chunk_v, 0.5f);
            }
        });
        hipFree(T);
    }
    return 0;
}
