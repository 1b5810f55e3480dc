// hbm_shapes.cpp — developer micro-benchmark: read bandwidth of one pass over an 8192 x 8192 fp32 matrix
// (268 MB, row pitch 32 KiB) as a function of the SHAPE in which a workgroup walks it.  Each workgroup
// (256 threads, 16-byte loads, 64 KiB in flight) reads `rows` rows x `seg` bytes per step and sums what it
// reads; the grid covers the matrix exactly once.  Used to choose the tile shape of the tracker's product
// kernels (kernels_bf16x2.hip).
//   hipcc --offload-arch=gfx950 -O2 -std=c++17 tools/hbm_shapes.cpp -o tools/hbm_shapes
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));

// A workgroup owns `rows` consecutive rows x `cols` floats starting at (r0, c0) and walks it in steps of
// rows x seg_f floats (seg_f * 4 = contiguous bytes per row and step); steps are rotated by `rot`.
template <bool NT>
__global__ __launch_bounds__(256) void walk(const float *G, long ld, int rows, int cols, int seg_f, int tiles_x, int rotate, float *out)
{
    const int bx = blockIdx.x % tiles_x, by = blockIdx.x / tiles_x;
    const long r0 = (long)by * rows, c0 = (long)bx * cols;
    const int nsteps = cols / seg_f;
    const int per_row = seg_f / 4;                 // float4 per row and step
    const int total = rows * per_row;              // float4 per step (multiple of 256 for the shapes used)
    const int rot = rotate ? (int)(blockIdx.x % (unsigned)nsteps) : 0;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < nsteps; s++) {
        int sp = s + rot;
        if (sp >= nsteps) sp -= nsteps;
        for (int f = threadIdx.x; f < total; f += 256) {
            const int r = f / per_row, c4 = f % per_row;
            const f32x4 *p = reinterpret_cast<const f32x4 *>(G + (r0 + r) * ld + c0 + (long)sp * seg_f + 4 * c4);
            const f32x4 v = NT ? __builtin_nontemporal_load(p) : *p;
            acc += v;
        }
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 123.456f) out[blockIdx.x] = acc[0];
}

int main()
{
    const int F = 8192;
    float *G, *out;
    hipMalloc(&G, (size_t)F * F * 4);
    hipMalloc(&out, 1 << 20);
    hipMemset(G, 0, (size_t)F * F * 4);
    hipStream_t s;
    hipStreamCreate(&s);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    struct Shape { int rows, cols, seg_f; };
    // rows x cols floats per workgroup (area 262144 floats = 1 MiB -> 256 workgroups), seg_f floats per row and step
    const Shape shapes[] = {
        {128, 2048, 128},   // 128 rows x 512 B   (the row-streaming product kernel)
        {128, 2048, 32},    // 128 rows x 128 B   (k-major fragments of the older kernel, transposed roles)
        {64, 4096, 256},    // 64 rows x 1 KiB
        {32, 8192, 512},    // 32 rows x 2 KiB
        {16, 8192, 1024},   // 16 rows x 4 KiB, 512 workgroups
        {8, 8192, 2048},    // 8 rows x 8 KiB, 1024 workgroups
        {1, 8192, 8192},    // whole rows, 8192 workgroups
        {256, 1024, 64},    // 256 rows x 256 B
    };
    for (const Shape &sh : shapes) {
        const int tiles_x = F / sh.cols, tiles_y = F / sh.rows;
        for (int rotate = 0; rotate < 2; rotate++)
            for (int nt = 0; nt < 2; nt++) {
                auto launch = [&] {
                    if (nt) hipLaunchKernelGGL(walk<true>, dim3(tiles_x * tiles_y), dim3(256), 0, s, G, (long)F, sh.rows, sh.cols, sh.seg_f, tiles_x, rotate, out);
                    else hipLaunchKernelGGL(walk<false>, dim3(tiles_x * tiles_y), dim3(256), 0, s, G, (long)F, sh.rows, sh.cols, sh.seg_f, tiles_x, rotate, out);
                };
                launch();
                hipStreamSynchronize(s);
                hipEventRecord(a, s);
                for (int i = 0; i < 10; i++) launch();
                hipEventRecord(b, s);
                hipEventSynchronize(b);
                float ms = 0.f;
                hipEventElapsedTime(&ms, a, b);
                ms /= 10;
                std::printf("%4d rows x %5d B per step, tile %4d x %5d floats, %5d workgroups, rotate %d, nt %d: %7.1f us  %.2f TB/s\n", sh.rows,
                            sh.seg_f * 4, sh.rows, sh.cols, tiles_x * tiles_y, rotate, nt, ms * 1e3, 4.0 * F * F / (ms * 1e-3) / 1e12);
            }
    }
    return 0;
}
