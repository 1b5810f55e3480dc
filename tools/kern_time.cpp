// kern_time.cpp — developer micro-benchmark of the single-workgroup kernels of the tracker
// (links against libdlco.so; not part of the product or of the tests).
//   hipcc --offload-arch=gfx950 -O2 -std=c++17 -I opencv-dlco_amd/csrc tools/kern_time.cpp \
//         -L opencv-dlco_amd -ldlco -Wl,-rpath,'$ORIGIN/../opencv-dlco_amd' -o tools/kern_time
#include "dlco_internal.hpp"

#include <cmath>
#include <cstdio>
#include <random>
#include <vector>

using namespace dlco;

static float time_ms(hipStream_t s, int reps, const std::function<void()> &fn)
{
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    fn();
    hipStreamSynchronize(s);
    hipEventRecord(a, s);
    for (int i = 0; i < reps; i++) fn();
    hipEventRecord(b, s);
    hipEventSynchronize(b);
    float ms = 0.f;
    hipEventElapsedTime(&ms, a, b);
    return ms / reps;
}

int main()
{
    hipStream_t s;
    hipStreamCreate(&s);
    std::mt19937 rng(7);
    std::normal_distribution<float> nd(0.f, 1.f);
    for (int n : {32, 64, 96, 128}) {
        // ---- Cholesky inverse on M = B B^T (rows of B random, nearly orthogonal after scaling) ----
        const int K = 512, ld = 1024;
        std::vector<double> B((size_t)n * K);
        for (auto &x : B) x = nd(rng);
        std::vector<float> M((size_t)n * ld, 0.f);
        for (int i = 0; i < n; i++)
            for (int j = 0; j < n; j++) {
                double acc = 0;
                for (int k = 0; k < K; k++) acc += B[(size_t)i * K + k] * B[(size_t)j * K + k];
                M[(size_t)i * ld + j] = (float)(acc / K);
            }
        DevBuf<float> dM, dL; DevBuf<int> dead;
        dM.alloc(M.size()); dL.alloc(M.size()); dead.alloc(128);
        hipMemcpy(dM.p, M.data(), M.size() * 4, hipMemcpyHostToDevice);
        hipMemset(dL.p, 0, M.size() * 4);
        const float ms = time_ms(s, 200, [&] { chol_inverse128(dM.p, ld, n, 1e-5f, dL.p, ld, dead.p, s); });
        std::vector<float> L(M.size());
        hipMemcpy(L.data(), dL.p, M.size() * 4, hipMemcpyDeviceToHost);
        double err = 0;
        for (int i = 0; i < n; i++)
            for (int j = 0; j < n; j++) {
                double acc = 0;                                   // (Linv M Linv^T)_ij
                for (int a = 0; a < n; a++) {
                    double t = 0;
                    for (int b = 0; b < n; b++) t += (double)M[(size_t)a * ld + b] * L[(size_t)j * ld + b];
                    acc += (double)L[(size_t)i * ld + a] * t;
                }
                err = std::max(err, std::fabs(acc - (i == j ? 1.0 : 0.0)));
            }
        std::printf("chol_inverse128 n=%3d  %.1f us  |Linv M Linv^T - I|max = %.2e\n", n, ms * 1e3, err);

        // ---- Jacobi on a nearly diagonal symmetric matrix (steady-state Rayleigh-Ritz) ------------
        std::vector<float> T((size_t)n * ld, 0.f);
        for (int i = 0; i < n; i++)
            for (int j = 0; j <= i; j++) {
                const float v = (i == j) ? (1.0f - 0.9f * i / n) : 2e-3f * nd(rng) / std::sqrt((float)n);
                T[(size_t)i * ld + j] = v; T[(size_t)j * ld + i] = v;
            }
        DevBuf<float> dT, dV, ev, work; DevBuf<int> sw;
        dT.alloc(T.size()); dV.alloc(T.size()); ev.alloc(n + 8); work.alloc(jacobi_work_floats(n)); sw.alloc(4);
        hipMemcpy(dT.p, T.data(), T.size() * 4, hipMemcpyHostToDevice);
        const float msj = time_ms(s, 100, [&] { jacobi_eigh(dT.p, ld, n, ev.p, dV.p, ld, work.p, sw.p, s); });
        int sweeps = 0;
        hipMemcpy(&sweeps, sw.p, 4, hipMemcpyDeviceToHost);
        std::vector<float> V(T.size()), e(n);
        hipMemcpy(V.data(), dV.p, T.size() * 4, hipMemcpyDeviceToHost);
        hipMemcpy(e.data(), ev.p, n * 4, hipMemcpyDeviceToHost);
        double res = 0, orth = 0;
        for (int j = 0; j < n; j++) {
            for (int i = 0; i < n; i++) {
                double acc = 0;
                for (int k = 0; k < n; k++) acc += (double)T[(size_t)i * ld + k] * V[(size_t)k * ld + j];
                res = std::max(res, std::fabs(acc - (double)e[j] * V[(size_t)i * ld + j]));
            }
            for (int k = 0; k < n; k++) {
                double acc = 0;
                for (int i = 0; i < n; i++) acc += (double)V[(size_t)i * ld + j] * V[(size_t)i * ld + k];
                orth = std::max(orth, std::fabs(acc - (j == k ? 1.0 : 0.0)));
            }
        }
        std::printf("jacobi_eigh     n=%3d  %.1f us  sweeps %d (%.1f us/sweep)  |TV - VE|max = %.2e  |V^T V - I|max = %.2e\n", n,
                    msj * 1e3, sweeps, msj * 1e3 / std::max(1, sweeps), res, orth);
    }
    return 0;
}
