// kern_time.cpp — developer micro-benchmark of the single-workgroup kernels of the tracker
// (links against libdlco.so; not part of the product or of the tests).
//   hipcc --offload-arch=gfx950 -O2 -std=c++17 -I opencv-dlco_amd/csrc tools/kern_time.cpp \
//         -L opencv-dlco_amd -ldlco -Wl,-rpath,'$ORIGIN/../opencv-dlco_amd' -o tools/kern_time
#include "dlco_internal.hpp"

#include <cmath>
#include <cstdio>
#include <functional>
#include <random>
#include <string>
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

int main(int argc, char **argv)
{
    const bool small_only = argc > 1 && std::string(argv[1]) == "small";
    const bool prod_only = argc > 1 && std::string(argv[1]) == "prod";
    hipStream_t s0;
    hipStreamCreate(&s0);
    if (prod_only) {
        // one pass of the tracker over a symmetric F x F matrix: two-way (filter) and three-way (Rayleigh-Ritz) split
        const int F = 8192, M = 96;
        std::mt19937 r0(11);
        std::normal_distribution<float> g0(0.f, 1.f);
        DevBuf<float> C, X, out, slab; DevBuf<char> ph, pl, pl2;
        C.alloc((size_t)F * F); X.alloc((size_t)128 * F); out.alloc((size_t)128 * F); slab.alloc(bf16x2_slab_floats(128, F, 8));   // the ks = 8 case below needs 8 slices
        ph.alloc(bf16x2_plane_bytes(128, F)); pl.alloc(bf16x2_plane_bytes(128, F)); pl2.alloc(bf16x2_plane_bytes(128, F));
        std::vector<float> h((size_t)128 * F);
        for (auto &x : h) x = g0(r0);
        hipMemcpy(X.p, h.data(), h.size() * 4, hipMemcpyHostToDevice);
        for (int i = 0; i < 64; i++) hipMemcpy(C.p + (size_t)i * 128 * F, h.data(), h.size() * 4, hipMemcpyHostToDevice);
        const float a = time_ms(s0, 30, [&] { skinny_product_bf16x2(X.p, F, M, C.p, F, F, F, 1.f, out.p, F, nullptr, 0.f, nullptr, 0.f, ph.p, pl.p, slab.p, s0); });
        const float b = time_ms(s0, 30, [&] { skinny_product_bf16x2(X.p, F, M, C.p, F, F, F, 1.f, out.p, F, nullptr, 0.f, nullptr, 0.f, ph.p, pl.p, slab.p, s0, 0, pl2.p); });
        const float c8 = time_ms(s0, 30, [&] { skinny_product_bf16x2(X.p, F, M, C.p, F, F, F, 1.f, out.p, F, nullptr, 0.f, nullptr, 0.f, ph.p, pl.p, slab.p, s0, 8); });
        const float t2 = time_ms(s0, 30, [&] { skinny_product_bf16x2(X.p, F, M, C.p, F, F, F, 1.f, out.p, F, nullptr, 0.f, nullptr, 0.f, ph.p, pl.p, slab.p, s0, 0, nullptr, true); });
        const float t3 = time_ms(s0, 30, [&] { skinny_product_bf16x2(X.p, F, M, C.p, F, F, F, 1.f, out.p, F, nullptr, 0.f, nullptr, 0.f, ph.p, pl.p, slab.p, s0, 0, pl2.p, true); });
        {   // symmetric G as packed upper tiles: every tile fetched once (skinny_sym_kernel)
            DevBuf<float> Gp, slab4, X160, out160; DevBuf<char> qh, ql;
            Gp.alloc(syrk_packed_floats(F)); slab4.alloc((size_t)4 * 160 * F); X160.alloc((size_t)160 * F); out160.alloc((size_t)160 * F);
            qh.alloc(bf16x2_plane_bytes(160, F)); ql.alloc(bf16x2_plane_bytes(160, F));
            hipMemcpy(X160.p, h.data(), (size_t)128 * F * 4, hipMemcpyHostToDevice);
            hipMemcpy(X160.p + (size_t)128 * F, h.data(), (size_t)32 * F * 4, hipMemcpyHostToDevice);
            syrk_pack_upper(C.p, F, F, Gp.p, s0);
            const float y2 = time_ms(s0, 30, [&] { skinny_product_sym(X.p, F, M, Gp.p, F, 1.f, out.p, F, nullptr, 0.f, nullptr, 0.f, ph.p, pl.p, slab4.p, s0); });
            const float y3 = time_ms(s0, 30, [&] { skinny_product_sym(X.p, F, M, Gp.p, F, 1.f, out.p, F, nullptr, 0.f, nullptr, 0.f, ph.p, pl.p, slab4.p, s0, pl2.p); });
            const float y5 = time_ms(s0, 30, [&] { skinny_product_sym(X160.p, F, 160, Gp.p, F, 1.f, out160.p, F, nullptr, 0.f, nullptr, 0.f, qh.p, ql.p, slab4.p, s0); });
            const float z5 = time_ms(s0, 30, [&] { skinny_product_bf16x2(X160.p, F, 160, C.p, F, F, F, 1.f, out160.p, F, nullptr, 0.f, nullptr, 0.f, qh.p, ql.p, slab4.p, s0); });
            std::printf("SYMMETRIC product on packed upper tiles (136 MB): M=96 two-way %.1f us, three-way %.1f us; M=160 two-way %.1f us (full-matrix kernel %.1f us)  [split + kernel + reduce]\n",
                        y2 * 1e3, y3 * 1e3, y5 * 1e3, z5 * 1e3);
        }
        std::printf("product on a TILED matrix (128 x 128 tiles contiguous): two-way %.1f us (%.2f TB/s), three-way %.1f us (%.2f TB/s)\n", t2 * 1e3,
                    4.0 * F * F / (t2 * 1e-3) / 1e12, t3 * 1e3, 4.0 * F * F / (t3 * 1e-3) / 1e12);
        std::printf("product M=%d F=%d: two-way %.1f us (%.2f TB/s), three-way %.1f us (%.2f TB/s), two-way ks=8 %.1f us  [split + kernel + reduce]\n", M, F,
                    a * 1e3, 4.0 * F * F / (a * 1e-3) / 1e12, b * 1e3, 4.0 * F * F / (b * 1e-3) / 1e12, c8 * 1e3);
        return 0;
    }
    hipStream_t s;
    hipStreamCreate(&s);
    std::mt19937 rng(7);
    std::normal_distribution<float> nd(0.f, 1.f);
    const bool big = argc > 2 && std::string(argv[2]) == "big";      // `kern_time small big`: the transient's block sizes too
    for (int n : {32, 64, 70, 96, 97, 128, 150, 160, 200, 256, 400, 800}) {
        if (n > 160 && !big) break;
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
        if (n <= CHOL_INV_MAX_N) {
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
        }

        // ---- Jacobi on a nearly diagonal symmetric matrix (steady-state Rayleigh-Ritz); eigenvectors = ROWS of V ----
        std::vector<float> T((size_t)n * ld, 0.f);
        for (int i = 0; i < n; i++)
            for (int j = 0; j <= i; j++) {
                const float v = (i == j) ? (1.0f - 0.9f * i / n) : 2e-3f * nd(rng) / std::sqrt((float)n);
                T[(size_t)i * ld + j] = v; T[(size_t)j * ld + i] = v;
            }
        DevBuf<float> dT, dV, ev, work; DevBuf<int> sw;
        dT.alloc(T.size()); dV.alloc(T.size()); ev.alloc(n + 8); work.alloc(jacobi_work_floats(n)); sw.alloc(4);
        hipMemcpy(dT.p, T.data(), T.size() * 4, hipMemcpyHostToDevice);
        const float msj = time_ms(s, n > 160 ? 5 : 100, [&] { jacobi_eigh(dT.p, ld, n, ev.p, dV.p, ld, work.p, sw.p, s); });
        int sweeps = 0;
        hipMemcpy(&sweeps, sw.p, 4, hipMemcpyDeviceToHost);
        std::vector<float> V(T.size()), e(n);
        hipMemcpy(V.data(), dV.p, T.size() * 4, hipMemcpyDeviceToHost);
        hipMemcpy(e.data(), ev.p, n * 4, hipMemcpyDeviceToHost);
        double res = 0, orth = 0;
        for (int j = 0; j < n; j++) {
            for (int i = 0; i < n; i++) {
                double acc = 0;
                for (int k = 0; k < n; k++) acc += (double)T[(size_t)i * ld + k] * V[(size_t)j * ld + k];
                res = std::max(res, std::fabs(acc - (double)e[j] * V[(size_t)j * ld + i]));
            }
            for (int k = 0; k < n; k++) {
                double acc = 0;
                for (int i = 0; i < n; i++) acc += (double)V[(size_t)j * ld + i] * V[(size_t)k * ld + i];
                orth = std::max(orth, std::fabs(acc - (j == k ? 1.0 : 0.0)));
            }
        }
        std::printf("jacobi_eigh     n=%3d  %.1f us  sweeps %d (%.1f us/sweep)  |TV - VE|max = %.2e  |V^T V - I|max = %.2e\n", n,
                    msj * 1e3, sweeps, msj * 1e3 / std::max(1, sweeps), res, orth);
        // ---- the same on a matrix that needs large rotations: M above (eigenvalues clustered around 1) and a wide spectrum ----
        for (int kind = 0; kind < 2; kind++) {
            std::vector<float> H((size_t)n * ld, 0.f);
            for (int i = 0; i < n; i++)
                for (int j = 0; j < n; j++) {
                    if (kind == 0) { H[(size_t)i * ld + j] = M[(size_t)i * ld + j]; continue; }
                    double acc = 0;                               // B diag(10^(-6 k / K)) B^T / K: six decades
                    for (int k = 0; k < K; k++) acc += B[(size_t)i * K + k] * B[(size_t)j * K + k] * std::pow(10.0, -6.0 * k / K) * (k % 7 == 0 ? 100.0 : 1.0);
                    H[(size_t)i * ld + j] = (float)(acc / K);
                }
            hipMemcpy(dT.p, H.data(), H.size() * 4, hipMemcpyHostToDevice);
            const float msh = time_ms(s, n > 160 ? 3 : 20, [&] { jacobi_eigh(dT.p, ld, n, ev.p, dV.p, ld, work.p, sw.p, s); });
            hipMemcpy(&sweeps, sw.p, 4, hipMemcpyDeviceToHost);
            hipMemcpy(V.data(), dV.p, T.size() * 4, hipMemcpyDeviceToHost);
            hipMemcpy(e.data(), ev.p, n * 4, hipMemcpyDeviceToHost);
            double resh = 0, orthh = 0, emax = 0;
            for (int j = 0; j < n; j++) {
                emax = std::max(emax, (double)std::fabs(e[j]));
                for (int i = 0; i < n; i++) {
                    double acc = 0;
                    for (int k = 0; k < n; k++) acc += (double)H[(size_t)i * ld + k] * V[(size_t)j * ld + k];
                    resh = std::max(resh, std::fabs(acc - (double)e[j] * V[(size_t)j * ld + i]));
                }
                for (int k = 0; k < n; k++) {
                    double acc = 0;
                    for (int i = 0; i < n; i++) acc += (double)V[(size_t)j * ld + i] * V[(size_t)k * ld + i];
                    orthh = std::max(orthh, std::fabs(acc - (j == k ? 1.0 : 0.0)));
                }
            }
            std::printf("jacobi_eigh     n=%3d  dense %s  %.1f us  sweeps %d  |HV - VE|max / |E|max = %.2e  |V^T V - I|max = %.2e\n", n,
                        kind == 0 ? "clustered" : "wide     ", msh * 1e3, sweeps, resh / emax, orthh);
        }
    }
    if (small_only) return 0;
    // ---- slab-mode kernels at the 8-GPU shape (F = 8192, slab of 1024 columns, global batch 1600+1600) ----
    {
        const int F = 8192, R = 4096, world = 8, cw = F / world;
        DevBuf<float> D, C, X, out, slab; DevBuf<int32_t> ids; DevBuf<float> w; DevBuf<int> kd; DevBuf<char> ph, pl;
        D.alloc((size_t)R * F); C.alloc((size_t)F * F); X.alloc((size_t)128 * F); out.alloc((size_t)128 * F);
        slab.alloc((size_t)32 * 128 * F / 4); ids.alloc(3200); w.alloc(3200); kd.alloc(4);
        ph.alloc(bf16x2_plane_bytes(128, F)); pl.alloc(bf16x2_plane_bytes(128, F));
        std::vector<float> h((size_t)R * F);
        for (auto &x : h) x = 0.1f * nd(rng);
        hipMemcpy(D.p, h.data(), h.size() * 4, hipMemcpyHostToDevice);
        hipMemcpy(X.p, h.data(), (size_t)128 * F * 4, hipMemcpyHostToDevice);
        hipMemset(C.p, 0, (size_t)F * F * 4);
        std::vector<int32_t> hid(3200); std::vector<float> hw(3200);
        for (int i = 0; i < 3200; i++) { hid[i] = (int)(rng() % R); hw[i] = (float)(1 + rng() % 50); }
        hipMemcpy(ids.p, hid.data(), 3200 * 4, hipMemcpyHostToDevice);
        hipMemcpy(w.p, hw.data(), 3200 * 4, hipMemcpyHostToDevice);
        for (int K : {0, 32, 64, 128, 224, 448, 896}) {                      // fixed cost vs per-K-tile cost of the SYRK
            hipMemcpy(kd.p, &K, 4, hipMemcpyHostToDevice);
            const float a = time_ms(s, 20, [&] { syrk_rda_f32(D.p, F, ids.p, nullptr, w.p, kd.p, 3200, F, 1e-3f, 0.5f, C.p, F, s); });
            const float b0 = time_ms(s, 20, [&] { syrk_rda_f32(D.p, F, ids.p, nullptr, w.p, kd.p, 3200, F, 1e-3f, 0.0f, C.p, F, s); });
            std::printf("syrk K=%4d  symmetric full: beta=0.5 %.1f us   beta=0 (no read of the old tile) %.1f us\n", K, a * 1e3, b0 * 1e3);
        }
        for (int K : {224, 1696}) {
            hipMemcpy(kd.p, &K, 4, hipMemcpyHostToDevice);
            const float a = time_ms(s, 20, [&] { syrk_rda_f32(D.p, F, ids.p, nullptr, w.p, kd.p, 3200, F, 1e-3f, 0.5f, C.p, F, s); });
            const float b = time_ms(s, 20, [&] { syrk_rda_f32(D.p, F, ids.p, nullptr, w.p, kd.p, 3200, F, 1e-3f, 0.5f, C.p, F, s, cw, cw); });
            std::printf("syrk K=%4d  symmetric full %.1f us   column slab 1/%d (dense) %.1f us\n", K, a * 1e3, world, b * 1e3);
        }
        // the per-rank kernels of the same-global-batch (strong scaling) mode at G = 2, 4, 8: column slab of the gradient over the
        // 305 active rows of the c2 step, slab of a 96-row tracker product (the projection's inputs)
        for (int G : {2, 4, 8}) {
            const int cwg = F / G, K = 305;
            hipMemcpy(kd.p, &K, 4, hipMemcpyHostToDevice);
            const float a = time_ms(s, 20, [&] { syrk_rda_f32(D.p, F, ids.p, nullptr, w.p, kd.p, 416, F, 1e-3f, 0.5f, C.p, F, s, cwg, cwg); });
            const float b = time_ms(s, 20, [&] { skinny_product_bf16x2(X.p, F, 96, C.p + cwg, F, cwg, F, 1.f, out.p + cwg, F, nullptr, 0.f, nullptr, 0.f, ph.p, pl.p, slab.p, s, 4 * G); });
            const float e = time_ms(s, 20, [&] { pack_cols(slab.p, out.p, F, cwg, cwg, 96, s); unpack_cols(out.p, F, slab.p, cwg, 96, G, s); });
            std::printf("G=%d: gradient column slab (K = 305 active rows, %d dense tiles) %.1f us; product slab 96 x 8192 x %d (two-way, ks %d) %.1f us; pack + unpack %.1f us\n",
                        G, (F / 128) * (cwg / 128), a * 1e3, cwg, 4 * G, b * 1e3, e * 1e3);
        }
        for (int M : {96}) {
            const float a = time_ms(s, 20, [&] { skinny_product_bf16x2(X.p, F, M, C.p, F, F, F, 1.f, out.p, F, nullptr, 0.f, nullptr, 0.f, ph.p, pl.p, slab.p, s); });
            const float b = time_ms(s, 20, [&] { skinny_product_bf16x2(X.p, F, M, C.p + cw, F, cw, F, 1.f, out.p + cw, F, nullptr, 0.f, nullptr, 0.f, ph.p, pl.p, slab.p, s, 32); });
            GemmArgs g;
            g.M = M; g.N = cw; g.K = F;
            g.A.p = X.p; g.A.ld = F; g.A.kmajor = false;
            g.B.p = C.p + cw; g.B.ld = F; g.B.kmajor = true;
            g.C = out.p + cw; g.ldc = F; g.split_k = 32; g.slab = slab.p;
            const float c = time_ms(s, 20, [&] { gemm_f32(g, s); });
            const float d = time_ms(s, 20, [&] { skinny_product_f32(X.p, F, M, 128, C.p, F, F, F, 1.f, out.p, F, nullptr, 0.f, nullptr, 0.f, s); });
            std::printf("product M=%d  bf16x2 full %.1f us, slab 1/%d (ks 32) %.1f us;  fp32 full (skinny) %.1f us, slab (gemm split 32) %.1f us\n",
                        M, a * 1e3, world, b * 1e3, d * 1e3, c * 1e3);
            const float e = time_ms(s, 20, [&] { pack_cols(slab.p, out.p, F, cw, cw, M, s); unpack_cols(out.p, F, slab.p, cw, M, world, s); });
            std::printf("pack + unpack %.1f us\n", e * 1e3);
        }
    }
    return 0;
}
