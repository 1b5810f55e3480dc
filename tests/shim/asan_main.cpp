// Sanitizer target for the host code ("ASan/UBSan CI target for host code"): one executable, built with
// -fsanitize=address,undefined, that drives every piece of host-side logic that compiles without a GPU -
//   * the product's pair indexing and cv::RNG restatement      opencv-dlco_amd/csrc/pair_index.hpp
//   * the tools' I/O layer (.npy directories, row streams)      opencv-dlco_amd/cli/dlco_io.hpp
//   * the export wire format                                     opencv-dlco_amd/cli/export_format.hpp
//   * the CPU oracle (test infrastructure)                       oracle/dlco_ref.c
// - on small inputs, cross-checking product against oracle on the way.  (host_comm.cpp and the CLIs' main() call HIP and
// cannot be linked here; their argument handling is covered by tests/test_cli_gpu.py.)  `make -C tests/shim asan` builds and
// runs it; tests/test_sanitizers.py does that in the CPU suite.  Exit code 0 = no sanitizer report and all checks passed.
#include "../../opencv-dlco_amd/cli/dlco_io.hpp"
#include "../../opencv-dlco_amd/cli/export_format.hpp"
#include "../../opencv-dlco_amd/csrc/pair_index.hpp"

extern "C" {
#include "../../oracle/dlco_ref.h"
}

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#define CHECK(c)                                                                          \
    do {                                                                                  \
        if (!(c)) { std::fprintf(stderr, "asan_main: check failed at line %d: %s\n", __LINE__, #c); return 1; } \
    } while (0)

static uint32_t lcg(uint32_t &s) { s = s * 1664525u + 1013904223u; return s >> 8; }
static float unif(uint32_t &s) { return (float)(lcg(s) & 0xffff) / 32768.0f - 1.0f; }

int main(int argc, char **argv)
{
    const std::string tmp = argc > 1 ? argv[1] : "/tmp";
    uint32_t seed = 12345u;

    // ---- pair indexing: product header against the oracle, ragged labels, tiny and empty classes ---------------------
    for (int N : {2, 7, 333, 4097}) {
        std::vector<uint8_t> lab(N);
        for (int i = 0; i < N; i++) lab[i] = (uint8_t)(lcg(seed) % 5 == 0 ? 2 : lcg(seed) % 2);
        dlco::PairIndex idx;
        idx.build(lab.data(), N);
        std::vector<int32_t> pos(N), neg(N);
        int np = 0, nn = 0;
        dlco_ref_build_index(lab.data(), N, pos.data(), &np, neg.data(), &nn);
        CHECK(np == (int)idx.pos.size() && nn == (int)idx.neg.size());
        CHECK((int)dlco_ref_split((size_t)np) == idx.n_pos_trn && (int)dlco_ref_split((size_t)nn) == idx.n_neg_trn);
        CHECK(np == 0 || std::memcmp(pos.data(), idx.pos.data(), np * sizeof(int32_t)) == 0);   // (an empty class has no storage)
        CHECK(nn == 0 || std::memcmp(neg.data(), idx.neg.data(), nn * sizeof(int32_t)) == 0);
    }
    {
        dlco::CvRng r(2215);
        uint64_t st = 2215;
        for (int i = 0; i < 1000; i++) CHECK(r.uniform(0, 200000) == dlco_ref_rng_uniform(&st, 0, 200000));
        float a = 0.f, b = 0.f;
        dlco::rda_coeffs(200u, 107000u, &a, &b);               // beyond the reference's 32-bit overflow point
        CHECK(std::isfinite(a) && std::isfinite(b));
    }

    // ---- the oracle's trainer, statistics and operators on a small problem (built-in loops and OpenBLAS if present) ----
    {
        const int N = 600, F = 24, B = 16;
        std::vector<float> D((size_t)N * F);
        std::vector<uint8_t> L(N);
        for (int i = 0; i < N; i++) {
            L[i] = (uint8_t)(i % 2 == 0);
            for (int f = 0; f < F; f++) D[(size_t)i * F + f] = unif(seed) * (L[i] ? 0.4f : 1.0f);
        }
        for (int order = 0; order < 2; order++) {
            dlco_ref_ctx *c = dlco_ref_create(D.data(), L.data(), N, F, B, 0.004f, 0.5f);
            CHECK(c != nullptr);
            dlco_ref_set_grad_order(c, order);
            for (int s = 0; s < 12; s++) CHECK(dlco_ref_step(c) == 0);
            float lo = 0.f, rg = 0.f, f95 = 0.f;
            double auc = 0.0;
            int dim = 0;
            dlco_ref_validate(c, &lo, &rg);
            dlco_ref_stats(c, &dim, &f95, &auc);
            CHECK(std::isfinite(lo) && std::isfinite(rg) && auc >= 0.0 && auc <= 1.0 && dim >= 0 && dim <= F);
            unsigned t = 0;
            int r = 0;
            std::vector<float> df((size_t)F * F), W((size_t)F * F), A((size_t)F * F), dl((size_t)F * F);
            dlco_ref_get_state(c, &t, &r, W.data(), A.data(), df.data(), dl.data());
            CHECK(t == 12u && r >= 0 && r <= F);
            dlco_ref_set_state(c, 3, df.data(), W.data(), r);
            CHECK(dlco_ref_step(c) == 0);
            dlco_ref_destroy(c);
        }
        // edges: one pair per class, all-equal distances (ties in the ROC sweep), the TPR == 0.95 boundary
        std::vector<float> d(40);
        std::vector<uint8_t> l(40);
        for (int i = 0; i < 40; i++) { d[i] = (float)(i / 4); l[i] = (uint8_t)(i % 2); }
        float f95 = 0.f;
        double auc = 0.0;
        dlco_ref_roc_stats(d.data(), l.data(), 40, &f95, &auc);
        CHECK(f95 >= 0.f && f95 <= 1.f);
        const float p1[1] = {0.5f}, n1[1] = {0.7f};
        CHECK(dlco_ref_hinge_sum(p1, 1, n1, 1) > 0.0);
        CHECK(dlco_ref_hinge_sum(p1, 0, n1, 1) == 0.0);
        int32_t rho[3], kap[3];
        const float pd[3] = {0.f, 1.f, 2.f}, nd[3] = {1.f, 1.f, 3.f};
        dlco_ref_viol_counts(pd, nd, 3, rho, kap);
        CHECK(rho[0] == 0 && rho[1] == 2 && rho[2] == 2);      // strict >: (0 + 1) > 1 is false
    }
    // ---- pr-learn restatement + descriptor ------------------------------------------------------------------------------
    {
        const int N = 400, F = 40;
        std::vector<float> D((size_t)N * F);
        std::vector<uint8_t> L(N);
        for (int i = 0; i < N; i++) {
            L[i] = (uint8_t)(i % 2 == 0);
            for (int f = 0; f < F; f++) D[(size_t)i * F + f] = std::fabs(unif(seed)) * (L[i] ? 0.3f : 1.0f);
        }
        dlco_ref_pr *p = dlco_ref_pr_create(D.data(), L.data(), N, F, 0.025f, 0.1f);
        CHECK(p != nullptr);
        dlco_ref_pr_steps(p, 500);
        float lo = 0.f, rg = 0.f;
        int nnz = 0;
        dlco_ref_pr_validate(p, &lo, &rg, &nnz);
        CHECK(std::isfinite(lo) && nnz >= 0 && nnz <= F);
        dlco_ref_pr_destroy(p);
        std::vector<uint8_t> patch(64 * 64);
        for (auto &px : patch) px = (uint8_t)(lcg(seed) & 0xff);
        std::vector<float> PT((size_t)4096 * 8);
        dlco_ref_get_desc(patch.data(), 8, 1.4f, 1, PT.data());
        for (float v : PT) CHECK(std::isfinite(v) && v >= 0.f);
    }
    // ---- I/O layer: .npy directory round trips, row streams with a ragged last block, bad files ---------------------------
    {
        const std::string dir = tmp + "/asan_io_dir";
        const size_t N = 301, F = 24;
        std::vector<float> D(N * F);
        std::vector<uint8_t> L(N);
        for (size_t i = 0; i < N * F; i++) D[i] = unif(seed);
        for (size_t i = 0; i < N; i++) L[i] = (uint8_t)(i & 1);
        {
            dlco_io::Writer w(dir);
            w.write<float>("Distance", D.data(), N, F);
            w.write<uint8_t>("Label", L.data(), N, 1);
        }
        std::vector<size_t> sh;
        std::vector<float> back;
        dlco_io::read_dataset<float>(dir, "Distance", sh, back);
        CHECK(sh.size() == 2 && sh[0] == N && sh[1] == F && back == D);
        std::vector<uint8_t> lb;
        dlco_io::read_dataset<uint8_t>(dir, "Label", sh, lb);
        CHECK(lb == L);
        {
            dlco_io::Writer w2(tmp + "/asan_io_stream");
            dlco_io::RowStream<float> rs(w2, "Distance", N, F, 128, 128, 9);
            for (size_t r0 = 0; r0 < N; r0 += 128) rs.write_rows(r0, std::min<size_t>(128, N - r0), D.data() + r0 * F);
        }
        dlco_io::read_dataset<float>(tmp + "/asan_io_stream", "Distance", sh, back);
        CHECK(back == D);
        bool threw = false;
        try { dlco_io::read_dataset<float>(dir, "NoSuchDataset", sh, back); } catch (const std::exception &) { threw = true; }
        CHECK(threw);
        threw = false;
        std::vector<int32_t> wrong;
        try { dlco_io::read_dataset<int32_t>(dir, "Distance", sh, wrong); } catch (const std::exception &) { threw = true; }
        CHECK(threw);                                           // wrong element type is refused, not reinterpreted
    }
    // ---- export format: filter selection and the sparse arrays ------------------------------------------------------------
    {
        const int rows = 12, cols = 64;
        std::vector<float> PR((size_t)rows * cols, 0.f), w(rows, 0.f);
        for (int r = 0; r < rows; r++) {
            w[r] = r % 3 == 0 ? 0.f : 0.1f * (float)r;
            for (int c = 0; c < cols; c++) PR[(size_t)r * cols + c] = (c / 8 == r % 8) ? 0.5f : 0.f;
        }
        int nsel = 0;
        const std::vector<float> sel = dlco_export::select_pr_filters(PR.data(), rows, cols, w.data(), rows, &nsel);
        CHECK(nsel > 0 && nsel <= rows && sel.size() == (size_t)nsel * cols);
        const auto runs = dlco_export::nonzero_runs(sel.data(), sel.size());
        CHECK(!runs.empty());
        FILE *f = std::fopen((tmp + "/asan_export.i").c_str(), "w");
        CHECK(f != nullptr);
        dlco_export::write_index_array(f, sel.data(), nsel, cols);
        dlco_export::write_value_array(f, sel.data(), nsel, cols);
        std::fclose(f);
    }
    std::printf("asan_main: all host checks passed\n");
    return 0;
}
