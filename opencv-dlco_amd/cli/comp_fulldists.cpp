// comp-fulldists — pooling-region distances for pr-learn over libdlco.so ("compute-pr").
//
// Command-line surface of the reference's tool (src/comp-fulldists.cpp:54-145):
//     comp-fulldists src_h5_filter_file src_h5_patches_file dst_h5_dist_file [-anglebins n] [-sigma f] [-norm 0|1]
// reads "PRFilters" [8*n_regions,64,64] f32 and "Indices" [pairs,4] i32 + "Patches" [n,64,64] u8, writes "Label"
// [pairs,1] u8 and "Distance" [pairs, n_regions] f32 (:285-369).  -anglebins must stay 8 (the library's
// transform is built for the 8 bins every caller of the reference uses).  Extra flag: -device N.
#include "../../include/dlco.h"
#include "dlco_io.hpp"

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <stdexcept>
#include <string>
#include <vector>

using std::cout;
using std::endl;

int main(int argc, char **argv)
{
    int nAngleBins = 8, bNorm = 1, device = 0;
    float InitSigma = 1.4f;
    bool help = false;
    const char *flt = nullptr, *img = nullptr, *dst = nullptr;
    const size_t sChunk = 128;                              // src/comp-fulldists.cpp:61
    for (int i = 1; i < argc; i++) {
        if (argv[i][0] == '-') {
            const bool has_val = i + 1 < argc;
            if (std::strcmp(argv[i], "-help") == 0) { help = true; continue; }
            if (std::strcmp(argv[i], "-anglebins") == 0 && has_val) { nAngleBins = atoi(argv[++i]); continue; }
            if (std::strcmp(argv[i], "-sigma") == 0 && has_val) { InitSigma = (float)atof(argv[++i]); continue; }
            if (std::strcmp(argv[i], "-norm") == 0 && has_val) { bNorm = atoi(argv[++i]); continue; }
            if (std::strcmp(argv[i], "-device") == 0 && has_val) { device = atoi(argv[++i]); continue; }
            cout << "ERROR: Invalid " << argv[i] << " option." << endl;
            help = true;
        } else {
            if (!flt) { flt = argv[i]; continue; }
            if (!img) { img = argv[i]; continue; }
            if (!dst) { dst = argv[i]; continue; }
            help = true;
        }
    }
    if (!flt || !img || !dst) help = true;
    if (help) {
        cout << endl;
        cout << "Usage: comp-fulldists src_h5_filter_file src_h5_patches_file dst_h5_dist_file" << endl;
        cout << "       -anglebins <0-255, 8=default> " << endl;
        cout << "       -sigma <0.0-10.0, 1.4=default> " << endl;
        cout << "       -norm <0,1 1=default> " << endl;
        cout << endl;
        return 1;
    }
    cout << "Bins: #" << nAngleBins << " Sigma: " << InitSigma << " bNorm: " << bNorm << endl;
    try {
        std::vector<size_t> fshape, ishape, pshape;
        std::vector<float> PRFilters;
        std::vector<int32_t> pairs;
        std::vector<uint8_t> patches;
        printf("ImageSet: [%s]\n", img);
        cout << "Load Indices." << endl;
        dlco_io::read_dataset<int32_t>(img, "Indices", ishape, pairs);
        if (ishape.size() != 2 || ishape[1] != 4) throw std::runtime_error("Indices must be [pairs,4]");
        cout << "Load Patches." << endl;
        dlco_io::read_dataset<uint8_t>(img, "Patches", pshape, patches);
        if (pshape.size() != 3 || pshape[1] != 64 || pshape[2] != 64) throw std::runtime_error("Patches must be [n,64,64]");
        dlco_io::term_progress(1.0, -1);
        cout << "Load PRFilters." << endl;
        dlco_io::read_dataset<float>(flt, "PRFilters", fshape, PRFilters);
        if (fshape.size() != 3 || fshape[1] * fshape[2] != 4096 || fshape[0] % 8 != 0) throw std::runtime_error("PRFilters must be [8*n,64,64]");
        dlco_io::term_progress(1.0, -1);
        const size_t npairs = ishape[0], regions = fshape[0] / 8;
        cout << "Export Pair Labels: #" << npairs << endl;
        dlco_io::term_progress(1.0, -1);

        dlco_desc_ctx *ctx = nullptr;
        if (dlco_desc_create(&ctx, InitSigma, nAngleBins, bNorm, device) != DLCO_OK) throw std::runtime_error(dlco_desc_last_error(nullptr));
        if (dlco_desc_set_filters(ctx, PRFilters.data(), (int32_t)fshape[0]) != DLCO_OK) throw std::runtime_error(dlco_desc_last_error(ctx));
        // one chunk of pairs at a time into hyperslabs of the chunked, deflate-9 datasets (src/comp-fulldists.cpp:
        // 270-283,300-369): host memory holds one chunk, and an interrupted run keeps the rows written so far
        struct Sink {
            dlco_io::RowStream<uint8_t> *lab; dlco_io::RowStream<float> *dst; size_t cols, total; bool nan = false; std::string err;
            static int put(void *u, int64_t row0, int64_t rows, const float *dist, const uint8_t *label)
            {
                Sink *k = static_cast<Sink *>(u);
                try {
                    for (size_t i = 0; i < (size_t)rows * k->cols; i++)
                        if (!(dist[i] == dist[i]) || dist[i] > 3.0e38f) { k->nan = true; return 1; }                          // checkRange per chunk, :361-365
                    k->lab->write_rows((size_t)row0, (size_t)rows, label);
                    k->dst->write_rows((size_t)row0, (size_t)rows, dist);
                } catch (const std::exception &e) { k->err = e.what(); return 2; }
                printf("\rStep: %zu / %zu", (size_t)(row0 + rows), k->total);
                fflush(stdout);
                return 0;
            }
        };
        const auto t0 = std::chrono::steady_clock::now();
        {
            dlco_io::Writer wr(dst);
            dlco_io::RowStream<uint8_t> ls(wr, "Label", npairs, 1, sChunk, 1, 9);
            dlco_io::RowStream<float> ds(wr, "Distance", npairs, regions, sChunk, 1, 9);
            Sink k{&ls, &ds, regions, npairs};
            const int rc = dlco_desc_full_dists_stream(ctx, patches.data(), (int64_t)pshape[0], pairs.data(), (int64_t)npairs, (int64_t)sChunk * 8,
                                                       &Sink::put, &k);
            if (k.nan) { cout << "\nDist contains NaN\n"; return 255; }
            if (!k.err.empty()) throw std::runtime_error(k.err);
            if (rc != DLCO_OK) throw std::runtime_error(dlco_desc_last_error(ctx));
        }
        const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        cout << "\nDone." << endl << endl;
        printf("Total: %.09f sec\n\n", sec);
        dlco_desc_destroy(ctx);
    } catch (const std::exception &e) {
        std::fprintf(stderr, "comp-fulldists: %s\n", e.what());
        return 2;
    }
    return 0;
}
