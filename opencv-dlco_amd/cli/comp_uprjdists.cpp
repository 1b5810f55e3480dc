// comp-uprjdists — descriptor generation for pj-learn over libdlco.so.
//
// Command-line surface of the reference's tool (src/comp-uprjdists.cpp:54-133):
//     comp-uprjdists src_h5_filter_file src_h5_img_file -prj src_h5_prj_file -id rowid -out dst_h5_filename
// reads "PRFilters" [n,64,64] f32, "Indices" [pairs,4] i32 + "Patches" [n,64,64] u8 and row `rowid` of "w",
// selects the pooling regions (SelectPRFilters) and writes "Label" [pairs,1] u8 and "Distance"
// [pairs, nsel*8] f32 (:254-349).  Each patch's descriptor is computed once on the GPU; the pair
// differences are formed there too.  Extra flag: -device N.  A path that does not end in .h5/.hdf5 is a
// directory of .npy files with the same dataset names (dlco_io.hpp).
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
    int widx = 0, device = 0;
    bool help = false;
    const char *flt = nullptr, *img = nullptr, *prj = nullptr, *out = nullptr;
    const size_t sChunk = 128;                              // src/comp-uprjdists.cpp:68
    for (int i = 1; i < argc; i++) {
        if (argv[i][0] == '-') {
            const bool has_val = i + 1 < argc;
            if (std::strcmp(argv[i], "-help") == 0) { help = true; continue; }
            if (std::strcmp(argv[i], "-prj") == 0 && has_val) { prj = argv[++i]; continue; }
            if (std::strcmp(argv[i], "-id") == 0 && has_val) { widx = atoi(argv[++i]); continue; }
            if (std::strcmp(argv[i], "-out") == 0 && has_val) { out = argv[++i]; continue; }
            if (std::strcmp(argv[i], "-device") == 0 && has_val) { device = atoi(argv[++i]); continue; }
            cout << "ERROR: Invalid " << argv[i] << " option." << endl;
            help = true;
        } else {
            if (!flt) { flt = argv[i]; continue; }
            if (!img) { img = argv[i]; continue; }
            help = true;
        }
    }
    if (!flt || !img || !out || !prj) help = true;
    if (help) {
        cout << endl;
        cout << "Usage: comp-uprjdists src_h5_filter_file src_h5_img_file" << endl;
        cout << "        -prj src_h5_prj_file -id src_h5_prj_matrix_rowid -out dsc_h5_filename" << endl;
        cout << endl;
        return 1;
    }
    try {
        std::vector<size_t> fshape, ishape, pshape, wshape;
        std::vector<float> PRFilters, w;
        std::vector<int32_t> pairs;
        std::vector<uint8_t> patches;
        cout << "Load PRFilters." << endl;
        dlco_io::read_dataset<float>(flt, "PRFilters", fshape, PRFilters);
        if (fshape.size() != 3 || fshape[1] * fshape[2] != 4096) throw std::runtime_error("PRFilters must be [n,64,64]");
        dlco_io::term_progress(1.0, -1);
        printf("ImageSet: [%s]\n", img);
        cout << "Load Indices." << endl;
        dlco_io::read_dataset<int32_t>(img, "Indices", ishape, pairs);
        if (ishape.size() != 2 || ishape[1] != 4) throw std::runtime_error("Indices must be [pairs,4]");
        cout << "Load Patches." << endl;
        dlco_io::read_dataset<uint8_t>(img, "Patches", pshape, patches);
        if (pshape.size() != 3 || pshape[1] != 64 || pshape[2] != 64) throw std::runtime_error("Patches must be [n,64,64]");
        dlco_io::term_progress(1.0, -1);
        printf("Load Learnt Filters: [%s]#%i\n", prj, widx);
        dlco_io::read_dataset<float>(prj, "w", wshape, w);
        if (wshape.size() != 2 || widx < 0 || (size_t)widx >= wshape[0]) throw std::runtime_error("w has no such row");
        const size_t wcols = wshape[1];
        if (wcols * 8 != fshape[0]) throw std::runtime_error("PRFilters needs 8 rows per column of w");
        const float *wrow = w.data() + (size_t)widx * wcols;

        int32_t nsel = 0;
        if (dlco_desc_select_filters(PRFilters.data(), (int32_t)fshape[0], 4096, wrow, (int32_t)wcols, nullptr, &nsel) != DLCO_OK)
            throw std::runtime_error("filter selection failed");
        if (nsel < 1) throw std::runtime_error("no pooling region selected (w has no positive entry on a non-zero filter)");
        std::vector<float> sPR((size_t)nsel * 4096);
        dlco_desc_select_filters(PRFilters.data(), (int32_t)fshape[0], 4096, wrow, (int32_t)wcols, sPR.data(), &nsel);
        printf("PRFilters: %i x %i\n", nsel, 4096);
        printf("Descriptor size: %i\n", nsel * 8);

        dlco_desc_ctx *ctx = nullptr;
        if (dlco_desc_create(&ctx, 1.4f, 8, 1, device) != DLCO_OK) throw std::runtime_error(dlco_desc_last_error(nullptr));
        if (dlco_desc_set_filters(ctx, sPR.data(), nsel) != DLCO_OK) throw std::runtime_error(dlco_desc_last_error(ctx));
        const size_t npairs = ishape[0], F = (size_t)nsel * 8;
        cout << "Export Pair Labels: #" << npairs << endl;
        dlco_io::term_progress(1.0, -1);
        cout << "Start Compute L1 distances." << endl;
        // rows leave the library a chunk of sChunk pairs at a time and go straight into hyperslabs of the two chunked,
        // deflate-9 datasets the reference creates (:254-256,289-290): host memory holds one chunk, not the matrix
        struct Sink {
            dlco_io::RowStream<uint8_t> *lab; dlco_io::RowStream<float> *dst; size_t cols, total; bool nan = false; std::string err;
            static int put(void *u, int64_t row0, int64_t rows, const float *dist, const uint8_t *label)
            {
                Sink *k = static_cast<Sink *>(u);
                try {
                    for (size_t i = 0; i < (size_t)rows * k->cols; i++)
                        if (!(dist[i] == dist[i]) || dist[i] > 3.0e38f || dist[i] < -3.0e38f) { k->nan = true; return 1; }   // checkRange per chunk, :341-345
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
            dlco_io::Writer wr(out);
            dlco_io::RowStream<uint8_t> ls(wr, "Label", npairs, 1, sChunk, 1, 9);
            dlco_io::RowStream<float> ds(wr, "Distance", npairs, F, sChunk, 1, 9);
            Sink k{&ls, &ds, F, npairs};
            const int rc = dlco_desc_pair_dists_stream(ctx, patches.data(), (int64_t)pshape[0], pairs.data(), (int64_t)npairs, (int64_t)sChunk * 64,
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
        std::fprintf(stderr, "comp-uprjdists: %s\n", e.what());
        return 2;
    }
    return 0;
}
