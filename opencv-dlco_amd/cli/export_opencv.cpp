// export-opencv — writes the "vgg_generated_XX.i" header that OpenCV's VGG descriptor compiles
// in, from the pooling-region filters, the learned pooling-region weights and the projection W
// that pj-learn saved.  Same argv, stdout and output bytes as the reference tool
// (src/export-opencv.cpp:57-394); inputs are HDF5 files, or directories of .npy files with the
// dataset names (PRFilters.npy, w.npy, W.npy) where HDF5 is absent.
#include "dlco_io.hpp"
#include "export_format.hpp"

#include <cstdlib>
#include <iostream>

using namespace dlco_io;

int main(int argc, char **argv)
{
    int widx = -1;
    bool help = false;
    const char *flt = nullptr, *prg = nullptr, *prj = nullptr, *outname = nullptr;
    if (argc < 1) std::exit(-argc);
    for (int i = 1; i < argc; i++) {
        if (argv[i][0] == '-') {
            if (std::strcmp(argv[i], "-help") == 0) { help = true; continue; }
            const bool has_val = i + 1 < argc;
            if (std::strcmp(argv[i], "-flt") == 0 && has_val) { flt = argv[++i]; continue; }
            if (std::strcmp(argv[i], "-prg") == 0 && has_val) { prg = argv[++i]; continue; }
            if (std::strcmp(argv[i], "-id") == 0 && has_val) { widx = std::atoi(argv[++i]); continue; }
            if (std::strcmp(argv[i], "-prj") == 0 && has_val) { prj = argv[++i]; continue; }
            std::cout << "ERROR: Invalid " << argv[i] << " option." << std::endl;
            help = true;
        } else {
            if (!outname) { outname = argv[i]; continue; }
            help = true;
        }
    }
    if (widx == -1 || !flt || !prg || !prj || !outname) help = true;
    if (help) {
        std::cout << std::endl;
        std::cout << "Usage: export-opencv -flt src_h5_filter_file" << std::endl;
        std::cout << "       -prg src_h5_prg_file -id src_h5_prg_matrix_rowid" << std::endl;
        std::cout << "       -prj src_h5_prj_file out_cc_headerfile" << std::endl;
        std::cout << std::endl;
        return 1;
    }
    try {
        std::vector<size_t> ps, ws, Ws;
        std::vector<float> PR, w_all, W;
        std::cout << "Load PRFilters:" << std::endl;
        read_dataset<float>(flt, "PRFilters", ps, PR);
        if (ps.size() != 3) throw std::runtime_error("PRFilters must have rank 3");
        const int n_filters = (int)ps[0], fsize = (int)(ps[1] * ps[2]);
        term_progress(1.0, -1);
        std::printf("Load Learnt Filters: [%s]#%i\n", prg, widx);
        read_dataset<float>(prg, "w", ws, w_all);
        if (ws.size() != 2 || widx < 0 || (size_t)widx >= ws[0]) throw std::runtime_error("w: row id out of range");
        const int nw = (int)ws[1];
        const float *w = w_all.data() + (size_t)widx * nw;
        std::printf("Load Learnt Projections: [%s]\n", prj);
        read_dataset<float>(prj, "W", Ws, W);
        if (Ws.size() != 2) throw std::runtime_error("W must have rank 2");
        if (nw * 8 != n_filters) throw std::runtime_error("w.cols * 8 != PRFilters.rows");        // CV_Assert, src/misc.cpp:86
        int n_sel = 0;
        const std::vector<float> sPR = dlco_export::select_pr_filters(PR.data(), n_filters, fsize, w, nw, &n_sel);
        std::printf("PRFilters: %i x %i [%i]\n", n_sel, fsize, n_sel * 8);
        std::printf("PJFilters: %i x [%i]\n", (int)Ws[0], (int)Ws[1]);
        if ((int)Ws[1] != n_sel * 8) {
            std::printf("ERROR: PJFilters [%i] not agree PRFilters [%i].\n", n_sel * 8, (int)Ws[1]);
            return 0;                                                                              // exit(0) in the reference
        }
        FILE *out = std::fopen(outname, "w");
        if (!out) throw std::runtime_error(std::string("cannot open ") + outname);
        dlco_export::write_vgg_header(out, prg, widx, prj, sPR.data(), n_sel, fsize, W.data(), (int)Ws[0], (int)Ws[1]);
        std::fclose(out);
    } catch (const std::exception &e) {
        std::fprintf(stderr, "export-opencv: %s\n", e.what());
        return 2;
    }
    return 0;
}
