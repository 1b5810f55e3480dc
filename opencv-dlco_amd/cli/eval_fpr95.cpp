// eval-fpr95 — FPR@95 / AUC of a learnt projection on a (possibly different) pair set.
//
//     eval-fpr95 <W file: .h5 with dataset "W", or directory with W.npy> <unproj file/dir> [-device n]
//
// The reference has no such binary: the numbers are produced inside pj-learn by ComputePJStats
// (src/misc.cpp:266-333, called at src/pj-learn.cpp:551); its stand-alone pj-stats.cpp is a
// commented-out stub.  This tool exposes exactly that routine (dlco_stats) and prints the
// reference's "Stat:" grammar so that workspace/09-pjstats.sh-style scraping keeps working.
#include "../../include/dlco.h"
#include "dlco_io.hpp"

#include <cstdlib>
#include <iostream>

int main(int argc, char **argv)
{
    const char *wpath = nullptr, *dpath = nullptr;
    int device = 0;
    for (int i = 1; i < argc; i++) {
        if (std::strcmp(argv[i], "-device") == 0 && i + 1 < argc) { device = atoi(argv[++i]); continue; }
        if (argv[i][0] == '-') { std::cout << "ERROR: Invalid " << argv[i] << " option." << std::endl; wpath = nullptr; break; }
        if (!wpath) wpath = argv[i];
        else if (!dpath) dpath = argv[i];
    }
    if (!wpath || !dpath) {
        std::cout << std::endl << "Usage: eval-fpr95  w_h5_file src_h5_dist_file [-device n]" << std::endl << std::endl;
        return 1;
    }
    try {
        std::vector<size_t> ws, ds, ls;
        std::vector<float> W, dists;
        std::vector<uint8_t> labels;
        dlco_io::read_dataset<float>(wpath, "W", ws, W);
        dlco_io::read_dataset<float>(dpath, "Distance", ds, dists);
        dlco_io::read_dataset<uint8_t>(dpath, "Label", ls, labels);
        if (ws.size() != 2 || ds.size() != 2 || ws[1] != ds[1]) throw std::runtime_error("W and Distance must be 2-D with equal column counts");
        dlco_cfg cfg;
        dlco_cfg_default(&cfg);
        cfg.F = (int)ds[1]; cfg.N = (int)ds[0]; cfg.B = 1; cfg.device = device;
        dlco_ctx *ctx = nullptr;
        if (dlco_ctx_create(&ctx, &cfg) != DLCO_OK) throw std::runtime_error(dlco_last_error(nullptr));
        if (dlco_set_data(ctx, dists.data(), labels.data()) != DLCO_OK) throw std::runtime_error(dlco_last_error(ctx));
        int32_t dim = 0;
        float fpr95 = 0.f;
        double auc = 0.0;
        if (dlco_stats(ctx, W.data(), (int)ws[0], &dim, &fpr95, &auc) != DLCO_OK) throw std::runtime_error(dlco_last_error(ctx));
        std::printf("Stat: Dim [%i] AUC: %.6f (%.6f) FPR95: %.2f (%.2f)\n", dim, auc, auc, fpr95 * 100, fpr95 * 100);
        dlco_ctx_destroy(ctx);
    } catch (const std::exception &e) {
        std::fprintf(stderr, "eval-fpr95: %s\n", e.what());
        return 2;
    }
    return 0;
}
