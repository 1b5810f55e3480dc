// pj-learn — command-line front end of libdlco.so with the reference's interface:
//
//     pj-learn src_h5_dist_file dst_h5_output_file [-mu f] [-gamma f] [-iters n] [-help]
//
// Same flags, defaults, usage text, exit codes and stdout grammar as the reference's main()
// (src/pj-learn.cpp:108-169,185-186,222-223,239-242,260-264,538-580): the reference's own
// scripts scrape these lines (workspace/08-pjlearn.sh:17, 09-pjstats.sh:28).  All compute is
// behind the C ABI of include/dlco.h; this file only parses, loads, prints and saves.
// Extra flags (defaults = the reference's hard-coded constants): -batch n, -logstep n,
// -seed n, -device n.
#include "../../include/dlco.h"
#include "dlco_io.hpp"

#include <chrono>
#include <cstdlib>
#include <iostream>

using std::cout;
using std::endl;

int main(int argc, char **argv)
{
    float mu = 0.001f, gamma = 0.500f;                 // src/pj-learn.cpp:89-90
    unsigned nIter = 50000, LogStep = 100, szBatch = 200;   // :91-93
    unsigned long long seed = 2215;                    // :225
    int device = 0;
    bool help = false;
    const char *src = nullptr, *dst = nullptr;

    for (int i = 1; i < argc; i++) {
        if (argv[i][0] == '-') {
            const bool has_val = i + 1 < argc;
            if (std::strcmp(argv[i], "-help") == 0) { help = true; continue; }
            if (std::strcmp(argv[i], "-mu") == 0 && has_val) { mu = (float)atof(argv[++i]); continue; }
            if (std::strcmp(argv[i], "-gamma") == 0 && has_val) { gamma = (float)atof(argv[++i]); continue; }
            if (std::strcmp(argv[i], "-iters") == 0 && has_val) { nIter = (unsigned)atoi(argv[++i]); continue; }
            if (std::strcmp(argv[i], "-batch") == 0 && has_val) { szBatch = (unsigned)atoi(argv[++i]); continue; }
            if (std::strcmp(argv[i], "-logstep") == 0 && has_val) { LogStep = (unsigned)atoi(argv[++i]); continue; }
            if (std::strcmp(argv[i], "-seed") == 0 && has_val) { seed = strtoull(argv[++i], nullptr, 10); continue; }
            if (std::strcmp(argv[i], "-device") == 0 && has_val) { device = atoi(argv[++i]); continue; }
            cout << "ERROR: Invalid " << argv[i] << " option." << endl;
            help = true;
        } else {
            if (!src) { src = argv[i]; continue; }
            if (!dst) { dst = argv[i]; continue; }
        }
    }
    if (!src || !dst) help = true;
    if (help) {
        // the reference's usage text, including its "pr-learn" name and stale defaults (:158-164)
        cout << endl;
        cout << "Usage: pr-learn  src_h5_dist_file dst_h5_output_file" << endl;
        cout << "       -mu <0.0-1.0, 0.025=default> " << endl;
        cout << "       -gamma <0.0-10.0, 0.10=default> " << endl;
        cout << "       -iters <0-N, 5000000=default> " << endl;
        cout << endl;
        return 1;
    }
    cout << "mu: " << mu << " gamma: " << gamma << " nIters: " << nIter << endl;

    try {
        std::vector<size_t> dshape, lshape;
        std::vector<float> dists;
        std::vector<uint8_t> labels;
        dlco_io::read_dataset<float>(src, "Distance", dshape, dists);
        dlco_io::read_dataset<uint8_t>(src, "Label", lshape, labels);
        if (dshape.size() != 2) throw std::runtime_error("Distance must be a 2-D dataset");
        const int nDists = (int)dshape[0], FeatDim = (int)dshape[1];
        if ((int)labels.size() < nDists) throw std::runtime_error("Label has fewer rows than Distance");
        cout << "Load Labels: " << nDists << endl;
        cout << "Load Distances: " << nDists << " x " << FeatDim << endl;
        int tick = -1;
        for (int i = 0; i < nDists; i += 128) tick = dlco_io::term_progress((double)i / (double)nDists, tick);
        dlco_io::term_progress(1.0, tick);

        dlco_cfg cfg;
        dlco_cfg_default(&cfg);
        cfg.F = FeatDim; cfg.N = nDists; cfg.B = (int)szBatch; cfg.mu = mu; cfg.gamma = gamma; cfg.seed = seed; cfg.device = device;
        dlco_ctx *ctx = nullptr;
        if (dlco_ctx_create(&ctx, &cfg) != DLCO_OK) throw std::runtime_error(dlco_last_error(nullptr));
        if (dlco_set_data(ctx, dists.data(), labels.data()) != DLCO_OK) throw std::runtime_error(dlco_last_error(ctx));
        dists.clear(); dists.shrink_to_fit();

        int32_t n_pos = 0, n_pos_trn = 0, n_neg = 0, n_neg_trn = 0;
        dlco_get_index(ctx, nullptr, &n_pos, &n_pos_trn, nullptr, &n_neg, &n_neg_trn);
        cout << "Positive samples #" << n_pos << endl;
        cout << "Negative samples #" << n_neg << endl;
        cout << "Positive train #" << n_pos_trn << endl;
        cout << "Negative train #" << n_neg_trn << endl;
        cout << "Positive valid #" << n_pos - n_pos_trn << endl;
        cout << "Negative valid #" << n_neg - n_neg_trn << endl;
        char name[256];
        int cc_major = 0, cc_minor = 0;
        dlco_device_name(ctx, name, sizeof(name), &cc_major, &cc_minor);
        cout << endl;
        cout << "Found GPU: " << name << endl;
        cout << "Compute Capability: " << cc_major << "." << cc_minor << endl;
        cout << endl;

        unsigned step = 0;
        auto train_start = std::chrono::steady_clock::now();
        for (unsigned t = 0; t <= nIter; t++) {
            if (dlco_step(ctx) != DLCO_OK) throw std::runtime_error(dlco_last_error(ctx));
            if (step == LogStep) {
                dlco_sync(ctx);
                const double ttime = std::chrono::duration<double>(std::chrono::steady_clock::now() - train_start).count();
                dlco_log_entry e;
                if (dlco_log_step(ctx, &e) != DLCO_OK) throw std::runtime_error(dlco_last_error(ctx));
                if (e.is_best) {
                    std::printf("Best: %i  Loss: %.6f Regul: %.6f Obj: %.6f (%.6f) Rank: %i (%i) Ttime: %.4f Vtime: %.4f\n", e.t,
                                e.loss_val, e.regul, e.obj, e.obj_best, e.rank, e.rank_best, ttime, e.vtime);
                    std::printf("Stat: Dim [%i] AUC: %.6f (%.6f) FPR95: %.2f (%.2f)%s\n", e.dim, e.auc, e.auc_best,
                                e.fpr95 * 100, e.fpr95_best * 100, e.saved ? " [saved]" : "");
                } else {
                    std::printf("Step: %i  Loss: %.6f Regul: %.6f Obj: %.6f (%.6f) Rank: %i (%i) Ttime: %.4f Vtime: %.4f\n", e.t,
                                e.loss_val, e.regul, e.obj, e.obj_best, e.rank, e.rank_best, ttime, e.vtime);
                }
                cout << std::flush;
                if (e.nonconv > 0)            // stdout is a wire format (scraped by the reference's scripts): warn on stderr
                    std::fprintf(stderr, "pj-learn: warning: %d of the last %u steps ended above the eigen tolerance (t = %u)\n",
                                 e.nonconv, LogStep, e.t);
                step = 0;
                train_start = std::chrono::steady_clock::now();
            }
            step++;
        }

        // src/pj-learn.cpp:592-597; nothing saved -> empty datasets, as the reference writes empty Mats
        int32_t r = 0;
        dlco_get_saved(ctx, nullptr, &r, nullptr);
        std::vector<float> W((size_t)r * FeatDim), A(r ? (size_t)FeatDim * FeatDim : 0);
        if (r) dlco_get_saved(ctx, W.data(), &r, A.data());
        dlco_io::Writer out(dst);
        out.write_f32("W", W.data(), (size_t)r, r ? (size_t)FeatDim : 0);
        out.write_f32("A", A.data(), r ? (size_t)FeatDim : 0, r ? (size_t)FeatDim : 0);
        dlco_ctx_destroy(ctx);
    } catch (const std::exception &e) {
        std::fprintf(stderr, "pj-learn: %s\n", e.what());
        return 2;
    }
    return 0;
}
