// pr-learn — command-line front end of the pooling-region stage with the reference's interface:
//
//     pr-learn src_h5_filter_file src_h5_dist_file dst_h5_output_file [-mu f] [-gamma f] [-maxdim n] [-iters n] [-help]
//
// Same flags, defaults, usage text, exit codes and stdout grammar as the reference's main()
// (src/pr-learn.cpp:78-176,200-262,282-287,366-403; the logs are scraped by workspace/05-prstats.sh).
// All compute is behind the dlco_pr_* block of include/dlco.h; this file parses, loads, prints and
// appends the saved weight vectors to the "w" dataset of the output file (:385-400).  Extra flags
// (defaults = the reference's constants): -logstep n, -seed n, -device n.
#include "../../include/dlco.h"
#include "dlco_io.hpp"

#include <cfloat>
#include <chrono>
#include <cstdlib>
#include <iostream>

using std::cout;
using std::endl;

int main(int argc, char **argv)
{
    float mu = 0.025f, gamma = 0.10f;                        // src/pr-learn.cpp:78-79
    int MaxDim = 640;                                        // :80
    unsigned nIter = 5000000, LogStep = 100000;              // :81-82
    unsigned long long seed = 2215;                          // :241
    int device = 0;
    bool help = false;
    const char *flt = nullptr, *src = nullptr, *dst = nullptr;

    for (int i = 1; i < argc; i++) {
        if (argv[i][0] == '-') {
            const bool has_val = i + 1 < argc;
            if (std::strcmp(argv[i], "-help") == 0) { help = true; continue; }
            if (std::strcmp(argv[i], "-mu") == 0 && has_val) { mu = (float)atof(argv[++i]); continue; }
            if (std::strcmp(argv[i], "-gamma") == 0 && has_val) { gamma = (float)atof(argv[++i]); continue; }
            if (std::strcmp(argv[i], "-maxdim") == 0 && has_val) { MaxDim = atoi(argv[++i]); continue; }
            if (std::strcmp(argv[i], "-iters") == 0 && has_val) { nIter = (unsigned)atoi(argv[++i]); continue; }
            if (std::strcmp(argv[i], "-logstep") == 0 && has_val) { LogStep = (unsigned)atoi(argv[++i]); continue; }
            if (std::strcmp(argv[i], "-seed") == 0 && has_val) { seed = strtoull(argv[++i], nullptr, 10); continue; }
            if (std::strcmp(argv[i], "-device") == 0 && has_val) { device = atoi(argv[++i]); continue; }
            cout << "ERROR: Invalid " << argv[i] << " option." << endl;
            help = true;
        } else {
            if (!flt) { flt = argv[i]; continue; }
            if (!src) { src = argv[i]; continue; }
            if (!dst) { dst = argv[i]; continue; }
        }
    }
    if (!flt || !src || !dst) help = true;
    if (LogStep < 1) { cout << "ERROR: -logstep must be at least 1." << endl; help = true; }
    if (help) {
        cout << endl;
        cout << "Usage: pr-learn  src_h5_filter_file" << endl;
        cout << "       src_h5_dist_file dst_h5_output_file" << endl;
        cout << "       -mu <0.0-1.0, 0.025=default> " << endl;
        cout << "       -gamma <0.0-10.0, 0.10=default> " << endl;
        cout << "       -maxdim <64-768, 640=default> " << endl;
        cout << "       -iters <0-N, 5000000=default> " << endl;
        cout << endl;
        return 1;
    }
    cout << "mu: " << mu << " gamma: " << gamma << " maxdim: " << MaxDim << " nIters: " << nIter << endl;

    try {
        std::vector<size_t> pshape, rshape, dshape, lshape;
        std::vector<float> PRParams, RingParams, dists;
        std::vector<uint8_t> labels;
        cout << "Load PRParams." << endl;
        dlco_io::read_dataset<float>(flt, "PRParams", pshape, PRParams);
        cout << "Load RingParams." << endl;
        dlco_io::read_dataset<float>(flt, "RingParams", rshape, RingParams);      // read like the reference; not used by this stage
        dlco_io::read_dataset<float>(src, "Distance", dshape, dists);
        dlco_io::read_dataset<uint8_t>(src, "Label", lshape, labels);
        if (dshape.size() != 2 || pshape.size() != 2) throw std::runtime_error("Distance and PRParams must be 2-D datasets");
        const int nDists = (int)dshape[0], FeatDim = (int)dshape[1];
        if ((int)labels.size() < nDists) throw std::runtime_error("Label has fewer rows than Distance");
        if ((int)pshape[0] < 8 * FeatDim) throw std::runtime_error("PRParams needs 8 rows per column of Distance");
        cout << "Load Labels: " << nDists << endl;
        cout << "Load Distances: " << nDists << " x " << FeatDim << endl;
        int tick = -1;
        for (int i = 0; i < nDists; i += 128) tick = dlco_io::term_progress((double)i / (double)nDists, tick);
        dlco_io::term_progress(1.0, tick);

        dlco_pr_ctx *ctx = nullptr;
        if (dlco_pr_create(&ctx, FeatDim, nDists, mu, gamma, seed, device) != DLCO_OK) throw std::runtime_error(dlco_pr_last_error(nullptr));
        if (dlco_pr_set_data(ctx, dists.data(), labels.data()) != DLCO_OK) throw std::runtime_error(dlco_pr_last_error(ctx));
        dists.clear(); dists.shrink_to_fit();
        int32_t n_pos = 0, n_pos_trn = 0, n_neg = 0, n_neg_trn = 0;
        dlco_pr_get_index(ctx, &n_pos, &n_pos_trn, &n_neg, &n_neg_trn);
        cout << "Positive samples #" << n_pos << endl;
        cout << "Negative samples #" << n_neg << endl;
        cout << "Positive train #" << n_pos_trn << endl;
        cout << "Negative train #" << n_neg_trn << endl;
        cout << "Positive valid #" << n_pos - n_pos_trn << endl;
        cout << "Negative valid #" << n_neg - n_neg_trn << endl;
        char name[256];
        int cc_major = 0, cc_minor = 0;
        dlco_pr_device_name(ctx, name, sizeof(name), &cc_major, &cc_minor);
        cout << endl;
        cout << "Found GPU: " << name << endl;
        cout << "Compute Capability: " << cc_major << "." << cc_minor << endl;
        cout << endl;

        // The reference logs inside iteration t when its step counter has reached LogStep, i.e. at
        // t = LogStep, 2 LogStep + ... with `step` restarting from 0 -> every LogStep iterations, the
        // first time after LogStep + 1 of them (src/pr-learn.cpp:331,419-422); t runs to nIter inclusive.
        float Obj_Best = FLT_MAX;
        int nnz_best = 0;
        std::vector<float> w(FeatDim), w_Best(FeatDim, 0.f);
        unsigned t_done = 0;                                   // iterations run so far (t = 0 .. t_done-1)
        unsigned long long next_log = LogStep;                 // t of the next log line: LogStep, 2 LogStep, ... (every t >= 1 for LogStep = 1)
        auto train_start = std::chrono::steady_clock::now();
        while (t_done <= nIter) {
            // run up to and including the next logging iteration
            const unsigned upto = next_log <= nIter ? (unsigned)next_log + 1 : nIter + 1; // iterations t < upto
            if (dlco_pr_steps(ctx, upto - t_done) != DLCO_OK) throw std::runtime_error(dlco_pr_last_error(ctx));
            t_done = upto;
            if (next_log > nIter) break;
            uint32_t tt = 0;
            dlco_pr_get_state(ctx, &tt, w.data(), nullptr);    // also synchronises
            const double ttime = std::chrono::duration<double>(std::chrono::steady_clock::now() - train_start).count();
            const auto v0 = std::chrono::steady_clock::now();
            float LossVal = 0.f, Regul = 0.f;
            int32_t nnz = 0;
            if (dlco_pr_validate(ctx, &LossVal, &Regul, &nnz) != DLCO_OK) throw std::runtime_error(dlco_pr_last_error(ctx));
            const double vtime = std::chrono::duration<double>(std::chrono::steady_clock::now() - v0).count();
            const unsigned t = (unsigned)next_log;
            if ((LossVal + Regul) < Obj_Best) {                // :364
                Obj_Best = LossVal + Regul;
                w_Best = w;
                nnz_best = nnz;
                std::printf("Best: %i  Loss: %.6f Regul: %.6f Obj: %.6f (%.6f)  NNZ: %i (%i)  Ttime: %.4f Vtime: %.4f\n", t, LossVal, Regul,
                            (LossVal + Regul), Obj_Best, nnz, nnz_best, ttime, vtime);
                int32_t nPR = 0, Dim = 0, nzDim = 0;
                float FPR95 = -1.f;
                double AUC = 0.0;
                if (dlco_pr_stats(ctx, w.data(), PRParams.data(), (int32_t)pshape[0], (int32_t)pshape[1], 8, -1, &nPR, &Dim, &nzDim, &FPR95,
                                  &AUC) != DLCO_OK)
                    throw std::runtime_error(dlco_pr_last_error(ctx));
                if (Dim <= MaxDim) {                           // :385-400
                    dlco_io::append_row_f32(dst, "w", w_Best.data(), (size_t)FeatDim);
                    std::printf("Stat: nPR #%i (#%i) Dim/MaxDim [%i/%i] AUC: %.6f FPR95: %.2f [saved]\n", nPR, nzDim, Dim, MaxDim, AUC, FPR95 * 100);
                } else {
                    std::printf("Stat: nPR #%i (#%i) Dim/MaxDim [%i/%i] AUC: %.6f FPR95: %.2f\n", nPR, nzDim, Dim, MaxDim, AUC, FPR95 * 100);
                }
            } else {
                std::printf("Step: %i  Loss: %.6f Regul: %.6f Obj: %.6f (%.6f)  NNZ: %i (%i)  Ttime: %.4f Vtime: %.4f\n", t, LossVal, Regul,
                            (LossVal + Regul), Obj_Best, nnz, nnz_best, ttime, vtime);
            }
            cout << std::flush;
            next_log += LogStep;
            train_start = std::chrono::steady_clock::now();
        }
        dlco_pr_destroy(ctx);
    } catch (const std::exception &e) {
        std::fprintf(stderr, "pr-learn: %s\n", e.what());
        return 2;
    }
    return 0;
}
