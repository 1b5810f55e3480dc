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
//
// Multi-GPU (the reference hard-codes cuda::setDevice(0), src/pj-learn.cpp:267):
//     -gpus N [-devices a,b,...] [-comm rccl|host] [-dp shard|allreduce]
// One process per GPU.  The parent loads the input once, then forks N ranks BEFORE anything has
// touched the GPU (the children inherit the loaded rows copy-on-write); rank 0 creates the
// ncclUniqueId, the parent relays its 128 bytes to the other ranks over pipes, and every rank runs
// the same dlco_step loop on a column-sharded context (cfg.shard: the -batch rows per class are the
// GLOBAL batch, split over the ranks; no F x F exchange) whose all-gathers the library issues itself
// through RCCL (dlco_comm_init).  `-comm host` selects the library's shared-memory fallback
// (dlco_comm_init_host) for machines without librccl or for ranks that share a device.  `-dp allreduce`
// is the exchange BASELINE configs[3] words: every rank keeps the whole dual average (cfg.shard = 0), the
// library all-gathers the 2B distances and all-reduces the F x F partial gradients (ncclAllReduce) each
// step, and the update is replicated.  Rank 0 prints
// the log and writes the result; a rank that fails exits non-zero and the parent then stops the
// others and exits with 3.  No process re-executes itself.
#include "../../include/dlco.h"
#include "dlco_io.hpp"

#include <signal.h>
#include <sys/mman.h>
#include <sys/wait.h>
#include <unistd.h>

#include <chrono>
#include <cstdlib>
#include <iostream>

using std::cout;
using std::endl;

namespace {

struct Options {
    float mu = 0.001f, gamma = 0.500f;                      // src/pj-learn.cpp:89-90
    unsigned nIter = 50000, LogStep = 100, szBatch = 200;   // :91-93
    unsigned long long seed = 2215;                         // :225
    int device = 0;
    int gpus = 1;
    std::vector<int> devices;
    std::string comm = "rccl";
    std::string dp = "shard";                               // multi-GPU layout: column-sharded dual average | replicated + all-reduce
    const char *src = nullptr, *dst = nullptr;
};

bool read_exact(int fd, void *buf, size_t n)
{
    char *p = static_cast<char *>(buf);
    while (n > 0) {
        const ssize_t k = ::read(fd, p, n);
        if (k <= 0) return false;
        p += k; n -= (size_t)k;
    }
    return true;
}

bool write_exact(int fd, const void *buf, size_t n)
{
    const char *p = static_cast<const char *>(buf);
    while (n > 0) {
        const ssize_t k = ::write(fd, p, n);
        if (k <= 0) return false;
        p += k; n -= (size_t)k;
    }
    return true;
}

// The training run of one rank (world == 1: the reference's single process).  id_in / id_out: pipe
// ends for the 128-byte ncclUniqueId (rank 0 writes it to id_out, the others read it from id_in).
int train(const Options &o, std::vector<float> &dists, const std::vector<uint8_t> &labels, int nDists, int FeatDim, int rank,
          int world, int id_in, int id_out, const std::string &shm_name)
{
    const bool talk = rank == 0;
    try {
        dlco_cfg cfg;
        dlco_cfg_default(&cfg);
        cfg.F = FeatDim; cfg.N = nDists; cfg.B = (int)o.szBatch; cfg.mu = o.mu; cfg.gamma = o.gamma; cfg.seed = o.seed;
        cfg.device = world > 1 ? o.devices[rank] : o.device;
        cfg.rank = rank; cfg.world = world; cfg.shard = (world > 1 && o.dp == "shard") ? 1 : 0;
        dlco_ctx *ctx = nullptr;
        if (dlco_ctx_create(&ctx, &cfg) != DLCO_OK) throw std::runtime_error(dlco_last_error(nullptr));
        if (dlco_set_data(ctx, dists.data(), labels.data()) != DLCO_OK) throw std::runtime_error(dlco_last_error(ctx));
        dists.clear(); dists.shrink_to_fit();
        if (world > 1) {
            if (o.comm == "host") {
                if (dlco_comm_init_host(ctx, shm_name.c_str()) != DLCO_OK) throw std::runtime_error(dlco_last_error(ctx));
            } else {
                unsigned char id[128];
                if (rank == 0) {
                    if (dlco_comm_unique_id(id, sizeof(id), nullptr) != DLCO_OK) throw std::runtime_error(dlco_last_error(nullptr));
                    if (!write_exact(id_out, id, sizeof(id))) throw std::runtime_error("cannot hand the RCCL id to the parent");
                } else if (!read_exact(id_in, id, sizeof(id))) {
                    throw std::runtime_error("did not receive the RCCL id");
                }
                if (dlco_comm_init(ctx, id, sizeof(id), nullptr) != DLCO_OK) throw std::runtime_error(dlco_last_error(ctx));
            }
        }

        if (talk) {
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
        }

        unsigned step = 0;
        auto train_start = std::chrono::steady_clock::now();
        for (unsigned t = 0; t <= o.nIter; t++) {
            if (dlco_step(ctx) != DLCO_OK) throw std::runtime_error(dlco_last_error(ctx));
            if (step == o.LogStep) {
                // W is replicated on every rank: rank 0 alone evaluates, prints and keeps W_Save / A_Save;
                // the others run ahead into the next step's first all-gather
                if (talk) {
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
                                     e.nonconv, o.LogStep, e.t);
                }
                step = 0;
                train_start = std::chrono::steady_clock::now();
            }
            step++;
        }

        if (talk) {
            // src/pj-learn.cpp:592-597; nothing saved -> empty datasets, as the reference writes empty Mats
            int32_t r = 0;
            dlco_get_saved(ctx, nullptr, &r, nullptr);
            std::vector<float> W((size_t)r * FeatDim), A(r ? (size_t)FeatDim * FeatDim : 0);
            if (r) dlco_get_saved(ctx, W.data(), &r, A.data());
            dlco_io::Writer out(o.dst);
            out.write_f32("W", W.data(), (size_t)r, r ? (size_t)FeatDim : 0);
            out.write_f32("A", A.data(), r ? (size_t)FeatDim : 0, r ? (size_t)FeatDim : 0);
        }
        dlco_sync(ctx);
        dlco_ctx_destroy(ctx);
    } catch (const std::exception &e) {
        std::fprintf(stderr, "pj-learn%s: %s\n", world > 1 ? (" [rank " + std::to_string(rank) + "]").c_str() : "", e.what());
        return 2;
    }
    return 0;
}

}  // namespace

int main(int argc, char **argv)
{
    Options o;
    bool help = false;

    for (int i = 1; i < argc; i++) {
        if (argv[i][0] == '-') {
            const bool has_val = i + 1 < argc;
            if (std::strcmp(argv[i], "-help") == 0) { help = true; continue; }
            if (std::strcmp(argv[i], "-mu") == 0 && has_val) { o.mu = (float)atof(argv[++i]); continue; }
            if (std::strcmp(argv[i], "-gamma") == 0 && has_val) { o.gamma = (float)atof(argv[++i]); continue; }
            if (std::strcmp(argv[i], "-iters") == 0 && has_val) { o.nIter = (unsigned)atoi(argv[++i]); continue; }
            if (std::strcmp(argv[i], "-batch") == 0 && has_val) { o.szBatch = (unsigned)atoi(argv[++i]); continue; }
            if (std::strcmp(argv[i], "-logstep") == 0 && has_val) { o.LogStep = (unsigned)atoi(argv[++i]); continue; }
            if (std::strcmp(argv[i], "-seed") == 0 && has_val) { o.seed = strtoull(argv[++i], nullptr, 10); continue; }
            if (std::strcmp(argv[i], "-device") == 0 && has_val) { o.device = atoi(argv[++i]); continue; }
            if (std::strcmp(argv[i], "-gpus") == 0 && has_val) { o.gpus = atoi(argv[++i]); continue; }
            if (std::strcmp(argv[i], "-comm") == 0 && has_val) { o.comm = argv[++i]; continue; }
            if (std::strcmp(argv[i], "-dp") == 0 && has_val) { o.dp = argv[++i]; continue; }
            if (std::strcmp(argv[i], "-devices") == 0 && has_val) {
                for (const char *p = argv[++i]; *p;) {
                    o.devices.push_back(atoi(p));
                    while (*p && *p != ',') p++;
                    if (*p == ',') p++;
                }
                continue;
            }
            cout << "ERROR: Invalid " << argv[i] << " option." << endl;
            help = true;
        } else {
            if (!o.src) { o.src = argv[i]; continue; }
            if (!o.dst) { o.dst = argv[i]; continue; }
        }
    }
    if (!o.src || !o.dst) help = true;
    if (o.gpus < 1 || (o.comm != "rccl" && o.comm != "host") || (o.dp != "shard" && o.dp != "allreduce")) help = true;
    if (o.LogStep < 1) { cout << "ERROR: -logstep must be at least 1." << endl; help = true; }
    if (o.gpus > 1) {
        if (o.devices.empty()) for (int g = 0; g < o.gpus; g++) o.devices.push_back(g);
        if ((int)o.devices.size() != o.gpus || o.szBatch % (unsigned)o.gpus != 0) {
            cout << "ERROR: -gpus N needs N entries in -devices and a -batch divisible by N." << endl;
            help = true;
        }
    }
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
    cout << "mu: " << o.mu << " gamma: " << o.gamma << " nIters: " << o.nIter << endl;

    std::vector<float> dists;
    std::vector<uint8_t> labels;
    int nDists = 0, FeatDim = 0;
    try {
        std::vector<size_t> dshape, lshape;
        dlco_io::read_dataset<float>(o.src, "Distance", dshape, dists);
        dlco_io::read_dataset<uint8_t>(o.src, "Label", lshape, labels);
        if (dshape.size() != 2) throw std::runtime_error("Distance must be a 2-D dataset");
        nDists = (int)dshape[0]; FeatDim = (int)dshape[1];
        if ((int)labels.size() < nDists) throw std::runtime_error("Label has fewer rows than Distance");
        cout << "Load Labels: " << nDists << endl;
        cout << "Load Distances: " << nDists << " x " << FeatDim << endl;
        int tick = -1;
        for (int i = 0; i < nDists; i += 128) tick = dlco_io::term_progress((double)i / (double)nDists, tick);
        dlco_io::term_progress(1.0, tick);
    } catch (const std::exception &e) {
        std::fprintf(stderr, "pj-learn: %s\n", e.what());
        return 2;
    }
    if (o.gpus == 1) return train(o, dists, labels, nDists, FeatDim, 0, 1, -1, -1, "");

    // ---- one process per GPU: fork before anything has touched the device ------------------------------
    cout << std::flush;
    std::fflush(stdout);
    const int N = o.gpus;
    const std::string shm_name = "/dlco_pj_" + std::to_string((long)getpid());
    int up[2] = {-1, -1};                                      // rank 0 -> parent (the RCCL id)
    std::vector<int> down_r(N, -1), down_w(N, -1);             // parent -> rank g
    if (pipe(up) != 0) { std::perror("pj-learn: pipe"); return 2; }
    for (int g = 1; g < N; g++) {
        int p[2];
        if (pipe(p) != 0) { std::perror("pj-learn: pipe"); return 2; }
        down_r[g] = p[0]; down_w[g] = p[1];
    }
    std::vector<pid_t> kids(N, -1);
    for (int g = 0; g < N; g++) {
        const pid_t pid = fork();
        if (pid < 0) { std::perror("pj-learn: fork"); for (int k = 0; k < g; k++) kill(kids[k], SIGTERM); return 2; }
        if (pid == 0) {
            close(up[0]);
            for (int k = 1; k < N; k++) { close(down_w[k]); if (k != g) close(down_r[k]); }
            if (g != 0) close(up[1]);
            const int rc = train(o, dists, labels, nDists, FeatDim, g, N, g ? down_r[g] : -1, g ? -1 : up[1], shm_name);
            std::fflush(stdout);
            _exit(rc);                                          // no atexit handlers of the parent's state in a child
        }
        kids[g] = pid;
    }
    dists.clear(); dists.shrink_to_fit();
    close(up[1]);
    for (int g = 1; g < N; g++) close(down_r[g]);
    // a rank that died before reading (bad -devices entry, failed create) leaves a pipe without a reader: the relay's
    // write must fail with EPIPE (write_exact returns false), not kill the parent, which still has to stop the others
    signal(SIGPIPE, SIG_IGN);
    if (o.comm == "rccl") {                                    // relay the id; a rank that died simply closes its pipe
        unsigned char id[128];
        if (read_exact(up[0], id, sizeof(id)))
            for (int g = 1; g < N; g++) (void)write_exact(down_w[g], id, sizeof(id));
    }
    close(up[0]);
    for (int g = 1; g < N; g++) close(down_w[g]);
    int failed = 0, left = N;
    while (left > 0) {
        int status = 0;
        const pid_t pid = wait(&status);
        if (pid < 0) break;
        left--;
        const bool ok = WIFEXITED(status) && WEXITSTATUS(status) == 0;
        if (!ok && !failed) {                                   // the others may sit in a collective: stop them
            failed = 1;
            for (int g = 0; g < N; g++) if (kids[g] != pid) kill(kids[g], SIGTERM);
        }
    }
    if (o.comm == "host") shm_unlink(shm_name.c_str());         // normally gone already (the last rank to attach unlinks it)
    return failed ? 3 : 0;
}
