// Test-only shim: exposes the product's host-side pair indexing (opencv-dlco_amd/csrc/pair_index.hpp,
// header-only, no HIP) through a C ABI so that the CPU test-suite can compare it bit for bit
// with the oracle without a GPU.  Built by tests/test_host_logic.py with g++.
#include "../../opencv-dlco_amd/csrc/pair_index.hpp"

#include <cstring>

extern "C" {

int shim_build_index(const uint8_t *labels, int N, int32_t *pos, int *n_pos, int *n_pos_trn, int32_t *neg, int *n_neg,
                     int *n_neg_trn)
{
    dlco::PairIndex idx;
    idx.build(labels, N);
    if (!idx.pos.empty()) std::memcpy(pos, idx.pos.data(), idx.pos.size() * sizeof(int32_t));
    if (!idx.neg.empty()) std::memcpy(neg, idx.neg.data(), idx.neg.size() * sizeof(int32_t));
    *n_pos = (int)idx.pos.size(); *n_neg = (int)idx.neg.size();
    *n_pos_trn = idx.n_pos_trn; *n_neg_trn = idx.n_neg_trn;
    return 0;
}

// `steps` batches of B interleaved (iPos, iNeg) draws, src/pj-learn.cpp:310-314
void shim_sample(uint64_t seed, int n_pos_trn, int n_neg_trn, int B, int steps, int32_t *ipos, int32_t *ineg)
{
    dlco::CvRng rng(seed);
    for (int s = 0; s < steps; s++)
        for (int k = 0; k < B; k++) {
            ipos[s * B + k] = rng.uniform(0, n_pos_trn);
            ineg[s * B + k] = rng.uniform(0, n_neg_trn);
        }
}

int shim_split(uint64_t n) { return dlco::PairIndex::split((size_t)n); }

void shim_rda_coeffs(uint32_t B, uint32_t t, float *w_dloss, float *w_dfavg) { dlco::rda_coeffs(B, t, w_dloss, w_dfavg); }

uint32_t shim_rng_next(uint64_t *state)
{
    dlco::CvRng r(*state);
    const uint32_t v = r.next();
    *state = r.state;
    return v;
}

}
