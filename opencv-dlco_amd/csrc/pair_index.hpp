// pair_index.hpp — host-side pair indexing of pj-learn (R1-R3): the integer part of the
// path that must be bit-exact.  Follows src/pj-learn.cpp:214-237,310-314 and the OpenCV
// routines those lines call (cv::RNG is a multiply-with-carry generator; cv::randShuffle
// without an RNG argument draws from the thread-default generator whose state starts at
// 0xFFFFFFFF, not from the `rng(2215)` object the reference declares next to it).
#pragma once

#include <cstddef>
#include <cstdint>
#include <vector>

namespace dlco {

struct CvRng {
    uint64_t state;
    explicit CvRng(uint64_t s = 0xffffffffULL) : state(s ? s : 0xffffffffULL) {}
    // RNG::next()
    uint32_t next()
    {
        state = (uint64_t)(uint32_t)state * 4164903690U + (uint32_t)(state >> 32);
        return (uint32_t)state;
    }
    // RNG::uniform(int a, int b)
    int uniform(int a, int b) { return a == b ? a : (int)(next() % (uint32_t)(b - a) + (uint32_t)a); }
};

// cv::randShuffle on a continuous 1-D array
inline void cv_rand_shuffle(std::vector<int32_t> &v, CvRng &rng)
{
    const uint32_t n = (uint32_t)v.size();
    for (uint32_t i = 0; i < n; i++) {
        const uint32_t j = rng.next() % n;
        const int32_t t = v[j]; v[j] = v[i]; v[i] = t;
    }
}

struct PairIndex {
    std::vector<int32_t> pos, neg;     // shuffled row ids of label 1 / label 0
    int n_pos_trn = 0, n_neg_trn = 0;  // first n_*_trn entries train, the rest validate

    void build(const uint8_t *labels, int N)
    {
        pos.clear(); neg.clear();
        for (int i = 0; i < N; i++) {
            if (labels[i] == 1) pos.push_back(i);
            if (labels[i] == 0) neg.push_back(i);
        }
        CvRng the_rng;                               // theRNG(): default state
        if (!pos.empty()) cv_rand_shuffle(pos, the_rng);
        if (!neg.empty()) cv_rand_shuffle(neg, the_rng);
        n_pos_trn = split(pos.size());
        n_neg_trn = split(neg.size());
    }
    // size_t nTrn = size() * nDiv with `const float nDiv = 0.80` (float multiply, truncation)
    static int split(size_t n)
    {
        const float nDiv = 0.80f;
        volatile float prod = (float)n * nDiv;
        return (int)(size_t)prod;
    }
};

// U1 coefficients (src/pj-learn.cpp:422): addWeighted(dfAvg, (double)t/(t+1), dLoss,
// 1.0f/(szBatch*szBatch*(t+1)), 0, dfAvg).  The reference forms the denominator in 32-bit
// unsigned arithmetic; with its B = 200 and t <= 50000 that never wraps.  Here B is the GLOBAL
// batch (1600 on 8 GPUs) and would wrap at t = 1677, so the product is formed in 64 bits:
// identical bits wherever the reference itself does not wrap.
inline void rda_coeffs(uint32_t B, uint32_t t, float *w_dloss, float *w_dfavg)
{
    *w_dfavg = (float)((double)t / ((double)t + 1.0));
    const uint64_t den = (uint64_t)B * (uint64_t)B * ((uint64_t)t + 1u);
    *w_dloss = 1.0f / (float)den;
}

}  // namespace dlco
