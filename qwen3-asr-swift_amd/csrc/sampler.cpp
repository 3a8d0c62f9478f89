// sampler.cpp -- CPU-side token selection of the slow decoding path.
// Reference: Qwen3ASRModel.pickNextToken (Sources/Qwen3ASR/Qwen3ASR.swift:449-520): HF sign-aware repetition
// penalty over the set of generated ids, no-repeat n-gram mask, Gumbel-max temperature sampling, argmax with the
// first maximum winning (strict '>').  The reference pulls the logits to the CPU for exactly this.
#include "qasr.h"
#include <cmath>
#include <cstdint>
#include <limits>
#include <unordered_set>
#include <vector>

namespace qasr {

static inline uint64_t splitmix64(uint64_t& s) {
    uint64_t z = (s += 0x9e3779b97f4a7c15ull);
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}

int32_t pick_next_token(const float* logits, int32_t vocab, const int32_t* generated, int32_t n_gen,
                        float repetition_penalty, int32_t ngram, float temperature, uint64_t* rng_state) {
    if (vocab <= 0) return 0;
    if (repetition_penalty == 0.0f) repetition_penalty = 1.0f;       // zero-initialised options = defaults
    const bool fast = repetition_penalty == 1.0f && ngram == 0 && temperature == 0.0f;
    std::vector<float> scratch;
    const float* scores = logits;
    if (!fast) {
        scratch.assign(logits, logits + vocab);
        // :469-480 -- positive logits divide, negative multiply
        if (repetition_penalty > 1.0f && n_gen > 0) {
            std::unordered_set<int32_t> seen(generated, generated + n_gen);
            for (int32_t t : seen)
                if (t >= 0 && t < vocab) {
                    float v = scratch[t];
                    scratch[t] = v > 0.0f ? v / repetition_penalty : v * repetition_penalty;
                }
        }
        // :484-500 -- forbid the token that completed an earlier occurrence of the last (n-1)-gram
        if (ngram > 0 && n_gen >= ngram - 1 && n_gen >= ngram) {
            const int32_t* last = generated + n_gen - (ngram - 1);
            for (int32_t i = 0; i + ngram <= n_gen; ++i) {
                bool same = true;
                for (int32_t j = 0; j < ngram - 1; ++j)
                    if (generated[i + j] != last[j]) { same = false; break; }
                if (!same) continue;
                int32_t f = generated[i + ngram - 1];
                if (f >= 0 && f < vocab) scratch[f] = -std::numeric_limits<float>::infinity();
            }
        }
        // :504-510 -- argmax(logits / T + Gumbel(0,1)),  u in [1e-6, 1]
        if (temperature > 0.0f) {
            uint64_t local = 0x243f6a8885a308d3ull;
            uint64_t& st = rng_state ? *rng_state : local;
            for (int32_t i = 0; i < vocab; ++i) {
                const double r = (double)(splitmix64(st) >> 11) * (1.0 / 9007199254740992.0);   // [0,1)
                const float u = (float)(1e-6 + r * (1.0 - 1e-6));
                scratch[i] = scratch[i] / temperature - logf(-logf(u));
            }
        }
        scores = scratch.data();
    }
    int32_t best = 0;
    float best_s = -std::numeric_limits<float>::infinity();
    for (int32_t i = 0; i < vocab; ++i)
        if (scores[i] > best_s) { best_s = scores[i]; best = i; }
    return best;
}

}  // namespace qasr

extern "C" int32_t qasr_pick_next_token(const float* logits, int32_t vocab, const int32_t* generated, int32_t n_generated,
                                        float repetition_penalty, int32_t no_repeat_ngram_size, float temperature,
                                        uint64_t* rng_state) {
    if (!logits || vocab <= 0 || (n_generated > 0 && !generated)) return -1;
    return qasr::pick_next_token(logits, vocab, generated, n_generated < 0 ? 0 : n_generated, repetition_penalty,
                                 no_repeat_ngram_size, temperature, rng_state);
}
