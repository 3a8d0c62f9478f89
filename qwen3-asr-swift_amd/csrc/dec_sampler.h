// dec_sampler.h -- pickNextToken of the non-default decoding options as a device kernel (dec_sampler.hip).
#pragma once
#include "dec_kernels.h"

namespace qasr {

// Device-side pickNextToken (Qwen3ASR.swift:449-520) for the non-default decoding options: one workgroup per batch row edits the
// row's f32 logits in place -- HF sign-aware repetition penalty over the distinct generated ids, then the no-repeat-n-gram bans --
// and scans it (with temperature > 0: logits / T + Gumbel noise from the same counter-based splitmix64 stream as csrc/sampler.cpp,
// row b, call lens[b] * V + i) for the first maximum.  Writes one (value, index) partial per row for greedy_finalize (n_parts = 1).
void sampler_pick_launch(float* logits, int V, GreedyState st, int B, float repetition_penalty, int ngram, float temperature,
                         unsigned long long seed, float* part_val, int* part_idx, hipStream_t s);

}  // namespace qasr
