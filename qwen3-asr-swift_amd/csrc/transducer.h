// transducer.h -- host logic of the Parakeet-TDT / Nemotron / Parakeet-EOU models (BASELINE configs[4]): the greedy loops that drive the
// prediction network and the joint, vocabulary decode, and the streaming session's chunk cutting.  The networks themselves are opaque
// CoreML bundles in the reference (`encoder.mlmodelc`, `decoder.mlmodelc`, `joint.mlmodelc`) and are supplied by the caller as callbacks.
//
// Reference: Sources/ParakeetASR/TDTGreedyDecoder.swift:45-205, Sources/NemotronStreamingASR/RNNTGreedyDecoder.swift:38-129,
// Sources/ParakeetStreamingASR/RNNTGreedyDecoder.swift:58-170, Sources/ParakeetASR/Vocabulary.swift:42-96,
// Sources/NemotronStreamingASR/Vocabulary.swift:31-77, Sources/NemotronStreamingASR/StreamingSession.swift:110-158.
#pragma once
#include "qasr.h"
#include <map>
#include <string>
#include <vector>

namespace qasr {

int argmax_first(const float* v, int n);                        // vDSP_maxvi / the scalar `>` scan: first maximum
float log_softmax_at(const float* logits, int n, int id);      // logit[id] - (log(sum exp(l - max)) + max), Float32
float transducer_confidence(const float* log_probs, int n);    // min(1, exp(mean)), 0 for n == 0

struct TransducerResult {
    std::vector<int32_t> tokens;
    std::vector<float> log_probs;
    bool eou = false;
};
// throws std::runtime_error when a callback returns non-zero
TransducerResult tdt_greedy(const qasr_transducer_config& c, const qasr_transducer_callbacks& cb, int encoded_length);
TransducerResult rnnt_greedy(const qasr_transducer_config& c, const qasr_transducer_callbacks& cb, int encoded_length, int frame_offset);

struct SpVocab {
    std::map<int32_t, std::string> table;
    int style = 0;                                              // 0: ParakeetVocabulary, 1: Nemotron / EOU vocabulary
    std::string decode(const int32_t* ids, int n) const;
    // words + confidences; mismatched lengths: style 0 -> one word (the whole text) with confidence 0, style 1 -> nothing
    void decode_words(const int32_t* ids, int n_ids, const float* log_probs, int n_lp, std::vector<std::string>& words,
                      std::vector<float>& conf) const;
    static SpVocab load_json(const std::string& path, int style);
};

// StreamingSession.pushAudio / finalize sample bookkeeping: chunks of `samples_per_chunk`, the buffer advances by `shift`
struct StreamChunker {
    int samples_per_chunk = 0, shift = 0;
    std::vector<float> buf;
    void push(const float* s, size_t n) { buf.insert(buf.end(), s, s + n); }
    bool pop(float* chunk);                                     // true: one chunk cut
    bool flush(float* chunk);                                   // true: the zero-padded remainder (buffer emptied)
};

}  // namespace qasr
