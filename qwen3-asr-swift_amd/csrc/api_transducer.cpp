// api_transducer.cpp -- extern "C" boundary of the Parakeet / Nemotron slice (include/qasr.h, last section).  Exceptions never cross it.
#include "nemo_mel.h"
#include "transducer.h"
#include <cstring>
#include <memory>
#include <string>
#include <vector>

struct qasr_nemo_mel {
    std::unique_ptr<qasr::NemoMel> impl;
    std::string last_error;
};
struct qasr_sp_vocab { qasr::SpVocab v; };
struct qasr_stream_chunker { qasr::StreamChunker c; };

static thread_local std::string g_nemo_create_error;

static int mfail(qasr_nemo_mel* m, int code, const std::string& msg) {
    if (code == QASR_ERR_HIP) (void)hipGetLastError();
    if (m) m->last_error = msg; else g_nemo_create_error = msg;
    return code;
}
#define NEMO_GUARD(m, body)                                                                   \
    try { body; return QASR_OK; }                                                             \
    catch (const qasr::HipError& ex) { return mfail(m, QASR_ERR_HIP, ex.what()); }            \
    catch (const std::length_error& ex) { return mfail(m, QASR_ERR_CAPACITY, ex.what()); }    \
    catch (const std::exception& ex) { return mfail(m, QASR_ERR_INVALID, ex.what()); }

static bool has(const std::string& s, const char* sub) { return s.find(sub) != std::string::npos; }

static void nemo_extract(qasr_nemo_mel* m, int variant, const float* const* pcm, const size_t* n, size_t B, const int32_t* stream_ids,
                         float* out, size_t stride, int32_t* mel_len, int fit) {
    // zero-length rows answer like the reference's `guard !audio.isEmpty` (zeros, melLength 0); the TDT extractor has no such guard
    // (it reads audio[0]) and refuses them
    std::vector<const float*> p2;
    std::vector<size_t> n2;
    std::vector<int32_t> sid2, rows;
    for (size_t b = 0; b < B; ++b) {
        if (n[b] == 0) {
            if (variant == QASR_NEMO_MEL_TDT) throw std::invalid_argument("nemo mel: empty clip (MelPreprocessor.extract reads audio[0])");
            if (mel_len) mel_len[b] = 0;
            continue;
        }
        if (!pcm[b]) throw std::invalid_argument("nemo mel: null clip");
        p2.push_back(pcm[b]); n2.push_back(n[b]); rows.push_back((int32_t)b);
        sid2.push_back(stream_ids ? stream_ids[b] : (int32_t)b);
    }
    if (rows.size() == B) {
        m->impl->extract(variant, pcm, n, B, stream_ids, out, stride, mel_len, fit);
    } else {
        int maxf = fit;
        if (maxf <= 0) { maxf = 1; for (size_t v : n2) maxf = std::max(maxf, qasr::nemo_num_frames((long)v)); }
        if (stride < (size_t)maxf) throw std::invalid_argument("nemo mel: stride smaller than the frame count");
        std::memset(out, 0, B * qasr::NEMO_NMELS * stride * sizeof(float));
        if (!rows.empty()) {
            std::vector<float> tmp(rows.size() * qasr::NEMO_NMELS * stride);
            std::vector<int32_t> ml(rows.size());
            m->impl->extract(variant, p2.data(), n2.data(), rows.size(), sid2.data(), tmp.data(), stride, ml.data(), maxf);
            for (size_t i = 0; i < rows.size(); ++i) {
                std::memcpy(out + (size_t)rows[i] * qasr::NEMO_NMELS * stride, tmp.data() + i * qasr::NEMO_NMELS * stride,
                            qasr::NEMO_NMELS * stride * sizeof(float));
                if (mel_len) mel_len[rows[i]] = ml[i];
            }
        }
    }
}

extern "C" {

int qasr_nemo_mel_create(int device, int max_streams, size_t max_samples, float fft_scale, qasr_nemo_mel** out) {
    if (!out) return QASR_ERR_INVALID;
    *out = nullptr;
    auto* m = new qasr_nemo_mel();
    try { m->impl = std::make_unique<qasr::NemoMel>(device, max_streams, (long)max_samples, fft_scale); }
    catch (const qasr::HipError& ex) { g_nemo_create_error = ex.what(); delete m; (void)hipGetLastError(); return QASR_ERR_HIP; }
    catch (const std::exception& ex) { g_nemo_create_error = ex.what(); delete m; return QASR_ERR_INVALID; }
    *out = m;
    return QASR_OK;
}
void qasr_nemo_mel_destroy(qasr_nemo_mel* m) { delete m; }
const char* qasr_nemo_mel_last_error(const qasr_nemo_mel* m) { return m ? m->last_error.c_str() : g_nemo_create_error.c_str(); }
int qasr_nemo_mel_num_frames(size_t n) { return qasr::nemo_num_frames((long)n); }
int qasr_nemo_mel_length(size_t n) { return qasr::nemo_mel_length((long)n); }

int qasr_nemo_mel_extract(qasr_nemo_mel* m, int variant, const float* const* pcm, const size_t* n, size_t B, const int32_t* stream_ids,
                          float* out, size_t stride, int32_t* mel_len, int fit) {
    if (!m || !m->impl) return QASR_ERR_INVALID;
    if (B == 0) return QASR_OK;
    if (!pcm || !n || !out) return mfail(m, QASR_ERR_INVALID, "nemo mel: null argument");
    NEMO_GUARD(m, nemo_extract(m, variant, pcm, n, B, stream_ids, out, stride, mel_len, fit));
}

int qasr_nemo_mel_reset_stats(qasr_nemo_mel* m, int stream) {
    if (!m || !m->impl) return QASR_ERR_INVALID;
    NEMO_GUARD(m, m->impl->reset_stats(stream));
}

int qasr_nemo_mel_timing(const qasr_nemo_mel* m, float* ms, int* was_graph) {
    if (!m || !m->impl) return QASR_ERR_INVALID;
    if (ms) *ms = m->impl->last_ms();
    if (was_graph) *was_graph = m->impl->last_was_graph() ? 1 : 0;
    return QASR_OK;
}

// ---- transducer loops -------------------------------------------------------------------------------------------------------
int qasr_transducer_default_config(const char* model, qasr_transducer_config* c) {
    if (!model || !c) return QASR_ERR_INVALID;
    std::string s = model;
    for (auto& ch : s) ch = (char)std::tolower((unsigned char)ch);
    std::memset(c, 0, sizeof(*c));
    c->eou_id = -1;
    c->max_symbols = 10;
    if (has(s, "nemotron")) {                                  // NemotronStreamingConfig.default
        c->vocab_size = 1024; c->blank_id = 1024;
    } else if (has(s, "eou")) {                                // ParakeetEOUConfig.default
        c->vocab_size = 1026; c->blank_id = 1026; c->eou_id = 1024;
    } else if (has(s, "tdt") || has(s, "parakeet")) {          // ParakeetConfig.default (Parakeet-TDT 0.6B v3)
        c->vocab_size = 8192; c->blank_id = 8192; c->n_durations = 5; c->first_text_id = 274;
        for (int i = 0; i < 5; ++i) c->durations[i] = i;
    } else return QASR_ERR_INVALID;
    return QASR_OK;
}

static bool cfg_ok(const qasr_transducer_config* c, const qasr_transducer_callbacks* cb, bool tdt) {
    if (!c || !cb || !cb->decoder_step || !cb->joint) return false;
    if (c->vocab_size <= 0 || c->vocab_size > (1 << 24) || c->blank_id < 0 || c->blank_id > c->vocab_size) return false;
    if (tdt && (c->n_durations <= 0 || c->n_durations > 8)) return false;
    if (!tdt && c->max_symbols <= 0) return false;
    return true;
}

static int emit(const qasr::TransducerResult& r, int32_t* tokens, float* log_probs, int32_t cap) {
    if ((int64_t)r.tokens.size() > cap) return -QASR_ERR_CAPACITY;
    for (size_t i = 0; i < r.tokens.size(); ++i) {
        tokens[i] = r.tokens[i];
        if (log_probs) log_probs[i] = r.log_probs[i];
    }
    return (int)r.tokens.size();
}

int qasr_tdt_greedy_decode(const qasr_transducer_config* cfg, const qasr_transducer_callbacks* cb, int32_t encoded_length, int32_t* tokens,
                           float* log_probs, int32_t cap, float* confidence) {
    if (!cfg_ok(cfg, cb, true) || encoded_length < 0 || cap < 0 || (!tokens && cap)) return -QASR_ERR_INVALID;
    try {
        const qasr::TransducerResult r = qasr::tdt_greedy(*cfg, *cb, encoded_length);
        if (confidence) *confidence = qasr::transducer_confidence(r.log_probs.data(), (int)r.log_probs.size());
        return emit(r, tokens, log_probs, cap);
    } catch (const std::exception&) { return -QASR_ERR_INVALID; }
}

int qasr_rnnt_greedy_decode(const qasr_transducer_config* cfg, const qasr_transducer_callbacks* cb, int32_t encoded_length, int32_t frame_offset,
                            int32_t* tokens, float* log_probs, int32_t cap, int32_t* eou_detected) {
    if (!cfg_ok(cfg, cb, false) || encoded_length < 0 || frame_offset < 0 || cap < 0 || (!tokens && cap)) return -QASR_ERR_INVALID;
    try {
        const qasr::TransducerResult r = qasr::rnnt_greedy(*cfg, *cb, encoded_length, frame_offset);
        if (eou_detected) *eou_detected = r.eou ? 1 : 0;
        return emit(r, tokens, log_probs, cap);
    } catch (const std::exception&) { return -QASR_ERR_INVALID; }
}

float qasr_log_softmax_at(const float* logits, int32_t n, int32_t id) {
    if (!logits || n <= 0 || id < 0 || id >= n) return 0.0f;
    return qasr::log_softmax_at(logits, n, id);
}
float qasr_transducer_confidence(const float* log_probs, int32_t n) {
    if (!log_probs || n <= 0) return 0.0f;
    return qasr::transducer_confidence(log_probs, n);
}

// ---- vocabulary -------------------------------------------------------------------------------------------------------------
int qasr_sp_vocab_create(const int32_t* ids, const char* const* pieces, size_t n, int style, qasr_sp_vocab** out) {
    if (!out || (style != 0 && style != 1) || (n && (!ids || !pieces))) return QASR_ERR_INVALID;
    try {
        auto v = std::make_unique<qasr_sp_vocab>();
        v->v.style = style;
        for (size_t i = 0; i < n; ++i) v->v.table[ids[i]] = pieces[i] ? pieces[i] : "";
        *out = v.release();
        return QASR_OK;
    } catch (...) { return QASR_ERR_INVALID; }          // allocation failure: nothing crosses the C boundary
}
int qasr_sp_vocab_load(const char* path, int style, qasr_sp_vocab** out) {
    if (!out || !path || (style != 0 && style != 1)) return QASR_ERR_INVALID;
    try {
        auto* v = new qasr_sp_vocab();
        try { v->v = qasr::SpVocab::load_json(path, style); } catch (...) { delete v; throw; }
        *out = v;
        return QASR_OK;
    } catch (const std::exception&) { return QASR_ERR_IO; }
}
void qasr_sp_vocab_destroy(qasr_sp_vocab* v) { delete v; }
int qasr_sp_vocab_count(const qasr_sp_vocab* v) { return v ? (int)v->v.table.size() : 0; }

int qasr_sp_vocab_decode(const qasr_sp_vocab* v, const int32_t* ids, int32_t n, char* buf, size_t cap) {
    if (!v || (!ids && n) || n < 0 || !buf || cap == 0) return -1;
    try {
        const std::string t = v->v.decode(ids, n);
        if (t.size() + 1 > cap) return -1;
        std::memcpy(buf, t.c_str(), t.size() + 1);
        return (int)t.size();
    } catch (...) { return -1; }
}

int qasr_sp_vocab_decode_words(const qasr_sp_vocab* v, const int32_t* ids, int32_t n_ids, const float* log_probs, int32_t n_lp, char* buf,
                               size_t cap, float* confidences, int32_t conf_cap) {
    if (!v || (!ids && n_ids) || (!log_probs && n_lp) || n_ids < 0 || n_lp < 0 || !buf || cap == 0 || conf_cap < 0 || (!confidences && conf_cap)) return -1;
    try {
        std::vector<std::string> words;
        std::vector<float> conf;
        v->v.decode_words(ids, n_ids, log_probs, n_lp, words, conf);
        std::string joined;
        for (size_t i = 0; i < words.size(); ++i) { if (i) joined += '\n'; joined += words[i]; }
        if (joined.size() + 1 > cap || (int64_t)words.size() > conf_cap) return -1;
        std::memcpy(buf, joined.c_str(), joined.size() + 1);
        for (size_t i = 0; i < conf.size(); ++i) confidences[i] = conf[i];
        return (int)words.size();
    } catch (...) { return -1; }
}

// ---- chunk cutting ----------------------------------------------------------------------------------------------------------
int qasr_stream_chunker_create(int32_t samples_per_chunk, int32_t shift, qasr_stream_chunker** out) {
    if (!out || samples_per_chunk <= 0 || shift <= 0 || shift > samples_per_chunk) return QASR_ERR_INVALID;
    try {
        auto c = std::make_unique<qasr_stream_chunker>();
        c->c.samples_per_chunk = samples_per_chunk;
        c->c.shift = shift;
        *out = c.release();
        return QASR_OK;
    } catch (...) { return QASR_ERR_INVALID; }
}
void qasr_stream_chunker_destroy(qasr_stream_chunker* c) { delete c; }
int qasr_stream_chunker_push(qasr_stream_chunker* c, const float* samples, size_t n) {
    if (!c || (!samples && n)) return QASR_ERR_INVALID;
    try { c->c.push(samples, n); } catch (const std::exception&) { return QASR_ERR_CAPACITY; }
    return QASR_OK;
}
int qasr_stream_chunker_pop(qasr_stream_chunker* c, float* chunk) { return (c && chunk && c->c.pop(chunk)) ? 1 : 0; }
int qasr_stream_chunker_flush(qasr_stream_chunker* c, float* chunk) { return (c && chunk && c->c.flush(chunk)) ? 1 : 0; }
size_t qasr_stream_chunker_buffered(const qasr_stream_chunker* c) { return c ? c->c.buf.size() : 0; }

}  // extern "C"
