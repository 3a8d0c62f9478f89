// transducer.cpp -- see transducer.h.  Pure host code (no HIP call).
#include "transducer.h"
#include "json.h"
#include <cmath>
#include <cstring>
#include <fstream>
#include <sstream>
#include <stdexcept>

namespace qasr {

int argmax_first(const float* v, int n) {
    int best = 0;
    float bv = v[0];
    for (int i = 1; i < n; ++i)
        if (v[i] > bv) { bv = v[i]; best = i; }
    return best;
}

float log_softmax_at(const float* logits, int n, int id) {          // TDTGreedyDecoder.swift:149-172
    float mx = logits[0];
    for (int i = 1; i < n; ++i) mx = logits[i] > mx ? logits[i] : mx;
    float s = 0.0f;
    for (int i = 0; i < n; ++i) s += expf(logits[i] - mx);
    return logits[id] - (logf(s) + mx);
}

float transducer_confidence(const float* lp, int n) {                // :135-141
    if (n <= 0) return 0.0f;
    float acc = 0.0f;
    for (int i = 0; i < n; ++i) acc += lp[i];
    const float c = expf(acc / (float)n);
    return c < 1.0f ? c : 1.0f;
}

static void call(int rc, const char* what) {
    if (rc != 0) throw std::runtime_error(std::string(what) + " callback failed with status " + std::to_string(rc));
}

// TDTGreedyDecoder.decode (:45-143): prime the prediction network with blank; blank -> next frame; token -> advance max(duration, 1)
// frames and feed it; pieces below first_text_id (language / control) are fed but not reported.
TransducerResult tdt_greedy(const qasr_transducer_config& c, const qasr_transducer_callbacks& cb, int encoded_length) {
    TransducerResult r;
    std::vector<float> tl((size_t)c.vocab_size + 1), dl((size_t)(c.n_durations > 0 ? c.n_durations : 1));
    call(cb.decoder_step(cb.ctx, c.blank_id), "decoder_step");
    int t = 0;
    while (t < encoded_length) {
        call(cb.joint(cb.ctx, t, tl.data(), dl.data()), "joint");
        const int tok = argmax_first(tl.data(), c.vocab_size + 1);
        if (tok == c.blank_id) { ++t; continue; }
        if (tok >= c.first_text_id) {
            r.tokens.push_back(tok);
            r.log_probs.push_back(log_softmax_at(tl.data(), c.vocab_size + 1, tok));
        }
        const int dur = c.durations[argmax_first(dl.data(), c.n_durations)];
        t += dur > 1 ? dur : 1;
        call(cb.decoder_step(cb.ctx, tok), "decoder_step");
    }
    return r;
}

// RNNTGreedyDecoder.decode (Nemotron :38-90; EOU :58-126): <= max_symbols per frame, blank moves on, the EOU id ends the decode
TransducerResult rnnt_greedy(const qasr_transducer_config& c, const qasr_transducer_callbacks& cb, int encoded_length, int frame_offset) {
    TransducerResult r;
    std::vector<float> tl((size_t)c.vocab_size + 1);
    for (int i = 0; i < encoded_length && !r.eou; ++i) {
        for (int k = 0; k < c.max_symbols; ++k) {
            call(cb.joint(cb.ctx, i + frame_offset, tl.data(), nullptr), "joint");
            const int tok = argmax_first(tl.data(), c.vocab_size + 1);
            if (tok == c.blank_id) break;
            if (c.eou_id >= 0 && tok == c.eou_id) { r.eou = true; break; }
            r.tokens.push_back(tok);
            r.log_probs.push_back(log_softmax_at(tl.data(), c.vocab_size + 1, tok));
            call(cb.decoder_step(cb.ctx, tok), "decoder_step");
        }
    }
    return r;
}

// ---- vocabulary -------------------------------------------------------------------------------------------------------------
static const char kMark[] = "\xE2\x96\x81";                        // U+2581

static std::string replace_all(std::string s, const std::string& from, const std::string& to) {
    size_t pos = 0;
    while ((pos = s.find(from, pos)) != std::string::npos) { s.replace(pos, from.size(), to); pos += to.size(); }
    return s;
}
static std::string trim_ws(const std::string& s) {                   // CharacterSet.whitespaces: space and tab occur in practice
    size_t a = 0, b = s.size();
    while (a < b && (s[a] == ' ' || s[a] == '\t')) ++a;
    while (b > a && (s[b - 1] == ' ' || s[b - 1] == '\t')) --b;
    return s.substr(a, b - a);
}
static bool has_mark_prefix(const std::string& s) { return s.compare(0, 3, kMark) == 0; }
static float word_conf(const std::vector<float>& lps) {
    float acc = 0.0f;
    for (float v : lps) acc += v;
    const float c = expf(acc / (float)lps.size());
    return c < 1.0f ? c : 1.0f;
}

std::string SpVocab::decode(const int32_t* ids, int n) const {
    std::string joined;
    for (int i = 0; i < n; ++i) {
        auto it = table.find(ids[i]);
        if (it == table.end()) continue;                             // unknown ids are skipped (Vocabulary.swift:47)
        joined += it->second;                                        // replacing the mark per piece or after the join is the same string
    }
    return trim_ws(replace_all(joined, kMark, " "));
}

void SpVocab::decode_words(const int32_t* ids, int n_ids, const float* lp, int n_lp, std::vector<std::string>& words, std::vector<float>& conf) const {
    words.clear();
    conf.clear();
    if (n_ids != n_lp) {
        if (style == 0) { words.push_back(decode(ids, n_ids)); conf.push_back(0.0f); }      // ParakeetASR/Vocabulary.swift:62-64
        return;                                                                              // Nemotron: [] (:44)
    }
    std::string cur;
    std::vector<float> lps;
    auto flush = [&]() {
        if (style == 0) { words.push_back(cur); conf.push_back(word_conf(lps)); return; }
        const std::string w = trim_ws(replace_all(cur, kMark, " "));
        if (!w.empty()) { words.push_back(w); conf.push_back(word_conf(lps)); }
    };
    for (int i = 0; i < n_ids; ++i) {
        auto it = table.find(ids[i]);
        if (it == table.end()) continue;
        const std::string& tok = it->second;
        if (has_mark_prefix(tok) && !cur.empty()) {
            flush();
            cur.clear();
            lps.clear();
        }
        cur += style == 0 ? replace_all(tok, kMark, "") : tok;
        lps.push_back(lp[i]);
    }
    if (!cur.empty()) flush();
}

SpVocab SpVocab::load_json(const std::string& path, int style) {     // {"0": "▁the", ...} (Vocabulary.swift:24-37)
    std::ifstream f(path, std::ios::binary);
    if (!f.good()) throw std::runtime_error("cannot open " + path);
    std::stringstream ss;
    ss << f.rdbuf();
    const std::string text = ss.str();
    Json j = JsonParser(text.data(), text.size()).parse();
    if (j.type != Json::Obj) throw std::runtime_error(path + ": expected a JSON object of id -> piece");
    SpVocab v;
    v.style = style;
    for (auto& kv : j.obj) {
        if (kv.second.type != Json::Str) throw std::runtime_error(path + ": piece of id " + kv.first + " is not a string");
        char* end = nullptr;
        const long id = std::strtol(kv.first.c_str(), &end, 10);
        if (end == kv.first.c_str() || *end != '\0') continue;      // `guard let id = Int(key) else { continue }`
        v.table[(int32_t)id] = kv.second.str;
    }
    return v;
}

// ---- chunk cutting ----------------------------------------------------------------------------------------------------------
bool StreamChunker::pop(float* chunk) {                               // StreamingSession.swift:118-127
    if ((int)buf.size() < samples_per_chunk) return false;
    std::memcpy(chunk, buf.data(), (size_t)samples_per_chunk * sizeof(float));
    const size_t drop = std::min((size_t)shift, buf.size());
    buf.erase(buf.begin(), buf.begin() + (long)drop);
    return true;
}

bool StreamChunker::flush(float* chunk) {                             // :133-139
    if (buf.empty()) return false;
    const size_t k = std::min(buf.size(), (size_t)samples_per_chunk);
    std::memcpy(chunk, buf.data(), k * sizeof(float));
    std::memset(chunk + k, 0, ((size_t)samples_per_chunk - k) * sizeof(float));
    buf.clear();
    return true;
}

}  // namespace qasr
