// aligner.cpp -- host logic of the forced aligner: word splitting with timestamp slots, LIS monotonicity
// fix-up, plateau detection and the align / alignLong drivers around Engine::align_forward.
//
// Reference: Sources/Qwen3ASR/TextPreprocessing.swift:48-93,103-127,165-335 (default path; the NLTokenizer
// languages are refused, see qasr_split_words), TimestampCorrection.swift:15-145,
// ForcedAligner.swift:97-215 (alignLong, plateau), :226-331 (align).
#include "engine.h"
#include "unicode_lnm.h"
#include <algorithm>
#include <cmath>
#include <cstring>

namespace qasr {

// ---- UTF-8 <-> scalars -----------------------------------------------------------------------------
static std::vector<uint32_t> utf8_scalars(const std::string& s) {
    std::vector<uint32_t> out;
    size_t i = 0;
    const size_t n = s.size();
    while (i < n) {
        const unsigned char c = (unsigned char)s[i];
        uint32_t cp = 0xFFFD;
        int len = 1;
        if (c < 0x80) cp = c;
        else if ((c >> 5) == 6 && i + 1 < n) { cp = ((c & 0x1Fu) << 6) | ((unsigned char)s[i + 1] & 0x3Fu); len = 2; }
        else if ((c >> 4) == 14 && i + 2 < n) {
            cp = ((c & 0x0Fu) << 12) | (((unsigned char)s[i + 1] & 0x3Fu) << 6) | ((unsigned char)s[i + 2] & 0x3Fu);
            len = 3;
        } else if ((c >> 3) == 30 && i + 3 < n) {
            cp = ((c & 0x07u) << 18) | (((unsigned char)s[i + 1] & 0x3Fu) << 12) | (((unsigned char)s[i + 2] & 0x3Fu) << 6) |
                 ((unsigned char)s[i + 3] & 0x3Fu);
            len = 4;
        }
        out.push_back(cp);
        i += (size_t)len;
    }
    return out;
}

static void append_utf8(std::string& s, uint32_t cp) {
    if (cp < 0x80) s.push_back((char)cp);
    else if (cp < 0x800) { s.push_back((char)(0xC0 | (cp >> 6))); s.push_back((char)(0x80 | (cp & 0x3F))); }
    else if (cp < 0x10000) {
        s.push_back((char)(0xE0 | (cp >> 12))); s.push_back((char)(0x80 | ((cp >> 6) & 0x3F))); s.push_back((char)(0x80 | (cp & 0x3F)));
    } else {
        s.push_back((char)(0xF0 | (cp >> 18))); s.push_back((char)(0x80 | ((cp >> 12) & 0x3F)));
        s.push_back((char)(0x80 | ((cp >> 6) & 0x3F))); s.push_back((char)(0x80 | (cp & 0x3F)));
    }
}

static std::string to_utf8(const std::vector<uint32_t>& v) {
    std::string s;
    for (uint32_t cp : v) append_utf8(s, cp);
    return s;
}

// ---- scalar classes ------------------------------------------------------------------------------
static bool is_lnm(uint32_t cp) {       // general category L*, N* or M* (tools/gen_unicode_lnm.py)
    int lo = 0, hi = kUnicodeLnmRanges - 1;
    while (lo <= hi) {
        const int mid = (lo + hi) / 2;
        if (cp < kUnicodeLnm[mid][0]) hi = mid - 1;
        else if (cp > kUnicodeLnm[mid][1]) lo = mid + 1;
        else return true;
    }
    return false;
}
static bool is_kept(uint32_t cp) { return cp == '\'' || is_lnm(cp); }                 // TextPreprocessing.swift:300-316
static bool is_space(uint32_t cp) {                                                    // Unicode White_Space
    return (cp >= 0x09 && cp <= 0x0D) || cp == 0x20 || cp == 0x85 || cp == 0xA0 || cp == 0x1680 || (cp >= 0x2000 && cp <= 0x200A) ||
           cp == 0x2028 || cp == 0x2029 || cp == 0x202F || cp == 0x205F || cp == 0x3000;
}
static bool is_han(uint32_t v) {                                                       // TextPreprocessing.swift:322-332
    return (v >= 0x4E00 && v <= 0x9FFF) || (v >= 0x3400 && v <= 0x4DBF) || (v >= 0x20000 && v <= 0x2A6DF) ||
           (v >= 0x2A700 && v <= 0x2B73F) || (v >= 0x2B740 && v <= 0x2B81F) || (v >= 0x2B820 && v <= 0x2CEAF) ||
           (v >= 0xF900 && v <= 0xFAFF);
}
static std::vector<uint32_t> clean(const std::vector<uint32_t>& t) {
    std::vector<uint32_t> o;
    for (uint32_t c : t) if (is_kept(c)) o.push_back(c);
    return o;
}

bool aligner_needs_nl_tokenizer(const std::string& language) {     // TextPreprocessing.swift:103-129
    std::string l = language;
    for (auto& c : l) c = (char)tolower((unsigned char)c);
    for (const char* n : {"japanese", "korean", "thai", "lao", "khmer", "burmese", "myanmar", "tibetan"})
        if (l.find(n) != std::string::npos) return true;
    for (const char* n : {"ja", "ko", "th", "lo", "km", "my", "bo"})
        if (l == n) return true;
    return false;
}

typedef std::pair<std::vector<uint32_t>, std::vector<uint32_t>> ScalarPair;   // (surface, cleaned)

static void pairs_for_segment(const std::vector<uint32_t>& seg, std::vector<ScalarPair>& out) {   // :207-263
    const size_t first = out.size();
    bool has_han = false;
    for (uint32_t c : seg) has_han |= is_han(c);
    if (!has_han) {
        std::vector<uint32_t> c = clean(seg);
        if (!c.empty()) out.push_back({seg, c});
        return;
    }
    std::vector<uint32_t> buf;
    auto flush = [&](bool before_han) {
        if (buf.empty()) return;
        std::vector<uint32_t> c = clean(buf);
        if (c.empty()) {
            if (out.size() > first) { auto& s = out.back().first; s.insert(s.end(), buf.begin(), buf.end()); buf.clear(); }
            else if (!before_han) buf.clear();
            return;
        }
        out.push_back({buf, c});
        buf.clear();
    };
    for (uint32_t c : seg) {
        if (is_han(c)) {
            flush(true);
            if (!buf.empty()) {                       // leading pure punctuation waiting for a Han anchor
                std::vector<uint32_t> s = buf;
                s.push_back(c);
                out.push_back({s, {c}});
                buf.clear();
            } else out.push_back({{c}, {c}});
        } else buf.push_back(c);
    }
    flush(false);
}

std::vector<std::pair<std::string, std::string>> aligner_split_word_pairs(const std::string& text) {   // :174-199
    std::vector<ScalarPair> pairs;
    std::vector<uint32_t> seg;
    auto end_segment = [&]() {
        if (seg.empty()) return;
        const size_t before = pairs.size();
        pairs_for_segment(seg, pairs);
        if (pairs.size() == before && !pairs.empty()) {          // pure punctuation: rides with the previous word
            auto& s = pairs.back().first;
            s.insert(s.end(), seg.begin(), seg.end());
        }
        seg.clear();
    };
    for (uint32_t c : utf8_scalars(text)) {
        if (is_space(c)) end_segment();
        else seg.push_back(c);
    }
    end_segment();
    std::vector<std::pair<std::string, std::string>> out;
    for (auto& p : pairs) out.push_back({to_utf8(p.first), to_utf8(p.second)});
    return out;
}

// ---- TimestampCorrection.swift ---------------------------------------------------------------------
std::vector<int32_t> aligner_lis_positions(const int32_t* a, size_t n) {   // :102-144
    std::vector<int32_t> pos;
    if (n == 0) return pos;
    std::vector<int32_t> tails, tail_idx, parent(n, -1);
    for (size_t i = 0; i < n; ++i) {
        size_t lo = 0, hi = tails.size();
        while (lo < hi) {
            const size_t mid = (lo + hi) / 2;
            if (tails[mid] < a[i]) lo = mid + 1;
            else hi = mid;
        }
        if (lo == tails.size()) { tails.push_back(a[i]); tail_idx.push_back((int32_t)i); }
        else { tails[lo] = a[i]; tail_idx[lo] = (int32_t)i; }
        parent[i] = lo > 0 ? tail_idx[lo - 1] : -1;
    }
    for (int32_t idx = tail_idx.back(); idx != -1; idx = parent[(size_t)idx]) pos.push_back(idx);
    std::reverse(pos.begin(), pos.end());
    return pos;
}

std::vector<int32_t> aligner_enforce_monotonicity(const int32_t* raw, size_t n) {   // :15-99
    std::vector<int32_t> out(raw, raw + n);
    if (n <= 1) return out;
    const std::vector<int32_t> lis = aligner_lis_positions(raw, n);
    if (lis.size() == n) return out;
    std::vector<char> in_lis(n, 0);
    for (int32_t p : lis) in_lis[(size_t)p] = 1;
    const int na = (int)lis.size();                      // anchors: (lis[k], raw[lis[k]])
    int a_idx = 0;
    for (int i = 0; i < (int)n; ++i) {
        if (in_lis[(size_t)i]) {
            for (int k = 0; k < na; ++k) if (lis[(size_t)k] == i) { a_idx = k; break; }
            continue;
        }
        int prev = -1, next = -1;
        if (a_idx < na && lis[(size_t)a_idx] < i) prev = a_idx;
        else if (a_idx > 0) prev = a_idx - 1;
        int nx = a_idx;
        while (nx < na && lis[(size_t)nx] <= i) ++nx;
        if (nx < na) next = nx;
        if (prev >= 0 && next >= 0) {
            const int pp = lis[(size_t)prev], np = lis[(size_t)next], pv = raw[pp], nv = raw[np];
            if (np - pp <= 3) out[(size_t)i] = (i - pp) <= (np - i) ? pv : nv;
            else {
                const float t = (float)(i - pp) / (float)(np - pp);
                out[(size_t)i] = pv + (int32_t)(t * (float)(nv - pv));
            }
        } else if (prev >= 0) out[(size_t)i] = raw[lis[(size_t)prev]];
        else if (next >= 0) out[(size_t)i] = raw[lis[(size_t)next]];
    }
    for (size_t i = 1; i < n; ++i)
        if (out[i] < out[i - 1]) out[i] = out[i - 1];
    return out;
}

int aligner_find_trailing_plateau(const float* starts, size_t n, float tol, int min_size) {   // ForcedAligner.swift:196-215
    // n == 0 or a non-positive min_size never reach the scan below (`n - 1` would wrap): "no plateau", like the reference's
    // `guard alignedWords.count > minSize` for its only caller (minSize = 5)
    if (n == 0 || min_size <= 0 || (long)n <= (long)min_size) return (int)n;
    size_t plateau = n;
    for (size_t i = n - 1; i >= 1; --i) {
        if (std::fabs(starts[i] - starts[i - 1]) < tol) plateau = i - 1;
        else break;
    }
    return (long)(n - plateau) >= (long)min_size ? (int)plateau : (int)n;
}

// ---- Engine side ---------------------------------------------------------------------------------
Engine::SlottedText Engine::prepare_alignment(const std::vector<std::pair<std::string, std::string>>& pairs) const {   // :48-93
    SlottedText st;
    for (auto& p : pairs) {
        std::vector<int32_t> toks = encode_text(p.second);
        if (toks.empty()) {
            if (!st.words.empty()) st.words.back() += p.first;
            continue;
        }
        st.ts_pos.push_back((int32_t)st.ids.size());
        st.ids.push_back(cfg_.tok_timestamp);
        st.ids.insert(st.ids.end(), toks.begin(), toks.end());
        st.ts_pos.push_back((int32_t)st.ids.size());
        st.ids.push_back(cfg_.tok_timestamp);
        st.words.push_back(p.first);
    }
    return st;
}

// align (long_form = false) / alignLong (true) on pre-split words; fills al_words / al_raw, returns the pass count
// long_text: the caller's text (qasr_align_long); the re-alignment passes then re-split it exactly like the reference
int Engine::align_words(const float* pcm, size_t n, const std::vector<std::pair<std::string, std::string>>& pairs_in, bool long_form,
                        const std::string* long_text) {
    std::string rem_text = long_text ? *long_text : std::string();
    al_words.clear();
    al_raw.clear();
    const float seg_t = cfg_.timestamp_segment_time;
    // alignLong's constants (ForcedAligner.swift:112-115)
    const float bypass_s = 240.0f, min_chunk_s = 5.0f, plateau_tol = 0.1f;
    const int plateau_min = 5;
    std::vector<std::pair<std::string, std::string>> pairs = pairs_in;
    const float* audio = pcm;
    size_t len = n;
    float offset = 0.0f;
    int pass = 1;
    while (len > 0 && !pairs.empty()) {
        const SlottedText st = prepare_alignment(pairs);
        if (st.words.empty()) break;
        std::vector<std::vector<int32_t>> raw;
        align_forward(&audio, &len, 1, {st.ids}, {st.ts_pos}, raw, nullptr);
        al_raw = raw[0];
        const std::vector<int32_t> fixed = aligner_enforce_monotonicity(al_raw.data(), al_raw.size());
        std::vector<AlignedWord> aligned;                  // ForcedAligner.swift:311-330
        for (size_t w = 0; w < st.words.size() && 2 * w + 1 < fixed.size(); ++w) {
            const float s = (float)fixed[2 * w] * seg_t, e = (float)fixed[2 * w + 1] * seg_t;
            aligned.push_back({st.words[w], s, std::max(e, s)});
        }
        auto append = [&](size_t count) {
            for (size_t i = 0; i < count; ++i) al_words.push_back({aligned[i].text, aligned[i].start + offset, aligned[i].end + offset});
        };
        if (aligned.empty()) break;
        const float duration = (float)len / 16000.0f;
        if (!long_form || duration <= bypass_s || (int)aligned.size() < plateau_min * 2) { append(aligned.size()); break; }
        std::vector<float> starts;
        for (auto& a : aligned) starts.push_back(a.start);
        const int plateau = aligner_find_trailing_plateau(starts.data(), starts.size(), plateau_tol, plateau_min);
        if (plateau == (int)aligned.size()) { append(aligned.size()); break; }
        // keep the reliable prefix, re-align the remaining audio with the remaining words (:150-175)
        if (plateau == 0) break;                           // the reference force-unwraps reliable.last: nothing reliable
        const float split_time = aligned[(size_t)plateau - 1].end;
        append((size_t)plateau);
        const size_t split_sample = (size_t)(split_time * 16000.0f);
        if (split_sample >= len) break;
        if ((float)(len - split_sample) / 16000.0f < min_chunk_s) break;
        if (long_text) {
            // ForcedAligner.swift:162-165: the remaining TEXT is split on single spaces (empty pieces dropped), the first
            // `plateau` pieces go, and the rest is joined and split into words again by the next align pass.  That is not
            // the same as dropping `plateau` aligned words when a punctuation-only piece was merged into its neighbour or a
            // piece encodes to no token (tests/golden/kat_aligner.json: align_long_resplit).
            std::vector<std::string> pieces;
            for (size_t i = 0; i < rem_text.size();) {
                size_t j = rem_text.find(' ', i);
                if (j == std::string::npos) j = rem_text.size();
                if (j > i) pieces.push_back(rem_text.substr(i, j - i));
                i = j + 1;
            }
            if ((size_t)plateau >= pieces.size()) break;
            rem_text.clear();
            for (size_t i = (size_t)plateau; i < pieces.size(); ++i) { if (!rem_text.empty()) rem_text += ' '; rem_text += pieces[i]; }
            pairs = aligner_split_word_pairs(rem_text);
        } else {
            // caller-split words (no text to re-split): drop the aligned words themselves
            if ((size_t)plateau >= pairs.size()) break;
            pairs.erase(pairs.begin(), pairs.begin() + plateau);
        }
        audio += split_sample;
        len -= split_sample;
        offset += split_time;
        if (++pass > 10) break;
    }
    al_view.clear();
    for (auto& w : al_words) al_view.push_back({w.text.c_str(), w.start, w.end});
    return pass;
}

// one pass over B clips (ForcedAligner.swift:226-331 per clip); clips whose text has no encodable word get no words
void Engine::align_batch(const float* const* pcm, const size_t* n, size_t B,
                         const std::vector<std::vector<std::pair<std::string, std::string>>>& pairs) {
    if (pairs.size() != B) throw std::invalid_argument("align_batch: one text per clip");
    std::vector<SlottedText> st(B);
    std::vector<std::vector<int32_t>> ids(B), ts(B), raw;
    for (size_t b = 0; b < B; ++b) {
        st[b] = prepare_alignment(pairs[b]);
        ids[b] = st[b].ids;
        ts[b] = st[b].ts_pos;
    }
    align_forward(pcm, n, B, ids, ts, raw, nullptr);
    al_batch.assign(B, AlignResult{});
    const float seg_t = cfg_.timestamp_segment_time;
    for (size_t b = 0; b < B; ++b) {
        AlignResult& r = al_batch[b];
        if (b < raw.size()) r.raw = raw[b];
        const std::vector<int32_t> fixed = aligner_enforce_monotonicity(r.raw.data(), r.raw.size());
        for (size_t w = 0; w < st[b].words.size() && 2 * w + 1 < fixed.size(); ++w) {
            const float s = (float)fixed[2 * w] * seg_t, e = (float)fixed[2 * w + 1] * seg_t;
            r.words.push_back({st[b].words[w], s, std::max(e, s)});
        }
        for (auto& w : r.words) r.view.push_back({w.text.c_str(), w.start, w.end});
    }
}

}  // namespace qasr
