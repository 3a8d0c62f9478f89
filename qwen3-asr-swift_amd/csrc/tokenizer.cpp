// tokenizer.cpp -- byte-level BPE detokeniser + "<asr_text>" post-strip.
// Reference: Sources/AudioCommon/Tokenizer.swift:37-89 (vocab.json + tokenizer_config.json
// added_tokens_decoder), :111-181 (decode, GPT-2 byte table), Sources/Qwen3ASR/Qwen3ASR.swift:283-289.
#include "engine.h"
#include "json.h"
#include <fstream>
#include <sstream>

namespace qasr {

// GPT-2 byte -> unicode table inverted: code point -> byte, or -1 (Tokenizer.swift:146-181)
static int unicode_to_byte(unsigned cp) {
    struct Table { int v[0x200]; };
    static const Table table = [] {               // initialised once, thread-safely (engines may detokenise from several threads)
        Table t;
        for (auto& x : t.v) x = -1;
        bool direct[256] = {};
        for (int b = 33; b <= 126; ++b) direct[b] = true;
        for (int b = 0xA1; b <= 0xAC; ++b) direct[b] = true;
        for (int b = 0xAE; b <= 0xFF; ++b) direct[b] = true;
        int n = 0;
        for (int b = 0; b < 256; ++b) {
            if (direct[b]) t.v[b] = b;
            else t.v[0x100 + n++] = b;
        }
        return t;
    }();
    return cp < 0x200 ? table.v[cp] : -1;
}

// decode one UTF-8 code point from a well-formed std::string (tokens come from JSON / callers)
static unsigned next_cp(const std::string& s, size_t& i, size_t& len) {
    unsigned char c = (unsigned char)s[i];
    unsigned cp;
    if (c < 0x80) { cp = c; len = 1; }
    else if ((c >> 5) == 6 && i + 1 < s.size()) { cp = ((c & 0x1F) << 6) | (s[i + 1] & 0x3F); len = 2; }
    else if ((c >> 4) == 14 && i + 2 < s.size()) { cp = ((c & 0x0F) << 12) | ((s[i + 1] & 0x3F) << 6) | (s[i + 2] & 0x3F); len = 3; }
    else if ((c >> 3) == 30 && i + 3 < s.size()) { cp = ((c & 0x07) << 18) | ((s[i + 1] & 0x3F) << 12) | ((s[i + 2] & 0x3F) << 6) | (s[i + 3] & 0x3F); len = 4; }
    else { cp = c; len = 1; }
    return cp;
}

// String(decoding:as: UTF8.self): invalid sequences -> U+FFFD per maximal subpart
static std::string sanitize_utf8(const std::string& in) {
    std::string out;
    const unsigned char* s = (const unsigned char*)in.data();
    size_t n = in.size(), i = 0;
    auto cont = [&](size_t k, unsigned lo = 0x80, unsigned hi = 0xBF) { return k < n && s[k] >= lo && s[k] <= hi; };
    while (i < n) {
        unsigned char c = s[i];
        size_t need = 0;
        unsigned lo = 0x80, hi = 0xBF;
        if (c < 0x80) { out += (char)c; ++i; continue; }
        else if (c >= 0xC2 && c <= 0xDF) need = 1;
        else if (c == 0xE0) { need = 2; lo = 0xA0; }
        else if ((c >= 0xE1 && c <= 0xEC) || c == 0xEE || c == 0xEF) need = 2;
        else if (c == 0xED) { need = 2; hi = 0x9F; }
        else if (c == 0xF0) { need = 3; lo = 0x90; }
        else if (c >= 0xF1 && c <= 0xF3) need = 3;
        else if (c == 0xF4) { need = 3; hi = 0x8F; }
        else { out += "\xEF\xBF\xBD"; ++i; continue; }
        size_t k = 1;
        bool ok = cont(i + 1, lo, hi);
        if (ok) { for (k = 2; k <= need; ++k) if (!cont(i + k)) { ok = false; break; } }
        if (ok) { out.append(in, i, need + 1); i += need + 1; }
        else { out += "\xEF\xBF\xBD"; i += k; }
    }
    return out;
}

// CharacterSet.whitespaces: Unicode Zs + TAB
static bool is_ws(unsigned cp) {
    return cp == 0x09 || cp == 0x20 || cp == 0xA0 || cp == 0x1680 || (cp >= 0x2000 && cp <= 0x200A) || cp == 0x202F ||
           cp == 0x205F || cp == 0x3000;
}

static std::string trim_ws(const std::string& s) {
    size_t b = 0, e = s.size();
    while (b < e) {
        size_t len;
        unsigned cp = next_cp(s, b, len);
        if (!is_ws(cp)) break;
        b += len;
    }
    while (e > b) {
        size_t k = e - 1;
        while (k > b && ((unsigned char)s[k] & 0xC0) == 0x80) --k;
        size_t len;
        unsigned cp = next_cp(s, k, len);
        if (k + len != e || !is_ws(cp)) break;
        e = k;
    }
    return s.substr(b, e - b);
}

void Engine::set_vocab(const int32_t* ids, const char* const* tokens, size_t n) {
    for (size_t i = 0; i < n; ++i) {
        id_to_token_[ids[i]] = tokens[i];
        token_to_id_[tokens[i]] = ids[i];
    }
}

// merges.txt (Tokenizer.swift:92-106): rank = line index, '#' lines and empty lines skipped
void Engine::set_merges(const std::string& text) {
    merge_rank_.clear();
    size_t pos = 0;
    int idx = 0;
    while (pos <= text.size()) {
        size_t nl = text.find('\n', pos);
        if (nl == std::string::npos) nl = text.size();
        std::string line = text.substr(pos, nl - pos);
        if (!line.empty() && line.back() == '\r') line.pop_back();
        if (!line.empty() && line[0] != '#') {
            size_t sp = line.find(' ');
            if (sp != std::string::npos && line.find(' ', sp + 1) == std::string::npos) merge_rank_[line] = idx;
        }
        ++idx;
        pos = nl + 1;
    }
}

static void put_cp(std::string& s, unsigned cp) {
    if (cp < 0x80) s += (char)cp;
    else if (cp < 0x800) { s += (char)(0xC0 | (cp >> 6)); s += (char)(0x80 | (cp & 0x3F)); }
    else { s += (char)(0xE0 | (cp >> 12)); s += (char)(0x80 | ((cp >> 6) & 0x3F)); s += (char)(0x80 | (cp & 0x3F)); }
}

// GPT-2 byte -> unicode (Tokenizer.swift:146-172)
static unsigned byte_to_unicode(unsigned char b) {
    struct Table { unsigned v[256]; };
    static const Table tbl = [] {
        Table t;
        bool direct[256] = {};
        for (int x = 33; x <= 126; ++x) direct[x] = true;
        for (int x = 0xA1; x <= 0xAC; ++x) direct[x] = true;
        for (int x = 0xAE; x <= 0xFF; ++x) direct[x] = true;
        int n = 0;
        for (int x = 0; x < 256; ++x) t.v[x] = direct[x] ? (unsigned)x : 0x100u + n++;
        return t;
    }();
    const unsigned* table = tbl.v;
    return table[b];
}

// Qwen3Tokenizer.encode (Tokenizer.swift:195-289): whitespace pre-tokenisation (the space / newline / tab
// starts the next word), byte-level mapping, lowest-rank-pair BPE; ids of pieces missing from the vocab are
// dropped; without merges: per-character lookup.
std::vector<int32_t> Engine::encode_text(const std::string& text) const {
    std::vector<int32_t> ids;
    if (merge_rank_.empty()) {
        for (size_t i = 0; i < text.size();) {
            size_t len;
            next_cp(text, i, len);
            auto it = token_to_id_.find(text.substr(i, len));
            if (it != token_to_id_.end()) ids.push_back(it->second);
            i += len;
        }
        return ids;
    }
    std::vector<std::string> words;
    std::string cur;
    for (char ch : text) {
        if (ch == ' ' || ch == '\n' || ch == '\t') {
            if (!cur.empty()) words.push_back(cur);
            cur.assign(1, ch);
        } else {
            cur += ch;
        }
    }
    if (!cur.empty()) words.push_back(cur);
    for (const std::string& w : words) {
        std::vector<std::string> pieces;
        for (unsigned char b : w) {
            std::string p;
            put_cp(p, byte_to_unicode(b));
            pieces.push_back(p);
        }
        while (pieces.size() > 1) {
            int best_rank = -1;
            size_t best_i = 0;
            for (size_t i = 0; i + 1 < pieces.size(); ++i) {
                auto it = merge_rank_.find(pieces[i] + " " + pieces[i + 1]);
                if (it != merge_rank_.end() && (best_rank < 0 || it->second < best_rank)) { best_rank = it->second; best_i = i; }
            }
            if (best_rank < 0) break;
            const std::string a = pieces[best_i], b = pieces[best_i + 1];
            std::vector<std::string> out;
            for (size_t i = 0; i < pieces.size();) {
                if (i + 1 < pieces.size() && pieces[i] == a && pieces[i + 1] == b) { out.push_back(a + b); i += 2; }
                else { out.push_back(pieces[i]); i += 1; }
            }
            pieces.swap(out);
        }
        for (const std::string& p : pieces) {
            auto it = token_to_id_.find(p);
            if (it != token_to_id_.end()) ids.push_back(it->second);
        }
    }
    return ids;
}

static bool read_file(const std::string& path, std::string& out) {
    std::ifstream f(path, std::ios::binary);
    if (!f) return false;
    std::stringstream ss;
    ss << f.rdbuf();
    out = ss.str();
    return true;
}

void Engine::load_vocab_files(const std::string& dir) {
    std::string txt;
    if (!read_file(dir + "/vocab.json", txt)) return;      // tokenizer is optional (Qwen3ASR.swift:644-649)
    Json v = JsonParser(txt.data(), txt.size()).parse();
    if (v.type != Json::Obj) throw std::runtime_error("vocab.json: expected {token: id}");
    for (auto& kv : v.obj)
        if (kv.second.type == Json::Num) { id_to_token_[(int32_t)kv.second.num] = kv.first; token_to_id_[kv.first] = (int32_t)kv.second.num; }
    if (read_file(dir + "/tokenizer_config.json", txt)) {
        Json c = JsonParser(txt.data(), txt.size()).parse();
        const Json* added = c.get("added_tokens_decoder");
        if (added && added->type == Json::Obj)
            for (auto& kv : added->obj) {
                const Json* content = kv.second.get("content");
                if (content && content->type == Json::Str) {
                    id_to_token_[(int32_t)std::stol(kv.first)] = content->str;
                    token_to_id_[content->str] = (int32_t)std::stol(kv.first);
                }
            }
    }
    if (read_file(dir + "/merges.txt", txt)) set_merges(txt);
}

std::string Engine::detokenize(const int32_t* tokens, int n, bool strip_asr_prefix) const {
    std::string buf;
    for (int t = 0; t < n; ++t) {
        auto it = id_to_token_.find(tokens[t]);
        if (it == id_to_token_.end()) continue;
        const std::string& tok = it->second;
        const bool lt = tok.size() >= 2 && tok.front() == '<' && tok.back() == '>';
        if (tok.size() >= 4 && tok.compare(0, 2, "<|") == 0 && tok.compare(tok.size() - 2, 2, "|>") == 0) continue;
        if (lt && tok.find('|') == std::string::npos) { buf += tok; continue; }
        for (size_t i = 0; i < tok.size();) {
            size_t len;
            unsigned cp = next_cp(tok, i, len);
            int b = unicode_to_byte(cp);
            if (b >= 0) buf += (char)b; else buf.append(tok, i, len);
            i += len;
        }
    }
    std::string text = trim_ws(sanitize_utf8(buf));
    if (strip_asr_prefix) {
        size_t p = text.find("<asr_text>");
        if (p != std::string::npos) text = trim_ws(text.substr(p + 10));
    }
    return text;
}

}  // namespace qasr
