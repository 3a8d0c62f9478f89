// json.h -- minimal JSON reader (objects, arrays, strings with \uXXXX, numbers, literals) for the
// safetensors header, vocab.json and tokenizer_config.json.  Not a general-purpose library.
#pragma once
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

namespace qasr {

struct Json {
    enum Type { Null, Bool, Num, Str, Arr, Obj } type = Null;
    bool b = false;
    double num = 0;
    std::string str;
    std::vector<Json> arr;
    std::vector<std::pair<std::string, Json>> obj;   // insertion order kept

    const Json* get(const std::string& k) const {
        for (auto& kv : obj) if (kv.first == k) return &kv.second;
        return nullptr;
    }
};

class JsonParser {
public:
    JsonParser(const char* p, size_t n) : p_(p), e_(p + n) {}
    Json parse() {
        Json j = value();
        ws();
        if (p_ != e_) fail("trailing characters");
        return j;
    }

private:
    const char *p_, *e_;
    int depth_ = 0;
    static constexpr int kMaxDepth = 64;          // a crafted header must not recurse the stack away
    [[noreturn]] void fail(const char* m) { throw std::runtime_error(std::string("json: ") + m); }
    void ws() { while (p_ < e_ && (*p_ == ' ' || *p_ == '\n' || *p_ == '\t' || *p_ == '\r')) ++p_; }
    static void put_utf8(std::string& s, unsigned cp) {
        if (cp < 0x80) s += (char)cp;
        else if (cp < 0x800) { s += (char)(0xC0 | (cp >> 6)); s += (char)(0x80 | (cp & 0x3F)); }
        else if (cp < 0x10000) { s += (char)(0xE0 | (cp >> 12)); s += (char)(0x80 | ((cp >> 6) & 0x3F)); s += (char)(0x80 | (cp & 0x3F)); }
        else { s += (char)(0xF0 | (cp >> 18)); s += (char)(0x80 | ((cp >> 12) & 0x3F)); s += (char)(0x80 | ((cp >> 6) & 0x3F)); s += (char)(0x80 | (cp & 0x3F)); }
    }
    unsigned hex4() {
        if (e_ - p_ < 4) fail("bad \\u escape");
        unsigned v = 0;
        for (int i = 0; i < 4; ++i) {
            char c = *p_++;
            v <<= 4;
            if (c >= '0' && c <= '9') v |= c - '0';
            else if (c >= 'a' && c <= 'f') v |= c - 'a' + 10;
            else if (c >= 'A' && c <= 'F') v |= c - 'A' + 10;
            else fail("bad hex digit");
        }
        return v;
    }
    std::string string() {
        if (p_ >= e_ || *p_ != '"') fail("expected string");
        ++p_;
        std::string s;
        while (true) {
            if (p_ >= e_) fail("unterminated string");
            char c = *p_++;
            if (c == '"') break;
            if (c != '\\') { s += c; continue; }
            if (p_ >= e_) fail("bad escape");
            char x = *p_++;
            switch (x) {
                case '"': s += '"'; break; case '\\': s += '\\'; break; case '/': s += '/'; break;
                case 'b': s += '\b'; break; case 'f': s += '\f'; break; case 'n': s += '\n'; break;
                case 'r': s += '\r'; break; case 't': s += '\t'; break;
                case 'u': {
                    unsigned cp = hex4();
                    if (cp >= 0xD800 && cp <= 0xDBFF && e_ - p_ >= 6 && p_[0] == '\\' && p_[1] == 'u') {
                        p_ += 2;
                        unsigned lo = hex4();
                        if (lo >= 0xDC00 && lo <= 0xDFFF) cp = 0x10000 + ((cp - 0xD800) << 10) + (lo - 0xDC00);
                    }
                    put_utf8(s, cp);
                    break;
                }
                default: fail("unknown escape");
            }
        }
        return s;
    }
    Json value() {
        ws();
        if (p_ >= e_) fail("unexpected end");
        Json j;
        char c = *p_;
        struct Depth { int& d; explicit Depth(int& x) : d(x) { ++d; } ~Depth() { --d; } } guard(depth_);
        if (depth_ > kMaxDepth) fail("nesting too deep");
        if (c == '{') {
            j.type = Json::Obj;
            ++p_;
            ws();
            if (p_ < e_ && *p_ == '}') { ++p_; return j; }
            while (true) {
                ws();
                std::string k = string();
                ws();
                if (p_ >= e_ || *p_ != ':') fail("expected ':'");
                ++p_;
                j.obj.emplace_back(std::move(k), value());
                ws();
                if (p_ < e_ && *p_ == ',') { ++p_; continue; }
                if (p_ < e_ && *p_ == '}') { ++p_; break; }
                fail("expected ',' or '}'");
            }
        } else if (c == '[') {
            j.type = Json::Arr;
            ++p_;
            ws();
            if (p_ < e_ && *p_ == ']') { ++p_; return j; }
            while (true) {
                j.arr.push_back(value());
                ws();
                if (p_ < e_ && *p_ == ',') { ++p_; continue; }
                if (p_ < e_ && *p_ == ']') { ++p_; break; }
                fail("expected ',' or ']'");
            }
        } else if (c == '"') {
            j.type = Json::Str;
            j.str = string();
        } else if (c == 't' && e_ - p_ >= 4 && std::string(p_, 4) == "true") { j.type = Json::Bool; j.b = true; p_ += 4; }
        else if (c == 'f' && e_ - p_ >= 5 && std::string(p_, 5) == "false") { j.type = Json::Bool; p_ += 5; }
        else if (c == 'n' && e_ - p_ >= 4 && std::string(p_, 4) == "null") { p_ += 4; }
        else {
            const char* s = p_;
            while (p_ < e_ && (std::string("+-0123456789.eE").find(*p_) != std::string::npos)) ++p_;
            if (s == p_) fail("unexpected character");
            j.type = Json::Num;
            j.num = std::stod(std::string(s, p_ - s));
        }
        return j;
    }
};

}  // namespace qasr
