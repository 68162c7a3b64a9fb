// json.h — small DOM JSON reader for the glTF loader (no dependencies).
#pragma once
#include <cstdlib>
#include <map>
#include <memory>
#include <string>
#include <vector>

namespace lpt {

struct Json {
    enum Kind { Null, Bool, Num, Str, Arr, Obj } kind = Null;
    bool b = false;
    double num = 0.0;
    std::string str;
    std::vector<Json> arr;
    std::vector<std::pair<std::string, Json>> obj;

    bool is_null() const { return kind == Null; }
    bool is_obj() const { return kind == Obj; }
    bool is_arr() const { return kind == Arr; }
    bool is_num() const { return kind == Num; }
    bool is_str() const { return kind == Str; }
    const Json *find(const char *key) const {
        if (kind != Obj) return nullptr;
        for (const auto &kv : obj)
            if (kv.first == key) return &kv.second;
        return nullptr;
    }
    const Json &at(const char *key) const {
        static const Json null_json;
        const Json *p = find(key);
        return p ? *p : null_json;
    }
    size_t size() const { return kind == Arr ? arr.size() : 0; }
    const Json &operator[](size_t i) const {
        static const Json null_json;
        return (kind == Arr && i < arr.size()) ? arr[i] : null_json;
    }
    double number(double dflt) const { return kind == Num ? num : dflt; }
    // saturating: a cast of an out-of-range or NaN double is undefined; callers range-check the result
    long long integer(long long dflt) const {
        if (kind != Num) return dflt;
        if (!(num == num)) return -1;
        if (num >= 4.0e18) return (long long)4000000000000000000LL;
        if (num <= -4.0e18) return -(long long)4000000000000000000LL;
        return (long long)num;
    }
};

class JsonParser {
   public:
    JsonParser(const char *p, size_t n) : p_(p), end_(p + n) {}
    bool parse(Json &out) {
        ws();
        if (!value(out, 0)) return false;
        ws();
        return p_ == end_;
    }

   private:
    const char *p_, *end_;
    void ws() {
        while (p_ < end_ && (*p_ == ' ' || *p_ == '\n' || *p_ == '\r' || *p_ == '\t')) ++p_;
    }
    bool lit(const char *s) {
        size_t n = strlen(s);
        if ((size_t)(end_ - p_) < n || memcmp(p_, s, n) != 0) return false;
        p_ += n;
        return true;
    }
    static void utf8(std::string &s, unsigned cp) {
        if (cp < 0x80) s.push_back((char)cp);
        else if (cp < 0x800) { s.push_back((char)(0xC0 | (cp >> 6))); s.push_back((char)(0x80 | (cp & 0x3F))); }
        else if (cp < 0x10000) { s.push_back((char)(0xE0 | (cp >> 12))); s.push_back((char)(0x80 | ((cp >> 6) & 0x3F))); s.push_back((char)(0x80 | (cp & 0x3F))); }
        else { s.push_back((char)(0xF0 | (cp >> 18))); s.push_back((char)(0x80 | ((cp >> 12) & 0x3F))); s.push_back((char)(0x80 | ((cp >> 6) & 0x3F))); s.push_back((char)(0x80 | (cp & 0x3F))); }
    }
    bool hex4(unsigned &v) {
        if (end_ - p_ < 4) return false;
        v = 0;
        for (int i = 0; i < 4; ++i) {
            char c = *p_++;
            v <<= 4;
            if (c >= '0' && c <= '9') v |= (unsigned)(c - '0');
            else if (c >= 'a' && c <= 'f') v |= (unsigned)(c - 'a' + 10);
            else if (c >= 'A' && c <= 'F') v |= (unsigned)(c - 'A' + 10);
            else return false;
        }
        return true;
    }
    bool string(std::string &s) {
        if (p_ >= end_ || *p_ != '"') return false;
        ++p_;
        while (p_ < end_) {
            char c = *p_++;
            if (c == '"') return true;
            if (c == '\\') {
                if (p_ >= end_) return false;
                char e = *p_++;
                switch (e) {
                    case '"': s.push_back('"'); break;
                    case '\\': s.push_back('\\'); break;
                    case '/': s.push_back('/'); break;
                    case 'b': s.push_back('\b'); break;
                    case 'f': s.push_back('\f'); break;
                    case 'n': s.push_back('\n'); break;
                    case 'r': s.push_back('\r'); break;
                    case 't': s.push_back('\t'); break;
                    case 'u': {
                        unsigned cp;
                        if (!hex4(cp)) return false;
                        if (cp >= 0xD800 && cp < 0xDC00 && end_ - p_ >= 6 && p_[0] == '\\' && p_[1] == 'u') {
                            p_ += 2;
                            unsigned lo;
                            if (!hex4(lo)) return false;
                            cp = 0x10000 + ((cp - 0xD800) << 10) + (lo - 0xDC00);
                        }
                        utf8(s, cp);
                        break;
                    }
                    default: return false;
                }
            } else s.push_back(c);
        }
        return false;
    }
    bool value(Json &out, int depth) {
        if (depth > 64 || p_ >= end_) return false;
        char c = *p_;
        if (c == '{') {
            ++p_;
            out.kind = Json::Obj;
            ws();
            if (p_ < end_ && *p_ == '}') { ++p_; return true; }
            for (;;) {
                ws();
                std::string k;
                if (!string(k)) return false;
                ws();
                if (p_ >= end_ || *p_++ != ':') return false;
                ws();
                out.obj.emplace_back(std::move(k), Json());
                if (!value(out.obj.back().second, depth + 1)) return false;
                ws();
                if (p_ >= end_) return false;
                if (*p_ == ',') { ++p_; continue; }
                if (*p_ == '}') { ++p_; return true; }
                return false;
            }
        }
        if (c == '[') {
            ++p_;
            out.kind = Json::Arr;
            ws();
            if (p_ < end_ && *p_ == ']') { ++p_; return true; }
            for (;;) {
                ws();
                out.arr.emplace_back();
                if (!value(out.arr.back(), depth + 1)) return false;
                ws();
                if (p_ >= end_) return false;
                if (*p_ == ',') { ++p_; continue; }
                if (*p_ == ']') { ++p_; return true; }
                return false;
            }
        }
        if (c == '"') { out.kind = Json::Str; return string(out.str); }
        if (c == 't') { out.kind = Json::Bool; out.b = true; return lit("true"); }
        if (c == 'f') { out.kind = Json::Bool; out.b = false; return lit("false"); }
        if (c == 'n') { out.kind = Json::Null; return lit("null"); }
        if (c == '-' || (c >= '0' && c <= '9')) {
            const char *s = p_;
            if (*p_ == '-') ++p_;
            while (p_ < end_ && ((*p_ >= '0' && *p_ <= '9') || *p_ == '.' || *p_ == 'e' || *p_ == 'E' || *p_ == '+' || *p_ == '-')) ++p_;
            std::string tmp(s, p_);
            char *endp = nullptr;
            out.num = strtod(tmp.c_str(), &endp);
            out.kind = Json::Num;
            return endp && *endp == 0;
        }
        return false;
    }
};

}  // namespace lpt
