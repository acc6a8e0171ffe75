// pth_json.h -- minimal JSON DOM for glTF 2.0 documents (stands in for the `gltf` crate's
// serde layer, gltf.rs:20).  Header-only, no dependencies.
#pragma once

#include <cstdlib>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

namespace pth {

struct Json {
    enum Type { Null, Bool, Number, String, Array, Object } type = Null;
    bool b = false;
    double num = 0.0;
    std::string str;
    std::vector<Json> arr;
    std::vector<std::pair<std::string, Json>> obj;   // insertion order kept

    bool is_null() const { return type == Null; }
    const Json *find(const std::string &k) const
    {
        if (type != Object) return nullptr;
        for (const auto &kv : obj) if (kv.first == k) return &kv.second;
        return nullptr;
    }
    bool has(const std::string &k) const { return find(k) != nullptr; }
    const Json &at(const std::string &k) const
    {
        const Json *j = find(k);
        if (!j) throw std::runtime_error("glTF: missing key '" + k + "'");
        return *j;
    }
    const Json &at(size_t i) const
    {
        if (type != Array || i >= arr.size()) throw std::runtime_error("glTF: array index out of range");
        return arr[i];
    }
    size_t size() const { return type == Array ? arr.size() : (type == Object ? obj.size() : 0); }
    double number(double dflt) const { return type == Number ? num : dflt; }
    double get_number(const std::string &k, double dflt) const { const Json *j = find(k); return j ? j->number(dflt) : dflt; }
    long long get_int(const std::string &k, long long dflt) const { const Json *j = find(k); return j && j->type == Number ? (long long)j->num : dflt; }
    std::string get_string(const std::string &k, const std::string &dflt) const { const Json *j = find(k); return j && j->type == String ? j->str : dflt; }
};

class JsonParser {
public:
    explicit JsonParser(const std::string &text) : s_(text) {}
    Json parse()
    {
        Json v = value();
        ws();
        if (p_ != s_.size()) fail("trailing characters");
        return v;
    }

private:
    const std::string &s_;
    size_t p_ = 0;
    [[noreturn]] void fail(const char *what) const { throw std::runtime_error(std::string("JSON: ") + what + " at byte " + std::to_string(p_)); }
    void ws() { while (p_ < s_.size() && (s_[p_] == ' ' || s_[p_] == '\t' || s_[p_] == '\n' || s_[p_] == '\r')) ++p_; }
    bool lit(const char *t)
    {
        size_t n = 0; while (t[n]) ++n;
        if (s_.compare(p_, n, t) == 0) { p_ += n; return true; }
        return false;
    }
    static void utf8(std::string &o, unsigned cp)
    {
        if (cp < 0x80) o += (char)cp;
        else if (cp < 0x800) { o += (char)(0xC0 | (cp >> 6)); o += (char)(0x80 | (cp & 0x3F)); }
        else { o += (char)(0xE0 | (cp >> 12)); o += (char)(0x80 | ((cp >> 6) & 0x3F)); o += (char)(0x80 | (cp & 0x3F)); }
    }
    std::string string()
    {
        if (s_[p_] != '"') fail("expected string");
        ++p_;
        std::string o;
        while (p_ < s_.size() && s_[p_] != '"') {
            char c = s_[p_++];
            if (c != '\\') { o += c; continue; }
            if (p_ >= s_.size()) fail("bad escape");
            char e = s_[p_++];
            switch (e) {
            case '"': o += '"'; break; case '\\': o += '\\'; break; case '/': o += '/'; break;
            case 'b': o += '\b'; break; case 'f': o += '\f'; break; case 'n': o += '\n'; break;
            case 'r': o += '\r'; break; case 't': o += '\t'; break;
            case 'u': {
                if (p_ + 4 > s_.size()) fail("bad \\u escape");
                unsigned cp = (unsigned)strtoul(s_.substr(p_, 4).c_str(), nullptr, 16);
                p_ += 4; utf8(o, cp); break;
            }
            default: fail("bad escape");
            }
        }
        if (p_ >= s_.size()) fail("unterminated string");
        ++p_;
        return o;
    }
    Json value()
    {
        ws();
        if (p_ >= s_.size()) fail("unexpected end");
        Json v;
        char c = s_[p_];
        if (c == '{') {
            v.type = Json::Object; ++p_; ws();
            if (p_ < s_.size() && s_[p_] == '}') { ++p_; return v; }
            for (;;) {
                ws();
                std::string k = string();
                ws();
                if (p_ >= s_.size() || s_[p_] != ':') fail("expected ':'");
                ++p_;
                v.obj.emplace_back(std::move(k), value());
                ws();
                if (p_ < s_.size() && s_[p_] == ',') { ++p_; continue; }
                if (p_ < s_.size() && s_[p_] == '}') { ++p_; break; }
                fail("expected ',' or '}'");
            }
        } else if (c == '[') {
            v.type = Json::Array; ++p_; ws();
            if (p_ < s_.size() && s_[p_] == ']') { ++p_; return v; }
            for (;;) {
                v.arr.push_back(value());
                ws();
                if (p_ < s_.size() && s_[p_] == ',') { ++p_; continue; }
                if (p_ < s_.size() && s_[p_] == ']') { ++p_; break; }
                fail("expected ',' or ']'");
            }
        } else if (c == '"') {
            v.type = Json::String; v.str = string();
        } else if (lit("true")) { v.type = Json::Bool; v.b = true; }
        else if (lit("false")) { v.type = Json::Bool; v.b = false; }
        else if (lit("null")) { v.type = Json::Null; }
        else {
            const char *start = s_.c_str() + p_;
            char *end = nullptr;
            v.num = strtod(start, &end);
            if (end == start) fail("bad value");
            v.type = Json::Number;
            p_ += (size_t)(end - start);
        }
        return v;
    }
};

}  // namespace pth
