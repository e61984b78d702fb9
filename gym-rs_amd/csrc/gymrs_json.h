// gymrs_json.h -- the little JSON the serde view needs (host only): a writer that formats numbers the way
// serde_json does (shortest decimal that round-trips the f64, a trailing ".0" on integral values, non-finite as
// null) and a recursive-descent reader for the flat objects that writer produces.  No dependency on the engine.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <string>
#include <vector>

namespace gymrs {
namespace json {

// serde_json prints an f64 with ryu's "pretty" layout (the `ryu` crate, src/pretty/mod.rs; recalled, the crate is not in this
// image): the shortest digit string d1..dn that parses back to the same double, value = 0.d1..dn x 10^kk, laid out as
//   0 <= kk - n and kk <= 16    digits, zeros, ".0"        10.0, 200.0, 100000.0, 1000000000000000.0
//   0 < kk <= 16                point inside               9.8, 0.5 is the next case, 12.34
//   -5 < kk <= 0                "0." zeros digits          0.02, 0.0025, 0.00001234
//   otherwise                   d1[.d2..dn]e(kk-1)         1e16, 1.5e-7, 1e-6   (no '+', no leading zeros)
// NaN and the infinities become null (serde_json's to_string).
inline std::string number(double v)
{
    if (!std::isfinite(v)) return "null";
    if (v == 0.0) return std::signbit(v) ? "-0.0" : "0.0";
    char buf[48];
    for (int prec = 0; prec <= 16; ++prec) { // shortest "%.{prec}e" that round-trips
        std::snprintf(buf, sizeof(buf), "%.*e", prec, v);
        if (std::strtod(buf, nullptr) == v) break;
    }
    std::string digits;
    const char* p = buf;
    const bool neg = *p == '-';
    if (neg) ++p;
    for (; *p && *p != 'e'; ++p)
        if (*p != '.') digits += *p;
    const int exp10 = std::atoi(p + 1);                  // value = d1.d2..dn x 10^exp10
    while (digits.size() > 1 && digits.back() == '0') digits.pop_back();
    const int n = (int)digits.size(), kk = exp10 + 1;    // value = 0.d1..dn x 10^kk
    std::string out = neg ? "-" : "";
    if (kk >= n && kk <= 16) {
        out += digits + std::string((size_t)(kk - n), '0') + ".0";
    } else if (kk > 0 && kk <= 16) {
        out += digits.substr(0, (size_t)kk) + "." + digits.substr((size_t)kk);
    } else if (kk > -5 && kk <= 0) {
        out += "0." + std::string((size_t)(-kk), '0') + digits;
    } else {
        out += digits.substr(0, 1);
        if (n > 1) out += "." + digits.substr(1);
        out += "e" + std::to_string(kk - 1);
    }
    return out;
}

inline std::string quoted(const char* text)
{
    std::string s = "\"";
    for (const char* p = text; *p; ++p) {
        if (*p == '"' || *p == '\\') s += '\\';
        s += *p;
    }
    return s + "\"";
}

// Builds one object: fields in call order, like serde's derive.
class Object {
public:
    Object& raw(const char* key, const std::string& value)
    {
        body_ += (body_.empty() ? "" : ",") + quoted(key) + ":" + value;
        return *this;
    }
    Object& num(const char* key, double v) { return raw(key, number(v)); }
    Object& uint(const char* key, unsigned long long v) { return raw(key, std::to_string(v)); }
    Object& str(const char* key, const char* v) { return raw(key, quoted(v)); }
    Object& null(const char* key) { return raw(key, "null"); }
    Object& obj(const char* key, const Object& o) { return raw(key, o.text()); }
    std::string text() const { return "{" + body_ + "}"; }

private:
    std::string body_;
};

// ---- reader -------------------------------------------------------------------------------------
struct Value {
    enum Kind { Null, Bool, Number, String, Array, ObjectK } kind = Null;
    double number = 0.0;
    bool boolean = false;
    std::string string;
    std::vector<Value> items;                          // Array
    std::vector<std::pair<std::string, Value>> fields; // ObjectK, in document order
    const Value* get(const char* key) const
    {
        for (const auto& f : fields)
            if (f.first == key) return &f.second;
        return nullptr;
    }
};

class Parser {
public:
    explicit Parser(const char* text) : p_(text) {}
    bool parse(Value& out)
    {
        if (!value(out, 0)) return false;
        ws();
        return *p_ == '\0';
    }
    const char* where() const { return p_; }

private:
    const char* p_;
    void ws()
    {
        while (*p_ == ' ' || *p_ == '\t' || *p_ == '\n' || *p_ == '\r') ++p_;
    }
    bool lit(const char* word)
    {
        const size_t n = std::strlen(word);
        if (std::strncmp(p_, word, n) != 0) return false;
        p_ += n;
        return true;
    }
    bool str(std::string& out)
    {
        if (*p_ != '"') return false;
        ++p_;
        out.clear();
        while (*p_ && *p_ != '"') {
            if (*p_ == '\\') {
                ++p_;
                switch (*p_) {
                case 'n': out += '\n'; break;
                case 't': out += '\t'; break;
                case 'r': out += '\r'; break;
                case 'b': out += '\b'; break;
                case 'f': out += '\f'; break;
                case 'u': // keep the escape verbatim: field names and enum variants here are ASCII
                    out += "\\u";
                    break;
                case '\0': return false;
                default: out += *p_; break;
                }
                ++p_;
            } else {
                out += *p_++;
            }
        }
        if (*p_ != '"') return false;
        ++p_;
        return true;
    }
    bool value(Value& out, int depth)
    {
        if (depth > 32) return false;
        ws();
        if (*p_ == '{') {
            ++p_;
            out.kind = Value::ObjectK;
            ws();
            if (*p_ == '}') {
                ++p_;
                return true;
            }
            for (;;) {
                ws();
                std::string key;
                if (!str(key)) return false;
                ws();
                if (*p_ != ':') return false;
                ++p_;
                Value v;
                if (!value(v, depth + 1)) return false;
                out.fields.emplace_back(std::move(key), std::move(v));
                ws();
                if (*p_ == ',') {
                    ++p_;
                    continue;
                }
                if (*p_ == '}') {
                    ++p_;
                    return true;
                }
                return false;
            }
        }
        if (*p_ == '[') {
            ++p_;
            out.kind = Value::Array;
            ws();
            if (*p_ == ']') {
                ++p_;
                return true;
            }
            for (;;) {
                Value v;
                if (!value(v, depth + 1)) return false;
                out.items.push_back(std::move(v));
                ws();
                if (*p_ == ',') {
                    ++p_;
                    continue;
                }
                if (*p_ == ']') {
                    ++p_;
                    return true;
                }
                return false;
            }
        }
        if (*p_ == '"') {
            out.kind = Value::String;
            return str(out.string);
        }
        if (lit("null")) {
            out.kind = Value::Null;
            return true;
        }
        if (lit("true")) {
            out.kind = Value::Bool;
            out.boolean = true;
            return true;
        }
        if (lit("false")) {
            out.kind = Value::Bool;
            return true;
        }
        char* end = nullptr;
        const double v = std::strtod(p_, &end);
        if (end == p_) return false;
        p_ = end;
        out.kind = Value::Number;
        out.number = v;
        return true;
    }
};

} // namespace json
} // namespace gymrs
