// json_min.hpp — the small part of JSON the bdd_solver configuration needs (objects, arrays, strings, numbers,
// true/false/null).  The reference uses nlohmann::json (src/bdd_solver/bdd_solver.cpp:468-475), which is not in
// this image; the driver only ever does contains(key) / value lookups with defaults.
#pragma once
#include <cctype>
#include <cstdlib>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

namespace bddmma_host {

class json {
public:
    enum kind_t { null_k, bool_k, number_k, string_k, array_k, object_k };
    kind_t kind = null_k;
    bool b = false;
    double num = 0;
    std::string str;
    std::vector<json> arr;
    std::map<std::string, json> obj;

    bool contains(const std::string& k) const { return kind == object_k && obj.count(k) > 0; }
    const json& operator[](const std::string& k) const
    {
        static const json none;
        auto it = obj.find(k);
        return it == obj.end() ? none : it->second;
    }
    bool is_object() const { return kind == object_k; }
    double number_or(const std::string& k, double d) const { return contains(k) && (*this)[k].kind == number_k ? (*this)[k].num : d; }
    bool bool_or(const std::string& k, bool d) const { return contains(k) && (*this)[k].kind == bool_k ? (*this)[k].b : d; }
    std::string string_or(const std::string& k, const std::string& d) const { return contains(k) && (*this)[k].kind == string_k ? (*this)[k].str : d; }

    static json parse(const std::string& s)
    {
        size_t p = 0;
        json v = value(s, p);
        ws(s, p);
        if (p != s.size()) fail(s, p, "trailing characters");
        return v;
    }

private:
    [[noreturn]] static void fail(const std::string& s, size_t p, const char* what)
    {
        throw std::runtime_error(std::string("json: ") + what + " at offset " + std::to_string(p) + " near '" + s.substr(p, 20) + "'");
    }
    static void ws(const std::string& s, size_t& p) { while (p < s.size() && std::isspace((unsigned char)s[p])) ++p; }
    static std::string string(const std::string& s, size_t& p)
    {
        if (s[p] != '"') fail(s, p, "expected string");
        std::string out;
        for (++p; p < s.size() && s[p] != '"'; ++p) {
            if (s[p] != '\\') { out += s[p]; continue; }
            if (++p >= s.size()) break;
            switch (s[p]) {
                case 'n': out += '\n'; break;
                case 't': out += '\t'; break;
                case 'r': out += '\r'; break;
                case 'b': out += '\b'; break;
                case 'f': out += '\f'; break;
                case 'u': {  // \uXXXX: only the Latin-1 range is materialised, the rest becomes '?'
                    if (p + 4 >= s.size()) fail(s, p, "bad \\u escape");
                    const unsigned cp = (unsigned)std::strtoul(s.substr(p + 1, 4).c_str(), nullptr, 16);
                    out += cp < 256 ? (char)cp : '?';
                    p += 4;
                    break;
                }
                default: out += s[p];  // \" \\ \/
            }
        }
        if (p >= s.size()) fail(s, p, "unterminated string");
        ++p;
        return out;
    }
    static json value(const std::string& s, size_t& p)
    {
        ws(s, p);
        if (p >= s.size()) fail(s, p, "unexpected end");
        json v;
        const char c = s[p];
        if (c == '{') {
            v.kind = object_k;
            ++p;
            ws(s, p);
            if (p < s.size() && s[p] == '}') { ++p; return v; }
            for (;;) {
                ws(s, p);
                std::string k = string(s, p);
                ws(s, p);
                if (p >= s.size() || s[p] != ':') fail(s, p, "expected ':'");
                ++p;
                v.obj[k] = value(s, p);
                ws(s, p);
                if (p < s.size() && s[p] == ',') { ++p; continue; }
                if (p < s.size() && s[p] == '}') { ++p; return v; }
                fail(s, p, "expected ',' or '}'");
            }
        }
        if (c == '[') {
            v.kind = array_k;
            ++p;
            ws(s, p);
            if (p < s.size() && s[p] == ']') { ++p; return v; }
            for (;;) {
                v.arr.push_back(value(s, p));
                ws(s, p);
                if (p < s.size() && s[p] == ',') { ++p; continue; }
                if (p < s.size() && s[p] == ']') { ++p; return v; }
                fail(s, p, "expected ',' or ']'");
            }
        }
        if (c == '"') { v.kind = string_k; v.str = string(s, p); return v; }
        if (s.compare(p, 4, "true") == 0) { v.kind = bool_k; v.b = true; p += 4; return v; }
        if (s.compare(p, 5, "false") == 0) { v.kind = bool_k; v.b = false; p += 5; return v; }
        if (s.compare(p, 4, "null") == 0) { p += 4; return v; }
        char* end = nullptr;
        v.num = std::strtod(s.c_str() + p, &end);
        if (end == s.c_str() + p) fail(s, p, "unexpected token");
        v.kind = number_k;
        p = (size_t)(end - s.c_str());
        return v;
    }
};

}  // namespace bddmma_host
