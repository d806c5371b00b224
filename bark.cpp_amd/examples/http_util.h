// http_util.h - the request-side helpers of batch_server.cpp (flat-JSON field access, 32-bit float WAV framing), header-only so that a CPU test
// can drive them without a device (tests/test_host_frontend.py).
#pragma once
#include <cstdint>
#include <cstdlib>
#include <string>

namespace barkhttp {

// the string value of `key` in a flat JSON object (escapes \" \\ \/ \n \t \r \b \f and \uXXXX incl. surrogate pairs are decoded to UTF-8)
inline bool json_string(const std::string & js, const char * key, std::string & out) {
    const std::string pat = std::string("\"") + key + "\"";
    size_t p = js.find(pat);
    if (p == std::string::npos) return false;
    p = js.find(':', p + pat.size());
    if (p == std::string::npos) return false;
    p++;
    while (p < js.size() && (js[p] == ' ' || js[p] == '\t' || js[p] == '\n' || js[p] == '\r')) p++;
    if (p >= js.size() || js[p] != '"') return false;            // the value of `key` must itself be a string ({"text": 12, "x": "y"} has no text)
    out.clear();
    for (size_t i = p + 1; i < js.size(); i++) {
        const char ch = js[i];
        if (ch == '"') return true;
        if (ch != '\\') { out.push_back(ch); continue; }
        if (++i >= js.size()) return false;
        switch (js[i]) {
            case 'n': out.push_back('\n'); break;
            case 't': out.push_back('\t'); break;
            case 'r': out.push_back('\r'); break;
            case 'b': out.push_back('\b'); break;
            case 'f': out.push_back('\f'); break;
            case 'u': {
                if (i + 4 >= js.size()) return false;
                unsigned cp = (unsigned) strtoul(js.substr(i + 1, 4).c_str(), nullptr, 16);
                i += 4;
                if (cp >= 0xD800 && cp < 0xDC00 && i + 6 < js.size() && js[i + 1] == '\\' && js[i + 2] == 'u') {      // surrogate pair -> one code point
                    const unsigned lo = (unsigned) strtoul(js.substr(i + 3, 4).c_str(), nullptr, 16);
                    if (lo >= 0xDC00 && lo < 0xE000) { cp = 0x10000 + ((cp - 0xD800) << 10) + (lo - 0xDC00); i += 6; }
                }
                if (cp >= 0x10000) {
                    out.push_back((char) (0xF0 | (cp >> 18))); out.push_back((char) (0x80 | ((cp >> 12) & 0x3F)));
                    out.push_back((char) (0x80 | ((cp >> 6) & 0x3F))); out.push_back((char) (0x80 | (cp & 0x3F)));
                }
                else if (cp < 0x80) out.push_back((char) cp);
                else if (cp < 0x800) { out.push_back((char) (0xC0 | (cp >> 6))); out.push_back((char) (0x80 | (cp & 0x3F))); }
                else { out.push_back((char) (0xE0 | (cp >> 12))); out.push_back((char) (0x80 | ((cp >> 6) & 0x3F))); out.push_back((char) (0x80 | (cp & 0x3F))); }
                break;
            }
            default: out.push_back(js[i]); break;            // \" \\ \/
        }
    }
    return false;
}
inline bool json_uint(const std::string & js, const char * key, uint32_t & out) {
    const std::string pat = std::string("\"") + key + "\"";
    size_t p = js.find(pat);
    if (p == std::string::npos) return false;
    p = js.find(':', p + pat.size());
    if (p == std::string::npos) return false;
    p++;
    while (p < js.size() && (js[p] == ' ' || js[p] == '\t')) p++;
    if (p >= js.size() || js[p] < '0' || js[p] > '9') return false;
    out = (uint32_t) strtoul(js.c_str() + p, nullptr, 10);
    return true;
}

inline std::string wav_f32(const float * pcm, int n, int rate) {
    auto u32 = [](std::string & s, uint32_t v) { s.append(reinterpret_cast<const char *>(&v), 4); };
    auto u16 = [](std::string & s, uint16_t v) { s.append(reinterpret_cast<const char *>(&v), 2); };
    std::string s;
    const uint32_t bytes = (uint32_t) n * 4;
    s += "RIFF"; u32(s, 36 + bytes); s += "WAVE";
    s += "fmt "; u32(s, 16); u16(s, 3 /* IEEE float */); u16(s, 1); u32(s, (uint32_t) rate); u32(s, (uint32_t) rate * 4); u16(s, 4); u16(s, 32);
    s += "data"; u32(s, bytes);
    s.append(reinterpret_cast<const char *>(pcm), bytes);
    return s;
}

}  // namespace barkhttp
