// quantize.cpp - bark_model_quantize: rewrites an f16/f32 model file with the GPT matmul weights in one of ggml's block
// formats (q4_0 is the input of BASELINE config 4; q4_1, q5_0, q5_1, q8_0 are the other types examples/quantize/main.cpp:29-35
// offers).  Behavioural contract: /root/reference/bark.cpp:272-478 (which tensors, record layout, ftype encoding) and
// :2234-2377 (vocab copied, three GPT sections quantised, codec copied verbatim).  The block arithmetic restates ggml's public
// reference quantisers (quantize_row_q*_ref), per 32 weights:
//   q4_0  d = max/-8  (max = the element of largest magnitude, sign kept)   q = min(15, (int8)(x/d + 8.5))
//   q4_1  d = (max - min)/15, m = min                                       q = min(15, (int8)((x - m)/d + 0.5))
//   q5_0  d = max/-16                                                        q = min(31, (int8)(x/d + 16.5)), fifth bit -> qh
//   q5_1  d = (max - min)/31, m = min                                       q = (uint8)((x - m)/d + 0.5),    fifth bit -> qh
//   q8_0  d = amax/127                                                       q = roundf(x/d)
// (x/d is evaluated as x * (1/d); layouts in quant_formats.h).  Pure host code; the HIP engine reads
// the result.
#include "model_file.h"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <regex>
#include <string>
#include <vector>

namespace barkhip {

namespace {
constexpr int kQK = 32;
constexpr int32_t kTypeF32 = 0, kTypeF16 = 1;
constexpr int32_t kQntVersion = 2, kQntFactor = 1000;                     // ggml: GGML_QNT_VERSION, GGML_QNT_VERSION_FACTOR

struct In {
    const uint8_t * p; size_t n; size_t pos = 0; bool ok = true;
    template <typename T> T get() { T v{}; if (pos + sizeof(T) > n) { ok = false; return v; } memcpy(&v, p + pos, sizeof(T)); pos += sizeof(T); return v; }
    const uint8_t * take(size_t k) { if (k > n - pos) { ok = false; return p; } const uint8_t * r = p + pos; pos += k; return r; }
};
struct Out {
    FILE * f; bool ok = true;
    void put(const void * d, size_t k) { if (k && fwrite(d, 1, k, f) != k) ok = false; }
    template <typename T> void val(T v) { put(&v, sizeof(T)); }
};

inline float h2f(uint16_t h) { return (float) __builtin_bit_cast(_Float16, h); }
inline uint16_t f2h(float f) { return __builtin_bit_cast(uint16_t, (_Float16) f); }

// one row of n floats (n % 32 == 0) -> n/32 blocks of fmt.block_bytes
void quantize_row(const QuantFormat & fmt, const float * x, uint8_t * out, int n) {
    for (int b = 0; b < n / kQK; b++) {
        const float * xb = x + b * kQK;
        uint8_t * ob = out + (size_t) b * (size_t) fmt.block_bytes;
        uint8_t * qs = ob + fmt.block_bytes - fmt.qs_bytes;
        if (fmt.id == QT_Q8_0) {
            float amax = 0.0f;
            for (int j = 0; j < kQK; j++) amax = std::max(amax, fabsf(xb[j]));
            const float d = amax / 127;
            const float id = d ? 1.0f / d : 0.0f;
            const uint16_t dh = f2h(d);
            memcpy(ob, &dh, 2);
            for (int j = 0; j < kQK; j++) qs[j] = (uint8_t) (int8_t) roundf(xb[j] * id);
            continue;
        }
        const int levels = fmt.has_high_bits ? 32 : 16;
        float d, base = 0.0f, off;
        if (fmt.has_min) {
            float mn = FLT_MAX, mx = -FLT_MAX;
            for (int j = 0; j < kQK; j++) { const float v = xb[j]; if (v < mn) mn = v; if (v > mx) mx = v; }
            d = (mx - mn) / (float) (levels - 1);
            base = mn; off = 0.5f;
            const uint16_t dh = f2h(d), mh = f2h(mn);
            memcpy(ob, &dh, 2); memcpy(ob + 2, &mh, 2);
        } else {
            float amax = 0.0f, mx = 0.0f;
            for (int j = 0; j < kQK; j++) { const float v = xb[j]; if (amax < fabsf(v)) { amax = fabsf(v); mx = v; } }
            d = mx / (float) -(levels / 2);
            off = (float) (levels / 2) + 0.5f;
            const uint16_t dh = f2h(d);
            memcpy(ob, &dh, 2);
        }
        const float id = d ? 1.0f / d : 0.0f;
        uint32_t qh = 0;
        for (int j = 0; j < kQK / 2; j++) {
            const float x0 = (fmt.has_min ? xb[j] - base : xb[j]) * id, x1 = (fmt.has_min ? xb[kQK / 2 + j] - base : xb[kQK / 2 + j]) * id;
            uint8_t q0, q1;
            if (fmt.id == QT_Q5_1) { q0 = (uint8_t) (x0 + off); q1 = (uint8_t) (x1 + off); }           // ggml does not clamp here
            else { q0 = (uint8_t) std::min(levels - 1, (int) (int8_t) (x0 + off)); q1 = (uint8_t) std::min(levels - 1, (int) (int8_t) (x1 + off)); }
            qs[j] = (uint8_t) ((q0 & 0x0F) | ((q1 & 0x0F) << 4));
            qh |= (uint32_t) ((q0 & 0x10u) >> 4) << (j + 0);
            qh |= (uint32_t) ((q1 & 0x10u) >> 4) << (j + kQK / 2);
        }
        if (fmt.has_high_bits) memcpy(ob + (fmt.has_min ? 4 : 2), &qh, 4);
    }
}
}  // namespace

bool model_quantize(const char * fname_inp, const char * fname_out, int ftype, std::string & err) {
    const QuantFormat * fmt = quant_format_by_ftype(ftype);
    if (!fmt) { err = "unsupported ggml_ftype " + std::to_string(ftype) + " (q4_0, q4_1, q5_0, q5_1, q8_0 are implemented)"; return false; }
    ModelFile mf;                                  // validates the input and maps it
    if (!mf.open(fname_inp, err)) return false;
    In in{mf.map, mf.map_size};
    FILE * f = fopen(fname_out, "wb");
    if (!f) { err = std::string("cannot open '") + fname_out + "' for writing"; return false; }
    Out out{f};
    // magic + vocabulary are copied (bark.cpp:2318-2345)
    out.val(in.get<uint32_t>());
    const int32_t n_vocab = in.get<int32_t>();
    out.val(n_vocab);
    for (int i = 0; i < n_vocab; i++) { const uint32_t len = in.get<uint32_t>(); out.val(len); out.put(in.take(len), len); }
    // tensors to quantise (bark.cpp:2283-2290), 2-D only (:372-373)
    static const std::regex to_quant[] = {std::regex("model/wte/.*"), std::regex("model/lm_head/.*"), std::regex("model/h.*/attn/c_attn/w"),
                                          std::regex("model/h.*/attn/c_proj/w"), std::regex("model/h.*/mlp/c_fc/w"), std::regex("model/h.*/mlp/c_proj/w")};
    std::vector<float> row;
    std::vector<uint8_t> qrow;
    for (int g = 0; g < 3 && in.ok; g++) {
        int32_t hp[10];
        for (int & v : hp) v = in.get<int32_t>();
        hp[9] = kQntVersion * kQntFactor + ftype;                                      // bark.cpp:2254
        out.put(hp, sizeof(hp));
        const int32_t n_tensors = in.get<int32_t>();
        out.val(n_tensors);
        for (int t = 0; t < n_tensors && in.ok; t++) {
            const int32_t n_dims = in.get<int32_t>(), name_len = in.get<int32_t>();
            int32_t ttype = in.get<int32_t>();
            int32_t ne[4] = {1, 1, 1, 1};
            for (int i = 0; i < n_dims && i < 4; i++) ne[i] = in.get<int32_t>();
            const std::string name((const char *) in.take((size_t) name_len), (size_t) name_len);
            const size_t nel = (size_t) ne[0] * ne[1] * ne[2] * ne[3];
            const size_t bpe = ttype == kTypeF16 ? 2 : 4;
            if (const QuantFormat * have = quant_format_by_type(ttype)) {
                err = "tensor '" + name + "' is already quantised (" + have->name + "): quantise the f16 / f32 file instead";
                fclose(f); return false;
            }
            const uint8_t * data = in.take(nel * bpe);
            if (!in.ok) break;
            bool quantize = false;
            for (const auto & re : to_quant) if (std::regex_match(name, re)) { quantize = true; break; }
            quantize = quantize && n_dims == 2;
            if (quantize && (ttype != kTypeF32 && ttype != kTypeF16)) { err = "tensor '" + name + "' is already quantised"; fclose(f); return false; }
            if (quantize && ne[0] % kQK != 0) { err = "tensor '" + name + "': row length is not a multiple of 32"; fclose(f); return false; }
            const int32_t otype = quantize ? fmt->ggml_type : ttype;
            out.val(n_dims); out.val(name_len); out.val(otype);
            for (int i = 0; i < n_dims; i++) out.val(ne[i]);
            out.put(name.data(), name.size());
            if (!quantize) { out.put(data, nel * bpe); continue; }
            row.resize((size_t) ne[0]); qrow.resize((size_t) ne[0] / kQK * (size_t) fmt->block_bytes);
            for (int r = 0; r < ne[1]; r++) {
                for (int k = 0; k < ne[0]; k++) {
                    if (ttype == kTypeF16) { uint16_t h; memcpy(&h, data + ((size_t) r * ne[0] + k) * 2, 2); row[(size_t) k] = h2f(h); }
                    else memcpy(&row[(size_t) k], data + ((size_t) r * ne[0] + k) * 4, 4);
                }
                quantize_row(*fmt, row.data(), qrow.data(), ne[0]);
                out.put(qrow.data(), qrow.size());
            }
        }
    }
    if (!in.ok) { err = "truncated input"; fclose(f); return false; }
    out.put(in.p + in.pos, in.n - in.pos);           // codec: copied verbatim, never quantised (bark.cpp:2366-2371)
    const bool ok = out.ok && fclose(f) == 0;
    if (!ok) err = "write error";
    return ok;
}

}  // namespace barkhip
