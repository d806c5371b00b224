// quant_formats.h - the ggml block formats bark_model_quantize can write (examples/quantize/main.cpp:29-35 lists q4_0, q4_1,
// q5_0, q5_1, q8_0) and the engine can read.  Numbers are ggml's public enums (enum ggml_type / enum ggml_ftype) and block
// structs (block_q4_0 ... block_q8_0, QK = 32), restated; SURVEY.md A.4 item 6.
//
//   type   ggml_type ggml_ftype  block bytes  layout                                   value of element j
//   q4_0       2         2           18       f16 d | 16 B nibbles                     (n_j - 8) d
//   q4_1       3         3           20       f16 d | f16 m | 16 B nibbles             n_j d + m
//   q5_0       6         8           22       f16 d | u32 qh | 16 B nibbles            ((n_j | h_j << 4) - 16) d
//   q5_1       7         9           24       f16 d | f16 m | u32 qh | 16 B nibbles    (n_j | h_j << 4) d + m
//   q8_0       8         7           34       f16 d | 32 x int8                        q_j d
// Nibble byte i holds element i (low) and element i + 16 (high); bit j of qh is the fifth bit of element j.
#pragma once
#include <cstdint>

namespace barkhip {

enum QuantId : int { QT_Q4_0 = 0, QT_Q4_1 = 1, QT_Q5_0 = 2, QT_Q5_1 = 3, QT_Q8_0 = 4, QT_COUNT = 5,
                     QT_F32 = 16 };      // not a block format: plain f32 weights (f32 model files) travelling through the same QMat handle

struct QuantFormat {
    int id; int ggml_type; int ggml_ftype; int block_bytes; bool has_min; bool has_high_bits; int qs_bytes; const char * name;
};

inline const QuantFormat * quant_formats() {
    static const QuantFormat f[QT_COUNT] = {
        {QT_Q4_0, 2, 2, 18, false, false, 16, "q4_0"},
        {QT_Q4_1, 3, 3, 20, true,  false, 16, "q4_1"},
        {QT_Q5_0, 6, 8, 22, false, true,  16, "q5_0"},
        {QT_Q5_1, 7, 9, 24, true,  true,  16, "q5_1"},
        {QT_Q8_0, 8, 7, 34, false, false, 32, "q8_0"},
    };
    return f;
}
inline const QuantFormat * quant_format_by_type(int ggml_type) {
    for (int i = 0; i < QT_COUNT; i++) if (quant_formats()[i].ggml_type == ggml_type) return quant_formats() + i;
    return nullptr;
}
inline const QuantFormat * quant_format_by_ftype(int ggml_ftype) {
    for (int i = 0; i < QT_COUNT; i++) if (quant_formats()[i].ggml_ftype == ggml_ftype) return quant_formats() + i;
    return nullptr;
}

}  // namespace barkhip
