// model_file.h - host-side reader of the reference's single-file model container
// ("ggml_weights.bin").  Layout restated from /root/reference/convert.py:59-110,202-322 and the
// reference loader bark.cpp:664-727,995-1068:
//
//   u32 magic 0x67676d6c | i32 n_vocab | n_vocab x {i32 len, bytes}
//   3 x GPT { 10 x i32 hparams | i32 n_tensors | n_tensors x record }      (semantic, coarse, fine)
//   u32 magic | 9 x i32 codec hparams | records until EOF
//   record = i32 n_dims, i32 name_len, i32 ttype, i32 dims[n_dims] (innermost first), name, raw data
//
// The file is mmapped; TensorRef::data points into the mapping (possibly unaligned).
#pragma once
#include "quant_formats.h"

#include <cstdint>
#include <map>
#include <string>
#include <vector>

namespace barkhip {

struct TensorRef {
    int32_t ttype = 0;                  // ggml_type: 0 f32, 1 f16, or one of the block formats of quant_formats.h
    int32_t n_dims = 0;
    int64_t ne[4] = {1, 1, 1, 1};       // ne[0] innermost
    const uint8_t * data = nullptr;
    int64_t nelements() const { return ne[0] * ne[1] * ne[2] * ne[3]; }
    size_t  nbytes() const {
        if (const QuantFormat * q = quant_format_by_type(ttype)) return (size_t) (nelements() / 32) * (size_t) q->block_bytes;
        return (size_t) nelements() * (ttype == 1 ? 2 : 4);
    }
};

struct GptHparams {   // bark.cpp:700-709 (file order)
    int32_t n_layer = 0, n_head = 0, n_embd = 0, block_size = 0, bias = 0, n_in_vocab = 0, n_out_vocab = 0,
            n_lm_heads = 0, n_wtes = 0, ftype = 0;
};

struct GptSection {
    GptHparams hp;
    std::map<std::string, TensorRef> tensors;
};

struct CodecHparams {   // convert.py:59-79
    int32_t in_channels = 0, hidden_dim = 0, n_filters = 0, kernel_size = 0, residual_kernel_size = 0, n_bins = 0,
            bandwidth = 0, sr = 0, ftype = 0;
};

struct ModelFile {
    const uint8_t * map = nullptr;
    size_t map_size = 0;
    std::vector<std::string> vocab;                 // id -> token
    GptSection gpt[3];                              // semantic, coarse, fine
    CodecHparams codec_hp;
    std::map<std::string, TensorRef> codec;

    ModelFile() = default;
    ModelFile(const ModelFile &) = delete;
    ModelFile & operator=(const ModelFile &) = delete;
    ~ModelFile();
    // Returns false (message in err) when the container itself is malformed (magic, counts, record headers, dims that
    // overflow or exceed the file, unknown types, truncation).  Shapes are checked against the hparams by the engine's loader.
    bool open(const char * path, std::string & err);
};

// bark_model_quantize (bark.cpp:2300-2377): f16/f32 file -> Q4_0 file (quantize.cpp)
bool model_quantize(const char * fname_inp, const char * fname_out, int ftype, std::string & err);

}  // namespace barkhip
