// tokenizer.h - host-side text front end of the semantic stage.
// Behavioural contract: /root/reference/bark.cpp:480-662 (utf8_len, strip_accents, bert_tokenize,
// bark_tokenize_input) including its quirks: no lower-casing, no CLS/SEP,
// at most n_max-1 word pieces, unknown bytes are skipped one at a time and force a "##" prefix.
#pragma once
#include <cstdint>
#include <string>
#include <unordered_map>
#include <vector>

namespace barkhip {

struct Vocab {
    std::unordered_map<std::string, int32_t> token_to_id;
    void build(const std::vector<std::string> & id_to_token);
};

// Latin-1 accented letters -> ASCII base letter (the 52 letters of bark.cpp:488-541).
std::string fold_accents(const std::string & text);

// Greedy longest-match WordPiece over regex-split words; writes at most n_max-1 ids.
int wordpiece_encode(const Vocab & vocab, const char * text, int32_t * out, int n_max, bool log_unknown);

struct PromptParams {
    int32_t block_size = 1024, text_encoding_offset = 10048, text_pad_token = 129595,
            semantic_pad_token = 10000, semantic_infer_token = 129599;
};
// 256 text slots (+offset, padded) | 256 semantic-history pads | infer token  -> 513 ids
std::vector<int32_t> build_semantic_prompt(const Vocab & vocab, const PromptParams & p, const char * text, bool log_unknown);

}  // namespace barkhip
