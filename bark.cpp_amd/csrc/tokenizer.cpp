// tokenizer.cpp - see tokenizer.h
#include "tokenizer.h"

#include <algorithm>
#include <cstdio>
#include <regex>

namespace barkhip {

void Vocab::build(const std::vector<std::string> & id_to_token) {
    token_to_id.clear();
    token_to_id.reserve(id_to_token.size() * 2);
    // later duplicates overwrite earlier ones, as std::map::operator[] does in the reference loader (bark.cpp:664-690)
    for (size_t i = 0; i < id_to_token.size(); i++) token_to_id[id_to_token[i]] = (int32_t) i;
}

static inline size_t utf8_span(unsigned char lead) {
    // sequence length from the high nibble of the lead byte (bark.cpp:480-484); continuation bytes count as 1
    return lead < 0xC0 ? 1 : lead < 0xE0 ? 2 : lead < 0xF0 ? 3 : 4;
}

std::string fold_accents(const std::string & text) {
    // All 52 mapped letters are two-byte sequences 0xC3 0x80..0xBF; second byte -> base letter (0 = unmapped).
    static const char kBase[64] = {
        /*80*/ 'A', 'A', 'A', 'A', 'A', 'A', 0,  'C', 'E', 'E', 'E', 'E', 'I', 'I', 'I', 'I',
        /*90*/ 0,  'N', 'O', 'O', 'O', 'O', 'O', 0,  0,  'U', 'U', 'U', 'U', 'Y', 0,  0,
        /*a0*/ 'a', 'a', 'a', 'a', 'a', 'a', 0,  'c', 'e', 'e', 'e', 'e', 'i', 'i', 'i', 'i',
        /*b0*/ 0,  'n', 'o', 'o', 'o', 'o', 'o', 0,  0,  'u', 'u', 'u', 'u', 'y', 0,  0};
    std::string out;
    out.reserve(text.size());
    for (size_t i = 0; i < text.size();) {
        const unsigned char lead = (unsigned char) text[i];
        const size_t n = utf8_span(lead);
        if (n == 2 && lead == 0xC3 && i + 1 < text.size()) {
            const unsigned char b = (unsigned char) text[i + 1];
            if (b >= 0x80 && b <= 0xBF && kBase[b - 0x80]) { out.push_back(kBase[b - 0x80]); i += 2; continue; }
        }
        out.append(text, i, n);      // std::string::append clamps at the end like substr does
        i += n;
    }
    return out;
}

int wordpiece_encode(const Vocab & vocab, const char * text, int32_t * out, int n_max, bool log_unknown) {
    std::string rest = fold_accents(text);
    std::vector<std::string> words;
    {
        // bark.cpp:575 - one punctuation mark, or a run of letters, or a run of digits (C locale, bytes)
        static const std::regex splitter(R"([[:punct:]]|[[:alpha:]]+|[[:digit:]]+)");
        std::smatch m;
        while (std::regex_search(rest, m, splitter)) {
            words.push_back(m.str(0));
            rest = m.suffix();
        }
    }
    int n_out = 0;
    for (const std::string & w : words) {
        if (w.empty()) continue;
        const int len = (int) w.size();
        bool continuation = false;
        int pos = 0;
        while (pos < len && n_out < n_max - 1) {                // budget check per piece (bark.cpp:598-599)
            int end = len;
            int32_t id = -1;
            for (; end > pos; --end) {
                std::string key = continuation ? "##" : "";
                key.append(w, (size_t) pos, (size_t) (end - pos));
                auto it = vocab.token_to_id.find(key);
                if (it != vocab.token_to_id.end()) { id = it->second; break; }
            }
            if (id >= 0) { out[n_out++] = id; pos = end; }
            else {
                if (log_unknown) fprintf(stderr, "bert_tokenize: unknown token '%c'\n", w[(size_t) pos]);
                pos += 1;                                        // skip one byte (bark.cpp:611-615)
            }
            continuation = true;
        }
    }
    return n_out;
}

std::vector<int32_t> build_semantic_prompt(const Vocab & vocab, const PromptParams & p, const char * text, bool log_unknown) {
    const int slots = std::min(p.block_size, 256);
    std::vector<int32_t> ids((size_t) slots, 0);
    const int n = wordpiece_encode(vocab, text, ids.data(), slots, log_unknown);
    for (int i = 0; i < slots; i++) ids[(size_t) i] = i < n ? ids[(size_t) i] + p.text_encoding_offset : p.text_pad_token;
    ids.insert(ids.end(), 256, p.semantic_pad_token);
    ids.push_back(p.semantic_infer_token);
    return ids;
}

}  // namespace barkhip
