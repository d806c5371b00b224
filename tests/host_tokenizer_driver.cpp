// Test infrastructure: drives the engine's HOST-side front end (model_file.cpp + tokenizer.cpp, no HIP) so that the CPU test
// suite can compare it with the oracle.  usage: driver <model.bin> <text>...   prints per text: the 513 prompt ids, then "|",
// then the plain WordPiece ids.
#include "model_file.h"
#include "tokenizer.h"

#include <cstdio>
#include <string>
#include <vector>

int main(int argc, char ** argv) {
    if (argc < 2) return 2;
    barkhip::ModelFile mf;
    std::string err;
    if (!mf.open(argv[1], err)) { fprintf(stderr, "%s\n", err.c_str()); return 1; }
    barkhip::Vocab vocab;
    vocab.build(mf.vocab);
    const barkhip::PromptParams pp;
    for (int i = 2; i < argc; i++) {
        const std::vector<int32_t> prompt = barkhip::build_semantic_prompt(vocab, pp, argv[i], false);
        for (int32_t v : prompt) printf("%d ", v);
        printf("| ");
        std::vector<int32_t> wp(256);
        const int n = barkhip::wordpiece_encode(vocab, argv[i], wp.data(), 256, false);
        for (int k = 0; k < n; k++) printf("%d ", wp[(size_t) k]);
        printf("\n");
    }
    return 0;
}
