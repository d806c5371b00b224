// Drives bark.cpp_amd/examples/http_util.h on the CPU: argv[1] = a JSON body; prints "text=<hex of the decoded string>" or "text=NONE", "seed=<n>" or
// "seed=NONE", and the hex of a WAV framing of three samples.
#include "http_util.h"

#include <cstdio>

int main(int argc, char ** argv) {
    const std::string body = argc > 1 ? argv[1] : "";
    std::string text; uint32_t seed = 0;
    if (barkhttp::json_string(body, "text", text)) { printf("text="); for (unsigned char ch : text) printf("%02x", ch); printf("\n"); } else printf("text=NONE\n");
    if (barkhttp::json_uint(body, "seed", seed)) printf("seed=%u\n", seed); else printf("seed=NONE\n");
    const float pcm[3] = {0.0f, 0.5f, -1.0f};
    const std::string wav = barkhttp::wav_f32(pcm, 3, 24000);
    printf("wav="); for (unsigned char ch : wav) printf("%02x", ch); printf("\n");
    return 0;
}
