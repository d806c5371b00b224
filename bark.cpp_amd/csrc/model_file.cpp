// model_file.cpp - see model_file.h
#include "model_file.h"

#include <cstring>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

namespace barkhip {

namespace {
constexpr uint32_t kMagic = 0x67676d6c;   // bark.cpp:1095-1102

struct Cursor {
    const uint8_t * base; size_t size; size_t pos = 0; bool ok = true;
    template <typename T> T get() {
        T v{};
        if (pos + sizeof(T) > size) { ok = false; return v; }
        memcpy(&v, base + pos, sizeof(T)); pos += sizeof(T);
        return v;
    }
    const uint8_t * take(size_t n) {
        if (n > size - pos) { ok = false; return base; }
        const uint8_t * p = base + pos; pos += n; return p;
    }
};

bool read_record(Cursor & c, std::string & name, TensorRef & t, std::string & err) {
    t = TensorRef();
    t.n_dims = c.get<int32_t>();
    const int32_t name_len = c.get<int32_t>();
    t.ttype = c.get<int32_t>();
    if (!c.ok || t.n_dims < 0 || t.n_dims > 4 || name_len < 0 || name_len > 1024) { err = "corrupt tensor record header"; return false; }
    int64_t count = 1;
    for (int i = 0; i < t.n_dims; i++) {
        t.ne[i] = c.get<int32_t>();
        if (t.ne[i] <= 0) { err = "corrupt tensor dims"; return false; }
        // the element count can never exceed what the file could hold (>= 0.5 byte per element): bounds it far below 2^63
        if (t.ne[i] > (int64_t) (2 * c.size) / count) { err = "tensor dims exceed the file size"; return false; }
        count *= t.ne[i];
    }
    const uint8_t * nm = c.take((size_t) name_len);
    if (!c.ok) { err = "truncated tensor name"; return false; }
    name.assign((const char *) nm, (size_t) name_len);
    const QuantFormat * qf = quant_format_by_type(t.ttype);
    if (t.ttype != 0 && t.ttype != 1 && !qf) {
        err = "tensor '" + name + "' has ggml type " + std::to_string(t.ttype) + ": only f32, f16, q4_0, q4_1, q5_0, q5_1 and q8_0 are supported";
        return false;
    }
    if (qf && (t.ne[0] % 32) != 0) { err = std::string(qf->name) + " tensor '" + name + "' has a row length that is not a multiple of 32"; return false; }
    t.data = c.take(t.nbytes());
    if (!c.ok) { err = "truncated data of tensor '" + name + "'"; return false; }
    return true;
}
}  // namespace

ModelFile::~ModelFile() {
    if (map) munmap((void *) map, map_size);
}

bool ModelFile::open(const char * path, std::string & err) {
    int fd = ::open(path, O_RDONLY);
    if (fd < 0) { err = std::string("cannot open '") + path + "'"; return false; }
    struct stat st;
    if (fstat(fd, &st) != 0 || st.st_size < 16) { close(fd); err = "cannot stat model file"; return false; }
    void * p = mmap(nullptr, (size_t) st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
    close(fd);
    if (p == MAP_FAILED) { err = "mmap failed"; return false; }
    map = (const uint8_t *) p; map_size = (size_t) st.st_size;

    Cursor c{map, map_size};
    if (c.get<uint32_t>() != kMagic) { err = "bad magic (not a bark ggml file)"; return false; }

    const int32_t n_vocab = c.get<int32_t>();                 // bark.cpp:664-690
    if (!c.ok || n_vocab < 0 || n_vocab > (1 << 24)) { err = "bad vocabulary size"; return false; }
    vocab.resize((size_t) n_vocab);
    for (int i = 0; i < n_vocab; i++) {
        const uint32_t len = c.get<uint32_t>();
        const uint8_t * w = c.take(len);
        if (!c.ok) { err = "truncated vocabulary"; return false; }
        vocab[(size_t) i].assign((const char *) w, len);
    }

    static const char * kNames[3] = {"semantic", "coarse", "fine"};
    for (int g = 0; g < 3; g++) {
        GptHparams & hp = gpt[g].hp;
        int32_t * f = &hp.n_layer;
        for (int i = 0; i < 10; i++) f[i] = c.get<int32_t>();
        if (!c.ok) { err = std::string("truncated hparams of ") + kNames[g]; return false; }
        // bark.cpp:711,727,2254 - quantised files carry 2000 + ggml_ftype
        // (GGML_QNT_VERSION_FACTOR = 1000; ggml_ftype 0 f32, 1 f16, 2 mostly q4_0)
        if (hp.ftype < 0 || (hp.ftype % 1000 > 1 && !quant_format_by_ftype(hp.ftype % 1000))) {
            err = std::string(kNames[g]) + ": model ftype " + std::to_string(hp.ftype) + " is not supported (f32, f16, q4_0, q4_1, q5_0, q5_1, q8_0)";
            return false;
        }
        if (hp.n_layer <= 0 || hp.n_layer > 128 || hp.n_head <= 0 || hp.n_embd <= 0 || hp.n_embd % hp.n_head != 0 ||
            hp.block_size <= 0 || hp.n_wtes <= 0 || hp.n_lm_heads <= 0) { err = std::string("implausible hparams of ") + kNames[g]; return false; }
        const int32_t n_tensors = c.get<int32_t>();
        if (!c.ok || n_tensors < 0 || n_tensors > 100000) { err = "bad tensor count"; return false; }
        for (int i = 0; i < n_tensors; i++) {
            std::string name; TensorRef t;
            if (!read_record(c, name, t, err)) return false;
            gpt[g].tensors[name] = t;
        }
    }

    if (c.get<uint32_t>() != kMagic) { err = "bad codec magic"; return false; }      // convert.py:303
    int32_t * f = &codec_hp.in_channels;
    for (int i = 0; i < 9; i++) f[i] = c.get<int32_t>();
    if (!c.ok) { err = "truncated codec hparams"; return false; }
    while (c.pos < c.size) {                                                          // convert.py:189-197 (no count)
        std::string name; TensorRef t;
        if (!read_record(c, name, t, err)) return false;
        codec[name] = t;
    }
    return true;
}

}  // namespace barkhip
