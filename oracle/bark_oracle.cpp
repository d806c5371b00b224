// =====================================================================================
//  bark_oracle.cpp — CPU ORACLE.  TEST INFRASTRUCTURE, NOT PRODUCT CODE.
//
//  A from-scratch CPU restatement of the one hot path this repository accelerates:
//  the reference's semantic / coarse / fine GPT stages, its sampling, its stage loops, its
//  tokenizer, and the EnCodec decoder it calls.  Only tests/, __graft_entry__.smoke() and the
//  `cpu_baseline` leg of bench.py may load this library; the HIP engine never links, loads or
//  calls it.
//
//  PARITY UNPINNED: the reference (PABannier/bark.cpp) ships no tests, golden vectors or
//  fixtures for this path, and it cannot be compiled here because its tensor
//  runtime (ggml) and its codec (PABannier/encodec.cpp, git submodule `encodec.cpp`, pinned
//  commit not recoverable) are absent from /root/reference.  This file therefore follows
//    * /root/reference/bark.cpp line by line for everything bark.cpp itself decides
//      (each function cites the lines it restates), and
//    * the published algorithms of the two absent dependencies: ggml's CPU operator
//      semantics (f16 weights x f16-rounded activations with f32 accumulation, f16-LUT tanh
//      GELU, double-accumulated LayerNorm / softmax sums) and the EnCodec 24 kHz decoder
//      (HF transformers modeling_encodec.py:82-450, the model convert.py converts from).
//  What it IS checked against (tests/test_oracle_golden.py, fixtures made by tools/make_hf_golden.py from HF transformers' Bark /
//  EnCodec - the PyTorch models convert.py converts from - on the same synthetic weights): forward passes of all three GPTs at toy,
//  bark-small and bark-large shapes (2.4e-6 on the logits with ggml's rounding points switched off, f16 rounding noise with them
//  on), a greedy semantic loop, HF's own coarse and fine `generate` loops id for id (incl. more than 1024 frames), and the EnCodec
//  decoder at its real dimensions (1.8e-6 of full scale).  That pins architecture, layouts, stage loops and windowing; it cannot pin
//  ggml's rounding points or its summation order - tests/test_order_sensitivity.py measures how much those matter.
//
//  Build: see oracle/Makefile (g++ -O3 -mavx2 -mfma -mf16c -fopenmp, -ffp-contract=off).
// =====================================================================================
#include <algorithm>
#include <cassert>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <random>
#include <regex>
#include <string>
#include <vector>

#include <fcntl.h>
#include <immintrin.h>
#include <omp.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include "mfma_f16_emu.h"

namespace {

// ------------------------------------------------------------------------------------
// small utilities
// ------------------------------------------------------------------------------------
static inline int64_t now_us() {
    return std::chrono::duration_cast<std::chrono::microseconds>(
               std::chrono::steady_clock::now().time_since_epoch()).count();
}

// Large buffers come from pre-faulted anonymous mappings: first-touch faults are very slow in
// the VMs this runs in.
static void * big_alloc(size_t nbytes) {
    if (nbytes == 0) nbytes = 64;
    void * p = mmap(nullptr, nbytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_POPULATE, -1, 0);
    if (p == MAP_FAILED) { fprintf(stderr, "oracle: mmap(%zu) failed\n", nbytes); abort(); }
    return p;
}
struct BigBuf {
    float * p = nullptr; size_t n = 0;
    void ensure(size_t count) {
        if (count <= n) return;
        if (p) munmap(p, n * sizeof(float));
        count = std::max(count, n + n / 2);     // geometric growth: attention scratch grows by one row per decode step
        p = (float *) big_alloc(count * sizeof(float)); n = count;
    }
    ~BigBuf() { if (p) munmap(p, n * sizeof(float)); }
};

static inline float    h2f(uint16_t h) { return _cvtsh_ss(h); }
static inline uint16_t f2h(float f)    { return _cvtss_sh(f, _MM_FROUND_TO_NEAREST_INT | _MM_FROUND_NO_EXC); }
static inline float    round_h(float f) { return h2f(f2h(f)); }

static inline float ld_f32(const uint8_t * p, size_t i) { float v; memcpy(&v, p + 4 * i, 4); return v; }
static inline uint16_t ld_u16(const uint8_t * p, size_t i) { uint16_t v; memcpy(&v, p + 2 * i, 2); return v; }

// ------------------------------------------------------------------------------------
// model file (layout: convert.py:59-110,202-322 ; bark.cpp:664-727,995-1068)
// ------------------------------------------------------------------------------------
struct Tensor {
    int      ttype = 0;            // ggml_type: 0 f32, 1 f16, 2 q4_0, 3 q4_1, 6 q5_0, 7 q5_1, 8 q8_0
    int      n_dims = 0;
    int64_t  ne[4] = {1, 1, 1, 1}; // ne[0] innermost
    const uint8_t * data = nullptr;  // into the file mapping; NOT necessarily aligned
    int64_t  nelements() const { return ne[0] * ne[1] * ne[2] * ne[3]; }
};

struct Reader {
    const uint8_t * base; size_t size; size_t pos = 0; bool ok = true;
    template <typename T> T get() {
        T v{}; if (pos + sizeof(T) > size) { ok = false; return v; }
        memcpy(&v, base + pos, sizeof(T)); pos += sizeof(T); return v;
    }
    const uint8_t * skip(size_t n) {
        if (pos + n > size) { ok = false; return base; }
        const uint8_t * p = base + pos; pos += n; return p;
    }
};

// ggml block formats of quantised files (block_q4_0 ... block_q8_0, 32 weights per block):
//   type 2 q4_0: f16 d | 16 B nibbles             w_j = (n_j - 8) d            18 bytes
//   type 3 q4_1: f16 d | f16 m | 16 B nibbles     w_j = n_j d + m              20 bytes
//   type 6 q5_0: f16 d | u32 qh | 16 B nibbles    w_j = ((n_j | h_j << 4) - 16) d   22 bytes
//   type 7 q5_1: f16 d | f16 m | u32 qh | nibbles w_j = (n_j | h_j << 4) d + m      24 bytes
//   type 8 q8_0: f16 d | 32 int8                  w_j = q_j d                  34 bytes
// nibble byte i = element i (low) and i + 16 (high); bit j of qh = fifth bit of element j.
static int qblock_bytes(int ttype) {
    switch (ttype) { case 2: return 18; case 3: return 20; case 6: return 22; case 7: return 24; case 8: return 34; default: return 0; }
}
// the 32 integer levels of a block and its (d, m)
static void qblock_unpack(int ttype, const uint8_t * blk, int (&q)[32], float & d, float & m) {
    d = h2f(ld_u16(blk, 0)); m = 0.0f;
    const bool has_m = ttype == 3 || ttype == 7, has_h = ttype == 6 || ttype == 7;
    if (has_m) m = h2f(ld_u16(blk, 1));
    if (ttype == 8) { for (int j = 0; j < 32; j++) q[j] = (int8_t) blk[2 + j]; return; }
    uint32_t qh = 0;
    const uint8_t * p = blk + 2 + (has_m ? 2 : 0);
    if (has_h) { memcpy(&qh, p, 4); p += 4; }
    for (int j = 0; j < 16; j++) {
        int lo = p[j] & 0x0F, hi = p[j] >> 4;
        if (has_h) { lo |= (int) ((qh >> j) & 1u) << 4; hi |= (int) ((qh >> (j + 16)) & 1u) << 4; }
        if (ttype == 2) { lo -= 8; hi -= 8; }
        if (ttype == 6) { lo -= 16; hi -= 16; }
        q[j] = lo; q[j + 16] = hi;
    }
}

static bool read_tensor_record(Reader & r, std::string & name, Tensor & t) {
    t = Tensor();
    t.n_dims = r.get<int32_t>();
    int32_t len = r.get<int32_t>();
    t.ttype = r.get<int32_t>();
    if (!r.ok || t.n_dims < 0 || t.n_dims > 4 || len < 0 || len > 4096) return false;
    for (int i = 0; i < t.n_dims; i++) t.ne[i] = r.get<int32_t>();
    const uint8_t * nm = r.skip(len);
    if (!r.ok) return false;
    name.assign((const char *) nm, len);
    if (t.ttype != 0 && t.ttype != 1 && !qblock_bytes(t.ttype)) {
        fprintf(stderr, "oracle: tensor '%s' has type %d; only f32 / f16 / q4_0 / q4_1 / q5_0 / q5_1 / q8_0 files are restated\n", name.c_str(), t.ttype);
        return false;
    }
    size_t bytes = qblock_bytes(t.ttype) ? (size_t) t.nelements() / 32 * (size_t) qblock_bytes(t.ttype) : (size_t) t.nelements() * (t.ttype == 1 ? 2 : 4);
    t.data = r.skip(bytes);
    return r.ok;
}

static std::vector<float> to_f32(const Tensor & t) {
    std::vector<float> v((size_t) t.nelements());
    for (size_t i = 0; i < v.size(); i++) v[i] = t.ttype == 1 ? h2f(ld_u16(t.data, i)) : ld_f32(t.data, i);
    return v;
}

// ------------------------------------------------------------------------------------
// dense kernels in CANONICAL summation order (DESIGN.md "Canonical numerics").
//
// ggml's CPU dot products sum in a SIMD-lane order that depends on the build (AVX2 / AVX-512 /
// NEON use different accumulator counts), so the reference itself is not bit-stable across
// hosts.  This oracle and the HIP engine therefore agree on ONE fixed order per operator, chosen
// so that a GPU can reproduce it exactly with IEEE fp32 fma/add:
//
//  C1  weight dot  y = sum_k w[k]*x[k]   (mul_mat with f16/f32 weights, K padded with zeros to 128):
//        the K axis is cut into 8-element chunks; chunk q belongs to chain (q mod 16); a chain
//        walks its chunks in ascending order, and inside a chunk its 8 elements in ascending
//        order, with acc = fmaf(w, x, acc) starting from +0.  The 16 chain sums are combined by
//        the pairwise tree ((c0+c1)+(c2+c3))+((c4+c5)+(c6+c7)) ... (butterfly xor 1,2,4,8).
//  C2  attention score  s = sum_d k[d]*q[d]  (head_dim 64): FOUR chains, one per block of 16 consecutive d, each walking its d in
//        ascending order with acc = fmaf(k, q, acc) from +0; combined as (c0 + c1) + (c2 + c3).  (The 16-d block is what one
//        workgroup of the decode QKV kernel produces of q, so a chain can be formed where its q values are born.)
//  C5  attention mix    o[d] = sum_j v[j][d]*p[j] : key j belongs to chain (j mod 16), chains walk
//        j ascending with acc = fmaf(v, p, acc); combined by the same 16-leaf tree as C1.
// ------------------------------------------------------------------------------------
static inline float hsum_tree16(__m256 lo, __m256 hi) {
    alignas(32) float a[16];
    _mm256_store_ps(a, lo); _mm256_store_ps(a + 8, hi);
    float p[8], q[4], r[2];
    for (int i = 0; i < 8; i++) p[i] = a[2 * i] + a[2 * i + 1];
    for (int i = 0; i < 4; i++) q[i] = p[2 * i] + p[2 * i + 1];
    for (int i = 0; i < 2; i++) r[i] = q[2 * i] + q[2 * i + 1];
    return r[0] + r[1];
}

static inline int canon_kp(int K) { return (K + 127) & ~127; }

// Chain-major image of one K-vector: dst[(b*8 + e)*16 + c] = src[b*128 + c*8 + e], zero padded.
// (Padding is exact: fmaf(0, 0, acc) == acc.)
static void canon_image_f32(const float * src, int K, float * dst) {
    const int Kp = canon_kp(K);
    for (int b = 0; b < Kp / 128; b++) for (int e = 0; e < 8; e++) for (int c = 0; c < 16; c++) {
        const int k = b * 128 + c * 8 + e;
        dst[(b * 8 + e) * 16 + c] = k < K ? src[k] : 0.0f;
    }
}
// Weight matrices are re-imaged once at load time (CanonW); f16 weights stay f16 (exact).
struct CanonW {
    int M = 0, K = 0, Kp = 0; bool f16 = false;
    const uint8_t * q4 = nullptr;                       // quantised weights stay in file order: [M][K/32] blocks
    int qtype = 0;                                      // their ggml_type (2 q4_0, 3 q4_1, 6 q5_0, 7 q5_1, 8 q8_0)
    uint8_t * data = nullptr; size_t bytes = 0;         // [M][Kp] in chain-major order
    const uint8_t * raw = nullptr;                      // the file's row-major [M][K] image (dot_order != 0)
    // fine model (Numerics::fine_mfma): products in the f16 matrix cores' order (mfma_f16_emu.h).  Image for the 8-lane restatement, built
    // on first use: rows in blocks of 8, mw[blk][k][8] f16 bit patterns, me[blk][k][8] the operand exponents (-100 for zeros)
    bool mfma = false;
    mutable std::vector<uint16_t> mw; mutable std::vector<int8_t> me;
    CanonW() = default;
    CanonW(const CanonW &) = delete; CanonW & operator=(const CanonW &) = delete;
    CanonW(CanonW && o) noexcept : M(o.M), K(o.K), Kp(o.Kp), f16(o.f16), q4(o.q4), qtype(o.qtype), data(o.data), bytes(o.bytes), raw(o.raw), mfma(o.mfma) { o.data = nullptr; }
    ~CanonW() { if (data) munmap(data, bytes); }
    void build(const uint8_t * src, bool src_f16, int M_, int K_) {
        M = M_; K = K_; Kp = canon_kp(K); f16 = src_f16; raw = src;
        const size_t es = f16 ? 2 : 4;
        bytes = (size_t) M * Kp * es;
        data = (uint8_t *) big_alloc(bytes);
        #pragma omp parallel for schedule(static)
        for (int m = 0; m < M; m++) {
            const uint8_t * s = src + (size_t) m * K * es; uint8_t * d = data + (size_t) m * Kp * es;
            for (int b = 0; b < Kp / 128; b++) for (int e = 0; e < 8; e++) for (int c = 0; c < 16; c++) {
                const int k = b * 128 + c * 8 + e; const size_t o = (size_t) (b * 8 + e) * 16 + c;
                if (k < K) memcpy(d + o * es, s + (size_t) k * es, es); else memset(d + o * es, 0, es);
            }
        }
    }
};

template <bool A16> static inline void canon_load16(const uint8_t * p, __m256 & lo, __m256 & hi) {
    if (A16) { lo = _mm256_cvtph_ps(_mm_loadu_si128((const __m128i *) p)); hi = _mm256_cvtph_ps(_mm_loadu_si128((const __m128i *) (p + 16))); }
    else     { lo = _mm256_loadu_ps((const float *) p); hi = _mm256_loadu_ps((const float *) p + 8); }
}

template <bool A16, int NB>
static inline void canon_dot(const uint8_t * w, const float * x, size_t ldx, int Kp, float * out, size_t ldo) {
    __m256 a0[NB], a1[NB];
    for (int r = 0; r < NB; r++) { a0[r] = _mm256_setzero_ps(); a1[r] = _mm256_setzero_ps(); }
    const int steps = Kp / 16;
    for (int p = 0; p < steps; p++) {
        __m256 w0, w1; canon_load16<A16>(w + (size_t) p * 16 * (A16 ? 2 : 4), w0, w1);
        for (int r = 0; r < NB; r++) {
            a0[r] = _mm256_fmadd_ps(w0, _mm256_loadu_ps(x + r * ldx + p * 16), a0[r]);
            a1[r] = _mm256_fmadd_ps(w1, _mm256_loadu_ps(x + r * ldx + p * 16 + 8), a1[r]);
        }
    }
    for (int r = 0; r < NB; r++) out[r * ldo] = hsum_tree16(a0[r], a1[r]);
}

struct Layer {
    std::vector<float> ln1_g, ln1_b, ln2_g, ln2_b;         // *_b empty when absent
    CanonW attn_w, proj_w, fc_w, mproj_w;                   // file layout [out][in] row-major (ggml ne0 = in), re-imaged for C1
    std::vector<float> attn_b, proj_b, fc_b, mproj_b;       // empty when absent
};

struct Gpt {
    int n_layer = 0, n_head = 0, n_embd = 0, block_size = 0, bias = 0, n_in = 0, n_out = 0, n_lm_heads = 0, n_wtes = 0, ftype = 0;
    std::vector<float> lnf_g, lnf_b, wpe;
    std::vector<Tensor> wtes;
    std::vector<CanonW> lm_heads;
    std::vector<Layer> layers;
    // KV cache, f32, [layer][position][n_embd]  (bark.cpp:976-991,1293-1300)
    float * mem_k = nullptr, * mem_v = nullptr;
    // statistics, same quotients as bark_print_statistics (bark.cpp:176-182)
    int64_t t_sample_us = 0, t_predict_us = 0, t_main_us = 0, n_sample = 0;
};

static bool load_gpt(Reader & r, Gpt & m, bool need_kv) {
    // hparam order: bark.cpp:700-709
    m.n_layer = r.get<int32_t>(); m.n_head = r.get<int32_t>(); m.n_embd = r.get<int32_t>();
    m.block_size = r.get<int32_t>(); m.bias = r.get<int32_t>(); m.n_in = r.get<int32_t>();
    m.n_out = r.get<int32_t>(); m.n_lm_heads = r.get<int32_t>(); m.n_wtes = r.get<int32_t>();
    m.ftype = r.get<int32_t>();
    if (!r.ok) return false;
    {   // bark.cpp:711,727,2254: quantised files carry 2000 + ggml_ftype; restated: f32 (0), f16 (1), q4_0 (2), q4_1 (3), q8_0 (7), q5_0 (8), q5_1 (9)
        const int ft = m.ftype % 1000;
        if (!(ft == 0 || ft == 1 || ft == 2 || ft == 3 || ft == 7 || ft == 8 || ft == 9)) {
            fprintf(stderr, "oracle: ftype %d is not restated\n", m.ftype);
            return false;
        }
    }
    if (m.n_layer <= 0 || m.n_layer > 256 || m.n_embd <= 0 || m.n_head <= 0 || m.n_embd % m.n_head) return false;
    m.layers.resize(m.n_layer); m.wtes.resize(m.n_wtes); m.lm_heads.resize(m.n_lm_heads);
    int32_t n_tensors = r.get<int32_t>();
    std::map<std::string, Tensor> tens;
    for (int i = 0; i < n_tensors; i++) {
        std::string name; Tensor t;
        if (!read_tensor_record(r, name, t)) return false;
        tens[name] = t;
    }
    auto need = [&](const std::string & n, Tensor & out, int64_t ne0, int64_t ne1) {
        auto it = tens.find(n);
        if (it == tens.end()) { fprintf(stderr, "oracle: missing tensor %s\n", n.c_str()); return false; }
        if (it->second.ne[0] != ne0 || it->second.ne[1] != ne1) {   // bark.cpp:1034
            fprintf(stderr, "oracle: tensor %s has shape [%lld,%lld], expected [%lld,%lld]\n", n.c_str(),
                    (long long) it->second.ne[0], (long long) it->second.ne[1], (long long) ne0, (long long) ne1);
            return false;
        }
        out = it->second; return true;
    };
    auto vec = [&](const std::string & n, std::vector<float> & out, int64_t ne0, bool required) {
        auto it = tens.find(n);
        if (it == tens.end()) { if (required) fprintf(stderr, "oracle: missing tensor %s\n", n.c_str()); return !required; }
        if (it->second.ne[0] != ne0) return false;
        out = to_f32(it->second); return true;
    };
    auto needw = [&](const std::string & n, CanonW & out, int64_t ne0, int64_t ne1) {
        Tensor t; if (!need(n, t, ne0, ne1)) return false;
        if (qblock_bytes(t.ttype)) { out.M = (int) ne1; out.K = (int) ne0; out.q4 = t.data; out.qtype = t.ttype; return ne0 % 32 == 0; }
        out.build(t.data, t.ttype == 1, (int) ne1, (int) ne0); return true;
    };
    const int E = m.n_embd;
    bool ok = true;
    for (int i = 0; i < m.n_wtes; i++) ok = ok && need("model/wte/" + std::to_string(i), m.wtes[i], E, m.n_in);
    for (int i = 0; i < m.n_lm_heads; i++) ok = ok && needw("model/lm_head/" + std::to_string(i), m.lm_heads[i], E, m.n_out);
    { Tensor t; ok = ok && need("model/wpe", t, E, m.block_size); if (ok) m.wpe = to_f32(t); }
    ok = ok && vec("model/ln_f/g", m.lnf_g, E, true) && vec("model/ln_f/b", m.lnf_b, E, false);
    for (int l = 0; l < m.n_layer && ok; l++) {
        std::string p = "model/h" + std::to_string(l);
        Layer & L = m.layers[l];
        ok = ok && vec(p + "/ln_1/g", L.ln1_g, E, true) && vec(p + "/ln_1/b", L.ln1_b, E, false);
        ok = ok && vec(p + "/ln_2/g", L.ln2_g, E, true) && vec(p + "/ln_2/b", L.ln2_b, E, false);
        ok = ok && needw(p + "/attn/c_attn/w", L.attn_w, E, 3 * E) && needw(p + "/attn/c_proj/w", L.proj_w, E, E);
        ok = ok && needw(p + "/mlp/c_fc/w", L.fc_w, E, 4 * E) && needw(p + "/mlp/c_proj/w", L.mproj_w, 4 * E, E);
        ok = ok && vec(p + "/attn/c_attn/b", L.attn_b, 3 * E, false) && vec(p + "/attn/c_proj/b", L.proj_b, E, false);
        ok = ok && vec(p + "/mlp/c_fc/b", L.fc_b, 4 * E, false) && vec(p + "/mlp/c_proj/b", L.mproj_b, E, false);
    }
    if (!ok) return false;
    if (need_kv) {
        size_t n = (size_t) m.n_layer * m.block_size * E;
        m.mem_k = (float *) big_alloc(n * 4); m.mem_v = (float *) big_alloc(n * 4);
    }
    return true;
}

// EnCodec decoder weights, all widened to f32 at load (exact)
// wm: the f16 image of the kernel for the matrix-core order C9m, built on first use: conv [cout][k * cin + ci], transposed conv one matrix
// per output phase r, [r][cout][tap * cin + ci] (tap 0 = the previous frame, kernel element r + stride; tap 1 = this frame, element r)
struct Conv { std::vector<float> w, b; int cout = 0, cin = 0, k = 0; mutable std::vector<CanonW> wm; mutable std::vector<uint16_t> wm_bits; };          // w[cout][cin][k]
struct ConvT { std::vector<float> w, b; int cin = 0, cout = 0, k = 0, stride = 0; mutable std::vector<CanonW> wm; mutable std::vector<uint16_t> wm_bits; }; // w[cin][cout][k]
struct Lstm { CanonW w_ih, w_hh; std::vector<float> b_ih, b_hh; };
struct Codec {
    int in_channels = 0, hidden_dim = 0, n_filters = 0, kernel = 0, res_kernel = 0, n_bins = 0, bandwidth = 0, sr = 0, ftype = 0;
    std::vector<std::vector<float>> codebooks;   // [q][n_bins][hidden_dim]
    Conv init, fin;
    Lstm lstm[2];
    struct Block { ConvT up; Conv c1, c2, sc; } blocks[4];
};

static bool load_codec(Reader & r, Codec & c) {
    uint32_t magic = r.get<uint32_t>();
    if (!r.ok || magic != 0x67676d6c) { fprintf(stderr, "oracle: bad codec magic\n"); return false; }
    c.in_channels = r.get<int32_t>(); c.hidden_dim = r.get<int32_t>(); c.n_filters = r.get<int32_t>();
    c.kernel = r.get<int32_t>(); c.res_kernel = r.get<int32_t>(); c.n_bins = r.get<int32_t>();
    c.bandwidth = r.get<int32_t>(); c.sr = r.get<int32_t>(); c.ftype = r.get<int32_t>();
    std::map<std::string, Tensor> tens;
    while (r.ok && r.pos < r.size) {
        std::string name; Tensor t;
        if (!read_tensor_record(r, name, t)) return false;
        tens[name] = t;
    }
    auto get = [&](const std::string & n, std::vector<float> & out) {
        auto it = tens.find(n);
        if (it == tens.end()) { fprintf(stderr, "oracle: missing codec tensor %s\n", n.c_str()); return false; }
        out = to_f32(it->second); return true;
    };
    // f32 codec files (convert.py without --use-f16): the decoder of this restatement (and of the engine) runs in the f16-weight
    // arithmetic - conv kernels meet an f16 im2col in ggml's mul_mat, which converts the f32 operand to f16 anyway; for the LSTM
    // matrices the same rounding is a stated simplification (the un-vendored encodec.cpp cannot be consulted).
    auto f16_bits = [](const Tensor & t) {
        std::vector<uint16_t> h((size_t) t.nelements());
        for (size_t i = 0; i < h.size(); i++) h[i] = f2h(ld_f32(t.data, i));
        return h;
    };
    auto getw = [&](const std::string & n, CanonW & out) {
        auto it = tens.find(n);
        if (it == tens.end()) { fprintf(stderr, "oracle: missing codec tensor %s\n", n.c_str()); return false; }
        if (it->second.ttype == 0) {
            const std::vector<uint16_t> h = f16_bits(it->second);
            out.build((const uint8_t *) h.data(), true, (int) it->second.ne[1], (int) it->second.ne[0]); return true;
        }
        out.build(it->second.data, it->second.ttype == 1, (int) it->second.ne[1], (int) it->second.ne[0]); return true;
    };
    auto getcw = [&](const std::string & n, std::vector<float> & out) {          // conv / convtr kernels
        if (!get(n, out)) return false;
        if (tens.find(n)->second.ttype == 0) for (float & v : out) v = round_h(v);
        return true;
    };
    auto conv = [&](const std::string & p, Conv & cv) {
        auto it = tens.find(p + ".weight");
        if (it == tens.end()) { fprintf(stderr, "oracle: missing codec tensor %s.weight\n", p.c_str()); return false; }
        cv.k = (int) it->second.ne[0]; cv.cin = (int) it->second.ne[1]; cv.cout = (int) it->second.ne[2];
        return getcw(p + ".weight", cv.w) && get(p + ".bias", cv.b) && (int) cv.b.size() == cv.cout;
    };
    auto convt = [&](const std::string & p, ConvT & cv, int stride) {
        auto it = tens.find(p + ".weight");
        if (it == tens.end()) { fprintf(stderr, "oracle: missing codec tensor %s.weight\n", p.c_str()); return false; }
        cv.k = (int) it->second.ne[0]; cv.cout = (int) it->second.ne[1]; cv.cin = (int) it->second.ne[2]; cv.stride = stride;
        return getcw(p + ".weight", cv.w) && get(p + ".bias", cv.b) && (int) cv.b.size() == cv.cout;
    };
    bool ok = conv("decoder.model.0.conv.conv", c.init);
    for (int l = 0; l < 2 && ok; l++) {
        std::string s = std::to_string(l);
        ok = getw("decoder.model.1.lstm.weight_ih_l" + s, c.lstm[l].w_ih) && getw("decoder.model.1.lstm.weight_hh_l" + s, c.lstm[l].w_hh) &&
             get("decoder.model.1.lstm.bias_ih_l" + s, c.lstm[l].b_ih) && get("decoder.model.1.lstm.bias_hh_l" + s, c.lstm[l].b_hh);
    }
    const int ratios[4] = {8, 5, 4, 2};    // EnCodec 24 kHz upsampling ratios (modeling_encodec.py:329-340)
    for (int i = 0; i < 4 && ok; i++) {
        int idx = 3 + 3 * i;
        ok = convt("decoder.model." + std::to_string(idx) + ".convtr.convtr", c.blocks[i].up, ratios[i]) &&
             conv("decoder.model." + std::to_string(idx + 1) + ".block.1.conv.conv", c.blocks[i].c1) &&
             conv("decoder.model." + std::to_string(idx + 1) + ".block.3.conv.conv", c.blocks[i].c2) &&
             conv("decoder.model." + std::to_string(idx + 1) + ".shortcut.conv.conv", c.blocks[i].sc);
    }
    ok = ok && conv("decoder.model.15.conv.conv", c.fin);
    for (int q = 0; ok; q++) {
        auto it = tens.find("quantizer.vq.layers." + std::to_string(q) + "._codebook.embed");
        if (it == tens.end()) break;
        c.codebooks.push_back(to_f32(it->second));
    }
    return ok && !c.codebooks.empty();
}

// ------------------------------------------------------------------------------------
// numerics switches — defaults restate ggml's CPU backend
// ------------------------------------------------------------------------------------
struct Numerics {
    int act_round_f16 = 1;   // mul_mat with f16 weights converts the f32 activation to f16 first
    int gelu_mode = 0;       // 0: tanh GELU through a 64K-entry f16->f16 table; 1: tanh GELU in f32; 2: erf GELU (HF)
    // Summation order of every dot product (weight matmuls, attention scores, attention mix).  0 is the canonical order the engine
    // reproduces bit for bit; the others exist ONLY to measure how much the greedy token stream depends on the order
    // (tools/order_sensitivity.py, tests/test_order_sensitivity.py):
    //   1  ggml's AVX2 order (ggml_vec_dot_f16 / ggml_vec_dot_f32 with GGML_F16_STEP = GGML_F32_STEP = 32, four 8-lane accumulators:
    //      element k goes to chain k mod 32 and is added by fma in ascending k; reduction (a0 + a2) + (a1 + a3), then lanes
    //      (l + l+4), then (0+1) + (2+3); a tail of K mod 32 elements is added one by one in double) - what the reference's CPU
    //      path computes on an AVX2 host, restated from upstream ggml (not in /root/reference: SURVEY.md A.4)
    //   2  one sequential fmaf chain over ascending k
    int dot_order = 0;
    // The fine model's weight products (f16 model files, act_round_f16 on, dot_order 0).  0 (default since round 6): the C1 chains as for the other
    // models - the restatement of the reference's arithmetic, what the engine's bark_generate_audio computes.  1: the order of CDNA4's
    // v_mfma_f32_32x32x16_f16, restated in mfma_f16_emu.h (C1m) - what the engine's lock-step jobs run their fine passes in.
    int fine_mfma = 0;
    // The codec's convolutions (every one whose input channel count is a multiple of 8) in the same matrix-core order, over the axis
    // kd = k * cin + ci (transposed conv: tap * cin + ci per output phase) - order C9m; 0: one fmaf chain in (ci, k) order (C9).
    int codec_mfma = 1;
};

struct Oracle {
    const uint8_t * map = nullptr; size_t map_size = 0;
    std::map<std::string, int32_t> token_to_id;
    Gpt sem, coarse, fine;
    Codec codec;
    Numerics num;
    std::vector<uint16_t> gelu_table;
    std::mt19937 rng;
    // scratch
    BigBuf x, xn, qkv, att, fc, tmp, scores, vt, logits, ximg;
    ~Oracle() {
        for (Gpt * g : {&sem, &coarse, &fine}) {
            size_t n = (size_t) g->n_layer * g->block_size * g->n_embd * 4;
            if (g->mem_k) munmap(g->mem_k, n);
            if (g->mem_v) munmap(g->mem_v, n);
        }
        if (map) munmap((void *) map, map_size);
    }
};

// ggml's GELU: 0.5x(1+tanh(sqrt(2/pi) x (1+0.044715x^2))), tabulated over all f16 inputs.
// Built with contraction disabled (see Makefile) so that any compiler yields the same table.
static float gelu_tanh_f32(float x) {
    const float a = 0.044715f, c = 0.79788456080286535587989211986876f;
    float x2 = x * x;
    float inner = 1.0f + a * x2;
    float arg = c * x * inner;
    float t = tanhf(arg);
    return 0.5f * x * (1.0f + t);
}
static void build_gelu_table(std::vector<uint16_t> & tab) {
    tab.resize(65536);
    for (uint32_t i = 0; i < 65536; i++) tab[i] = f2h(gelu_tanh_f32(h2f((uint16_t) i)));
}
static inline float gelu_apply(const Oracle & o, float x) {
    switch (o.num.gelu_mode) {
        case 0:
            if (x <= -10.0f) return 0.0f;
            if (x >= 10.0f) return x;
            return h2f(o.gelu_table[f2h(x)]);
        case 1: return gelu_tanh_f32(x);
        default: return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
    }
}

// ------------------------------------------------------------------------------------
// elementwise pieces with ggml-CPU rounding points
// ------------------------------------------------------------------------------------
// ggml_norm + ggml_mul(g) [+ ggml_add(b)]  (bark.cpp:1265-1274): sums in double, eps on the variance
static void layer_norm_row(const float * x, float * y, int E, const float * g, const float * b) {
    double sum = 0.0;
    for (int i = 0; i < E; i++) sum += (double) x[i];
    float mean = (float) (sum / E);
    double sum2 = 0.0;
    for (int i = 0; i < E; i++) { float v = x[i] - mean; y[i] = v; sum2 += (double) (v * v); }
    float variance = (float) (sum2 / E);
    const float scale = 1.0f / sqrtf(variance + 1e-5f);   // EPS_NORM, bark.cpp:30
    for (int i = 0; i < E; i++) {
        float v = y[i] * scale;
        v = v * g[i];
        if (b) v = v + b[i];
        y[i] = v;
    }
}
static void round_rows(const Oracle & o, float * x, size_t n, int nth = 1) {
    if (!o.num.act_round_f16) return;
    #pragma omp parallel for schedule(static) num_threads(nth) if (nth > 1 && n >= 65536)
    for (size_t i = 0; i < n; i++) x[i] = round_h(x[i]);
}
// C4e: the exponential of ggml's f32 soft_max (bark.cpp:1322,1513 -> ggml_soft_max_inplace -> ggml_vec_soft_max_f32).  ggml (third-party, its
// submodule is absent from /root/reference) evaluates it with ggml_v_expf on the SIMD body of a row: the vector expf of ARM's optimised
// routines - n = round(x log2 e) by adding 1.5 x 2^23, b = x - n ln2 (hi / lo split), 2^n built from the exponent bits, a degree-5 polynomial
// in b, all fused multiply-adds.  Restated here operation for operation with that published routine's constants, as ONE scalar definition for
// every element (ggml's scalar expf tail depends on the host's vector width).  x = s - max <= 0; below n = -125 the
// value is defined as +0.  Every operation is a correctly rounded IEEE single operation: the same bits on any CPU and on CDNA4.
// (Rounds 1 - 5 defined C4 as (float) exp((double) x): neither ggml's arithmetic nor cheap on a GPU.)
static inline float canon_expf(float x) {
    const float r = 0x1.8p23f;
    const float z = fmaf(x, 0x1.715476p+0f, r);
    const float n = z - r;
    float b = fmaf(-n, 0x1.62e4p-1f, x);
    b = fmaf(-n, 0x1.7f7d1cp-20f, b);
    uint32_t zb; memcpy(&zb, &z, 4);
    const uint32_t kb = (zb << 23) + 0x3f800000u;
    float k; memcpy(&k, &kb, 4);                                  // 2^n
    const float u = b * b;
    const float j = fmaf(fmaf(fmaf(0x1.0e4020p-7f, b, 0x1.573e2ep-5f), u, fmaf(0x1.555e66p-3f, b, 0x1.fffdb6p-2f)), u, 0x1.ffffecp-1f * b);
    return n < -125.0f ? 0.0f : fmaf(k, j, k);
}
extern "C" float bark_oracle_canon_expf(float x) { return canon_expf(x); }       // tests: the routine against exact exponentials

// ggml_soft_max over one row of `n` valid entries: max, exp (C4e), double sum, scale by (float)(1/sum).
static void softmax_row(float * s, int n) {
    float mx = -INFINITY;
    for (int i = 0; i < n; i++) mx = std::max(mx, s[i]);
    double sum = 0.0;
    for (int i = 0; i < n; i++) { float e = canon_expf(s[i] - mx); s[i] = e; sum += (double) e; }
    const float inv = (float) (1.0 / sum);
    for (int i = 0; i < n; i++) s[i] *= inv;
}

// C[n*ldc + m] = C1-dot(W[m], B[n])   (B rows: f32, already holding f16-rounded values where ggml rounds)
// ggml's quantised products (ggml_vec_dot_q4_0_q8_0, _q4_1_q8_1, _q5_0_q8_0, _q5_1_q8_1, _q8_0_q8_0): mul_mat first quantises
// the f32 activation row to q8_0 / q8_1 blocks: d = amax / 127, q = roundf(x / d), d stored as f16,
// and for q8_1 also s = f16(d * sum q) with the unrounded d.  Per 32-element block, sumi = sum w_int * q8 (exact int32) and
//   q4_0:        t = ((float) sumi * dw) * dx
//   q5_0, q8_0:  t = (dw * dx) * (float) sumi
//   q4_1, q5_1:  t = (dw * dx) * (float) sumi + mw * sx
// Canonical order C1q: block b belongs to chain (b mod 16), chains add their t in ascending b from +0, then the C1 tree.
struct Q8Row { std::vector<int8_t> q; std::vector<float> d, s; };
static void quantize_row_q8(const float * x, int K, Q8Row & r) {
    r.q.resize((size_t) K); r.d.resize((size_t) K / 32); r.s.resize((size_t) K / 32);
    for (int b = 0; b < K / 32; b++) {
        float amax = 0.0f;
        for (int j = 0; j < 32; j++) amax = std::max(amax, fabsf(x[b * 32 + j]));
        const float d = amax / 127.0f;                       // amax / ((1 << 7) - 1)
        const float id = d ? 1.0f / d : 0.0f;
        r.d[(size_t) b] = round_h(d);                        // the block scale is stored as f16
        int sum = 0;
        for (int j = 0; j < 32; j++) { const int8_t q = (int8_t) roundf(x[b * 32 + j] * id); r.q[(size_t) b * 32 + j] = q; sum += q; }
        r.s[(size_t) b] = round_h((float) sum * d);          // q8_1: y.s = FP16(sum * d)
    }
}
static float dot_q_q8(int qtype, const uint8_t * wrow, const Q8Row & x, int K) {
    float acc[16];
    for (float & a : acc) a = 0.0f;
    const int bb = qblock_bytes(qtype);
    for (int b = 0; b < K / 32; b++) {
        int w[32]; float dw, mw;
        qblock_unpack(qtype, wrow + (size_t) b * bb, w, dw, mw);
        const int8_t * q8 = x.q.data() + (size_t) b * 32;
        int sumi = 0;
        for (int j = 0; j < 32; j++) sumi += w[j] * q8[j];
        const float dx = x.d[(size_t) b];
        float t;
        if (qtype == 2) t = ((float) sumi * dw) * dx;
        else {
            const float dd = dw * dx;
            t = dd * (float) sumi;
            if (qtype == 3 || qtype == 7) { const float ms = mw * x.s[(size_t) b]; t = t + ms; }
        }
        acc[b & 15] = acc[b & 15] + t;
    }
    for (int st = 1; st < 16; st <<= 1) for (int c = 0; c < 16; c += 2 * st) acc[c] = acc[c] + acc[c + st];
    return acc[0];
}
// the same value as dot_q_q8 for a weight row that has been unpacked once (levels as int8, block scales as floats): the block sums are
// exact integers and every float operation is the one dot_q_q8 performs, in the same order
static float dot_unpacked_q8(int qtype, const int8_t * wq, const float * dws, const float * mws, const Q8Row & x, int K) {
    float acc[16];
    for (float & a : acc) a = 0.0f;
    for (int b = 0; b < K / 32; b++) {
        const int8_t * w = wq + (size_t) b * 32, * q8 = x.q.data() + (size_t) b * 32;
        int sumi = 0;
        for (int j = 0; j < 32; j++) sumi += (int) w[j] * (int) q8[j];
        const float dw = dws[b], dx = x.d[(size_t) b];
        float t;
        if (qtype == 2) t = ((float) sumi * dw) * dx;
        else {
            const float dd = dw * dx;
            t = dd * (float) sumi;
            if (qtype == 3 || qtype == 7) { const float ms = mws[b] * x.s[(size_t) b]; t = t + ms; }
        }
        acc[b & 15] = acc[b & 15] + t;
    }
    for (int st = 1; st < 16; st <<= 1) for (int c = 0; c < 16; c += 2 * st) acc[c] = acc[c] + acc[c + st];
    return acc[0];
}
static void gemm_q4(const CanonW & W, const float * B, size_t ldb, float * C, size_t ldc, int M, int N, int K, int nth) {
    std::vector<Q8Row> rows((size_t) N);
    #pragma omp parallel for schedule(static) num_threads(nth) if (nth > 1 && N >= 16)
    for (int n = 0; n < N; n++) quantize_row_q8(B + (size_t) n * ldb, K, rows[(size_t) n]);
    const int bb = qblock_bytes(W.qtype);
    const size_t rb = (size_t) K / 32 * (size_t) bb;
    #pragma omp parallel num_threads(nth) if (nth > 1 && (int64_t) M * N * K > 65536)
    {
        std::vector<int8_t> wq((size_t) K);
        std::vector<float> dws((size_t) K / 32), mws((size_t) K / 32);
        #pragma omp for schedule(static)
        for (int m = 0; m < M; m++) {
            // levels of every format fit int8: q4_0 -8..7, q4_1 0..15, q5_0 -16..15, q5_1 0..31, q8_0 -128..127
            for (int b = 0; b < K / 32; b++) {
                int w[32]; float dw, mw;
                qblock_unpack(W.qtype, W.q4 + (size_t) m * rb + (size_t) b * bb, w, dw, mw);
                for (int j = 0; j < 32; j++) wq[(size_t) b * 32 + j] = (int8_t) w[j];
                dws[(size_t) b] = dw; mws[(size_t) b] = mw;
            }
            for (int n = 0; n < N; n++) C[(size_t) n * ldc + m] = dot_unpacked_q8(W.qtype, wq.data(), dws.data(), mws.data(), rows[(size_t) n], K);
        }
    }
}


// ---- study orders (Numerics::dot_order != 0): the same products, other summation orders ------------------------------------------
// ggml AVX2: 32 chains (4 accumulators x 8 lanes), step 32; w: K floats or f16 bit patterns (row-major), x: K floats
static inline float hsum_ggml_avx2(__m256 a0, __m256 a1, __m256 a2, __m256 a3) {
    a0 = _mm256_add_ps(a0, a2); a1 = _mm256_add_ps(a1, a3);          // GGML_F32x8_REDUCE: offset 2, then offset 1
    a0 = _mm256_add_ps(a0, a1);
    const __m128 t0 = _mm_add_ps(_mm256_castps256_ps128(a0), _mm256_extractf128_ps(a0, 1));
    const __m128 t1 = _mm_hadd_ps(t0, t0);
    return _mm_cvtss_f32(_mm_hadd_ps(t1, t1));
}
template <bool W16> static inline float dot_ggml_avx2(const uint8_t * w, const float * x, int K) {
    __m256 a[4] = {_mm256_setzero_ps(), _mm256_setzero_ps(), _mm256_setzero_ps(), _mm256_setzero_ps()};
    const int np = K & ~31;
    for (int i = 0; i < np; i += 32)
        for (int j = 0; j < 4; j++) {
            const __m256 wv = W16 ? _mm256_cvtph_ps(_mm_loadu_si128((const __m128i *) (w + (size_t) (i + 8 * j) * 2))) : _mm256_loadu_ps((const float *) w + i + 8 * j);
            a[j] = _mm256_fmadd_ps(wv, _mm256_loadu_ps(x + i + 8 * j), a[j]);
        }
    double sumf = (double) hsum_ggml_avx2(a[0], a[1], a[2], a[3]);
    for (int i = np; i < K; i++) {                                    // leftovers: float product added in ggml_float (double)
        const float wf = W16 ? h2f(((const uint16_t *) w)[i]) : ((const float *) w)[i];
        sumf += (double) (wf * x[i]);
    }
    return (float) sumf;
}
template <bool W16> static inline float dot_sequential(const uint8_t * w, const float * x, int K) {
    float acc = 0.0f;
    for (int i = 0; i < K; i++) acc = fmaf(W16 ? h2f(((const uint16_t *) w)[i]) : ((const float *) w)[i], x[i], acc);
    return acc;
}
static void gemm_w_alt(const Oracle & o, const CanonW & W, const float * B, size_t ldb, float * C, size_t ldc, int M, int N, int K, int nth) {
    const size_t rb = (size_t) K * (W.f16 ? 2 : 4);
    #pragma omp parallel for schedule(static) num_threads(nth) if (nth > 1 && (int64_t) M * N * K > 65536)
    for (int m = 0; m < M; m++) {
        const uint8_t * w = W.raw + (size_t) m * rb;
        for (int n = 0; n < N; n++) {
            const float * x = B + (size_t) n * ldb;
            float y;
            if (o.num.dot_order == 1) y = W.f16 ? dot_ggml_avx2<true>(w, x, K) : dot_ggml_avx2<false>(w, x, K);
            else                      y = W.f16 ? dot_sequential<true>(w, x, K) : dot_sequential<false>(w, x, K);
            C[(size_t) n * ldc + m] = y;
        }
    }
}

// ---- C1m: the f16 matrix cores' order (mfma_f16_emu.h states the arithmetic; this is the same thing for 8 output rows at a time) ---------
// One group of 8 products per lane: p_k = w_k x_k (exact in f32), E = max_k (e_w + e_x), every p_k scaled by 2^(24 - E) and truncated toward
// zero by the float -> int conversion, summed as integers; the accumulator is joined in double precision (every intermediate is an integer
// below 2^35 times a power of two: exact), floor to the 32-bit window, floor to 32 leading bits, round to nearest even by the double -> float
// conversion.  tests/test_mfma_f16_emu.py holds it against the scalar statement and against device dumps.
static inline __m256d pow2_pd(__m128i e) {                // 2^e for four int32 exponents (|e| < 1000)
    return _mm256_castsi256_pd(_mm256_slli_epi64(_mm256_add_epi64(_mm256_cvtepi32_epi64(e), _mm256_set1_epi64x(1023)), 52));
}
static inline __m128 mfma_join4(__m128 acc, __m128i S, __m128i E) {
    const __m128i ab = _mm_castps_si128(acc);
    const __m128i eacc = _mm_sub_epi32(_mm_and_si128(_mm_srli_epi32(ab, 23), _mm_set1_epi32(255)), _mm_set1_epi32(127));
    const __m128i lsbp = _mm_sub_epi32(E, _mm_set1_epi32(24));
    const __m128i lsb = _mm_max_epi32(_mm_sub_epi32(eacc, _mm_set1_epi32(32)), lsbp);
    const __m256d f1 = pow2_pd(_mm_sub_epi32(_mm_setzero_si128(), lsb));          // 2^-lsb
    const __m256d f2 = pow2_pd(_mm_sub_epi32(lsbp, lsb));                          // 2^(lsb_p - lsb) <= 1
    const __m256d a1 = _mm256_floor_pd(_mm256_mul_pd(_mm256_cvtps_pd(acc), f1));
    const __m256d s1 = _mm256_floor_pd(_mm256_mul_pd(_mm256_cvtepi32_pd(S), f2));
    const __m256d T = _mm256_add_pd(a1, s1);
    // 32 leading bits, floor: |T| is an integer (>= 1 or 0), its biased exponent field clamped to 1023 so that T == 0 scales harmlessly
    __m256i eb = _mm256_and_si256(_mm256_srli_epi64(_mm256_castpd_si256(T), 52), _mm256_set1_epi64x(0x7ff));
    eb = _mm256_max_epi32(eb, _mm256_set1_epi64x(1023));
    const __m256d g    = _mm256_castsi256_pd(_mm256_slli_epi64(_mm256_sub_epi64(_mm256_set1_epi64x(2077), eb), 52));     // 2^(31 - eT)
    const __m256d ginv = _mm256_castsi256_pd(_mm256_slli_epi64(_mm256_sub_epi64(eb, _mm256_set1_epi64x(31)), 52));       // 2^(eT - 31)
    const __m256d T2 = _mm256_mul_pd(_mm256_floor_pd(_mm256_mul_pd(T, g)), ginv);
    return _mm256_cvtpd_ps(_mm256_mul_pd(T2, pow2_pd(lsb)));
}
static void mfma_build_image(const CanonW & W) {
    const int M = W.M, K = W.K, nb = (M + 7) / 8;
    W.mw.assign((size_t) nb * K * 8, 0); W.me.assign((size_t) nb * K * 8, (int8_t) -100);
    const uint16_t * src = (const uint16_t *) W.raw;
    for (int m = 0; m < M; m++) for (int k = 0; k < K; k++) {
        const uint16_t h = src[(size_t) m * K + k];
        const size_t o = ((size_t) (m >> 3) * K + k) * 8 + (m & 7);
        W.mw[o] = h;
        const mfma_emu::H16 d = mfma_emu::decode_h16(h);
        W.me[o] = d.m ? (int8_t) d.e : (int8_t) -100;
    }
}
// ---- the same for 16 output rows at a time where the host has AVX-512 (F + DQ; runtime dispatch, the build stays x86-64-v3) -------------
#define ORC_AVX512 __attribute__((target("avx512f,avx512dq,avx512bw,avx512vl")))
ORC_AVX512 static inline __m512d pow2_pd8(__m256i e) {
    return _mm512_castsi512_pd(_mm512_slli_epi64(_mm512_add_epi64(_mm512_cvtepi32_epi64(e), _mm512_set1_epi64(1023)), 52));
}
ORC_AVX512 static inline __m256 mfma_join8(__m256 acc, __m256i S, __m256i E) {
    const __m256i ab = _mm256_castps_si256(acc);
    const __m256i eacc = _mm256_sub_epi32(_mm256_and_si256(_mm256_srli_epi32(ab, 23), _mm256_set1_epi32(255)), _mm256_set1_epi32(127));
    const __m256i lsbp = _mm256_sub_epi32(E, _mm256_set1_epi32(24));
    const __m256i lsb = _mm256_max_epi32(_mm256_sub_epi32(eacc, _mm256_set1_epi32(32)), lsbp);
    const __m512d f1 = pow2_pd8(_mm256_sub_epi32(_mm256_setzero_si256(), lsb));
    const __m512d f2 = pow2_pd8(_mm256_sub_epi32(lsbp, lsb));
    const __m512d a1 = _mm512_roundscale_pd(_mm512_mul_pd(_mm512_cvtps_pd(acc), f1), _MM_FROUND_TO_NEG_INF | _MM_FROUND_NO_EXC);
    const __m512d s1 = _mm512_roundscale_pd(_mm512_mul_pd(_mm512_cvtepi32_pd(S), f2), _MM_FROUND_TO_NEG_INF | _MM_FROUND_NO_EXC);
    const __m512d T = _mm512_add_pd(a1, s1);
    __m512i eb = _mm512_and_si512(_mm512_srli_epi64(_mm512_castpd_si512(T), 52), _mm512_set1_epi64(0x7ff));
    eb = _mm512_max_epi64(eb, _mm512_set1_epi64(1023));
    const __m512d g    = _mm512_castsi512_pd(_mm512_slli_epi64(_mm512_sub_epi64(_mm512_set1_epi64(2077), eb), 52));
    const __m512d ginv = _mm512_castsi512_pd(_mm512_slli_epi64(_mm512_sub_epi64(eb, _mm512_set1_epi64(31)), 52));
    const __m512d T2 = _mm512_mul_pd(_mm512_roundscale_pd(_mm512_mul_pd(T, g), _MM_FROUND_TO_NEG_INF | _MM_FROUND_NO_EXC), ginv);
    return _mm512_cvtpd_ps(_mm512_mul_pd(T2, pow2_pd8(lsb)));
}
// rows [n0, n1) of B against the row blocks [b0, b1) of W (b0 even, pairs of 8-row blocks = 16 lanes; an odd last block runs with its upper half on zeros)
ORC_AVX512 static void gemm_mfma_rows_avx512(const CanonW & W, const float * B, size_t ldb, float * C, size_t ldc, int M, int K, int n0, int n1, int nb, int32_t * xe) {
    for (int n = n0; n < n1; n++) {
        const float * x = B + (size_t) n * ldb;
        for (int k = 0; k < K; k++) {
            const mfma_emu::H16 d = mfma_emu::decode_h16(f2h(x[k]));
            xe[k] = d.m ? d.e : -100;
        }
        for (int b = 0; b < nb; b += 2) {
            const bool two = b + 1 < nb;
            const uint16_t * mw0 = W.mw.data() + (size_t) b * K * 8, * mw1 = two ? mw0 + (size_t) K * 8 : mw0;
            const int8_t * me0 = W.me.data() + (size_t) b * K * 8, * me1 = two ? me0 + (size_t) K * 8 : me0;
            __m512 acc = _mm512_setzero_ps();
            for (int g = 0; g < K; g += 8) {
                __m512 p[8];
                __m512i E = _mm512_set1_epi32(-60);
                for (int k = 0; k < 8; k++) {
                    const __m256i w16 = _mm256_inserti128_si256(_mm256_castsi128_si256(_mm_loadu_si128((const __m128i *) (mw0 + (size_t) (g + k) * 8))),
                                                                _mm_loadu_si128((const __m128i *) (mw1 + (size_t) (g + k) * 8)), 1);
                    p[k] = _mm512_mul_ps(_mm512_cvtph_ps(w16), _mm512_set1_ps(x[g + k]));
                    const __m128i e8 = _mm_unpacklo_epi64(_mm_loadl_epi64((const __m128i *) (me0 + (size_t) (g + k) * 8)), _mm_loadl_epi64((const __m128i *) (me1 + (size_t) (g + k) * 8)));
                    E = _mm512_max_epi32(E, _mm512_add_epi32(_mm512_cvtepi8_epi32(e8), _mm512_set1_epi32(xe[g + k])));
                }
                const __m512 invq = _mm512_castsi512_ps(_mm512_slli_epi32(_mm512_sub_epi32(_mm512_set1_epi32(127 + 24), E), 23));
                __m512i S = _mm512_setzero_si512();
                for (int k = 0; k < 8; k++) S = _mm512_add_epi32(S, _mm512_cvttps_epi32(_mm512_mul_ps(p[k], invq)));
                const __m256 lo = mfma_join8(_mm512_castps512_ps256(acc), _mm512_castsi512_si256(S), _mm512_castsi512_si256(E));
                const __m256 hi = mfma_join8(_mm512_extractf32x8_ps(acc, 1), _mm512_extracti32x8_epi32(S, 1), _mm512_extracti32x8_epi32(E, 1));
                acc = _mm512_insertf32x8(_mm512_castps256_ps512(lo), hi, 1);
            }
            alignas(64) float out[16];
            _mm512_store_ps(out, acc);
            const int cnt = std::min(two ? 16 : 8, M - 8 * b);
            for (int i = 0; i < cnt; i++) C[(size_t) n * ldc + 8 * b + i] = out[i];
        }
    }
}
static bool host_has_avx512() {
    static const bool ok = __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512dq") && __builtin_cpu_supports("avx512bw") &&
                           __builtin_cpu_supports("avx512vl") && !getenv("ORACLE_NO_AVX512");
    return ok;
}

// C[n*ldc + m] = C1m-dot(W[m], B[n]); B rows hold f16-representable values
static void gemm_mfma(const CanonW & W, const float * B, size_t ldb, float * C, size_t ldc, int M, int N, int K, int nth) {
    assert((K & 7) == 0 && W.raw);
    if (W.mw.empty()) mfma_build_image(W);
    const int nb = (M + 7) / 8;
    if (host_has_avx512()) {
        #pragma omp parallel num_threads(nth) if (nth > 1 && N >= 4)
        {
            std::vector<int32_t> xe((size_t) K);
            #pragma omp for schedule(dynamic, 4)
            for (int n = 0; n < N; n++) gemm_mfma_rows_avx512(W, B, ldb, C, ldc, M, K, n, n + 1, nb, xe.data());
        }
        return;
    }
    #pragma omp parallel num_threads(nth) if (nth > 1 && N >= 4)
    {
        std::vector<int32_t> xe((size_t) K);
        #pragma omp for schedule(dynamic, 4)
        for (int n = 0; n < N; n++) {
            const float * x = B + (size_t) n * ldb;
            for (int k = 0; k < K; k++) {
                const uint16_t h = f2h(x[k]);
                assert(h2f(h) == x[k] || x[k] != x[k]);
                const mfma_emu::H16 d = mfma_emu::decode_h16(h);
                xe[(size_t) k] = d.m ? d.e : -100;
            }
            for (int b = 0; b < nb; b++) {
                const uint16_t * mw = W.mw.data() + (size_t) b * K * 8;
                const int8_t * me = W.me.data() + (size_t) b * K * 8;
                __m256 acc = _mm256_setzero_ps();
                for (int g = 0; g < K; g += 8) {
                    __m256 p[8];
                    __m256i E = _mm256_set1_epi32(-60);                        // all-zero groups: S = 0 and the accumulator passes through
                    for (int k = 0; k < 8; k++) {
                        const __m256 w = _mm256_cvtph_ps(_mm_loadu_si128((const __m128i *) (mw + (size_t) (g + k) * 8)));
                        p[k] = _mm256_mul_ps(w, _mm256_broadcast_ss(x + g + k));
                        const __m256i e = _mm256_add_epi32(_mm256_cvtepi8_epi32(_mm_loadl_epi64((const __m128i *) (me + (size_t) (g + k) * 8))), _mm256_set1_epi32(xe[(size_t) (g + k)]));
                        E = _mm256_max_epi32(E, e);
                    }
                    const __m256 invq = _mm256_castsi256_ps(_mm256_slli_epi32(_mm256_sub_epi32(_mm256_set1_epi32(127 + 24), E), 23));      // 2^(24 - E)
                    __m256i S = _mm256_setzero_si256();
                    for (int k = 0; k < 8; k++) S = _mm256_add_epi32(S, _mm256_cvttps_epi32(_mm256_mul_ps(p[k], invq)));
                    const __m128 lo = mfma_join4(_mm256_castps256_ps128(acc), _mm256_castsi256_si128(S), _mm256_castsi256_si128(E));
                    const __m128 hi = mfma_join4(_mm256_extractf128_ps(acc, 1), _mm256_extracti128_si256(S, 1), _mm256_extracti128_si256(E, 1));
                    acc = _mm256_set_m128(hi, lo);
                }
                alignas(32) float out[8];
                _mm256_store_ps(out, acc);
                const int cnt = std::min(8, M - 8 * b);
                for (int i = 0; i < cnt; i++) C[(size_t) n * ldc + 8 * b + i] = out[i];
            }
        }
    }
}

static void gemm_w(Oracle & o, const CanonW & W, const float * B, size_t ldb, float * C, size_t ldc, int M, int N, int K, int nth) {
    assert(M == W.M && K == W.K);
    if (W.q4) { gemm_q4(W, B, ldb, C, ldc, M, N, K, nth); return; }
    if (W.mfma && W.f16 && W.raw && o.num.fine_mfma && o.num.act_round_f16 && o.num.dot_order == 0 && (K & 7) == 0) { gemm_mfma(W, B, ldb, C, ldc, M, N, K, nth); return; }
    if (o.num.dot_order != 0 && W.raw) { gemm_w_alt(o, W, B, ldb, C, ldc, M, N, K, nth); return; }
    const int Kp = W.Kp;
    o.ximg.ensure((size_t) N * Kp);
    float * xi = o.ximg.p;
    #pragma omp parallel for schedule(static) num_threads(nth) if (nth > 1 && N >= 16)
    for (int n = 0; n < N; n++) canon_image_f32(B + (size_t) n * ldb, K, xi + (size_t) n * Kp);
    const size_t rb = (size_t) Kp * (W.f16 ? 2 : 4);
    if (N >= 16) {
        // many rows: four x images (4 Kp floats, L1-sized) stay put while the weight rows stream past them; every output is still
        // its own canon_dot chain set, so the loop order does not touch a single bit
        constexpr int RB = 4;                                    // (6 rows per block measured the same)
        const int nb = (N + RB - 1) / RB;
        #pragma omp parallel for schedule(static) num_threads(nth) if (nth > 1)
        for (int b = 0; b < nb; b++) {
            const int n = RB * b, cnt = std::min(RB, N - n);
            const float * xb = xi + (size_t) n * Kp;
            float * cb = C + (size_t) n * ldc;
            for (int m = 0; m < M; m++) {
                const uint8_t * w = W.data + (size_t) m * rb;
                if (cnt == RB) { if (W.f16) canon_dot<true, RB>(w, xb, Kp, Kp, cb + m, ldc); else canon_dot<false, RB>(w, xb, Kp, Kp, cb + m, ldc); }
                else for (int r = 0; r < cnt; r++) { if (W.f16) canon_dot<true, 1>(w, xb + (size_t) r * Kp, Kp, Kp, cb + (size_t) r * ldc + m, ldc); else canon_dot<false, 1>(w, xb + (size_t) r * Kp, Kp, Kp, cb + (size_t) r * ldc + m, ldc); }
            }
        }
        return;
    }
    #pragma omp parallel for schedule(static) num_threads(nth) if (nth > 1 && (int64_t) M * N * K > 65536)
    for (int m = 0; m < M; m++) {
        const uint8_t * w = W.data + (size_t) m * rb;
        int n = 0;
        if (W.f16) {
            for (; n + 4 <= N; n += 4) canon_dot<true, 4>(w, xi + (size_t) n * Kp, Kp, Kp, C + (size_t) n * ldc + m, ldc);
            for (; n < N; n++)         canon_dot<true, 1>(w, xi + (size_t) n * Kp, Kp, Kp, C + (size_t) n * ldc + m, ldc);
        } else {
            for (; n + 4 <= N; n += 4) canon_dot<false, 4>(w, xi + (size_t) n * Kp, Kp, Kp, C + (size_t) n * ldc + m, ldc);
            for (; n < N; n++)         canon_dot<false, 1>(w, xi + (size_t) n * Kp, Kp, Kp, C + (size_t) n * ldc + m, ldc);
        }
    }
}

// Multi-head attention for N query rows against `ctx_total` cached rows.
//   q: rows at q[i*ldq + h*64 ...]; K/V rows at kc[j*E + h*64 ...]; out[i*E + h*64 ...]
//   causal: query i sees keys j <= n_past + i (ggml_diag_mask_inf(n_past), bark.cpp:1320)
// Summation orders: scores C2, mix C5 (see the dense-kernel header).  Masked keys carry p == 0 in
// ggml and fmaf(v, 0, acc) == acc, so they are simply skipped here.
static void attention(Oracle & o, const float * q, size_t ldq, const float * kc, const float * vc, float * out,
                      int N, int ctx_total, int n_past, bool causal, int E, int H, int nth) {
    const int D = E / H;
    const float scale = 1.0f / sqrtf((float) E / H);      // bark.cpp:1318
    const int ctx8 = (ctx_total + 7) & ~7;
    o.scores.ensure((size_t) nth * ctx8);
    o.vt.ensure((size_t) nth * D * ctx8);
    assert(D == 64);                                       // C2 is stated for head_dim 64 (every Bark model)
    // tasks = heads x blocks of query rows (a block re-transposes its head's K: 64 ctx copies against >= 64 ctx fmaf per row), so that
    // 12 heads do not leave 8 threads waiting for a second round; every output row is computed exactly as before
    const int RB = (N >= 256 && o.num.dot_order == 0) ? 4 : 1;
    #pragma omp parallel for schedule(dynamic, 1) num_threads(nth) if (nth > 1)
    for (int task = 0; task < H * RB; task++) {
        const int h = task / RB, rb = task % RB;
        const int i_lo = (int) ((int64_t) N * rb / RB), i_hi = (int) ((int64_t) N * (rb + 1) / RB);
        const int tid = omp_get_thread_num();
        float * row = o.scores.p + (size_t) tid * ctx8;
        float * Kt = o.vt.p + (size_t) tid * D * ctx8;                    // Kt[d][j]
        for (int j = 0; j < ctx_total; j++) for (int d = 0; d < D; d++) Kt[(size_t) d * ctx8 + j] = kc[(size_t) j * E + h * D + d];
        for (int j = ctx_total; j < ctx8; j++) for (int d = 0; d < D; d++) Kt[(size_t) d * ctx8 + j] = 0.0f;
        if (o.num.dot_order != 0) {
            // study orders: scores = dot over d (K row x Q row), mix = dot over the keys (V_trans row x probabilities), each in the
            // selected order (ggml: vec_dot_f32 on f32 operands, bark.cpp:1316,1333); softmax as in the canonical path
            std::vector<float> vcol((size_t) ctx8);
            for (int i = 0; i < N; i++) {
                const float * qi = q + (size_t) i * ldq + h * D;
                const int valid = causal ? std::min(ctx_total, n_past + i + 1) : ctx_total;
                for (int j = 0; j < valid; j++) {
                    const float * kj = kc + (size_t) j * E + h * D;
                    row[j] = (o.num.dot_order == 1 ? dot_ggml_avx2<false>((const uint8_t *) kj, qi, D) : dot_sequential<false>((const uint8_t *) kj, qi, D)) * scale;
                }
                softmax_row(row, valid);
                float * oi = out + (size_t) i * E + h * D;
                for (int d = 0; d < D; d++) {
                    for (int j = 0; j < valid; j++) vcol[(size_t) j] = vc[(size_t) j * E + h * D + d];
                    oi[d] = o.num.dot_order == 1 ? dot_ggml_avx2<false>((const uint8_t *) vcol.data(), row, valid) : dot_sequential<false>((const uint8_t *) vcol.data(), row, valid);
                }
            }
            continue;
        }
        for (int i = i_lo; i < i_hi; i++) {
            const float * qi = q + (size_t) i * ldq + h * D;
            const int valid = causal ? std::min(ctx_total, n_past + i + 1) : ctx_total;
            // C2: s[j] = four chains over the 16-d blocks of fmaf(K[j][d], Q[i][d], acc), then (c0 + c1) + (c2 + c3)   (f32 x f32, bark.cpp:1316)
            for (int j0 = 0; j0 < valid; j0 += 8) {
                __m256 blk[4];
                for (int b = 0; b < 4; b++) {
                    __m256 acc = _mm256_setzero_ps();
                    for (int d = 16 * b; d < 16 * b + 16; d++) acc = _mm256_fmadd_ps(_mm256_loadu_ps(Kt + (size_t) d * ctx8 + j0), _mm256_set1_ps(qi[d]), acc);
                    blk[b] = acc;
                }
                _mm256_storeu_ps(row + j0, _mm256_add_ps(_mm256_add_ps(blk[0], blk[1]), _mm256_add_ps(blk[2], blk[3])));
            }
            for (int j = 0; j < valid; j++) row[j] *= scale;                // ggml_scale_inplace, bark.cpp:1318
            softmax_row(row, valid);
            // C5: out[d] = 16 chains over j, tree-combined   (V_trans . P, bark.cpp:1324-1333)
            float * oi = out + (size_t) i * E + h * D;
            for (int d0 = 0; d0 < D; d0 += 8) {
                __m256 acc[16];
                for (int c = 0; c < 16; c++) acc[c] = _mm256_setzero_ps();
                const float * vp = vc + h * D + d0;
                int j = 0;
                for (; j + 16 <= valid; j += 16)
                    for (int c = 0; c < 16; c++)
                        acc[c] = _mm256_fmadd_ps(_mm256_loadu_ps(vp + (size_t) (j + c) * E), _mm256_set1_ps(row[j + c]), acc[c]);
                for (int c = 0; j + c < valid; c++)
                    acc[c] = _mm256_fmadd_ps(_mm256_loadu_ps(vp + (size_t) (j + c) * E), _mm256_set1_ps(row[j + c]), acc[c]);
                for (int c = 0; c < 16; c += 2) acc[c] = _mm256_add_ps(acc[c], acc[c + 1]);
                for (int c = 0; c < 16; c += 4) acc[c] = _mm256_add_ps(acc[c], acc[c + 2]);
                for (int c = 0; c < 16; c += 8) acc[c] = _mm256_add_ps(acc[c], acc[c + 4]);
                _mm256_storeu_ps(oi + d0, _mm256_add_ps(acc[0], acc[8]));
            }
        }
    }
}

static void add_bias_rows(float * y, size_t ld, int N, int M, const std::vector<float> & b, int nth = 1) {
    if (b.empty()) return;
    #pragma omp parallel for schedule(static) num_threads(nth) if (nth > 1 && N >= 16)
    for (int i = 0; i < N; i++) for (int m = 0; m < M; m++) y[i * ld + m] += b[m];
}

// One transformer block on N rows; `x` is updated in place.  Shared by the causal models
// (bark.cpp:1261-1389) and the fine model (bark.cpp:1474-1562), which differ only in mask/cache.
static void block_forward(Oracle & o, Gpt & m, int il, float * x, int N, int n_past, bool causal_cached, int nth) {
    const int E = m.n_embd, H = m.n_head;
    Layer & L = m.layers[il];
    o.xn.ensure((size_t) N * E); o.qkv.ensure((size_t) N * 3 * E); o.att.ensure((size_t) N * E);
    o.fc.ensure((size_t) N * 4 * E); o.tmp.ensure((size_t) N * E);
    float * xn = o.xn.p, * qkv = o.qkv.p, * att = o.att.p, * fc = o.fc.p, * tmp = o.tmp.p;

    #pragma omp parallel for schedule(static) num_threads(nth) if (nth > 1 && N >= 16)
    for (int i = 0; i < N; i++) layer_norm_row(x + (size_t) i * E, xn + (size_t) i * E, E, L.ln1_g.data(), L.ln1_b.empty() ? nullptr : L.ln1_b.data());
    if (L.attn_w.f16) round_rows(o, xn, (size_t) N * E, nth);     // f16 weights: activation -> f16; quantised weights: activation -> q8 inside the product; f32 weights: f32 x f32
    gemm_w(o, L.attn_w, xn, E, qkv, 3 * E, 3 * E, N, E, nth);
    add_bias_rows(qkv, 3 * E, N, 3 * E, L.attn_b, nth);

    if (causal_cached) {
        float * kc = m.mem_k + (size_t) il * m.block_size * E, * vc = m.mem_v + (size_t) il * m.block_size * E;
        for (int i = 0; i < N; i++) {
            memcpy(kc + (size_t) (n_past + i) * E, qkv + (size_t) i * 3 * E + E, E * 4);
            memcpy(vc + (size_t) (n_past + i) * E, qkv + (size_t) i * 3 * E + 2 * E, E * 4);
        }
        attention(o, qkv, 3 * E, kc, vc, att, N, n_past + N, n_past, true, E, H, nth);
    } else {
        // fine model: K/V are this pass's own rows, no mask (bark.cpp:1495-1523)
        o.logits.ensure((size_t) 2 * N * E);
        float * kc = o.logits.p, * vc = o.logits.p + (size_t) N * E;
        for (int i = 0; i < N; i++) {
            memcpy(kc + (size_t) i * E, qkv + (size_t) i * 3 * E + E, E * 4);
            memcpy(vc + (size_t) i * E, qkv + (size_t) i * 3 * E + 2 * E, E * 4);
        }
        attention(o, qkv, 3 * E, kc, vc, att, N, N, 0, false, E, H, nth);
    }

    if (L.proj_w.f16) round_rows(o, att, (size_t) N * E, nth);
    gemm_w(o, L.proj_w, att, E, tmp, E, E, N, E, nth);
    add_bias_rows(tmp, E, N, E, L.proj_b, nth);
    #pragma omp parallel for schedule(static) num_threads(nth) if (nth > 1 && N >= 16)
    for (size_t i = 0; i < (size_t) N * E; i++) x[i] = tmp[i] + x[i];          // cur + inpL  (bark.cpp:1352)

    #pragma omp parallel for schedule(static) num_threads(nth) if (nth > 1 && N >= 16)
    for (int i = 0; i < N; i++) layer_norm_row(x + (size_t) i * E, xn + (size_t) i * E, E, L.ln2_g.data(), L.ln2_b.empty() ? nullptr : L.ln2_b.data());
    if (L.fc_w.f16) round_rows(o, xn, (size_t) N * E, nth);
    gemm_w(o, L.fc_w, xn, E, fc, 4 * E, 4 * E, N, E, nth);
    add_bias_rows(fc, 4 * E, N, 4 * E, L.fc_b, nth);
    #pragma omp parallel for schedule(static) num_threads(nth) if (nth > 1 && N >= 16)
    for (size_t i = 0; i < (size_t) N * 4 * E; i++) fc[i] = gelu_apply(o, fc[i]);
    if (L.mproj_w.f16) round_rows(o, fc, (size_t) N * 4 * E, nth);
    gemm_w(o, L.mproj_w, fc, 4 * E, tmp, E, E, N, 4 * E, nth);
    add_bias_rows(tmp, E, N, E, L.mproj_b, nth);
    #pragma omp parallel for schedule(static) num_threads(nth) if (nth > 1 && N >= 16)
    for (size_t i = 0; i < (size_t) N * E; i++) x[i] = tmp[i] + x[i];          // cur + inpFF (bark.cpp:1388)
}

static void embed_row(const Tensor & wte, int id, float * out, int E) {
    if (const int bb = qblock_bytes(wte.ttype)) {
        // ggml_get_rows dequantises the row: w = q * d (+ m)
        const uint8_t * row = wte.data + (size_t) id * (E / 32) * bb;
        const bool has_m = wte.ttype == 3 || wte.ttype == 7;
        for (int b = 0; b < E / 32; b++) {
            int q[32]; float d, m;
            qblock_unpack(wte.ttype, row + (size_t) b * bb, q, d, m);
            for (int j = 0; j < 32; j++) { const float v = (float) q[j] * d; out[b * 32 + j] = has_m ? v + m : v; }
        }
        return;
    }
    const uint8_t * row = wte.data + (size_t) id * E * (wte.ttype == 1 ? 2 : 4);
    for (int e = 0; e < E; e++) out[e] = wte.ttype == 1 ? h2f(ld_u16(row, e)) : ld_f32(row, e);
}

// bark_build_gpt_graph + bark_eval_encoder_internal (bark.cpp:1186-1414,1586-1643).
// `tokens`: n_tokens ids; with merge_ctx at n_past == 0 the 513-id prompt collapses to 257 rows.
// Writes n_out logits of the LAST row; advances *n_past exactly as the reference does.
static bool gpt_eval(Oracle & o, Gpt & m, const int32_t * tokens, int n_tokens, int * n_past, bool merge_ctx, float * logits, int nth) {
    const int64_t t0 = now_us();
    const int E = m.n_embd;
    int N = n_tokens;
    const bool merge = merge_ctx && *n_past == 0;
    if (*n_past > 0 && N != 1) return false;                               // bark.cpp:1227
    if (merge) { if (N != 513) return false; N -= 256; }                   // bark.cpp:1231-1232
    if (*n_past + N > m.block_size) return false;
    for (int i = 0; i < n_tokens; i++) if (tokens[i] < 0 || tokens[i] >= m.n_in) return false;
    o.x.ensure((size_t) N * E);
    float * x = o.x.p;
    std::vector<float> a(E), b(E);
    for (int i = 0; i < N; i++) {
        float * xi = x + (size_t) i * E;
        if (merge && i < 256) {                                            // bark.cpp:1237-1248
            embed_row(m.wtes[0], tokens[i], a.data(), E);
            embed_row(m.wtes[0], tokens[256 + i], b.data(), E);
            for (int e = 0; e < E; e++) xi[e] = a[e] + b[e];
        } else if (merge) {
            embed_row(m.wtes[0], tokens[512], xi, E);
        } else {
            embed_row(m.wtes[0], tokens[i], xi, E);
        }
        const float * pe = m.wpe.data() + (size_t) (*n_past + i) * E;      // position = i + n_past (bark.cpp:1617-1620)
        for (int e = 0; e < E; e++) xi[e] = xi[e] + pe[e];
    }
    for (int il = 0; il < m.n_layer; il++) block_forward(o, m, il, x, N, *n_past, true, nth);
    // final norm + LM head on the last row only (bark.cpp:1391-1405)
    std::vector<float> last(E);
    layer_norm_row(x + (size_t) (N - 1) * E, last.data(), E, m.lnf_g.data(), m.lnf_b.empty() ? nullptr : m.lnf_b.data());
    if (m.lm_heads[0].f16) round_rows(o, last.data(), E);
    gemm_w(o, m.lm_heads[0], last.data(), E, logits, m.n_out, m.n_out, 1, E, nth);
    *n_past += N;
    m.t_predict_us += now_us() - t0;
    return true;
}

// bark_build_fine_gpt_graph + bark_eval_fine_encoder_internal (bark.cpp:1416-1584,1907-1959).
// tokens: [8][1024] codebook-major.  logits: [1024][n_out].
static bool fine_eval(Oracle & o, const int32_t * tokens, int nn, float * logits, int nth) {
    Gpt & m = o.fine;
    const int64_t t0 = now_us();
    const int E = m.n_embd, N = 1024;
    if (nn < 1 || nn >= m.n_wtes || nn - 1 >= m.n_lm_heads) return false;
    o.x.ensure((size_t) N * E);
    float * x = o.x.p;
    std::vector<float> r(E);
    for (int i = 0; i < N; i++) {
        float * xi = x + (size_t) i * E;
        for (int e = 0; e < E; e++) xi[e] = 0.0f;                          // ggml_set_zero(tok_emb), bark.cpp:1936-1937
        for (int c = 0; c <= nn; c++) {                                    // bark.cpp:1457-1463
            int id = tokens[c * N + i];
            if (id < 0 || id >= m.n_in) return false;
            embed_row(m.wtes[c], id, r.data(), E);
            for (int e = 0; e < E; e++) xi[e] = xi[e] + r[e];
        }
        const float * pe = m.wpe.data() + (size_t) i * E;
        for (int e = 0; e < E; e++) xi[e] = xi[e] + pe[e];
    }
    for (int il = 0; il < m.n_layer; il++) block_forward(o, m, il, x, N, 0, false, nth);
    o.xn.ensure((size_t) N * E);
    for (int i = 0; i < N; i++) layer_norm_row(x + (size_t) i * E, o.xn.p + (size_t) i * E, E, m.lnf_g.data(), m.lnf_b.empty() ? nullptr : m.lnf_b.data());
    if (m.lm_heads[nn - 1].f16) round_rows(o, o.xn.p, (size_t) N * E);
    gemm_w(o, m.lm_heads[nn - 1], o.xn.p, E, logits, m.n_out, m.n_out, N, E, nth);   // lm_heads[codebook_idx - n_codes_given]
    m.t_predict_us += now_us() - t0;
    return true;
}

// ------------------------------------------------------------------------------------
// sampling (bark.cpp:184-270)
// ------------------------------------------------------------------------------------
static void softmax_inplace(std::vector<float> & logits) {
    float maxl = -INFINITY;
    for (float l : logits) maxl = std::max(maxl, l);
    float sum = 0.0f;
    for (float & l : logits) { l = (float) exp((double) (l - maxl)); sum += l; }   // ::exp(double) on a float argument
    for (float & l : logits) l /= sum;
}
static int32_t sample_argmax(std::vector<float> & logits, float * eos_p) {
    for (float & l : logits) l /= 0.7f;
    softmax_inplace(logits);
    if (eos_p) *eos_p = logits.back();
    int next = 0; float maxl = -INFINITY;
    for (int i = 0; i < (int) logits.size(); i++) if (logits[i] > maxl) { maxl = logits[i]; next = i; }
    return next;
}
static int32_t sample_multinomial(std::vector<float> & logits, std::mt19937 & rng, float temp, float * eos_p) {
    for (float & l : logits) l /= temp;
    softmax_inplace(logits);
    std::discrete_distribution<int32_t> dist(logits.begin(), logits.end());
    int next = dist(rng);
    if (eos_p) *eos_p = logits.back();
    return next;
}
static int32_t gpt_sample(std::vector<float> & logits, std::mt19937 & rng, float temp, float * eos_p, Gpt & m) {
    const int64_t t0 = now_us();
    int32_t r = temp == 0.0f ? sample_argmax(logits, eos_p) : sample_multinomial(logits, rng, temp, eos_p);
    m.t_sample_us += now_us() - t0; m.n_sample += 1;
    return r;
}

// ------------------------------------------------------------------------------------
// tokenizer (bark.cpp:480-662)
// ------------------------------------------------------------------------------------
static size_t utf8_len(char c) {
    static const size_t lookup[16] = {1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 2, 2, 3, 4};
    return lookup[((uint8_t) c) >> 4];
}
static std::string strip_accents(const std::string & in) {
    // Latin-1 letters folded to ASCII (bark.cpp:488-541), as (utf-8 bytes c3 XX -> ascii)
    static const struct { unsigned char lo, hi; char to; } ranges[] = {
        {0x80, 0x85, 'A'}, {0xa0, 0xa5, 'a'}, {0x88, 0x8b, 'E'}, {0xa8, 0xab, 'e'}, {0x8c, 0x8f, 'I'}, {0xac, 0xaf, 'i'},
        {0x92, 0x96, 'O'}, {0xb2, 0xb6, 'o'}, {0x99, 0x9c, 'U'}, {0xb9, 0xbc, 'u'}, {0x9d, 0x9d, 'Y'}, {0xbd, 0xbd, 'y'},
        {0x87, 0x87, 'C'}, {0xa7, 0xa7, 'c'}, {0x91, 0x91, 'N'}, {0xb1, 0xb1, 'n'}};
    std::string out;
    for (size_t i = 0; i < in.size();) {
        size_t len = utf8_len(in[i]);
        bool mapped = false;
        if (len == 2 && i + 1 < in.size() && (unsigned char) in[i] == 0xc3) {
            unsigned char b = (unsigned char) in[i + 1];
            for (auto & r : ranges) if (b >= r.lo && b <= r.hi) { out += r.to; mapped = true; break; }
        }
        if (!mapped) out += in.substr(i, len);
        i += len;
    }
    return out;
}
static void bert_tokenize(const Oracle & o, const char * text, int32_t * tokens, int32_t * n_tokens, int32_t n_max) {
    std::string str = strip_accents(text);
    std::vector<std::string> words;
    {
        std::regex re(R"([[:punct:]]|[[:alpha:]]+|[[:digit:]]+)");
        std::smatch m;
        while (std::regex_search(str, m, re)) {
            for (std::string x : m) words.push_back(x);
            str = m.suffix();
        }
    }
    int32_t t = 0;
    for (const auto & word : words) {
        if (word.empty()) continue;
        std::string prefix;
        int i = 0, n = (int) word.size();
        while (i < n) {
            if (t >= n_max - 1) break;                                       // bark.cpp:598-599
            int j = n; bool hit = false;
            while (j > i) {
                auto it = o.token_to_id.find(prefix + word.substr(i, j - i));
                if (it != o.token_to_id.end()) { tokens[t++] = it->second; i = j; prefix = "##"; hit = true; break; }
                --j;
            }
            if (!hit) { prefix = "##"; ++i; }                               // unknown byte: skip it (bark.cpp:611-615)
        }
    }
    *n_tokens = t;
}

struct Params {   // mirrors bark_context_params (bark.h:81-141) minus callbacks
    float temp = 0.7f, fine_temp = 0.5f, min_eos_p = 0.2f;
    int sliding_window_size = 60, max_coarse_history = 630, sample_rate = 24000, target_bandwidth = 6;
    int n_steps_text_encoder = 768, text_pad_token = 129595, text_encoding_offset = 10048;
    float semantic_rate_hz = 49.9f;
    int semantic_pad_token = 10000, semantic_vocab_size = 10000, semantic_infer_token = 129599;
    float coarse_rate_hz = 75.0f;
    int coarse_infer_token = 12050, coarse_semantic_pad_token = 12048, n_coarse_codebooks = 2, n_fine_codebooks = 8, codebook_size = 1024;
};

// bark_tokenize_input (bark.cpp:622-662) -> 513 ids
static void tokenize_input(const Oracle & o, const Params & p, const char * text, std::vector<int32_t> & out) {
    const int max_ctx = std::min(o.sem.block_size, 256);
    std::vector<int32_t> tokens(max_ctx, 0);
    int32_t n_tokens = 0;
    bert_tokenize(o, text, tokens.data(), &n_tokens, max_ctx);
    for (auto & t : tokens) t += p.text_encoding_offset;
    for (int i = n_tokens; i < max_ctx; i++) tokens[i] = p.text_pad_token;
    for (int i = 0; i < 256; i++) tokens.push_back(p.semantic_pad_token);
    tokens.push_back(p.semantic_infer_token);
    out = tokens;
}

// ------------------------------------------------------------------------------------
// stage loops
// ------------------------------------------------------------------------------------
// bark_eval_text_encoder (bark.cpp:1645-1701).  eos_trace (optional) receives eos_p per step.
static bool semantic_stage(Oracle & o, const Params & p, const std::vector<int32_t> & prompt, std::vector<int32_t> & out,
                           std::vector<float> * eos_trace, int nth) {
    Gpt & m = o.sem;
    const int64_t t0 = now_us();
    std::vector<int32_t> input = prompt;
    std::vector<float> logits(m.n_out);
    float eos_p = 0; int n_past = 0;
    out.clear();
    for (int i = 0; i < p.n_steps_text_encoder; i++) {
        if (!gpt_eval(o, m, input.data(), (int) input.size(), &n_past, true, logits.data(), nth)) return false;
        input.clear();
        // NOTE the reference samples over ALL n_out logits; its sliced copy is unused (bark.cpp:1682-1688)
        std::vector<float> l = logits;
        int32_t next = gpt_sample(l, o.rng, p.temp, &eos_p, m);
        if (eos_trace) eos_trace->push_back(eos_p);
        if (next == p.semantic_vocab_size || eos_p >= p.min_eos_p) break;
        input.push_back(next); out.push_back(next);
    }
    m.t_main_us = now_us() - t0;
    return true;
}

// bark_eval_coarse_encoder (bark.cpp:1745-1863) -> out_coarse [T][2]
static bool coarse_stage(Oracle & o, const Params & p, const std::vector<int32_t> & semantic, std::vector<int32_t> & out_flat, int nth) {
    Gpt & m = o.coarse;
    const int64_t t0 = now_us();
    std::vector<int32_t> out;
    std::vector<float> logits(m.n_out);
    const float stc_ratio = p.coarse_rate_hz / p.semantic_rate_hz * p.n_coarse_codebooks;
    const int max_semantic_history = (int) floorf(p.max_coarse_history / stc_ratio);
    const int n_steps = (int) (floorf(semantic.size() * stc_ratio / p.n_coarse_codebooks) * p.n_coarse_codebooks);
    if (n_steps <= 0) return false;
    const int n_window_steps = (int) ceilf((float) n_steps / p.sliding_window_size);
    int step_idx = 0;
    for (int i = 0; i < n_window_steps; i++) {
        const int semantic_idx = (int) roundf(step_idx / stc_ratio);
        std::vector<int32_t> input_in(semantic.begin() + std::max(semantic_idx - max_semantic_history, 0), semantic.end());
        size_t original_size = input_in.size();
        input_in.resize(256);
        for (size_t ix = original_size; ix < 256; ix++) input_in[ix] = p.coarse_semantic_pad_token;
        input_in.push_back(p.coarse_infer_token);
        const int nh = std::min(p.max_coarse_history, (int) out.size());
        input_in.insert(input_in.end(), out.end() - nh, out.end());
        int n_past = 0;
        for (int j = 0; j < p.sliding_window_size; j++) {
            if (step_idx >= n_steps) continue;
            if (!gpt_eval(o, m, input_in.data(), (int) input_in.size(), &n_past, false, logits.data(), nth)) return false;
            input_in.clear();
            const bool is_major = step_idx % p.n_coarse_codebooks == 0;
            const int start_idx = p.semantic_vocab_size + (1 - is_major) * p.codebook_size;
            const int end_idx = p.semantic_vocab_size + (2 - is_major) * p.codebook_size;
            std::vector<float> relevant(logits.begin() + start_idx, logits.begin() + end_idx);
            int32_t next = gpt_sample(relevant, o.rng, p.temp, nullptr, m);
            next += start_idx;
            input_in.push_back(next); out.push_back(next);
            step_idx += 1;
        }
    }
    out_flat.clear();
    for (size_t i = 0; i + 1 < out.size(); i += 2) {
        out_flat.push_back(out[i] - p.semantic_vocab_size);
        out_flat.push_back(out[i + 1] - p.semantic_vocab_size - p.codebook_size);
    }
    m.t_main_us = now_us() - t0;
    return true;
}

// bark_eval_fine_encoder (bark.cpp:1961-2059).  coarse: [T][2] -> fine: [T][8].
// T > 1024 (several windows): implemented; the one place where the reference's store index is undefined behaviour is
// restated from the algorithm it was ported from - see the comment at the store below.
static bool fine_stage(Oracle & o, const Params & p, const std::vector<int32_t> & coarse, std::vector<int32_t> & fine_out, int nth) {
    Gpt & m = o.fine;
    const int64_t t0 = now_us();
    const int nc = p.n_coarse_codebooks, nf = p.n_fine_codebooks, cs = p.codebook_size;
    const int T = (int) coarse.size() / nc;
    if (T <= 0 || nf != 8) return false;
    std::vector<std::vector<int32_t>> in_arr;
    for (int i = 0; i < T; i++) {
        std::vector<int32_t> row(coarse.begin() + i * nc, coarse.begin() + (i + 1) * nc);
        for (int j = nc; j < nf; j++) row.push_back(cs);
        in_arr.push_back(row);
    }
    int n_remove_from_end = 0;
    if (T < 1024) { n_remove_from_end = 1024 - T; for (int i = T; i < 1024; i++) in_arr.push_back(std::vector<int32_t>(nf, cs)); }
    const int n_loops = std::max(0, (int) ceilf((in_arr.size() - 1024) / 512.f)) + 1;
    std::vector<float> logits((size_t) 1024 * m.n_out);
    for (int n = 0; n < n_loops; n++) {
        const int start_idx = std::min(n * 512, (int) in_arr.size() - 1024);
        const int start_fill_idx = std::min(n * 512, (int) in_arr.size() - 512);
        const int rel = start_fill_idx - start_idx;
        std::vector<int32_t> in_buffer;
        for (int i = 0; i < nf; i++) for (int j = start_idx; j < start_idx + 1024; j++) in_buffer.push_back(in_arr[j][i]);
        for (int nn = nc; nn < nf; nn++) {
            if (!fine_eval(o, in_buffer.data(), nn, logits.data(), nth)) return false;
            for (int i = 0; i < 1024; i++) {
                std::vector<float> relevant(logits.begin() + (size_t) i * m.n_out, logits.begin() + (size_t) i * m.n_out + cs);
                int32_t next = gpt_sample(relevant, o.rng, p.fine_temp, nullptr, m);
                // The reference stores at [rel + i] (bark.cpp:2037), which for T > 1024 (rel > 0) runs into the next codebook's
                // row and past the end of the buffer (undefined behaviour).  Restated as the algorithm it was
                // ported from (suno-ai/bark generation.py, generate_fine: in_buffer[rel:, nn] = preds[rel:]): every position is
                // sampled - the random stream advances exactly as in the reference - and positions >= rel keep their sample.
                // For T <= 1024 (rel == 0) both readings coincide.
                if (i >= rel) in_buffer[nn * 1024 + i] = next;
            }
        }
        for (int nn = nc; nn < nf; nn++) for (int j = 0; j < cs - rel; j++) in_arr[start_fill_idx + j][nn] = in_buffer[nn * 1024 + rel + j];
    }
    if (n_remove_from_end > 0) in_arr.resize(in_arr.size() - n_remove_from_end);
    fine_out.clear();
    for (auto & row : in_arr) for (int v : row) fine_out.push_back(v);
    m.t_main_us = now_us() - t0;
    return true;
}

// ------------------------------------------------------------------------------------
// EnCodec decoder (HF modeling_encodec.py; numerics: ggml conv/LSTM matmuls take f16-rounded
// activations against f16 weights with f32 accumulation, everything else f32)
// ------------------------------------------------------------------------------------
// Canonical transcendental: evaluated in double and rounded once to float, so that the host libm and
// the GPU's device libm (both < 1 ulp in double) produce the same float.
static float elu(float x) { return x > 0.f ? x : (float) expm1((double) x); }

// EncodecConv1d._pad1d with mode="reflect" (modeling_encodec.py:140-157), causal: (left = k - stride, right = extra)
static std::vector<float> reflect_pad(const std::vector<float> & x, int C, int T, int left, int right, int & Tp) {
    int len = T, extra = 0;
    const int max_pad = std::max(left, right);
    std::vector<float> src = x;
    if (len <= max_pad) {      // too short to reflect: zero-extend on the right first
        extra = max_pad - len + 1;
        std::vector<float> s2((size_t) C * (len + extra), 0.f);
        for (int c = 0; c < C; c++) for (int t = 0; t < len; t++) s2[(size_t) c * (len + extra) + t] = x[(size_t) c * len + t];
        src.swap(s2); len += extra;
    }
    int full = left + len + right;
    std::vector<float> out((size_t) C * full);
    for (int c = 0; c < C; c++) {
        const float * s = src.data() + (size_t) c * len; float * d = out.data() + (size_t) c * full;
        for (int i = 0; i < left; i++) d[i] = s[left - i];
        for (int i = 0; i < len; i++) d[left + i] = s[i];
        for (int i = 0; i < right; i++) d[left + len + i] = s[len - 2 - i];
    }
    Tp = full - extra;
    if (extra) {
        std::vector<float> o2((size_t) C * Tp);
        for (int c = 0; c < C; c++) for (int t = 0; t < Tp; t++) o2[(size_t) c * Tp + t] = out[(size_t) c * full + t];
        return o2;
    }
    return out;
}

// C9m: the convolution as a product on the f16 matrix cores (gemm_mfma), y[t][co] = C1m-dot(Wm[co], col_t) over kd = k * cin + ci
static bool conv_uses_mfma(const Oracle & o, int cin) { return o.num.codec_mfma && o.num.act_round_f16 && o.num.dot_order == 0 && (cin & 7) == 0; }
static std::vector<float> conv1d_mfma(const Conv & cv, const std::vector<float> & xp, int Tp, int T, int nth) {
    const int kd = cv.k * cv.cin;
    if (cv.wm.empty()) {
        cv.wm_bits.resize((size_t) cv.cout * kd);
        for (int co = 0; co < cv.cout; co++) for (int ci = 0; ci < cv.cin; ci++) for (int k = 0; k < cv.k; k++)
            cv.wm_bits[(size_t) co * kd + (size_t) k * cv.cin + ci] = f2h(cv.w[((size_t) co * cv.cin + ci) * cv.k + k]);
        cv.wm.resize(1);
        CanonW & W = cv.wm[0]; W.M = cv.cout; W.K = kd; W.f16 = true; W.mfma = true; W.raw = (const uint8_t *) cv.wm_bits.data();
    }
    std::vector<float> X((size_t) T * kd), Y((size_t) T * cv.cout);
    #pragma omp parallel for schedule(static) num_threads(nth) if (nth > 1)
    for (int t = 0; t < T; t++)
        for (int k = 0; k < cv.k; k++) for (int ci = 0; ci < cv.cin; ci++) X[(size_t) t * kd + (size_t) k * cv.cin + ci] = xp[(size_t) ci * Tp + t + k];
    gemm_mfma(cv.wm[0], X.data(), kd, Y.data(), cv.cout, cv.cout, T, kd, nth);
    std::vector<float> y((size_t) cv.cout * T);
    for (int co = 0; co < cv.cout; co++) for (int t = 0; t < T; t++) y[(size_t) co * T + t] = Y[(size_t) t * cv.cout + co] + cv.b[co];
    return y;
}

// causal stride-1 conv: out[co][t] = b[co] + sum_{ci,k} w[co][ci][k] * xpad[ci][t + k]   (modeling_encodec.py:159-176)
static std::vector<float> conv1d(const Oracle & o, const Conv & cv, const std::vector<float> & x, int T, int nth) {
    int Tp = 0;
    std::vector<float> xp = reflect_pad(x, cv.cin, T, cv.k - 1, 0, Tp);
    if (o.num.act_round_f16) for (float & v : xp) v = round_h(v);       // im2col to f16
    if (conv_uses_mfma(o, cv.cin)) return conv1d_mfma(cv, xp, Tp, T, nth);
    std::vector<float> y((size_t) cv.cout * T);
    #pragma omp parallel for schedule(static) num_threads(nth) if (nth > 1)
    for (int co = 0; co < cv.cout; co++) {
        float * yo = y.data() + (size_t) co * T;
        for (int t = 0; t < T; t++) yo[t] = 0.f;
        for (int ci = 0; ci < cv.cin; ci++) {
            const float * xi = xp.data() + (size_t) ci * Tp;
            for (int k = 0; k < cv.k; k++) {
                const float w = cv.w[((size_t) co * cv.cin + ci) * cv.k + k];
                for (int t = 0; t < T; t++) yo[t] = fmaf(w, xi[t + k], yo[t]);
            }
        }
        for (int t = 0; t < T; t++) yo[t] += cv.b[co];
    }
    return y;
}

// causal transposed conv, full output (T-1)*s + k trimmed by (k - s) on the right (modeling_encodec.py:206-233)
static std::vector<float> convtr1d(const Oracle & o, const ConvT & cv, const std::vector<float> & x, int T, int & Tout, int nth) {
    const int s = cv.stride, K = cv.k;
    Tout = T * s;                                  // (T-1)*s + K - (K - s)
    std::vector<float> xr = x;
    if (o.num.act_round_f16) for (float & v : xr) v = round_h(v);
    if (conv_uses_mfma(o, cv.cin) && K == 2 * s) {
        // C9m: output phase r (to = q s + r) is one product over kd = tap * cin + ci: tap 0 = frame q - 1 with kernel element r + s, tap 1 = frame q
        // with element r (the frame in front of the first one is zero)
        const int kd = 2 * cv.cin;
        if (cv.wm.empty()) {
            cv.wm_bits.resize((size_t) s * cv.cout * kd);
            for (int r = 0; r < s; r++) for (int co = 0; co < cv.cout; co++) for (int ci = 0; ci < cv.cin; ci++) {
                const float * w = cv.w.data() + ((size_t) ci * cv.cout + co) * K;
                cv.wm_bits[((size_t) r * cv.cout + co) * kd + ci] = f2h(w[r + s]);
                cv.wm_bits[((size_t) r * cv.cout + co) * kd + cv.cin + ci] = f2h(w[r]);
            }
            cv.wm.resize((size_t) s);
            for (int r = 0; r < s; r++) { CanonW & W = cv.wm[(size_t) r]; W.M = cv.cout; W.K = kd; W.f16 = true; W.mfma = true; W.raw = (const uint8_t *) (cv.wm_bits.data() + (size_t) r * cv.cout * kd); }
        }
        std::vector<float> X((size_t) T * kd), Y((size_t) T * cv.cout), ym((size_t) cv.cout * Tout);
        for (int q = 0; q < T; q++) for (int ci = 0; ci < cv.cin; ci++) {
            X[(size_t) q * kd + ci] = q > 0 ? xr[(size_t) ci * T + q - 1] : 0.0f;
            X[(size_t) q * kd + cv.cin + ci] = xr[(size_t) ci * T + q];
        }
        for (int r = 0; r < s; r++) {
            gemm_mfma(cv.wm[(size_t) r], X.data(), kd, Y.data(), cv.cout, cv.cout, T, kd, nth);
            for (int co = 0; co < cv.cout; co++) for (int q = 0; q < T; q++) ym[(size_t) co * Tout + (size_t) q * s + r] = Y[(size_t) q * cv.cout + co] + cv.b[co];
        }
        return ym;
    }
    std::vector<float> y((size_t) cv.cout * Tout);
    #pragma omp parallel for schedule(static) num_threads(nth) if (nth > 1)
    for (int co = 0; co < cv.cout; co++) {
        float * yo = y.data() + (size_t) co * Tout;
        for (int t = 0; t < Tout; t++) yo[t] = 0.f;
        for (int ci = 0; ci < cv.cin; ci++) {
            const float * xi = xr.data() + (size_t) ci * T;
            const float * w = cv.w.data() + ((size_t) ci * cv.cout + co) * K;
            for (int t = 0; t < T; t++) {
                const float xv = xi[t];
                for (int k = 0; k < K; k++) { int to = t * s + k; if (to < Tout) yo[to] = fmaf(w[k], xv, yo[to]); }
            }
        }
        for (int t = 0; t < Tout; t++) yo[t] += cv.b[co];
    }
    return y;
}

// one LSTM layer over [D][T] (modeling_encodec.py:236-249; PyTorch gate order i,f,g,o)
static std::vector<float> lstm_layer(const Oracle & o, const Lstm & L, const std::vector<float> & x, int D, int T, int nth) {
    std::vector<float> hs((size_t) D * T), h(D, 0.f), c(D, 0.f), xt(D), hr(D), gi(4 * D), gh(4 * D);
    for (int t = 0; t < T; t++) {
        for (int d = 0; d < D; d++) { xt[d] = x[(size_t) d * T + t]; hr[d] = h[d]; }
        if (o.num.act_round_f16) { for (float & v : xt) v = round_h(v); for (float & v : hr) v = round_h(v); }
        gemm_w(const_cast<Oracle &>(o), L.w_ih, xt.data(), D, gi.data(), 4 * D, 4 * D, 1, D, nth);     // C1 order
        gemm_w(const_cast<Oracle &>(o), L.w_hh, hr.data(), D, gh.data(), 4 * D, 4 * D, 1, D, nth);
        for (int d = 0; d < D; d++) {
            auto gate = [&](int g) { return (gi[g * D + d] + L.b_ih[g * D + d]) + (gh[g * D + d] + L.b_hh[g * D + d]); };
            const float i_t = 1.f / (1.f + (float) exp((double) (-gate(0))));
            const float f_t = 1.f / (1.f + (float) exp((double) (-gate(1))));
            const float g_t = (float) tanh((double) gate(2));
            const float o_t = 1.f / (1.f + (float) exp((double) (-gate(3))));
            c[d] = f_t * c[d] + i_t * g_t;
            h[d] = o_t * (float) tanh((double) c[d]);
            hs[(size_t) d * T + t] = h[d];
        }
    }
    return hs;
}

// codes: [n_q][T] (time contiguous, bark.cpp:2153-2161).  pcm: 320*T samples (for the 24 kHz ratios).
// tap: if tap_stage >= 0, *tap receives the activation after that stage (0 first conv, 1 LSTM+skip, 2..5 up-blocks)
static bool codec_decode(Oracle & o, const int32_t * codes, int n_q, int T, std::vector<float> & pcm, int nth, int tap_stage = -1, std::vector<float> * tap = nullptr) {
    Codec & c = o.codec;
    const int H = c.hidden_dim;
    if (n_q <= 0 || n_q > (int) c.codebooks.size() || T <= 0) return false;
    // RVQ decode: quantized_out = 0 + sum_q embed_q[code]  (modeling_encodec.py:440-448)
    std::vector<float> z((size_t) H * T, 0.f);
    for (int q = 0; q < n_q; q++) for (int t = 0; t < T; t++) {
        int id = codes[(size_t) q * T + t];
        if (id < 0 || id >= c.n_bins) return false;
        const float * e = c.codebooks[q].data() + (size_t) id * H;
        for (int d = 0; d < H; d++) z[(size_t) d * T + t] = z[(size_t) d * T + t] + e[d];
    }
    std::vector<float> x = conv1d(o, c.init, z, T, nth);
    const int D = c.init.cout;
    if (tap_stage == 0) *tap = x;
    {   // 2-layer LSTM with skip connection
        std::vector<float> y = lstm_layer(o, c.lstm[0], x, D, T, nth);
        y = lstm_layer(o, c.lstm[1], y, D, T, nth);
        for (size_t i = 0; i < x.size(); i++) x[i] = y[i] + x[i];
    }
    if (tap_stage == 1) *tap = x;
    int Tc = T;
    for (int b = 0; b < 4; b++) {
        auto & B = c.blocks[b];
        for (float & v : x) v = elu(v);
        int Tn = 0;
        x = convtr1d(o, B.up, x, Tc, Tn, nth); Tc = Tn;
        // residual block: shortcut(x) + conv2(elu(conv1(elu(x))))   (modeling_encodec.py:252-282)
        std::vector<float> r = x;
        for (float & v : r) v = elu(v);
        r = conv1d(o, B.c1, r, Tc, nth);
        for (float & v : r) v = elu(v);
        r = conv1d(o, B.c2, r, Tc, nth);
        std::vector<float> s = conv1d(o, B.sc, x, Tc, nth);
        for (size_t i = 0; i < s.size(); i++) s[i] = s[i] + r[i];
        x.swap(s);
        if (tap_stage == 2 + b) *tap = x;
    }
    for (float & v : x) v = elu(v);
    pcm = conv1d(o, c.fin, x, Tc, nth);
    return true;
}

static Oracle * oracle_open(const char * path) {
    int fd = open(path, O_RDONLY);
    if (fd < 0) { fprintf(stderr, "oracle: cannot open %s\n", path); return nullptr; }
    struct stat st; fstat(fd, &st);
    void * p = mmap(nullptr, st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
    close(fd);
    if (p == MAP_FAILED) return nullptr;
    std::unique_ptr<Oracle> o(new Oracle());
    o->map = (const uint8_t *) p; o->map_size = st.st_size;
    Reader r{o->map, o->map_size};
    if (r.get<uint32_t>() != 0x67676d6c) { fprintf(stderr, "oracle: bad magic\n"); return nullptr; }
    int32_t n_vocab = r.get<int32_t>();
    for (int i = 0; i < n_vocab && r.ok; i++) {
        uint32_t len = r.get<uint32_t>();
        const uint8_t * w = r.skip(len);
        if (r.ok) o->token_to_id[std::string((const char *) w, len)] = i;
    }
    if (!r.ok) return nullptr;
    if (!load_gpt(r, o->sem, true) || !load_gpt(r, o->coarse, true) || !load_gpt(r, o->fine, false)) return nullptr;
    if (!load_codec(r, o->codec)) return nullptr;
    for (Layer & L : o->fine.layers) L.attn_w.mfma = L.proj_w.mfma = L.fc_w.mfma = L.mproj_w.mfma = true;
    for (CanonW & W : o->fine.lm_heads) W.mfma = true;
    build_gelu_table(o->gelu_table);
    return o.release();
}

}  // namespace

// =====================================================================================
// C ABI used by the tests (ctypes)
// =====================================================================================
extern "C" {

void * orc_open(const char * path) { return oracle_open(path); }
void   orc_close(void * h) { delete (Oracle *) h; }
void   orc_set_numerics(void * h, int act_round_f16, int gelu_mode) { auto * o = (Oracle *) h; o->num.act_round_f16 = act_round_f16; o->num.gelu_mode = gelu_mode; }
void   orc_set_dot_order(void * h, int dot_order) { ((Oracle *) h)->num.dot_order = dot_order; }      // study modes, see Numerics
void   orc_set_codec_mfma(void * h, int on) { ((Oracle *) h)->num.codec_mfma = on; }                      // 0: the codec's convolutions as (ci, k) fmaf chains (Numerics::codec_mfma)
void   orc_set_fine_mfma(void * h, int on) { ((Oracle *) h)->num.fine_mfma = on; }                        // 0: C1 chains for the fine model too (Numerics::fine_mfma)
// y[n][m] = C1m-dot(w[m], x[n]) through the 8-lane restatement (tests/test_mfma_f16_emu.py): w [M][K] f16 bits, x [N][K] floats (f16-representable)
void   orc_test_mfma_gemm(const uint16_t * w, const float * x, int M, int N, int K, float * y) {
    CanonW W; W.M = M; W.K = K; W.f16 = true; W.raw = (const uint8_t *) w; W.mfma = true;
    gemm_mfma(W, x, K, y, M, M, N, K, 1);
}
void   orc_seed(void * h, uint32_t seed) { ((Oracle *) h)->rng = std::mt19937(seed); }
const uint16_t * orc_gelu_table(void * h) { return ((Oracle *) h)->gelu_table.data(); }

// ---- unit hooks: the canonical summation orders, exercised in isolation (tests/test_canon_orders.py) ----
// y = C1-dot(w, x): w holds K f16 bit patterns, x K floats (used as given)
float orc_test_wdot(const uint16_t * w, const float * x, int K) {
    Oracle o; CanonW W; W.build((const uint8_t *) w, true, 1, K);
    float y = 0.f;
    gemm_w(o, W, x, K, &y, 1, 1, 1, K, 1);
    return y;
}
// single-head attention (E = H * 64 with H = 1): q [N][64], kc/vc [ctx][64], out [N][64]
void orc_test_attention(const float * q, const float * kc, const float * vc, int N, int ctx, int n_past, int causal, float * out) {
    Oracle o;
    attention(o, q, 64, kc, vc, out, N, ctx, n_past, causal != 0, 64, 1, 1);
}
// y = C1q dot of one q4_0 row (K/32 blocks of 18 bytes) with the f32 row x (quantised to q8_0 as mul_mat does)
float orc_test_q4dot(const uint8_t * blocks, const float * x, int K) {
    Q8Row r; quantize_row_q8(x, K, r);
    return dot_q_q8(2, blocks, r, K);
}
// the same for any restated block format (qtype = ggml_type of the weight row)
float orc_test_qdot(int qtype, const uint8_t * blocks, const float * x, int K) {
    if (!qblock_bytes(qtype)) return NAN;
    Q8Row r; quantize_row_q8(x, K, r);
    return dot_q_q8(qtype, blocks, r, K);
}
void orc_test_layer_norm(const float * x, float * y, int E, const float * g, const float * b) { layer_norm_row(x, y, E, g, b); }

// which: 0 semantic, 1 coarse, 2 fine.  out[10]: n_layer,n_head,n_embd,block_size,bias,n_in,n_out,n_lm_heads,n_wtes,ftype
void orc_hparams(void * h, int which, int32_t * out) {
    auto * o = (Oracle *) h; Gpt & m = which == 0 ? o->sem : which == 1 ? o->coarse : o->fine;
    int32_t v[10] = {m.n_layer, m.n_head, m.n_embd, m.block_size, m.bias, m.n_in, m.n_out, m.n_lm_heads, m.n_wtes, m.ftype};
    memcpy(out, v, sizeof(v));
}
int orc_tokenize(void * h, const char * text, int32_t * out513) {
    auto * o = (Oracle *) h; Params p; std::vector<int32_t> t; tokenize_input(*o, p, text, t);
    memcpy(out513, t.data(), t.size() * 4); return (int) t.size();
}
int orc_bert_tokenize(void * h, const char * text, int32_t * out, int n_max) {
    int32_t n = 0; bert_tokenize(*(Oracle *) h, text, out, &n, n_max); return n;
}
// one causal-model evaluation; returns the new n_past or -1
int orc_gpt_eval(void * h, int which, const int32_t * tokens, int n_tokens, int n_past, int merge_ctx, float * logits, int nth) {
    auto * o = (Oracle *) h; Gpt & m = which == 0 ? o->sem : o->coarse;
    int np = n_past;
    if (!gpt_eval(*o, m, tokens, n_tokens, &np, merge_ctx != 0, logits, nth)) return -1;
    return np;
}
int orc_fine_eval(void * h, const int32_t * tokens8x1024, int nn, float * logits, int nth) {
    return fine_eval(*(Oracle *) h, tokens8x1024, nn, logits, nth) ? 0 : -1;
}

struct orc_params {   // flat mirror of Params for ctypes
    float temp, fine_temp, min_eos_p; int32_t sliding_window_size, max_coarse_history, n_steps_text_encoder;
};
static Params make_params(const orc_params * q) {
    Params p; if (q) { p.temp = q->temp; p.fine_temp = q->fine_temp; p.min_eos_p = q->min_eos_p;
        p.sliding_window_size = q->sliding_window_size; p.max_coarse_history = q->max_coarse_history; p.n_steps_text_encoder = q->n_steps_text_encoder; }
    return p;
}
static void reset_stats(Oracle * o) { for (Gpt * g : {&o->sem, &o->coarse, &o->fine}) { g->t_sample_us = g->t_predict_us = g->t_main_us = 0; g->n_sample = 0; } }

int orc_semantic(void * h, const orc_params * q, const int32_t * prompt513, int32_t * out, float * eos_trace, int nth) {
    auto * o = (Oracle *) h; Params p = make_params(q);
    std::vector<int32_t> prompt(prompt513, prompt513 + 513), res; std::vector<float> tr;
    if (!semantic_stage(*o, p, prompt, res, eos_trace ? &tr : nullptr, nth)) return -1;
    memcpy(out, res.data(), res.size() * 4);
    if (eos_trace) memcpy(eos_trace, tr.data(), tr.size() * 4);
    return (int) res.size();
}
int orc_coarse(void * h, const orc_params * q, const int32_t * semantic, int n_sem, int32_t * out_Tx2, int nth) {
    auto * o = (Oracle *) h; Params p = make_params(q);
    std::vector<int32_t> sem(semantic, semantic + n_sem), res;
    if (!coarse_stage(*o, p, sem, res, nth)) return -1;
    memcpy(out_Tx2, res.data(), res.size() * 4);
    return (int) res.size() / 2;
}
int orc_fine(void * h, const orc_params * q, const int32_t * coarse_Tx2, int T, int32_t * out_Tx8, int nth) {
    auto * o = (Oracle *) h; Params p = make_params(q);
    std::vector<int32_t> c(coarse_Tx2, coarse_Tx2 + (size_t) T * 2), res;
    if (!fine_stage(*o, p, c, res, nth)) return -1;
    memcpy(out_Tx8, res.data(), res.size() * 4);
    return (int) res.size() / 8;
}
// activation after codec stage `stage` (see codec_decode); returns the element count or -1
int orc_codec_tap(void * h, const int32_t * codes, int n_q, int T, int stage, float * out, int capacity, int nth) {
    std::vector<float> pcm, tap;
    if (!codec_decode(*(Oracle *) h, codes, n_q, T, pcm, nth, stage, &tap)) return -1;
    if ((int) tap.size() > capacity) return -1;
    memcpy(out, tap.data(), tap.size() * 4);
    return (int) tap.size();
}
// codes: [n_q][T]; pcm must hold 320*T floats for the 24 kHz ratios; returns sample count
int orc_codec_decode(void * h, const int32_t * codes, int n_q, int T, float * pcm, int nth) {
    std::vector<float> out;
    if (!codec_decode(*(Oracle *) h, codes, n_q, T, out, nth)) return -1;
    memcpy(pcm, out.data(), out.size() * 4);
    return (int) out.size();
}

struct orc_result {
    int32_t n_semantic, n_frames, n_samples;
    int64_t t_eval_us, t_semantic_us, t_coarse_us, t_fine_us, t_codec_us;
    int64_t t_predict_semantic_us, t_predict_coarse_us, t_predict_fine_us;
    int64_t n_sample_semantic, n_sample_coarse, n_sample_fine;
};
// bark_generate_audio (bark.cpp:2125-2172).  Buffers: semantic[768], coarse[1024*2], fine[1024*8], pcm[1024*320]
int orc_generate(void * h, const orc_params * q, const char * text, int32_t * semantic, int32_t * coarse, int32_t * fine,
                 float * pcm, orc_result * res, int nth) {
    auto * o = (Oracle *) h; Params p = make_params(q);
    reset_stats(o);
    const int64_t t0 = now_us();
    std::vector<int32_t> prompt, sem, co, fi;
    tokenize_input(*o, p, text, prompt);
    if (!semantic_stage(*o, p, prompt, sem, nullptr, nth)) return -1;
    if (sem.empty()) return -2;
    if (!coarse_stage(*o, p, sem, co, nth)) return -3;
    if (!fine_stage(*o, p, co, fi, nth)) return -4;
    const int T = (int) fi.size() / 8;
    std::vector<int32_t> codes((size_t) 8 * T);
    for (int c = 0; c < 8; c++) for (int t = 0; t < T; t++) codes[(size_t) c * T + t] = fi[(size_t) t * 8 + c];   // bark.cpp:2153-2159
    const int64_t tc = now_us();
    std::vector<float> audio;
    if (!codec_decode(*o, codes.data(), 8, T, audio, nth)) return -5;
    const int64_t t1 = now_us();
    if (semantic) memcpy(semantic, sem.data(), sem.size() * 4);
    if (coarse) memcpy(coarse, co.data(), co.size() * 4);
    if (fine) memcpy(fine, fi.data(), fi.size() * 4);
    if (pcm) memcpy(pcm, audio.data(), audio.size() * 4);
    if (res) {
        res->n_semantic = (int) sem.size(); res->n_frames = T; res->n_samples = (int) audio.size();
        res->t_eval_us = t1 - t0; res->t_semantic_us = o->sem.t_main_us; res->t_coarse_us = o->coarse.t_main_us; res->t_fine_us = o->fine.t_main_us;
        res->t_codec_us = t1 - tc;
        res->t_predict_semantic_us = o->sem.t_predict_us; res->t_predict_coarse_us = o->coarse.t_predict_us; res->t_predict_fine_us = o->fine.t_predict_us;
        res->n_sample_semantic = o->sem.n_sample; res->n_sample_coarse = o->coarse.n_sample; res->n_sample_fine = o->fine.n_sample;
    }
    return 0;
}

}  // extern "C"
