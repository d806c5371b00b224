// engine.hip - model upload, stage loops and orchestration of the MI355X Bark engine.
// Control flow restates /root/reference/bark.cpp (cited per function); all tensor math runs in the
// HIP kernels of kernels.hip / codec_kernels.hip.  No CPU fallback exists: without a HIP device
// bark_load_model fails.
#include "engine.h"

#include <algorithm>
#include <chrono>
#include <climits>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <stdexcept>

using namespace barkhip;

namespace {

#define HIP_OK(expr)                                                                                         \
    do {                                                                                                     \
        hipError_t _e = (expr);                                                                              \
        if (_e != hipSuccess)                                                                                \
            throw std::runtime_error(std::string("HIP error: ") + hipGetErrorString(_e) + " at " #expr);   \
    } while (0)

inline int64_t now_us() {
    return std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

template <typename T> T * dev_alloc(bark_context * ctx, size_t count) {
    void * p = nullptr;
    HIP_OK(hipMalloc(&p, std::max<size_t>(count, 1) * sizeof(T)));
    ctx->allocs.push_back(p);
    return (T *) p;
}

// ---- weight slab ---------------------------------------------------------------------------------
struct SlabPlan {
    struct Item { const uint8_t * src; size_t bytes; size_t off; };
    std::vector<Item> items;
    size_t total = 0;
    size_t add(const TensorRef & t) {
        const size_t off = total;
        items.push_back({t.data, t.nbytes(), off});
        total = (total + t.nbytes() + 255) & ~(size_t) 255;
        return off;
    }
};

const TensorRef & need(const std::map<std::string, TensorRef> & m, const std::string & name, int ttype, int64_t ne0, int64_t ne1) {
    auto it = m.find(name);
    if (it == m.end()) throw std::runtime_error("missing tensor '" + name + "'");
    const TensorRef & t = it->second;
    if (ne0 > 0 && (t.ne[0] != ne0 || (ne1 > 0 && t.ne[1] != ne1)))       // shape check on ne[0], ne[1] (bark.cpp:1034)
        throw std::runtime_error("tensor '" + name + "' has an unexpected shape");
    if (t.ttype != ttype)
        throw std::runtime_error("tensor '" + name + "' is " + (quant_format_by_type(t.ttype) ? quant_format_by_type(t.ttype)->name : t.ttype ? "f16" : "f32") +
                                 ", expected " + (ttype == 1 ? "f16" : ttype == 0 ? "f32" : "another type"));
    return t;
}
// a weight matrix: f16, or q4_0 (uploaded later as a QMat)
const TensorRef & need_w(const std::map<std::string, TensorRef> & m, const std::string & name, int64_t ne0, int64_t ne1) {
    auto it = m.find(name);
    if (it == m.end()) throw std::runtime_error("missing tensor '" + name + "'");
    return need(m, name, it->second.ttype, ne0, ne1);                      // f32, f16 or a block format: all accepted
}
QMat q4_rows(const QMat & w, size_t row0, int K) {
    QMat r = w;
    if (w.qt == QT_F32) { r.qs = w.qs + row0 * (size_t) K * 4; return r; }
    const size_t nb = row0 * (size_t) (K / 32);
    r.d = w.d + nb; r.qs = w.qs + nb * (size_t) quant_formats()[w.qt].qs_bytes;
    if (w.m) r.m = w.m + nb;
    if (w.qh) r.qh = w.qh + nb;
    return r;
}
const TensorRef * maybe(const std::map<std::string, TensorRef> & m, const std::string & name, int ttype, int64_t ne0) {
    auto it = m.find(name);
    if (it == m.end()) return nullptr;
    if (it->second.ne[0] != ne0 || it->second.ttype != ttype) throw std::runtime_error("tensor '" + name + "' has an unexpected shape/type");
    return &it->second;
}

// ggml's GELU table: tanh approximation tabulated over every f16 input.
// Written without fused multiply-adds (the file is built with -ffp-contract=off) so the table is the
// same on every host compiler.
float gelu_tanh_host(float x) {
    const float a = 0.044715f, c = 0.79788456080286535587989211986876f;
    const float x2 = x * x;
    const float inner = 1.0f + a * x2;
    const float arg = c * x * inner;
    const float t = tanhf(arg);
    return 0.5f * x * (1.0f + t);
}

}  // namespace

bark_context::~bark_context() {
    (void) hipSetDevice(device);
    for (auto & g : gpt) {
        if (g.decode_graph) (void) hipGraphExecDestroy(g.decode_graph);
        if (g.bench_graph) (void) hipGraphExecDestroy(g.bench_graph);
    }
    for (auto & g : batch.graph) if (g) (void) hipGraphExecDestroy(g);
    for (auto & g : lstm_graphs) if (g.exec) (void) hipGraphExecDestroy(g.exec);
    for (auto & g : fine_graphs) if (g) (void) hipGraphExecDestroy(g);
    for (void * p : allocs) (void) hipFree(p);
    if (stream) (void) hipStreamDestroy(stream);
}
bark_context::SharedWeights::~SharedWeights() {
    (void) hipSetDevice(device);
    if (slab) (void) hipFree(slab);
    if (codebooks) (void) hipFree(codebooks);
    for (void * p : extra) (void) hipFree(p);
}

namespace barkhip {

void engine_invalidate_graphs(bark_context * ctx) {
    for (auto & g : ctx->batch.graph) if (g) { (void) hipGraphExecDestroy(g); g = nullptr; }
    for (auto & g : ctx->fine_graphs) if (g) { (void) hipGraphExecDestroy(g); g = nullptr; }
    for (auto & g : ctx->gpt) {
        if (g.decode_graph) { (void) hipGraphExecDestroy(g.decode_graph); g.decode_graph = nullptr; }
        if (g.bench_graph) { (void) hipGraphExecDestroy(g.bench_graph); g.bench_graph = nullptr; }
    }
}

// per-context mutable state: stream, KV caches, activation scratch, GELU table
static void init_runtime(bark_context * ctxp) {
    struct Holder { bark_context * p; bark_context * get() const { return p; } bark_context * operator->() const { return p; } } ctx{ctxp};
    // ---- KV caches, scratch ------------------------------------------------------------------------
    const int P = ctx->P;
    for (int g = 0; g < 2; g++) {
        GptModel & m = ctx->gpt[g];
        m.kv_layer_stride = (size_t) m.hp.n_embd * P;
        m.kcache = dev_alloc<float>(ctx.get(), m.kv_layer_stride * m.hp.n_layer);     // bark.cpp:976-991
        m.vcache = dev_alloc<float>(ctx.get(), m.kv_layer_stride * m.hp.n_layer);
    }
    {
        GptModel & m = ctx->gpt[2];
        m.kv_layer_stride = 0;
        m.kcache = dev_alloc<float>(ctx.get(), (size_t) m.hp.n_embd * P);
        m.vcache = dev_alloc<float>(ctx.get(), (size_t) m.hp.n_embd * P);
    }
    const size_t NE = (size_t) P * ctx->max_E;
    ctx->x = dev_alloc<float>(ctx.get(), NE);
    ctx->q = dev_alloc<float>(ctx.get(), NE);
    ctx->xn = dev_alloc<half_t>(ctx.get(), NE);
    ctx->att = dev_alloc<half_t>(ctx.get(), NE);
    ctx->hbuf = dev_alloc<half_t>(ctx.get(), NE * 4);
    ctx->scores = dev_alloc<float>(ctx.get(), (size_t) ctx->max_H * P * P);
    if (ctx->any_q4) {
        ctx->att32 = dev_alloc<float>(ctx.get(), NE);
        ctx->h32 = dev_alloc<float>(ctx.get(), NE * 4);
        const size_t nT = (size_t) (4 * ctx->max_E / 32) * 1024;
        ctx->xq.q = dev_alloc<int8_t>(ctx.get(), NE * 4);
        ctx->xq.d = dev_alloc<float>(ctx.get(), NE * 4 / 32);
        ctx->xq.s = dev_alloc<float>(ctx.get(), NE * 4 / 32);
        ctx->xq.dT = dev_alloc<float>(ctx.get(), nT);
        ctx->xq.sT = dev_alloc<float>(ctx.get(), nT);
        HIP_OK(hipMemset(ctx->xq.dT, 0, nT * sizeof(float)));
        HIP_OK(hipMemset(ctx->xq.sT, 0, nT * sizeof(float)));
        if (ctx->any_w32) ctx->xn32 = dev_alloc<float>(ctx.get(), NE);
    }
    size_t n_logits = (size_t) 1024 * ctx->gpt[2].hp.n_out_vocab;
    for (int g = 0; g < 2; g++) n_logits = std::max(n_logits, (size_t) ctx->gpt[g].hp.n_out_vocab);
    ctx->logits = dev_alloc<float>(ctx.get(), n_logits);
    ctx->d_tokens = dev_alloc<int32_t>(ctx.get(), 8 * 1024);
    ctx->d_out_tokens = dev_alloc<int32_t>(ctx.get(), 2048);
    ctx->d_eos_trace = dev_alloc<float>(ctx.get(), 2048);
    ctx->d_state = dev_alloc<StepState>(ctx.get(), 1);
    ctx->d_lstm_t = dev_alloc<int>(ctx.get(), 2);
    ctx->d_u = dev_alloc<double>(ctx.get(), 8192);
    { const char * e = getenv("BARK_HIP_HOST_SAMPLING"); ctx->host_sampling = e && atoi(e) != 0; }
    ctx->d_hmax = dev_alloc<unsigned>(ctx.get(), 64);
    HIP_OK(hipMemset(ctx->d_hmax, 0, 64 * sizeof(unsigned)));
    {
        std::vector<uint16_t> lut(65536);
        for (uint32_t i = 0; i < 65536; i++) {
            const uint16_t bits = (uint16_t) i;
            const _Float16 h = __builtin_bit_cast(_Float16, bits);
            const _Float16 r = (_Float16) gelu_tanh_host((float) h);
            lut[i] = __builtin_bit_cast(uint16_t, r);
        }
        ctx->d_gelu_lut = dev_alloc<uint16_t>(ctx.get(), 65536);
        HIP_OK(hipMemcpy(ctx->d_gelu_lut, lut.data(), 65536 * 2, hipMemcpyHostToDevice));
    }
}

// bark_load_model_from_file (bark.cpp:1080-1163): parse the container, upload every tensor of the hot path.
bark_context * engine_load(const char * path, const bark_context_params & params, uint32_t seed) {
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0)
        throw std::runtime_error("no HIP device available (this engine has no CPU path)");
    std::unique_ptr<bark_context> ctx(new bark_context());
    ctx->params = params;
    ctx->rng = std::mt19937(seed);                       // bark.cpp:1179
    if (const char * e = getenv("BARK_HIP_DEVICE")) ctx->device = atoi(e);
    else (void) hipGetDevice(&ctx->device);
    if (ctx->device < 0 || ctx->device >= n_dev) throw std::runtime_error("BARK_HIP_DEVICE out of range");
    HIP_OK(hipSetDevice(ctx->device));
    if (const char * e = getenv("BARK_HIP_GRAPH")) ctx->use_graph = atoi(e) != 0;
    HIP_OK(hipStreamCreate(&ctx->stream));
    init_kernel_attributes();

    ModelFile mf;
    std::string err;
    if (!mf.open(path, err)) throw std::runtime_error(std::string("failed to read '") + path + "': " + err);
    ctx->vocab.build(mf.vocab);

    // ---- plan the slab -----------------------------------------------------------------------------
    SlabPlan plan;
    struct Fix { const void ** dst; size_t off; };
    std::vector<Fix> fixes;
    auto place = [&](const TensorRef & t, const void ** dst) { fixes.push_back({dst, plan.add(t)}); };
    struct Q4Job { const TensorRef * t; QMat * dst; };
    std::vector<Q4Job> q4_jobs;
    int n_w16 = 0, n_wq4 = 0, n_w32 = 0;
    GptModel * cur_model = nullptr;
    auto place_w = [&](const TensorRef & t, const half_t ** dst16, QMat * dstq) {
        if (quant_format_by_type(t.ttype)) { q4_jobs.push_back({&t, dstq}); n_wq4++; }
        else if (t.ttype == 0) { place(t, (const void **) &dstq->qs); dstq->qt = QT_F32; n_w32++; }       // f32 file: plain f32 rows behind the QMat handle
        else { place(t, (const void **) dst16); n_w16++; }
        (void) cur_model;
    };

    for (int g = 0; g < 3; g++) {
        GptModel & m = ctx->gpt[g];
        m.hp = mf.gpt[g].hp;
        const auto & T = mf.gpt[g].tensors;
        const int E = m.hp.n_embd;
        if (E / m.hp.n_head != 64) throw std::runtime_error("head_dim must be 64");
        if (E % 128 != 0 || E > 1024) throw std::runtime_error("n_embd must be a multiple of 128 and <= 1024");
        if (m.hp.block_size != 1024) throw std::runtime_error("block_size must be 1024");
        if (m.hp.n_wtes > 8 || m.hp.n_lm_heads > 8 || m.hp.n_layer > 64) throw std::runtime_error("unsupported GPT shape");
        m.layers.resize((size_t) m.hp.n_layer);
        n_w16 = n_wq4 = n_w32 = 0;
        for (int i = 0; i < m.hp.n_wtes; i++) place_w(need_w(T, "model/wte/" + std::to_string(i), E, m.hp.n_in_vocab), &m.wte[i], &m.wte_q[i]);
        for (int i = 0; i < m.hp.n_lm_heads; i++) place_w(need_w(T, "model/lm_head/" + std::to_string(i), E, m.hp.n_out_vocab), &m.lm_head[i], &m.lm_head_q[i]);
        place(need(T, "model/wpe", 0, E, m.hp.block_size), (const void **) &m.wpe);
        place(need(T, "model/ln_f/g", 0, E, 0), (const void **) &m.lnf_g);
        if (auto * t = maybe(T, "model/ln_f/b", 0, E)) place(*t, (const void **) &m.lnf_b);
        for (int l = 0; l < m.hp.n_layer; l++) {
            const std::string p = "model/h" + std::to_string(l);
            GptModel::Layer & L = m.layers[(size_t) l];
            place(need(T, p + "/ln_1/g", 0, E, 0), (const void **) &L.ln1_g);
            place(need(T, p + "/ln_2/g", 0, E, 0), (const void **) &L.ln2_g);
            if (auto * t = maybe(T, p + "/ln_1/b", 0, E)) place(*t, (const void **) &L.ln1_b);
            if (auto * t = maybe(T, p + "/ln_2/b", 0, E)) place(*t, (const void **) &L.ln2_b);
            place_w(need_w(T, p + "/attn/c_attn/w", E, 3 * E), &L.attn_w, &L.attn_q);
            place_w(need_w(T, p + "/attn/c_proj/w", E, E), &L.proj_w, &L.proj_q);
            place_w(need_w(T, p + "/mlp/c_fc/w", E, 4 * E), &L.fc_w, &L.fc_q);
            place_w(need_w(T, p + "/mlp/c_proj/w", 4 * E, E), &L.mproj_w, &L.mproj_q);
            if (auto * t = maybe(T, p + "/attn/c_attn/b", 0, 3 * E)) place(*t, (const void **) &L.attn_b);
            if (auto * t = maybe(T, p + "/attn/c_proj/b", 0, E)) place(*t, (const void **) &L.proj_b);
            if (auto * t = maybe(T, p + "/mlp/c_fc/b", 0, 4 * E)) place(*t, (const void **) &L.fc_b);
            if (auto * t = maybe(T, p + "/mlp/c_proj/b", 0, E)) place(*t, (const void **) &L.mproj_b);
        }
        // bark_model_quantize converts every matrix of a model or none (bark.cpp:2277-2289)
        if ((n_w16 > 0) + (n_wq4 > 0) + (n_w32 > 0) > 1) throw std::runtime_error("model mixes f32 / f16 / quantised weight matrices");
        m.w32 = n_w32 > 0;
        m.q4 = n_wq4 > 0 || m.w32;                        // both keep the activations in f32 between the products
        ctx->any_q4 = ctx->any_q4 || m.q4;
        ctx->any_w32 = ctx->any_w32 || m.w32;
        ctx->max_E = std::max(ctx->max_E, E);
        ctx->max_H = std::max(ctx->max_H, m.hp.n_head);
    }
    if (ctx->gpt[2].hp.n_wtes != 8 || ctx->gpt[2].hp.n_lm_heads < 6) throw std::runtime_error("fine model must have 8 embeddings and >= 6 heads");

    // ---- codec -------------------------------------------------------------------------------------
    CodecModel & cm = ctx->codec;
    cm.hp = mf.codec_hp;
    // f32 codec weights (convert.py without --use-f16) are rounded to f16 here and then run in the f16-weight arithmetic, like
    // the oracle: conv kernels meet an f16 im2col in ggml's mul_mat anyway; for the LSTM matrices it is a stated simplification
    std::map<std::string, TensorRef> codec_w16;
    std::deque<std::vector<uint16_t>> codec_w16_store;
    auto codec_weight = [&](const std::string & name, int64_t ne0, int64_t ne1) -> const TensorRef & {
        auto it = mf.codec.find(name);
        if (it == mf.codec.end()) throw std::runtime_error("missing tensor '" + name + "'");
        if (it->second.ttype != 0) return need(mf.codec, name, 1, ne0, ne1);
        auto have = codec_w16.find(name);
        if (have != codec_w16.end()) return have->second;
        const TensorRef & t = need(mf.codec, name, 0, ne0, ne1);
        codec_w16_store.emplace_back((size_t) t.nelements());
        std::vector<uint16_t> & h = codec_w16_store.back();
        for (size_t i = 0; i < h.size(); i++) { float f; memcpy(&f, t.data + 4 * i, 4); h[i] = __builtin_bit_cast(uint16_t, (_Float16) f); }
        TensorRef r = t; r.ttype = 1; r.data = (const uint8_t *) h.data();
        return codec_w16[name] = r;
    };
    {
        const auto & T = mf.codec;
        auto conv = [&](const std::string & p, CodecModel::Conv & cv) {
            const TensorRef & w = codec_weight(p + ".weight", 0, 0);
            cv.k = (int) w.ne[0]; cv.cin = (int) w.ne[1]; cv.cout = (int) w.ne[2];
            place(w, (const void **) &cv.w);
            const TensorRef & b = need(T, p + ".bias", 0, 0, 0);
            if (b.nelements() != cv.cout) throw std::runtime_error("codec bias size mismatch at " + p);
            place(b, (const void **) &cv.b);
        };
        auto convt = [&](const std::string & p, CodecModel::ConvT & cv, int stride) {
            const TensorRef & w = codec_weight(p + ".weight", 0, 0);
            cv.k = (int) w.ne[0]; cv.cout = (int) w.ne[1]; cv.cin = (int) w.ne[2]; cv.stride = stride;
            place(w, (const void **) &cv.w);
            const TensorRef & b = need(T, p + ".bias", 0, 0, 0);
            if (b.nelements() != cv.cout) throw std::runtime_error("codec bias size mismatch at " + p);
            place(b, (const void **) &cv.b);
        };
        conv("decoder.model.0.conv.conv", cm.init);
        cm.D = cm.init.cout;
        if (cm.D % 128 != 0) throw std::runtime_error("codec LSTM width must be a multiple of 128");
        for (int l = 0; l < 2; l++) {
            const std::string s = std::to_string(l);
            place(codec_weight("decoder.model.1.lstm.weight_ih_l" + s, cm.D, 4 * cm.D), (const void **) &cm.lstm[l].w_ih);
            place(codec_weight("decoder.model.1.lstm.weight_hh_l" + s, cm.D, 4 * cm.D), (const void **) &cm.lstm[l].w_hh);
            place(need(T, "decoder.model.1.lstm.bias_ih_l" + s, 0, 4 * cm.D, 0), (const void **) &cm.lstm[l].b_ih);
            place(need(T, "decoder.model.1.lstm.bias_hh_l" + s, 0, 4 * cm.D, 0), (const void **) &cm.lstm[l].b_hh);
        }
        static const int ratios[4] = {8, 5, 4, 2};          // EnCodec 24 kHz upsampling ratios (modeling_encodec.py:329-340)
        for (int i = 0; i < 4; i++) {
            const int idx = 3 + 3 * i;
            convt("decoder.model." + std::to_string(idx) + ".convtr.convtr", cm.blocks[i].up, ratios[i]);
            conv("decoder.model." + std::to_string(idx + 1) + ".block.1.conv.conv", cm.blocks[i].c1);
            conv("decoder.model." + std::to_string(idx + 1) + ".block.3.conv.conv", cm.blocks[i].c2);
            conv("decoder.model." + std::to_string(idx + 1) + ".shortcut.conv.conv", cm.blocks[i].sc);
        }
        conv("decoder.model.15.conv.conv", cm.fin);
        // codebooks are uploaded contiguously (separate allocation below)
        while (T.count("quantizer.vq.layers." + std::to_string(cm.n_q) + "._codebook.embed")) cm.n_q++;
        if (cm.n_q == 0) throw std::runtime_error("codec has no codebooks");
    }

    // ---- upload ------------------------------------------------------------------------------------
    ctx->weight_bytes = plan.total;
    ctx->weights = std::make_shared<bark_context::SharedWeights>();
    ctx->weights->device = ctx->device;
    HIP_OK(hipMalloc(&ctx->weights->slab, plan.total));
    {
        // stage through pinned memory in 32 MiB pieces (the mapping is pageable and possibly unaligned)
        const size_t kStage = 32u << 20;
        void * stage = nullptr;
        HIP_OK(hipHostMalloc(&stage, kStage, hipHostMallocDefault));
        for (const auto & it : plan.items) {
            for (size_t done = 0; done < it.bytes; done += kStage) {
                const size_t n = std::min(kStage, it.bytes - done);
                memcpy(stage, it.src + done, n);
                HIP_OK(hipMemcpy((uint8_t *) ctx->weights->slab + it.off + done, stage, n, hipMemcpyHostToDevice));
            }
        }
        (void) hipHostFree(stage);
    }
    for (const auto & f : fixes) *f.dst = (const uint8_t *) ctx->weights->slab + f.off;
    for (const auto & j : q4_jobs) {
        // ggml blocks (f16 d [| f16 m] [| u32 qh] | level bytes) -> one device array per field: aligned vector loads of the levels
        const QuantFormat & qf = *quant_format_by_type(j.t->ttype);
        const size_t nb = (size_t) j.t->nelements() / 32;
        std::vector<uint16_t> d(nb), mn(qf.has_min ? nb : 0);
        std::vector<uint32_t> qh(qf.has_high_bits ? nb : 0);
        std::vector<uint8_t> qs(nb * (size_t) qf.qs_bytes);
        for (size_t b = 0; b < nb; b++) {
            const uint8_t * blk = j.t->data + b * (size_t) qf.block_bytes;
            size_t pos = 0;
            memcpy(&d[b], blk, 2); pos = 2;
            if (qf.has_min) { memcpy(&mn[b], blk + pos, 2); pos += 2; }
            if (qf.has_high_bits) { memcpy(&qh[b], blk + pos, 4); pos += 4; }
            memcpy(&qs[b * (size_t) qf.qs_bytes], blk + pos, (size_t) qf.qs_bytes);
        }
        auto upload = [&](const void * src, size_t bytes) -> void * {
            void * p = nullptr;
            HIP_OK(hipMalloc(&p, bytes)); ctx->weights->extra.push_back(p);
            HIP_OK(hipMemcpy(p, src, bytes, hipMemcpyHostToDevice));
            return p;
        };
        j.dst->qt = qf.id;
        j.dst->d = (const half_t *) upload(d.data(), nb * 2);
        j.dst->qs = (const uint8_t *) upload(qs.data(), qs.size());
        if (qf.has_min) j.dst->m = (const half_t *) upload(mn.data(), nb * 2);
        if (qf.has_high_bits) j.dst->qh = (const uint32_t *) upload(qh.data(), nb * 4);
        ctx->weight_bytes += nb * (size_t) qf.block_bytes;
    }
    {
        // f32 copies of the codec's conv weights (19 MB of f16 in the file): exact, and wave-uniform f32 weights become
        // scalar loads / SGPR operands in the register-blocked conv kernels
        auto widen = [&](const TensorRef & t) -> const float * {
            std::vector<float> f((size_t) t.nelements());
            for (size_t i = 0; i < f.size(); i++) { uint16_t b; memcpy(&b, t.data + 2 * i, 2); f[i] = (float) __builtin_bit_cast(_Float16, b); }
            float * d = nullptr;
            HIP_OK(hipMalloc((void **) &d, f.size() * sizeof(float)));
            ctx->weights->extra.push_back(d);
            HIP_OK(hipMemcpy(d, f.data(), f.size() * sizeof(float), hipMemcpyHostToDevice));
            return d;
        };
        auto cw = [&](const std::string & p) { return widen(codec_weight(p + ".weight", 0, 0)); };
        cm.init.w32 = cw("decoder.model.0.conv.conv");
        for (int i = 0; i < 4; i++) {
            const int idx = 3 + 3 * i;
            cm.blocks[i].up.w32 = cw("decoder.model." + std::to_string(idx) + ".convtr.convtr");
            cm.blocks[i].c1.w32 = cw("decoder.model." + std::to_string(idx + 1) + ".block.1.conv.conv");
            cm.blocks[i].c2.w32 = cw("decoder.model." + std::to_string(idx + 1) + ".block.3.conv.conv");
            cm.blocks[i].sc.w32 = cw("decoder.model." + std::to_string(idx + 1) + ".shortcut.conv.conv");
        }
        cm.fin.w32 = cw("decoder.model.15.conv.conv");
    }
    {
        const size_t per = (size_t) cm.hp.n_bins * cm.hp.hidden_dim;
        float * cb = nullptr;
        HIP_OK(hipMalloc((void **) &cb, per * cm.n_q * sizeof(float)));
        ctx->weights->codebooks = cb;
        for (int q = 0; q < cm.n_q; q++) {
            const TensorRef & t = need(mf.codec, "quantizer.vq.layers." + std::to_string(q) + "._codebook.embed", 0, cm.hp.hidden_dim, cm.hp.n_bins);
            HIP_OK(hipMemcpy(cb + per * q, t.data, per * 4, hipMemcpyHostToDevice));
        }
        cm.codebooks = cb;
    }

    init_runtime(ctx.get());
    hipDeviceProp_t prop;
    HIP_OK(hipGetDeviceProperties(&prop, ctx->device));
    char buf[512];
    const GptModel & g0 = ctx->gpt[0];
    const char * wfmt = g0.w32 ? "f32" : g0.q4 ? quant_formats()[g0.layers[0].attn_q.qt].name : "f16";
    snprintf(buf, sizeof(buf), "bark-mi355x engine on %s (%s, %d CUs), %s weights %.1f MB, n_embd %d/%d/%d, layers %d/%d/%d, graph=%d",
             prop.name, prop.gcnArchName, prop.multiProcessorCount, wfmt, ctx->weight_bytes / 1e6, ctx->gpt[0].hp.n_embd, ctx->gpt[1].hp.n_embd,
             ctx->gpt[2].hp.n_embd, ctx->gpt[0].hp.n_layer, ctx->gpt[1].hp.n_layer, ctx->gpt[2].hp.n_layer, (int) ctx->use_graph);
    ctx->description = buf;
    if (params.verbosity >= MEDIUM) fprintf(stderr, "%s\n", buf);
    return ctx.release();
}

bark_context * engine_clone(bark_context * src, uint32_t seed) {
    HIP_OK(hipSetDevice(src->device));
    std::unique_ptr<bark_context> ctx(new bark_context());
    ctx->params = src->params;
    ctx->rng = std::mt19937(seed);
    ctx->vocab = src->vocab;
    for (int g = 0; g < 3; g++) {
        ctx->gpt[g] = src->gpt[g];
        ctx->gpt[g].kcache = ctx->gpt[g].vcache = nullptr;
        ctx->gpt[g].decode_graph = ctx->gpt[g].bench_graph = nullptr;
    }
    ctx->codec = src->codec;
    ctx->device = src->device; ctx->use_graph = src->use_graph;
    ctx->weights = src->weights; ctx->weight_bytes = src->weight_bytes;
    ctx->max_E = src->max_E; ctx->max_H = src->max_H; ctx->P = src->P; ctx->any_q4 = src->any_q4; ctx->any_w32 = src->any_w32;
    HIP_OK(hipStreamCreate(&ctx->stream));
    init_runtime(ctx.get());
    ctx->description = src->description + " (clone)";
    return ctx.release();
}

// ---------------------------------------------------------------------------------------------------
// GPT building blocks
// ---------------------------------------------------------------------------------------------------
namespace {

// bytes per weight of the model's matrices: f16 2, f32 4, block formats block_bytes / 32
double weight_bytes_per_element(const GptModel & m) {
    if (m.w32) return 4.0;
    if (m.q4) return quant_formats()[m.layers[0].attn_q.qt].block_bytes / 32.0;
    return 2.0;
}
float * layer_k(const GptModel & m, int l) { return m.kcache + m.kv_layer_stride * (size_t) l; }
float * layer_v(const GptModel & m, int l) { return m.vcache + m.kv_layer_stride * (size_t) l; }

// N > 1 rows through all layers (bark.cpp:1261-1389 causal, :1474-1562 fine); x holds the embeddings.
void run_layers_rows(bark_context * c, GptModel & m, int N, bool causal, float * kbase = nullptr, float * vbase = nullptr, int pos0 = 0) {
    const int E = m.hp.n_embd, H = m.hp.n_head, P = c->P;
    hipStream_t s = c->stream;
    // kbase / vbase: another utterance slot's cache (batched decode); default: the context's own cache
    auto layer_k = [&](const GptModel & mm, int l) { return (kbase ? kbase : mm.kcache) + mm.kv_layer_stride * (size_t) l; };
    auto layer_v = [&](const GptModel & mm, int l) { return (vbase ? vbase : mm.vcache) + mm.kv_layer_stride * (size_t) l; };
    for (int l = 0; l < m.hp.n_layer; l++) {
        const GptModel::Layer & L = m.layers[(size_t) l];
        // f16 weights: activations are rounded to f16 rows (xn / att / hbuf); q4_0 weights: f32 rows quantised to q8_0 (xq8 / xd8)
        if (m.w32)     launch_ln_rows_f32(s, c->x, N, E, L.ln1_g, L.ln1_b, c->xn32);
        else if (m.q4) launch_q8_rows(s, c->x, N, E, L.ln1_g, L.ln1_b, c->xq);
        else           launch_ln_rows(s, c->x, N, E, L.ln1_g, L.ln1_b, c->xn);
        LinArgs a;
        a.W = L.attn_w; a.wq = L.attn_q; a.M = 3 * E; a.K = E; a.N = N; a.x_f16 = c->xn; a.xq = c->xq; if (m.w32) a.x_f32 = c->xn32; a.bias = L.attn_b; a.epi = EPI_QKV;
        a.q = c->q; a.kc = layer_k(m, l); a.vc = layer_v(m, l); a.E = E; a.P = P; a.pos0 = pos0;
        launch_linear(s, a);
        AttnPrefillArgs at;
        at.q = c->q; at.ldq = E; at.kc = layer_k(m, l); at.vc = layer_v(m, l); at.H = H; at.P = P; at.N = N; at.n_past = pos0;
        at.causal = causal ? 1 : 0; at.scores = c->scores; at.att = c->att; at.ld_att = E; at.att32 = m.q4 ? c->att32 : nullptr;
        { static const int dbg = getenv("BARK_HIP_ATTN_DBG") ? atoi(getenv("BARK_HIP_ATTN_DBG")) : 0; at.dbg = dbg; }
        launch_attn_prefill(s, at);
        if (m.q4 && !m.w32) launch_q8_rows(s, c->att32, N, E, nullptr, nullptr, c->xq);
        LinArgs p;
        p.W = L.proj_w; p.wq = L.proj_q; p.M = E; p.K = E; p.N = N; p.x_f16 = c->att; p.xq = c->xq; if (m.w32) p.x_f32 = c->att32; p.bias = L.proj_b; p.epi = EPI_RESID; p.res = c->x;
        launch_linear(s, p);
        if (m.w32)     launch_ln_rows_f32(s, c->x, N, E, L.ln2_g, L.ln2_b, c->xn32);
        else if (m.q4) launch_q8_rows(s, c->x, N, E, L.ln2_g, L.ln2_b, c->xq);
        else           launch_ln_rows(s, c->x, N, E, L.ln2_g, L.ln2_b, c->xn);
        LinArgs f;
        f.W = L.fc_w; f.wq = L.fc_q; f.M = 4 * E; f.K = E; f.N = N; f.x_f16 = c->xn; f.xq = c->xq; if (m.w32) f.x_f32 = c->xn32; f.bias = L.fc_b; f.epi = EPI_GELU;
        f.out_h = c->hbuf; f.out_h32 = m.q4 ? c->h32 : nullptr; f.lut = c->d_gelu_lut;
        launch_linear(s, f);
        if (m.q4 && !m.w32) launch_q8_rows(s, c->h32, N, 4 * E, nullptr, nullptr, c->xq);
        LinArgs o;
        o.W = L.mproj_w; o.wq = L.mproj_q; o.M = E; o.K = 4 * E; o.N = N; o.x_f16 = c->hbuf; o.xq = c->xq; if (m.w32) o.x_f32 = c->h32; o.bias = L.mproj_b; o.epi = EPI_RESID; o.res = c->x;
        launch_linear(s, o);
    }
}

// one token through all layers; position / token come from the device-resident StepState
void run_layers_decode(bark_context * c, GptModel & m) {
    const int E = m.hp.n_embd, H = m.hp.n_head, P = c->P;
    hipStream_t s = c->stream;
    for (int l = 0; l < m.hp.n_layer; l++) {
        const GptModel::Layer & L = m.layers[(size_t) l];
        LinArgs a;
        a.W = L.attn_w; a.wq = L.attn_q; a.M = 3 * E; a.K = E; a.N = 1; a.x_f32 = c->x; a.ln_g = L.ln1_g; a.ln_b = L.ln1_b; a.bias = L.attn_b;
        a.epi = EPI_QKV; a.q = c->q; a.kc = layer_k(m, l); a.vc = layer_v(m, l); a.E = E; a.P = P; a.pos0 = 0; a.st = c->d_state;
        launch_linear(s, a);
        AttnDecodeArgs at;
        at.q = c->q; at.kc = layer_k(m, l); at.vc = layer_v(m, l); at.H = H; at.P = P; at.st = c->d_state; at.att = c->att; at.scores = c->scores; at.hmax = c->d_hmax;
        at.att32 = m.q4 ? c->att32 : nullptr;
        launch_attn_decode(s, at);
        LinArgs p;
        p.W = L.proj_w; p.wq = L.proj_q; p.M = E; p.K = E; p.N = 1; if (m.q4) p.x_f32 = c->att32; else p.x_f16 = c->att; p.bias = L.proj_b; p.epi = EPI_RESID; p.res = c->x;
        launch_linear(s, p);
        LinArgs f;
        f.W = L.fc_w; f.wq = L.fc_q; f.M = 4 * E; f.K = E; f.N = 1; f.x_f32 = c->x; f.ln_g = L.ln2_g; f.ln_b = L.ln2_b; f.bias = L.fc_b;
        f.epi = EPI_GELU; f.out_h = c->hbuf; f.out_h32 = m.q4 ? c->h32 : nullptr; f.lut = c->d_gelu_lut;
        launch_linear(s, f);
        LinArgs o;
        o.W = L.mproj_w; o.wq = L.mproj_q; o.M = E; o.K = 4 * E; o.N = 1; if (m.q4) o.x_f32 = c->h32; else o.x_f16 = c->hbuf; o.bias = L.mproj_b; o.epi = EPI_RESID; o.res = c->x;
        launch_linear(s, o);
    }
}

// final LayerNorm + LM head on ONE row (bark.cpp:1391-1405): rows [row0, row0 + n_rows) of the head,
// or the parity-selected codebook window of the coarse model.
void run_lm_head(bark_context * c, GptModel & m, const float * xrow, int row0, int n_rows, int parity_rows) {
    LinArgs a;
    if (m.q4) a.wq = q4_rows(m.lm_head_q[0], (size_t) row0, m.hp.n_embd); else a.W = m.lm_head[0] + (size_t) row0 * m.hp.n_embd;
    a.M = n_rows; a.K = m.hp.n_embd; a.N = 1;
    a.x_f32 = xrow; a.ln_g = m.lnf_g; a.ln_b = m.lnf_b; a.epi = EPI_LOGITS; a.out = c->logits; a.ld_out = n_rows;
    a.parity_rows = parity_rows; a.st = c->d_state;
    launch_linear(c->stream, a);
}

void set_state(bark_context * c, const StepState & st) {
    HIP_OK(hipMemsetAsync(c->d_hmax, 0, 64 * sizeof(unsigned), c->stream));     // decode-attention row maxima (kernels.hip)
    HIP_OK(hipMemcpyAsync(c->d_state, &st, sizeof(st), hipMemcpyHostToDevice, c->stream));
    HIP_OK(hipStreamSynchronize(c->stream));       // `st` is a stack object
}
StepState get_state(bark_context * c) {
    StepState st;
    HIP_OK(hipMemcpyAsync(&st, c->d_state, sizeof(st), hipMemcpyDeviceToHost, c->stream));
    HIP_OK(hipStreamSynchronize(c->stream));
    return st;
}
StepState fresh_state() {
    StepState st{};
    st.eos_step = INT32_MAX;
    return st;
}

void upload_tokens(bark_context * c, const int32_t * tok, size_t n) {
    HIP_OK(hipMemcpyAsync(c->d_tokens, tok, n * 4, hipMemcpyHostToDevice, c->stream));
    HIP_OK(hipStreamSynchronize(c->stream));
}

void check_ids(const int32_t * tok, size_t n, int n_in, const char * what) {
    for (size_t i = 0; i < n; i++)
        if (tok[i] < 0 || tok[i] >= n_in) throw std::runtime_error(std::string(what) + ": token id out of range");
}

// prompt rows -> x, all layers.  merge: the 513-id semantic prompt collapses to 257 rows (bark.cpp:1231-1248)
// pos0 > 0: the cache already holds rows 0..pos0-1 of this very sequence (prefix reuse); d_tokens then holds only the new ids
int run_prefill(bark_context * c, GptModel & m, int n_tokens, bool merge, float * kbase = nullptr, float * vbase = nullptr, int pos0 = 0) {
    const int N = merge ? n_tokens - 256 : n_tokens;
    EmbedArgs e;
    e.wte = m.wte[0]; e.wte_q = m.wte_q[0]; e.wpe = m.wpe; e.E = m.hp.n_embd; e.n_in = m.hp.n_in_vocab; e.P = c->P; e.tokens = c->d_tokens; e.n_rows = N; e.merge = merge ? 1 : 0; e.pos0 = pos0; e.x = c->x;
    launch_embed_causal(c->stream, e);
    run_layers_rows(c, m, N, true, kbase, vbase, pos0);
    return N;
}

struct StageCfg {            // what differs between the semantic and the coarse decode step
    int which; int mode; int lm_row0, lm_rows, parity_rows; int token_base; float min_eos_p; int eos_token; float temp;
};
StageCfg stage_cfg(bark_context * c, int which) {
    const bark_context_params & p = c->params;
    StageCfg s{};
    s.which = which;
    s.temp = p.temp;
    if (which == 0) {
        // the reference samples over ALL n_out logits (bark.cpp:1682-1688)
        s.mode = 0; s.lm_row0 = 0; s.lm_rows = c->gpt[0].hp.n_out_vocab; s.parity_rows = 0; s.token_base = 0;
        s.min_eos_p = p.min_eos_p; s.eos_token = p.semantic_vocab_size;
    } else {
        // only the active codebook's window is sampled (bark.cpp:1829-1835) -> only its 1024 rows are evaluated
        s.mode = 1; s.lm_row0 = p.semantic_vocab_size; s.lm_rows = p.codebook_size; s.parity_rows = p.codebook_size;
        s.token_base = p.semantic_vocab_size; s.min_eos_p = 0.f; s.eos_token = -1;
    }
    return s;
}

void run_sample(bark_context * c, const StageCfg & s, int n_past_add) {
    SampleArgs a;
    a.logits = c->logits; a.n = s.lm_rows; a.mode = s.mode; a.min_eos_p = s.min_eos_p; a.eos_token = s.eos_token;
    a.token_base = s.token_base; a.n_past_add = n_past_add; a.out_tokens = c->d_out_tokens;
    a.eos_trace = s.mode == 0 ? c->d_eos_trace : nullptr; a.st = c->d_state;
    a.temp = s.temp; a.u = c->d_u;
    const GptModel & m = c->gpt[s.which];
    a.wte = m.wte[0]; a.wte_q = m.wte_q[0]; a.wpe = m.wpe; a.E = m.hp.n_embd; a.n_in = m.hp.n_in_vocab; a.P = c->P; a.x = c->x;
    launch_sample_greedy(c->stream, a);
}

// [embed(state) ->] layers -> LM head [-> greedy sample + embedding of the sampled token]
// In the greedy loop the previous sample kernel has already written x, so the step starts at the layers.
void enqueue_decode_step(bark_context * c, const StageCfg & s, bool sample, int n_past_add, bool embed = true) {
    GptModel & m = c->gpt[s.which];
    if (embed) {
        EmbedArgs e;
        e.wte = m.wte[0]; e.wte_q = m.wte_q[0]; e.wpe = m.wpe; e.E = m.hp.n_embd; e.n_in = m.hp.n_in_vocab; e.P = c->P; e.n_rows = 1; e.st = c->d_state; e.x = c->x;
        launch_embed_causal(c->stream, e);
    }
    run_layers_decode(c, m);
    run_lm_head(c, m, c->x, s.lm_row0, s.lm_rows, s.parity_rows);
    if (sample) run_sample(c, s, n_past_add);
}

hipGraphExec_t capture_decode(bark_context * c, const StageCfg & s, int n_past_add) {
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    HIP_OK(hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
    try { enqueue_decode_step(c, s, true, n_past_add, false); }
    catch (...) { hipGraph_t g2 = nullptr; (void) hipStreamEndCapture(c->stream, &g2); if (g2) (void) hipGraphDestroy(g2); throw; }
    HIP_OK(hipStreamEndCapture(c->stream, &graph));
    HIP_OK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
    (void) hipGraphDestroy(graph);
    return exec;
}

void decode_step_greedy(bark_context * c, const StageCfg & s) {
    GptModel & m = c->gpt[s.which];
    if (c->use_graph) {
        if (!m.decode_graph) m.decode_graph = capture_decode(c, s, 1);
        HIP_OK(hipGraphLaunch(m.decode_graph, c->stream));
        c->stats.graph_replays++;
    } else {
        enqueue_decode_step(c, s, true, 1, false);
    }
}

// ---- host-side sampling (temp > 0, or settling a near tie): bark.cpp:184-270 -------------------------
void softmax_host(std::vector<float> & l) {
    float mx = -INFINITY;
    for (float v : l) mx = std::max(mx, v);
    float sum = 0.0f;
    for (float & v : l) { v = (float) exp((double) (v - mx)); sum += v; }
    for (float & v : l) v /= sum;
}
int sample_host(std::vector<float> & l, std::mt19937 & rng, float temp, float * eos_p) {
    if (temp == 0.0f) {                                  // gpt_argmax_sample
        for (float & v : l) v /= 0.7f;
        softmax_host(l);
        if (eos_p) *eos_p = l.back();
        int best = 0; float mx = -INFINITY;
        for (int i = 0; i < (int) l.size(); i++) if (l[(size_t) i] > mx) { mx = l[(size_t) i]; best = i; }
        return best;
    }
    for (float & v : l) v /= temp;                       // gpt_multinomial_sample
    softmax_host(l);
    std::discrete_distribution<int32_t> dist(l.begin(), l.end());
    const int next = dist(rng);
    if (eos_p) *eos_p = l.back();
    return next;
}

std::vector<float> fetch_logits(bark_context * c, size_t n) {
    std::vector<float> l(n);
    HIP_OK(hipMemcpyAsync(l.data(), c->logits, n * 4, hipMemcpyDeviceToHost, c->stream));
    HIP_OK(hipStreamSynchronize(c->stream));
    return l;
}

// Uniform draws for `n` multinomial samples, taken from a COPY of the context's generator exactly as
// std::discrete_distribution would take them (one std::generate_canonical<double, 53> per sample = two mt19937 words);
// consume_uniforms() then advances the real generator by the samples that were actually used, so the random stream
// stays aligned with the reference's (bark.cpp:201-221) even when a stage stops early.
void upload_uniforms(bark_context * c, int n) {
    if (n > 8192) throw std::runtime_error("too many samples in one stage");
    std::mt19937 tmp = c->rng;
    std::vector<double> u((size_t) n);
    for (auto & v : u) v = std::generate_canonical<double, 53>(tmp);
    HIP_OK(hipMemcpyAsync(c->d_u, u.data(), (size_t) n * sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIP_OK(hipStreamSynchronize(c->stream));
}
void consume_uniforms(bark_context * c, int n_used) { c->rng.discard(2ull * (unsigned long long) n_used); }

void progress(bark_context * c, bark_encoding_step step, int pct) {
    if (c->params.progress_callback) c->params.progress_callback(c, step, pct, c->params.progress_callback_user_data);
}

}  // namespace

// ---------------------------------------------------------------------------------------------------
// test / binding hooks: one evaluation with full logits (bark_eval_encoder_internal, bark.cpp:1586-1643)
// ---------------------------------------------------------------------------------------------------
int engine_gpt_eval(bark_context * c, int which, const int32_t * tokens, int n_tokens, int n_past, bool merge_ctx, float * logits) {
    if (which < 0 || which > 1) throw std::runtime_error("gpt_eval: which must be 0 or 1");
    HIP_OK(hipSetDevice(c->device));
    GptModel & m = c->gpt[which];
    const bool merge = merge_ctx && n_past == 0;
    if (n_tokens <= 0) throw std::runtime_error("gpt_eval: no tokens");
    if (n_past > 0 && n_tokens != 1) throw std::runtime_error("gpt_eval: n_past > 0 needs exactly one token");   // bark.cpp:1227
    if (merge && n_tokens != 513) throw std::runtime_error("gpt_eval: merged prompt must hold 513 ids");        // bark.cpp:1231
    const int N = merge ? n_tokens - 256 : n_tokens;
    if (n_past + N > m.hp.block_size) throw std::runtime_error("gpt_eval: context overflow");
    check_ids(tokens, (size_t) n_tokens, m.hp.n_in_vocab, "gpt_eval");
    const int n_out = m.hp.n_out_vocab;
    if (N > 1) {
        upload_tokens(c, tokens, (size_t) n_tokens);
        run_prefill(c, m, n_tokens, merge);
        StepState st = fresh_state();
        set_state(c, st);
        run_lm_head(c, m, c->x + (size_t) (N - 1) * m.hp.n_embd, 0, n_out, 0);
    } else if (n_past == 0) {
        // a single-token prompt: same kernels as a decode step at position 0
        StepState st = fresh_state(); st.n_past = 0; st.cur_token = tokens[0];
        set_state(c, st);
        StageCfg s = stage_cfg(c, which); s.lm_row0 = 0; s.lm_rows = n_out; s.parity_rows = 0;
        enqueue_decode_step(c, s, false, 1);
    } else {
        StepState st = fresh_state(); st.n_past = n_past; st.cur_token = tokens[0];
        set_state(c, st);
        StageCfg s = stage_cfg(c, which); s.lm_row0 = 0; s.lm_rows = n_out; s.parity_rows = 0;
        enqueue_decode_step(c, s, false, 1);
    }
    HIP_OK(hipMemcpyAsync(logits, c->logits, (size_t) n_out * 4, hipMemcpyDeviceToHost, c->stream));
    HIP_OK(hipStreamSynchronize(c->stream));
    return n_past + N;
}

namespace {
// one fine forward (bark_build_fine_gpt_graph, bark.cpp:1416-1584): d_tokens holds [8][1024]; logits -> c->logits [1024][n_rows]
void run_fine_forward(bark_context * c, int nn, int n_rows) {
    GptModel & m = c->gpt[2];
    const int E = m.hp.n_embd;
    launch_embed_fine(c->stream, m.wte, m.wte_q, m.wpe, E, m.hp.n_in_vocab, c->d_tokens, nn, c->x);
    run_layers_rows(c, m, 1024, false);
    if (m.w32)     launch_ln_rows_f32(c->stream, c->x, 1024, E, m.lnf_g, m.lnf_b, c->xn32);
    else if (m.q4) launch_q8_rows(c->stream, c->x, 1024, E, m.lnf_g, m.lnf_b, c->xq);
    else           launch_ln_rows(c->stream, c->x, 1024, E, m.lnf_g, m.lnf_b, c->xn);
    LinArgs a;
    a.W = m.lm_head[nn - 1]; a.wq = m.lm_head_q[nn - 1]; a.xq = c->xq; if (m.w32) a.x_f32 = c->xn32; a.M = n_rows; a.K = E; a.N = 1024; a.x_f16 = c->xn; a.epi = EPI_LOGITS; a.out = c->logits; a.ld_out = n_rows;
    launch_linear(c->stream, a);                           // lm_heads[codebook_idx - n_codes_given], bark.cpp:1573
}
}  // namespace

void engine_fine_eval(bark_context * c, const int32_t * tokens_8x1024, int nn, float * logits) {
    HIP_OK(hipSetDevice(c->device));
    GptModel & m = c->gpt[2];
    if (nn < 1 || nn >= m.hp.n_wtes || nn - 1 >= m.hp.n_lm_heads) throw std::runtime_error("fine_eval: bad codebook index");
    check_ids(tokens_8x1024, 8 * 1024, m.hp.n_in_vocab, "fine_eval");
    upload_tokens(c, tokens_8x1024, 8 * 1024);
    run_fine_forward(c, nn, m.hp.n_out_vocab);
    HIP_OK(hipMemcpyAsync(logits, c->logits, (size_t) 1024 * m.hp.n_out_vocab * 4, hipMemcpyDeviceToHost, c->stream));
    HIP_OK(hipStreamSynchronize(c->stream));
}

// ---------------------------------------------------------------------------------------------------
// semantic stage: bark_eval_text_encoder (bark.cpp:1645-1701)
// ---------------------------------------------------------------------------------------------------
std::vector<int32_t> engine_semantic(bark_context * c, const std::vector<int32_t> & prompt, std::vector<float> * eos_trace) {
    HIP_OK(hipSetDevice(c->device));
    const bark_context_params & p = c->params;
    GptModel & m = c->gpt[0];
    if (prompt.size() != 513) throw std::runtime_error("semantic: prompt must hold 513 ids");
    check_ids(prompt.data(), prompt.size(), m.hp.n_in_vocab, "semantic");
    // 257 prompt rows + one row per further step must fit the context (the reference would overrun it)
    const int n_steps = std::max(0, std::min(p.n_steps_text_encoder, m.hp.block_size - 257 + 1));
    const StageCfg s = stage_cfg(c, 0);
    std::vector<int32_t> out;
    if (n_steps == 0) return out;
    upload_tokens(c, prompt.data(), prompt.size());
    StepState st = fresh_state();
    set_state(c, st);
    const bool greedy = p.temp == 0.0f || !c->host_sampling;      // "greedy" == sampled on the device (argmax or multinomial)
    if (p.temp != 0.0f && greedy) upload_uniforms(c, n_steps);
    const int N = run_prefill(c, m, 513, true);
    run_lm_head(c, m, c->x + (size_t) (N - 1) * m.hp.n_embd, s.lm_row0, s.lm_rows, 0);
    if (greedy) {
        run_sample(c, s, N);
        progress(c, SEMANTIC, 100 * 1 / std::max(1, p.n_steps_text_encoder));
        int issued = 1;
        StepState cur{};
        while (true) {
            const int batch_end = std::min(n_steps, issued + 32);
            for (; issued < batch_end; issued++) {
                decode_step_greedy(c, s);
                progress(c, SEMANTIC, 100 * (issued + 1) / std::max(1, p.n_steps_text_encoder));
            }
            cur = get_state(c);                                        // poll the stop rule every 32 steps
            if (cur.eos_step != INT32_MAX || issued >= n_steps) break;
        }
        const int n_keep = std::min(cur.eos_step, issued);
        out.resize((size_t) n_keep);
        if (n_keep) HIP_OK(hipMemcpy(out.data(), c->d_out_tokens, (size_t) n_keep * 4, hipMemcpyDeviceToHost));
        if (eos_trace) {
            const int n_tr = std::min(issued, cur.eos_step == INT32_MAX ? issued : cur.eos_step + 1);
            eos_trace->resize((size_t) n_tr);
            if (n_tr) HIP_OK(hipMemcpy(eos_trace->data(), c->d_eos_trace, (size_t) n_tr * 4, hipMemcpyDeviceToHost));
        }
        const int n_used = std::min(issued, cur.eos_step == INT32_MAX ? issued : cur.eos_step + 1);
        c->stats.n_sample_semantic += n_used;
        c->stats.n_near_tie += cur.near_tie;
        if (p.temp != 0.0f) consume_uniforms(c, n_used);
    } else {
        int n_past = N;
        for (int i = 0; i < n_steps; i++) {
            if (i > 0) {
                StepState h = fresh_state(); h.n_past = n_past; h.cur_token = out.back();
                set_state(c, h);
                enqueue_decode_step(c, s, false, 1);
                n_past += 1;
            }
            std::vector<float> l = fetch_logits(c, (size_t) s.lm_rows);
            float eos_p = 0.f;
            const int next = sample_host(l, c->rng, p.temp, &eos_p);
            c->stats.n_sample_semantic++;
            if (eos_trace) eos_trace->push_back(eos_p);
            progress(c, SEMANTIC, 100 * (i + 1) / std::max(1, p.n_steps_text_encoder));
            if (next == p.semantic_vocab_size || eos_p >= p.min_eos_p) break;      // bark.cpp:1690
            out.push_back(next);
        }
    }
    return out;
}

// ---------------------------------------------------------------------------------------------------
// coarse stage: bark_eval_coarse_encoder (bark.cpp:1745-1863)
// ---------------------------------------------------------------------------------------------------
std::vector<int32_t> engine_coarse(bark_context * c, const std::vector<int32_t> & semantic) {
    HIP_OK(hipSetDevice(c->device));
    const bark_context_params & p = c->params;
    GptModel & m = c->gpt[1];
    if (p.n_coarse_codebooks != 2 || p.codebook_size != 1024) throw std::runtime_error("coarse: only 2 codebooks of 1024 entries are supported");
    if (p.sliding_window_size <= 0 || p.max_coarse_history < 0) throw std::runtime_error("coarse: bad window parameters");
    check_ids(semantic.data(), semantic.size(), m.hp.n_in_vocab, "coarse");
    const StageCfg s = stage_cfg(c, 1);
    if (s.lm_row0 + 2 * s.lm_rows > m.hp.n_out_vocab) throw std::runtime_error("coarse: vocabulary too small");
    const float stc_ratio = p.coarse_rate_hz / p.semantic_rate_hz * p.n_coarse_codebooks;                        // bark.cpp:1757
    const int max_semantic_history = (int) floorf(p.max_coarse_history / stc_ratio);
    const int n_steps = (int) (floorf(semantic.size() * stc_ratio / p.n_coarse_codebooks) * p.n_coarse_codebooks);  // bark.cpp:1775-1779
    if (n_steps <= 0) throw std::runtime_error("coarse: no steps to run");
    const int n_windows = (int) ceilf((float) n_steps / p.sliding_window_size);
    const bool greedy = p.temp == 0.0f || !c->host_sampling;
    if (p.temp != 0.0f && greedy) upload_uniforms(c, n_steps);
    std::vector<int32_t> out;            // offset ids, as fed back into the model
    std::vector<int32_t> cached;         // token ids whose K/V rows are valid in the cache (prefix reuse)
    static const bool reuse_prefix = !getenv("BARK_HIP_NO_PREFIX_REUSE");
    int step_idx = 0;
    for (int w = 0; w < n_windows; w++) {
        // window prompt (bark.cpp:1787-1807)
        const int semantic_idx = (int) roundf(step_idx / stc_ratio);
        std::vector<int32_t> in(semantic.begin() + std::max(semantic_idx - max_semantic_history, 0), semantic.end());
        const size_t had = in.size();
        in.resize(256);
        for (size_t i = had; i < 256; i++) in[i] = p.coarse_semantic_pad_token;
        in.push_back(p.coarse_infer_token);
        const int nh = std::min(p.max_coarse_history, (int) out.size());
        in.insert(in.end(), out.end() - nh, out.end());
        const int N = (int) in.size();
        const int steps_here = std::min(p.sliding_window_size, n_steps - step_idx);
        if (N + steps_here - 1 > m.hp.block_size) throw std::runtime_error("coarse: window exceeds the context");
        check_ids(in.data(), in.size(), m.hp.n_in_vocab, "coarse");
        // Prefix reuse: the cache still holds the rows of the previous window's prompt and of the tokens decoded after
        // it.  While the semantic slice does not move and the history is not truncated, the new prompt is that very
        // sequence plus ONE token, so the reference's full re-evaluation (n_past = 0, bark.cpp:1809) collapses to a
        // decode step; in general rows [0, L) are kept and only [L, N) are evaluated.  Row i depends on rows <= i only
        // and both paths use the same canonical arithmetic, so the logits are bit-identical either way.
        int L = 0;
        if (greedy && reuse_prefix) {
            while (L < N && L < (int) cached.size() && cached[(size_t) L] == in[(size_t) L]) L++;
            if (L >= N) L = N - 1;
        }
        const int rows = N - L;
        StepState st = fresh_state(); st.step = step_idx; st.n_past = L; st.cur_token = in[(size_t) L];
        set_state(c, st);
        if (greedy && rows == 1) {
            EmbedArgs e;
            e.wte = m.wte[0]; e.wte_q = m.wte_q[0]; e.wpe = m.wpe; e.E = m.hp.n_embd; e.n_in = m.hp.n_in_vocab; e.P = c->P; e.n_rows = 1; e.st = c->d_state; e.x = c->x;
            launch_embed_causal(c->stream, e);
            decode_step_greedy(c, s);
            c->stats.n_prefix_rows_reused += L;
        } else {
            upload_tokens(c, in.data() + L, (size_t) rows);
            run_prefill(c, m, rows, false, nullptr, nullptr, L);
            c->stats.n_prefix_rows_reused += L;
        }
        if (greedy) {
            if (rows > 1) {
                run_lm_head(c, m, c->x + (size_t) (rows - 1) * m.hp.n_embd, s.lm_row0, s.lm_rows, s.parity_rows);
                run_sample(c, s, rows);
            }
            progress(c, COARSE, 100 * (step_idx + 1) / n_steps);
            for (int j = 1; j < steps_here; j++) {
                decode_step_greedy(c, s);
                progress(c, COARSE, 100 * (step_idx + j + 1) / n_steps);
            }
            const StepState cur = get_state(c);
            std::vector<int32_t> got((size_t) steps_here);
            HIP_OK(hipMemcpy(got.data(), c->d_out_tokens, (size_t) steps_here * 4, hipMemcpyDeviceToHost));
            out.insert(out.end(), got.begin(), got.end());
            cached = in;                                               // rows now in the cache: the prompt + every token fed back
            cached.insert(cached.end(), got.begin(), got.end() - 1);
            step_idx += steps_here;
            c->stats.n_sample_coarse += steps_here;
            c->stats.n_near_tie += cur.near_tie;
        } else {
            int n_past = N;
            for (int j = 0; j < steps_here; j++) {
                const int parity = step_idx % 2;
                if (j == 0) {
                    run_lm_head(c, m, c->x + (size_t) (N - 1) * m.hp.n_embd, s.lm_row0 + parity * s.lm_rows, s.lm_rows, 0);
                } else {
                    StepState h = fresh_state(); h.n_past = n_past; h.cur_token = out.back(); h.step = step_idx;
                    set_state(c, h);
                    StageCfg s2 = s; s2.lm_row0 = s.lm_row0 + parity * s.lm_rows; s2.parity_rows = 0;
                    enqueue_decode_step(c, s2, false, 1);
                    n_past += 1;
                }
                std::vector<float> l = fetch_logits(c, (size_t) s.lm_rows);
                int next = sample_host(l, c->rng, p.temp, nullptr);
                next += s.lm_row0 + parity * s.lm_rows;
                out.push_back(next);
                step_idx += 1;
                c->stats.n_sample_coarse++;
                progress(c, COARSE, 100 * step_idx / n_steps);
            }
        }
    }
    if (p.temp != 0.0f && greedy) consume_uniforms(c, n_steps);
    // de-offset into [T][2] (bark.cpp:1851-1857)
    std::vector<int32_t> res;
    for (size_t i = 0; i + 1 < out.size(); i += 2) {
        res.push_back(out[i] - p.semantic_vocab_size);
        res.push_back(out[i + 1] - p.semantic_vocab_size - p.codebook_size);
    }
    return res;
}

// ---------------------------------------------------------------------------------------------------
// fine stage: bark_eval_fine_encoder (bark.cpp:1961-2059).  T > 1024: sliding windows of 1024 frames with a hop of 512,
// as in the algorithm the reference was ported from (its own indexing is undefined there).
// ---------------------------------------------------------------------------------------------------
std::vector<int32_t> engine_fine(bark_context * c, const std::vector<int32_t> & coarse) {
    HIP_OK(hipSetDevice(c->device));
    const bark_context_params & p = c->params;
    GptModel & m = c->gpt[2];
    const int nc = p.n_coarse_codebooks, nf = p.n_fine_codebooks, cs = p.codebook_size;
    if (nc != 2 || nf != 8 || cs != 1024) throw std::runtime_error("fine: only 2 -> 8 codebooks of 1024 entries are supported");
    const int T = (int) coarse.size() / nc;
    if (T <= 0 || T > 8192) throw std::runtime_error("fine: number of frames must be in 1..8192");
    for (int32_t v : coarse) if (v < 0 || v >= cs) throw std::runtime_error("fine: coarse code out of range");
    // in_arr [L][8]: coarse rows, channels 2..7 and the time padding filled with `cs` (bark.cpp:1983-1996)
    const int L = std::max(T, 1024);
    std::vector<int32_t> in_arr((size_t) L * 8, cs);
    for (int t = 0; t < T; t++) for (int ch = 0; ch < nc; ch++) in_arr[(size_t) t * 8 + ch] = coarse[(size_t) t * nc + ch];
    const int n_loops = std::max(0, (int) ceilf((float) (L - 1024) / 512.f)) + 1;          // bark.cpp:1998
    const bool greedy = p.fine_temp == 0.0f;
    const bool device_multinomial = !greedy && !c->host_sampling;
    StepState st = fresh_state();
    set_state(c, st);
    std::vector<int32_t> buf((size_t) 8 * 1024);
    for (int n = 0; n < n_loops; n++) {
        // window n (bark.cpp:2002-2013).  T <= 1024: one window, start_idx == 0, rel == 0.  For longer inputs the reference
        // stores its samples at [rel + i] and runs out of the buffer (undefined behaviour); this engine and
        // the oracle implement the algorithm it was ported from (suno-ai/bark generate_fine): all 1024 positions are sampled
        // (the random stream advances as in the reference) and positions >= rel keep their sample.
        const int start_idx = std::min(n * 512, L - 1024);
        const int start_fill_idx = std::min(n * 512, L - 512);
        const int rel = start_fill_idx - start_idx;
        for (int ch = 0; ch < 8; ch++) for (int j = 0; j < 1024; j++) buf[(size_t) ch * 1024 + j] = in_arr[(size_t) (start_idx + j) * 8 + ch];
        upload_tokens(c, buf.data(), buf.size());
        if (device_multinomial) upload_uniforms(c, (nf - nc) * 1024);
        for (int nn = nc; nn < nf; nn++) {
            progress(c, FINE, 100 * (n * (nf - nc) + (nn - nc + 1)) / (n_loops * (nf - nc)));
            // rel > 0 (only the last windows of a long input): picks go to a scratch row, then positions >= rel are copied in
            int32_t * pick_dst = rel == 0 ? c->d_tokens + (size_t) nn * 1024 : c->d_out_tokens;
            if (greedy || device_multinomial) {
                // one pass = embed -> 12 layers -> head -> per-row pick, captured once per codebook as a hipGraph
                auto enqueue = [&] {
                    run_fine_forward(c, nn, cs);               // only logits [0, 1024) of each row are sampled (bark.cpp:2031)
                    if (greedy) launch_argmax_rows(c->stream, c->logits, cs, 1024, cs, pick_dst, 1, c->d_state);
                    else launch_sample_rows_multinomial(c->stream, c->logits, cs, 1024, cs, p.fine_temp, c->d_u + (size_t) (nn - nc) * 1024, pick_dst, 1);
                };
                if (c->use_graph && rel == 0) {
                    hipGraphExec_t & g = c->fine_graphs[nn];
                    if (!g) {
                        hipGraph_t graph = nullptr;
                        HIP_OK(hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
                        try { enqueue(); }
                        catch (...) { hipGraph_t g2 = nullptr; (void) hipStreamEndCapture(c->stream, &g2); if (g2) (void) hipGraphDestroy(g2); throw; }
                        HIP_OK(hipStreamEndCapture(c->stream, &graph));
                        HIP_OK(hipGraphInstantiate(&g, graph, nullptr, nullptr, 0));
                        (void) hipGraphDestroy(graph);
                    }
                    HIP_OK(hipGraphLaunch(g, c->stream));
                    c->stats.graph_replays++;
                } else {
                    enqueue();
                }
                if (rel > 0)
                    HIP_OK(hipMemcpyAsync(c->d_tokens + (size_t) nn * 1024 + rel, c->d_out_tokens + rel, (size_t) (1024 - rel) * 4, hipMemcpyDeviceToDevice, c->stream));
            } else {
                const int n_out = m.hp.n_out_vocab;
                run_fine_forward(c, nn, n_out);
                std::vector<float> l = fetch_logits(c, (size_t) 1024 * n_out);
                std::vector<int32_t> ch(1024);
                for (int i = 0; i < 1024; i++) {
                    std::vector<float> relv(l.begin() + (size_t) i * n_out, l.begin() + (size_t) i * n_out + cs);
                    ch[(size_t) i] = sample_host(relv, c->rng, p.fine_temp, nullptr);
                }
                HIP_OK(hipMemcpyAsync(c->d_tokens + (size_t) nn * 1024 + rel, ch.data() + rel, (size_t) (1024 - rel) * 4, hipMemcpyHostToDevice, c->stream));
                HIP_OK(hipStreamSynchronize(c->stream));
            }
            c->stats.n_sample_fine += 1024;
        }
        if (device_multinomial) consume_uniforms(c, (nf - nc) * 1024);
        HIP_OK(hipMemcpyAsync(buf.data(), c->d_tokens, buf.size() * 4, hipMemcpyDeviceToHost, c->stream));
        HIP_OK(hipStreamSynchronize(c->stream));
        for (int nn = nc; nn < nf; nn++)                                                     // bark.cpp:2041-2046
            for (int j = 0; j < cs - rel; j++) in_arr[(size_t) (start_fill_idx + j) * 8 + nn] = buf[(size_t) nn * 1024 + rel + j];
    }
    const StepState cur = get_state(c);
    c->stats.n_near_tie += cur.near_tie;
    in_arr.resize((size_t) T * 8);                                                           // strip the time padding
    return in_arr;
}

// ---------------------------------------------------------------------------------------------------
// EnCodec decode (encodec_decompress_audio call site, bark.cpp:2143-2167)
// ---------------------------------------------------------------------------------------------------
std::vector<float> engine_codec_decode(bark_context * c, const int32_t * codes, int n_q, int T, int tap_stage, std::vector<float> * tap) {
    HIP_OK(hipSetDevice(c->device));
    CodecModel & cm = c->codec;
    if (n_q <= 0 || n_q > cm.n_q || T <= 0 || T > 4096) throw std::runtime_error("codec: bad code matrix shape");
    for (size_t i = 0; i < (size_t) n_q * T; i++) if (codes[i] < 0 || codes[i] >= cm.hp.n_bins) throw std::runtime_error("codec: code out of range");
    hipStream_t s = c->stream;
    const int D = cm.D;
    int hop = 1; for (auto & b : cm.blocks) hop *= b.up.stride;
    // largest activation: channels x time at every stage
    size_t need = (size_t) std::max(cm.hp.hidden_dim, D) * T;
    { int ch = D, tt = T; for (auto & b : cm.blocks) { ch = b.up.cout; tt *= b.up.stride; need = std::max(need, (size_t) ch * tt); } }
    if (need > c->cbuf_elems) {
        for (auto & b : c->cbuf) b = dev_alloc<float>(c, need);
        c->cbuf_h = dev_alloc<half_t>(c, need);
        c->cbuf_elems = need;
    }
    if ((size_t) T > c->c_T) {
        c->c_gi = dev_alloc<float>(c, (size_t) T * 4 * D);
        c->c_cell = dev_alloc<float>(c, (size_t) D);
        c->c_hseq_h = dev_alloc<half_t>(c, (size_t) T * D);
        c->c_xt_h = dev_alloc<half_t>(c, (size_t) T * D);
        c->c_T = (size_t) T;
    }
    if ((size_t) n_q * T > c->d_codes_elems) { c->d_codes = dev_alloc<int32_t>(c, (size_t) n_q * T); c->d_codes_elems = (size_t) n_q * T; }
    HIP_OK(hipMemcpyAsync(c->d_codes, codes, (size_t) n_q * T * 4, hipMemcpyHostToDevice, s));
    float * A = c->cbuf[0], * B = c->cbuf[1], * R = c->cbuf[2];
    half_t * Hh = c->cbuf_h;
    static const bool blocked = !getenv("BARK_HIP_CODEC_NAIVE");      // register-blocked convs (default) vs the one-output-per-thread kernels

    auto conv = [&](const CodecModel::Conv & cv, const float * in, bool elu, int Tc, const float * add, float * out) {
        launch_act_round(s, in, (size_t) cv.cin * Tc, elu ? 1 : 0, Hh);
        if (blocked && cv.w32 && conv1d_f32w_supported(cv.k)) launch_conv1d_f32w(s, cv.w32, cv.b, cv.cout, cv.cin, cv.k, Hh, Tc, add, out);
        else launch_conv1d(s, cv.w, cv.b, cv.cout, cv.cin, cv.k, Hh, Tc, add, out);
    };
    // RVQ de-embedding, first conv
    launch_rvq_gather(s, cm.codebooks, cm.hp.n_bins, cm.hp.hidden_dim, c->d_codes, n_q, T, A);
    conv(cm.init, A, false, T, nullptr, B);                            // B = x [D][T]
    // 2-layer LSTM + skip (modeling_encodec.py:236-249)
    const float * lin = B;
    for (int l = 0; l < 2; l++) {
        const half_t * seq_h;
        if (l == 0) { launch_transpose_round(s, lin, D, T, c->c_xt_h); seq_h = c->c_xt_h; }
        else { HIP_OK(hipMemcpyAsync(c->c_xt_h, c->c_hseq_h, (size_t) T * D * sizeof(half_t), hipMemcpyDeviceToDevice, s)); seq_h = c->c_xt_h; }
        LinArgs g;
        g.W = cm.lstm[l].w_ih; g.M = 4 * D; g.K = D; g.N = T; g.x_f16 = seq_h; g.epi = EPI_LOGITS; g.out = c->c_gi; g.ld_out = 4 * D;
        launch_linear(s, g);
        float * hseq = (l == 0) ? A : R;                                // layer outputs [D][T]
        // T strictly sequential steps.  64 of them are captured once per context and layer as a hipGraph whose nodes
        // take their step index from a device counter (base) + the node's offset, so one graph serves every T.
        LstmStepArgs a;
        a.w_hh = cm.lstm[l].w_hh; a.b_ih = cm.lstm[l].b_ih; a.b_hh = cm.lstm[l].b_hh; a.gi = c->c_gi;
        a.c = c->c_cell; a.hseq_h = c->c_hseq_h; a.hseq = hseq; a.T = T; a.D = D;
        if (!c->use_graph) {
            for (int t = 0; t < T; t++) { a.t = t; launch_lstm_step(s, a); }
        } else {
            constexpr int kBlock = 64;
            auto & slot = c->lstm_graphs[l];
            if (slot.exec && (slot.hseq != hseq || slot.gi != c->c_gi)) { (void) hipGraphExecDestroy(slot.exec); slot.exec = nullptr; }
            if (!slot.exec) {
                hipGraph_t graph = nullptr;
                HIP_OK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
                a.t_base = c->d_lstm_t;
                for (int i = 0; i < kBlock; i++) { a.t = i; launch_lstm_step(s, a); }
                launch_add_int(s, c->d_lstm_t, kBlock);
                HIP_OK(hipStreamEndCapture(s, &graph));
                HIP_OK(hipGraphInstantiate(&slot.exec, graph, nullptr, nullptr, 0));
                (void) hipGraphDestroy(graph);
                slot.T = T; slot.hseq = hseq; slot.gi = c->c_gi;
            }
            const int hdr[2] = {0, T};
            HIP_OK(hipMemcpyAsync(c->d_lstm_t, hdr, sizeof(hdr), hipMemcpyHostToDevice, s));
            HIP_OK(hipStreamSynchronize(s));                         // hdr is a stack object
            for (int t0 = 0; t0 < T; t0 += kBlock) HIP_OK(hipGraphLaunch(slot.exec, s));
        }
    }
    auto grab = [&](int stage, const float * buf, size_t n) {
        if (tap_stage != stage || !tap) return;
        tap->resize(n);
        HIP_OK(hipMemcpyAsync(tap->data(), buf, n * 4, hipMemcpyDeviceToHost, s));
        HIP_OK(hipStreamSynchronize(s));
    };
    grab(0, B, (size_t) D * T);
    launch_add(s, R, B, (size_t) D * T, A);                            // y + x ; A = x
    grab(1, A, (size_t) D * T);
    float * cur = A, * other = B;
    int Tc = T;
    for (int b = 0; b < 4; b++) {
        const CodecModel::Block & bl = cm.blocks[b];
        launch_act_round(s, cur, (size_t) bl.up.cin * Tc, 1, Hh);
        if (blocked && bl.up.w32 && bl.up.k == 2 * bl.up.stride) launch_convtr1d_f32w(s, bl.up.w32, bl.up.b, bl.up.cin, bl.up.cout, bl.up.k, bl.up.stride, Hh, Tc, other);
        else launch_convtr1d(s, bl.up.w, bl.up.b, bl.up.cin, bl.up.cout, bl.up.k, bl.up.stride, Hh, Tc, other);
        Tc *= bl.up.stride;
        std::swap(cur, other);                                          // cur = upsampled x
        // residual block: shortcut(x) + conv2(elu(conv1(elu(x))))   (modeling_encodec.py:252-282)
        conv(bl.c1, cur, true, Tc, nullptr, R);
        conv(bl.c2, R, true, Tc, nullptr, other);                       // other = r
        conv(bl.sc, cur, false, Tc, other, R);                          // R = shortcut(x) + r
        std::swap(cur, R);
        // keep three distinct buffers: cur (result), other, R (old x)
        grab(2 + b, cur, (size_t) bl.up.cout * Tc);
    }
    conv(cm.fin, cur, true, Tc, nullptr, other);
    std::vector<float> pcm((size_t) Tc);
    HIP_OK(hipMemcpyAsync(pcm.data(), other, (size_t) Tc * 4, hipMemcpyDeviceToHost, s));
    HIP_OK(hipStreamSynchronize(s));
    (void) hop;
    return pcm;
}

// ---------------------------------------------------------------------------------------------------
// bark_generate_audio (bark.cpp:2125-2172)
// ---------------------------------------------------------------------------------------------------
bool engine_generate(bark_context * c, const char * text) {
    HIP_OK(hipSetDevice(c->device));
    const int64_t t0 = now_us();
    const int64_t t_load = c->stats.t_load_us;
    c->stats = bark_hip_stats{};
    c->stats.t_load_us = t_load;
    c->audio.clear(); c->semantic_tokens.clear(); c->coarse_tokens.clear(); c->fine_tokens.clear();
    PromptParams pp;
    pp.block_size = c->gpt[0].hp.block_size; pp.text_encoding_offset = c->params.text_encoding_offset; pp.text_pad_token = c->params.text_pad_token;
    pp.semantic_pad_token = c->params.semantic_pad_token; pp.semantic_infer_token = c->params.semantic_infer_token;
    c->tokens = build_semantic_prompt(c->vocab, pp, text, true);
    if (c->params.verbosity >= MEDIUM) {
        fprintf(stderr, "bark_tokenize_input: prompt: '%s'\nbark_tokenize_input: number of tokens in prompt = %zu, first 8 tokens:", text, c->tokens.size());
        for (int i = 0; i < 8 && i < (int) c->tokens.size(); i++) fprintf(stderr, " %d", c->tokens[(size_t) i]);
        fprintf(stderr, "\n");
    }
    int64_t t = now_us();
    c->semantic_tokens = engine_semantic(c, c->tokens, nullptr);
    c->stats.t_semantic_us = now_us() - t;
    c->stats.n_semantic = (int32_t) c->semantic_tokens.size();
    if (c->semantic_tokens.empty()) { fprintf(stderr, "bark_generate_audio: the semantic stage produced no tokens\n"); return false; }
    t = now_us();
    c->coarse_tokens = engine_coarse(c, c->semantic_tokens);
    c->stats.t_coarse_us = now_us() - t;
    t = now_us();
    c->fine_tokens = engine_fine(c, c->coarse_tokens);
    c->stats.t_fine_us = now_us() - t;
    const int T = (int) c->fine_tokens.size() / 8;
    c->stats.n_frames = T;
    std::vector<int32_t> codes((size_t) 8 * T);
    for (int ch = 0; ch < 8; ch++) for (int i = 0; i < T; i++) codes[(size_t) ch * T + i] = c->fine_tokens[(size_t) i * 8 + ch];   // bark.cpp:2153-2159
    t = now_us();
    c->audio = engine_codec_decode(c, codes.data(), 8, T, -1, nullptr);
    c->stats.t_codec_us = now_us() - t;
    c->stats.n_samples = (int32_t) c->audio.size();
    c->stats.t_eval_us = now_us() - t0;
    if (c->params.verbosity >= MEDIUM) {
        auto line = [](const char * name, int64_t n, int64_t us) {
            fprintf(stderr, "%s: %8.2f ms / %lld samples (%.3f ms per sample)\n", name, us / 1000.0, (long long) n, n ? us / 1000.0 / n : 0.0);
        };
        line("semantic", c->stats.n_sample_semantic, c->stats.t_semantic_us);
        line("coarse  ", c->stats.n_sample_coarse, c->stats.t_coarse_us);
        line("fine    ", c->stats.n_sample_fine, c->stats.t_fine_us);
        fprintf(stderr, "codec   : %8.2f ms / %d frames ; total %8.2f ms for %.2f s of audio\n", c->stats.t_codec_us / 1000.0, T,
                c->stats.t_eval_us / 1000.0, c->audio.size() / (double) c->params.sample_rate);
    }
    return true;
}

// ---------------------------------------------------------------------------------------------------
// batched decode: B utterances advance in lock step through the semantic and coarse decode loops; every decode
// kernel processes all slots (weights are read from HBM once per step instead of once per utterance), prefill,
// fine passes and the codec still run per utterance.  Per-slot arithmetic is exactly the single-utterance one.
// ---------------------------------------------------------------------------------------------------
namespace {

void ensure_batch(bark_context * c, int B) {
    bark_context::Batch & bb = c->batch;
    if (bb.cap >= B) return;
    if (bb.cap) throw std::runtime_error("batch capacity is fixed by the first bark_hip_generate_batch call of a context");
    const int E = c->max_E;
    for (int g = 0; g < 2; g++) {
        GptModel & m = c->gpt[g];
        bb.slot_stride[g] = m.kv_layer_stride * m.hp.n_layer;
        bb.kc[g] = dev_alloc<float>(c, bb.slot_stride[g] * B);
        bb.vc[g] = dev_alloc<float>(c, bb.slot_stride[g] * B);
    }
    bb.ld_logits = 0;
    for (int g = 0; g < 2; g++) bb.ld_logits = std::max(bb.ld_logits, (size_t) c->gpt[g].hp.n_out_vocab);
    bb.x = dev_alloc<float>(c, (size_t) B * E);
    bb.q = dev_alloc<float>(c, (size_t) B * E);
    bb.att = dev_alloc<half_t>(c, (size_t) B * E);
    bb.h = dev_alloc<half_t>(c, (size_t) B * 4 * E);
    bb.logits = dev_alloc<float>(c, (size_t) B * bb.ld_logits);
    bb.state = dev_alloc<StepState>(c, (size_t) B);
    bb.ln_stats = dev_alloc<float>(c, (size_t) B * 2);
    bb.out_tokens = dev_alloc<int32_t>(c, (size_t) B * 2048);
    bb.eos_trace = dev_alloc<float>(c, (size_t) B * 2048);
    bb.u = dev_alloc<double>(c, (size_t) B * 8192);
    if (c->any_q4) { bb.att32 = dev_alloc<float>(c, (size_t) B * E); bb.h32 = dev_alloc<float>(c, (size_t) B * 4 * E); }
    bb.cap = B;
}

void set_slot_state(bark_context * c, int slot, const StepState & st) {
    HIP_OK(hipMemcpyAsync(c->batch.state + slot, &st, sizeof(st), hipMemcpyHostToDevice, c->stream));
    HIP_OK(hipStreamSynchronize(c->stream));
}
std::vector<StepState> get_slot_states(bark_context * c, int B) {
    std::vector<StepState> st((size_t) B);
    HIP_OK(hipMemcpyAsync(st.data(), c->batch.state, sizeof(StepState) * B, hipMemcpyDeviceToHost, c->stream));
    HIP_OK(hipStreamSynchronize(c->stream));
    return st;
}

// all slots: layers -> LM head -> greedy sample (+ embedding of the sampled token)
void enqueue_batch_step(bark_context * c, const StageCfg & s, int B, const bark_context::Batch & bb) {
    GptModel & m = c->gpt[s.which];
    const int E = m.hp.n_embd, H = m.hp.n_head, P = c->P;
    hipStream_t st = c->stream;
    float * kc0 = bb.kc[s.which], * vc0 = bb.vc[s.which];
    const size_t slot = bb.slot_stride[s.which];
    // LayerNorm statistics: recomputed inside every GEMV wave for small batches (an extra launch costs ~2 us), hoisted into
    // ln_stats_kernel for large ones (measured cross-over on MI355X between 16 and 32 slots)
    const bool hoist = B >= 24 && !m.q4;
    for (int l = 0; l < m.hp.n_layer; l++) {
        const GptModel::Layer & L = m.layers[(size_t) l];
        float * kl = kc0 + m.kv_layer_stride * (size_t) l, * vl = vc0 + m.kv_layer_stride * (size_t) l;
        if (hoist) launch_ln_stats(st, bb.x, B, E, bb.ln_stats);
        LinArgs a;
        a.batched = 1; a.nbatch = B; a.kv_slot_stride = slot; a.ln_stats = hoist ? bb.ln_stats : nullptr;
        a.W = L.attn_w; a.wq = L.attn_q; a.M = 3 * E; a.K = E; a.N = 1; a.x_f32 = bb.x; a.ln_g = L.ln1_g; a.ln_b = L.ln1_b; a.bias = L.attn_b;
        a.epi = EPI_QKV; a.q = bb.q; a.kc = kl; a.vc = vl; a.E = E; a.P = P; a.pos0 = 0; a.st = bb.state;
        launch_linear(st, a);
        AttnDecodeArgs at;
        at.q = bb.q; at.kc = kl; at.vc = vl; at.H = H; at.P = P; at.st = bb.state; at.att = bb.att; at.scores = c->scores; at.hmax = c->d_hmax;
        at.nbatch = B; at.kv_slot_stride = slot; at.att32 = m.q4 ? bb.att32 : nullptr;
        launch_attn_decode_part(st, at, 4);
        LinArgs p;
        p.batched = 1; p.nbatch = B;
        p.W = L.proj_w; p.wq = L.proj_q; p.M = E; p.K = E; p.N = 1; if (m.q4) p.x_f32 = bb.att32; else p.x_f16 = bb.att; p.bias = L.proj_b; p.epi = EPI_RESID; p.res = bb.x;
        launch_linear(st, p);
        if (hoist) launch_ln_stats(st, bb.x, B, E, bb.ln_stats);
        LinArgs f;
        f.batched = 1; f.nbatch = B; f.ln_stats = hoist ? bb.ln_stats : nullptr;
        f.W = L.fc_w; f.wq = L.fc_q; f.M = 4 * E; f.K = E; f.N = 1; f.x_f32 = bb.x; f.ln_g = L.ln2_g; f.ln_b = L.ln2_b; f.bias = L.fc_b;
        f.epi = EPI_GELU; f.out_h = bb.h; f.out_h32 = m.q4 ? bb.h32 : nullptr; f.lut = c->d_gelu_lut;
        launch_linear(st, f);
        LinArgs o;
        o.batched = 1; o.nbatch = B;
        o.W = L.mproj_w; o.wq = L.mproj_q; o.M = E; o.K = 4 * E; o.N = 1; if (m.q4) o.x_f32 = bb.h32; else o.x_f16 = bb.h; o.bias = L.mproj_b; o.epi = EPI_RESID; o.res = bb.x;
        launch_linear(st, o);
    }
    if (hoist) launch_ln_stats(st, bb.x, B, E, bb.ln_stats);
    LinArgs h;
    h.batched = 1; h.nbatch = B; h.ln_stats = hoist ? bb.ln_stats : nullptr;
    if (m.q4) h.wq = q4_rows(m.lm_head_q[0], (size_t) s.lm_row0, E); else h.W = m.lm_head[0] + (size_t) s.lm_row0 * E;
    h.M = s.lm_rows; h.K = E; h.N = 1; h.x_f32 = bb.x; h.ln_g = m.lnf_g; h.ln_b = m.lnf_b;
    h.epi = EPI_LOGITS; h.out = bb.logits; h.ld_out = (int) bb.ld_logits; h.parity_rows = s.parity_rows; h.st = bb.state;
    launch_linear(st, h);
    SampleArgs sa;
    sa.logits = bb.logits; sa.n = s.lm_rows; sa.mode = s.mode; sa.min_eos_p = s.min_eos_p; sa.eos_token = s.eos_token;
    sa.token_base = s.token_base; sa.n_past_add = 1; sa.out_tokens = bb.out_tokens; sa.eos_trace = s.mode == 0 ? bb.eos_trace : nullptr;
    sa.st = bb.state; sa.nbatch = B; sa.ld_logits = (int) bb.ld_logits; sa.out_stride = 2048;
    sa.temp = s.temp; sa.u = bb.u; sa.u_stride = 8192;
    sa.wte = m.wte[0]; sa.wte_q = m.wte_q[0]; sa.wpe = m.wpe; sa.E = E; sa.n_in = m.hp.n_in_vocab; sa.P = P; sa.x = bb.x;
    launch_sample_greedy(st, sa);
}

void batch_step(bark_context * c, const StageCfg & s, int B) {
    bark_context::Batch & bb = c->batch;
    if (c->use_graph) {
        if (bb.graph[s.which] && bb.graph_B[s.which] != B) { (void) hipGraphExecDestroy(bb.graph[s.which]); bb.graph[s.which] = nullptr; }
        if (!bb.graph[s.which]) {
            hipGraph_t graph = nullptr;
            HIP_OK(hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
            try { enqueue_batch_step(c, s, B, bb); }
            catch (...) { hipGraph_t g2 = nullptr; (void) hipStreamEndCapture(c->stream, &g2); if (g2) (void) hipGraphDestroy(g2); throw; }
            HIP_OK(hipStreamEndCapture(c->stream, &graph));
            HIP_OK(hipGraphInstantiate(&bb.graph[s.which], graph, nullptr, nullptr, 0));
            (void) hipGraphDestroy(graph);
            bb.graph_B[s.which] = B;
        }
        HIP_OK(hipGraphLaunch(bb.graph[s.which], c->stream));
        c->stats.graph_replays++;
    } else {
        enqueue_batch_step(c, s, B, bb);
    }
}

// the buffers of slot b alone, as a batch of one
bark_context::Batch slot_view(const bark_context * c, const StageCfg & s, int b) {
    bark_context::Batch v = c->batch;
    const size_t E = (size_t) c->gpt[s.which].hp.n_embd;
    for (int g = 0; g < 2; g++) { v.kc[g] += v.slot_stride[g] * (size_t) b; v.vc[g] += v.slot_stride[g] * (size_t) b; }
    v.x += E * b; v.q += E * b; v.att += E * b; v.h += 4 * E * b; v.logits += v.ld_logits * (size_t) b;
    if (v.att32) { v.att32 += E * b; v.h32 += 4 * E * b; }
    v.state += b; v.out_tokens += (size_t) b * 2048; v.eos_trace += (size_t) b * 2048; v.ln_stats += 2 * (size_t) b; v.u += (size_t) b * 8192;
    v.graph[0] = v.graph[1] = nullptr;
    return v;
}
void embed_slot(bark_context * c, const StageCfg & s, int b) {
    GptModel & m = c->gpt[s.which];
    EmbedArgs e;
    e.wte = m.wte[0]; e.wte_q = m.wte_q[0]; e.wpe = m.wpe; e.E = m.hp.n_embd; e.n_in = m.hp.n_in_vocab; e.P = c->P; e.n_rows = 1;
    e.st = c->batch.state + b; e.x = c->batch.x + (size_t) b * m.hp.n_embd;
    launch_embed_causal(c->stream, e);
}

// temp > 0: the next `n` uniform draws of a slot's own generator, taken from a COPY as in upload_uniforms()
void upload_slot_uniforms(bark_context * c, int slot, const std::mt19937 & rng, int n) {
    if (n > 8192) throw std::runtime_error("too many samples in one stage");
    std::mt19937 tmp = rng;
    std::vector<double> u((size_t) std::max(n, 1));
    for (auto & v : u) v = std::generate_canonical<double, 53>(tmp);
    HIP_OK(hipMemcpyAsync(c->batch.u + (size_t) slot * 8192, u.data(), (size_t) n * sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIP_OK(hipStreamSynchronize(c->stream));
}

// prompt of one slot through the model (single-utterance kernels, the slot's own cache), first sample of the slot
// L > 0: rows [0, L) of the prompt are already in the slot's cache (prefix reuse); only ids[L..] are evaluated
void batch_prefill_and_sample(bark_context * c, const StageCfg & s, int slot, const std::vector<int32_t> & ids, bool merge, int step0, int L = 0) {
    GptModel & m = c->gpt[s.which];
    bark_context::Batch & bb = c->batch;
    check_ids(ids.data(), ids.size(), m.hp.n_in_vocab, "batch prefill");
    upload_tokens(c, ids.data() + L, ids.size() - (size_t) L);
    StepState st = fresh_state(); st.step = step0; st.n_past = L;
    set_slot_state(c, slot, st);
    float * kb = bb.kc[s.which] + bb.slot_stride[s.which] * (size_t) slot, * vb = bb.vc[s.which] + bb.slot_stride[s.which] * (size_t) slot;
    const int N = run_prefill(c, m, (int) ids.size() - L, merge, kb, vb, L);
    LinArgs h;
    if (m.q4) h.wq = q4_rows(m.lm_head_q[0], (size_t) s.lm_row0, m.hp.n_embd); else h.W = m.lm_head[0] + (size_t) s.lm_row0 * m.hp.n_embd;
    h.M = s.lm_rows; h.K = m.hp.n_embd; h.N = 1;
    h.x_f32 = c->x + (size_t) (N - 1) * m.hp.n_embd; h.ln_g = m.lnf_g; h.ln_b = m.lnf_b; h.epi = EPI_LOGITS;
    h.out = bb.logits + bb.ld_logits * (size_t) slot; h.ld_out = (int) bb.ld_logits; h.parity_rows = s.parity_rows; h.st = bb.state + slot;
    launch_linear(c->stream, h);
    SampleArgs sa;
    sa.logits = bb.logits + bb.ld_logits * (size_t) slot; sa.n = s.lm_rows; sa.mode = s.mode; sa.min_eos_p = s.min_eos_p; sa.eos_token = s.eos_token;
    sa.token_base = s.token_base; sa.n_past_add = N; sa.out_tokens = bb.out_tokens + (size_t) slot * 2048;
    sa.eos_trace = s.mode == 0 ? bb.eos_trace + (size_t) slot * 2048 : nullptr; sa.st = bb.state + slot;
    sa.temp = s.temp; sa.u = bb.u + (size_t) slot * 8192;
    sa.wte = m.wte[0]; sa.wte_q = m.wte_q[0]; sa.wpe = m.wpe; sa.E = m.hp.n_embd; sa.n_in = m.hp.n_in_vocab; sa.P = c->P; sa.x = bb.x + (size_t) slot * m.hp.n_embd;
    launch_sample_greedy(c->stream, sa);
}

}  // namespace

int engine_generate_batch(bark_context * c, const char * const * texts, int n, const uint32_t * seeds) {
    HIP_OK(hipSetDevice(c->device));
    const bark_context_params & p = c->params;
    if (n <= 0 || n > 32) throw std::runtime_error("generate_batch: batch size must be in 1..32");
    c->batch_results.assign((size_t) n, bark_context::BatchResult());
    // one generator per utterance (bark.cpp:1179 seeds one per context): utterance i of a batch is what a fresh context with
    // seed seeds[i] would generate.  Without explicit seeds they are drawn from the context's generator, in order.
    std::vector<std::mt19937> slot_rng((size_t) n);
    for (int i = 0; i < n; i++) slot_rng[(size_t) i] = std::mt19937(seeds ? seeds[i] : (uint32_t) c->rng());
    const bool sampled = p.temp != 0.0f;
    if (c->host_sampling || c->gpt[0].hp.n_embd != c->gpt[1].hp.n_embd || c->any_w32) {
        // host-side sampling and f32 model files keep one utterance in flight: fall back to the sequential loop
        int good = 0;
        for (int i = 0; i < n; i++) {
            bark_context::BatchResult & r = c->batch_results[(size_t) i];
            std::swap(c->rng, slot_rng[(size_t) i]);
            try { r.ok = engine_generate(c, texts[i]); } catch (...) { std::swap(c->rng, slot_rng[(size_t) i]); throw; }
            std::swap(c->rng, slot_rng[(size_t) i]);
            if (r.ok) { r.semantic = c->semantic_tokens; r.coarse = c->coarse_tokens; r.fine = c->fine_tokens; r.audio = c->audio; good++; }
        }
        return good;
    }
    const int64_t t0 = now_us();
    const int64_t t_load = c->stats.t_load_us;
    c->stats = bark_hip_stats{};
    c->stats.t_load_us = t_load;
    const int B = n;
    ensure_batch(c, c->batch.cap ? c->batch.cap : std::max(B, 8));
    bark_context::Batch & bb = c->batch;
    if (B > bb.cap) throw std::runtime_error("generate_batch: batch larger than the capacity fixed by the first call");
    HIP_OK(hipMemsetAsync(c->d_hmax, 0, 64 * sizeof(unsigned), c->stream));

    // ---- semantic (bark.cpp:1645-1701), lock step -----------------------------------------------------
    int64_t t = now_us();
    {
        GptModel & m = c->gpt[0];
        const StageCfg s = stage_cfg(c, 0);
        const int n_steps = std::max(0, std::min(p.n_steps_text_encoder, m.hp.block_size - 257 + 1));
        PromptParams pp;
        pp.block_size = m.hp.block_size; pp.text_encoding_offset = p.text_encoding_offset; pp.text_pad_token = p.text_pad_token;
        pp.semantic_pad_token = p.semantic_pad_token; pp.semantic_infer_token = p.semantic_infer_token;
        if (n_steps > 0) {
            if (sampled) for (int b = 0; b < B; b++) upload_slot_uniforms(c, b, slot_rng[(size_t) b], n_steps);
            for (int b = 0; b < B; b++) batch_prefill_and_sample(c, s, b, build_semantic_prompt(c->vocab, pp, texts[b], true), true, 0);
            int issued = 1;
            std::vector<StepState> st;
            while (true) {
                const int batch_end = std::min(n_steps, issued + 32);
                for (; issued < batch_end; issued++) { batch_step(c, s, B); progress(c, SEMANTIC, 100 * (issued + 1) / std::max(1, p.n_steps_text_encoder)); }
                st = get_slot_states(c, B);
                bool all_done = true;
                for (auto & v : st) all_done = all_done && v.eos_step != INT32_MAX;
                if (all_done || issued >= n_steps) break;
            }
            for (int b = 0; b < B; b++) {
                const int keep = std::min(st[(size_t) b].eos_step, issued);
                auto & out = c->batch_results[(size_t) b].semantic;
                out.resize((size_t) keep);
                if (keep) HIP_OK(hipMemcpy(out.data(), bb.out_tokens + (size_t) b * 2048, (size_t) keep * 4, hipMemcpyDeviceToHost));
                const int n_used = std::min(issued, st[(size_t) b].eos_step == INT32_MAX ? issued : st[(size_t) b].eos_step + 1);
                c->stats.n_sample_semantic += n_used;
                if (sampled) slot_rng[(size_t) b].discard(2ull * (unsigned long long) n_used);      // as consume_uniforms()
                c->stats.n_near_tie += st[(size_t) b].near_tie;
            }
        }
    }
    c->stats.t_semantic_us = now_us() - t;

    // ---- coarse (bark.cpp:1745-1863), windows in lock step ---------------------------------------------
    t = now_us();
    std::vector<std::vector<int32_t>> coarse_out((size_t) B);
    {
        GptModel & m = c->gpt[1];
        const StageCfg s = stage_cfg(c, 1);
        if (p.n_coarse_codebooks != 2 || p.codebook_size != 1024 || p.sliding_window_size <= 0 || p.max_coarse_history < 0)
            throw std::runtime_error("coarse: unsupported parameters");
        const float stc_ratio = p.coarse_rate_hz / p.semantic_rate_hz * p.n_coarse_codebooks;
        const int max_semantic_history = (int) floorf(p.max_coarse_history / stc_ratio);
        std::vector<int> n_steps((size_t) B, 0), step_idx((size_t) B, 0);
        int max_windows = 0;
        for (int b = 0; b < B; b++) {
            const auto & sem = c->batch_results[(size_t) b].semantic;
            if (sem.empty()) continue;
            n_steps[(size_t) b] = (int) (floorf(sem.size() * stc_ratio / p.n_coarse_codebooks) * p.n_coarse_codebooks);
            max_windows = std::max(max_windows, (int) ceilf((float) n_steps[(size_t) b] / p.sliding_window_size));
            if (sampled) upload_slot_uniforms(c, b, slot_rng[(size_t) b], n_steps[(size_t) b]);          // indexed by the slot's step_idx
        }
        std::vector<std::vector<int32_t>> cached((size_t) B);          // per slot: ids whose K/V rows are in its cache
        static const bool reuse_prefix = !getenv("BARK_HIP_NO_PREFIX_REUSE");
        for (int w = 0; w < max_windows; w++) {
            int max_here = 0;
            std::vector<int> here((size_t) B, 0), Ls((size_t) B, 0);
            std::vector<std::vector<int32_t>> ins((size_t) B);
            bool all_single = true;                                      // every live slot needs exactly one new row
            for (int b = 0; b < B; b++) {
                if (step_idx[(size_t) b] >= n_steps[(size_t) b]) continue;
                const auto & sem = c->batch_results[(size_t) b].semantic;
                auto & out = coarse_out[(size_t) b];
                const int semantic_idx = (int) roundf(step_idx[(size_t) b] / stc_ratio);
                std::vector<int32_t> in(sem.begin() + std::max(semantic_idx - max_semantic_history, 0), sem.end());
                const size_t had = in.size();
                in.resize(256);
                for (size_t i = had; i < 256; i++) in[i] = p.coarse_semantic_pad_token;
                in.push_back(p.coarse_infer_token);
                const int nh = std::min(p.max_coarse_history, (int) out.size());
                in.insert(in.end(), out.end() - nh, out.end());
                here[(size_t) b] = std::min(p.sliding_window_size, n_steps[(size_t) b] - step_idx[(size_t) b]);
                if ((int) in.size() + here[(size_t) b] - 1 > m.hp.block_size) throw std::runtime_error("coarse: window exceeds the context");
                check_ids(in.data(), in.size(), m.hp.n_in_vocab, "coarse");
                int L = 0;
                if (reuse_prefix) {
                    const auto & cd = cached[(size_t) b];
                    while (L < (int) in.size() && L < (int) cd.size() && cd[(size_t) L] == in[(size_t) L]) L++;
                    if (L >= (int) in.size()) L = (int) in.size() - 1;
                }
                Ls[(size_t) b] = L;
                if ((int) in.size() - L != 1) all_single = false;
                ins[(size_t) b] = std::move(in);
                max_here = std::max(max_here, here[(size_t) b]);
            }
            int lock_steps = max_here - 1;                               // batched steps after every live slot has its first sample
            for (int b = 0; b < B; b++) {
                if (!here[(size_t) b]) {                                 // finished (or empty) slot: park it at position 0
                    StepState idle = fresh_state(); idle.cur_token = 0;
                    idle.step = w * p.sliding_window_size;          // same codebook parity as the live slots (slot 0's step selects the LM-head rows)
                    set_slot_state(c, b, idle);
                    continue;
                }
                const auto & in = ins[(size_t) b];
                const int L = Ls[(size_t) b];
                c->stats.n_prefix_rows_reused += L;
                if ((int) in.size() - L == 1) {
                    // the prompt is the cached sequence plus one token: a decode step (prefix reuse, see engine_coarse)
                    StepState st1 = fresh_state(); st1.step = step_idx[(size_t) b]; st1.n_past = L; st1.cur_token = in[(size_t) L];
                    set_slot_state(c, b, st1);
                    embed_slot(c, s, b);
                    if (!all_single) enqueue_batch_step(c, s, 1, slot_view(c, s, b));      // mixed window: this slot alone, eagerly
                } else {
                    batch_prefill_and_sample(c, s, b, in, false, step_idx[(size_t) b], L);
                }
            }
            if (all_single && max_here > 0) lock_steps = max_here;       // the first sample of the window is a lock-step too
            for (int j = 0; j < lock_steps; j++) batch_step(c, s, B);
            const std::vector<StepState> st = get_slot_states(c, B);
            for (int b = 0; b < B; b++) {
                if (!here[(size_t) b]) continue;
                std::vector<int32_t> got((size_t) here[(size_t) b]);
                HIP_OK(hipMemcpy(got.data(), bb.out_tokens + (size_t) b * 2048, got.size() * 4, hipMemcpyDeviceToHost));
                coarse_out[(size_t) b].insert(coarse_out[(size_t) b].end(), got.begin(), got.end());
                // rows now in the slot's cache: its prompt and every token fed back (a parked tail of lock steps past `here`
                // wrote further rows, but those are never matched because the ids are not recorded)
                cached[(size_t) b] = ins[(size_t) b];
                cached[(size_t) b].insert(cached[(size_t) b].end(), got.begin(), got.end() - 1);
                step_idx[(size_t) b] += here[(size_t) b];
                c->stats.n_sample_coarse += here[(size_t) b];
                c->stats.n_near_tie += st[(size_t) b].near_tie;
            }
            progress(c, COARSE, 100 * (w + 1) / std::max(1, max_windows));
        }
        for (int b = 0; b < B; b++) {
            if (sampled) slot_rng[(size_t) b].discard(2ull * (unsigned long long) n_steps[(size_t) b]);
            auto & res = c->batch_results[(size_t) b].coarse;
            const auto & out = coarse_out[(size_t) b];
            for (size_t i = 0; i + 1 < out.size(); i += 2) {
                res.push_back(out[i] - p.semantic_vocab_size);
                res.push_back(out[i + 1] - p.semantic_vocab_size - p.codebook_size);
            }
        }
    }
    c->stats.t_coarse_us = now_us() - t;

    // ---- fine + codec, one utterance at a time ------------------------------------------------------------
    int good = 0;
    for (int b = 0; b < B; b++) {
        bark_context::BatchResult & r = c->batch_results[(size_t) b];
        if (r.coarse.empty()) continue;
        t = now_us();
        std::swap(c->rng, slot_rng[(size_t) b]);                        // the fine stage draws from the utterance's generator
        try { r.fine = engine_fine(c, r.coarse); } catch (...) { std::swap(c->rng, slot_rng[(size_t) b]); throw; }
        std::swap(c->rng, slot_rng[(size_t) b]);
        c->stats.t_fine_us += now_us() - t;
        const int T = (int) r.fine.size() / 8;
        std::vector<int32_t> codes((size_t) 8 * T);
        for (int ch = 0; ch < 8; ch++) for (int i = 0; i < T; i++) codes[(size_t) ch * T + i] = r.fine[(size_t) i * 8 + ch];
        t = now_us();
        r.audio = engine_codec_decode(c, codes.data(), 8, T, -1, nullptr);
        c->stats.t_codec_us += now_us() - t;
        c->stats.n_frames += T; c->stats.n_samples += (int32_t) r.audio.size(); c->stats.n_semantic += (int32_t) r.semantic.size();
        r.ok = true; good++;
    }
    c->stats.t_eval_us = now_us() - t0;
    return good;
}

// ---------------------------------------------------------------------------------------------------
// timing hooks for bench.py (hipEvents on the engine's own stream)
// ---------------------------------------------------------------------------------------------------
double engine_time_decode_step(bark_context * c, int which, int ctxlen, int iters, double * bytes_per_step) {
    if (which < 0 || which > 1) throw std::runtime_error("time_decode_step: which must be 0 or 1");
    HIP_OK(hipSetDevice(c->device));
    GptModel & m = c->gpt[which];
    ctxlen = std::max(1, std::min(ctxlen, m.hp.block_size));
    const StageCfg s = stage_cfg(c, which);
    StepState st = fresh_state(); st.n_past = ctxlen - 1; st.cur_token = 1;
    set_state(c, st);
    // the cache rows below ctxlen hold whatever the last run left; timing does not depend on the values,
    // but keep them finite: zero them once
    HIP_OK(hipMemsetAsync(m.kcache, 0, m.kv_layer_stride * m.hp.n_layer * 4, c->stream));
    HIP_OK(hipMemsetAsync(m.vcache, 0, m.kv_layer_stride * m.hp.n_layer * 4, c->stream));
    if (!m.bench_graph) m.bench_graph = capture_decode(c, s, 0);       // n_past does not advance
    for (int i = 0; i < 3; i++) HIP_OK(hipGraphLaunch(m.bench_graph, c->stream));
    set_state(c, st);
    hipEvent_t e0, e1;
    HIP_OK(hipEventCreate(&e0)); HIP_OK(hipEventCreate(&e1));
    HIP_OK(hipEventRecord(e0, c->stream));
    for (int i = 0; i < iters; i++) {
        HIP_OK(hipGraphLaunch(m.bench_graph, c->stream));
        if ((i & 1023) == 1023) set_state(c, st);                        // out_tokens holds 2048 entries
    }
    HIP_OK(hipEventRecord(e1, c->stream));
    HIP_OK(hipEventSynchronize(e1));
    float ms = 0.f;
    HIP_OK(hipEventElapsedTime(&ms, e0, e1));
    (void) hipEventDestroy(e0); (void) hipEventDestroy(e1);
    if (bytes_per_step) {
        const double E = m.hp.n_embd, L = m.hp.n_layer;
        // f16 weights of all layers + evaluated LM-head rows + f32 K and V rows read
        const double wb = weight_bytes_per_element(m);
        *bytes_per_step = L * 12.0 * E * E * wb + (double) s.lm_rows * E * wb + 2.0 * ctxlen * E * L * 4.0;
    }
    return (double) ms * 1000.0 / std::max(1, iters);
}

// One decode GEMV, launched `iters` times back to back while rotating through the layers' weights (so that the
// stream comes from HBM / Infinity Cache like in a real step, not from a hot L2).  op: 0 LN+QKV, 1 proj,
// 2 LN+FC+GELU, 3 mlp proj.  Returns the average device time per launch in microseconds.
double engine_time_gemv(bark_context * c, int which, int op, int iters, double * bytes_per_launch) {
    if (which < 0 || which > 1 || op < 0 || op > 12) throw std::runtime_error("time_gemv: bad arguments");
    const bool attn = op >= 8;            // 8: attn_scores_kernel, 9: attn_mix_kernel, 10: both (context = n_past + 1 = 641)
    const int attn_op = op;
    const bool hot = op >= 4 && !attn;
    op &= 3;
    HIP_OK(hipSetDevice(c->device));
    GptModel & m = c->gpt[which];
    const int E = m.hp.n_embd, P = c->P;
    StepState st = fresh_state(); st.n_past = attn ? 640 : 100; st.cur_token = 1;
    set_state(c, st);
    if (attn) {
        HIP_OK(hipMemsetAsync(m.kcache, 0, m.kv_layer_stride * m.hp.n_layer * 4, c->stream));
        HIP_OK(hipMemsetAsync(m.vcache, 0, m.kv_layer_stride * m.hp.n_layer * 4, c->stream));
        HIP_OK(hipMemsetAsync(c->q, 0, (size_t) E * 4, c->stream));
    }
    HIP_OK(hipMemsetAsync(c->x, 0, (size_t) E * 4, c->stream));
    HIP_OK(hipMemsetAsync(c->att, 0, (size_t) E * 2, c->stream));
    HIP_OK(hipMemsetAsync(c->hbuf, 0, (size_t) 4 * E * 2, c->stream));
    if (m.q4) { HIP_OK(hipMemsetAsync(c->att32, 0, (size_t) E * 4, c->stream)); HIP_OK(hipMemsetAsync(c->h32, 0, (size_t) 4 * E * 4, c->stream)); }
    auto launch = [&](int l) {
        const GptModel::Layer & L = m.layers[(size_t) l];
        if (attn) {
            AttnDecodeArgs at;
            at.q = c->q; at.kc = layer_k(m, l); at.vc = layer_v(m, l); at.H = m.hp.n_head; at.P = P; at.st = c->d_state; at.att = c->att;
            at.scores = c->scores; at.hmax = c->d_hmax;
            launch_attn_decode_part(c->stream, at, attn_op == 8 ? 1 : attn_op == 9 ? 2 : attn_op == 10 ? 3 : attn_op == 11 ? 4 : 5);
            return;
        }
        LinArgs a;
        a.N = 1;
        switch (op) {
            case 0: a.W = L.attn_w; a.wq = L.attn_q; a.M = 3 * E; a.K = E; a.x_f32 = c->x; a.ln_g = L.ln1_g; a.ln_b = L.ln1_b; a.bias = L.attn_b; a.epi = EPI_QKV;
                    a.q = c->q; a.kc = layer_k(m, l); a.vc = layer_v(m, l); a.E = E; a.P = P; a.st = c->d_state; break;
            case 1: a.W = L.proj_w; a.wq = L.proj_q; a.M = E; a.K = E; if (m.q4) a.x_f32 = c->att32; else a.x_f16 = c->att; a.bias = L.proj_b; a.epi = EPI_RESID; a.res = c->x; break;
            case 2: a.W = L.fc_w; a.wq = L.fc_q; a.M = 4 * E; a.K = E; a.x_f32 = c->x; a.ln_g = L.ln2_g; a.ln_b = L.ln2_b; a.bias = L.fc_b; a.epi = EPI_GELU;
                    a.out_h = c->hbuf; a.out_h32 = m.q4 ? c->h32 : nullptr; a.lut = c->d_gelu_lut; break;
            default: a.W = L.mproj_w; a.wq = L.mproj_q; a.M = E; a.K = 4 * E; if (m.q4) a.x_f32 = c->h32; else a.x_f16 = c->hbuf; a.bias = L.mproj_b; a.epi = EPI_RESID; a.res = c->x; break;
        }
        launch_linear(c->stream, a);
    };
    // op >= 4 ("hot"): always layer 0, so the weights stay in L2; otherwise rotate through the layers.
    // The launches are captured into one hipGraph (48 nodes) so that the host launch rate does not bound the result.
    const int per_graph = 48;
    hipGraph_t graph = nullptr; hipGraphExec_t exec = nullptr;
    HIP_OK(hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < per_graph; i++) launch(hot ? 0 : i % m.hp.n_layer);
    HIP_OK(hipStreamEndCapture(c->stream, &graph));
    HIP_OK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
    (void) hipGraphDestroy(graph);
    HIP_OK(hipGraphLaunch(exec, c->stream));
    const int reps = std::max(1, iters / per_graph);
    hipEvent_t e0, e1;
    HIP_OK(hipEventCreate(&e0)); HIP_OK(hipEventCreate(&e1));
    HIP_OK(hipEventRecord(e0, c->stream));
    for (int i = 0; i < reps; i++) HIP_OK(hipGraphLaunch(exec, c->stream));
    HIP_OK(hipEventRecord(e1, c->stream));
    HIP_OK(hipEventSynchronize(e1));
    float ms = 0.f;
    HIP_OK(hipEventElapsedTime(&ms, e0, e1));
    (void) hipEventDestroy(e0); (void) hipEventDestroy(e1);
    (void) hipGraphExecDestroy(exec);
    iters = reps * per_graph;
    if (bytes_per_launch) {
        const double Ed = E;
        const double w = op == 0 ? 3 * Ed * Ed : op == 1 ? Ed * Ed : 4 * Ed * Ed;
        *bytes_per_launch = w * weight_bytes_per_element(m);          // the weight matrix; vectors are < 1 % of it
    }
    return (double) ms * 1000.0 / std::max(1, iters);
}

double engine_time_fine_pass(bark_context * c, int iters, double * flops_per_pass) {
    HIP_OK(hipSetDevice(c->device));
    GptModel & m = c->gpt[2];
    std::vector<int32_t> buf((size_t) 8 * 1024);
    for (size_t i = 0; i < buf.size(); i++) buf[i] = (int32_t) ((i * 2654435761u) >> 22) & 1023;
    upload_tokens(c, buf.data(), buf.size());
    run_fine_forward(c, 4, 1024);
    HIP_OK(hipStreamSynchronize(c->stream));
    hipEvent_t e0, e1;
    HIP_OK(hipEventCreate(&e0)); HIP_OK(hipEventCreate(&e1));
    HIP_OK(hipEventRecord(e0, c->stream));
    for (int i = 0; i < iters; i++) run_fine_forward(c, 2 + i % 6, 1024);
    HIP_OK(hipEventRecord(e1, c->stream));
    HIP_OK(hipEventSynchronize(e1));
    float ms = 0.f;
    HIP_OK(hipEventElapsedTime(&ms, e0, e1));
    (void) hipEventDestroy(e0); (void) hipEventDestroy(e1);
    if (flops_per_pass) {
        const double E = m.hp.n_embd, L = m.hp.n_layer, N = 1024;
        *flops_per_pass = 2.0 * N * (L * 12.0 * E * E + 1024.0 * E) + 4.0 * N * N * E * L;
    }
    return (double) ms * 1000.0 / std::max(1, iters);
}

}  // namespace barkhip
