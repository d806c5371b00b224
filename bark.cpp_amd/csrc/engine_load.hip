// engine_load.hip - model upload and per-context runtime state of the MI355X Bark engine (bark_load_model_from_file,
// /root/reference/bark.cpp:1080-1163): the container is parsed on the host, every tensor of the hot path goes into one device
// slab (block-quantised matrices into per-field arrays), KV caches and activation scratch are allocated per context.
#include "engine_internal.h"

#include <algorithm>
#include <climits>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <stdexcept>

using namespace barkhip;
using namespace barkhip::detail;

namespace {

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

// ne0 / ne1 > 0: the tensor must have exactly that rank-1 / rank-2 shape (every further dim 1; bark.cpp:1034 compares ne[0], ne[1]);
// ne0 == 0: any shape (the caller checks it)
const TensorRef & need(const std::map<std::string, TensorRef> & m, const std::string & name, int ttype, int64_t ne0, int64_t ne1) {
    auto it = m.find(name);
    if (it == m.end()) throw std::runtime_error("missing tensor '" + name + "'");
    const TensorRef & t = it->second;
    if (ne0 > 0) {
        const int rank = ne1 > 0 ? 2 : 1;
        bool ok = t.n_dims >= 1 && t.n_dims <= 4 && t.ne[0] == ne0 && (rank == 1 || t.ne[1] == ne1);
        for (int i = rank; i < 4; i++) ok = ok && t.ne[i] == 1;
        if (!ok) throw std::runtime_error("tensor '" + name + "' has an unexpected shape");
    }
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
const TensorRef * maybe(const std::map<std::string, TensorRef> & m, const std::string & name, int ttype, int64_t ne0) {
    auto it = m.find(name);
    if (it == m.end()) return nullptr;
    if (it->second.ne[0] != ne0 || it->second.nelements() != ne0 || it->second.ttype != ttype) throw std::runtime_error("tensor '" + name + "' has an unexpected shape/type");
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
    delete tail;
    (void) hipSetDevice(device);
    try { (void) barkhip::detail::guard_check(this, "bark_free"); } catch (...) { }
    for (auto & g : gpt) {
        for (auto & e : g.decode_graph) if (e) (void) hipGraphExecDestroy(e);
        for (auto & e : g.decode_graph8) if (e) (void) hipGraphExecDestroy(e);
        if (g.bench_graph) (void) hipGraphExecDestroy(g.bench_graph);
    }
    for (auto & g : batch_graphs) if (g.second) (void) hipGraphExecDestroy(g.second);
    if (lstm_graph.exec) (void) hipGraphExecDestroy(lstm_graph.exec);
    if (codec_graph.exec) (void) hipGraphExecDestroy(codec_graph.exec);
    for (auto & g : fine_graphs) if (g) (void) hipGraphExecDestroy(g);
    for (void * p : allocs) (void) hipFree(p);
    if (batch.h_ids) (void) hipHostFree(batch.h_ids);
    if (batch.h_state) (void) hipHostFree(batch.h_state);
    if (batch.h_state_in) (void) hipHostFree(batch.h_state_in);
    if (stream) (void) hipStreamDestroy(stream);
}
bark_context::SharedWeights::~SharedWeights() {
    (void) hipSetDevice(device);
    if (slab) (void) hipFree(slab);
    if (codebooks) (void) hipFree(codebooks);
    for (void * p : extra) (void) hipFree(p);
}

namespace barkhip {

namespace detail {
int guard_check(bark_context * ctx, const char * where) {
    if (ctx->guarded.empty() || !ctx->stream) return 0;
    int bad = 0;
    std::vector<unsigned char> h(kGuardBytes);
    HIP_OK(hipStreamSynchronize(ctx->stream));
    for (size_t i = 0; i < ctx->guarded.size(); i++) {
        for (int side = 0; side < 2; side++) {
            const char * g = (const char *) ctx->guarded[i].first + (side ? kGuardBytes + ctx->guarded[i].second : 0);
            HIP_OK(hipMemcpyAsync(h.data(), g, kGuardBytes, hipMemcpyDeviceToHost, ctx->stream));
            HIP_OK(hipStreamSynchronize(ctx->stream));
            size_t first = kGuardBytes, n = 0;
            for (size_t k = 0; k < kGuardBytes; k++) if (h[k] != 0xA5) { if (first == kGuardBytes) first = k; n++; }
            if (n) {
                bad++;
                fprintf(stderr, "bark-hip GUARD DAMAGED (%s, ctx %p): allocation %zu (%zu bytes), %s band: %zu bytes overwritten, first at offset %zu (value 0x%02x)\n",
                        where, (void *) ctx, i, ctx->guarded[i].second, side ? "upper" : "lower", n, first, h[first]);
                HIP_OK(hipMemsetAsync((void *) g, 0xA5, kGuardBytes, ctx->stream));          // report once
            }
        }
    }
    HIP_OK(hipStreamSynchronize(ctx->stream));
    return bad;
}
}  // namespace detail

void engine_invalidate_graphs(bark_context * ctx) {
    if (ctx->tail) engine_invalidate_graphs(ctx->tail);       // the clone that runs the tail of lock-step jobs replays graphs with the same constants
    for (auto & g : ctx->batch_graphs) if (g.second) (void) hipGraphExecDestroy(g.second);
    ctx->batch_graphs.clear();
    for (auto & g : ctx->fine_graphs) if (g) { (void) hipGraphExecDestroy(g); g = nullptr; }
    for (auto & g : ctx->gpt) {
        for (auto & e : g.decode_graph) if (e) { (void) hipGraphExecDestroy(e); e = nullptr; }
        for (auto & e : g.decode_graph8) if (e) { (void) hipGraphExecDestroy(e); e = nullptr; }
        if (g.bench_graph) { (void) hipGraphExecDestroy(g.bench_graph); g.bench_graph = nullptr; }
    }
}

// per-context mutable state: stream, KV caches, activation scratch, GELU table
static void init_runtime(bark_context * ctxp, bool weights_uploaded_now) {
    struct Holder { bark_context * p; bark_context * get() const { return p; } bark_context * operator->() const { return p; } } ctx{ctxp};
    // ---- KV caches, scratch ------------------------------------------------------------------------
    const int P = ctx->P;
    for (int g = 0; g < 2; g++) {
        GptModel & m = ctx->gpt[g];
        m.kv_layer_stride = (size_t) m.hp.n_embd * P;
        m.kcache = dev_alloc<float>(ctx.get(), m.kv_layer_stride * m.hp.n_layer);     // bark.cpp:976-991
        m.vcache = dev_alloc<float>(ctx.get(), m.kv_layer_stride * m.hp.n_layer);
        m.vtcache = dev_alloc<float>(ctx.get(), m.kv_layer_stride * m.hp.n_layer);
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
    if (ctx->fast_gemm) {
        ctx->q16 = dev_alloc<half_t>(ctx.get(), NE);
        ctx->k16 = dev_alloc<half_t>(ctx.get(), NE);
        ctx->vt16 = dev_alloc<half_t>(ctx.get(), NE);
    }
    ctx->ps = dev_alloc<float>(ctx.get(), (size_t) ctx->max_H * P * 4);
    ctx->knew = dev_alloc<float>(ctx.get(), (size_t) ctx->max_E);
    HIP_OK(hipMemsetAsync(ctx->ps, 0, (size_t) ctx->max_H * P * 4 * sizeof(float), ctx->stream));
    if (ctx->any_q4) {
        ctx->att32 = dev_alloc<float>(ctx.get(), NE);
        ctx->h32 = dev_alloc<float>(ctx.get(), NE * 4);
        const size_t nT = (size_t) (4 * ctx->max_E / 32) * 1024;
        ctx->xq.q = dev_alloc<int8_t>(ctx.get(), NE * 4);
        ctx->xq.d = dev_alloc<float>(ctx.get(), NE * 4 / 32);
        ctx->xq.s = dev_alloc<float>(ctx.get(), NE * 4 / 32);
        ctx->xq.dT = dev_alloc<float>(ctx.get(), nT);
        ctx->xq.sT = dev_alloc<float>(ctx.get(), nT);
        HIP_OK(hipMemsetAsync(ctx->xq.dT, 0, nT * sizeof(float), ctx->stream));
        HIP_OK(hipMemsetAsync(ctx->xq.sT, 0, nT * sizeof(float), ctx->stream));
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
    {
        std::vector<uint16_t> lut(65536);
        for (uint32_t i = 0; i < 65536; i++) {
            const uint16_t bits = (uint16_t) i;
            const _Float16 h = __builtin_bit_cast(_Float16, bits);
            const _Float16 r = (_Float16) gelu_tanh_host((float) h);
            lut[i] = __builtin_bit_cast(uint16_t, r);
        }
        ctx->d_gelu_lut = dev_alloc<uint16_t>(ctx.get(), 65536);
        HIP_OK(hipMemcpyAsync(ctx->d_gelu_lut, lut.data(), 65536 * 2, hipMemcpyHostToDevice, ctx->stream));
        HIP_OK(hipStreamSynchronize(ctx->stream));               // `lut` leaves scope
    }
    // everything the load put on the legacy stream (weight uploads) or on this one is in place before the first kernel of the non-blocking stream;
    // a clone shares weights that are long in place and must not stall the other streams of a running server with a device-wide synchronisation
    HIP_OK(hipStreamSynchronize(ctx->stream));
    if (weights_uploaded_now) HIP_OK(hipDeviceSynchronize());
}

// bark_load_model_from_file (bark.cpp:1080-1163): parse the container, upload every tensor of the hot path.
bark_context * engine_load(const char * path, const bark_context_params & params, uint32_t seed, int device) {
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0)
        throw std::runtime_error("no HIP device available (this engine has no CPU path)");
    std::unique_ptr<bark_context> ctx(new bark_context());
    ctx->params = params;
    ctx->rng = std::mt19937(seed);                       // bark.cpp:1179
    // device: the caller's (bark_hip_load_model_on_device: one process, several GPUs), else BARK_HIP_DEVICE (one process per GPU), else the current one
    if (device >= 0) ctx->device = device;
    else if (const char * e = getenv("BARK_HIP_DEVICE")) ctx->device = atoi(e);
    else (void) hipGetDevice(&ctx->device);
    if (ctx->device < 0 || ctx->device >= n_dev) throw std::runtime_error("HIP device ordinal out of range (BARK_HIP_DEVICE / bark_hip_load_model_on_device)");
    HIP_OK(hipSetDevice(ctx->device));
    if (const char * e = getenv("BARK_HIP_GRAPH")) ctx->use_graph = atoi(e) != 0;
    if (const char * e = getenv("BARK_HIP_FAST_GEMM")) {
        // "1" selects the tolerance route; anything else but "0" / "" is refused instead of silently meaning something (it once selected an
        // instruction-order variant of the canonical route)
        if (!strcmp(e, "1")) ctx->fast_gemm = 1;
        else if (strcmp(e, "0") && *e) throw std::runtime_error("BARK_HIP_FAST_GEMM accepts 0 or 1");
    }
    if (const char * e = getenv("BARK_HIP_FINE_ORDER")) {
        // order of the fine model's products on f16 files (bark_context::fine_order): "c1" / "c1m" everywhere; unset or "" = the default policy
        if (!strcmp(e, "c1")) ctx->fine_order = 1;
        else if (!strcmp(e, "c1m")) ctx->fine_order = 2;
        else if (*e) throw std::runtime_error("BARK_HIP_FINE_ORDER accepts c1 or c1m");
    }
    // non-blocking: the legacy stream neither waits for this one nor is refused while it captures a graph - contexts (clones) of one process run
    // from several host threads, and a plain hipMemcpy of one thread must not collide with a capture of another (tools/staggered_jobs.py)
    HIP_OK(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
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
        // the decode GEMV is instantiated for K / 128 in {1, 2, 4, 6, 8} (and 4x those): kernels.hip
        if (E != 128 && E != 256 && E != 512 && E != 768 && E != 1024) throw std::runtime_error("n_embd must be one of 128, 256, 512, 768, 1024");
        // the sampling kernels hold up to 12288 logits in registers (misc_kernels.hip)
        if (m.hp.n_in_vocab <= 0 || m.hp.n_out_vocab <= 0 || m.hp.n_out_vocab > 12288) throw std::runtime_error("vocabulary sizes must be in 1..12288 (output) and positive (input)");
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
    // the fine stage predicts codebook nn = 2..7 with lm_heads[nn - 1] (n_codes_given = 1, bark.cpp:1573) over logits [0, 1024)
    if (ctx->gpt[2].hp.n_wtes != 8 || ctx->gpt[2].hp.n_lm_heads < 7) throw std::runtime_error("fine model must have 8 embeddings and >= 7 heads");
    if (ctx->gpt[2].hp.n_out_vocab < 1024 || ctx->gpt[2].hp.n_in_vocab < 1025) throw std::runtime_error("fine model vocabulary is smaller than a codebook");

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
            if (w.n_dims != 3 || w.ne[3] != 1 || w.ne[0] > 64 || w.ne[1] > 4096 || w.ne[2] > 4096) throw std::runtime_error("codec conv weight '" + p + "' has an unexpected shape");
            cv.k = (int) w.ne[0]; cv.cin = (int) w.ne[1]; cv.cout = (int) w.ne[2];
            place(w, (const void **) &cv.w);
            const TensorRef & b = need(T, p + ".bias", 0, 0, 0);
            if (b.nelements() != cv.cout) throw std::runtime_error("codec bias size mismatch at " + p);
            place(b, (const void **) &cv.b);
        };
        auto convt = [&](const std::string & p, CodecModel::ConvT & cv, int stride) {
            const TensorRef & w = codec_weight(p + ".weight", 0, 0);
            if (w.n_dims != 3 || w.ne[3] != 1 || w.ne[0] < stride || w.ne[0] > 64 || w.ne[1] > 4096 || w.ne[2] > 4096) throw std::runtime_error("codec transposed-conv weight '" + p + "' has an unexpected shape");
            if (w.ne[0] != 2 * stride) throw std::runtime_error("codec transposed conv '" + p + "': kernel size must be twice the stride (EnCodec's)");
            cv.k = (int) w.ne[0]; cv.cout = (int) w.ne[1]; cv.cin = (int) w.ne[2]; cv.stride = stride;
            place(w, (const void **) &cv.w);
            const TensorRef & b = need(T, p + ".bias", 0, 0, 0);
            if (b.nelements() != cv.cout) throw std::runtime_error("codec bias size mismatch at " + p);
            place(b, (const void **) &cv.b);
        };
        if (cm.hp.hidden_dim <= 0 || cm.hp.hidden_dim > 4096 || cm.hp.n_bins <= 0 || cm.hp.n_bins > (1 << 20)) throw std::runtime_error("implausible codec hparams");
        conv("decoder.model.0.conv.conv", cm.init);
        if (cm.init.cin != cm.hp.hidden_dim) throw std::runtime_error("codec: the first conv does not take hidden_dim channels");
        cm.D = cm.init.cout;
        if (cm.D != 128 && cm.D != 256 && cm.D != 512 && cm.D != 1024) throw std::runtime_error("codec LSTM width must be 128, 256, 512 or 1024");
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
        // channel continuity of the decoder stack (modeling_encodec.py:316-347)
        int ch = cm.D;
        bool chain_ok = true;
        for (int i = 0; i < 4; i++) {
            const CodecModel::Block & b = cm.blocks[i];
            chain_ok = chain_ok && b.up.cin == ch && b.c1.cin == b.up.cout && b.c2.cin == b.c1.cout && b.c2.cout == b.up.cout &&
                       b.sc.cin == b.up.cout && b.sc.cout == b.up.cout;
            ch = b.up.cout;
        }
        chain_ok = chain_ok && cm.fin.cin == ch && cm.fin.cout == 1;
        if (!chain_ok) throw std::runtime_error("codec: decoder convolutions do not chain (channel counts)");
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
        // kernel images for the f16 matrix cores (order C9m): rows = output channels padded to 32, columns kd padded to 16 with zeros
        auto upload_h = [&](const std::vector<uint16_t> & h) -> const half_t * {
            void * d = nullptr;
            HIP_OK(hipMalloc(&d, h.size() * 2));
            ctx->weights->extra.push_back(d);
            HIP_OK(hipMemcpy(d, h.data(), h.size() * 2, hipMemcpyHostToDevice));
            return (const half_t *) d;
        };
        auto conv_image = [&](const std::string & p, CodecModel::Conv & cv) {
            if (cv.cin & 7) return;
            const TensorRef & t = codec_weight(p + ".weight", 0, 0);                 // [cout][cin][k] f16
            const int kd = cv.k * cv.cin, kd16 = (kd + 15) & ~15, c32 = (cv.cout + 31) & ~31;
            std::vector<uint16_t> img((size_t) c32 * kd16, 0);
            for (int co = 0; co < cv.cout; co++) for (int ci = 0; ci < cv.cin; ci++) for (int k = 0; k < cv.k; k++)
                memcpy(&img[(size_t) co * kd16 + (size_t) k * cv.cin + ci], t.data + 2 * (((size_t) co * cv.cin + ci) * cv.k + k), 2);
            cv.wm = upload_h(img);
        };
        auto convt_image = [&](const std::string & p, CodecModel::ConvT & cv) {
            if (cv.cin & 7) return;
            const TensorRef & t = codec_weight(p + ".weight", 0, 0);                 // [cin][cout][k] f16, k = 2 stride
            const int s = cv.stride, kd = 2 * cv.cin, kd16 = (kd + 15) & ~15, c32 = (cv.cout + 31) & ~31;
            std::vector<uint16_t> img((size_t) s * c32 * kd16, 0);
            for (int r = 0; r < s; r++) for (int co = 0; co < cv.cout; co++) for (int ci = 0; ci < cv.cin; ci++) {
                const uint8_t * w = t.data + 2 * (((size_t) ci * cv.cout + co) * cv.k);
                memcpy(&img[((size_t) r * c32 + co) * kd16 + ci], w + 2 * (r + s), 2);              // tap 0: the previous frame, kernel element r + s
                memcpy(&img[((size_t) r * c32 + co) * kd16 + cv.cin + ci], w + 2 * r, 2);           // tap 1: this frame, element r
            }
            cv.wm = upload_h(img);
        };
        conv_image("decoder.model.0.conv.conv", cm.init);
        for (int i = 0; i < 4; i++) {
            const int idx = 3 + 3 * i;
            convt_image("decoder.model." + std::to_string(idx) + ".convtr.convtr", cm.blocks[i].up);
            conv_image("decoder.model." + std::to_string(idx + 1) + ".block.1.conv.conv", cm.blocks[i].c1);
            conv_image("decoder.model." + std::to_string(idx + 1) + ".block.3.conv.conv", cm.blocks[i].c2);
            conv_image("decoder.model." + std::to_string(idx + 1) + ".shortcut.conv.conv", cm.blocks[i].sc);
        }
        conv_image("decoder.model.15.conv.conv", cm.fin);
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

    init_runtime(ctx.get(), true);
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
        ctx->gpt[g].kcache = ctx->gpt[g].vcache = ctx->gpt[g].vtcache = nullptr;
        for (auto & e : ctx->gpt[g].decode_graph) e = nullptr;
        for (auto & e : ctx->gpt[g].decode_graph8) e = nullptr;
        ctx->gpt[g].bench_graph = nullptr;
    }
    ctx->codec = src->codec;
    ctx->device = src->device; ctx->use_graph = src->use_graph; ctx->fast_gemm = src->fast_gemm; ctx->fine_order = src->fine_order;
    ctx->weights = src->weights; ctx->weight_bytes = src->weight_bytes;
    ctx->max_E = src->max_E; ctx->max_H = src->max_H; ctx->P = src->P; ctx->any_q4 = src->any_q4; ctx->any_w32 = src->any_w32;
    HIP_OK(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
    init_runtime(ctx.get(), false);
    ctx->description = src->description + " (clone)";
    return ctx.release();
}

}  // namespace barkhip
