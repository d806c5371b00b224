// engine.hip - building blocks, stage loops and orchestration of the MI355X Bark engine.
// Control flow restates /root/reference/bark.cpp (cited per function); all tensor math runs in the HIP kernels of
// kernels.hip / quant_kernels.hip / attention_kernels.hip / misc_kernels.hip / codec_kernels.hip.  No CPU fallback exists:
// without a HIP device bark_load_model fails (engine_load.hip).  Lock-step batching: engine_batch.hip; bench hooks:
// engine_timing.hip.
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

namespace barkhip {

// ---------------------------------------------------------------------------------------------------
// GPT building blocks
// ---------------------------------------------------------------------------------------------------
namespace detail {

// bytes per weight of the model's matrices: f16 2, f32 4, block formats block_bytes / 32
double weight_bytes_per_element(const GptModel & m) {
    if (m.w32) return 4.0;
    if (m.q4) return quant_formats()[m.layers[0].attn_q.qt].block_bytes / 32.0;
    return 2.0;
}
float * layer_k(const GptModel & m, int l) { return m.kcache + m.kv_layer_stride * (size_t) l; }
float * layer_v(const GptModel & m, int l) { return m.vcache + m.kv_layer_stride * (size_t) l; }
float * layer_vt(const GptModel & m, int l) { return m.vtcache ? m.vtcache + m.kv_layer_stride * (size_t) l : nullptr; }

// The fine model's products of an f16 file: C1 chains on the f32 matrix cores (gemm_kernel; the restated reference order, the oracle's default) or C1m on
// the f16 matrix cores (gemm_f16_tile_kernel; the oracle follows with set_fine_mfma(True)) - bark_context::fine_order.  Default policy: C1 for
// bark_generate_audio and the stage entry points (config 2 is bit-exact to the restated reference end to end), C1m inside lock-step jobs, whose fine
// stage is a third of the wall clock under C1.  BARK_HIP_CROSSCHECK bit 8 (256) = C1 everywhere (as in rounds 4 - 5).
bool fine_products_on_f16_mfma(const bark_context * c, const GptModel & m, bool causal) {
    if (causal || &m != &c->gpt[2] || m.q4 || m.w32 || (m.hp.n_embd & 63) != 0 || (crosscheck_mask() & 256)) return false;
    return c->fine_order == 2 || (c->fine_order == 0 && c->in_job);
}

RowBufs own_rows(bark_context * c) {
    RowBufs r; r.x = c->x; r.q = c->q; r.xn = c->xn; r.att = c->att; r.hbuf = c->hbuf; r.q16 = c->q16; r.k16 = c->k16; r.vt16 = c->vt16;
    r.logits = c->logits; r.tokens = c->d_tokens; r.plane = 1024;
    return r;
}

// N > 1 rows through all layers (bark.cpp:1261-1389 causal, :1474-1562 fine); x holds the embeddings.
void run_layers_rows(bark_context * c, GptModel & m, int N, bool causal, float * kbase, float * vbase, int pos0, const RowBufs * rbp, int seq, size_t kv_seq_stride,
                     const SeqTab * seqtab) {
    const int E = m.hp.n_embd, H = m.hp.n_head, P = c->P;
    hipStream_t s = c->stream;
    const RowBufs own = own_rows(c);
    const RowBufs & rb = rbp ? *rbp : own;
    if (rbp && (m.q4 || m.w32)) throw std::runtime_error("row scratch other than the context's own: f16 model files only");
    // fine model, f16 file: the products run on the f16 matrix cores, whose accumulation IS the canonical order of that model (C1m: stated in
    // oracle/mfma_f16_emu.h from device probes; the fine model never shares a row with a decode step, so the other models keep C1)
    const int fast = fine_products_on_f16_mfma(c, m, causal) ? 1 : (!m.q4 && !m.w32) ? c->fast_gemm : 0;
    const int Z = seq > 0 ? N / seq : 1;
    const bool flash = c->fast_gemm && fast && !causal && pos0 == 0 && N % 1024 == 0 && (seq == 0 || seq == 1024) && (rbp || (!kbase && !vbase)) && rb.q16 && E % 64 == 0;
    // kbase / vbase: another utterance slot's cache (batched decode) or the fine batch's; default: the context's own cache
    auto layer_k = [&](const GptModel & mm, int l) { return (kbase ? kbase : mm.kcache) + mm.kv_layer_stride * (size_t) l; };
    auto layer_v = [&](const GptModel & mm, int l) { return (vbase ? vbase : mm.vcache) + mm.kv_layer_stride * (size_t) l; };
    for (int l = 0; l < m.hp.n_layer; l++) {
        const GptModel::Layer & L = m.layers[(size_t) l];
        // f16 weights: activations are rounded to f16 rows (xn / att / hbuf); q4_0 weights: f32 rows quantised to q8_0 (xq8 / xd8)
        if (m.w32)     launch_ln_rows_f32(s, rb.x, N, E, L.ln1_g, L.ln1_b, c->xn32);
        else if (m.q4) launch_q8_rows(s, rb.x, N, E, L.ln1_g, L.ln1_b, c->xq);
        else           launch_ln_rows(s, rb.x, N, E, L.ln1_g, L.ln1_b, rb.xn);
        LinArgs a;
        a.W = L.attn_w; a.wq = L.attn_q; a.M = 3 * E; a.K = E; a.N = N; a.x_f16 = rb.xn; a.xq = c->xq; if (m.w32) a.x_f32 = c->xn32; a.bias = L.attn_b; a.epi = EPI_QKV;
        a.q = rb.q; a.kc = layer_k(m, l); a.vc = layer_v(m, l); a.E = E; a.P = P; a.pos0 = pos0;
        if (!kbase && !vbase) a.vt = detail::layer_vt(m, l);      // the context's own cache keeps the K-layout copy of V too
        a.seq = seq; a.kv_slot_stride = kv_seq_stride; a.seqtab = seqtab;
        a.fast = fast;
        if (flash) {
            // tolerance route of the fine model: q / k / v leave the product as the f16 operands of the flash attention (no KV cache)
            a.epi = EPI_QKV16; a.q16 = rb.q16; a.k16 = rb.k16; a.vt16 = rb.vt16; a.seq = 1024;
            launch_linear(s, a);
            AttnFlashArgs fa;
            fa.q16 = rb.q16; fa.k16 = rb.k16; fa.vt16 = rb.vt16; fa.H = H; fa.E = E; fa.S = 1024; fa.Z = N / 1024; fa.att = rb.att; fa.ld_att = E;
            launch_attn_flash(s, fa);
        } else {
            launch_linear(s, a);
            AttnPrefillArgs at;
            at.q = rb.q; at.ldq = E; at.kc = layer_k(m, l); at.vc = layer_v(m, l); at.H = H; at.P = P; at.N = seq > 0 ? seq : N; at.n_past = pos0;
            at.causal = causal ? 1 : 0; at.att = rb.att; at.ld_att = E; at.att32 = m.q4 ? c->att32 : nullptr;
            at.Z = Z; at.kv_seq_stride = kv_seq_stride; at.seqtab = seqtab;
            launch_attn_prefill(s, at);
        }
        if (m.q4 && !m.w32) launch_q8_rows(s, c->att32, N, E, nullptr, nullptr, c->xq);
        LinArgs p;
        p.W = L.proj_w; p.wq = L.proj_q; p.M = E; p.K = E; p.N = N; p.x_f16 = rb.att; p.xq = c->xq; if (m.w32) p.x_f32 = c->att32; p.bias = L.proj_b; p.epi = EPI_RESID; p.res = rb.x;
        p.fast = fast;
        launch_linear(s, p);
        if (m.w32)     launch_ln_rows_f32(s, rb.x, N, E, L.ln2_g, L.ln2_b, c->xn32);
        else if (m.q4) launch_q8_rows(s, rb.x, N, E, L.ln2_g, L.ln2_b, c->xq);
        else           launch_ln_rows(s, rb.x, N, E, L.ln2_g, L.ln2_b, rb.xn);
        LinArgs f;
        f.W = L.fc_w; f.wq = L.fc_q; f.M = 4 * E; f.K = E; f.N = N; f.x_f16 = rb.xn; f.xq = c->xq; if (m.w32) f.x_f32 = c->xn32; f.bias = L.fc_b; f.epi = EPI_GELU;
        f.out_h = rb.hbuf; f.out_h32 = m.q4 ? c->h32 : nullptr; f.lut = c->d_gelu_lut;
        f.fast = fast;
        launch_linear(s, f);
        if (m.q4 && !m.w32) launch_q8_rows(s, c->h32, N, 4 * E, nullptr, nullptr, c->xq);
        LinArgs o;
        o.W = L.mproj_w; o.wq = L.mproj_q; o.M = E; o.K = 4 * E; o.N = N; o.x_f16 = rb.hbuf; o.xq = c->xq; if (m.w32) o.x_f32 = c->h32; o.bias = L.mproj_b; o.epi = EPI_RESID; o.res = rb.x;
        o.fast = fast;
        launch_linear(s, o);
    }
}

// one token through all layers; position / token come from the device-resident StepState
void run_layers_decode(bark_context * c, GptModel & m) {
    const int E = m.hp.n_embd, H = m.hp.n_head, P = c->P;
    hipStream_t s = c->stream;
    for (int l = 0; l < m.hp.n_layer; l++) {
        const GptModel::Layer & L = m.layers[(size_t) l];
        const bool ps = !(crosscheck_mask() & 4) && !m.w32 && P == 1024 && m.vtcache && E <= 1024;
        LinArgs a;
        a.W = L.attn_w; a.wq = L.attn_q; a.M = 3 * E; a.K = E; a.N = 1; a.x_f32 = c->x; a.ln_g = L.ln1_g; a.ln_b = L.ln1_b; a.bias = L.attn_b;
        a.epi = EPI_QKV; a.q = c->q; a.kc = layer_k(m, l); a.vc = layer_v(m, l); a.vt = layer_vt(m, l); a.E = E; a.P = P; a.pos0 = 0; a.st = c->d_state;
        // f16 weights: the QKV kernel also forms the partial scores of the cached keys (C2 blocks), attn_ps_kernel finishes them
        if (ps) { a.ps = c->ps; a.knew = c->knew; a.ng = c->decode_ng; }
        BARK_TRACE_SET(c, a, (a.M + 3) / 4 + 4 * (E / 4));        // room for all four copies of the q workgroups
        launch_linear(s, a);
        AttnDecodeArgs at;
        at.q = c->q; at.kc = layer_k(m, l); at.vc = layer_v(m, l); at.H = H; at.P = P; at.st = c->d_state; at.att = c->att;
        at.att32 = m.q4 ? c->att32 : nullptr;
        if (ps) { at.ps = c->ps; at.knew = c->knew; at.ng = c->decode_ng; at.vt = layer_vt(m, l); }
        BARK_TRACE_SET(c, at, 8 * 16 * ((H + 7) / 8) * 16);      // up to 16 slices per head, 16 waves per workgroup
        launch_attn_decode(s, at);
        LinArgs p;
        p.W = L.proj_w; p.wq = L.proj_q; p.M = E; p.K = E; p.N = 1; if (m.q4) p.x_f32 = c->att32; else p.x_f16 = c->att; p.bias = L.proj_b; p.epi = EPI_RESID; p.res = c->x;
        BARK_TRACE_SET(c, p, (p.M + 3) / 4);
        launch_linear(s, p);
        LinArgs f;
        f.W = L.fc_w; f.wq = L.fc_q; f.M = 4 * E; f.K = E; f.N = 1; f.x_f32 = c->x; f.ln_g = L.ln2_g; f.ln_b = L.ln2_b; f.bias = L.fc_b;
        f.epi = EPI_GELU; f.out_h = c->hbuf; f.out_h32 = m.q4 ? c->h32 : nullptr; f.lut = c->d_gelu_lut;
        BARK_TRACE_SET(c, f, (f.M + 3) / 4);
        launch_linear(s, f);
        LinArgs o;
        o.W = L.mproj_w; o.wq = L.mproj_q; o.M = E; o.K = 4 * E; o.N = 1; if (m.q4) o.x_f32 = c->h32; else o.x_f16 = c->hbuf; o.bias = L.mproj_b; o.epi = EPI_RESID; o.res = c->x;
        BARK_TRACE_SET(c, o, (o.M + 3) / 4);
        launch_linear(s, o);
    }
}

// final LayerNorm + LM head on ONE row (bark.cpp:1391-1405): rows [row0, row0 + n_rows) of the head,
// or the parity-selected codebook window of the coarse model.
void run_lm_head(bark_context * c, GptModel & m, const float * xrow, int row0, int n_rows, int parity_rows, float out_div) {
    LinArgs a;
    if (m.q4) a.wq = q4_rows(m.lm_head_q[0], (size_t) row0, m.hp.n_embd); else a.W = m.lm_head[0] + (size_t) row0 * m.hp.n_embd;
    a.M = n_rows; a.K = m.hp.n_embd; a.N = 1;
    a.x_f32 = xrow; a.ln_g = m.lnf_g; a.ln_b = m.lnf_b; a.epi = EPI_LOGITS; a.out = c->logits; a.ld_out = n_rows;
    a.parity_rows = parity_rows; a.st = c->d_state; a.out_div = out_div;
    BARK_TRACE_SET(c, a, (a.M + 3) / 4);
    launch_linear(c->stream, a);
}

void set_state(bark_context * c, const StepState & st) {
    HIP_OK(hipMemcpyAsync(c->d_state, &st, sizeof(st), hipMemcpyHostToDevice, c->stream));
    HIP_OK(hipStreamSynchronize(c->stream));       // `st` is a stack object
}
StepState get_state(bark_context * c) {
    StepState st;
    HIP_OK(hipMemcpyAsync(&st, c->d_state, sizeof(st), hipMemcpyDeviceToHost, c->stream));
    HIP_OK(hipStreamSynchronize(c->stream));
    if (st.fault) throw std::runtime_error("decode step launched with a context bound below the cached keys (partial scores incomplete)");
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
int run_prefill(bark_context * c, GptModel & m, int n_tokens, bool merge, float * kbase, float * vbase, int pos0) {
    const int N = merge ? n_tokens - 256 : n_tokens;
    EmbedArgs e;
    e.wte = m.wte[0]; e.wte_q = m.wte_q[0]; e.wpe = m.wpe; e.E = m.hp.n_embd; e.n_in = m.hp.n_in_vocab; e.P = c->P; e.tokens = c->d_tokens; e.n_rows = N; e.merge = merge ? 1 : 0; e.pos0 = pos0; e.x = c->x;
    launch_embed_causal(c->stream, e);
    run_layers_rows(c, m, N, true, kbase, vbase, pos0);
    return N;
}

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

void run_sample(bark_context * c, const StageCfg & s, int n_past_add, bool prescaled) {
    SampleArgs a;
    a.prescaled = prescaled ? 1 : 0;
    a.logits = c->logits; a.n = s.lm_rows; a.mode = s.mode; a.min_eos_p = s.min_eos_p; a.eos_token = s.eos_token;
    a.token_base = s.token_base; a.n_past_add = n_past_add; a.out_tokens = c->d_out_tokens;
    a.eos_trace = s.mode == 0 ? c->d_eos_trace : nullptr; a.st = c->d_state;
    a.temp = s.temp; a.u = c->d_u;
    { static const int force_exact = getenv("BARK_HIP_EXACT_SAMPLING") ? atoi(getenv("BARK_HIP_EXACT_SAMPLING")) : 0; a.force_exact = force_exact; }
    const GptModel & m = c->gpt[s.which];
    a.wte = m.wte[0]; a.wte_q = m.wte_q[0]; a.wpe = m.wpe; a.E = m.hp.n_embd; a.n_in = m.hp.n_in_vocab; a.P = c->P; a.x = c->x;
    BARK_TRACE_SET(c, a, 16);
    launch_sample_greedy(c->stream, a);
}

// [embed(state) ->] layers -> LM head [-> greedy sample + embedding of the sampled token]
// In the greedy loop the previous sample kernel has already written x, so the step starts at the layers.
void enqueue_decode_step(bark_context * c, const StageCfg & s, bool sample, int n_past_add, bool embed) {
    GptModel & m = c->gpt[s.which];
    if (embed) {
        EmbedArgs e;
        e.wte = m.wte[0]; e.wte_q = m.wte_q[0]; e.wpe = m.wpe; e.E = m.hp.n_embd; e.n_in = m.hp.n_in_vocab; e.P = c->P; e.n_rows = 1; e.st = c->d_state; e.x = c->x;
        launch_embed_causal(c->stream, e);
    }
    run_layers_decode(c, m);
    // greedy decode step: the LM head divides by 0.7 itself (2512 waves instead of one workgroup doing 10 048 divisions)
    const bool prescale = sample && s.temp == 0.0f;
    run_lm_head(c, m, c->x, s.lm_row0, s.lm_rows, s.parity_rows, prescale ? 0.7f : 0.0f);
    if (sample) run_sample(c, s, n_past_add, prescale);
}

hipGraphExec_t capture_decode(bark_context * c, const StageCfg & s, int n_past_add, int n_steps, int ng) {
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    const int ng_saved = c->decode_ng;
    c->decode_ng = std::max(1, std::min(ng, 4));
    HIP_OK(hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
    try { for (int i = 0; i < n_steps; i++) enqueue_decode_step(c, s, true, n_past_add, false); }
    catch (...) { c->decode_ng = ng_saved; hipGraph_t g2 = nullptr; (void) hipStreamEndCapture(c->stream, &g2); if (g2) (void) hipGraphDestroy(g2); throw; }
    c->decode_ng = ng_saved;
    HIP_OK(hipStreamEndCapture(c->stream, &graph));
    HIP_OK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
    (void) hipGraphDestroy(graph);
    return exec;
}

// n consecutive decode steps; the cache holds n_past rows before the first of them (step k then attends over n_past + k + 1 keys).
// Between two graph launches the GPU idles longer than between two kernels of one graph, so runs of steps are replayed from an
// eight-step graph (the state lives on the device: a step needs nothing from the host) and the remainder from the one-step graph.
// The graph variant is picked by the number of 256-key groups the launch can touch: its kernels request those keys' partial scores
// and value rows at wave launch instead of waiting ~0.5 us for the context length to arrive from the device-resident state.
void decode_steps_greedy(bark_context * c, const StageCfg & s, int n, int n_past) {
    GptModel & m = c->gpt[s.which];
    if (!c->use_graph) {
        struct Restore { bark_context * c; ~Restore() { c->decode_ng = 4; } } restore{c};      // also when a launch throws
        for (int k = 0; k < n; k++) { c->decode_ng = std::min(4, (n_past + k + 256) / 256); enqueue_decode_step(c, s, true, 1, false); }
        return;
    }
    int k = 0;
    while (k < n) {
        const int g = n - k >= 8 ? 8 : 1;
        const int ng = std::min(4, (n_past + k + g + 255) / 256);      // ctx of the last step of this launch = n_past + k + g
        hipGraphExec_t & e = g == 8 ? m.decode_graph8[ng] : m.decode_graph[ng];
        if (!e) e = capture_decode(c, s, 1, g, ng);
        HIP_OK(hipGraphLaunch(e, c->stream));
        c->stats.graph_replays++;
        k += g;
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

}  // namespace detail

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

namespace detail {
// one fine forward (bark_build_fine_gpt_graph, bark.cpp:1416-1584): tokens [8][plane] hold Z windows of 1024 positions back to back;
// logits -> rb.logits [Z * 1024][n_rows]
void run_fine_forward(bark_context * c, int nn, int n_rows, const RowBufs * rbp, int Z) {
    GptModel & m = c->gpt[2];
    const int E = m.hp.n_embd, N = 1024 * Z;
    const RowBufs own = own_rows(c);
    const RowBufs & rb = rbp ? *rbp : own;
    launch_embed_fine(c->stream, m.wte, m.wte_q, m.wpe, E, m.hp.n_in_vocab, rb.tokens, nn, rb.x, N, rb.plane);
    if (rbp) run_layers_rows(c, m, N, false, c->fine_batch.kc, c->fine_batch.vc, 0, rbp, 1024, (size_t) E * c->P);
    else     run_layers_rows(c, m, N, false);
    if (m.w32)     launch_ln_rows_f32(c->stream, rb.x, N, E, m.lnf_g, m.lnf_b, c->xn32);
    else if (m.q4) launch_q8_rows(c->stream, rb.x, N, E, m.lnf_g, m.lnf_b, c->xq);
    else           launch_ln_rows(c->stream, rb.x, N, E, m.lnf_g, m.lnf_b, rb.xn);
    LinArgs a;
    a.W = m.lm_head[nn - 1]; a.wq = m.lm_head_q[nn - 1]; a.xq = c->xq; if (m.w32) a.x_f32 = c->xn32; a.M = n_rows; a.K = E; a.N = N; a.x_f16 = rb.xn; a.epi = EPI_LOGITS; a.out = rb.logits; a.ld_out = n_rows;
    a.fast = fine_products_on_f16_mfma(c, m, false) ? 1 : (!m.q4 && !m.w32) ? c->fast_gemm : 0;
    launch_linear(c->stream, a);                           // lm_heads[codebook_idx - n_codes_given], bark.cpp:1573
}
}  // namespace detail

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
            decode_steps_greedy(c, s, batch_end - issued, N + issued - 1);
            for (; issued < batch_end; issued++) progress(c, SEMANTIC, 100 * (issued + 1) / std::max(1, p.n_steps_text_encoder));
            cur = get_state(c);                                        // poll the stop rule every 32 steps
            if (cur.eos_step != INT32_MAX || issued >= n_steps) break;
        }
        const int n_keep = std::min(cur.eos_step, issued);
        out.resize((size_t) n_keep);
        if (n_keep) copy_to_host(c, out.data(), c->d_out_tokens, (size_t) n_keep * 4);
        if (eos_trace) {
            const int n_tr = std::min(issued, cur.eos_step == INT32_MAX ? issued : cur.eos_step + 1);
            eos_trace->resize((size_t) n_tr);
            if (n_tr) copy_to_host(c, eos_trace->data(), c->d_eos_trace, (size_t) n_tr * 4);
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
    const bool reuse_prefix = !(crosscheck_mask() & 8);
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
            decode_steps_greedy(c, s, 1, L);
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
            decode_steps_greedy(c, s, steps_here - 1, N);
            for (int j = 1; j < steps_here; j++) progress(c, COARSE, 100 * (step_idx + j + 1) / n_steps);
            const StepState cur = get_state(c);
            std::vector<int32_t> got((size_t) steps_here);
            copy_to_host(c, got.data(), c->d_out_tokens, (size_t) steps_here * 4);
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
                    else launch_sample_rows_multinomial(c->stream, c->logits, cs, 1024, cs, p.fine_temp, c->d_u + (size_t) (nn - nc) * 1024, pick_dst, 1, c->d_state);
                };
                if (c->use_graph && rel == 0) {
                    hipGraphExec_t & g = c->fine_graphs[nn + (fine_products_on_f16_mfma(c, m, false) ? 8 : 0)];
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
// The fine stage of several utterances at once (lock-step batches): window n of every utterance that has one goes through the six
// forward passes side by side - rows z * 1024 .. z * 1024 + 1023 of every activation are utterance z's window, the attention runs per
// window (grid.z), the products see 1024 Z rows.  Per utterance this is engine_fine's arithmetic, row for row.
// ---------------------------------------------------------------------------------------------------
namespace detail {
void ensure_fine_batch(bark_context * c, int Z) {
    bark_context::FineBatch & fb = c->fine_batch;
    if (fb.cap >= Z) return;
    GptModel & m = c->gpt[2];
    const size_t R = (size_t) Z * 1024, E = (size_t) m.hp.n_embd;
    fb.x = dev_alloc<float>(c, R * E); fb.q = dev_alloc<float>(c, R * E);
    fb.xn = dev_alloc<half_t>(c, R * E); fb.att = dev_alloc<half_t>(c, R * E); fb.hbuf = dev_alloc<half_t>(c, R * E * 4);
    fb.kc = dev_alloc<float>(c, (size_t) Z * E * c->P); fb.vc = dev_alloc<float>(c, (size_t) Z * E * c->P);
    if (c->fast_gemm) { fb.q16 = dev_alloc<half_t>(c, R * E); fb.k16 = dev_alloc<half_t>(c, R * E); fb.vt16 = dev_alloc<half_t>(c, R * E); }
    fb.logits = dev_alloc<float>(c, R * 1024);
    fb.tokens = dev_alloc<int32_t>(c, 8 * R); fb.picks = dev_alloc<int32_t>(c, R);
    fb.u = dev_alloc<double>(c, 6 * R);
    fb.cap = Z;                                             // (a smaller earlier allocation stays with the context until it is freed)
}
RowBufs fine_batch_rows(bark_context * c, int Z) {
    const bark_context::FineBatch & fb = c->fine_batch;
    RowBufs rb; rb.x = fb.x; rb.q = fb.q; rb.xn = fb.xn; rb.att = fb.att; rb.hbuf = fb.hbuf; rb.q16 = fb.q16; rb.k16 = fb.k16; rb.vt16 = fb.vt16;
    rb.logits = fb.logits; rb.tokens = fb.tokens; rb.plane = Z * 1024;
    return rb;
}
}  // namespace detail

std::vector<std::vector<int32_t>> engine_fine_many(bark_context * c, const std::vector<const std::vector<int32_t> *> & coarse, std::vector<std::mt19937> * rngs) {
    HIP_OK(hipSetDevice(c->device));
    const bark_context_params & p = c->params;
    GptModel & m = c->gpt[2];
    const int nc = p.n_coarse_codebooks, nf = p.n_fine_codebooks, cs = p.codebook_size;
    if (nc != 2 || nf != 8 || cs != 1024) throw std::runtime_error("fine: only 2 -> 8 codebooks of 1024 entries are supported");
    if (m.q4 || m.w32 || c->host_sampling) throw std::runtime_error("fine_many: f16 model files, device sampling");
    const JobScope job(c);                                   // windows side by side are the job's pass: products in C1m under the default policy
    const int U = (int) coarse.size();
    const bool greedy = p.fine_temp == 0.0f;
    if (!greedy && (!rngs || (int) rngs->size() != U)) throw std::runtime_error("fine_many: one generator per utterance is needed for fine_temp > 0");
    struct Utt { int T, L, n_loops; std::vector<int32_t> in_arr; };
    std::vector<Utt> us((size_t) U);
    int max_loops = 0;
    for (int u = 0; u < U; u++) {
        const std::vector<int32_t> & co = *coarse[(size_t) u];
        Utt & t = us[(size_t) u];
        t.T = (int) co.size() / nc;
        if (t.T <= 0 || t.T > 8192) throw std::runtime_error("fine: number of frames must be in 1..8192");
        for (int32_t v : co) if (v < 0 || v >= cs) throw std::runtime_error("fine: coarse code out of range");
        t.L = std::max(t.T, 1024);
        t.in_arr.assign((size_t) t.L * 8, cs);                                              // bark.cpp:1983-1996
        for (int i = 0; i < t.T; i++) for (int ch = 0; ch < nc; ch++) t.in_arr[(size_t) i * 8 + ch] = co[(size_t) i * nc + ch];
        t.n_loops = std::max(0, (int) ceilf((float) (t.L - 1024) / 512.f)) + 1;             // bark.cpp:1998
        max_loops = std::max(max_loops, t.n_loops);
    }
    ensure_fine_batch(c, U);
    bark_context::FineBatch & fb = c->fine_batch;
    StepState st = fresh_state();
    set_state(c, st);
    std::vector<int32_t> buf;
    std::vector<double> ubuf;
    for (int n = 0; n < max_loops; n++) {
        std::vector<int> act, rels, fills;
        for (int u = 0; u < U; u++) if (n < us[(size_t) u].n_loops) act.push_back(u);
        const int Z = (int) act.size(), R = Z * 1024;
        buf.assign((size_t) 8 * R, cs);
        bool any_rel = false;
        for (int z = 0; z < Z; z++) {
            const Utt & t = us[(size_t) act[(size_t) z]];
            const int start_idx = std::min(n * 512, t.L - 1024), start_fill_idx = std::min(n * 512, t.L - 512);      // bark.cpp:2002-2013
            rels.push_back(start_fill_idx - start_idx); fills.push_back(start_fill_idx);
            any_rel = any_rel || rels.back() > 0;
            for (int ch = 0; ch < 8; ch++) for (int j = 0; j < 1024; j++) buf[(size_t) ch * R + (size_t) z * 1024 + j] = t.in_arr[(size_t) (start_idx + j) * 8 + ch];
        }
        HIP_OK(hipMemcpyAsync(fb.tokens, buf.data(), buf.size() * 4, hipMemcpyHostToDevice, c->stream));
        if (!greedy) {
            // utterance z draws (nf - nc) * 1024 uniforms per window, codebook-major, from a COPY of its generator (upload_uniforms)
            ubuf.assign((size_t) 6 * R, 0.0);
            for (int z = 0; z < Z; z++) {
                std::mt19937 tmp = (*rngs)[(size_t) act[(size_t) z]];
                for (int k = 0; k < nf - nc; k++) for (int j = 0; j < 1024; j++) ubuf[(size_t) k * R + (size_t) z * 1024 + j] = std::generate_canonical<double, 53>(tmp);
            }
            HIP_OK(hipMemcpyAsync(fb.u, ubuf.data(), ubuf.size() * 8, hipMemcpyHostToDevice, c->stream));
        }
        HIP_OK(hipStreamSynchronize(c->stream));
        const RowBufs rb = fine_batch_rows(c, Z);
        for (int nn = nc; nn < nf; nn++) {
            progress(c, FINE, 100 * (n * (nf - nc) + (nn - nc + 1)) / (max_loops * (nf - nc)));
            // a window with rel > 0 (the last ones of a long utterance) keeps the positions below rel: picks go to a scratch row first
            int32_t * pick_dst = any_rel ? fb.picks : fb.tokens + (size_t) nn * R;
            run_fine_forward(c, nn, cs, &rb, Z);
            if (greedy) launch_argmax_rows(c->stream, fb.logits, cs, R, cs, pick_dst, 1, c->d_state);
            else launch_sample_rows_multinomial(c->stream, fb.logits, cs, R, cs, p.fine_temp, fb.u + (size_t) (nn - nc) * R, pick_dst, 1, c->d_state);
            if (any_rel)
                for (int z = 0; z < Z; z++)
                    HIP_OK(hipMemcpyAsync(fb.tokens + (size_t) nn * R + (size_t) z * 1024 + rels[(size_t) z], fb.picks + (size_t) z * 1024 + rels[(size_t) z],
                                          (size_t) (1024 - rels[(size_t) z]) * 4, hipMemcpyDeviceToDevice, c->stream));
            c->stats.n_sample_fine += R;
        }
        HIP_OK(hipMemcpyAsync(buf.data(), fb.tokens, buf.size() * 4, hipMemcpyDeviceToHost, c->stream));
        HIP_OK(hipStreamSynchronize(c->stream));
        for (int z = 0; z < Z; z++) {
            Utt & t = us[(size_t) act[(size_t) z]];
            const int rel = rels[(size_t) z];
            if (!greedy) (*rngs)[(size_t) act[(size_t) z]].discard(2ull * (unsigned long long) ((nf - nc) * 1024));
            for (int nn = nc; nn < nf; nn++)                                                 // bark.cpp:2041-2046
                for (int j = 0; j < cs - rel; j++) t.in_arr[(size_t) (fills[(size_t) z] + j) * 8 + nn] = buf[(size_t) nn * R + (size_t) z * 1024 + rel + j];
        }
    }
    const StepState cur = get_state(c);
    c->stats.n_near_tie += cur.near_tie;
    std::vector<std::vector<int32_t>> out((size_t) U);
    for (int u = 0; u < U; u++) { us[(size_t) u].in_arr.resize((size_t) us[(size_t) u].T * 8); out[(size_t) u] = std::move(us[(size_t) u].in_arr); }
    return out;
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

}  // namespace barkhip
