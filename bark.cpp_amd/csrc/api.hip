// api.hip - C ABI of libbark.so: the reference's bark.h surface (drop-in) plus bark_mi355x.h.
// No exception crosses the boundary; failures are reported as nullptr / false / negative counts with
// a diagnostic on stderr, like the reference (bark.cpp:1174-1177, 2379-2401).
#include "engine.h"

#include <chrono>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <thread>
#include <vector>

using namespace barkhip;

namespace {
int64_t wall_us() {
    return std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
// noexcept shell of every entry point.  Kernel launches are not checked one by one (hipLaunchKernelGGL), so the sticky launch
// error is read here: a launch that was rejected means missing results, which must fail the call instead of returning them.
template <typename F> auto guarded(const char * what, decltype(std::declval<F>()()) fail, F && f, bool uses_gpu = true) -> decltype(f()) {
    try {
        if (!uses_gpu) return f();                                   // host-only entry points (bark_model_quantize) work without a device
        (void) hipGetLastError();
        auto r = f();
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess) { fprintf(stderr, "%s: a kernel launch failed: %s\n", what, hipGetErrorString(e)); return fail; }
        return r;
    }
    catch (const std::exception & e) { fprintf(stderr, "%s: %s\n", what, e.what()); }
    catch (...) { fprintf(stderr, "%s: unknown failure\n", what); }
    return fail;
}
}  // namespace

extern "C" {

// defaults: /root/reference/bark.cpp:2202-2232
struct bark_context_params bark_context_default_params(void) {
    struct bark_context_params p;
    memset(&p, 0, sizeof(p));
    p.verbosity = LOW;
    p.temp = 0.7f;
    p.fine_temp = 0.5f;
    p.min_eos_p = 0.2f;
    p.sliding_window_size = 60;
    p.max_coarse_history = 630;
    p.sample_rate = 24000;
    p.target_bandwidth = 6;
    p.cls_token_id = 101;
    p.sep_token_id = 102;
    p.n_steps_text_encoder = 768;
    p.text_pad_token = 129595;
    p.text_encoding_offset = 10048;
    p.semantic_rate_hz = 49.9f;
    p.semantic_pad_token = 10000;
    p.semantic_vocab_size = 10000;
    p.semantic_infer_token = 129599;
    p.coarse_rate_hz = 75.0f;
    p.coarse_infer_token = 12050;
    p.coarse_semantic_pad_token = 12048;
    p.n_coarse_codebooks = 2;
    p.n_fine_codebooks = 8;
    p.codebook_size = 1024;
    p.progress_callback = nullptr;
    p.progress_callback_user_data = nullptr;
    return p;
}

struct bark_context * bark_load_model(const char * model_path, struct bark_context_params params, uint32_t seed) {
    const int64_t t0 = wall_us();
    if (!model_path) { fprintf(stderr, "bark_load_model: null path\n"); return nullptr; }
    bark_context * ctx = guarded("bark_load_model", (bark_context *) nullptr, [&] { return engine_load(model_path, params, seed); });
    if (ctx) ctx->stats.t_load_us = wall_us() - t0;
    return ctx;
}

bool bark_generate_audio(struct bark_context * bctx, const char * text, int n_threads) {
    (void) n_threads;      // CPU-backend hint in the reference (bark.cpp:1623-1625); the HIP engine has no use for it
    if (!bctx || !text) { fprintf(stderr, "bark_generate_audio: null argument\n"); return false; }
    return guarded("bark_generate_audio", false, [&] { return engine_generate(bctx, text); });
}

float * bark_get_audio_data(struct bark_context * bctx) { return (bctx && !bctx->audio.empty()) ? bctx->audio.data() : nullptr; }
int bark_get_audio_data_size(struct bark_context * bctx) { return bctx ? (int) bctx->audio.size() : 0; }
int64_t bark_get_load_time(struct bark_context * bctx) { return bctx ? bctx->stats.t_load_us : 0; }
int64_t bark_get_eval_time(struct bark_context * bctx) { return bctx ? bctx->stats.t_eval_us : 0; }

// Deviation from the reference, kept on purpose: bark.cpp:2403-2407 zeroes the WHOLE statistics struct and bark_generate_audio
// calls it first (bark.cpp:2131), so the reference's bark_get_load_time reads 0 after any generate call (examples/main prints "load time = 0.00 ms").
// Here the load time survives a reset: it describes the context, not a call.
void bark_reset_statistics(struct bark_context * bctx) {
    if (!bctx) return;
    const int64_t t_load = bctx->stats.t_load_us;
    bctx->stats = bark_hip_stats{};
    bctx->stats.t_load_us = t_load;
}

bool bark_model_quantize(const char * fname_inp, const char * fname_out, enum ggml_ftype ftype) {
    if (!fname_inp || !fname_out) { fprintf(stderr, "bark_model_quantize: null path\n"); return false; }
    return guarded("bark_model_quantize", false, [&] {
        std::string err;
        if (!model_quantize(fname_inp, fname_out, (int) ftype, err)) { fprintf(stderr, "bark_model_quantize: %s\n", err.c_str()); return false; }
        return true;
    }, /*uses_gpu=*/false);
}

void bark_free(struct bark_context * bctx) { delete bctx; }

// ---- ggml.h shim --------------------------------------------------------------------------------------
void ggml_time_init(void) {}
int64_t ggml_time_us(void) { return wall_us(); }
int64_t ggml_time_ms(void) { return wall_us() / 1000; }
struct ggml_context * ggml_init(struct ggml_init_params params) { (void) params; return nullptr; }
void ggml_free(struct ggml_context * ctx) { (void) ctx; }

// ---- bark_mi355x.h -------------------------------------------------------------------------------------
int bark_hip_hparams(struct bark_context * bctx, int which, int32_t * out10) {
    if (!bctx || which < 0 || which > 2 || !out10) return -1;
    const GptHparams & h = bctx->gpt[which].hp;
    const int32_t v[10] = {h.n_layer, h.n_head, h.n_embd, h.block_size, h.bias, h.n_in_vocab, h.n_out_vocab, h.n_lm_heads, h.n_wtes, h.ftype};
    memcpy(out10, v, sizeof(v));
    return 0;
}

void bark_hip_set_params(struct bark_context * bctx, struct bark_context_params params) {
    if (!bctx) return;
    bctx->params = params;
    guarded("bark_hip_set_params", 0, [&] { engine_invalidate_graphs(bctx); return 0; });   // sampling constants are baked into the graphs
}

int bark_hip_tokenize(struct bark_context * bctx, const char * text, int32_t * out513) {
    if (!bctx || !text || !out513) return -1;
    return guarded("bark_hip_tokenize", -1, [&] {
        PromptParams pp;
        pp.block_size = bctx->gpt[0].hp.block_size; pp.text_encoding_offset = bctx->params.text_encoding_offset;
        pp.text_pad_token = bctx->params.text_pad_token; pp.semantic_pad_token = bctx->params.semantic_pad_token;
        pp.semantic_infer_token = bctx->params.semantic_infer_token;
        std::vector<int32_t> ids = build_semantic_prompt(bctx->vocab, pp, text, false);
        memcpy(out513, ids.data(), ids.size() * 4);
        return (int) ids.size();
    });
}

int bark_hip_bert_tokenize(struct bark_context * bctx, const char * text, int32_t * out, int n_max) {
    if (!bctx || !text || !out || n_max <= 0) return -1;
    return guarded("bark_hip_bert_tokenize", -1, [&] { return wordpiece_encode(bctx->vocab, text, out, n_max, false); });
}

int bark_hip_gpt_eval(struct bark_context * bctx, int which, const int32_t * tokens, int n_tokens, int n_past, int merge_ctx, float * logits) {
    if (!bctx || !tokens || !logits) return -1;
    return guarded("bark_hip_gpt_eval", -1, [&] { return engine_gpt_eval(bctx, which, tokens, n_tokens, n_past, merge_ctx != 0, logits); });
}

int bark_hip_fine_eval(struct bark_context * bctx, const int32_t * tokens_8x1024, int nn, float * logits) {
    if (!bctx || !tokens_8x1024 || !logits) return -1;
    return guarded("bark_hip_fine_eval", -1, [&] { engine_fine_eval(bctx, tokens_8x1024, nn, logits); return 0; });
}

int bark_hip_semantic(struct bark_context * bctx, const int32_t * prompt513, int32_t * out, int capacity, float * eos_trace) {
    if (!bctx || !prompt513 || !out || capacity < 0) return -1;
    return guarded("bark_hip_semantic", -1, [&] {
        std::vector<int32_t> prompt(prompt513, prompt513 + 513);
        std::vector<float> tr;
        std::vector<int32_t> r = engine_semantic(bctx, prompt, eos_trace ? &tr : nullptr);
        // the trace holds one entry per evaluated step: at most one more than the ids kept
        if ((int) r.size() > capacity || (eos_trace && (int) tr.size() > capacity + 1)) throw std::runtime_error("output buffer too small");
        if (!r.empty()) memcpy(out, r.data(), r.size() * 4);
        if (eos_trace && !tr.empty()) memcpy(eos_trace, tr.data(), tr.size() * 4);
        return (int) r.size();
    });
}

int bark_hip_coarse(struct bark_context * bctx, const int32_t * semantic, int n_semantic, int32_t * out_Tx2, int capacity_rows) {
    if (!bctx || !semantic || n_semantic <= 0 || !out_Tx2) return -1;
    return guarded("bark_hip_coarse", -1, [&] {
        std::vector<int32_t> r = engine_coarse(bctx, std::vector<int32_t>(semantic, semantic + n_semantic));
        if ((int) (r.size() / 2) > capacity_rows) throw std::runtime_error("output buffer too small");
        memcpy(out_Tx2, r.data(), r.size() * 4);
        return (int) r.size() / 2;
    });
}

int bark_hip_fine(struct bark_context * bctx, const int32_t * coarse_Tx2, int T, int32_t * out_Tx8, int capacity_rows) {
    if (!bctx || !coarse_Tx2 || T <= 0 || !out_Tx8) return -1;
    return guarded("bark_hip_fine", -1, [&] {
        std::vector<int32_t> r = engine_fine(bctx, std::vector<int32_t>(coarse_Tx2, coarse_Tx2 + (size_t) T * 2));
        if ((int) (r.size() / 8) > capacity_rows) throw std::runtime_error("output buffer too small");
        memcpy(out_Tx8, r.data(), r.size() * 4);
        return (int) r.size() / 8;
    });
}

int bark_hip_fine_many(struct bark_context * bctx, const int32_t * coarse_concat, const int * T, int n, int32_t * out_concat, int capacity_rows) {
    if (!bctx || !coarse_concat || !T || n <= 0 || n > 64 || !out_concat) return -1;          // 64 windows side by side: 270 MB of logits
    return guarded("bark_hip_fine_many", -1, [&] {
        std::vector<std::vector<int32_t>> co((size_t) n);
        std::vector<const std::vector<int32_t> *> ptr;
        std::vector<std::mt19937> rngs;
        size_t off = 0, total = 0;
        for (int i = 0; i < n; i++) {
            if (T[i] <= 0) throw std::runtime_error("fine_many: every utterance needs at least one frame");
            co[(size_t) i].assign(coarse_concat + off * 2, coarse_concat + (off + (size_t) T[i]) * 2);
            off += (size_t) T[i]; total += (size_t) T[i];
            ptr.push_back(&co[(size_t) i]);
            rngs.push_back(std::mt19937((uint32_t) bctx->rng()));
        }
        if ((long) total > (long) capacity_rows) throw std::runtime_error("output buffer too small");
        std::vector<std::vector<int32_t>> r = engine_fine_many(bctx, ptr, &rngs);
        off = 0;
        for (int i = 0; i < n; i++) { memcpy(out_concat + off * 8, r[(size_t) i].data(), r[(size_t) i].size() * 4); off += r[(size_t) i].size() / 8; }
        return (int) total;
    });
}

struct bark_context * bark_hip_load_model_on_device(const char * model_path, struct bark_context_params params, uint32_t seed, int device) {
    if (!model_path) { fprintf(stderr, "bark_hip_load_model_on_device: null path\n"); return nullptr; }
    if (device < 0) { fprintf(stderr, "bark_hip_load_model_on_device: negative device ordinal\n"); return nullptr; }
    const int64_t t0 = wall_us();
    bark_context * ctx = guarded("bark_hip_load_model_on_device", (bark_context *) nullptr, [&] { return engine_load(model_path, params, seed, device); });
    if (ctx) ctx->stats.t_load_us = wall_us() - t0;
    return ctx;
}

int bark_hip_set_fine_order(struct bark_context * bctx, int order) {
    if (!bctx || order < 0 || order > 2) return -1;
    return guarded("bark_hip_set_fine_order", -1, [&] {
        bctx->fine_order = order;
        if (bctx->tail) bctx->tail->fine_order = order;
        return 0;
    });
}

int bark_hip_codec_decode(struct bark_context * bctx, const int32_t * codes, int n_q, int T, float * pcm, int capacity) {
    if (!bctx || !codes || !pcm) return -1;
    return guarded("bark_hip_codec_decode", -1, [&] {
        std::vector<float> r = engine_codec_decode(bctx, codes, n_q, T, -1, nullptr);
        if ((int) r.size() > capacity) throw std::runtime_error("output buffer too small");
        memcpy(pcm, r.data(), r.size() * 4);
        return (int) r.size();
    });
}

int bark_hip_codec_tap(struct bark_context * bctx, const int32_t * codes, int n_q, int T, int stage, float * out, int capacity) {
    if (!bctx || !codes || !out) return -1;
    return guarded("bark_hip_codec_tap", -1, [&] {
        std::vector<float> tap;
        engine_codec_decode(bctx, codes, n_q, T, stage, &tap);
        if ((int) tap.size() > capacity) return -1;
        memcpy(out, tap.data(), tap.size() * 4);
        return (int) tap.size();
    });
}

struct bark_context * bark_hip_clone_context(struct bark_context * src, uint32_t seed) {
    if (!src) return nullptr;
    return guarded("bark_hip_clone_context", (bark_context *) nullptr, [&] { return engine_clone(src, seed); });
}

int bark_hip_generate_audio_batch(struct bark_context ** ctxs, const char * const * texts, int n) {
    if (!ctxs || !texts || n <= 0) return 0;
    std::vector<char> ok((size_t) n, 0);
    std::vector<std::thread> th;
    for (int i = 0; i < n; i++)
        th.emplace_back([&, i] { ok[(size_t) i] = (ctxs[i] && texts[i]) ? (char) guarded("bark_hip_generate_audio_batch", false, [&] { return engine_generate(ctxs[i], texts[i]); }) : 0; });
    for (auto & t : th) t.join();
    int good = 0;
    for (char c : ok) good += c;
    return good;
}

int bark_hip_generate_batch(struct bark_context * bctx, const char * const * texts, int n) {
    if (!bctx || !texts || n <= 0) return -1;
    for (int i = 0; i < n; i++) if (!texts[i]) return -1;
    return guarded("bark_hip_generate_batch", -1, [&] { return engine_generate_batch(bctx, texts, n, nullptr); });
}
int bark_hip_generate_batch_seeded(struct bark_context * bctx, const char * const * texts, int n, const uint32_t * seeds) {
    if (!bctx || !texts || !seeds || n <= 0) return -1;
    for (int i = 0; i < n; i++) if (!texts[i]) return -1;
    return guarded("bark_hip_generate_batch_seeded", -1, [&] { return engine_generate_batch(bctx, texts, n, seeds); });
}
int bark_hip_generate_batch_ex(struct bark_context * bctx, const char * const * texts, int n, const struct bark_hip_request_params * per_utterance) {
    if (!bctx || !texts || !per_utterance || n <= 0) return -1;
    for (int i = 0; i < n; i++) if (!texts[i]) return -1;
    return guarded("bark_hip_generate_batch_ex", -1, [&] { return engine_generate_batch(bctx, texts, n, nullptr, per_utterance); });
}
int bark_hip_profile_lock_step(struct bark_context * bctx, int which, int n_slots, int ctx, int reps, char * json_out, int capacity) {
    if (!bctx || !json_out || capacity < 2) return -1;
    return guarded("bark_hip_profile_lock_step", -1, [&] {
        std::vector<std::pair<std::string, double>> tl;
        engine_profile_lock_step(bctx, which, n_slots, ctx, reps, tl);
        std::string js = "[";
        for (size_t i = 0; i < tl.size(); i++) {
            char buf[160];
            snprintf(buf, sizeof(buf), "%s{\"site\": \"%s\", \"us\": %.3f}", i ? ", " : "", tl[i].first.c_str(), tl[i].second);
            js += buf;
        }
        js += "]";
        if ((int) js.size() + 1 > capacity) return -1;
        memcpy(json_out, js.c_str(), js.size() + 1);
        return (int) js.size();
    });
}
int bark_hip_reserve_batch(struct bark_context * bctx, int slots) {
    if (!bctx) return -1;
    return guarded("bark_hip_reserve_batch", -1, [&] { engine_reserve_batch(bctx, slots); return 0; });
}
int bark_hip_batch_audio(struct bark_context * bctx, int i, float ** data) {
    if (!bctx || i < 0 || i >= (int) bctx->batch_results.size() || !bctx->batch_results[(size_t) i].ok) return -1;
    if (data) *data = bctx->batch_results[(size_t) i].audio.data();
    return (int) bctx->batch_results[(size_t) i].audio.size();
}
int bark_hip_batch_tokens(struct bark_context * bctx, int i, int stage, int32_t * out, int capacity) {
    if (!bctx || i < 0 || i >= (int) bctx->batch_results.size() || stage < 0 || stage > 2) return -1;
    const auto & r = bctx->batch_results[(size_t) i];
    const std::vector<int32_t> & v = stage == 0 ? r.semantic : stage == 1 ? r.coarse : r.fine;
    if (!out || capacity < (int) v.size()) return -1;
    if (!v.empty()) memcpy(out, v.data(), v.size() * 4);
    return (int) v.size();
}

static int copy_out(const std::vector<int32_t> & v, int per_row, int32_t * out, int capacity_rows) {
    const int rows = (int) v.size() / per_row;
    if (!out || capacity_rows < rows) return -1;
    if (rows) memcpy(out, v.data(), v.size() * 4);
    return rows;
}
int bark_hip_get_semantic_tokens(struct bark_context * bctx, int32_t * out, int capacity) { return bctx ? copy_out(bctx->semantic_tokens, 1, out, capacity) : -1; }
int bark_hip_get_coarse_tokens(struct bark_context * bctx, int32_t * out, int capacity_rows) { return bctx ? copy_out(bctx->coarse_tokens, 2, out, capacity_rows) : -1; }
int bark_hip_get_fine_tokens(struct bark_context * bctx, int32_t * out, int capacity_rows) { return bctx ? copy_out(bctx->fine_tokens, 8, out, capacity_rows) : -1; }

void bark_hip_get_stats(struct bark_context * bctx, struct bark_hip_stats * out) { if (bctx && out) *out = bctx->stats; }

double bark_hip_time_decode_step(struct bark_context * bctx, int which, int ctx, int iters, double * bytes_per_step) {
    if (!bctx) return -1.0;
    return guarded("bark_hip_time_decode_step", -1.0, [&] { return engine_time_decode_step(bctx, which, ctx, iters, bytes_per_step); });
}
double bark_hip_time_gemv(struct bark_context * bctx, int which, int op, int iters, double * bytes_per_launch) {
    if (!bctx) return -1.0;
    return guarded("bark_hip_time_gemv", -1.0, [&] { return engine_time_gemv(bctx, which, op, iters, bytes_per_launch); });
}
double bark_hip_time_fine_passes(struct bark_context * bctx, int n_windows, int iters, double * flops_per_pass) {
    if (!bctx) return -1.0;
    return guarded("bark_hip_time_fine_passes", -1.0, [&] { return engine_time_fine_pass(bctx, iters, flops_per_pass, n_windows); });
}

double bark_hip_time_fine_pass(struct bark_context * bctx, int iters, double * flops_per_pass) {
    if (!bctx) return -1.0;
    return guarded("bark_hip_time_fine_pass", -1.0, [&] { return engine_time_fine_pass(bctx, iters, flops_per_pass); });
}
double bark_hip_time_slots(struct bark_context * bctx, int which, int op, int n_slots, int kind, int ctx, int iters) {
    if (!bctx) return -1.0;
    return guarded("bark_hip_time_slots", -1.0, [&] { return engine_time_slots(bctx, which, op, n_slots, kind, ctx, iters); });
}
#ifdef BARK_TRACE
__attribute__((visibility("default"))) int bark_hip_trace_decode_step(struct bark_context * bctx, int which, int ctx, int replays, unsigned long long * out6, int cap_records) {
    if (!bctx || !out6) return -1;
    return guarded("bark_hip_trace_decode_step", -1, [&] { return engine_trace_decode_step(bctx, which, ctx, replays, out6, cap_records); });
}
#endif
const char * bark_hip_describe(struct bark_context * bctx) { return bctx ? bctx->description.c_str() : "no context"; }

}  // extern "C"
