// engine_internal.h - declarations shared by the engine's translation units (engine_load.hip: model upload; engine.hip:
// building blocks and the stage loops; engine_batch.hip: lock-step batching; engine_timing.hip: bench hooks).  Not an API.
#pragma once
#include "engine.h"

#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <stdexcept>
#include <string>
#include <vector>

#define HIP_OK(expr)                                                                                         \
    do {                                                                                                     \
        hipError_t _e = (expr);                                                                              \
        if (_e != hipSuccess)                                                                                \
            throw std::runtime_error(std::string("HIP error: ") + hipGetErrorString(_e) + " at " #expr);   \
    } while (0)

// diagnostic build: hand the context's trace log to a kernel's argument block and number the launch
#ifdef BARK_TRACE
#define BARK_TRACE_SET(c, args, nwaves) do { (args).tr.rec = (c)->trace_rec; (args).tr.pos = (c)->trace_pos; (args).tr.cap = (c)->trace_cap; (args).tr.kid = (c)->trace_kid++; (args).tr.base = (c)->trace_base; (args).tr.per_replay = (c)->trace_per_replay; (c)->trace_base += (unsigned) (nwaves); } while (0)
#else
#define BARK_TRACE_SET(c, args, nwaves) do { } while (0)
#endif

namespace barkhip { namespace detail {

inline int64_t now_us() {
    return std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// BARK_HIP_POISON=1 (diagnostic): every run-time allocation (KV caches, slot rows, scratch, partial scores) starts as 0xFF bytes - NaN as
// f32 / f16, -1 as an id - so that a read of memory nobody has written shows up in the results whatever a fresh box happens to hand out
// (the GPU suite is expected to pass unchanged under it)
inline bool poison_allocations() {
    static const bool v = getenv("BARK_HIP_POISON") && atoi(getenv("BARK_HIP_POISON")) != 0;
    return v;
}
// BARK_HIP_GUARD=1 (diagnostic): every run-time allocation sits between two 64 KB guard bands filled with 0xA5; guard_check() (engine_load.hip)
// verifies them at the end of every lock-step job and when the context is freed - a write of any kernel of the process that lands just outside one of
// this context's buffers (or runs over from a neighbour) is reported on stderr with the buffer's index and the offset
constexpr size_t kGuardBytes = 64 * 1024;
inline bool guard_allocations() {
    static const bool v = getenv("BARK_HIP_GUARD") && atoi(getenv("BARK_HIP_GUARD")) != 0;
    return v;
}
template <typename T> T * dev_alloc(bark_context * ctx, size_t count) {
    void * p = nullptr;
    const size_t bytes = std::max<size_t>(count, 1) * sizeof(T);
    if (guard_allocations() && ctx->stream) {
        const size_t body = (bytes + 255) & ~(size_t) 255;
        HIP_OK(hipMalloc(&p, body + 2 * kGuardBytes));
        ctx->allocs.push_back(p);
        ctx->guarded.push_back({p, body});
        HIP_OK(hipMemsetAsync(p, 0xA5, body + 2 * kGuardBytes, ctx->stream));
        if (poison_allocations()) HIP_OK(hipMemsetAsync((char *) p + kGuardBytes, 0xFF, body, ctx->stream));
        HIP_OK(hipStreamSynchronize(ctx->stream));
        return (T *) ((char *) p + kGuardBytes);
    }
    HIP_OK(hipMalloc(&p, bytes));
    ctx->allocs.push_back(p);
    if (poison_allocations() && ctx->stream) {
        HIP_OK(hipMemsetAsync(p, 0xFF, bytes, ctx->stream));
        HIP_OK(hipStreamSynchronize(ctx->stream));
    }
    return (T *) p;
}
int guard_check(bark_context * ctx, const char * where);      // number of damaged guard bands (0 without BARK_HIP_GUARD)

// device -> host on the context's own (non-blocking) stream, complete on return: never the legacy stream, which refuses work while ANY blocking
// stream of the process captures a graph (other contexts on other host threads do)
inline void copy_to_host(bark_context * c, void * dst, const void * src, size_t bytes) {
    HIP_OK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream));
    HIP_OK(hipStreamSynchronize(c->stream));
}

// rows [row0, ...) of a weight matrix behind a QMat handle (block formats or f32)
inline QMat q4_rows(const QMat & w, size_t row0, int K) {
    QMat r = w;
    if (w.qt == QT_F32) { r.qs = w.qs + row0 * (size_t) K * 4; return r; }
    const size_t nb = row0 * (size_t) (K / 32);
    r.d = w.d + nb; r.qs = w.qs + nb * (size_t) quant_formats()[w.qt].qs_bytes;
    if (w.m) r.m = w.m + nb;
    if (w.qh) r.qh = w.qh + nb;
    return r;
}

// ---- building blocks (engine.hip) ------------------------------------------------------------------
double weight_bytes_per_element(const GptModel & m);
float * layer_k(const GptModel & m, int l);
float * layer_v(const GptModel & m, int l);
float * layer_vt(const GptModel & m, int l);
// the many-row scratch a forward pass works in: the context's own (P rows), or the fine batch's (several windows back to back)
struct RowBufs { float * x, * q; half_t * xn, * att, * hbuf, * q16, * k16, * vt16; float * logits; const int32_t * tokens; int plane; };
RowBufs own_rows(bark_context * c);
bool fine_products_on_f16_mfma(const bark_context * c, const GptModel & m, bool causal);
// marks a context as running a lock-step job / a side-by-side fine pass for the lifetime of the object (bark_context::fine_order's default policy)
struct JobScope {
    bark_context * c; bool prev;
    explicit JobScope(bark_context * ctx) : c(ctx), prev(ctx->in_job) { ctx->in_job = true; }
    ~JobScope() { c->in_job = prev; }
    JobScope(const JobScope &) = delete; JobScope & operator=(const JobScope &) = delete;
};
// seq > 0: the N rows are N / seq independent sequences (fine windows), sequence z with its cache at kbase / vbase + z * kv_seq_stride
void run_layers_rows(bark_context * c, GptModel & m, int N, bool causal, float * kbase = nullptr, float * vbase = nullptr, int pos0 = 0,
                     const RowBufs * rb = nullptr, int seq = 0, size_t kv_seq_stride = 0, const SeqTab * seqtab = nullptr);
void run_layers_decode(bark_context * c, GptModel & m);
void run_lm_head(bark_context * c, GptModel & m, const float * xrow, int row0, int n_rows, int parity_rows, float out_div = 0.0f);
void set_state(bark_context * c, const StepState & st);
StepState get_state(bark_context * c);
StepState fresh_state();
void upload_tokens(bark_context * c, const int32_t * tok, size_t n);
void check_ids(const int32_t * tok, size_t n, int n_in, const char * what);
int run_prefill(bark_context * c, GptModel & m, int n_tokens, bool merge, float * kbase = nullptr, float * vbase = nullptr, int pos0 = 0);
struct StageCfg {            // what differs between the semantic and the coarse decode step
    int which; int mode; int lm_row0, lm_rows, parity_rows; int token_base; float min_eos_p; int eos_token; float temp;
};
StageCfg stage_cfg(bark_context * c, int which);
void run_sample(bark_context * c, const StageCfg & s, int n_past_add, bool prescaled = false);
void enqueue_decode_step(bark_context * c, const StageCfg & s, bool sample, int n_past_add, bool embed = true);
hipGraphExec_t capture_decode(bark_context * c, const StageCfg & s, int n_past_add, int n_steps = 1, int ng = 4);
void decode_steps_greedy(bark_context * c, const StageCfg & s, int n, int n_past);
int sample_host(std::vector<float> & l, std::mt19937 & rng, float temp, float * eos_p);
std::vector<float> fetch_logits(bark_context * c, size_t n);
void upload_uniforms(bark_context * c, int n);
void consume_uniforms(bark_context * c, int n_used);
void progress(bark_context * c, bark_encoding_step step, int pct);
void run_fine_forward(bark_context * c, int nn, int n_rows, const RowBufs * rb = nullptr, int Z = 1);
void ensure_fine_batch(bark_context * c, int Z);            // scratch of engine_fine_many for Z windows side by side
RowBufs fine_batch_rows(bark_context * c, int Z);

} }  // namespace barkhip::detail
