// engine_timing.hip - hipEvent-timed replays of single kernels / decode steps / fine passes for bench.py's roofline block.
#include "engine_internal.h"

#include <algorithm>
#include <climits>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <memory>
#include <stdexcept>

using namespace barkhip;
using namespace barkhip::detail;

namespace barkhip {

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
    if (m.vtcache) HIP_OK(hipMemsetAsync(m.vtcache, 0, m.kv_layer_stride * m.hp.n_layer * 4, c->stream));
    // the graph the stage loops replay: eight steps per launch; n_past does not advance here
    const int per_graph = 8;
    if (m.bench_graph) { (void) hipGraphExecDestroy(m.bench_graph); m.bench_graph = nullptr; }     // the variant depends on the context length
    m.bench_graph = capture_decode(c, s, 0, per_graph, (ctxlen + 255) / 256);
    for (int i = 0; i < 3; i++) HIP_OK(hipGraphLaunch(m.bench_graph, c->stream));
    set_state(c, st);
    iters = std::max(1, iters / per_graph) * per_graph;
    hipEvent_t e0, e1;
    HIP_OK(hipEventCreate(&e0)); HIP_OK(hipEventCreate(&e1));
    HIP_OK(hipEventRecord(e0, c->stream));
    for (int i = 0; i < iters; i += per_graph) {
        HIP_OK(hipGraphLaunch(m.bench_graph, c->stream));
        if (((i / per_graph) & 127) == 127) set_state(c, st);             // out_tokens holds 2048 entries
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
    if (which < 0 || which > 1 || op < 0 || op > 7) throw std::runtime_error("time_gemv: bad arguments");
    const bool hot = op >= 4;
    op &= 3;
    HIP_OK(hipSetDevice(c->device));
    GptModel & m = c->gpt[which];
    const int E = m.hp.n_embd, P = c->P;
    StepState st = fresh_state(); st.n_past = 100; st.cur_token = 1;
    set_state(c, st);
    HIP_OK(hipMemsetAsync(c->x, 0, (size_t) E * 4, c->stream));
    HIP_OK(hipMemsetAsync(c->att, 0, (size_t) E * 2, c->stream));
    HIP_OK(hipMemsetAsync(c->hbuf, 0, (size_t) 4 * E * 2, c->stream));
    if (m.q4) { HIP_OK(hipMemsetAsync(c->att32, 0, (size_t) E * 4, c->stream)); HIP_OK(hipMemsetAsync(c->h32, 0, (size_t) 4 * E * 4, c->stream)); }
    auto launch = [&](int l) {
        const GptModel::Layer & L = m.layers[(size_t) l];
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

#ifdef BARK_TRACE
// diagnostic build: `replays` consecutive replays of one decode step with per-wave time stamps; returns the record count
int engine_trace_decode_step(bark_context * c, int which, int ctxlen, int replays, unsigned long long * out6, int cap_records) {
    HIP_OK(hipSetDevice(c->device));
    GptModel & m = c->gpt[which];
    const StageCfg s = stage_cfg(c, which);
    StepState st = fresh_state(); st.n_past = ctxlen - 1; st.cur_token = 1;
    set_state(c, st);
    if (!c->trace_rec) {
        c->trace_cap = 1u << 19;
        c->trace_rec = dev_alloc<unsigned long long>(c, (size_t) c->trace_cap * 8);
        c->trace_pos = dev_alloc<unsigned>(c, 1);
    }
    HIP_OK(hipMemsetAsync(m.kcache, 0, m.kv_layer_stride * m.hp.n_layer * 4, c->stream));
    HIP_OK(hipMemsetAsync(m.vcache, 0, m.kv_layer_stride * m.hp.n_layer * 4, c->stream));
    if (m.vtcache) HIP_OK(hipMemsetAsync(m.vtcache, 0, m.kv_layer_stride * m.hp.n_layer * 4, c->stream));
    // first capture counts the waves of a step, the second bakes the per-replay stride into the kernels' arguments
    c->trace_kid = 0; c->trace_base = 0; c->trace_per_replay = 0;
    hipGraphExec_t g = capture_decode(c, s, 0, 1, (ctxlen + 255) / 256);
    (void) hipGraphExecDestroy(g);
    c->trace_per_replay = c->trace_base; c->trace_kid = 0; c->trace_base = 0;
    g = capture_decode(c, s, 0, 1, (ctxlen + 255) / 256);
    HIP_OK(hipMemsetAsync(c->trace_rec, 0, (size_t) c->trace_cap * 8 * sizeof(unsigned long long), c->stream));
    for (int i = 0; i < 5; i++) { HIP_OK(hipMemsetAsync(c->trace_pos, 0, sizeof(unsigned), c->stream)); HIP_OK(hipGraphLaunch(g, c->stream)); }
    HIP_OK(hipStreamSynchronize(c->stream));
    set_state(c, st);
    HIP_OK(hipMemsetAsync(c->trace_pos, 0, sizeof(unsigned), c->stream));
    for (int i = 0; i < replays; i++) HIP_OK(hipGraphLaunch(g, c->stream));
    HIP_OK(hipStreamSynchronize(c->stream));
    unsigned n = c->trace_per_replay * (unsigned) replays;
    n = std::min(n, std::min(c->trace_cap, (unsigned) std::max(0, cap_records)));
    copy_to_host(c, out6, c->trace_rec, (size_t) n * 8 * sizeof(unsigned long long));
    (void) hipGraphExecDestroy(g);
    return (int) n;
}
#endif

double engine_time_fine_pass(bark_context * c, int iters, double * flops_per_pass, int Z) {
    HIP_OK(hipSetDevice(c->device));
    GptModel & m = c->gpt[2];
    if (Z < 1 || Z > 32) throw std::runtime_error("time_fine_pass: 1..32 windows");
    std::vector<int32_t> buf((size_t) 8 * 1024 * Z);
    for (size_t i = 0; i < buf.size(); i++) buf[i] = (int32_t) ((i * 2654435761u) >> 22) & 1023;
    RowBufs rb = own_rows(c);
    if (Z > 1) {
        if (m.q4 || m.w32) throw std::runtime_error("time_fine_pass: several windows side by side need an f16 model file");
        ensure_fine_batch(c, Z);
        rb = fine_batch_rows(c, Z);
        HIP_OK(hipMemcpyAsync(c->fine_batch.tokens, buf.data(), buf.size() * 4, hipMemcpyHostToDevice, c->stream));
        HIP_OK(hipStreamSynchronize(c->stream));
    } else upload_tokens(c, buf.data(), buf.size());
    // Z > 1 is the pass of a lock-step job (engine_fine_many): C1m products under the default policy of bark_context::fine_order
    std::unique_ptr<JobScope> job;
    if (Z > 1) job.reset(new JobScope(c));
    auto pass = [&](int nn) { if (Z > 1) run_fine_forward(c, nn, 1024, &rb, Z); else run_fine_forward(c, nn, 1024); };
    pass(4);
    HIP_OK(hipStreamSynchronize(c->stream));
    hipEvent_t e0, e1;
    HIP_OK(hipEventCreate(&e0)); HIP_OK(hipEventCreate(&e1));
    HIP_OK(hipEventRecord(e0, c->stream));
    for (int i = 0; i < iters; i++) pass(2 + i % 6);
    HIP_OK(hipEventRecord(e1, c->stream));
    HIP_OK(hipEventSynchronize(e1));
    float ms = 0.f;
    HIP_OK(hipEventElapsedTime(&ms, e0, e1));
    (void) hipEventDestroy(e0); (void) hipEventDestroy(e1);
    if (flops_per_pass) {
        const double E = m.hp.n_embd, L = m.hp.n_layer, N = 1024;
        *flops_per_pass = Z * (2.0 * N * (L * 12.0 * E * E + 1024.0 * E) + 4.0 * N * N * E * L);     // SURVEY.md 8(d), per window
    }
    return (double) ms * 1000.0 / std::max(1, iters);
}

}  // namespace barkhip
