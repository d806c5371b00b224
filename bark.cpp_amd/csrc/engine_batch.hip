// engine_batch.hip - bark_hip_generate_batch: several utterances in lock step on one context.
#include "engine_internal.h"

#include <algorithm>
#include <climits>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <deque>
#include <exception>
#include <mutex>
#include <stdexcept>
#include <thread>

using namespace barkhip;
using namespace barkhip::detail;

namespace barkhip {

// ---------------------------------------------------------------------------------------------------
// batched decode: B utterances advance in lock step through the semantic and coarse decode loops; every decode
// kernel processes all slots (weights are read from HBM once per step instead of once per utterance); the prompts of all
// slots go through the model in one pass (batch_prefill_many), the fine windows of several utterances side by side
// (engine_fine_many, engine.hip), the codec of all utterances in one pass (engine_codec.hip).  Per-slot arithmetic is
// exactly the single-utterance one.
// ---------------------------------------------------------------------------------------------------
namespace {

constexpr int kMaxSlots = 64;                            // lock-step slots of a context (KV caches of both causal models per slot: 151 MB at bark-small)

// Lock steps over few live slots (f16 files, n_embd <= 1024, block_size 1024) do not use the matrix-core tiles of gemm_slots16_kernel / the one-CU
// attention of attn_fused_kernel.  Measured on MI355X (graph-replayed coarse step at context 640, bark-small, us; profiles/r05_few_slot_routes_round4_kernels.txt):
//     live slots                                   2     4     6     8    12    16    24    32
//     matrix-core / per-pair route (round 4)     390   477   521   434   506   555   688   753
//     per-slot products only                     330   389   414   395   471   512   682   787
//     + partial scores where q is born           277   319   362   388   478   544   761   892
// so: up to kFewSlotsScores live slots the QKV product runs per slot and forms the partial attention scores of the cached keys where q is born
// (gemv_ln_slots_ps_kernel, attention on them: attn_fused_ps_kernel); up to kFewSlotsProducts the FC product and the two out-projections run per slot
// on the VALU (weights out of the XCD's L2 after the first slot).  A slot-group form (8 slots inside the workgroup, weights requested once) was built
// and measured in round 5 too: slower at every size (494 us at 8 slots: its epilogues serialise per slot; profiles/r05_few_slot_routes_slot_group_kernels.txt).
// BARK_HIP_FEW_SLOTS=<products>[,<scores>] moves the cross-overs (0: off) - the A/B handle of tools/r05_sweep.py and of the route-equality test.
constexpr int kFewSlotsProducts = 16, kFewSlotsScores = 8;
int few_slots_max(int which = 0) {
    static const std::pair<int, int> v = [] {
        const char * e = getenv("BARK_HIP_FEW_SLOTS");
        int n = kFewSlotsProducts, na = kFewSlotsScores;
        if (e) { n = na = atoi(e); if (const char * k = strchr(e, ',')) na = atoi(k + 1); }
        auto ok = [](int x) { return x >= 2 && x <= kMaxSlots ? x : 0; };
        return std::make_pair(ok(n), ok(na));
    }();
    return which ? v.second : v.first;
}

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
    bb.sc = dev_alloc<float>(c, (size_t) B * c->max_H * c->P);
    if (few_slots_max(1) > 0) bb.ps = dev_alloc<float>(c, (size_t) std::min(B, few_slots_max(1)) * c->max_H * 4 * c->P);
    if (c->any_q4) { bb.att32 = dev_alloc<float>(c, (size_t) B * E); bb.h32 = dev_alloc<float>(c, (size_t) B * 4 * E); }
    bb.slot_par = dev_alloc<float>(c, (size_t) 2 * B);               // [0, B): temperatures, [B, 2 B): min_eos_p
    c->h_slot_par.assign((size_t) 2 * B, 0.0f);
    // pinned landing zone of the per-window / per-poll read-back (ids of all slots + their states): read_back() below
    HIP_OK(hipHostMalloc((void **) &bb.h_ids, (size_t) B * 2048 * sizeof(int32_t), hipHostMallocDefault));
    HIP_OK(hipHostMalloc((void **) &bb.h_state, (size_t) B * sizeof(StepState), hipHostMallocDefault));
    HIP_OK(hipHostMalloc((void **) &bb.h_state_in, (size_t) B * sizeof(StepState), hipHostMallocDefault));
    bb.cap = B;
}

// How the host learns what the lock steps produced: the sampled ids of the live slots (rows of 2048) and their stage states, in one pair of copies
// into pinned memory behind a synchronisation of the stream (the copy engine reads device memory that no kernel is writing any more, and nothing
// depends on how the runtime stages a pageable destination).  Round 5 measured this form against round 4's (pageable destination, copies enqueued
// straight behind the lock steps) while hunting the r04 divergence: both forms read back exactly what the device holds (a second copy after one
// more synchronisation never differed in 800 stress iterations) - the divergence was on the device.
void read_back(bark_context * c, int B, std::vector<int32_t> * ids, std::vector<StepState> & st) {
    bark_context::Batch & bb = c->batch;
    st.resize((size_t) B);
    if (ids) ids->resize((size_t) B * 2048);
    HIP_OK(hipStreamSynchronize(c->stream));
    if (ids) HIP_OK(hipMemcpyAsync(bb.h_ids, bb.out_tokens, ids->size() * 4, hipMemcpyDeviceToHost, c->stream));
    HIP_OK(hipMemcpyAsync(bb.h_state, bb.state, sizeof(StepState) * B, hipMemcpyDeviceToHost, c->stream));
    HIP_OK(hipStreamSynchronize(c->stream));
    if (ids) memcpy(ids->data(), bb.h_ids, ids->size() * 4);
    memcpy(st.data(), bb.h_state, sizeof(StepState) * B);
}
// the slots' own sampling parameters (bark_hip_request_params): host mirror -> device
void upload_slot_params(bark_context * c) {
    bark_context::Batch & bb = c->batch;
    HIP_OK(hipMemcpyAsync(bb.slot_par, c->h_slot_par.data(), c->h_slot_par.size() * sizeof(float), hipMemcpyHostToDevice, c->stream));
    HIP_OK(hipStreamSynchronize(c->stream));
}
// 1: greedy slots among [slot0, slot0 + n), 2: sampled slots among them
int slot_kinds(const bark_context * c, int slot0, int n) {
    int k = 0;
    for (int b = slot0; b < slot0 + n; b++) k |= c->h_slot_par[(size_t) b] > 0.0f ? 2 : 1;
    return k;
}
// the sampler's arguments for slots [slot0, slot0 + nb) of the batch
SampleArgs slot_sample_args(bark_context * c, const StageCfg & s, const bark_context::Batch & bb, int slot0, int nb, int n_past_add) {
    const GptModel & m = c->gpt[s.which];
    SampleArgs sa;
    { static const int force_exact = getenv("BARK_HIP_EXACT_SAMPLING") ? atoi(getenv("BARK_HIP_EXACT_SAMPLING")) : 0; sa.force_exact = force_exact; }
    sa.logits = bb.logits + bb.ld_logits * (size_t) slot0; sa.n = s.lm_rows; sa.mode = s.mode; sa.min_eos_p = s.min_eos_p; sa.eos_token = s.eos_token;
    sa.token_base = s.token_base; sa.n_past_add = n_past_add; sa.out_tokens = bb.out_tokens + (size_t) slot0 * 2048;
    sa.eos_trace = s.mode == 0 ? bb.eos_trace + (size_t) slot0 * 2048 : nullptr;
    sa.st = bb.state + slot0; sa.nbatch = nb; sa.ld_logits = (int) bb.ld_logits; sa.out_stride = 2048;
    sa.temp = s.temp; sa.u = bb.u + (size_t) slot0 * 8192; sa.u_stride = 8192;
    // the parameter arrays belong to the context's own batch (`bb` may be a slot_view of it): slot b's entries sit at [b] and [cap + b]
    const int first = (int) (bb.state - c->batch.state) + slot0;
    sa.slot_temp = c->batch.slot_par + first; sa.slot_min_eos_p = s.mode == 0 ? c->batch.slot_par + c->batch.cap + first : nullptr;
    sa.kinds = slot_kinds(c, first, nb);
    sa.wte = m.wte[0]; sa.wte_q = m.wte_q[0]; sa.wpe = m.wpe; sa.E = m.hp.n_embd; sa.n_in = m.hp.n_in_vocab; sa.P = c->P; sa.x = bb.x + (size_t) slot0 * m.hp.n_embd;
    return sa;
}

void set_slot_state(bark_context * c, int slot, const StepState & st) {
    HIP_OK(hipMemcpyAsync(c->batch.state + slot, &st, sizeof(st), hipMemcpyHostToDevice, c->stream));
    HIP_OK(hipStreamSynchronize(c->stream));
}

// all slots: layers -> LM head -> greedy sample (+ embedding of the sampled token)
void enqueue_batch_step(bark_context * c, const StageCfg & s, int B, const bark_context::Batch & bb) {
    GptModel & m = c->gpt[s.which];
    const int E = m.hp.n_embd, H = m.hp.n_head, P = c->P;
    hipStream_t st = c->stream;
    // time line of a lock step (engine_profile_lock_step): an event behind every launch site, named
    auto mark = [&](const char * name) {
        if (!c->step_marks) return;
        hipEvent_t e; HIP_OK(hipEventCreate(&e)); HIP_OK(hipEventRecord(e, st));
        c->step_marks->emplace_back(name, e);
    };
    float * kc0 = bb.kc[s.which], * vc0 = bb.vc[s.which];
    const size_t slot = bb.slot_stride[s.which];
    // LayerNorm statistics: recomputed inside every GEMV wave for small batches (an extra launch costs ~2 us), hoisted into
    // ln_stats_kernel for large ones (measured cross-over on MI355X between 16 and 32 slots)
    const bool hoist = B >= 24 && !m.q4;
    // B >= 8, f16 weights: the products of the step run once for all slots on the f32 matrix cores (gemm_slots16_kernel, kernels.hip);
    // rows are normalised to f16 by ln_rows_kernel first.  Same C1 chains as the GEMV path: results do not depend on the route
    // (tools/check_routes.py, test_cross_check_routes_give_the_same_bits).  BARK_HIP_CROSSCHECK bit 1 forces the VALU
    // GEMV per pair of slots everywhere (the cross-check route, and the only one for fewer than 8 slots and for quantised files).
    const bool mfma = !(crosscheck_mask() & 2) && B >= 8 && !m.q4;
    auto product = [&](LinArgs & a, const float * ln_g, const float * ln_b) {
        if (!mfma) { a.ln_g = ln_g; a.ln_b = ln_b; launch_linear(st, a); return; }
        // the two out-projections (input rows f16 already, 768 output rows): the VALU GEMV per pair of slots stays ahead for few slots
        // (tools/time_slots.py, us per launch VALU / matrix cores: proj 2.9 / 3.5 at 8 slots, 3.3 / 3.7 at 16, 4.4 / 3.7 at 32; MLP proj 5.1 / 8.0
        // at 8, 9.0 / 8.8 at 16, 12.7 / 8.9 at 32)
        if (!ln_g && (a.K == a.M ? B < 24 : B < 16)) { launch_linear(st, a); return; }
        // LayerNorm of the slot rows: inside the product kernel (BARK_HIP_CROSSCHECK bit 7 keeps the launch of its own, the cross-check route)
        // - measured per launch (QKV, small): 8 slots 5.7 us fused against 4.0 + 1.9 separate, 32 slots 8.7 against 5.7 + 2.0 (two slot tiles: twice
        // the workgroups repeat the LayerNorm), whole batches +2 % at 8 slots, -1 % at 32 (profiles/r03_ln_fused_slots.txt): fused for one slot tile
        if (ln_g && linear_slots_fuses_ln(a.K) && B <= 16 && !(crosscheck_mask() & 128)) { a.ln_g = ln_g; a.ln_b = ln_b; }
        else if (ln_g) { launch_ln_rows(st, a.x_f32, B, a.K, ln_g, ln_b, c->xn); a.x_f16 = c->xn; a.x_f32 = nullptr; }
        a.ln_stats = nullptr;
        launch_linear_slots(st, a);
    };
    // few live slots: per-slot products, partial scores where q is born + the attention on them (see few_slots_max)
    const bool few_ok = !m.q4 && !m.w32 && P == 1024 && E <= 1024 && (E & 127) == 0 && B >= 2 && !(crosscheck_mask() & 2);
    const bool few = few_ok && B <= few_slots_max(0);
    const bool slot_ps = few_ok && bb.ps && B <= few_slots_max(1);
    for (int l = 0; l < m.hp.n_layer; l++) {
        const GptModel::Layer & L = m.layers[(size_t) l];
        float * kl = kc0 + m.kv_layer_stride * (size_t) l, * vl = vc0 + m.kv_layer_stride * (size_t) l;
        if (hoist && !mfma) launch_ln_stats(st, bb.x, B, E, bb.ln_stats);
        LinArgs a;
        a.batched = 1; a.nbatch = B; a.kv_slot_stride = slot; a.ln_stats = hoist ? bb.ln_stats : nullptr;
        a.W = L.attn_w; a.wq = L.attn_q; a.M = 3 * E; a.K = E; a.N = 1; a.x_f32 = bb.x; a.bias = L.attn_b;
        a.epi = EPI_QKV; a.q = bb.q; a.kc = kl; a.vc = vl; a.E = E; a.P = P; a.pos0 = 0; a.st = bb.state;
        if (slot_ps) { a.ln_g = L.ln1_g; a.ln_b = L.ln1_b; a.ln_stats = nullptr; a.ps = bb.ps; launch_linear_slots_ps(st, a); }
        else product(a, L.ln1_g, L.ln1_b);
        mark("ln1+qkv");
        AttnDecodeArgs at;
        at.q = bb.q; at.kc = kl; at.vc = vl; at.H = H; at.P = P; at.st = bb.state; at.att = bb.att;
        at.nbatch = B; at.kv_slot_stride = slot; at.att32 = m.q4 ? bb.att32 : nullptr;
        at.sc = bb.sc;
        if (slot_ps) at.ps = bb.ps;
        launch_attn_decode(st, at);
        mark("attention");
        LinArgs p;
        p.batched = 1; p.nbatch = B;
        p.W = L.proj_w; p.wq = L.proj_q; p.M = E; p.K = E; p.N = 1; if (m.q4) p.x_f32 = bb.att32; else p.x_f16 = bb.att; p.bias = L.proj_b; p.epi = EPI_RESID; p.res = bb.x;
        if (few) launch_linear_slots_gemv(st, p); else product(p, nullptr, nullptr);
        mark("proj");
        if (hoist && !mfma) launch_ln_stats(st, bb.x, B, E, bb.ln_stats);
        LinArgs f;
        f.batched = 1; f.nbatch = B; f.ln_stats = hoist ? bb.ln_stats : nullptr;
        f.W = L.fc_w; f.wq = L.fc_q; f.M = 4 * E; f.K = E; f.N = 1; f.x_f32 = bb.x; f.bias = L.fc_b;
        f.epi = EPI_GELU; f.out_h = bb.h; f.out_h32 = m.q4 ? bb.h32 : nullptr; f.lut = c->d_gelu_lut;
        if (few) { f.ln_g = L.ln2_g; f.ln_b = L.ln2_b; f.ln_stats = nullptr; f.E = E; launch_linear_slots_ps(st, f); }
        else product(f, L.ln2_g, L.ln2_b);
        mark("ln2+fc+gelu");
        LinArgs o;
        o.batched = 1; o.nbatch = B;
        o.W = L.mproj_w; o.wq = L.mproj_q; o.M = E; o.K = 4 * E; o.N = 1; if (m.q4) o.x_f32 = bb.h32; else o.x_f16 = bb.h; o.bias = L.mproj_b; o.epi = EPI_RESID; o.res = bb.x;
        if (few) launch_linear_slots_gemv(st, o); else product(o, nullptr, nullptr);
        mark("mlp_proj");
    }
    if (hoist && !mfma) launch_ln_stats(st, bb.x, B, E, bb.ln_stats);
    LinArgs h;
    h.batched = 1; h.nbatch = B; h.ln_stats = hoist ? bb.ln_stats : nullptr;
    if (m.q4) h.wq = q4_rows(m.lm_head_q[0], (size_t) s.lm_row0, E); else h.W = m.lm_head[0] + (size_t) s.lm_row0 * E;
    h.M = s.lm_rows; h.K = E; h.N = 1; h.x_f32 = bb.x;
    h.epi = EPI_LOGITS; h.out = bb.logits; h.ld_out = (int) bb.ld_logits; h.parity_rows = s.parity_rows; h.st = bb.state;
    product(h, m.lnf_g, m.lnf_b);
    mark("lnf+lm_head");
    launch_sample_greedy(st, slot_sample_args(c, s, bb, 0, B, 1));
    mark("sample+embed");
}

void batch_step(bark_context * c, const StageCfg & s, int B) {
    bark_context::Batch & bb = c->batch;
    if (c->use_graph) {
        // one captured lock step per (model, active slots, kinds of sampling among them); a batch that shrinks slot by slot meets each
        // size once per context (the executables are kept)
        const int key = s.which | (B << 1) | (slot_kinds(c, 0, B) << 12);
        hipGraphExec_t & exec = c->batch_graphs[key];
        if (!exec) {
            hipGraph_t graph = nullptr;
            HIP_OK(hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
            try { enqueue_batch_step(c, s, B, bb); }
            catch (...) { hipGraph_t g2 = nullptr; (void) hipStreamEndCapture(c->stream, &g2); if (g2) (void) hipGraphDestroy(g2); throw; }
            HIP_OK(hipStreamEndCapture(c->stream, &graph));
            HIP_OK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
            (void) hipGraphDestroy(graph);
        }
        HIP_OK(hipGraphLaunch(exec, c->stream));
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
    launch_sample_greedy(c->stream, slot_sample_args(c, s, bb, slot, 1, N));
}


// The prompts of several slots through the model in ONE pass: sequence z = slot slots[z] evaluates ids[z][L[z] ..] behind the L[z] rows its
// cache already holds.  Every activation holds the sequences back to back, `seq` rows each (the longest, rounded up; shorter ones are
// padded - padding rows store nothing); the products see all rows at once (thousands instead of a few hundred: whole rounds of tiles),
// the causal attention runs per sequence against the slot's own cache.  Row for row the arithmetic of batch_prefill_and_sample.
void batch_prefill_many(bark_context * c, const StageCfg & s, const std::vector<int> & slots, const std::vector<const std::vector<int32_t> *> & ids,
                        const std::vector<int> & Ls, bool merge, const std::vector<int> & step0) {
    GptModel & m = c->gpt[s.which];
    bark_context::Batch & bb = c->batch;
    const int Z = (int) slots.size(), E = m.hp.n_embd, P = c->P;
    if (Z == 0) return;
    if (!bb.pf_x) {
        const size_t R = (size_t) bb.cap * P, ME = (size_t) c->max_E;
        bb.pf_x = dev_alloc<float>(c, R * ME); bb.pf_q = dev_alloc<float>(c, R * ME);
        bb.pf_xn = dev_alloc<half_t>(c, R * ME); bb.pf_att = dev_alloc<half_t>(c, R * ME); bb.pf_h = dev_alloc<half_t>(c, R * ME * 4);
        bb.pf_tokens = dev_alloc<int32_t>(c, R); bb.pf_tab = dev_alloc<SeqTab>(c, (size_t) bb.cap);
    }
    std::vector<SeqTab> tab((size_t) Z);
    int max_rows = 0, tok_stride = 0;
    for (int z = 0; z < Z; z++) {
        check_ids(ids[(size_t) z]->data(), ids[(size_t) z]->size(), m.hp.n_in_vocab, "batch prefill");
        const int n_tok = (int) ids[(size_t) z]->size() - Ls[(size_t) z], rows = merge ? n_tok - 256 : n_tok;
        if (rows < 1 || Ls[(size_t) z] + rows > P || (merge && Ls[(size_t) z])) throw std::runtime_error("batch prefill: bad prompt length");
        tab[(size_t) z] = SeqTab{slots[(size_t) z], Ls[(size_t) z], rows, 0};
        max_rows = std::max(max_rows, rows); tok_stride = std::max(tok_stride, n_tok);
    }
    const int seq = std::min(P, (max_rows + 63) & ~63);
    if ((size_t) Z * (size_t) std::max(seq, tok_stride) > (size_t) bb.cap * P) throw std::runtime_error("batch prefill: scratch too small");
    std::vector<int32_t> tok((size_t) Z * tok_stride, 0);
    std::vector<StepState> sts((size_t) Z, fresh_state());
    for (int z = 0; z < Z; z++) {
        std::copy(ids[(size_t) z]->begin() + Ls[(size_t) z], ids[(size_t) z]->end(), tok.begin() + (size_t) z * tok_stride);
        sts[(size_t) z].step = step0[(size_t) z]; sts[(size_t) z].n_past = Ls[(size_t) z];
    }
    hipStream_t st = c->stream;
    HIP_OK(hipMemcpyAsync(bb.pf_tokens, tok.data(), tok.size() * 4, hipMemcpyHostToDevice, st));
    HIP_OK(hipMemcpyAsync(bb.pf_tab, tab.data(), tab.size() * sizeof(SeqTab), hipMemcpyHostToDevice, st));
    for (int z = 0; z < Z; z++) HIP_OK(hipMemcpyAsync(bb.state + slots[(size_t) z], &sts[(size_t) z], sizeof(StepState), hipMemcpyHostToDevice, st));
    HIP_OK(hipMemsetAsync(bb.pf_x, 0, (size_t) Z * seq * E * sizeof(float), st));          // padding rows: zeros (their results are never read)
    for (int z = 0; z < Z; z++) {
        EmbedArgs e;
        e.wte = m.wte[0]; e.wte_q = m.wte_q[0]; e.wpe = m.wpe; e.E = E; e.n_in = m.hp.n_in_vocab; e.P = P; e.tokens = bb.pf_tokens + (size_t) z * tok_stride;
        e.n_rows = tab[(size_t) z].len; e.merge = merge ? 1 : 0; e.pos0 = Ls[(size_t) z]; e.x = bb.pf_x + (size_t) z * seq * E;
        launch_embed_causal(st, e);
    }
    RowBufs rb; rb.x = bb.pf_x; rb.q = bb.pf_q; rb.xn = bb.pf_xn; rb.att = bb.pf_att; rb.hbuf = bb.pf_h; rb.q16 = rb.k16 = rb.vt16 = nullptr;
    rb.logits = nullptr; rb.tokens = nullptr; rb.plane = 0;
    run_layers_rows(c, m, Z * seq, true, bb.kc[s.which], bb.vc[s.which], 0, &rb, seq, bb.slot_stride[s.which], bb.pf_tab);
    for (int z = 0; z < Z; z++) {
        const int slot = slots[(size_t) z], N = tab[(size_t) z].len;
        LinArgs h;
        h.W = m.lm_head[0] + (size_t) s.lm_row0 * E; h.M = s.lm_rows; h.K = E; h.N = 1;
        h.x_f32 = bb.pf_x + ((size_t) z * seq + (size_t) (N - 1)) * E; h.ln_g = m.lnf_g; h.ln_b = m.lnf_b; h.epi = EPI_LOGITS;
        h.out = bb.logits + bb.ld_logits * (size_t) slot; h.ld_out = (int) bb.ld_logits; h.parity_rows = s.parity_rows; h.st = bb.state + slot;
        launch_linear(st, h);
        launch_sample_greedy(st, slot_sample_args(c, s, bb, slot, 1, N));
    }
    HIP_OK(hipStreamSynchronize(st));                           // tok / tab / sts are stack objects
}

}  // namespace

// One lock-step decode kernel for B slots, launched back to back (hipGraph of 48 nodes) while rotating through the layers' weights.
// op: 0 QKV, 1 attention out-proj, 2 FC + GELU, 3 MLP out-proj, 4 LayerNorm of the B rows to f16, 5 attention of every slot at context
// `ctxlen`.  kind: route of the products - 0 the VALU GEMV (LayerNorm fused for ops 0 / 2), >= 1 launch_linear_slots(kind) on rows
// that are already normalised.  Returns the average device time per launch in microseconds.
double engine_time_slots(bark_context * c, int which, int op, int B, int kind, int ctxlen, int iters) {
    if (which < 0 || which > 1 || op < 0 || op > 5 || B < 1 || B > kMaxSlots) throw std::runtime_error("time_slots: bad arguments");
    HIP_OK(hipSetDevice(c->device));
    GptModel & m = c->gpt[which];
    if (m.q4 || c->any_w32) throw std::runtime_error("time_slots: f16 model files only");
    ensure_batch(c, c->batch.cap ? c->batch.cap : std::max(B, 8));
    bark_context::Batch & bb = c->batch;
    if (B > bb.cap) throw std::runtime_error("time_slots: batch larger than the capacity fixed by the first call");
    const int E = m.hp.n_embd, H = m.hp.n_head, P = c->P;
    ctxlen = std::max(2, std::min(ctxlen, m.hp.block_size));
    std::vector<StepState> sts((size_t) B, fresh_state());
    for (auto & v : sts) { v.n_past = ctxlen - 1; v.cur_token = 1; }
    HIP_OK(hipMemcpyAsync(bb.state, sts.data(), sizeof(StepState) * B, hipMemcpyHostToDevice, c->stream));
    HIP_OK(hipMemsetAsync(bb.x, 0, (size_t) B * E * 4, c->stream));
    HIP_OK(hipMemsetAsync(bb.q, 0, (size_t) B * E * 4, c->stream));
    HIP_OK(hipMemsetAsync(bb.att, 0, (size_t) B * E * 2, c->stream));
    HIP_OK(hipMemsetAsync(bb.h, 0, (size_t) B * 4 * E * 2, c->stream));
    HIP_OK(hipMemsetAsync(bb.ln_stats, 0, (size_t) B * 2 * 4, c->stream));
    HIP_OK(hipMemsetAsync(c->xn, 0, (size_t) B * 4 * E * 2, c->stream));
    if (op == 5) {
        HIP_OK(hipMemsetAsync(bb.kc[which], 0, bb.slot_stride[which] * (size_t) B * 4, c->stream));
        HIP_OK(hipMemsetAsync(bb.vc[which], 0, bb.slot_stride[which] * (size_t) B * 4, c->stream));
    }
    const size_t slot = bb.slot_stride[which];
    auto launch = [&](int l) {
        const GptModel::Layer & L = m.layers[(size_t) l];
        float * kl = bb.kc[which] + m.kv_layer_stride * (size_t) l, * vl = bb.vc[which] + m.kv_layer_stride * (size_t) l;
        if (op == 4) { launch_ln_rows(c->stream, bb.x, B, E, L.ln1_g, L.ln1_b, c->xn); return; }
        if (op == 5) {
            AttnDecodeArgs at;
            at.q = bb.q; at.kc = kl; at.vc = vl; at.H = H; at.P = P; at.st = bb.state; at.att = bb.att;
            at.nbatch = B; at.kv_slot_stride = slot;
            if (kind != 0) at.sc = bb.sc;                        // kind 0: the one-workgroup-per-(head, slot) kernel, for the A/B
            launch_attn_decode(c->stream, at);
            return;
        }
        LinArgs a;
        a.batched = 1; a.nbatch = B; a.kv_slot_stride = slot; a.N = 1;
        switch (op) {
            case 0: a.W = L.attn_w; a.M = 3 * E; a.K = E; a.bias = L.attn_b; a.epi = EPI_QKV; a.q = bb.q; a.kc = kl; a.vc = vl; a.E = E; a.P = P; a.st = bb.state;
                    if (kind == 0) { a.x_f32 = bb.x; a.ln_g = L.ln1_g; a.ln_b = L.ln1_b; a.ln_stats = B >= 24 ? bb.ln_stats : nullptr; }
                    else if (kind == 6) { a.x_f32 = bb.x; a.ln_g = L.ln1_g; a.ln_b = L.ln1_b; }          // LayerNorm fused into the matrix-core product
                    else a.x_f16 = c->xn;
                    break;
            case 1: a.W = L.proj_w; a.M = E; a.K = E; a.x_f16 = bb.att; a.bias = L.proj_b; a.epi = EPI_RESID; a.res = bb.x; break;
            case 2: a.W = L.fc_w; a.M = 4 * E; a.K = E; a.bias = L.fc_b; a.epi = EPI_GELU; a.out_h = bb.h; a.lut = c->d_gelu_lut;
                    if (kind == 0) { a.x_f32 = bb.x; a.ln_g = L.ln2_g; a.ln_b = L.ln2_b; a.ln_stats = B >= 24 ? bb.ln_stats : nullptr; }
                    else if (kind == 6) { a.x_f32 = bb.x; a.ln_g = L.ln2_g; a.ln_b = L.ln2_b; }
                    else a.x_f16 = c->xn;
                    break;
            default: a.W = L.mproj_w; a.M = E; a.K = 4 * E; a.x_f16 = bb.h; a.bias = L.mproj_b; a.epi = EPI_RESID; a.res = bb.x; break;
        }
        if (kind == 0) launch_linear(c->stream, a); else launch_linear_slots(c->stream, a);
    };
    const int per_graph = 48;
    hipGraph_t graph = nullptr; hipGraphExec_t exec = nullptr;
    HIP_OK(hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
    try { for (int i = 0; i < per_graph; i++) launch(i % m.hp.n_layer); }
    catch (...) { hipGraph_t g2 = nullptr; (void) hipStreamEndCapture(c->stream, &g2); if (g2) (void) hipGraphDestroy(g2); throw; }
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
    return (double) ms * 1000.0 / (reps * per_graph);
}

// Time line of ONE lock step over B slots at context `ctxlen`: the step is enqueued eagerly `reps` times with an event behind every launch
// site of enqueue_batch_step; out[i] = {site, average microseconds from the previous event to this one} - kernel time plus the gap in front
// of it, in launch order (the profiler view of the lock-step path: rocprofv3 cannot follow bark_hip_generate_batch, profiles/README.md).
// which: 0 semantic, 1 coarse.  The step's total is also returned as the last entry "step (graph replay)": the same step replayed from a hipGraph.
void engine_profile_lock_step(bark_context * c, int which, int B, int ctxlen, int reps, std::vector<std::pair<std::string, double>> & out) {
    if (which < 0 || which > 1 || B < 1 || B > kMaxSlots || reps < 1) throw std::runtime_error("profile_lock_step: bad arguments");
    HIP_OK(hipSetDevice(c->device));
    GptModel & m = c->gpt[which];
    ensure_batch(c, c->batch.cap ? c->batch.cap : std::max(B, 8));
    bark_context::Batch & bb = c->batch;
    if (B > bb.cap) throw std::runtime_error("profile_lock_step: more slots than the capacity fixed by the first call");
    const StageCfg s = stage_cfg(c, which);
    ctxlen = std::max(2, std::min(ctxlen, m.hp.block_size - reps - 40));
    auto reset = [&] {
        std::vector<StepState> sts((size_t) B, fresh_state());
        for (auto & v : sts) { v.n_past = ctxlen - 1; v.cur_token = 1; }
        HIP_OK(hipMemcpyAsync(bb.state, sts.data(), sizeof(StepState) * B, hipMemcpyHostToDevice, c->stream));
        HIP_OK(hipStreamSynchronize(c->stream));
    };
    HIP_OK(hipMemsetAsync(bb.x, 0, (size_t) B * m.hp.n_embd * 4, c->stream));
    HIP_OK(hipMemsetAsync(bb.kc[which], 0, bb.slot_stride[which] * (size_t) B * 4, c->stream));
    HIP_OK(hipMemsetAsync(bb.vc[which], 0, bb.slot_stride[which] * (size_t) B * 4, c->stream));
    for (int b = 0; b < B; b++) { c->h_slot_par[(size_t) b] = 0.0f; c->h_slot_par[(size_t) bb.cap + b] = 0.2f; }
    upload_slot_params(c);
    reset();
    enqueue_batch_step(c, s, B, bb);                               // warm-up (first-use costs)
    reset();
    std::vector<std::pair<std::string, hipEvent_t>> marks;
    std::vector<double> acc;
    std::vector<std::string> names;
    for (int r = 0; r < reps; r++) {
        marks.clear();
        hipEvent_t e0; HIP_OK(hipEventCreate(&e0)); HIP_OK(hipEventRecord(e0, c->stream));
        c->step_marks = &marks;
        try { enqueue_batch_step(c, s, B, bb); } catch (...) { c->step_marks = nullptr; throw; }
        c->step_marks = nullptr;
        HIP_OK(hipStreamSynchronize(c->stream));
        if (acc.empty()) { acc.assign(marks.size(), 0.0); for (auto & mk : marks) names.push_back(mk.first); }
        hipEvent_t prev = e0;
        for (size_t i = 0; i < marks.size(); i++) {
            float ms = 0.f; HIP_OK(hipEventElapsedTime(&ms, prev, marks[i].second));
            acc[i] += (double) ms * 1000.0;
            prev = marks[i].second;
        }
        (void) hipEventDestroy(e0);
        for (auto & mk : marks) (void) hipEventDestroy(mk.second);
    }
    out.clear();
    for (size_t i = 0; i < acc.size(); i++) out.emplace_back(names[i], acc[i] / reps);
    // the same step from a hipGraph (what the stage loops replay)
    reset();
    batch_step(c, s, B);
    HIP_OK(hipStreamSynchronize(c->stream));
    reset();
    hipEvent_t g0, g1; HIP_OK(hipEventCreate(&g0)); HIP_OK(hipEventCreate(&g1));
    HIP_OK(hipEventRecord(g0, c->stream));
    for (int r = 0; r < reps; r++) batch_step(c, s, B);
    HIP_OK(hipEventRecord(g1, c->stream));
    HIP_OK(hipEventSynchronize(g1));
    float ms = 0.f; HIP_OK(hipEventElapsedTime(&ms, g0, g1));
    (void) hipEventDestroy(g0); (void) hipEventDestroy(g1);
    out.emplace_back("step (graph replay)", (double) ms * 1000.0 / reps);
}

namespace { bool ensure_tail_context(bark_context * c); bool tail_stream_enabled(); }

void engine_reserve_batch(bark_context * c, int slots) {
    HIP_OK(hipSetDevice(c->device));
    if (slots < 1 || slots > kMaxSlots) throw std::runtime_error("reserve_batch: 1..64 slots");
    if (c->gpt[0].hp.n_embd != c->gpt[1].hp.n_embd || c->any_w32) return;          // these contexts run batches sequentially
    ensure_batch(c, std::max(slots, 8));
    // the clone the tail of a job runs on: made here, off the hot path, instead of inside the first job
    if (slots > 1 && tail_stream_enabled() && !c->host_sampling) (void) ensure_tail_context(c);
}

namespace {

// one utterance of a job on its way through the stages
struct Utt {
    std::string text;
    bark_hip_request_params rp{};
    std::mt19937 rng;
    std::vector<int32_t> coarse_out;                     // raw coarse ids of the windows done so far
    std::vector<int32_t> cached;                         // coarse: ids whose K / V rows sit in the utterance's slot cache
    int n_steps = 0, step_idx = 0;                       // coarse steps in total / done
    int issued = 0, cap = 0;                             // semantic: lock steps run for it / its step cap
};

// slot `from` becomes slot `to` (the batch stays a compact range of slots when an utterance in its middle retires): caches of model g,
// the decode row, the stage state, the sampled ids, the uniform draws
void move_slot(bark_context * c, int g, int from, int to) {
    bark_context::Batch & bb = c->batch;
    const GptModel & m = c->gpt[g];
    hipStream_t st = c->stream;
    const size_t ss = bb.slot_stride[g];
    HIP_OK(hipMemcpyAsync(bb.kc[g] + ss * (size_t) to, bb.kc[g] + ss * (size_t) from, ss * sizeof(float), hipMemcpyDeviceToDevice, st));
    HIP_OK(hipMemcpyAsync(bb.vc[g] + ss * (size_t) to, bb.vc[g] + ss * (size_t) from, ss * sizeof(float), hipMemcpyDeviceToDevice, st));
    HIP_OK(hipMemcpyAsync(bb.x + (size_t) to * m.hp.n_embd, bb.x + (size_t) from * m.hp.n_embd, (size_t) m.hp.n_embd * sizeof(float), hipMemcpyDeviceToDevice, st));
    HIP_OK(hipMemcpyAsync(bb.state + to, bb.state + from, sizeof(StepState), hipMemcpyDeviceToDevice, st));
    HIP_OK(hipMemcpyAsync(bb.out_tokens + (size_t) to * 2048, bb.out_tokens + (size_t) from * 2048, 2048 * sizeof(int32_t), hipMemcpyDeviceToDevice, st));
    HIP_OK(hipMemcpyAsync(bb.eos_trace + (size_t) to * 2048, bb.eos_trace + (size_t) from * 2048, 2048 * sizeof(float), hipMemcpyDeviceToDevice, st));
    HIP_OK(hipMemcpyAsync(bb.u + (size_t) to * 8192, bb.u + (size_t) from * 8192, 8192 * sizeof(double), hipMemcpyDeviceToDevice, st));
    c->h_slot_par[(size_t) to] = c->h_slot_par[(size_t) from];
    c->h_slot_par[(size_t) bb.cap + to] = c->h_slot_par[(size_t) bb.cap + from];
}

void set_slot_params(bark_context * c, int slot, const Utt & u) {
    c->h_slot_par[(size_t) slot] = u.rp.temp;
    c->h_slot_par[(size_t) c->batch.cap + slot] = u.rp.min_eos_p;
}


// The clone a job's tail runs on (JobTail below): a second set of KV caches, scratch, fine-batch and codec buffers on a stream of its own, kept
// until bark_free - i.e. a context that runs lock-step jobs holds about twice the run-time memory of one that does not.  Created by
// bark_hip_reserve_batch / the request collector (explicit, off the hot path) or by the first job with more than one utterance.
// false: no clone could be made (said once on stderr); the job keeps its tail on its own stream.
// BARK_HIP_TAIL_STREAM=0 keeps the tail behind the coarse stage on the job's own stream; read per call: tests flip it
bool tail_stream_enabled() { const char * e = getenv("BARK_HIP_TAIL_STREAM"); return !(e && !strcmp(e, "0")); }

bool ensure_tail_context(bark_context * c) {
    if (c->tail) return true;
    try {
        c->tail = engine_clone(c, 0);
        // the helper's stream yields to the decode chain: lowest priority (workgroups of the chain's small kernels are dispatched first whenever a CU
        // frees up).  (Round 4 also tried confining it to a CU mask - hipExtStreamCreateWithCUMask yields a BLOCKING stream, unusable beside threads
        // that capture graphs - and normal priority; neither paid, both knobs are gone.)
        {
            HIP_OK(hipStreamSynchronize(c->tail->stream));
            HIP_OK(hipStreamDestroy(c->tail->stream)); c->tail->stream = nullptr;
            int least = 0, greatest = 0;
            HIP_OK(hipDeviceGetStreamPriorityRange(&least, &greatest));
            HIP_OK(hipStreamCreateWithPriority(&c->tail->stream, hipStreamNonBlocking, least));
        }
        return true;
    } catch (const std::exception & e) {
        // e.g. no memory for a second set of caches and scratch
        static bool said = false;
        if (!said) { fprintf(stderr, "bark-hip: no second stream for the tail of lock-step jobs (%s)\n", e.what()); said = true; }
        delete c->tail; c->tail = nullptr;
        return false;
    }
}

// The tail of a job: fine passes (bark.cpp:1961-2059) and codec (bark.cpp:2143-2167) of the utterances that have left the coarse stage.
// With a second context (a clone: own non-blocking stream, scratch and graphs, the same weights) it runs on a helper thread WHILE the lock
// steps of the remaining utterances go on: a lock step is a chain of ~100 small dependent kernels that leaves most of the chip idle, the fine
// passes are wide matrix-core kernels (tools/staggered_jobs.py: two job streams a second apart deliver 14 % more than one).  Utterances
// arrive when their last coarse window is done; the helper takes up to `chunk` of one fine temperature at a time (their windows side by side,
// engine_fine_many), the codec runs over what has accumulated whenever nobody waits for a fine pass (at most 32 per pass).  Per-utterance
// results do not depend on the grouping (every utterance is checked against its own oracle run, tests/test_gpu_batch_ragged.py).
// inline_ctx: BARK_HIP_TAIL_STREAM=0 - the same steps on the job's own context after the coarse stage (the A/B arm, and the only form for
// host-side sampling).  The progress callback stays on the calling thread (bark.h contract): the helper reports nothing.
struct JobTail {
    bark_context * c;                                            // the job's context: parameters, results
    bark_context * t;                                            // the context the tail runs on (c itself: inline)
    std::vector<Utt> & us;
    std::mutex mu; std::condition_variable cv;
    std::deque<int> pending;                                     // utterances whose coarse ids are complete
    bool closed = false, abandon = false;
    std::exception_ptr err;
    std::thread th;
    std::vector<int> codec_wait;
    int good = 0;
    float saved_fine_temp = 0.0f;
    std::unique_ptr<JobScope> tail_scope;                        // the clone runs the job's fine passes: same order of the products as the job's context

    JobTail(bark_context * job, std::vector<Utt> & utts, bool second_stream) : c(job), t(job), us(utts) {
        if (second_stream && !ensure_tail_context(c)) second_stream = false;
        if (second_stream) {
            t = c->tail;
            // its captured fine passes bake the fine temperature (bark_hip_set_params on the job's context drops them too: engine_invalidate_graphs)
            if (t->params.fine_temp != c->params.fine_temp) drop_fine_graphs_of(t);
            t->params = c->params;
            t->params.progress_callback = nullptr; t->params.progress_callback_user_data = nullptr;
            const int64_t t_load = t->stats.t_load_us;
            t->stats = bark_hip_stats{}; t->stats.t_load_us = t_load;
        }
        saved_fine_temp = t->params.fine_temp;
        if (t != c) { t->fine_order = c->fine_order; tail_scope.reset(new JobScope(t)); th = std::thread([this] { run(); }); }
    }
    ~JobTail() {
        { std::lock_guard<std::mutex> g(mu); closed = true; abandon = true; }
        cv.notify_all();
        if (th.joinable()) th.join();
        tail_scope.reset();
    }
    static void drop_fine_graphs_of(bark_context * x) { for (auto & g : x->fine_graphs) if (g) { (void) hipGraphExecDestroy(g); g = nullptr; } }
    void drop_fine_graphs() { drop_fine_graphs_of(t); }
    void push(const std::vector<int> & utts) {
        if (utts.empty()) return;
        { std::lock_guard<std::mutex> g(mu); for (int b : utts) pending.push_back(b); }
        cv.notify_all();
    }
    // everything pushed so far is carried through; rethrows what the helper met
    void finish() {
        { std::lock_guard<std::mutex> g(mu); closed = true; }
        cv.notify_all();
        if (t != c) { if (th.joinable()) th.join(); }
        else run();
        if (err) std::rethrow_exception(err);
        if (t != c) {
            c->stats.n_sample_fine += t->stats.n_sample_fine; c->stats.n_near_tie += t->stats.n_near_tie;
            c->stats.t_fine_us += t->stats.t_fine_us; c->stats.t_codec_us += t->stats.t_codec_us;
            c->stats.n_frames += t->stats.n_frames; c->stats.n_semantic += t->stats.n_semantic; c->stats.n_samples += t->stats.n_samples;
            c->stats.graph_replays += t->stats.graph_replays;
        }
    }
    void fine_chunk(const std::vector<int> & take, bool many) {
        const int64_t t0 = now_us();
        const float ft = us[(size_t) take[0]].rp.fine_temp;
        if (t->params.fine_temp != ft) {
            // the per-utterance fine loop replays one captured forward pass + pick per codebook, and a capture bakes the kind of pick and its
            // temperature: a chunk with another fine temperature needs fresh ones
            drop_fine_graphs();
            t->params.fine_temp = ft;
        }
        if (many) {
            std::vector<const std::vector<int32_t> *> co;
            std::vector<std::mt19937> rr;
            for (int b : take) { co.push_back(&c->batch_results[(size_t) b].coarse); rr.push_back(us[(size_t) b].rng); }
            std::vector<std::vector<int32_t>> fine = engine_fine_many(t, co, &rr);
            for (size_t k = 0; k < take.size(); k++) { c->batch_results[(size_t) take[k]].fine = std::move(fine[k]); us[(size_t) take[k]].rng = rr[k]; }
        } else {
            for (int b : take) {
                bark_context::BatchResult & r = c->batch_results[(size_t) b];
                std::swap(t->rng, us[(size_t) b].rng);                                // the fine stage draws from the utterance's generator
                try { r.fine = engine_fine(t, r.coarse); } catch (...) { std::swap(t->rng, us[(size_t) b].rng); throw; }
                std::swap(t->rng, us[(size_t) b].rng);
            }
        }
        t->stats.t_fine_us += now_us() - t0;
    }
    void codec_pass(size_t count) {
        const int64_t t0 = now_us();
        std::vector<std::vector<int32_t>> codes; std::vector<const int32_t *> cp; std::vector<int> Ts;
        for (size_t k = 0; k < count; k++) {
            const bark_context::BatchResult & r = c->batch_results[(size_t) codec_wait[k]];
            const int T = (int) r.fine.size() / 8;
            std::vector<int32_t> cd((size_t) 8 * T);
            for (int ch = 0; ch < 8; ch++) for (int i = 0; i < T; i++) cd[(size_t) ch * T + i] = r.fine[(size_t) i * 8 + ch];      // bark.cpp:2153-2159
            codes.push_back(std::move(cd)); Ts.push_back(T);
            t->stats.n_frames += T; t->stats.n_semantic += (int32_t) r.semantic.size();
        }
        for (auto & cd : codes) cp.push_back(cd.data());
        std::vector<std::vector<float>> pcm = engine_codec_decode_many(t, cp, 8, Ts, -1, nullptr);
        for (size_t k = 0; k < count; k++) {
            bark_context::BatchResult & r = c->batch_results[(size_t) codec_wait[k]];
            r.audio = std::move(pcm[k]);
            t->stats.n_samples += (int32_t) r.audio.size();
            r.ok = true; good++;
        }
        codec_wait.erase(codec_wait.begin(), codec_wait.begin() + (long) count);
        t->stats.t_codec_us += now_us() - t0;
    }
    void run() {
        try {
            HIP_OK(hipSetDevice(t->device));
            static const int chunk_env = getenv("BARK_HIP_FINE_BATCH") ? atoi(getenv("BARK_HIP_FINE_BATCH")) : 8;
            const bool many = chunk_env > 1 && !t->gpt[2].q4 && !t->gpt[2].w32 && !t->host_sampling;
            const size_t chunk = (size_t) std::max(1, chunk_env);
            while (true) {
                std::vector<int> take;
                bool drained = false;
                {
                    std::unique_lock<std::mutex> lk(mu);
                    cv.wait(lk, [&] { return closed || !pending.empty(); });
                    if (abandon) break;
                    if (!pending.empty()) {
                        const float ft = us[(size_t) pending.front()].rp.fine_temp;
                        for (auto it = pending.begin(); it != pending.end() && take.size() < chunk;) {
                            if (us[(size_t) *it].rp.fine_temp == ft) { take.push_back(*it); it = pending.erase(it); } else ++it;
                        }
                    }
                    drained = pending.empty();
                    if (take.empty() && closed && codec_wait.empty()) break;
                }
                if (!take.empty()) {
                    fine_chunk(take, many);
                    std::sort(take.begin(), take.end());
                    codec_wait.insert(codec_wait.end(), take.begin(), take.end());
                    std::lock_guard<std::mutex> g(mu);
                    drained = pending.empty();
                }
                while (codec_wait.size() >= 32 || (drained && !codec_wait.empty())) codec_pass(std::min<size_t>(32, codec_wait.size()));
            }
        } catch (...) { err = std::current_exception(); }
        if (t->params.fine_temp != saved_fine_temp) { t->params.fine_temp = saved_fine_temp; drop_fine_graphs(); }
    }
};

}  // namespace

// A job of n utterances on the context's lock-step slots (bark.cpp:2125-2172 per utterance).  The stages run one after the other for the
// whole job; inside the semantic and the coarse stage the utterances travel through S = capacity slots: a slot whose utterance has
// finished (its own step cap / stop rule, its own number of coarse windows) is handed to the next waiting utterance, and once nobody
// waits the batch is compacted (the last slot moves into the hole), so every lock step runs over live utterances only.
int engine_generate_batch(bark_context * c, const char * const * texts, int n, const uint32_t * seeds, const bark_hip_request_params * rps, const BatchAdmit * admit) {
    HIP_OK(hipSetDevice(c->device));
    const JobScope job(c);
    const bark_context_params & p = c->params;
    if (n <= 0 || n > 4096) throw std::runtime_error("generate_batch: 1..4096 utterances per call");
    if (admit && (c->host_sampling || c->gpt[0].hp.n_embd != c->gpt[1].hp.n_embd || c->any_w32)) admit = nullptr;     // the sequential fallback takes the job as given
    c->batch_results.assign((size_t) n, bark_context::BatchResult());
    // one generator per utterance (bark.cpp:1179 seeds one per context): utterance i of a batch is what a fresh context with
    // seed seeds[i] would generate.  Without explicit seeds they are drawn from the context's generator, in order.
    std::vector<Utt> us((size_t) n);
    for (int i = 0; i < n; i++) {
        Utt & u = us[(size_t) i];
        u.text = texts[i];
        if (rps) u.rp = rps[i];
        else { u.rp.temp = p.temp; u.rp.fine_temp = p.fine_temp; u.rp.min_eos_p = p.min_eos_p; u.rp.n_steps_text_encoder = p.n_steps_text_encoder; u.rp.seed = seeds ? seeds[i] : (uint32_t) c->rng(); }
        if (rps && seeds) u.rp.seed = seeds[i];
        if (!(u.rp.temp >= 0.0f) || !(u.rp.fine_temp >= 0.0f)) throw std::runtime_error("generate_batch: temperatures must be >= 0");
        u.rng = std::mt19937(u.rp.seed);
    }
    if (c->host_sampling || c->gpt[0].hp.n_embd != c->gpt[1].hp.n_embd || c->any_w32) {
        // host-side sampling and f32 model files keep one utterance in flight: fall back to the sequential loop
        int good = 0;
        const bark_context_params saved = c->params;
        for (int i = 0; i < n; i++) {
            bark_context::BatchResult & r = c->batch_results[(size_t) i];
            Utt & u = us[(size_t) i];
            // the captured graphs bake the sampling constants: an utterance with other parameters than its predecessor needs fresh ones (as bark_hip_set_params)
            if (c->params.temp != u.rp.temp || c->params.fine_temp != u.rp.fine_temp || c->params.min_eos_p != u.rp.min_eos_p) engine_invalidate_graphs(c);
            c->params.temp = u.rp.temp; c->params.fine_temp = u.rp.fine_temp; c->params.min_eos_p = u.rp.min_eos_p; c->params.n_steps_text_encoder = u.rp.n_steps_text_encoder;
            std::swap(c->rng, u.rng);
            try { r.ok = engine_generate(c, u.text.c_str()); } catch (...) { std::swap(c->rng, u.rng); c->params = saved; engine_invalidate_graphs(c); throw; }
            std::swap(c->rng, u.rng);
            if (r.ok) { r.semantic = c->semantic_tokens; r.coarse = c->coarse_tokens; r.fine = c->fine_tokens; r.audio = c->audio; good++; }
        }
        if (c->params.temp != saved.temp || c->params.fine_temp != saved.fine_temp || c->params.min_eos_p != saved.min_eos_p) engine_invalidate_graphs(c);
        c->params = saved;
        return good;
    }
    const int64_t t0 = now_us();
    const int64_t t_load = c->stats.t_load_us;
    c->stats = bark_hip_stats{};
    c->stats.t_load_us = t_load;
    // the prompts of all slots in one pass (batch_prefill_many) instead of slot by slot; BARK_HIP_CROSSCHECK bit 4 (16) keeps the slot-by-slot route
    const bool prefill_many = !(crosscheck_mask() & 16);
    ensure_batch(c, c->batch.cap ? c->batch.cap : std::max(std::min(n, kMaxSlots), 8));
    bark_context::Batch & bb = c->batch;
    const int S = bb.cap;

    // ---- semantic (bark.cpp:1645-1701): lock steps over the live slots, slots refilled from the queue ------------------------------------
    int64_t t = now_us();
    {
        GptModel & m = c->gpt[0];
        const StageCfg s = stage_cfg(c, 0);
        const int cap_max = m.hp.block_size - 257 + 1;              // 257 prompt rows + one row per further step must fit the context
        PromptParams pp;
        pp.block_size = m.hp.block_size; pp.text_encoding_offset = p.text_encoding_offset; pp.text_pad_token = p.text_pad_token;
        pp.semantic_pad_token = p.semantic_pad_token; pp.semantic_infer_token = p.semantic_infer_token;
        std::deque<int> queue;
        int total_steps = 0, done_steps = 0, last_pct = 0;
        for (int i = 0; i < n; i++) {
            us[(size_t) i].cap = std::max(0, std::min(us[(size_t) i].rp.n_steps_text_encoder, cap_max));
            if (us[(size_t) i].cap > 0) { queue.push_back(i); total_steps += us[(size_t) i].cap; }
        }
        std::vector<int> slot_utt;                                  // slot -> utterance, a compact range [0, active)
        std::vector<std::vector<int32_t>> prompts((size_t) n);
        std::vector<int32_t> ids_all;                               // the slots' ids at a poll at which somebody retires: one copy for all of them
        auto retire = [&](int slot, const StepState & st) {
            Utt & u = us[(size_t) slot_utt[(size_t) slot]];
            const int keep = std::min(st.eos_step, u.cap);
            auto & out = c->batch_results[(size_t) slot_utt[(size_t) slot]].semantic;
            out.assign(ids_all.begin() + (long) slot * 2048, ids_all.begin() + (long) slot * 2048 + keep);
            const int n_used = st.eos_step == INT32_MAX ? u.cap : std::min(u.cap, st.eos_step + 1);
            c->stats.n_sample_semantic += n_used;
            if (u.rp.temp > 0.0f) u.rng.discard(2ull * (unsigned long long) n_used);        // as consume_uniforms()
            c->stats.n_near_tie += st.near_tie;
            done_steps += u.cap;
        };
        while (!queue.empty() || !slot_utt.empty()) {
            // continuous admission (the request collector, batcher.hip): while slots are free and nobody of this job waits for them, requests
            // that arrived in the meantime join the job - they ride along through the remaining stages
            while (admit && queue.empty() && (int) slot_utt.size() < S && n < admit->max_job) {
                Utt u;
                if (!admit->next(u.text, u.rp)) break;
                u.rng = std::mt19937(u.rp.seed);
                u.cap = std::max(0, std::min(u.rp.n_steps_text_encoder, cap_max));
                us.push_back(std::move(u)); c->batch_results.emplace_back(); prompts.emplace_back();
                if (us.back().cap > 0 && us.back().rp.temp >= 0.0f && us.back().rp.fine_temp >= 0.0f) { queue.push_back(n); total_steps += us.back().cap; }
                n++;
            }
            // hand free slots to waiting utterances: their prompts go through the model in one pass, which also takes their first sample
            std::vector<int> fresh;
            while (!queue.empty() && (int) slot_utt.size() < S) { fresh.push_back((int) slot_utt.size()); slot_utt.push_back(queue.front()); queue.pop_front(); }
            if (!fresh.empty()) {
                for (int slot : fresh) {
                    Utt & u = us[(size_t) slot_utt[(size_t) slot]];
                    set_slot_params(c, slot, u);
                    if (u.rp.temp > 0.0f) upload_slot_uniforms(c, slot, u.rng, u.cap);
                    prompts[(size_t) slot_utt[(size_t) slot]] = build_semantic_prompt(c->vocab, pp, u.text.c_str(), true);
                    u.issued = 1;
                }
                upload_slot_params(c);
                if (prefill_many && !m.q4 && !m.w32) {
                    std::vector<const std::vector<int32_t> *> ids; std::vector<int> l0(fresh.size(), 0), s0(fresh.size(), 0);
                    for (int slot : fresh) ids.push_back(&prompts[(size_t) slot_utt[(size_t) slot]]);
                    batch_prefill_many(c, s, fresh, ids, l0, true, s0);
                } else {
                    for (int slot : fresh) batch_prefill_and_sample(c, s, slot, prompts[(size_t) slot_utt[(size_t) slot]], true, 0);
                }
            }
            const int B = (int) slot_utt.size();
            // lock steps until the next poll: at most 32, and no further than the live utterance with the FEWEST steps left needs - a slot is looked at
            // exactly when its cap is reached, so no lock step is spent on an utterance that is known to be finished and the queue refills its slot at once
            // (an utterance that meets its STOP RULE between two polls keeps stepping until the next one, at most 31 steps; its further ids are discarded).
            // issued <= cap <= cap_max for every live slot, so no step runs past the end of a slot's context.
            int least_left = INT32_MAX;
            for (int b = 0; b < B; b++) { const Utt & u = us[(size_t) slot_utt[(size_t) b]]; least_left = std::min(least_left, u.cap - u.issued); }
            const int k = B ? std::max(0, std::min(32, least_left)) : 0;
            for (int j = 0; j < k; j++) batch_step(c, s, B);
            for (int b = 0; b < B; b++) us[(size_t) slot_utt[(size_t) b]].issued += k;
            std::vector<StepState> st;
            read_back(c, B, nullptr, st);
            for (auto & v : st) if (v.fault) throw std::runtime_error("lock step launched with a context bound below the cached keys");
            // retire from the back so that a move never touches a slot that is still to be looked at
            bool moved = false, any_done = false;
            for (int b = 0; b < B; b++) any_done = any_done || !(st[(size_t) b].eos_step == INT32_MAX && us[(size_t) slot_utt[(size_t) b]].issued < us[(size_t) slot_utt[(size_t) b]].cap);
            if (any_done) read_back(c, B, &ids_all, st);          // the stream is idle: one more copy, taken only when somebody leaves
            for (int b = B - 1; b >= 0; b--) {
                const Utt & u = us[(size_t) slot_utt[(size_t) b]];
                if (st[(size_t) b].eos_step == INT32_MAX && u.issued < u.cap) continue;
                retire(b, st[(size_t) b]);
                const int last = (int) slot_utt.size() - 1;
                // the last slot moves into the hole: the live slots stay a compact range, and the refill appends behind them
                if (b != last) { move_slot(c, 0, last, b); slot_utt[(size_t) b] = slot_utt[(size_t) last]; moved = true; }
                slot_utt.pop_back();
            }
            if (moved) upload_slot_params(c);
            // admissions grow total_steps: the reported percentage never goes backwards
            last_pct = std::max(last_pct, total_steps ? (int) (100ll * done_steps / total_steps) : 100);
            progress(c, SEMANTIC, last_pct);
        }
    }
    c->stats.t_semantic_us = now_us() - t;

    // the job's tail on a second stream (JobTail): utterances are handed over as they leave the coarse stage
    JobTail tail(c, us, tail_stream_enabled() && n > 1);

    // ---- coarse (bark.cpp:1745-1863): windows in lock step, slots refilled at window boundaries ------------------------------------------
    t = now_us();
    {
        GptModel & m = c->gpt[1];
        const StageCfg s = stage_cfg(c, 1);
        if (p.n_coarse_codebooks != 2 || p.codebook_size != 1024 || p.sliding_window_size <= 0 || p.max_coarse_history < 0)
            throw std::runtime_error("coarse: unsupported parameters");
        if (s.lm_row0 + 2 * s.lm_rows > m.hp.n_out_vocab) throw std::runtime_error("coarse: vocabulary too small");
        const float stc_ratio = p.coarse_rate_hz / p.semantic_rate_hz * p.n_coarse_codebooks;
        const int max_semantic_history = (int) floorf(p.max_coarse_history / stc_ratio);
        std::deque<int> queue;
        long total_steps = 0, done_steps = 0;
        for (int i = 0; i < n; i++) {
            const auto & sem = c->batch_results[(size_t) i].semantic;
            if (sem.empty()) continue;
            us[(size_t) i].n_steps = (int) (floorf(sem.size() * stc_ratio / p.n_coarse_codebooks) * p.n_coarse_codebooks);
            if (us[(size_t) i].n_steps > 0) { queue.push_back(i); total_steps += us[(size_t) i].n_steps; }
        }
        const bool reuse_prefix = !(crosscheck_mask() & 8);
        std::vector<int> slot_utt;
        while (!queue.empty() || !slot_utt.empty()) {
            bool params_dirty = false;
            // the slots of a lock step share the codebook parity of their step (slot 0's step selects the LM-head rows of everyone): with an odd
            // sliding window the live utterances' step counts are odd in every second window, and a newcomer (step 0) has to wait for an even one
            const bool parity_ok = slot_utt.empty() || (us[(size_t) slot_utt[0]].step_idx & 1) == 0;
            while (parity_ok && !queue.empty() && (int) slot_utt.size() < S) {
                const int slot = (int) slot_utt.size();
                slot_utt.push_back(queue.front()); queue.pop_front();
                Utt & u = us[(size_t) slot_utt.back()];
                set_slot_params(c, slot, u); params_dirty = true;
                if (u.rp.temp > 0.0f) upload_slot_uniforms(c, slot, u.rng, u.n_steps);          // indexed by the utterance's step_idx
                u.cached.clear();
            }
            if (params_dirty) upload_slot_params(c);
            const int B = (int) slot_utt.size();
            // ---- one window for every live slot ----
            int max_here = 0;
            std::vector<int> here((size_t) B, 0), Ls((size_t) B, 0);
            std::vector<std::vector<int32_t>> ins((size_t) B);
            bool all_single = true;                                      // every slot needs exactly one new row
            for (int b = 0; b < B; b++) {
                Utt & u = us[(size_t) slot_utt[(size_t) b]];
                const auto & sem = c->batch_results[(size_t) slot_utt[(size_t) b]].semantic;
                const auto & out = u.coarse_out;
                const int semantic_idx = (int) roundf(u.step_idx / stc_ratio);
                std::vector<int32_t> in(sem.begin() + std::max(semantic_idx - max_semantic_history, 0), sem.end());
                const size_t had = in.size();
                in.resize(256);
                for (size_t i = had; i < 256; i++) in[i] = p.coarse_semantic_pad_token;
                in.push_back(p.coarse_infer_token);
                const int nh = std::min(p.max_coarse_history, (int) out.size());
                in.insert(in.end(), out.end() - nh, out.end());
                here[(size_t) b] = std::min(p.sliding_window_size, u.n_steps - u.step_idx);
                if ((int) in.size() + here[(size_t) b] - 1 > m.hp.block_size) throw std::runtime_error("coarse: window exceeds the context");
                check_ids(in.data(), in.size(), m.hp.n_in_vocab, "coarse");
                int L = 0;
                if (reuse_prefix) {
                    const auto & cd = u.cached;
                    while (L < (int) in.size() && L < (int) cd.size() && cd[(size_t) L] == in[(size_t) L]) L++;
                    if (L >= (int) in.size()) L = (int) in.size() - 1;
                }
                Ls[(size_t) b] = L;
                if ((int) in.size() - L != 1) all_single = false;
                ins[(size_t) b] = std::move(in);
                max_here = std::max(max_here, here[(size_t) b]);
            }
            // a slot whose last window is shorter than the others' steps on to the end of the window: its rows must fit its context too
            for (int b = 0; b < B; b++)
                if ((int) ins[(size_t) b].size() + max_here - 1 > m.hp.block_size) throw std::runtime_error("coarse: a lock-step window exceeds the context of a slot (history + sliding window too long for a batch)");
            int lock_steps = max_here - 1;                               // batched steps after every slot has its first sample
            std::vector<int> pf_slots, pf_L, pf_step; std::vector<const std::vector<int32_t> *> pf_ids;
            // states of the slots that continue with a decode step: uploaded without a host synchronisation per slot (64 slots x 13 windows of
            // 25 us round trips otherwise) from a pinned staging row per slot, which is not touched again before read_back() below has
            // synchronised the stream
            StepState * stage_st = bb.h_state_in;
            for (int b = 0; b < B; b++) {
                Utt & u = us[(size_t) slot_utt[(size_t) b]];
                const auto & in = ins[(size_t) b];
                const int L = Ls[(size_t) b];
                c->stats.n_prefix_rows_reused += L;
                if ((int) in.size() - L == 1) {
                    // the prompt is the cached sequence plus one token: a decode step (prefix reuse, see engine_coarse)
                    StepState & st1 = stage_st[b];
                    st1 = fresh_state(); st1.step = u.step_idx; st1.n_past = L; st1.cur_token = in[(size_t) L];
                    HIP_OK(hipMemcpyAsync(bb.state + b, &st1, sizeof(StepState), hipMemcpyHostToDevice, c->stream));
                    embed_slot(c, s, b);
                    if (!all_single) enqueue_batch_step(c, s, 1, slot_view(c, s, b));      // mixed window: this slot alone, eagerly
                } else if (prefill_many && !m.q4 && !m.w32) {
                    pf_slots.push_back(b); pf_ids.push_back(&ins[(size_t) b]); pf_L.push_back(L); pf_step.push_back(u.step_idx);
                } else {
                    batch_prefill_and_sample(c, s, b, in, false, u.step_idx, L);
                }
            }
            if (!pf_slots.empty()) batch_prefill_many(c, s, pf_slots, pf_ids, pf_L, false, pf_step);
            if (all_single && max_here > 0) lock_steps = max_here;       // the first sample of the window is a lock step too
            // a slot whose last window is shorter than the others' keeps stepping to the end of the window (its further ids are discarded;
            // window prompt + sliding_window_size rows fit the context by the check above, as every window is even the parity stays shared)
            for (int j = 0; j < lock_steps; j++) batch_step(c, s, B);
            // the window's ids of all slots in ONE copy (rows of 2048 per slot, 512 KB at 64 slots) and their states
            std::vector<int32_t> ids_all;
            std::vector<StepState> st;
            read_back(c, B, &ids_all, st);
            for (int b = 0; b < B; b++) {
                Utt & u = us[(size_t) slot_utt[(size_t) b]];
                if (st[(size_t) b].fault) throw std::runtime_error("lock step launched with a context bound below the cached keys");
                const std::vector<int32_t> got(ids_all.begin() + (long) b * 2048, ids_all.begin() + (long) b * 2048 + here[(size_t) b]);
                u.coarse_out.insert(u.coarse_out.end(), got.begin(), got.end());
                // rows now in the slot's cache: its prompt and every token fed back (steps past `here` wrote further rows, but those are
                // never matched because the ids are not recorded)
                u.cached = ins[(size_t) b];
                u.cached.insert(u.cached.end(), got.begin(), got.end() - 1);
                u.step_idx += here[(size_t) b];
                done_steps += here[(size_t) b];
                c->stats.n_sample_coarse += here[(size_t) b];
                c->stats.n_near_tie += st[(size_t) b].near_tie;
            }
            // retire the finished utterances (from the back: a move never touches a slot still to be looked at)
            bool moved = false;
            std::vector<int> retired;
            for (int b = B - 1; b >= 0; b--) {
                Utt & u = us[(size_t) slot_utt[(size_t) b]];
                if (u.step_idx < u.n_steps) continue;
                if (u.rp.temp > 0.0f) u.rng.discard(2ull * (unsigned long long) u.n_steps);
                auto & res = c->batch_results[(size_t) slot_utt[(size_t) b]].coarse;
                for (size_t i = 0; i + 1 < u.coarse_out.size(); i += 2) {
                    res.push_back(u.coarse_out[i] - p.semantic_vocab_size);
                    res.push_back(u.coarse_out[i + 1] - p.semantic_vocab_size - p.codebook_size);
                }
                if (!res.empty()) retired.push_back(slot_utt[(size_t) b]);
                const int last = (int) slot_utt.size() - 1;
                if (b != last) { move_slot(c, 1, last, b); slot_utt[(size_t) b] = slot_utt[(size_t) last]; moved = true; }
                slot_utt.pop_back();
            }
            if (moved) upload_slot_params(c);
            std::sort(retired.begin(), retired.end());
            tail.push(retired);                                          // all of a window boundary at once: the helper groups them into passes
            progress(c, COARSE, total_steps ? (int) (100 * done_steps / total_steps) : 100);
        }
    }
    c->stats.t_coarse_us = now_us() - t;

    // ---- fine + codec: whatever the helper has not finished yet (everything, without a second stream) ------------------------------------
    tail.finish();
    if (guard_allocations()) { (void) guard_check(c, "end of a lock-step job"); if (c->tail) (void) guard_check(c->tail, "end of a lock-step job, tail clone"); }
    if (tail.t != c) progress(c, FINE, 100);                         // the helper reports nothing (the callback belongs to the calling thread)
    const int good = tail.good;
    c->stats.t_eval_us = now_us() - t0;
    return good;
}

}  // namespace barkhip
