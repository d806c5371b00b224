// engine.h - the MI355X Bark engine behind bark.h / bark_mi355x.h.
//
// Pipeline (reference call stack /root/reference/bark.cpp:2125-2172): tokenise -> semantic GPT ->
// coarse GPT (sliding windows) -> fine GPT (6 non-causal passes per window) -> EnCodec decoder.
// Weights live in one device slab; activations, KV caches and the per-stage StepState are
// device-resident; a decode step is a fixed kernel sequence captured once per model in a hipGraph
// and replayed per token with no host-side parameter updates.
#pragma once
#include "bark.h"
#include "bark_mi355x.h"
#include "kernels.h"
#include "model_file.h"
#include "tokenizer.h"

#include <functional>
#include <map>
#include <memory>
#include <random>
#include <string>
#include <vector>

namespace barkhip {

struct GptModel {
    GptHparams hp;
    const half_t * wte[8] = {};
    const half_t * lm_head[8] = {};
    // quantised model files (bark_model_quantize output): every weight matrix is a QMat instead of an f16 pointer
    bool q4 = false;                                    // activations stay f32 (quantised or f32 weights)
    bool w32 = false;                                   // f32 weights (QMat with qt == QT_F32)
    QMat wte_q[8], lm_head_q[8];
    const float * wpe = nullptr, * lnf_g = nullptr, * lnf_b = nullptr;
    struct Layer {
        const float * ln1_g = nullptr, * ln1_b = nullptr, * ln2_g = nullptr, * ln2_b = nullptr;
        const half_t * attn_w = nullptr, * proj_w = nullptr, * fc_w = nullptr, * mproj_w = nullptr;
        QMat attn_q, proj_q, fc_q, mproj_q;
        const float * attn_b = nullptr, * proj_b = nullptr, * fc_b = nullptr, * mproj_b = nullptr;
    };
    std::vector<Layer> layers;
    float * kcache = nullptr, * vcache = nullptr;       // [L][H][16][P][4] / [L][H][P][64] f32; fine model: L = 1 scratch
    float * vtcache = nullptr;                          // second copy of V in the K layout [L][H][16][P][4] (semantic / coarse): the decode attention
                                                        // reads 4 value dims of consecutive keys as one contiguous stream
    size_t kv_layer_stride = 0;                         // floats per layer (0 for the fine model's shared scratch)
    // layers -> LM head -> sample + embedding of the next token; [ng]: variant whose kernels request the keys below 256 ng without
    // waiting for the context length (the host knows how many rows the cache holds when it launches a step)
    hipGraphExec_t decode_graph[5] = {};
    hipGraphExec_t decode_graph8[5] = {};               // eight such steps in one graph: one launch gap per eight tokens
    hipGraphExec_t bench_graph = nullptr;               // same, without advancing n_past (timing hook)
};

struct CodecModel {
    CodecHparams hp;
    int n_q = 0, D = 0;
    const float * codebooks = nullptr;                  // [n_q][n_bins][hidden_dim]
    // wm: kernel image for the matrix-core order C9m, [phases][cout32][kd16] f16 zero padded, kd = k * cin + ci (convT: one image per output
    // phase over tap * cin + ci), nullptr when cin is not a multiple of 8; w32: f32 copy of the file's kernel (exact) for order C9
    struct Conv { const half_t * w = nullptr; const float * w32 = nullptr; const float * b = nullptr; const half_t * wm = nullptr; int cout = 0, cin = 0, k = 0; };
    struct ConvT { const half_t * w = nullptr; const float * w32 = nullptr; const float * b = nullptr; const half_t * wm = nullptr; int cin = 0, cout = 0, k = 0, stride = 0; };
    struct Lstm { const half_t * w_ih = nullptr, * w_hh = nullptr; const float * b_ih = nullptr, * b_hh = nullptr; };
    Conv init, fin;
    Lstm lstm[2];
    struct Block { ConvT up; Conv c1, c2, sc; } blocks[4];
};

}  // namespace barkhip

// The opaque handle of bark.h.
struct bark_context {
    bark_context_params params;
    std::mt19937 rng;

    barkhip::Vocab vocab;
    barkhip::GptModel gpt[3];
    barkhip::CodecModel codec;
    int device = 0;
    hipStream_t stream = nullptr;
    bool use_graph = true;
    // Order of the fine model's weight products on f16 model files.  0 (default policy): C1 - the restated reference order, on the
    // f32 matrix cores - for bark_generate_audio and the stage-level entry points, C1m - the f16 matrix cores' own accumulation - inside lock-step jobs
    // (in_job) and side-by-side fine passes; 1: C1 everywhere; 2: C1m everywhere (rounds 4 - 5).  BARK_HIP_FINE_ORDER=c1|c1m, bark_hip_set_fine_order.
    int fine_order = 0;
    bool in_job = false;                 // a lock-step job (engine_generate_batch) or a side-by-side fine pass is running on this context
    int fast_gemm = 0;                   // BARK_HIP_FAST_GEMM=1: N > 1 products on the f16 matrix cores in hardware accumulation order (non-canonical, tolerance mode)
    int decode_ng = 4;                                  // key groups (of 256) the decode kernels being enqueued may assume: ctx <= 256 ng

    // device memory.  The weight slab (and the codec codebooks) are immutable after load and shared by every
    // context cloned from this one (bark_hip_clone_context): replicas on one GPU stream the same bytes.
    struct SharedWeights { void * slab = nullptr; void * codebooks = nullptr; std::vector<void *> extra; int device = 0; ~SharedWeights(); };
    std::shared_ptr<SharedWeights> weights;
    size_t weight_bytes = 0;
    std::vector<void *> allocs;                         // everything else (freed in destroy)
    std::vector<std::pair<void *, size_t>> guarded;     // BARK_HIP_GUARD: {base of the allocation, bytes between its two guard bands}
    // GPT scratch
    float * x = nullptr, * q = nullptr, * logits = nullptr;
    float * knew = nullptr;                             // [E] K row appended by the current decode step (fixed-address copy)
    float * ps = nullptr;                               // [H][4][P] partial attention scores of a decode step (QKV kernel -> attn_ps_kernel)
    barkhip::half_t * xn = nullptr, * att = nullptr, * hbuf = nullptr;
    barkhip::half_t * q16 = nullptr, * k16 = nullptr, * vt16 = nullptr;      // tolerance route (fast_gemm): f16 operands of the flash attention, [P][E] each
    // quantised models: activations stay f32 between the products and are quantised to q8 rows (xq) in front of each
    bool any_q4 = false;
    float * att32 = nullptr, * h32 = nullptr; barkhip::Q8Scratch xq;
    bool any_w32 = false; float * xn32 = nullptr;       // f32 model files: LayerNorm-ed rows without f16 rounding
    int32_t * d_tokens = nullptr, * d_out_tokens = nullptr;
    float * d_eos_trace = nullptr;
    barkhip::StepState * d_state = nullptr;
    double * d_u = nullptr;                             // uniform draws for on-device multinomial sampling (8192)
    bool host_sampling = false;                         // BARK_HIP_HOST_SAMPLING=1: sample temp > 0 on the host (A/B path)
    uint16_t * d_gelu_lut = nullptr;
    int max_E = 0, max_H = 0, P = 1024;
    // codec scratch (grown on demand)
    float * cbuf[3] = {nullptr, nullptr, nullptr}; size_t cbuf_elems = 0;
    barkhip::half_t * cbuf_hh[3] = {nullptr, nullptr, nullptr};
    float * c_gi = nullptr, * c_cell = nullptr, * c_cell2 = nullptr; barkhip::half_t * c_hseq_h = nullptr, * c_xt_h = nullptr, * c_hseq2_h = nullptr; size_t c_T = 0;
    int32_t * d_codes = nullptr; size_t d_codes_elems = 0;
    hipGraphExec_t fine_graphs[16] = {};                // one captured forward pass + pick per predicted codebook, [8 * (products in C1m) + codebook]
    int * d_lstm_t = nullptr;                           // step counter of the replayed LSTM block
    int * d_codec_T = nullptr;                          // frame counts of the utterances being decoded and their prefix sums (CodecBatch)
    struct LstmGraph { hipGraphExec_t exec = nullptr; int B = 0; const float * out = nullptr; const float * gi = nullptr; } lstm_graph;    // 64 wave-front steps
    struct CodecGraph { hipGraphExec_t exec = nullptr; std::vector<int> T; const float * buf = nullptr; float * out = nullptr; int tmul = 0; } codec_graph;   // conv stack behind the LSTM

    // batched decode (several utterances in lock step on this context, bark_hip_generate_batch): per-slot KV caches and decode rows,
    // the row scratch of the all-slots prefill; FineBatch below holds the scratch of the side-by-side fine passes
    struct Batch {
        int cap = 0;
        float * kc[2] = {nullptr, nullptr}, * vc[2] = {nullptr, nullptr}; size_t slot_stride[2] = {0, 0};
        float * x = nullptr, * q = nullptr, * logits = nullptr; barkhip::half_t * att = nullptr, * h = nullptr;
        barkhip::StepState * state = nullptr; int32_t * out_tokens = nullptr; float * eos_trace = nullptr; float * ln_stats = nullptr;
        float * att32 = nullptr, * h32 = nullptr;        // quantised models: f32 activations per slot
        float * sc = nullptr;                            // [cap][max_H][P] attention scores of a lock step (scores kernel -> mix kernel)
        float * ps = nullptr;                            // [cap][max_H][4][P] partial scores per slot (few-slot lock steps: QKV kernel -> attn_fused_ps_kernel)
        double * u = nullptr;                            // [cap][8192] uniform draws of the slots' own generators (temp > 0)
        size_t ld_logits = 0;
        float * slot_par = nullptr;                      // the slots' own temperatures [cap] and min_eos_p [cap] (bark_hip_request_params)
        // pinned host memory (hipHostMalloc, freed by the context): where the sampled ids [cap][2048] and the states [cap] of the live slots land at
        // a poll / window end, and the staging rows of the states uploaded at a window start
        int32_t * h_ids = nullptr; barkhip::StepState * h_state = nullptr, * h_state_in = nullptr;
        // window prompts of all slots in ONE pass (batch_prefill_many): row scratch for cap * P rows, the prompts' ids, the sequence table
        float * pf_x = nullptr, * pf_q = nullptr; barkhip::half_t * pf_xn = nullptr, * pf_att = nullptr, * pf_h = nullptr;
        int32_t * pf_tokens = nullptr; barkhip::SeqTab * pf_tab = nullptr;
    } batch;
    std::vector<float> h_slot_par;                      // host mirror of batch.slot_par
    std::vector<std::pair<std::string, hipEvent_t>> * step_marks = nullptr;      // engine_profile_lock_step: events behind the launch sites of a lock step
    std::map<int, hipGraphExec_t> batch_graphs;         // captured lock steps by (model, active slots, kinds of sampling among them)
    // fine windows of several utterances in one forward pass (engine_fine_many): rows = cap * 1024
    struct FineBatch {
        int cap = 0;
        float * x = nullptr, * q = nullptr, * logits = nullptr, * kc = nullptr, * vc = nullptr;
        barkhip::half_t * xn = nullptr, * att = nullptr, * hbuf = nullptr, * q16 = nullptr, * k16 = nullptr, * vt16 = nullptr;
        int32_t * tokens = nullptr, * picks = nullptr;   // [8][cap * 1024] window ids (codebook-major planes), [cap * 1024] scratch picks
        double * u = nullptr;                            // [6][cap * 1024] uniform draws (fine_temp > 0)
    } fine_batch;
    struct BatchResult { std::vector<int32_t> semantic, coarse, fine; std::vector<float> audio; bool ok = false; };
    std::vector<BatchResult> batch_results;
    bark_context * tail = nullptr;                      // clone that runs the fine passes / codec of a lock-step job beside its decode chain (engine_batch.hip: JobTail)

#ifdef BARK_TRACE
    unsigned long long * trace_rec = nullptr; unsigned * trace_pos = nullptr; unsigned trace_cap = 0; int trace_kid = 0; unsigned trace_base = 0, trace_per_replay = 0;
#endif
    // results of the last generate call
    std::vector<int32_t> tokens, semantic_tokens, coarse_tokens, fine_tokens;
    std::vector<float> audio;
    std::vector<float> eos_trace;
    bark_hip_stats stats{};
    std::string description;

    ~bark_context();
};

namespace barkhip {

// All functions throw std::runtime_error on failure; the C API catches at the boundary.
bark_context * engine_load(const char * path, const bark_context_params & params, uint32_t seed, int device = -1);      // device < 0: BARK_HIP_DEVICE / the current device
bark_context * engine_clone(bark_context * src, uint32_t seed);          // same weights, own stream / caches / scratch
void engine_invalidate_graphs(bark_context * ctx);

int  engine_gpt_eval(bark_context * ctx, int which, const int32_t * tokens, int n_tokens, int n_past, bool merge_ctx, float * logits);
void engine_fine_eval(bark_context * ctx, const int32_t * tokens_8x1024, int nn, float * logits);

std::vector<int32_t> engine_semantic(bark_context * ctx, const std::vector<int32_t> & prompt, std::vector<float> * eos_trace);
std::vector<int32_t> engine_coarse(bark_context * ctx, const std::vector<int32_t> & semantic);          // [T][2]
std::vector<int32_t> engine_fine(bark_context * ctx, const std::vector<int32_t> & coarse_Tx2);          // [T][8]
// the fine stage of several utterances, their windows side by side in every forward pass (f16 model files; per-utterance results are those
// of engine_fine); rngs: one generator per utterance (fine_temp > 0), advanced as engine_fine advances the context's
std::vector<std::vector<int32_t>> engine_fine_many(bark_context * ctx, const std::vector<const std::vector<int32_t> *> & coarse, std::vector<std::mt19937> * rngs);
// tap_stage >= 0: *tap receives the activation after that stage (0 first conv, 1 LSTM+skip, 2..5 up-blocks)
std::vector<float>   engine_codec_decode(bark_context * ctx, const int32_t * codes, int n_q, int T, int tap_stage, std::vector<float> * tap);
// all utterances of a batch in one pass (codes[b]: [n_q][T[b]]); the launches of one utterance serve all of them
std::vector<std::vector<float>> engine_codec_decode_many(bark_context * ctx, const std::vector<const int32_t *> & codes, int n_q, const std::vector<int> & T,
                                                         int tap_stage, std::vector<float> * tap);
bool engine_generate(bark_context * ctx, const char * text);
// seeds: one std::mt19937 seed per utterance (temp > 0); nullptr: drawn from the context's generator, in order.  Returns #ok
// Continuous admission: while the semantic stage of a job has free slots and nobody of the job waits for them, next() may hand over further
// requests (false: none pending); they join the job - results are appended behind the n given ones - up to max_job utterances in total.
struct BatchAdmit { std::function<bool(std::string & text, bark_hip_request_params & rp)> next; int max_job = 0; };
// rps: per-utterance parameters (nullptr: the context's for everyone); seeds override rps[i].seed when both are given
int  engine_generate_batch(bark_context * ctx, const char * const * texts, int n, const uint32_t * seeds, const bark_hip_request_params * rps = nullptr,
                           const BatchAdmit * admit = nullptr);
void engine_reserve_batch(bark_context * ctx, int slots);          // fixes the lock-step capacity (otherwise the first batch call does)

double engine_time_decode_step(bark_context * ctx, int which, int ctxlen, int iters, double * bytes_per_step);
double engine_time_gemv(bark_context * ctx, int which, int op, int iters, double * bytes_per_launch);
double engine_time_fine_pass(bark_context * ctx, int iters, double * flops_per_pass, int Z = 1);     // Z windows side by side (engine_fine_many's pass)
double engine_time_slots(bark_context * c, int which, int op, int B, int kind, int ctxlen, int iters);
void engine_profile_lock_step(bark_context * c, int which, int B, int ctxlen, int reps, std::vector<std::pair<std::string, double>> & out);
#ifdef BARK_TRACE
int engine_trace_decode_step(bark_context * ctx, int which, int ctxlen, int replays, unsigned long long * out6, int cap_records);
#endif

}  // namespace barkhip
