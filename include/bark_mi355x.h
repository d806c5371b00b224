/*
 * bark_mi355x.h - engine extensions of the MI355X-native Bark library (libbark.so).
 *
 * bark.h stays byte-compatible with the reference; everything the reference API cannot express
 * (device selection, stage-level entry points that the parity tests and benchmarks drive,
 * statistics beyond load/eval time) is declared here with plain C types.
 *
 * Each stage-level entry point replaces one internal function of the reference and is what a
 * maintainer would bind if the reference exposed it (file:line cites /root/reference/bark.cpp):
 *
 *   bark_hip_tokenize        bark_tokenize_input                 bark.cpp:622-662
 *   bark_hip_bert_tokenize   bert_tokenize                       bark.cpp:558-620
 *   bark_hip_gpt_eval        bark_eval_encoder_internal          bark.cpp:1586-1643
 *   bark_hip_fine_eval       bark_eval_fine_encoder_internal     bark.cpp:1907-1959
 *   bark_hip_semantic        bark_forward_text_encoder           bark.cpp:1645-1743
 *   bark_hip_coarse          bark_forward_coarse_encoder         bark.cpp:1745-1905
 *   bark_hip_fine            bark_forward_fine_encoder           bark.cpp:1961-2104
 *   bark_hip_codec_decode    encodec_decompress_audio call site  bark.cpp:2143-2167
 *
 * All of them run on the context's HIP stream and block until their result is on the host.
 * Environment (read at bark_load_model):
 *   BARK_HIP_DEVICE=<n>   HIP device ordinal (default: current device)
 *   BARK_HIP_GRAPH=0|1    replay decode steps from a captured hipGraph (default 1)
 *   (the other switches: INTEGRATION.md section 2)
 *
 * Numerics.  Tokenizer, sampling rules and the token ids of the semantic and coarse stages follow the reference's f16-product / f32-accumulate
 * arithmetic in one fixed summation order (bit-exact against the repository's CPU oracle, which restates it).  The fine model's weight products
 * and the codec's convolutions run on the f16 matrix cores in the hardware's own accumulation order: their outputs are bit-exact against the
 * oracle's emulation of that order and WITHIN TOLERANCE of the reference restatement, not bit-equal to it - fine logits within 2.5e-3, >= 98 % of
 * the fine ids identical on the same coarse input, codec SNR >= 55 dB on the same codes (tests/test_order_divergence.py; BARK_HIP_CROSSCHECK=1280
 * keeps the reference restatement on the device).  Parity against ggml itself is unpinned: the reference cannot be built here.
 */
#pragma once
#include "bark.h"

#ifdef __cplusplus
extern "C" {
#endif

/* which GPT: 0 semantic, 1 coarse, 2 fine */
enum bark_hip_model { BARK_HIP_SEMANTIC = 0, BARK_HIP_COARSE = 1, BARK_HIP_FINE = 2 };

/* out[10] = n_layer, n_head, n_embd, block_size, bias, n_in_vocab, n_out_vocab, n_lm_heads, n_wtes, ftype */
BARK_API int bark_hip_hparams(struct bark_context * bctx, int which, int32_t * out10);

/* Replace the generation parameters of a live context (same struct as bark_load_model takes). */
BARK_API void bark_hip_set_params(struct bark_context * bctx, struct bark_context_params params);

/* 513-id semantic prompt for `text`; returns 513 or <0. */
BARK_API int bark_hip_tokenize(struct bark_context * bctx, const char * text, int32_t * out513);
/* raw WordPiece ids (no offset / padding); returns the count. */
BARK_API int bark_hip_bert_tokenize(struct bark_context * bctx, const char * text, int32_t * out, int n_max);

/* One causal-model evaluation (which = 0|1).  n_tokens ids at positions n_past..; with merge_ctx
 * and n_past == 0 the 513-id prompt collapses to 257 rows.  Writes n_out logits of the last row.
 * Returns the new n_past, or -1. */
BARK_API int bark_hip_gpt_eval(struct bark_context * bctx, int which, const int32_t * tokens, int n_tokens,
                               int n_past, int merge_ctx, float * logits);

/* One fine-model forward: tokens [8][1024] codebook-major, codebook nn in 2..7.
 * logits [1024][n_out].  Returns 0 or -1. */
BARK_API int bark_hip_fine_eval(struct bark_context * bctx, const int32_t * tokens_8x1024, int nn, float * logits);

/* Stage loops (sampled on the device: argmax when temp == 0, multinomial otherwise).  Every output buffer comes with its
 * capacity; a result that does not fit fails the call (-1) instead of overrunning the buffer.
 * semantic: prompt513 -> out (capacity ids; eos_trace, when given, holds capacity + 1 floats); returns count or -1.
 * coarse  : semantic ids -> out [T][2] (capacity_rows rows; T = floor(n_semantic * 75 / 49.9 * 2 / 2)); returns T or -1.
 * fine    : coarse [T][2] -> out [T][8] (capacity_rows rows); returns T or -1 (T <= 8192). */
BARK_API int bark_hip_semantic(struct bark_context * bctx, const int32_t * prompt513, int32_t * out, int capacity, float * eos_trace);
BARK_API int bark_hip_coarse(struct bark_context * bctx, const int32_t * semantic, int n_semantic, int32_t * out_Tx2, int capacity_rows);
BARK_API int bark_hip_fine(struct bark_context * bctx, const int32_t * coarse_Tx2, int T, int32_t * out_Tx8, int capacity_rows);
/* The fine stage of n <= 64 utterances with their windows side by side in every forward pass (what bark_hip_generate_batch runs; f16 model
 * files).  Greedy, or device multinomial with ONE std::mt19937 PER UTTERANCE, each seeded with the next draw of the context's generator
 * (as the utterances of bark_hip_generate_batch are: with fine_temp > 0, fine_many(n = 1) is therefore not bark_hip_fine on the same context).  coarse: the utterances' [T_i][2]
 * arrays back to back, out: their [T_i][8] results back to back (capacity_rows >= sum T_i).  Returns sum T_i or -1. */
BARK_API int bark_hip_fine_many(struct bark_context * bctx, const int32_t * coarse_concat, const int * T, int n, int32_t * out_concat, int capacity_rows);

/* EnCodec decode: codes [n_q][T] (time contiguous) -> pcm (capacity floats; 320 * T are produced). Returns samples or -1. */
BARK_API int bark_hip_codec_decode(struct bark_context * bctx, const int32_t * codes, int n_q, int T, float * pcm, int capacity);

/* Per-layer parity tap of the codec: activation [C][T'] after stage 0 (first conv), 1 (LSTM + skip),
 * 2..5 (the four upsampling blocks).  Returns the element count or -1. */
BARK_API int bark_hip_codec_tap(struct bark_context * bctx, const int32_t * codes, int n_q, int T, int stage, float * out, int capacity);

/* Replicas on one GPU: a clone shares the (immutable) device weights of `src` and owns its stream, KV caches and
 * scratch, so several utterances can be in flight on one device.  Free clones and the original in any order. */
BARK_API struct bark_context * bark_hip_clone_context(struct bark_context * src, uint32_t seed);

/* Runs bark_generate_audio for n (context, text) pairs concurrently, one host thread + one HIP stream each
 * (no collectives: utterances are independent).  Returns the number of successful generations; results are read
 * from each context with bark_get_audio_data[_size]. */
BARK_API int bark_hip_generate_audio_batch(struct bark_context ** ctxs, const char * const * texts, int n);

/* In-engine batching: a job of n utterances (n <= 4096) travels through the lock-step slots of ONE context (up to 64; the first call or
 * bark_hip_reserve_batch fixes the number): the live slots advance together through the semantic and coarse decode loops, so every decode
 * kernel reads the weights once per step for all of them; a slot whose utterance has finished (its own step cap / stop rule, its own number
 * of coarse windows) is handed to the next waiting utterance, and when nobody waits the batch is compacted, so no lock step is spent on a
 * finished utterance.  The prompts of the slots go through the model in one pass, the fine windows of 8 utterances side by side, the codec
 * of the utterances in one pass.  Per-utterance results are bit-identical to bark_generate_audio on a fresh context with the same
 * parameters, whatever the job size, the slot count or the company an utterance travels in.  With temp > 0 every
 * utterance has its own std::mt19937 (the reference seeds one per context, bark.cpp:1179): utterance i is what a context
 * loaded with seed seeds[i] would generate; the unseeded call draws those seeds from the context's generator, in order.
 * BARK_HIP_HOST_SAMPLING degrades the call to a sequential loop.
 * Returns the number of utterances that produced audio.  Results: bark_hip_batch_audio / bark_hip_batch_tokens. */
BARK_API int bark_hip_generate_batch(struct bark_context * bctx, const char * const * texts, int n);
BARK_API int bark_hip_generate_batch_seeded(struct bark_context * bctx, const char * const * texts, int n, const uint32_t * seeds);
/* The same with per-utterance parameters - the fields of bark_context_params a request may set for itself (the loops they steer:
 * bark.cpp:1669-1695 step cap and stop rule, :201-247 temperatures); everything else comes from the context.  A server mixes
 * requests with different settings in one job; tests make slots stop at chosen steps. */
struct bark_hip_request_params {
    float    temp;                  /* semantic / coarse sampling temperature, 0 = greedy (bark_context_params::temp)  */
    float    fine_temp;             /* fine sampling temperature, 0 = greedy                                           */
    float    min_eos_p;             /* stop rule of the semantic loop                                                  */
    int32_t  n_steps_text_encoder;  /* step cap of the semantic loop                                                   */
    uint32_t seed;                  /* seed of the utterance's own std::mt19937 (temp > 0 / fine_temp > 0)             */
};
BARK_API int bark_hip_generate_batch_ex(struct bark_context * bctx, const char * const * texts, int n, const struct bark_hip_request_params * per_utterance);
/* Fixes the number of lock-step slots (1..64; at least 8 are allocated) before the first job; returns 0 or -1. */
BARK_API int bark_hip_reserve_batch(struct bark_context * bctx, int slots);
/* audio of utterance i of the last batch: returns the sample count (-1 on error), *data points into the context */
BARK_API int bark_hip_batch_audio(struct bark_context * bctx, int i, float ** data);
/* token stream of utterance i: stage 0 semantic, 1 coarse [T][2], 2 fine [T][8]; returns the id count or -1 */
BARK_API int bark_hip_batch_tokens(struct bark_context * bctx, int i, int stage, int32_t * out, int capacity);

/* Request collector in front of bark_hip_generate_batch - what a server puts where the reference's example holds one mutex around
 * bark_generate_audio (examples/server/server.cpp:76-94,128-163).  Any number of host threads submit; one worker thread owns `bctx`
 * (nobody else may use it while the batcher lives), collects pending requests - up to max_batch (<= 256; the context's slots, at most 64,
 * serve them as one job), waiting at most max_wait_ms
 * for a batch to fill once one request is pending - and runs them as ONE lock-step batch.  A request's result is what a fresh context
 * seeded with its `seed` generates, whatever batch it travelled in (seed is irrelevant for temp == 0).
 *   submit: thread-safe, returns a ticket > 0 (or -1).
 *   wait  : blocks until the request is done; copies the PCM (24 kHz mono) and returns the sample count, -1 if the generation failed,
 *           -(2 + samples) if `capacity` is too small (the ticket stays valid).  A ticket is consumed by a successful or failed wait.
 *   free  : serves what is pending, then stops the worker.  No thread may still be inside submit / wait of this batcher; tickets that were
 *           never waited for are dropped. */
struct bark_hip_batcher;
BARK_API struct bark_hip_batcher * bark_hip_batcher_create(struct bark_context * bctx, int max_batch, int max_wait_ms);
/* n_streams (1 .. 4) job streams: workers 1 .. n_streams-1 run on clones of `bctx` (bark_hip_clone_context; owned by the batcher) and serve the
 * same queue, so up to n_streams jobs are in flight on the GPU at once - two decode chains share the chip (two streams of 64-slot jobs:
 * +14 % prompts/s under load).  Results do not depend on the stream a request travelled in. */
BARK_API struct bark_hip_batcher * bark_hip_batcher_create_ex(struct bark_context * bctx, int max_batch, int max_wait_ms, int n_streams);
// One process, several GPUs.  bark_load_model (bark.h / bark.cpp:1165) takes the device from BARK_HIP_DEVICE or the current device; this variant takes it as an
// argument.  bark_hip_batcher_create_multi: a request collector whose workers are the caller's contexts - one per GPU - behind one queue (request-level
// data parallelism: no collective inside an utterance; the reference's server holds one context behind one mutex, examples/server/server.cpp:76-94).  The
// collector does not own the contexts; request defaults come from ctxs[0].  nullptr on a bad argument (null / repeated context, n_ctx outside 1 .. 64).
BARK_API struct bark_context * bark_hip_load_model_on_device(const char * model_path, struct bark_context_params params, uint32_t seed, int device);
BARK_API struct bark_hip_batcher * bark_hip_batcher_create_multi(struct bark_context * const * ctxs, int n_ctx, int max_batch, int max_wait_ms);
BARK_API int64_t bark_hip_batcher_submit(struct bark_hip_batcher * b, const char * text, uint32_t seed);
/* a request with its own parameters (nullptr: the context's) */
BARK_API int64_t bark_hip_batcher_submit_ex(struct bark_hip_batcher * b, const char * text, const struct bark_hip_request_params * params);
BARK_API int bark_hip_batcher_wait(struct bark_hip_batcher * b, int64_t ticket, float * pcm, int capacity);
BARK_API void bark_hip_batcher_stats(struct bark_hip_batcher * b, int * n_batches, int * n_requests, int * largest_batch);
/* requests that joined a job that was already running (continuous admission: while the semantic stage of a job has free slots, requests
 * arriving in the meantime are taken along, up to max_batch per job) */
BARK_API int bark_hip_batcher_admitted(struct bark_hip_batcher * b);
BARK_API void bark_hip_batcher_free(struct bark_hip_batcher * b);

/* Token streams of the last bark_generate_audio call (copied out; returns counts). */
BARK_API int bark_hip_get_semantic_tokens(struct bark_context * bctx, int32_t * out, int capacity);
BARK_API int bark_hip_get_coarse_tokens(struct bark_context * bctx, int32_t * out_Tx2, int capacity_rows);
BARK_API int bark_hip_get_fine_tokens(struct bark_context * bctx, int32_t * out_Tx8, int capacity_rows);

/* Detailed statistics of the last bark_generate_audio call. */
struct bark_hip_stats {
    int64_t t_load_us, t_eval_us;
    int64_t t_semantic_us, t_coarse_us, t_fine_us, t_codec_us;   /* host wall clock per stage        */
    int64_t n_sample_semantic, n_sample_coarse, n_sample_fine;   /* same quotient as bark.cpp:176-182 */
    int32_t n_semantic, n_frames, n_samples;
    int32_t n_near_tie;                                          /* greedy picks settled on the host  */
    int32_t graph_replays;                                       /* hipGraph launches issued          */
    int32_t n_prefix_rows_reused;                                /* coarse prompt rows served from the KV cache */
};
BARK_API void bark_hip_get_stats(struct bark_context * bctx, struct bark_hip_stats * out);

/* Micro-benchmark hook used by bench.py for the roofline line: runs `iters` decode steps of model
 * `which` at context length `ctx` on the context's stream and returns the average device time of
 * one step in microseconds (hipEvents on that stream); *bytes_per_step receives the algorithmic
 * bytes of one step (weights + KV rows read).  Returns <0 on error. */
BARK_API double bark_hip_time_decode_step(struct bark_context * bctx, int which, int ctx, int iters,
                                          double * bytes_per_step);

/* Device time (us) of ONE decode GEMV kernel launch (gemv_kernel), averaged over `iters` back-to-back launches that
 * rotate through the layers' weights.  op: 0 LN+QKV, 1 attention out-proj, 2 LN+FC+GELU, 3 MLP out-proj.
 * *bytes_per_launch receives the algorithmic bytes (the weight matrix in the file's format: f16, f32 or blocks). */
BARK_API double bark_hip_time_gemv(struct bark_context * bctx, int which, int op, int iters, double * bytes_per_launch);

/* Device time (us) of ONE lock-step decode kernel over n_slots utterance slots (the kernels of bark_hip_generate_batch), averaged over
 * `iters` back-to-back launches that rotate through the layers' weights.  op: 0 QKV, 1 attention out-proj, 2 FC + GELU, 3 MLP out-proj,
 * 4 LayerNorm of the slot rows, 5 attention of every slot at context `ctx`.  kind: 0 = the VALU GEMV per pair of slots with the LayerNorm fused
 * (ops 0 / 2) and, for op 5, one workgroup per (head, slot); 6 = the matrix-core product with the LayerNorm fused (ops 0 / 2); any other value =
 * the matrix-core product on normalised f16 rows and, for op 5, the scores + mix pair of launches where the engine would use it.  f16 model files only. */
BARK_API double bark_hip_time_slots(struct bark_context * bctx, int which, int op, int n_slots, int kind, int ctx, int iters);

/* Time line of ONE lock step over n_slots slots of model `which` (0 semantic, 1 coarse) at context `ctx`: the step is enqueued eagerly `reps`
 * times with a HIP event behind every launch site; writes a JSON array [{"site": ..., "us": average time from the previous event}, ...] in
 * launch order (kernel time + the gap in front of it), closed by {"site": "step (graph replay)", "us": the same step replayed from its
 * hipGraph}.  Returns the length written, or -1 (error / capacity too small). */
BARK_API int bark_hip_profile_lock_step(struct bark_context * bctx, int which, int n_slots, int ctx, int reps, char * json_out, int capacity);

/* Device time (us) of one fine forward pass (N = 1024), averaged over iters. */
BARK_API double bark_hip_time_fine_pass(struct bark_context * bctx, int iters, double * flops_per_pass);
/* the same for n_windows fine windows side by side (the forward pass of bark_hip_fine_many / of a lock-step batch); flops for all windows */
BARK_API double bark_hip_time_fine_passes(struct bark_context * bctx, int n_windows, int iters, double * flops_per_pass);

/* Library / device description (static string). */
// Order of the fine model's weight products on f16 model files (bark.cpp:1489,1533,1552,1558,1573: ggml_mul_mat of the fine graph).
//   0  default policy: C1 (the restated reference order; f32 matrix cores) for bark_generate_audio and the stage-level entry points - greedy fine ids are
//      bit-equal to the CPU restatement of the reference - and C1m (the f16 matrix cores' own accumulation, >= 98 % of the ids equal, logits within 2.5e-3)
//      inside lock-step jobs (bark_hip_generate_batch*, the request collector) and bark_hip_fine_many;
//   1  C1 everywhere;   2  C1m everywhere (the behaviour of rounds 4 - 5).     Environment: BARK_HIP_FINE_ORDER=c1|c1m at load.  Returns 0, -1 on a bad argument.
BARK_API int bark_hip_set_fine_order(struct bark_context * bctx, int order);
BARK_API const char * bark_hip_describe(struct bark_context * bctx);

#ifdef __cplusplus
}
#endif
