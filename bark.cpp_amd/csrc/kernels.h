// kernels.h - launch interface of the HIP kernels (kernels.hip, codec_kernels.hip).
//
// Numerics contract ("canonical numerics"): every kernel reproduces, with IEEE fp32
// fma/add and the summation orders C1/C2/C5, exactly what the CPU oracle computes, so that greedy
// token ids are bit-identical.  All code is built with -ffp-contract=off; fused multiply-adds are
// written explicitly (fmaf / MFMA).
#pragma once
#include "quant_formats.h"

#include <hip/hip_runtime.h>
#include <cstdint>

namespace barkhip {

typedef _Float16 half_t;

// Launch-side rejection of a shape no kernel is instantiated for: throws std::runtime_error (printf-style message), which the
// C API turns into nullptr / false / -1 like every other failure (api.hip) - the host process is never aborted.
[[noreturn]] void kernel_fail(const char * fmt, ...) __attribute__((format(printf, 1, 2)));

// Device-resident state of one autoregressive stage; kernels read/advance it so that a captured
// hipGraph of one decode step can be replayed without host-side parameter updates.
struct StepState {
    int32_t n_past;        // rows already in the KV cache == position of the next token
    int32_t cur_token;     // token embedded by the next decode step
    int32_t step;          // stage step counter (coarse: parity selects the logit slice)
    int32_t eos_step;      // first step whose sample met the stop rule (INT32_MAX if none)
    int32_t near_tie;      // samples settled by the exact path (near tie of the two largest logits, or eos_p next to min_eos_p)
    int32_t n_out;         // sampled ids written to out_tokens so far
    float   last_eos_p;
    int32_t fault;         // set by a kernel whose launch-time assumption about the context does not hold (1: more cached keys than the
                           // partial-score copies of the QKV kernel cover); the stage loops turn it into an error
};

// A ggml block-quantised matrix (quant_formats.h), re-laid out at load time into one array per field: scales d [M][K/32]
// f16, minima m (q4_1 / q5_1), fifth bits qh (q5_0 / q5_1) and the level bytes qs [M][K/32][16] ([..][32] for q8_0;
// low nibble of byte j = element j of the block, high nibble = element j + 16)
struct QMat {
    const half_t * d = nullptr; const uint8_t * qs = nullptr; const half_t * m = nullptr; const uint32_t * qh = nullptr;
    int qt = 0;                           // QuantId; QT_F32: qs points at plain f32 weights [M][K] (f32 model files), d / m / qh unused
};
// q8 image of N <= 1024 activation rows: levels q [N][K] int8, scales d and s = f16(d * sum q) as [N][K/32] and block-major
// [K/32][1024] (the i8-MFMA kernel reads 16 consecutive rows of one block)
struct Q8Scratch { int8_t * q = nullptr; float * d = nullptr, * dT = nullptr, * s = nullptr, * sT = nullptr; };

// Diagnostic build only (-DBARK_TRACE, tools/trace_decode.py): every wave of the decode kernels appends one record
// {kid | xcc << 16, workgroup | wave << 24, t_entry, t_kernarg, t_operands, t_done} (s_memrealtime, 100 MHz) to a device log.
#ifdef BARK_TRACE
struct TraceSink { unsigned long long * rec = nullptr; unsigned * pos = nullptr; unsigned cap = 0; int kid = 0; unsigned base = 0, per_replay = 0; };   // slot = *pos (replay) * per_replay + base + wave index: no atomics
#define BARK_TRACE_FIELD TraceSink tr;
#else
#define BARK_TRACE_FIELD
#endif

// One sequence of a multi-sequence causal prefill (the window prompts of a lock-step batch in ONE pass): its rows sit at [z seq, z seq + len)
// of every activation, continue its cache (utterance slot `slot`) at position pos0; rows len .. seq - 1 are padding
struct SeqTab { int slot, pos0, len, pad; };

enum LinEpi { EPI_QKV = 0, EPI_RESID = 1, EPI_GELU = 2, EPI_LOGITS = 3, EPI_QKV16 = 4 };      // EPI_QKV16: tolerance route only (fast_kernels.hip)

// One linear operator  y[n][m] = epi( C1dot(W[m], x[n]) + bias[m] )  for n < N, m < M.
struct LinArgs {
    const half_t * W = nullptr; int M = 0, K = 0;     // [M][K] f16, row-major ([out][in])
    int N = 1;
    // input rows: either f16 [N][K] ...
    const half_t * x_f16 = nullptr;
    // ... or (decode GEMV only) one f32 row normalised in the kernel prologue (LayerNorm fused)
    const float * x_f32 = nullptr; const float * ln_g = nullptr; const float * ln_b = nullptr;
    const float * ln_stats = nullptr;     // batched decode: {mean, 1/sqrt(var+eps)} per row from ln_stats_kernel (else computed in the kernel)
    // quantised weights (wq.qs != nullptr): the input row is f32 (x_f32, LayerNorm applied when ln_g != nullptr) and is
    // quantised to q8 blocks inside the kernel, as ggml's mul_mat does for a quantised src0; N > 1: rows already quantised
    // by launch_q8_rows into xq
    QMat wq;
    Q8Scratch xq;
    const float * bias = nullptr;
    int epi = EPI_LOGITS;
    // EPI_QKV: m < E -> q ; E <= m < 2E -> K cache ; else V cache, at position pos0 (+ st->n_past) + n
    float * q = nullptr; float * kc = nullptr; float * vc = nullptr; int E = 0, P = 0; int pos0 = 0;
    float * vt = nullptr;                 // optional second V store in the K layout [H][16][P][4] (read by attn_ps_kernel)
    const StepState * st = nullptr;
    // EPI_RESID: res[n][m] = (dot + bias) + res[n][m]
    float * res = nullptr;
    // EPI_GELU: out_h[n][m] = f16(gelu_lut(dot + bias))
    half_t * out_h = nullptr; const uint16_t * lut = nullptr;
    float * out_h32 = nullptr;            // q4_0 path: the GELU output stays f32 (it is quantised to q8_0 by the next product)
    // EPI_LOGITS: out[n*ld_out + m] = dot (+ bias)
    float * out = nullptr; int ld_out = 0;
    // EPI_QKV, decode, f16 weights: the workgroup that produces 16 consecutive q values (one C2 block of one head) also forms that
    // block's partial score against every cached key: ps[(h * 4 + block) * P + j], j < n_past (attn_ps_kernel finishes the sum)
    float * ps = nullptr; int ng = 4;     // ng: the context holds at most 256 ng keys (only ng - 1 copies of the q workgroups are launched)
    float * knew = nullptr;               // with ps: the appended K row is also stored here ([E], position independent), so that the attention kernel
                                          // can request it before it knows the context length
    float out_div = 0.0f;                 // != 0: out = (dot + bias) / out_div - the sampler's `l /= 0.7f` (bark.cpp:226-228) done where the logit is born
    // coarse LM head: only the 1024 logits of the active codebook are needed (bark.cpp:1829-1833);
    // the row window starts at parity_rows * (st->step & 1) rows into W (and bias)
    int parity_rows = 0;
    // batched decode (several utterances in lock step): row n of x / q / res / out_h / out is sequence slot n, which has
    // its own StepState st[n] and its own KV cache at kc/vc + n * kv_slot_stride
    int batched = 0, nbatch = 1; size_t kv_slot_stride = 0;
    // N > 1, f16 weights, opt-in (BARK_HIP_FAST_GEMM=1): v_mfma_f32_32x32x16_f16 with the matrix core's own f32 accumulation order
    // (gemm_f16_tile_kernel, fast_kernels.hip).  Same operands and roundings as the canonical product (R1), only the ORDER of the f32
    // additions differs from C1: results agree to f32 rounding noise, not bit for bit, so this route is never the one the parity tests check.
    int fast = 0;
    // EPI_QKV16 (fast route, fine model): q (x 0.125) and k as f16 rows [N][E], v transposed [N / seq][E][seq] with permuted keys - the
    // operands of attn_flash_f16_kernel; N is a whole number of sequences of `seq` rows
    half_t * q16 = nullptr, * k16 = nullptr, * vt16 = nullptr; int seq = 0;
    // EPI_QKV with seq > 0 (N > 1, not batched): the rows are N / seq independent sequences; row n is position pos0 + n % seq of sequence
    // n / seq, whose cache starts kv_slot_stride floats behind its predecessor's - or, with seqtab, position seqtab[z].pos0 + n % seq of the
    // cache of slot seqtab[z].slot (rows at or beyond seqtab[z].len are padding and store nothing)
    const SeqTab * seqtab = nullptr;
    BARK_TRACE_FIELD
};
void launch_linear(hipStream_t s, const LinArgs & a);
// lock-step decode of up to 32 slots on the f32 matrix cores (a.batched, f16 rows a.x_f16 [nbatch][K], f16 weights)
void launch_linear_slots(hipStream_t s, const LinArgs & a);                // gemm_slots16_kernel; x_f32 + ln_g: the LayerNorm of the rows fused (when linear_slots_fuses_ln(K))
bool linear_slots_fuses_ln(int K);
// lock steps at few slots: the QKV product per slot as in the single-utterance step, forming the partial scores of the cached keys (a.ps:
// [nbatch][H][4][P]) that attn_fused_ps_kernel finishes; a.x_f32 [nbatch][K] + LayerNorm, batched epilogue; a.ps == nullptr: no copies (FC product)
void launch_linear_slots_ps(hipStream_t s, const LinArgs & a);
// ... and their out-projections as the single-utterance GEMV with a slot dimension (a.x_f16 [nbatch][K])
void launch_linear_slots_gemv(hipStream_t s, const LinArgs & a);

// x[i] = wte[tok] (+ wte[tok2] for merged prompt rows) + wpe[pos]      (bark.cpp:1220-1259)
struct EmbedArgs {
    const half_t * wte = nullptr; const float * wpe = nullptr; int E = 0, n_in = 0, P = 1024;
    QMat wte_q;                           // q4_0 embedding table (rows are dequantised, ggml_get_rows)
    const int32_t * tokens = nullptr;      // n_tokens ids (prefill) - ignored when st != nullptr
    int n_rows = 1; int merge = 0;         // merge: 513 ids -> 257 rows
    int pos0 = 0;
    const StepState * st = nullptr;        // decode: token = st->cur_token, pos = st->n_past
    float * x = nullptr;
};
void launch_embed_causal(hipStream_t s, const EmbedArgs & a);
// fine: x[i] = sum_{c<=nn} wte_c[tok[c][i]] + wpe[i]                   (bark.cpp:1450-1472)
// n_rows rows = n_rows / 1024 windows back to back (row i sits at position i % 1024); tokens [8][plane], plane >= n_rows
void launch_embed_fine(hipStream_t s, const half_t * const wte[8], const QMat * wte_q, const float * wpe, int E, int n_in,
                       const int32_t * tokens, int nn, float * x, int n_rows = 1024, int plane = 1024);

void launch_ln_stats(hipStream_t s, const float * x, int N, int E, float * stats);
// q8 quantisation of N <= 1024 f32 rows of length K (LayerNorm first when ln_g != nullptr)
void launch_q8_rows(hipStream_t s, const float * x, int N, int K, const float * ln_g, const float * ln_b, const Q8Scratch & out);
void launch_ln_rows(hipStream_t s, const float * x, int N, int E, const float * g, const float * b, half_t * out);
void launch_ln_rows_f32(hipStream_t s, const float * x, int N, int E, const float * g, const float * b, float * out);   // no f16 rounding (f32 weights)

// Single-query attention over the KV cache (decode step): q [E] f32, ctx = st->n_past + 1 keys.
struct AttnDecodeArgs {
    const float * q = nullptr; const float * kc = nullptr; const float * vc = nullptr;
    int H = 0, P = 0; const StepState * st = nullptr; half_t * att = nullptr;
    float * att32 = nullptr;              // q4_0 path: attention output kept in f32
    const float * vt = nullptr;           // V in the K layout [H][16][P][4] (attn_ps_kernel)
    int ng = 4;                           // the context holds at most 256 ng keys: those keys are requested at wave launch
    const float * knew = nullptr;         // [E] the K row this step appended (copy at a fixed address, see LinArgs::knew)
    const float * ps = nullptr;           // [H][4][P] partial scores of the cached keys from the QKV kernel (LinArgs::ps); nullptr: scores are formed here
    int nbatch = 1; size_t kv_slot_stride = 0;   // batched decode: slot b uses q/att + b*E, st[b], kc/vc + b*kv_slot_stride
    float * sc = nullptr;                  // lock-step batches: [nbatch][H][P] scores between attn_slots_scores_kernel and attn_slots_mix_kernel (nullptr: attn_fused_kernel)
    BARK_TRACE_FIELD
};
void launch_attn_decode(hipStream_t s, const AttnDecodeArgs & a);      // a.ps set: attn_ps_kernel; otherwise attn_fused_kernel (one workgroup per head and slot)

// Multi-query attention (prefill / fine): N queries at positions n_past.., keys 0..n_past+N-1.
struct AttnPrefillArgs {
    const float * q = nullptr; int ldq = 0; const float * kc = nullptr; const float * vc = nullptr;
    int H = 0, P = 0, N = 0, n_past = 0; int causal = 1;
    half_t * att = nullptr; int ld_att = 0;
    float * att32 = nullptr;              // q4_0 path: attention output kept in f32 (same leading dimension)
    // Z > 1 (fine windows of several utterances in one launch, grid.z): sequence z owns rows [z N, (z + 1) N) of q / att and the cache at
    // kc / vc + z kv_seq_stride
    int Z = 1; size_t kv_seq_stride = 0;
    const SeqTab * seqtab = nullptr;      // ragged causal sequences: N rows per sequence in q / att, sequence z has seqtab[z].len of them, continues
                                          // the cache of slot seqtab[z].slot (kc / vc + slot * kv_seq_stride) at seqtab[z].pos0
};
void launch_attn_prefill(hipStream_t s, const AttnPrefillArgs & a);
// Tolerance route of the fine model's attention (non-causal, whole sequences): flash-style on the f16 matrix cores, operands from EPI_QKV16
struct AttnFlashArgs {
    const half_t * q16 = nullptr, * k16 = nullptr, * vt16 = nullptr;
    int H = 0, E = 0, S = 0, Z = 1;          // S keys / queries per sequence, Z sequences
    half_t * att = nullptr; int ld_att = 0;   // [Z * S][ld_att]
};
void launch_attn_flash(hipStream_t s, const AttnFlashArgs & a);
void launch_linear_fast(hipStream_t s, const LinArgs & a);           // gemm_f16_tile_kernel
void init_fast_attributes();

// Greedy pick (gpt_argmax_sample, bark.cpp:223-247) over logits[0..n); advances *st.
struct SampleArgs {
    const float * logits = nullptr; int n = 0;
    int mode = 0;                          // 0 semantic (stop rule on eos token / eos_p), 1 coarse
    float temp = 0.0f;                     // 0: greedy (gpt_argmax_sample); > 0: multinomial with the uniform draws u[st->step]
    const double * u = nullptr;
    int u_stride = 0;                      // batched decode: slot b reads u + b * u_stride
    float min_eos_p = 0.2f; int eos_token = 10000;
    int prescaled = 0;                     // the logits have already been divided by 0.7 (LinArgs::out_div of the LM head)
    int force_exact = 0;                   // every sample takes the exact path (the reference's sequential float softmax); tests
    int token_base = 0;                    // coarse: added to the pick (slice start); semantic 0
    int n_past_add = 1;                    // rows the evaluated step appended to the KV cache (prefill: N)
    int32_t * out_tokens = nullptr; float * eos_trace = nullptr; StepState * st = nullptr;
    int nbatch = 1; int ld_logits = 0; int out_stride = 0;   // batched decode: slot b reads logits + b*ld_logits, writes out_tokens + b*out_stride, x + b*E
    // lock-step batches whose utterances carry their own parameters: slot b samples with slot_temp[b] (0: greedy) and stops on slot_min_eos_p[b].
    // With slot_temp set the launch runs the greedy kernel for the slots with temperature 0 and / or the multinomial kernel for the others
    // (`kinds`: 1 greedy slots present, 2 sampled slots present); a kernel leaves the other kind's slots untouched.
    const float * slot_temp = nullptr; const float * slot_min_eos_p = nullptr; int kinds = 0;
    // embedding of the sampled token for the NEXT decode step, written by the same kernel (x == nullptr: skip)
    const half_t * wte = nullptr; const float * wpe = nullptr; int E = 0, n_in = 0, P = 1024; float * x = nullptr;
    QMat wte_q;
    BARK_TRACE_FIELD
};
void launch_sample_greedy(hipStream_t s, const SampleArgs & a);
// fine: per-row greedy pick over the first n_cols of each row -> out[i*out_stride]
// fine stage, fine_temp > 0: row r picks with the uniform draw u[r]
// st (optional): near_tie counts the picks settled by the exact path (u within 1e-6 of a bin boundary)
void launch_sample_rows_multinomial(hipStream_t s, const float * logits, int ld, int n_rows, int n_cols, float temp, const double * u,
                                    int32_t * out, int out_stride, StepState * st = nullptr);
void launch_argmax_rows(hipStream_t s, const float * logits, int ld, int n_rows, int n_cols, int32_t * out,
                        int out_stride, StepState * st);

// BARK_HIP_CROSSCHECK (bit mask, read once per process): the slower routes kept to check the default ones against, bit for bit
//   1  N > 1 products through the one-row-per-wave kernels instead of the matrix cores (gemv_rows_kernel / gemm_q_rows_kernel)
//   2  lock-step decode products on the VALU GEMV per pair of slots instead of gemm_slots16_kernel
//   4  decode attention without the QKV kernel's partial scores (attn_fused_kernel instead of attn_ps_kernel)
//   8  every coarse window re-evaluated from its first row (no prefix reuse), as the reference does
//  16  lock-step batches: the prompts of the slots through the model slot by slot instead of all in one pass (batch_prefill_many)
//  32  lock-step batches: the decode attention always as one workgroup per (head, slot) (attn_fused_kernel), also above 256 pairs
//  64  lock-step batches: the decode attention always as the scores / mix pair of launches, also below 256 (head, slot) pairs
// 128  lock-step batches: the LayerNorm of the slot rows as a launch of its own in front of the QKV / FC / LM-head products instead of inside them
// 256  fine model: products as C1 chains on the f32 matrix cores (the canonical order of rounds 1 - 3) instead of C1m on the f16 matrix cores;
//      the oracle follows with set_fine_mfma(False)
// 1024 codec: every convolution as one fmaf chain in (ci, k) order (C9) instead of the f16 matrix cores' order (C9m); oracle: set_codec_mfma(False)
// 512  fine model: the attention of whole windows through attn_rows_kernel (scores in an LDS tile) instead of attn_window_kernel (scores in registers)
int crosscheck_mask();
int xcd_panel_width(int n_tiles, int ncol);            // column-panel width of the XCD-aware tile order (device_utils.h: panel_tile)
void init_kernel_attributes();
// internal: per-file pieces of the above and the weight-type specific back ends of launch_linear
void init_attention_attributes();
void init_quant_attributes();
void launch_linear_q(hipStream_t s, const LinArgs & a);
void launch_linear_w32(hipStream_t s, const LinArgs & a);

// ---- EnCodec decoder (codec_kernels.hip) ---------------------------------------------------------
// Activations are time-major [row][C] f32 (row = frame at the stage's rate); every conv / LSTM matmul consumes them rounded to f16
// (ggml im2col / mul_mat) and accumulates in f32: convolutions in the order of the f16 matrix cores over
// kd = k * cin + ci (C9m; C9 - one fmaf chain in (ci, k) order - where the input channel count is not a multiple of 8), bias added last.
// Several utterances in one launch: utterance z has T[z] frames and owns rows [tm Tpre[z], tm Tpre[z + 1]) of every activation at a stage
// with upsampling factor tm (Tpre = exclusive prefix sums of T).  T == nullptr: one utterance whose length is the launch's T argument.
struct CodecBatch { const int * T = nullptr; const int * Tpre = nullptr; int B = 1; };
void launch_rvq_gather(hipStream_t s, const float * codebooks, int n_bins, int Hd, const int32_t * codes, int n_q, int T, int rows_total, float * z, const CodecBatch & cb);
// out_h = f16(elu ? ELU(x) : x), n elements
void launch_act_round(hipStream_t s, const float * x, size_t n, int elu, half_t * out_h);
// One convolution (convT = 0: causal stride-1 conv with reflect padding on the left, K taps) or transposed convolution (convT = 1: causal,
// stride nphase, kernel 2 nphase, output trimmed to rows_in * nphase) over time-major rows:
//   y[orow][co] = bias[co] + dot(kernel image of co, operand rows) (+ add[orow][co]);   orow = row (conv) or row * nphase + phase (convT)
// W: kernel image for C9m [nphase][cout32][kd16] f16, zero padded (nullptr: order C9 on w32, the f32 copy of the file's kernel -
// conv [cout][cin][K], convT [cin][cout][K]); outputs: any of y (f32), yh_raw (f16), yh_elu (f16 of ELU) - what the next operators consume
struct ConvTmArgs {
    const half_t * W = nullptr; const float * w32 = nullptr; const float * bias = nullptr;
    int cin = 0, cout = 0, cout32 = 0, K = 0, kd = 0, kd16 = 0, nphase = 1, convT = 0;
    const half_t * xh = nullptr; const float * add = nullptr;
    float * y = nullptr; half_t * yh_raw = nullptr, * yh_elu = nullptr;
    int rows_in = 0, tm_in = 1, T_single = 0;
    CodecBatch cb;
};
void launch_conv_tm(hipStream_t s, const ConvTmArgs & a);
// Both LSTM layers as a wave front (PyTorch gate order i,f,g,o): launch i runs layer 1 at step i and layer 2 at step i - 1 (whose input
// projection W_ih2 h1[i-1] is formed in the same kernel, C1 order - the bits of the row-batched product), so a sequence of T frames takes
// T + 1 dependent launches instead of 2 T - for every utterance of the batch at once.
// Layer 1: gi1 = W_ih1 x for all t (launch_linear), w_hh1, biases, cells c1 [B][D], f16 outputs h1 [sum T][D].
// Layer 2: w_ih2, w_hh2, biases, cells c2, f16 outputs h2 and the f32 sequences out2 ([D][T] per utterance).
struct LstmPairArgs {
    const float * gi1 = nullptr; const half_t * w_hh1 = nullptr; const float * b_ih1 = nullptr, * b_hh1 = nullptr; float * c1 = nullptr; half_t * h1 = nullptr;
    const half_t * w_ih2 = nullptr, * w_hh2 = nullptr; const float * b_ih2 = nullptr, * b_hh2 = nullptr; float * c2 = nullptr; half_t * h2 = nullptr;
    float * out2 = nullptr; int T = 0, D = 0;
    int t = 0; const int * t_base = nullptr;          // launch index, or offset inside a replayed block added to t_base[0] (then t_base[1] holds T)
    CodecBatch cb;
};
void launch_lstm_pair_step(hipStream_t s, const LstmPairArgs & a);
void launch_add_int(hipStream_t s, int * p, int v);       // *p += v
void launch_add(hipStream_t s, const float * a, const float * b, size_t n, float * out);

}  // namespace barkhip
