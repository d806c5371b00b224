// kernels.hip - linear operators with f16 weights of the MI355X-native Bark engine (gfx950 / CDNA4, wave64).
//
// Every kernel follows the canonical numerics of DESIGN.md (orders C1/C2/C5, explicit fmaf, double
// accumulated LayerNorm / softmax sums, f16 rounding points of ggml's CPU backend) so that the CPU
// oracle reproduces its results bit for bit.  Built with -ffp-contract=off.
//
//   decode (N = 1)      : gemv_kernel (16 lanes per output row = the 16 chains of C1, coalesced
//                         16-byte weight loads, LayerNorm fused as prologue, bias / residual /
//                         GELU-LUT / KV-append fused as epilogue), gemv_batch_kernel (lock-step batch).
//   prefill / fine (N>1): gemm_kernel (v_mfma_f32_32x32x2_f32: exact f32 fma chains; the 16 chains
//                         of C1 live in 8 waves x 2 accumulator sets and meet in LDS).
// Other kernel files: quant_kernels.hip (block-quantised and f32 weights), attention_kernels.hip (decode / prefill
// attention), misc_kernels.hip (embeddings, LayerNorm rows, sampling), codec_kernels.hip (EnCodec decoder).
#include "device_utils.h"

#include <algorithm>
#include <cfloat>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <stdexcept>

namespace barkhip {

int crosscheck_mask() {
    static const int mask = getenv("BARK_HIP_CROSSCHECK") ? atoi(getenv("BARK_HIP_CROSSCHECK")) : 0;
    return mask;
}

int xcd_panel_width(int n_tiles, int ncol) {
    // one XCD works on n_tiles / 8 consecutive tiles: a near-square block of them shares the fewest operand slices
    int pw = 1;
    while (pw * pw < (n_tiles + 7) / 8) pw++;
    return std::max(1, std::min(pw, ncol));
}

void kernel_fail(const char * fmt, ...) {
    char buf[256];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    throw std::runtime_error(buf);
}

// ------------------------------------------------------------------------------------------------
// decode GEMV.  One wave = 4 output rows x 16 lanes; lane c of a row owns chain c of C1, i.e. the
// 16-byte chunks c, c+16, c+32, ... of that weight row: the wave's loads are four fully used
// 256-byte row segments per instruction.  x is an f16 vector (the LayerNorm-fused products use gemv_ln_wg_kernel below).
// ------------------------------------------------------------------------------------------------
// NBLK = K / 128 is a compile-time constant so that every load of a lane (weights, x, LayerNorm
// parameters) is issued up front with no control flow in between: the kernel is one memory round
// trip deep.  One wave per workgroup (4 output rows) spreads the rows over as many CUs as possible.
// The operands the first loads need (weight / input pointers, row count, row window) are explicit leading kernel parameters: built
// with -amdgpu-kernarg-preload-count the hardware delivers them in SGPRs at wave launch, so the weight stream is requested without
// the ~0.2 us kernel-argument round trip the in-kernel time line shows in front of every kernel; the struct carries the rest.
template <int NBLK>
__global__ __launch_bounds__(64) void gemv_kernel(const half_t * __restrict__ W, const half_t * __restrict__ x_f16, const int M, const int parity_rows, const LinArgs a) {
    TRACE_T0();
    TRACE_T1(M);
    const int lane = threadIdx.x;
    const int c = lane & 15, rg = lane >> 4;
    const int m = blockIdx.x * 4 + rg;
    const int row_off = parity_rows ? parity_rows * (a.st->step & 1) : 0;
    const bool live = m < M;                                // whole 16-lane groups are live or dead together
    constexpr int K = NBLK * 128;
    const half_t * wrow = W + (size_t) (row_off + (live ? m : 0)) * K + (c << 3);
    const half_t * xrow = x_f16 + (c << 3);
    // every chunk of the row is requested before the first fmaf (up to 2 x 32 x 16 bytes per lane: a one-wave workgroup may
    // use all 512 registers).  The chain of a lane is K / 16 dependent fmaf long whatever the load schedule; with batches of
    // 8 chunks the K = 3072 product paid two more exposed memory round trips (1.8 us between "arguments ready" and "dot done").
    half8 wv[NBLK], xv[NBLK];
    #pragma unroll
    for (int i = 0; i < NBLK; i++) { wv[i] = ld_half8_w(wrow + (i << 7)); xv[i] = ld_half8(xrow + (i << 7)); }
    __builtin_amdgcn_sched_barrier(0);                    // the streams above go out on the preloaded arguments alone; the struct is read behind them
    const EpiPre pre = epilogue_prefetch(a, 0, live ? m : 0, row_off);
    float acc = 0.0f;
    #pragma unroll
    for (int i = 0; i < NBLK; i++) {
        #pragma unroll
        for (int e = 0; e < 8; e++) acc = fmaf((float) wv[i][e], (float) xv[i][e], acc);
    }
    TRACE_T2(acc);
    acc = wave_xor_add16(acc);
    if (live && c == 0) linear_epilogue_pre(a, 0, m, acc, pre);
    TRACE_END(a.tr);
}


// ------------------------------------------------------------------------------------------------
// LayerNorm-fused decode GEMV, workgroup form.  The in-kernel time line (tools/trace_decode.py) showed the one-wave form above
// spending 2.2 us between "arguments ready" and "dot product done" where the plain GEMV needs 0.6: every wave normalised the
// whole row on its own - ~1500 dependent VALU instructions (fp64 sums, two fp64 divisions, 48 scale / round steps per lane) on a
// SIMD that holds nothing else.  Here the 256 threads of a workgroup (4 waves = 16 output rows) normalise the row ONCE: E / 256
// elements per thread, wave sums by DPP, the four wave sums meet in LDS, and the f16-rounded row is published in LDS (1.5 KB) from
// where every lane reads its 16-byte chunks.  Same arithmetic per element as before (ggml_norm: double sums, bark.cpp:1265-1274).
// ------------------------------------------------------------------------------------------------
// PS = true (QKV product of a decode step): the 16 rows of a workgroup below n_embd are one 16-d block of one head's q.  C2 sums a
// score as four such blocks, so this workgroup can form its block's partial score against every cached key right here, where q is
// born: the K stream of the attention (the per-CU bandwidth bound of every fused decode-attention kernel tried) is spread over the
// 4 H q-workgroups of this launch, 64 bytes per key each, requested together with the weights.
// Leading parameters = what the first loads need (preloaded into SGPRs at wave launch, see gemv_kernel): every stream of this kernel -
// weights, the f32 row, LayerNorm parameters, the K quads of the partial scores - is requested before the argument struct is read.
template <int NBLK, bool LNB, bool PS>
__global__ __launch_bounds__(256) void gemv_ln_wg_kernel(const half_t * __restrict__ W, const float * __restrict__ x_f32, const float * __restrict__ ln_g,
                                                         const float * __restrict__ ln_b, const float * __restrict__ kc, const StepState * __restrict__ st, const int M,
                                                         const int parity_rows, const int E, const int kpc, const LinArgs a) {
    TRACE_T0();
    TRACE_T1(M);
    constexpr int K = NBLK * 128;
    constexpr int EPT = K / 64;                              // row elements per lane of the normalising wave
    __shared__ __attribute__((aligned(16))) half_t xs[K];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 15, rg = lane >> 4;
    // PS: behind the M / 16 main workgroups the grid carries copies of the q workgroups.  Copy r repeats the LayerNorm and the 16 q rows
    // (its weights come out of the XCD's L2: ids congruent mod 8 share it with the main workgroup) and scores the keys r kpc .. r kpc +
    // kpc - 1; it writes nothing but partial scores.  A CU pulls only ~24 bytes/ns from the memory side and the launch is as slow as its
    // busiest CU: nobody carries both the 24 KB of weights of 16 rows and a K stream, and the host sizes kpc (256 .. 512 keys, 64 bytes
    // each) so that main workgroups + copies do not outnumber the CUs (two workgroups on one CU were the long pole before).
    [[maybe_unused]] const int n_main = (M + 15) >> 4, n_q = E >> 4;
    [[maybe_unused]] const bool copy = PS && (int) blockIdx.x >= n_main;
    [[maybe_unused]] const int rep = copy ? ((int) blockIdx.x - n_main) / n_q : 0;
    const int wg = copy ? ((int) blockIdx.x - n_main) % n_q : (int) blockIdx.x;
    const int m = (wg * 4 + wave) * 4 + rg;
    const int row_off = parity_rows ? parity_rows * (st->step & 1) : 0;
    const bool live = m < M;
    const half_t * wrow = W + (size_t) (row_off + (live ? m : 0)) * K + (c << 3);
    [[maybe_unused]] unsigned long long stamp_a = 0, stamp_b = 0;       // diagnostic build: row arrived, normalised row written
    // the row to normalise is requested FIRST: loads return in order, and behind 6 KB of weights per lane group the 3 KB row arrived
    // after 1.3 us (in-kernel time line) - the whole LayerNorm waited for the weight stream it was meant to overlap
    float xv[EPT], gv[EPT], bv[EPT];
    if (wave == 0) {
        #pragma unroll
        for (int i = 0; i < EPT; i++) xv[i] = x_f32[lane + 64 * i];
        #pragma unroll
        for (int i = 0; i < EPT; i++) {
            gv[i] = ln_g[lane + 64 * i];
            if constexpr (LNB) bv[i] = ln_b[lane + 64 * i]; else bv[i] = 0.0f;
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    half8 wv[NBLK];
    #pragma unroll
    for (int b = 0; b < NBLK; b++) wv[b] = ld_half8_w(wrow + (b << 7));
    // partial scores: keys rep * kpc + tid (+ 256); d-quads 4 blk .. 4 blk + 3 of head hq
    [[maybe_unused]] float4 kq[2][4];
    [[maybe_unused]] const int m0 = wg * 16;
    [[maybe_unused]] const bool is_q = copy;
    [[maybe_unused]] const int hq = m0 >> 6, blk = (m0 >> 4) & 3;
    // the copies' K quads are requested only when the normalised row is in LDS: asked for at wave launch, 2 MB of K in front of every
    // memory channel delayed the 3 KB row all workgroups wait for by 0.6 us (in-kernel time line: row arrived at 1.4 us against 0.8 us
    // without the K stream); they are needed last, after the dot product
    auto load_kq = [&] {
        const BufRsrc kr = buf_rsrc(reinterpret_cast<const float4 *>(kc) + ((size_t) hq * 16 + 4 * blk) * 1024 + rep * kpc);     // PS implies P == 1024
        #pragma unroll
        for (int i = 0; i < 4; i++) kq[0][i] = buf_ld_f4(kr, (unsigned) tid * 16u, (unsigned) i * 16384u);    // rows beyond n_past hold stale bits; their scores are not stored
        if (tid + 256 < kpc) {
            #pragma unroll
            for (int i = 0; i < 4; i++) kq[1][i] = buf_ld_f4(kr, (unsigned) tid * 16u + 4096u, (unsigned) i * 16384u);
        }
    };
    // the context length (EPI_QKV: where the K / V rows go, which copies have keys) is requested on the preloaded state pointer, not
    // through the argument struct: two dependent scalar round trips in front of the LayerNorm cost the QKV kernel 0.6 us
    const int n_past_now = st ? st->n_past : 0;
    __builtin_amdgcn_sched_barrier(0);                        // everything above goes out on the preloaded arguments alone
    EpiPre pre = epilogue_prefetch(a, 0, live ? m : 0, row_off);
    if (PS || a.epi == EPI_QKV) pre.n_past = n_past_now;      // same value epilogue_prefetch reads through the struct (that load is dead now)
    // (a copy whose keys are not in the context yet runs to the end and stores nothing: leaving early would put the arrival of the
    // context length in front of the LayerNorm; the host launches only the copies the context bound needs)
    if (wave == 0) {
        // ggml_norm (+mul, +add): double sums, eps on the variance (bark.cpp:1265-1274); four partial sums per lane keep the fp64
        // chains short (the order of a double sum of floats changes the rounded float result with probability ~2^-29)
        double p1[4] = {0.0, 0.0, 0.0, 0.0};
        #pragma unroll
        for (int i = 0; i < EPT; i++) p1[i & 3] += (double) xv[i];
        TRACE_SET(stamp_a, (float) p1[0]);
        const double s1 = wave_sum((p1[0] + p1[1]) + (p1[2] + p1[3]));
        const float mean = (float) div_by_const<K>(s1);
        double p2[4] = {0.0, 0.0, 0.0, 0.0};
        #pragma unroll
        for (int i = 0; i < EPT; i++) { xv[i] = xv[i] - mean; p2[i & 3] += (double) (xv[i] * xv[i]); }
        const double s2 = wave_sum((p2[0] + p2[1]) + (p2[2] + p2[3]));
        const float var = (float) div_by_const<K>(s2);
        const float scale = 1.0f / sqrtf(var + 1e-5f);
        #pragma unroll
        for (int i = 0; i < EPT; i++) {
            float v = xv[i] * scale;
            v = v * gv[i];
            if constexpr (LNB) v = v + bv[i];
            xs[lane + 64 * i] = to_half(v);                    // mul_mat converts the activation to f16 first
        }
        TRACE_SET(stamp_b, scale);
    }
    __syncthreads();
    if constexpr (PS) { if (is_q) load_kq(); }
    float acc = 0.0f;
    #pragma unroll
    for (int b = 0; b < NBLK; b++) {
        const half8 xh = *reinterpret_cast<const half8 *>(xs + ((b * 16 + c) << 3));
        #pragma unroll
        for (int e = 0; e < 8; e++) acc = fmaf((float) wv[b][e], (float) xh[e], acc);
    }
    TRACE_T2(acc);
    acc = wave_xor_add16(acc);
    if (live && c == 0 && !copy) linear_epilogue_pre(a, 0, m, acc, pre);
    if constexpr (PS) {
        // the copies cover the keys below (copies per q block) x kpc; a launch whose bound on the context was too small must not pass silently
        if (blockIdx.x == 0 && tid == 0 && pre.n_past > (((int) gridDim.x - n_main) / n_q) * kpc) const_cast<StepState *>(st)->fault = 1;
        __shared__ float qs[16];
        if (is_q) {                                              // uniform per workgroup
            if (c == 0) qs[wave * 4 + rg] = a.bias ? acc + pre.bias : acc;       // the q value the epilogue stores
            __syncthreads();
            float qb[16];
            #pragma unroll
            for (int i = 0; i < 16; i++) qb[i] = qs[i];
            const int j = rep * kpc + tid;
            if (j < pre.n_past) a.ps[((size_t) hq * 4 + blk) * a.P + j] = score_block_f4(kq[0], qb);             // [H][4][P]: a workgroup's keys are contiguous
            if (tid + 256 < kpc && j + 256 < pre.n_past) a.ps[((size_t) hq * 4 + blk) * a.P + j + 256] = score_block_f4(kq[1], qb);
        }
    }
#ifdef BARK_TRACE
    trace_emit(a.tr, _tr0, _tr1, _tr2, trace_clock(), stamp_a, stamp_b);
#endif
}

template <int NBLK>
static void launch_gemv_n(hipStream_t s, const LinArgs & a) {
    if (!a.x_f32) {
        hipLaunchKernelGGL((gemv_kernel<NBLK>), dim3((a.M + 3) / 4), dim3(64), 0, s, a.W, a.x_f16, a.M, a.parity_rows, a);
        return;
    }
    if constexpr (NBLK <= 8) {
        const dim3 g16((a.M + 15) / 16), b256(256);
        const float * kc = a.kc;
        if (a.ps && a.epi == EPI_QKV && a.P == 1024) {
            // copies of the q workgroups: as many per q block as fit beside the main workgroups on 256 CUs (at most 2), each
            // scoring kpc = 256 or 512 of the up to 256 ng keys the context may hold
            const int n_main = (a.M + 15) / 16, n_q = a.E / 16, keys = 256 * std::max(1, std::min(a.ng, 4));
            const int fit = std::max(1, std::min(2, (256 - n_main) / n_q));
            const int n_copy = std::max((keys + 511) / 512, std::min(fit, keys / 256));      // a copy scores at most 512 keys (two per thread)
            const int kpc = ((keys + n_copy - 1) / n_copy + 127) / 128 * 128;                 // 256, 384 or 512
            const dim3 gps(n_main + n_copy * n_q);
            if (a.ln_b) hipLaunchKernelGGL((gemv_ln_wg_kernel<NBLK, true, true>), gps, b256, 0, s, a.W, a.x_f32, a.ln_g, a.ln_b, kc, a.st, a.M, a.parity_rows, a.E, kpc, a);
            else        hipLaunchKernelGGL((gemv_ln_wg_kernel<NBLK, false, true>), gps, b256, 0, s, a.W, a.x_f32, a.ln_g, a.ln_b, kc, a.st, a.M, a.parity_rows, a.E, kpc, a);
        }
        else if (a.ln_b) hipLaunchKernelGGL((gemv_ln_wg_kernel<NBLK, true, false>), g16, b256, 0, s, a.W, a.x_f32, a.ln_g, a.ln_b, kc, a.st, a.M, a.parity_rows, a.E, 0, a);
        else             hipLaunchKernelGGL((gemv_ln_wg_kernel<NBLK, false, false>), g16, b256, 0, s, a.W, a.x_f32, a.ln_g, a.ln_b, kc, a.st, a.M, a.parity_rows, a.E, 0, a);
    } else { kernel_fail("bark-hip: LayerNorm-fused GEMV supports n_embd <= 1024"); }
}

// ------------------------------------------------------------------------------------------------
// Lock steps at FEW slots (engine_batch.hip: kFewSlotsScores / kFewSlotsProducts; measured in round 5, profiles/r05_few_slot_routes_*.txt): the QKV product as gemv_ln_wg_kernel<PS>
// with a slot dimension (blockIdx.y).  At 8 slots a lock step is a latency chain whose longest link is the attention: one workgroup per
// (head, slot, value half) pulls the pair's whole K (164 KB at 640 keys) through one CU to form scores.  The single-utterance step does not:
// the workgroup that produces 16 consecutive q values also forms that C2 block's partial score against every cached key, spread over the
// q workgroups' copies.  This kernel does the same per slot: slot b's workgroups normalise ITS row, read the same weight rows (from the XCD's L2
// after the first slot: ids congruent mod 8 share it, grid.x is a multiple of 8), write q / K / V of slot b through the batched epilogue and the
// partial scores to ps + b * H * 4 * P; attn_fused_ps_kernel (attention_kernels.hip) finishes them.  Same arithmetic per element as
// gemv_ln_wg_kernel (C6 LayerNorm, C1 chains, C2 blocks): bit-equal to the lock step's gemm_slots16_kernel + attn_fused_kernel route.
// A copy whose keys lie beyond its slot's context leaves once the context length has arrived (the lock step's graph is captured for any context,
// so copies for all 1024 keys are launched).
// ------------------------------------------------------------------------------------------------
// PS = false: the same per-slot LayerNorm-fused product without copies and partial scores, for the FC product of a lock step at few slots
// (any batched epilogue) - ahead of gemm_slots16_kernel<LNF> up to 16 slots.
template <int NBLK, bool LNB, bool PS>
__global__ __launch_bounds__(256) void gemv_ln_slots_ps_kernel(const half_t * __restrict__ W, const float * __restrict__ X, const float * __restrict__ ln_g,
                                                               const float * __restrict__ ln_b, const float * __restrict__ kc0, const StepState * __restrict__ st0, const int M,
                                                               const int E, const int kpc, const LinArgs a) {
    constexpr int K = NBLK * 128;
    constexpr int EPT = K / 64;
    __shared__ __attribute__((aligned(16))) half_t xs[K];
    __shared__ float qs[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 15, rg = lane >> 4;
    const int slot = blockIdx.y;
    const float * __restrict__ x_f32 = X + (size_t) slot * K;
    const StepState * __restrict__ st = st0 + slot;
    const float * __restrict__ kc = kc0 + (size_t) slot * a.kv_slot_stride;
    [[maybe_unused]] const int n_main = (M + 15) >> 4, n_q = PS ? E >> 4 : 1;
    const bool copy = PS && (int) blockIdx.x >= n_main;
    [[maybe_unused]] const int rep = copy ? ((int) blockIdx.x - n_main) / n_q : 0;
    const int wg = copy ? ((int) blockIdx.x - n_main) % n_q : (int) blockIdx.x;
    const int m = (wg * 4 + wave) * 4 + rg;
    const bool live = m < M;
    const half_t * wrow = W + (size_t) (live ? m : 0) * K + (c << 3);
    float xv[EPT], gv[EPT], bv[EPT];
    if (wave == 0) {
        #pragma unroll
        for (int i = 0; i < EPT; i++) xv[i] = x_f32[lane + 64 * i];
        #pragma unroll
        for (int i = 0; i < EPT; i++) {
            gv[i] = ln_g[lane + 64 * i];
            if constexpr (LNB) bv[i] = ln_b[lane + 64 * i]; else bv[i] = 0.0f;
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    half8 wv[NBLK];
    #pragma unroll
    for (int b = 0; b < NBLK; b++) wv[b] = ld_half8_w(wrow + (b << 7));
    [[maybe_unused]] float4 kq[2][4];
    [[maybe_unused]] const int m0 = wg * 16;
    [[maybe_unused]] const int hq = m0 >> 6, blk = (m0 >> 4) & 3;
    [[maybe_unused]] int n_past = 0;
    if constexpr (PS) n_past = st->n_past;
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (PS) { if (copy && rep * kpc >= n_past) return; }      // uniform per workgroup: none of this copy's keys is cached yet
    const EpiPre pre = epilogue_prefetch(a, slot, live ? m : 0, 0);       // batched: bias, the slot's context length
    if (wave == 0) {
        // ggml_norm (+mul, +add) exactly as gemv_ln_wg_kernel: double sums in four partial chains per lane, Markstein division by the row length
        double p1[4] = {0.0, 0.0, 0.0, 0.0};
        #pragma unroll
        for (int i = 0; i < EPT; i++) p1[i & 3] += (double) xv[i];
        const double s1 = wave_sum((p1[0] + p1[1]) + (p1[2] + p1[3]));
        const float mean = (float) div_by_const<K>(s1);
        double p2[4] = {0.0, 0.0, 0.0, 0.0};
        #pragma unroll
        for (int i = 0; i < EPT; i++) { xv[i] = xv[i] - mean; p2[i & 3] += (double) (xv[i] * xv[i]); }
        const double s2 = wave_sum((p2[0] + p2[1]) + (p2[2] + p2[3]));
        const float var = (float) div_by_const<K>(s2);
        const float scale = 1.0f / sqrtf(var + 1e-5f);
        #pragma unroll
        for (int i = 0; i < EPT; i++) {
            float v = xv[i] * scale;
            v = v * gv[i];
            if constexpr (LNB) v = v + bv[i];
            xs[lane + 64 * i] = to_half(v);
        }
    }
    __syncthreads();
    if constexpr (PS) {
        if (copy) {
            const BufRsrc kr = buf_rsrc(reinterpret_cast<const float4 *>(kc) + ((size_t) hq * 16 + 4 * blk) * 1024 + rep * kpc);     // P == 1024
            #pragma unroll
            for (int i = 0; i < 4; i++) kq[0][i] = buf_ld_f4(kr, (unsigned) tid * 16u, (unsigned) i * 16384u);
            if (tid + 256 < kpc) {
                #pragma unroll
                for (int i = 0; i < 4; i++) kq[1][i] = buf_ld_f4(kr, (unsigned) tid * 16u + 4096u, (unsigned) i * 16384u);
            }
        }
    }
    float acc = 0.0f;
    #pragma unroll
    for (int b = 0; b < NBLK; b++) {
        const half8 xh = *reinterpret_cast<const half8 *>(xs + ((b * 16 + c) << 3));
        #pragma unroll
        for (int e = 0; e < 8; e++) acc = fmaf((float) wv[b][e], (float) xh[e], acc);
    }
    acc = wave_xor_add16(acc);
    if (live && c == 0 && !copy) linear_epilogue_pre(a, slot, m, acc, pre);
    if constexpr (PS) {
        // the copies cover the keys below (copies per q block) x kpc; a launch whose bound on the context was too small must not pass silently
        if (blockIdx.x == 0 && tid == 0 && n_past > (((int) gridDim.x - n_main) / n_q) * kpc) const_cast<StepState *>(st)->fault = 1;
        if (copy) {                                              // uniform per workgroup
            if (c == 0) qs[wave * 4 + rg] = a.bias ? acc + pre.bias : acc;       // the q value the epilogue stores
            __syncthreads();
            float qb[16];
            #pragma unroll
            for (int i = 0; i < 16; i++) qb[i] = qs[i];
            float * __restrict__ psl = a.ps + (size_t) slot * (size_t) (E >> 6) * 4 * a.P + ((size_t) hq * 4 + blk) * a.P;
            const int j = rep * kpc + tid;
            if (j < n_past) psl[j] = score_block_f4(kq[0], qb);
            if (tid + 256 < kpc && j + 256 < n_past) psl[j + 256] = score_block_f4(kq[1], qb);
        }
    }
}

// The plain decode GEMV (gemv_kernel: one wave = 4 output rows, every chunk of the row requested up front) with a slot dimension - the out-projections of
// a lock step at few slots (ahead of gemv_batch_kernel, two slots per wave sharing the weight loads, up to 16 slots).  Slot b reads
// its f16 row X + b K and finishes through the batched epilogue; the weight rows come out of the XCD's L2 after the first slot (grid.x is a multiple of 8
// for every bark shape).  Same C1 chains.
template <int NBLK>
__global__ __launch_bounds__(64) void gemv_slots_kernel(const half_t * __restrict__ W, const half_t * __restrict__ X, const int M, const LinArgs a) {
    const int lane = threadIdx.x;
    const int c = lane & 15, rg = lane >> 4;
    const int m = blockIdx.x * 4 + rg;
    const int slot = blockIdx.y;
    const bool live = m < M;
    constexpr int K = NBLK * 128;
    const half_t * wrow = W + (size_t) (live ? m : 0) * K + (c << 3);
    const half_t * xrow = X + (size_t) slot * K + (c << 3);
    half8 wv[NBLK], xv[NBLK];
    #pragma unroll
    for (int i = 0; i < NBLK; i++) { wv[i] = ld_half8_w(wrow + (i << 7)); xv[i] = ld_half8(xrow + (i << 7)); }
    __builtin_amdgcn_sched_barrier(0);
    const EpiPre pre = epilogue_prefetch(a, slot, live ? m : 0, 0);
    float acc = 0.0f;
    #pragma unroll
    for (int i = 0; i < NBLK; i++) {
        #pragma unroll
        for (int e = 0; e < 8; e++) acc = fmaf((float) wv[i][e], (float) xv[i][e], acc);
    }
    acc = wave_xor_add16(acc);
    if (live && c == 0) linear_epilogue_pre(a, slot, m, acc, pre);
}
template <int NBLK>
static void launch_gemv_slots_n(hipStream_t s, const LinArgs & a) {
    hipLaunchKernelGGL((gemv_slots_kernel<NBLK>), dim3((a.M + 3) / 4, a.nbatch), dim3(64), 0, s, a.W, a.x_f16, a.M, a);
}
void launch_linear_slots_gemv(hipStream_t s, const LinArgs & a) {
    if (!a.batched || !a.x_f16 || a.x_f32 || !a.W || a.wq.qs || (a.K & 127) != 0 || a.K > 4096 || a.parity_rows || a.epi == EPI_QKV)
        kernel_fail("bark-hip: the per-slot plain GEMV takes f16 rows, f16 weights and a residual / GELU / logits epilogue");
    switch (a.K >> 7) {
        case 1:  launch_gemv_slots_n<1>(s, a); break;
        case 2:  launch_gemv_slots_n<2>(s, a); break;
        case 4:  launch_gemv_slots_n<4>(s, a); break;
        case 6:  launch_gemv_slots_n<6>(s, a); break;
        case 8:  launch_gemv_slots_n<8>(s, a); break;
        case 16: launch_gemv_slots_n<16>(s, a); break;
        case 24: launch_gemv_slots_n<24>(s, a); break;
        case 32: launch_gemv_slots_n<32>(s, a); break;
        default: kernel_fail("bark-hip: unsupported K=%d in the per-slot plain GEMV", a.K);
    }
}

template <int NBLK>
static void launch_slots_ps_n(hipStream_t s, const LinArgs & a) {
    if constexpr (NBLK <= 8) {
        const int n_main = (a.M + 15) / 16, n_q = a.E / 16;
        // the lock step's graph serves every context: copies for all 1024 keys, two per q block of 512 keys each (as the single-utterance launch at ng = 4)
        const int n_copy = 2, kpc = 512;
        const dim3 grid(n_main + n_copy * n_q, a.nbatch), b256(256);
        if (!a.ps) {
            const dim3 g0(n_main, a.nbatch);
            if (a.ln_b) hipLaunchKernelGGL((gemv_ln_slots_ps_kernel<NBLK, true, false>), g0, b256, 0, s, a.W, a.x_f32, a.ln_g, a.ln_b, a.kc, a.st, a.M, a.E, 0, a);
            else        hipLaunchKernelGGL((gemv_ln_slots_ps_kernel<NBLK, false, false>), g0, b256, 0, s, a.W, a.x_f32, a.ln_g, a.ln_b, a.kc, a.st, a.M, a.E, 0, a);
        }
        else if (a.ln_b) hipLaunchKernelGGL((gemv_ln_slots_ps_kernel<NBLK, true, true>), grid, b256, 0, s, a.W, a.x_f32, a.ln_g, a.ln_b, a.kc, a.st, a.M, a.E, kpc, a);
        else             hipLaunchKernelGGL((gemv_ln_slots_ps_kernel<NBLK, false, true>), grid, b256, 0, s, a.W, a.x_f32, a.ln_g, a.ln_b, a.kc, a.st, a.M, a.E, kpc, a);
    } else { kernel_fail("bark-hip: the per-slot partial-score QKV product supports n_embd <= 1024"); }
}
void launch_linear_slots_ps(hipStream_t s, const LinArgs & a) {
    if (!a.batched || !a.x_f32 || !a.ln_g || !a.W || a.wq.qs || (a.K & 127) != 0 || a.knew || a.vt || a.parity_rows)
        kernel_fail("bark-hip: the per-slot LayerNorm-fused product takes f32 rows + LayerNorm and f16 weights, K %% 128 == 0");
    if (a.ps && (a.epi != EPI_QKV || a.P != 1024 || a.M != 3 * a.E || a.K != a.E || !a.st))
        kernel_fail("bark-hip: the per-slot partial-score QKV product needs block_size 1024");
    if (a.epi == EPI_QKV && !a.st) kernel_fail("bark-hip: a QKV product needs the slots' states");
    switch (a.K >> 7) {
        case 1: launch_slots_ps_n<1>(s, a); break;
        case 2: launch_slots_ps_n<2>(s, a); break;
        case 4: launch_slots_ps_n<4>(s, a); break;
        case 6: launch_slots_ps_n<6>(s, a); break;
        case 8: launch_slots_ps_n<8>(s, a); break;
        default: kernel_fail("bark-hip: unsupported K=%d in the per-slot partial-score QKV product", a.K);
    }
}

// Batched decode GEMV (several utterances in lock step): grid.y walks the sequence slots, BPW slots per wave.
// The weight rows of a workgroup column are read from HBM once (same XCD L2 for every grid.y: grid.x is a multiple
// of 8) and each slot's dot product is the same C1 chain as in gemv_kernel, so results do not depend on the batch.
template <int NBLK, bool LN, bool LNB, int BPW>
__global__ __launch_bounds__(64) void gemv_batch_kernel(const LinArgs a) {
    const int lane = threadIdx.x;
    const int c = lane & 15, rg = lane >> 4;
    const int m = blockIdx.x * 4 + rg;
    const int b0 = blockIdx.y * BPW;
    constexpr int K = NBLK * 128;
    const int row_off = a.parity_rows ? a.parity_rows * (a.st->step & 1) : 0;
    const bool live = m < a.M;
    const half_t * wrow = a.W + (size_t) (row_off + (live ? m : 0)) * K + (c << 3);
    EpiPre pre[BPW];
    int bi[BPW];
    #pragma unroll
    for (int bb = 0; bb < BPW; bb++) { bi[bb] = min(b0 + bb, a.nbatch - 1); pre[bb] = epilogue_prefetch(a, bi[bb], live ? m : 0, row_off); }
    float acc[BPW];
    #pragma unroll
    for (int bb = 0; bb < BPW; bb++) acc[bb] = 0.0f;

    if constexpr (LN) {
        static_assert(NBLK <= 8, "LayerNorm-fused GEMV keeps the row in registers (n_embd <= 1024)");
        half8 wv[NBLK];
        float4 xa[BPW][NBLK][2], ga[NBLK][2], ba[NBLK][2];
        #pragma unroll
        for (int b = 0; b < NBLK; b++) {
            const int k0 = (b * 16 + c) << 3;
            wv[b] = ld_half8(wrow + (b << 7));
            #pragma unroll
            for (int bb = 0; bb < BPW; bb++) {
                const float * xrow = a.x_f32 + (size_t) bi[bb] * K;
                xa[bb][b][0] = *reinterpret_cast<const float4 *>(xrow + k0); xa[bb][b][1] = *reinterpret_cast<const float4 *>(xrow + k0 + 4);
            }
            ga[b][0] = *reinterpret_cast<const float4 *>(a.ln_g + k0);  ga[b][1] = *reinterpret_cast<const float4 *>(a.ln_g + k0 + 4);
            if constexpr (LNB) { ba[b][0] = *reinterpret_cast<const float4 *>(a.ln_b + k0); ba[b][1] = *reinterpret_cast<const float4 *>(a.ln_b + k0 + 4); }
        }
        #pragma unroll
        for (int bb = 0; bb < BPW; bb++) {
            float xr[NBLK][8];
            #pragma unroll
            for (int b = 0; b < NBLK; b++) {
                xr[b][0] = xa[bb][b][0].x; xr[b][1] = xa[bb][b][0].y; xr[b][2] = xa[bb][b][0].z; xr[b][3] = xa[bb][b][0].w;
                xr[b][4] = xa[bb][b][1].x; xr[b][5] = xa[bb][b][1].y; xr[b][6] = xa[bb][b][1].z; xr[b][7] = xa[bb][b][1].w;
            }
            float mean, scale;
            if (a.ln_stats) {
                // statistics hoisted into ln_stats_kernel: otherwise every (row group, slot pair) wave would redo them
                mean = a.ln_stats[2 * bi[bb]]; scale = a.ln_stats[2 * bi[bb] + 1];
                #pragma unroll
                for (int b = 0; b < NBLK; b++)
                    #pragma unroll
                    for (int e = 0; e < 8; e++) xr[b][e] = xr[b][e] - mean;
            } else {
                double p1[4] = {0.0, 0.0, 0.0, 0.0};
                #pragma unroll
                for (int b = 0; b < NBLK; b++) {
                    if ((b & 3) == rg) {
                        #pragma unroll
                        for (int e = 0; e < 8; e++) p1[e & 3] += (double) xr[b][e];
                    }
                }
                const double s1 = wave_sum((p1[0] + p1[1]) + (p1[2] + p1[3]));
                mean = (float) (s1 / (double) K);
                double p2[4] = {0.0, 0.0, 0.0, 0.0};
                #pragma unroll
                for (int b = 0; b < NBLK; b++) {
                    #pragma unroll
                    for (int e = 0; e < 8; e++) { const float v = xr[b][e] - mean; xr[b][e] = v; if ((b & 3) == rg) p2[e & 3] += (double) (v * v); }
                }
                const double s2 = wave_sum((p2[0] + p2[1]) + (p2[2] + p2[3]));
                const float var = (float) (s2 / (double) K);
                scale = 1.0f / sqrtf(var + 1e-5f);
            }
            #pragma unroll
            for (int b = 0; b < NBLK; b++) {
                const float gg[8] = {ga[b][0].x, ga[b][0].y, ga[b][0].z, ga[b][0].w, ga[b][1].x, ga[b][1].y, ga[b][1].z, ga[b][1].w};
                float bv[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                if constexpr (LNB) { bv[0] = ba[b][0].x; bv[1] = ba[b][0].y; bv[2] = ba[b][0].z; bv[3] = ba[b][0].w; bv[4] = ba[b][1].x; bv[5] = ba[b][1].y; bv[6] = ba[b][1].z; bv[7] = ba[b][1].w; }
                #pragma unroll
                for (int e = 0; e < 8; e++) {
                    float v = xr[b][e] * scale;
                    v = v * gg[e];
                    if constexpr (LNB) v = v + bv[e];
                    acc[bb] = fmaf((float) wv[b][e], (float) to_half(v), acc[bb]);
                }
            }
        }
    } else {
        constexpr int G = NBLK < 8 ? NBLK : 8;
        static_assert(NBLK % G == 0, "K/128 must be <= 8 or a multiple of 8");
        #pragma unroll
        for (int g = 0; g < NBLK / G; g++) {
            half8 wv[G], xv[BPW][G];
            #pragma unroll
            for (int i = 0; i < G; i++) {
                wv[i] = ld_half8(wrow + ((g * G + i) << 7));
                #pragma unroll
                for (int bb = 0; bb < BPW; bb++) xv[bb][i] = ld_half8(a.x_f16 + (size_t) bi[bb] * K + (c << 3) + ((g * G + i) << 7));
            }
            #pragma unroll
            for (int bb = 0; bb < BPW; bb++)
                #pragma unroll
                for (int i = 0; i < G; i++)
                    #pragma unroll
                    for (int e = 0; e < 8; e++) acc[bb] = fmaf((float) wv[i][e], (float) xv[bb][i][e], acc[bb]);
        }
    }
    #pragma unroll
    for (int bb = 0; bb < BPW; bb++) {
        const float r = wave_xor_add16(acc[bb]);
        if (live && c == 0 && b0 + bb < a.nbatch) linear_epilogue_pre(a, b0 + bb, m, r, pre[bb]);
    }
}

template <int NBLK>
static void launch_gemv_batch_n(hipStream_t s, const LinArgs & a) {
    constexpr int BPW = 2;
    dim3 grid((a.M + 3) / 4, (a.nbatch + BPW - 1) / BPW), block(64);
    if (a.x_f32) {
        if constexpr (NBLK <= 8) {
            if (a.ln_b) hipLaunchKernelGGL((gemv_batch_kernel<NBLK, true, true, BPW>), grid, block, 0, s, a);
            else        hipLaunchKernelGGL((gemv_batch_kernel<NBLK, true, false, BPW>), grid, block, 0, s, a);
        } else { kernel_fail("bark-hip: LayerNorm-fused GEMV supports n_embd <= 1024"); }
    } else {
        hipLaunchKernelGGL((gemv_batch_kernel<NBLK, false, false, BPW>), grid, block, 0, s, a);
    }
}

// Same operator for several input rows at once (N small, or as a cross-check of gemm_kernel):
// grid.y indexes the input row; no LayerNorm prologue.
template <int MAXB>
__global__ __launch_bounds__(256) void gemv_rows_kernel(const LinArgs a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 15, rg = lane >> 4;
    const int m = (blockIdx.x * 4 + wave) * 4 + rg;
    const int n = blockIdx.y;
    const int K = a.K, nblk = K >> 7;
    if (m >= a.M) return;
    const half_t * wrow = a.W + (size_t) m * K + (c << 3);
    const half_t * xrow = a.x_f16 + (size_t) n * K + (c << 3);
    float acc = 0.0f;
    #pragma unroll 4
    for (int b = 0; b < nblk; b++) {
        const half8 wv = ld_half8(wrow + (b << 7)), xv = ld_half8(xrow + (b << 7));
        #pragma unroll
        for (int e = 0; e < 8; e++) acc = fmaf((float) wv[e], (float) xv[e], acc);
    }
    acc = wave_xor_add16(acc);
    if (c == 0) linear_epilogue(a, n, m, acc, 0);
}

// ------------------------------------------------------------------------------------------------
// exact GEMM on the f32 matrix cores.  v_mfma_f32_32x32x2_f32 is bit-for-bit a k-ordered fmaf chain
// (cdna_hip_programming.md section 3), so one accumulator register == one chain of C1.
// Workgroup: 8 waves, output tile 64 (rows n) x 64 (cols m).  Wave w owns chains 2w and 2w+1 for the
// whole tile (2 x 2 MFMA tiles x 2 chains = 8 accumulators) and issues 4 MFMAs (k pairs) per tile, chain and 128-element K block.
// Operands are staged through LDS as f16 whole rows, fetched once per workgroup (round 3; before, every wave fetched 16-byte pieces of
// the same rows itself: 3.6 % slower, profiles/r03_gemm_lds.txt).  The 16 chains meet in LDS and are added in the C1 tree order.
// ------------------------------------------------------------------------------------------------
// Two tile shapes (round 6): 64 x 64, and 32 rows x 96 columns for the products whose 64 x 64 tiles fill the chip's 256 persistent workgroups badly - one
// fine window's QKV (576 tiles: 2.25 rounds), proj and MLP-proj (192 tiles: three quarters of the CUs) become 768 / 256 / 256 tiles.  Same chains, same
// order per accumulator: the bits do not depend on the shape (launch_linear picks by the share of busy workgroup rounds).
constexpr int GL_LD = 136;                                        // halfs per staged row (272 bytes: the 16-byte chunks a ds_read_b128 takes from 16 consecutive rows cover all 64 banks)
template <int TN, int TM> constexpr int gemm_lds_bytes() {       // the reduction tree of the epilogue (8 x TN x TM f32); the two staging buffers re-use it
    return 8 * TN * TM * 4 > 2 * (TN + TM) * GL_LD * 2 ? 8 * TN * TM * 4 : 2 * (TN + TM) * GL_LD * 2;
}
typedef float floatx4 __attribute__((ext_vector_type(4)));
DEVINL float half_of(const uint4 & u, int e) {                 // element e (compile-time) of eight packed f16 values, widened
    const unsigned word = (e >> 1) == 0 ? u.x : (e >> 1) == 1 ? u.y : (e >> 1) == 2 ? u.z : u.w;
    return (float) __builtin_bit_cast(half_t, (unsigned short) ((e & 1) ? (word >> 16) : word));
}
DEVINL uint4 ld_u4(const half_t * p) { return *reinterpret_cast<const uint4 *>(p); }
template <int TN, int TM>
__global__ __launch_bounds__(512) void gemm_kernel(const LinArgs a, const int ncol, const int nrow, const int pw) {
    extern __shared__ __attribute__((aligned(16))) float lds[];      // [8][TN][TM]
    constexpr int NI = TN / 32, NJ = TM / 32;                    // 32 x 32 MFMA tiles per chain along n and m
    constexpr int CX = TN * 16 / 512, CW = TM * 16 / 512;        // 16-byte chunks per thread and K block (x rows, weight rows)
    static_assert(TN % 32 == 0 && TM % 32 == 0 && CX >= 1 && CW >= 1 && CX * 512 == TN * 16 && CW * 512 == TM * 16, "tile shape");
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    // Persistent: the grid is one workgroup per CU (128 KB of LDS each); workgroup b walks the virtual ids b, b + grid, ... (same XCD: the grid is
    // a multiple of 8) through the XCD-aware tile order, and requests the first K block of its NEXT tile before the epilogue of the current one,
    // so that the ~2 us a first request takes overlap the reduction tree and the stores instead of standing in front of every tile's MFMAs.
    const int ntiles = ncol * nrow;
    int vb = blockIdx.x;
    if (vb >= ntiles) return;
    int n0, m0;
    const int K = a.K, nblk = K >> 7;
    floatx16 acc[2][NI][NJ];

    // Operand staging: the workgroup fetches the K block (128 elements = 256 bytes per row) of its 64 x rows and 64 weight rows ONCE, as
    // whole rows (a wave covers 4 rows x 256 contiguous bytes; gemm_kernel's waves fetch 16-byte pieces of those rows separately and depend
    // on finding each other's lines in L1), keeps them f16 in LDS (row stride 272 bytes: the 16-byte chunks a ds_read_b128 takes from 16
    // consecutive rows cover all 64 banks) and every wave reads the chunks of ITS two chains from there; conversions stay in the matrix waves.
    // Two LDS buffers, one barrier per K block: block b + 1 travels through registers (requested one block = ~4096 matrix-core cycles ahead)
    // and is written behind the MFMAs of block b.  The staging buffers (68 KB) are re-used by the reduction tree of the epilogue (128 KB).
    half_t * stg = reinterpret_cast<half_t *>(lds);              // [buffer][x rows TN | w rows TM][GL_LD]
    constexpr int CMAX = CX > CW ? CX : CW;
    const half_t * xsrc[CX]; const half_t * wsrc[CW]; int sdst[CMAX];
    #pragma unroll
    for (int i = 0; i < CMAX; i++) { const int c = threadIdx.x + 512 * i; sdst[i] = (c >> 4) * GL_LD + (c & 15) * 8; }
    auto setup_tile = [&](int id) {
        int trow, tcol;
        panel_tile(xcd_rank(id, ntiles), nrow, ncol, pw, trow, tcol);      // XCD-aware tile order (device_utils.h)
        n0 = trow * TN; m0 = tcol * TM;
        #pragma unroll
        for (int i = 0; i < CX; i++) { const int c = threadIdx.x + 512 * i, row = c >> 4, ch = c & 15; xsrc[i] = a.x_f16 + (size_t) min(n0 + row, a.N - 1) * K + ch * 8; }
        #pragma unroll
        for (int i = 0; i < CW; i++) { const int c = threadIdx.x + 512 * i, row = c >> 4, ch = c & 15; wsrc[i] = a.W + (size_t) min(m0 + row, a.M - 1) * K + ch * 8; }
    };
    uint4 rx[CX], rw[CW];
#define GL_FETCH(B) { _Pragma("unroll") for (int i = 0; i < CX; i++) rx[i] = ld_u4(xsrc[i] + ((B) << 7));                    \
                      _Pragma("unroll") for (int i = 0; i < CW; i++) rw[i] = ld_u4(wsrc[i] + ((B) << 7)); }
#define GL_STAGE(BUF) { half_t * d_ = stg + (BUF) * (TN + TM) * GL_LD;                                                      \
        _Pragma("unroll") for (int i = 0; i < CX; i++) *reinterpret_cast<uint4 *>(d_ + sdst[i]) = rx[i];                    \
        _Pragma("unroll") for (int i = 0; i < CW; i++) *reinterpret_cast<uint4 *>(d_ + TN * GL_LD + sdst[i]) = rw[i]; }
    half8 xa0[2][NI], wb0[2][NJ];
#define GEMM_LOAD_BLOCK(XA, WB, BUF)                                                                     \
    _Pragma("unroll") for (int s = 0; s < 2; s++) {                                                      \
        const half_t * src_ = stg + (BUF) * (TN + TM) * GL_LD + ((2 * w + s) << 3);                       \
        _Pragma("unroll") for (int t = 0; t < NI; t++) XA[s][t] = *reinterpret_cast<const half8 *>(src_ + (t * 32 + l31) * GL_LD);               \
        _Pragma("unroll") for (int t = 0; t < NJ; t++) WB[s][t] = *reinterpret_cast<const half8 *>(src_ + TN * GL_LD + (t * 32 + l31) * GL_LD);  \
    }
#define GEMM_MFMA_BLOCK(XA, WB)                                                                          \
    _Pragma("unroll") for (int s = 0; s < 2; s++) {                                                      \
        uint4 xu[NI], wu[NJ];                                                                            \
        _Pragma("unroll") for (int t = 0; t < NI; t++) xu[t] = __builtin_bit_cast(uint4, XA[s][t]);      \
        _Pragma("unroll") for (int t = 0; t < NJ; t++) wu[t] = __builtin_bit_cast(uint4, WB[s][t]);      \
        _Pragma("unroll") for (int kp = 0; kp < 4; kp++) {                                               \
            float av[NI], bv[NJ];                                                                        \
            _Pragma("unroll") for (int t = 0; t < NI; t++) {                                             \
                const unsigned xr = kp == 0 ? xu[t].x : kp == 1 ? xu[t].y : kp == 2 ? xu[t].z : xu[t].w; \
                av[t] = (float) __builtin_bit_cast(half_t, (unsigned short) (xr >> sh16));               \
            }                                                                                            \
            _Pragma("unroll") for (int t = 0; t < NJ; t++) {                                             \
                const unsigned wr = kp == 0 ? wu[t].x : kp == 1 ? wu[t].y : kp == 2 ? wu[t].z : wu[t].w; \
                bv[t] = (float) __builtin_bit_cast(half_t, (unsigned short) (wr >> sh16));               \
            }                                                                                            \
            _Pragma("unroll") for (int i = 0; i < NI; i++)                                               \
                _Pragma("unroll") for (int j = 0; j < NJ; j++)                                           \
                    acc[s][i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[s][i][j], 0, 0, 0); \
        }                                                                                                \
    }
    // lanes 32-63 feed the odd element of each f16 pair (the MFMA's second k slot): one per-lane shift selects it
    const unsigned sh16 = half ? 16u : 0u;
    setup_tile(vb);
    GL_FETCH(0)
    while (true) {
    #pragma unroll
    for (int s = 0; s < 2; s++) for (int i = 0; i < NI; i++) for (int j = 0; j < NJ; j++)
        for (int r = 0; r < 16; r++) acc[s][i][j][r] = 0.0f;
    GL_STAGE(0)
    if (nblk > 1) GL_FETCH(1)
    __syncthreads();
    int b = 0;
    // branch-free steady state (a conditional request would make hipcc wait for vmcnt(0) at the staging stores), guarded last blocks behind it
    for (; b + 2 < nblk; b++) {
        GEMM_LOAD_BLOCK(xa0, wb0, b & 1)
        GEMM_MFMA_BLOCK(xa0, wb0)
        GL_STAGE((b + 1) & 1)
        GL_FETCH(b + 2)
        __syncthreads();
    }
    for (; b < nblk; b++) {
        GEMM_LOAD_BLOCK(xa0, wb0, b & 1)
        GEMM_MFMA_BLOCK(xa0, wb0)
        if (b + 1 < nblk) GL_STAGE((b + 1) & 1)
        __syncthreads();
    }
    const int cn0 = n0, cm0 = m0;                               // the tile being finished
    const int nvb = vb + (int) gridDim.x;
    const bool more = nvb < ntiles;
    if (more) { setup_tile(nvb); GL_FETCH(0) }                  // in flight during the epilogue
    // chain pair (2w, 2w+1) -> LDS; accumulator register r of lane l holds row (r&3)+8(r>>2)+4*half, col l31
    float * mine = lds + (size_t) w * (TN * TM);
    #pragma unroll
    for (int i = 0; i < NI; i++)
        #pragma unroll
        for (int j = 0; j < NJ; j++)
            #pragma unroll
            for (int r = 0; r < 16; r++) {
                const int row = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half, col = j * 32 + l31;
                mine[row * TM + col] = acc[0][i][j][r] + acc[1][i][j][r];
            }
    __syncthreads();
    // each thread finishes 2 x 4 adjacent outputs: every epilogue operand is fetched as one 16-byte access and all
    // of them are requested before the first is used
    constexpr int Q4 = TN * TM / 4, ER = (Q4 + 511) / 512, TM4 = TM / 4;      // float4 outputs of the tile, rounds of 512 threads over them
    float4 bias4[ER], res4[ER];
    int nn[ER], mm[ER];
    #pragma unroll
    for (int r = 0; r < ER; r++) {
        const int idx4 = threadIdx.x + 512 * r;
        nn[r] = cn0 + idx4 / TM4; mm[r] = cm0 + ((idx4 % TM4) << 2);
        const bool ok = idx4 < Q4 && nn[r] < a.N && mm[r] < a.M;
        bias4[r] = (a.bias && ok) ? *reinterpret_cast<const float4 *>(a.bias + mm[r]) : float4{0.f, 0.f, 0.f, 0.f};
        res4[r] = (a.epi == EPI_RESID && ok) ? *reinterpret_cast<const float4 *>(a.res + (size_t) nn[r] * a.M + mm[r]) : float4{0.f, 0.f, 0.f, 0.f};
    }
    #pragma unroll
    for (int r = 0; r < ER; r++) {
        const int idx4 = threadIdx.x + 512 * r;
        if (idx4 >= Q4) continue;
        float4 p[8];
        #pragma unroll
        for (int q = 0; q < 8; q++) p[q] = *reinterpret_cast<const float4 *>(lds + q * (TN * TM) + idx4 * 4);
        float v[4];
        v[0] = ((p[0].x + p[1].x) + (p[2].x + p[3].x)) + ((p[4].x + p[5].x) + (p[6].x + p[7].x));
        v[1] = ((p[0].y + p[1].y) + (p[2].y + p[3].y)) + ((p[4].y + p[5].y) + (p[6].y + p[7].y));
        v[2] = ((p[0].z + p[1].z) + (p[2].z + p[3].z)) + ((p[4].z + p[5].z) + (p[6].z + p[7].z));
        v[3] = ((p[0].w + p[1].w) + (p[2].w + p[3].w)) + ((p[4].w + p[5].w) + (p[6].w + p[7].w));
        const int n = nn[r], m = mm[r];
        if (n >= a.N || m >= a.M) continue;
        if (a.bias) { v[0] = v[0] + bias4[r].x; v[1] = v[1] + bias4[r].y; v[2] = v[2] + bias4[r].z; v[3] = v[3] + bias4[r].w; }
        switch (a.epi) {
            case EPI_QKV: {
                const int E = a.E;
                float4 o = {v[0], v[1], v[2], v[3]};
                if (m < E) { *reinterpret_cast<float4 *>(a.q + (size_t) n * E + m) = o; break; }
                const int zq = a.seq ? n / a.seq : 0;                   // several sequences back to back (fine windows / window prompts of a batch)
                int pos = a.pos0 + (a.st ? a.st->n_past : 0) + (n - zq * a.seq);
                size_t zoff = (size_t) zq * a.kv_slot_stride;
                if (a.seqtab) {
                    const SeqTab t = a.seqtab[zq];
                    if (n - zq * a.seq >= t.len) break;                 // padding row
                    pos = t.pos0 + (n - zq * a.seq); zoff = (size_t) t.slot * a.kv_slot_stride;
                }
                const int m2 = m < 2 * E ? m - E : m - 2 * E;
                const int h = m2 >> 6, d = m2 & 63;                     // d is a multiple of 4: one d-quad of the K layout
                if (m < 2 * E) *reinterpret_cast<float4 *>(a.kc + zoff + kc_index(h, d, pos, a.P)) = o;
                else         { *reinterpret_cast<float4 *>(a.vc + zoff + vc_index(h, d, pos, a.P)) = o; if (a.vt) *reinterpret_cast<float4 *>(a.vt + zoff + kc_index(h, d, pos, a.P)) = o; }
                break;
            }
            case EPI_RESID: {                                           // cur + inpL (bark.cpp:1352,1388)
                float4 o = {v[0] + res4[r].x, v[1] + res4[r].y, v[2] + res4[r].z, v[3] + res4[r].w};
                *reinterpret_cast<float4 *>(a.res + (size_t) n * a.M + m) = o;
                break;
            }
            case EPI_GELU: {
                half_t g[4];
                #pragma unroll
                for (int e = 0; e < 4; e++) g[e] = gelu_lut_apply(v[e], a.lut);
                *reinterpret_cast<uint2 *>(a.out_h + (size_t) n * a.M + m) = __builtin_bit_cast(uint2, g);
                break;
            }
            default: {
                float4 o = {v[0], v[1], v[2], v[3]};
                *reinterpret_cast<float4 *>(a.out + (size_t) n * a.ld_out + m) = o;
                break;
            }
        }
    }
    if (!more) break;
    vb = nvb;
    __syncthreads();                                            // every partial sum has been read: the staging buffers may be written again
    }
#undef GL_FETCH
#undef GL_STAGE
#undef GEMM_LOAD_BLOCK
#undef GEMM_MFMA_BLOCK
}

// ------------------------------------------------------------------------------------------------
// Lock-step decode product: y[slot][m] for up to 32 utterance slots at once, every weight read ONCE per step for all slots.
// v_mfma_f32_16x16x1_4b_f32 issues FOUR independent 16 x 16 x 1 blocks, each one fused multiply-add per element: block g of wave w is
// chain 4 w + g of C1, so a wave walks four chains of a (16 weight rows x 16 slots) tile element by element and a 256-thread
// workgroup (one wave per SIMD) holds all 16 chains.  Measured on the device (tools/probes/mfma_rate_probe.hip, mfma_dep_probe.hip;
// profiles/r03_mfma_*_probe.txt): this form issues every 32 cycles with ONE dependent accumulator (32 multiply-adds per clock and
// SIMD = the f32 matrix-core peak), D register 4 b + (i % 4) of lane 16 (i / 4) + j holds element (i, j) of block b; the 4 x 4 x 1
// form of round 2 needs 15.6 cycles per dependent issue for a quarter of the work and re-read the slots' rows for every 4 weight
// rows.  A VALU instruction between two dependent MFMAs costs ~43 cycles, so a block's 16 conversions come first and its 8 MFMAs
// back to back (47.6 cycles per MFMA for one wave alone, 32 when a second wave of the SIMD fills the conversion gaps).
//   lane (g, r): weight row m0 + r and slot s0 + r of chain 4 w + g: the 16-byte chunk at block * 256 + (4 w + g) * 16 bytes of either
//   row - a wave's load instruction covers 16 rows x 64 contiguous bytes, and nobody loads a byte twice.
//   Chains 4 w .. 4 w + 3 meet inside a lane ((d0 + d1) + (d2 + d3) over the block registers = C1's tree levels xor 1, xor 2), the four
//   waves through LDS (levels xor 4, xor 8); thread t then finishes output (row t % 16, slot t / 16) with the decode epilogues.
// Every load of a lane is requested before the first conversion (K <= 1024), or in two groups in flight (K up to 4096); the in-order
// vmcnt lets block b's MFMAs start when its two chunks have landed.  32-slot tiles (two accumulators sharing the weight conversions)
// measured slower at every shape: half the workgroups, and the 768-row products then fill 48 CUs.
// Per launch at 32 slots, bark-small (tools/time_slots.py, QKV / proj / FC / MLP proj): 5.7 / 3.7 / 5.5 / 8.9 us against 9.2 / 4.9 / 9.1 /
// 11.2 for the 4 x 4 x 1 kernel and 17.1 / 4.4 / 22.7 / 12.7 for the VALU GEMV per pair of slots (profiles/r03_slots16_times.txt).
// ------------------------------------------------------------------------------------------------
// LNF: the LayerNorm of the slot rows happens HERE instead of in a launch of its own (ln_rows_vec_kernel, 2.0 us + a kernel boundary, twice per
// layer): wave w normalises slots s0 + 4 w .. + 3 with ln_rows_vec_kernel's arithmetic (the row in registers, lane l holds elements 4 l + 256 i:
// the same partition of the double sums, the same operations per element - bit-equal), the f16 rows go to LDS (row stride K + 8 halfs: the 16
// rows a load instruction touches cover all 64 banks), and the weight rows are requested BEFORE that, so their way from HBM overlaps it.
template <int NBLK, bool LNF>
__global__ __launch_bounds__(256) void gemm_slots16_kernel(const half_t * __restrict__ W, const half_t * __restrict__ X, const int M, const int parity_rows,
                                                           const LinArgs a) {
    constexpr int K = NBLK * 128;
    constexpr int G = NBLK <= 8 ? NBLK : 8, NG = NBLK / G;       // blocks per load group
    static_assert(NBLK % G == 0, "K/128 must be <= 8 or a multiple of 8");
    static_assert(!LNF || (NG == 1 && NBLK % 2 == 0), "the fused LayerNorm needs the whole row in one load group and K % 256 == 0");
    __shared__ float red[4][16][17];
    constexpr int XS_LD = K + 8;
    __shared__ __attribute__((aligned(16))) half_t xs[LNF ? 16 * XS_LD : 8];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int g = lane >> 4, r = lane & 15;
    const int m0 = blockIdx.x * 16, s0 = blockIdx.y * 16;
    const int row_off = parity_rows ? parity_rows * (a.st->step & 1) : 0;
    const int coff = (4 * w + g) << 3;
    const half_t * wrow = W + (size_t) (row_off + min(m0 + r, M - 1)) * K + coff;
    const half_t * xrow = X + (size_t) min(s0 + r, a.nbatch - 1) * K + coff;
    floatx16 acc;
    #pragma unroll
    for (int i = 0; i < 16; i++) acc[i] = 0.0f;
    uint4 wa[G], xa[G], wb[NG > 1 ? G : 1], xb[NG > 1 ? G : 1];
#define SLOTS16_LOAD(WV, XV, GI)                                                                            \
    _Pragma("unroll") for (int i = 0; i < G; i++) {                                                          \
        WV[i] = ld_u4(wrow + (((GI) * G + i) << 7));                                                         \
        if constexpr (!LNF) XV[i] = ld_u4(xrow + (((GI) * G + i) << 7));                                     \
    }
#define SLOTS16_MFMA(WV, XV)                                                                                 \
    _Pragma("unroll") for (int i = 0; i < G; i++) {                                                          \
        float wf[8], xf[8];                                                                                  \
        _Pragma("unroll") for (int e = 0; e < 8; e++) { wf[e] = half_of(WV[i], e); xf[e] = half_of(XV[i], e); } \
        __builtin_amdgcn_sched_barrier(0);                                                                   \
        _Pragma("unroll") for (int e = 0; e < 8; e++) acc = __builtin_amdgcn_mfma_f32_16x16x1f32(wf[e], xf[e], acc, 0, 0, 0); \
        __builtin_amdgcn_sched_barrier(0);                                                                   \
    }
    SLOTS16_LOAD(wa, xa, 0)
    // the epilogue's operands (bias, residual, context length): thread t finishes row t % 16 of slot t / 16
    const int em = m0 + (threadIdx.x & 15), en = s0 + (threadIdx.x >> 4);
    const bool live = em < M && en < a.nbatch;
    const EpiPre pre = epilogue_prefetch(a, live ? en : 0, live ? em : 0, row_off);
    if constexpr (LNF) {
        constexpr int NV = K / 256;
        float4 gg[NV], bb[NV];
        #pragma unroll
        for (int i = 0; i < NV; i++) {
            gg[i] = *reinterpret_cast<const float4 *>(a.ln_g + 4 * lane + 256 * i);
            bb[i] = a.ln_b ? *reinterpret_cast<const float4 *>(a.ln_b + 4 * lane + 256 * i) : float4{0.f, 0.f, 0.f, 0.f};
        }
        float4 v[4][NV];
        #pragma unroll
        for (int q = 0; q < 4; q++) {
            const float * xr = a.x_f32 + (size_t) min(s0 + 4 * w + q, a.nbatch - 1) * K + 4 * lane;
            #pragma unroll
            for (int i = 0; i < NV; i++) v[q][i] = *reinterpret_cast<const float4 *>(xr + 256 * i);
        }
        #pragma unroll
        for (int q = 0; q < 4; q++) {
            double s1 = 0.0;
            #pragma unroll
            for (int i = 0; i < NV; i++) { s1 += (double) v[q][i].x; s1 += (double) v[q][i].y; s1 += (double) v[q][i].z; s1 += (double) v[q][i].w; }
            s1 = wave_sum(s1);
            const float mean = (float) (s1 / (double) K);
            double s2 = 0.0;
            #pragma unroll
            for (int i = 0; i < NV; i++) {
                float4 & u = v[q][i];
                u.x = u.x - mean; u.y = u.y - mean; u.z = u.z - mean; u.w = u.w - mean;
                s2 += (double) (u.x * u.x); s2 += (double) (u.y * u.y); s2 += (double) (u.z * u.z); s2 += (double) (u.w * u.w);
            }
            s2 = wave_sum(s2);
            const float var = (float) (s2 / (double) K);
            const float scale = 1.0f / sqrtf(var + 1e-5f);
            half_t * o = xs + (4 * w + q) * XS_LD + 4 * lane;
            #pragma unroll
            for (int i = 0; i < NV; i++) {
                float t[4] = {v[q][i].x * scale, v[q][i].y * scale, v[q][i].z * scale, v[q][i].w * scale};
                t[0] = t[0] * gg[i].x; t[1] = t[1] * gg[i].y; t[2] = t[2] * gg[i].z; t[3] = t[3] * gg[i].w;
                if (a.ln_b) { t[0] = t[0] + bb[i].x; t[1] = t[1] + bb[i].y; t[2] = t[2] + bb[i].z; t[3] = t[3] + bb[i].w; }
                half_t h[4];
                #pragma unroll
                for (int e = 0; e < 4; e++) h[e] = to_half(t[e]);
                *reinterpret_cast<uint2 *>(o + 256 * i) = __builtin_bit_cast(uint2, h);
            }
        }
        __syncthreads();
        #pragma unroll
        for (int i = 0; i < G; i++) xa[i] = *reinterpret_cast<const uint4 *>(xs + r * XS_LD + coff + (i << 7));
    }
    #pragma unroll
    for (int gi = 0; gi < NG; gi += 2) {
        if constexpr (NG > 1) { if (gi + 1 < NG) { SLOTS16_LOAD(wb, xb, gi + 1) } }
        __builtin_amdgcn_sched_barrier(0);                       // the next group's loads stay ahead of this group's MFMAs
        SLOTS16_MFMA(wa, xa)
        if constexpr (NG > 1) {
            if (gi + 1 < NG) {
                if (gi + 2 < NG) { SLOTS16_LOAD(wa, xa, gi + 2) }
                __builtin_amdgcn_sched_barrier(0);
                SLOTS16_MFMA(wb, xb)
            }
        }
    }
#undef SLOTS16_LOAD
#undef SLOTS16_MFMA
    // D register 4 b + v of lane (g, r): chain 4 w + b, weight row 4 g + v, slot r
    #pragma unroll
    for (int v = 0; v < 4; v++) red[w][r][4 * g + v] = (acc[v] + acc[4 + v]) + (acc[8 + v] + acc[12 + v]);
    __syncthreads();
    const int row = threadIdx.x & 15, sl = threadIdx.x >> 4;
    const float v = (red[0][sl][row] + red[1][sl][row]) + (red[2][sl][row] + red[3][sl][row]);
    if (live) linear_epilogue_pre(a, en, em, v, pre);
}
template <int NBLK>
static void launch_slots16_n(hipStream_t s, const LinArgs & a) {
    const dim3 grid((a.M + 15) / 16, (a.nbatch + 15) / 16);
    if constexpr (NBLK <= 8 && NBLK % 2 == 0) {
        if (a.ln_g) { hipLaunchKernelGGL((gemm_slots16_kernel<NBLK, true>), grid, dim3(256), 0, s, a.W, a.x_f16, a.M, a.parity_rows, a); return; }
    }
    if (a.ln_g || !a.x_f16) kernel_fail("bark-hip: the lock-step MFMA product fuses the LayerNorm only for n_embd %% 256 == 0, n_embd <= 1024");
    hipLaunchKernelGGL((gemm_slots16_kernel<NBLK, false>), grid, dim3(256), 0, s, a.W, a.x_f16, a.M, a.parity_rows, a);
}
bool linear_slots_fuses_ln(int K) { return K % 256 == 0 && K <= 1024; }

void launch_linear_slots(hipStream_t s, const LinArgs & a) {
    if (!a.batched || (!a.x_f16 && !(a.x_f32 && a.ln_g)) || a.wq.qs || (a.K & 127) != 0 || a.nbatch > 64) kernel_fail("bark-hip: the lock-step MFMA product takes f16 rows (or f32 rows + LayerNorm) of up to 64 slots and f16 weights");
    switch (a.K >> 7) {
        case 1:  launch_slots16_n<1>(s, a); break;
        case 2:  launch_slots16_n<2>(s, a); break;
        case 4:  launch_slots16_n<4>(s, a); break;
        case 6:  launch_slots16_n<6>(s, a); break;
        case 8:  launch_slots16_n<8>(s, a); break;
        case 16: launch_slots16_n<16>(s, a); break;
        case 24: launch_slots16_n<24>(s, a); break;
        case 32: launch_slots16_n<32>(s, a); break;
        default: kernel_fail("bark-hip: unsupported K=%d in the lock-step MFMA product", a.K);
    }
}

void launch_linear(hipStream_t s, const LinArgs & a) {
    if (a.wq.qs && a.wq.qt == QT_F32) { launch_linear_w32(s, a); return; }
    if (a.wq.qs) { launch_linear_q(s, a); return; }
    if ((a.K & 127) != 0 || a.K > 4096) { kernel_fail("bark-hip: unsupported K=%d in linear op", a.K); }
    if (a.N > 1 && ((a.M & 3) || (a.epi == EPI_LOGITS && (a.ld_out & 3)))) { kernel_fail("bark-hip: batched linear op needs M %% 4 == 0"); }
    const int nblk = a.K >> 7;
    if (a.batched) {
        switch (nblk) {
            case 1:  launch_gemv_batch_n<1>(s, a); break;
            case 2:  launch_gemv_batch_n<2>(s, a); break;
            case 4:  launch_gemv_batch_n<4>(s, a); break;
            case 6:  launch_gemv_batch_n<6>(s, a); break;
            case 8:  launch_gemv_batch_n<8>(s, a); break;
            case 16: launch_gemv_batch_n<16>(s, a); break;
            case 24: launch_gemv_batch_n<24>(s, a); break;
            case 32: launch_gemv_batch_n<32>(s, a); break;
            default: kernel_fail("bark-hip: unsupported K=%d in batched decode GEMV", a.K);
        }
        return;
    }
    if (a.N == 1) {
        switch (nblk) {           // n_embd in {128, 256, 512, 768, 1024} and 4x those
            case 1: launch_gemv_n<1>(s, a); break;
            case 2: launch_gemv_n<2>(s, a); break;
            case 4: launch_gemv_n<4>(s, a); break;
            case 6: launch_gemv_n<6>(s, a); break;
            case 8: launch_gemv_n<8>(s, a); break;
            case 16: launch_gemv_n<16>(s, a); break;
            case 24: launch_gemv_n<24>(s, a); break;
            case 32: launch_gemv_n<32>(s, a); break;
            default: kernel_fail("bark-hip: unsupported K=%d in decode GEMV", a.K);
        }
        return;
    }
    if (a.x_f32 || a.parity_rows) { kernel_fail("bark-hip: batched linear op needs f16 rows"); }
    if (a.fast == 1) { launch_linear_fast(s, a); return; }          // f16 matrix cores: the fine model's canonical order C1m, or the tolerance route
    if (crosscheck_mask() & 1) {
        dim3 grid((a.M + 15) / 16, a.N), block(256);
        hipLaunchKernelGGL((gemv_rows_kernel<32>), grid, block, 0, s, a);
        return;
    }
    if (a.epi == EPI_QKV16) kernel_fail("bark-hip: the f16 QKV epilogue exists on the tolerance route only");
    static const int n_cu = [] { hipDeviceProp_t p; int d = 0; return (hipGetDevice(&d) == hipSuccess && hipGetDeviceProperties(&p, d) == hipSuccess) ? p.multiProcessorCount : 256; }();
    const int gmax = std::max(8, n_cu / 8 * 8);                                         // persistent: one workgroup per CU
    // tile shape by the share of busy workgroup rounds: tiles / (rounds x workgroups); 64 x 64 unless 32 x 96 keeps clearly more of the chip busy
    auto tiles = [&](int tn, int tm) { return (long) ((a.M + tm - 1) / tm) * ((a.N + tn - 1) / tn); };
    auto busy = [&](long t) { return (double) t / (double) (((t + gmax - 1) / gmax) * gmax); };
    const bool wide = !(crosscheck_mask() & 2048) && a.M % 96 == 0 && busy(tiles(32, 96)) > busy(tiles(64, 64)) + 0.08;
    if (wide) {
        const int ncol = (a.M + 95) / 96, nrow = (a.N + 31) / 32;
        constexpr int lds_bytes = gemm_lds_bytes<32, 96>();
        hipLaunchKernelGGL((gemm_kernel<32, 96>), dim3(std::min(ncol * nrow, gmax)), dim3(512), lds_bytes, s, a, ncol, nrow, xcd_panel_width(ncol * nrow, ncol));
    } else {
        const int ncol = (a.M + 63) / 64, nrow = (a.N + 63) / 64;
        constexpr int lds_bytes = gemm_lds_bytes<64, 64>();
        hipLaunchKernelGGL((gemm_kernel<64, 64>), dim3(std::min(ncol * nrow, gmax)), dim3(512), lds_bytes, s, a, ncol, nrow, xcd_panel_width(ncol * nrow, ncol));
    }
}

void init_kernel_attributes() {
    (void) hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_kernel<64, 64>), hipFuncAttributeMaxDynamicSharedMemorySize, gemm_lds_bytes<64, 64>());
    (void) hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_kernel<32, 96>), hipFuncAttributeMaxDynamicSharedMemorySize, gemm_lds_bytes<32, 96>());
    init_attention_attributes();
    init_quant_attributes();
    init_fast_attributes();
}

}  // namespace barkhip
