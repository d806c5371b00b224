// misc_kernels.hip - embeddings, LayerNorm over rows, and the sampling kernels (greedy rule C8, multinomial picks aligned
// with std::discrete_distribution, per-row picks of the fine stage).
#include "device_utils.h"

#include <cfloat>
#include <cstdio>
#include <cstdlib>

namespace barkhip {

// ------------------------------------------------------------------------------------------------
// embeddings
// ------------------------------------------------------------------------------------------------
__global__ void embed_causal_kernel(const EmbedArgs a) {
    const int i = blockIdx.x;
    int tok, tok2 = -1, pos;
    if (a.st) { tok = a.st->cur_token; pos = min(a.st->n_past, a.P - 1); }
    else {
        pos = a.pos0 + i;
        if (a.merge) { if (i < 256) { tok = a.tokens[i]; tok2 = a.tokens[256 + i]; } else tok = a.tokens[512]; }
        else tok = a.tokens[i];
    }
    tok = min(max(tok, 0), a.n_in - 1);
    if (tok2 >= 0) tok2 = min(tok2, a.n_in - 1);
    const float * pe = a.wpe + (size_t) pos * a.E;
    float * out = a.x + (size_t) i * a.E;
    for (int e = threadIdx.x; e < a.E; e += blockDim.x) {
        float v = wte_elem(a.wte, a.wte_q, a.E, tok, e);
        if (tok2 >= 0) v = v + wte_elem(a.wte, a.wte_q, a.E, tok2, e);   // wte[text] + wte[history]  (bark.cpp:1237-1248)
        out[e] = v + pe[e];
    }
}
void launch_embed_causal(hipStream_t s, const EmbedArgs & a) {
    hipLaunchKernelGGL(embed_causal_kernel, dim3(a.n_rows), dim3(256), 0, s, a);
}

struct FineEmbedArgs { const half_t * wte[8]; QMat wte_q[8]; const float * wpe; int E, n_in; const int32_t * tok; int nn; float * x; int plane; };
__global__ void embed_fine_kernel(const FineEmbedArgs a) {
    const int i = blockIdx.x;
    float * out = a.x + (size_t) i * a.E;
    const float * pe = a.wpe + (size_t) (i & 1023) * a.E;
    for (int e = threadIdx.x; e < a.E; e += blockDim.x) {
        float v = 0.0f;                                     // ggml_set_zero(tok_emb), bark.cpp:1936-1937
        for (int cb = 0; cb <= a.nn; cb++) {
            int id = a.tok[(size_t) cb * a.plane + i];
            id = min(max(id, 0), a.n_in - 1);
            v = v + wte_elem(a.wte[cb], a.wte_q[cb], a.E, id, e);
        }
        out[e] = v + pe[e];
    }
}
void launch_embed_fine(hipStream_t s, const half_t * const wte[8], const QMat * wte_q, const float * wpe, int E, int n_in,
                       const int32_t * tokens, int nn, float * x, int n_rows, int plane) {
    FineEmbedArgs a; for (int i = 0; i < 8; i++) { a.wte[i] = wte[i]; a.wte_q[i] = wte_q[i]; }
    a.wpe = wpe; a.E = E; a.n_in = n_in; a.tok = tokens; a.nn = nn; a.x = x; a.plane = plane;
    hipLaunchKernelGGL(embed_fine_kernel, dim3(n_rows), dim3(256), 0, s, a);
}

// ------------------------------------------------------------------------------------------------
// LayerNorm over rows -> f16 (one wave per row)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ln_rows_kernel(const float * x, int N, int E, const float * g, const float * b, half_t * out) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= N) return;
    const float * xr = x + (size_t) row * E;
    double s1 = 0.0;
    for (int e = lane; e < E; e += 64) s1 += (double) xr[e];
    s1 = wave_sum(s1);
    const float mean = (float) (s1 / (double) E);
    double s2 = 0.0;
    for (int e = lane; e < E; e += 64) { const float v = xr[e] - mean; s2 += (double) (v * v); }
    s2 = wave_sum(s2);
    const float var = (float) (s2 / (double) E);
    const float scale = 1.0f / sqrtf(var + 1e-5f);
    half_t * o = out + (size_t) row * E;
    for (int e = lane; e < E; e += 64) {
        float v = (xr[e] - mean) * scale;
        v = v * g[e];
        if (b) v = v + b[e];
        o[e] = to_half(v);
    }
}
// LayerNorm statistics only (batched decode): stats[row] = {mean, 1/sqrt(var + eps)}, same arithmetic as ln_rows_kernel
__global__ __launch_bounds__(256) void ln_stats_kernel(const float * x, int N, int E, float * stats) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= N) return;
    const float * xr = x + (size_t) row * E;
    double s1 = 0.0;
    for (int e = lane; e < E; e += 64) s1 += (double) xr[e];
    s1 = wave_sum(s1);
    const float mean = (float) (s1 / (double) E);
    double s2 = 0.0;
    for (int e = lane; e < E; e += 64) { const float v = xr[e] - mean; s2 += (double) (v * v); }
    s2 = wave_sum(s2);
    const float var = (float) (s2 / (double) E);
    if (lane == 0) { stats[2 * row] = mean; stats[2 * row + 1] = 1.0f / sqrtf(var + 1e-5f); }
}
void launch_ln_stats(hipStream_t s, const float * x, int N, int E, float * stats) {
    hipLaunchKernelGGL(ln_stats_kernel, dim3((N + 3) / 4), dim3(256), 0, s, x, N, E, stats);
}
// the same LayerNorm without the f16 rounding of the result: input of products with f32 weights
__global__ __launch_bounds__(256) void ln_rows_f32_kernel(const float * x, int N, int E, const float * g, const float * b, float * out) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= N) return;
    const float * xr = x + (size_t) row * E;
    double s1 = 0.0;
    for (int e = lane; e < E; e += 64) s1 += (double) xr[e];
    s1 = wave_sum(s1);
    const float mean = (float) (s1 / (double) E);
    double s2 = 0.0;
    for (int e = lane; e < E; e += 64) { const float v = xr[e] - mean; s2 += (double) (v * v); }
    s2 = wave_sum(s2);
    const float var = (float) (s2 / (double) E);
    const float scale = 1.0f / sqrtf(var + 1e-5f);
    float * o = out + (size_t) row * E;
    for (int e = lane; e < E; e += 64) {
        float v = (xr[e] - mean) * scale;
        v = v * g[e];
        if (b) v = v + b[e];
        o[e] = v;
    }
}
void launch_ln_rows_f32(hipStream_t s, const float * x, int N, int E, const float * g, const float * b, float * out) {
    hipLaunchKernelGGL(ln_rows_f32_kernel, dim3((N + 3) / 4), dim3(256), 0, s, x, N, E, g, b, out);
}
// E = 256 NV: the row stays in registers (one 16-byte load per 256 elements and lane), statistics and output from the same copy -
// one trip to memory instead of three dependent ones.  Same operations per element as ln_rows_kernel; the double sums are formed over a
// different partition of the row (they are exact or off by 2^-53 relative either way, long before the rounding to float).
template <int NV>
__global__ __launch_bounds__(256) void ln_rows_vec_kernel(const float * __restrict__ x, int N, const float * __restrict__ g, const float * __restrict__ b,
                                                          half_t * __restrict__ out) {
    constexpr int E = NV * 256;
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= N) return;
    const float * xr = x + (size_t) row * E + 4 * lane;
    float4 v[NV], gg[NV], bb[NV];
    #pragma unroll
    for (int i = 0; i < NV; i++) v[i] = *reinterpret_cast<const float4 *>(xr + 256 * i);
    #pragma unroll
    for (int i = 0; i < NV; i++) {
        gg[i] = *reinterpret_cast<const float4 *>(g + 4 * lane + 256 * i);
        bb[i] = b ? *reinterpret_cast<const float4 *>(b + 4 * lane + 256 * i) : float4{0.f, 0.f, 0.f, 0.f};
    }
    double s1 = 0.0;
    #pragma unroll
    for (int i = 0; i < NV; i++) { s1 += (double) v[i].x; s1 += (double) v[i].y; s1 += (double) v[i].z; s1 += (double) v[i].w; }
    s1 = wave_sum(s1);
    const float mean = (float) (s1 / (double) E);
    double s2 = 0.0;
    #pragma unroll
    for (int i = 0; i < NV; i++) {
        v[i].x = v[i].x - mean; v[i].y = v[i].y - mean; v[i].z = v[i].z - mean; v[i].w = v[i].w - mean;
        s2 += (double) (v[i].x * v[i].x); s2 += (double) (v[i].y * v[i].y); s2 += (double) (v[i].z * v[i].z); s2 += (double) (v[i].w * v[i].w);
    }
    s2 = wave_sum(s2);
    const float var = (float) (s2 / (double) E);
    const float scale = 1.0f / sqrtf(var + 1e-5f);
    half_t * o = out + (size_t) row * E + 4 * lane;
    #pragma unroll
    for (int i = 0; i < NV; i++) {
        float r[4] = {v[i].x * scale, v[i].y * scale, v[i].z * scale, v[i].w * scale};
        r[0] = r[0] * gg[i].x; r[1] = r[1] * gg[i].y; r[2] = r[2] * gg[i].z; r[3] = r[3] * gg[i].w;
        if (b) { r[0] = r[0] + bb[i].x; r[1] = r[1] + bb[i].y; r[2] = r[2] + bb[i].z; r[3] = r[3] + bb[i].w; }
        half_t h[4];
        #pragma unroll
        for (int e = 0; e < 4; e++) h[e] = to_half(r[e]);
        *reinterpret_cast<uint2 *>(o + 256 * i) = __builtin_bit_cast(uint2, h);
    }
}
void launch_ln_rows(hipStream_t s, const float * x, int N, int E, const float * g, const float * b, half_t * out) {
    const dim3 grid((N + 3) / 4), block(256);
    if (E == 768)  { hipLaunchKernelGGL((ln_rows_vec_kernel<3>), grid, block, 0, s, x, N, g, b, out); return; }
    if (E == 1024) { hipLaunchKernelGGL((ln_rows_vec_kernel<4>), grid, block, 0, s, x, N, g, b, out); return; }
    if (E == 512)  { hipLaunchKernelGGL((ln_rows_vec_kernel<2>), grid, block, 0, s, x, N, g, b, out); return; }
    if (E == 256)  { hipLaunchKernelGGL((ln_rows_vec_kernel<1>), grid, block, 0, s, x, N, g, b, out); return; }
    hipLaunchKernelGGL(ln_rows_kernel, grid, block, 0, s, x, N, E, g, b, out);       // other widths (toy models): three trips over the row
}


// ------------------------------------------------------------------------------------------------
// greedy sampling (gpt_argmax_sample, bark.cpp:223-247): l /= 0.7; softmax; first index of the largest probability;
// eos_p = probability of the last logit (semantic stage: the stop rule compares it with min_eos_p, bark.cpp:1690).
//
// The reference's softmax is a SEQUENTIAL float sum over all logits of e_i = (float) exp((double)(l_i - max)) followed by one float
// division per element.  Evaluating that literally takes one CU ~5 us per token (10 048 double-precision exps: fp64 throughput of
// a single CU; the in-kernel time line showed it as the longest tail of the whole step) plus a 10 048-long dependent add chain.
// Fast path, provably the reference's decision:
//   * pick: p_i = RN(e_i / sum) is monotone in e_i, so the winner is the first index whose e_i is 1.0f (d_i = l_i - max >=
//     -2^-25), PROVIDED no other logit lies within kNearTie = 4e-7 of the maximum: any e_i <= 1 - 4e-7 is more than two float
//     ulps below 1 and stays below after the division by the common sum.
//   * stop rule: eos ~= e_last / S with S a tree sum of v_exp_f32 terms (relative error < 1e-5 against the exact sum, while the
//     reference's own sequential sum may be off by up to n * 2^-24 = 6e-4); the decision eos_p >= min_eos_p is taken from it only
//     when eos is further than kEosBand = 2e-3 (relative) from the threshold.
// Anything else - a near tie, or eos inside the band - takes the exact path: every e_i in double precision, the reference's
// sequential float sum by one thread out of LDS, p_i = e_i / sum, first strict maximum.  That path returns the reference's bits
// (token AND eos_p); SampleArgs::force_exact (BARK_HIP_EXACT_SAMPLING=1) makes it the only path, which is how it is tested.
// ------------------------------------------------------------------------------------------------
constexpr float kTieCut = -2.98023223876953125e-08f;     // -2^-25
constexpr float kNearTie = -4.0e-7f;
constexpr float kEosBand = 2.0e-3f;
constexpr double kPickMargin = 1.0e-6;

// the reference's multinomial pick, literally, by ONE thread: e[0..n) exponentials (float), u the sample's uniform draw
DEVINL int multinomial_exact(const float * e, int n, double u, float * p_last) {
    float fs = 0.0f;
    for (int i = 0; i < n; i++) fs += e[i];                    // sequential float sum (bark.cpp:191-195)
    double sum = 0.0;
    for (int i = 0; i < n; i++) sum += (double) (e[i] / fs);   // std::accumulate over the float probabilities
    double run = 0.0;
    int pick = n - 1;
    for (int i = 0; i < n; i++) {
        run += (double) (e[i] / fs) / sum;                     // normalise, partial_sum
        if ((i == n - 1 ? 1.0 : run) >= u) { pick = i; break; }
    }
    if (p_last) *p_last = e[n - 1] / fs;
    return pick;
}

// leading parameters: what the first loads need (preloaded into SGPRs at wave launch, see gemv_kernel in kernels.hip)
__global__ __launch_bounds__(1024) void sample_greedy_kernel(const float * __restrict__ logits0, StepState * st0, const int n, const int ld_logits, const SampleArgs a) {
    TRACE_T0();
    TRACE_T1(n);
    __shared__ float red_f[16];
    __shared__ int red_i[16];
    __shared__ int red_c[16];
    __shared__ float red_s[16];
    __shared__ float e_all[12288];                             // exact path only
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int slot = blockIdx.x;                               // sequence slot (batched decode); 0 otherwise
    if (a.slot_temp && a.slot_temp[slot] > 0.0f) return;       // a sampled slot of a mixed batch: sample_multinomial_kernel's
    const float min_eos_p = a.slot_min_eos_p ? a.slot_min_eos_p[slot] : a.min_eos_p;
    const float * logits = logits0 + (size_t) slot * ld_logits;
    StepState * st = st0 + slot;
    // everything the end of the kernel needs from memory is requested now: the stage state, and the position row of the next
    // step's embedding (its token row can only be fetched once the pick is known)
    // VOLATILE loads: thread 0 rewrites the state at the END of this kernel, and on the fast path no barrier separates that store from the other
    // waves' last use of `step` (the coarse stage's codebook parity in `tok`).  As plain loads the compiler is free to sink them below the barriers
    // in between - it did: the scalar load of st->step sat behind the second __syncthreads(), so a wave that fell microseconds behind wave 0
    // (other contexts' kernels competing for the SIMD) could read the step thread 0 had already advanced, take the other codebook's offset and
    // write 64 elements of the NEXT token's embedding from the wrong row: one wrong sample for one slot, about once per 10^6 lock steps under 8+
    // concurrent contexts - round 4's red GPU suite.  A volatile access is not moved across the barrier intrinsic.
#ifdef BARK_DIAG_PLAIN_STATE_LOADS                              // tools/r05_state_race_demo.sh: the kernel as it was up to round 4, to show the race
    const int np_next = st->n_past + a.n_past_add;
    const int step = st->step;
#else
    const int np_next = *reinterpret_cast<const volatile int32_t *>(&st->n_past) + a.n_past_add;
    const int step = *reinterpret_cast<const volatile int32_t *>(&st->step);
#endif
    StepState s0{};
    if (tid == 0) s0 = *st;
    constexpr int MAXV = 12;                                   // up to 12288 logits
    float sv[MAXV];
    #pragma unroll
    for (int k = 0; k < MAXV; k++) {
        const int i = tid + 1024 * k;
        sv[k] = i < n ? logits[i] : -INFINITY;
    }
    float last_logit = logits[n - 1];
    const bool embed = a.x != nullptr && tid < a.E;
    float pe_v = 0.0f;
    if (embed && np_next < a.P) pe_v = a.wpe[(size_t) np_next * a.E + tid];
    float mx = -INFINITY;
    if (!a.prescaled) {                                        // gpt_argmax_sample divides by 0.7 whatever the temperature (bark.cpp:226-228);
        #pragma unroll                                         // the decode step's LM head has already done it (LinArgs::out_div)
        for (int k = 0; k < MAXV; k++) sv[k] = sv[k] / 0.7f;
        last_logit = last_logit / 0.7f;
    }
    #pragma unroll
    for (int k = 0; k < MAXV; k++) mx = fmaxf(mx, sv[k]);
    mx = wave_max(mx);
    if (lane == 0) red_f[wave] = mx;
    __syncthreads();
    mx = red_f[0];
    #pragma unroll
    for (int i = 1; i < 16; i++) mx = fmaxf(mx, red_f[i]);
    TRACE_T2(mx);
    int best = INT32_MAX, close = 0;
    float sum = 0.0f;
    #pragma unroll
    for (int k = 0; k < MAXV; k++) {
        const float d = sv[k] - mx;                            // -inf beyond n: never best, never close, exp2 = 0
        const int i = tid + 1024 * k;
        if (d >= kTieCut && i < best) best = i;
        if (d >= kNearTie) close++;
        if (a.mode == 0) sum += __builtin_amdgcn_exp2f(d * 1.44269504088896340736f);      // v_exp_f32: 1 ulp, fast path only
    }
    best = wave_min_i32(best); close = wave_add_i32(close); sum = wave_add_f32(sum);
    if (lane == 0) { red_i[wave] = best; red_c[wave] = close; red_s[wave] = sum; }
    __syncthreads();
#ifdef BARK_DIAG_LAG_WAVES                                      // the same demo: every wave but the one that rewrites the state falls ~N x 4 us behind
    {   // one asm statement (no control flow for the compiler to schedule around): waves 1 .. 15 sleep BARK_DIAG_LAG_WAVES x ~4 us, then the scalar
        // data cache is invalidated - what the dispatch of any other kernel on a neighbouring CU does to it.  (Without the invalidation the sunk
        // s_load of st->step HITS the line the wave's earlier scalar loads of the state brought in and returns the old, i.e. right, value: the race
        // needs a lagging wave AND a concurrent dispatch in that window, which is why it took 8+ busy contexts and ~10^6 lock steps to show.)
        const int w_uniform = __builtin_amdgcn_readfirstlane(wave);
        int n_sleep = w_uniform ? BARK_DIAG_LAG_WAVES : 0;
        asm volatile("s_cmp_eq_u32 %0, 0\n\ts_cbranch_scc1 2f\n1:\n\ts_sleep 127\n\ts_sub_u32 %0, %0, 1\n\ts_cmp_lg_u32 %0, 0\n\ts_cbranch_scc1 1b\n\ts_dcache_inv\n\ts_waitcnt lgkmcnt(0)\n2:" : "+s"(n_sleep) :: "scc");     // (no "memory" clobber: it would pin the plain load in front of this statement)
    }
#endif
    // every thread finishes the reduction itself (same order, same bits): no broadcast, no further barrier on the fast path
    best = red_i[0]; close = red_c[0]; sum = red_s[0];
    #pragma unroll
    for (int i = 1; i < 16; i++) { best = min(best, red_i[i]); close += red_c[i]; sum += red_s[i]; }
    float eos_p = 0.0f;
    bool exact = a.force_exact || close > 1;
    if (a.mode == 0) {
        eos_p = (float) exp((double) (last_logit - mx)) / sum;
        if (fabsf(eos_p - min_eos_p) <= kEosBand * min_eos_p) exact = true;
    }
    if (exact) {                                               // uniform: the reference's arithmetic, literally
        #pragma unroll
        for (int k = 0; k < MAXV; k++) {
            const int i = tid + 1024 * k;
            if (i < a.n) e_all[i] = (float) exp((double) (sv[k] - mx));
        }
        __syncthreads();
        if (tid == 0) {
            float fs = 0.0f;
            for (int i = 0; i < a.n; i++) fs += e_all[i];      // float sum in index order (bark.cpp:191-195)
            red_s[0] = fs;
        }
        __syncthreads();
        const float fs = red_s[0];
        // first strict maximum of p_i = e_i / sum (bark.cpp:197-198, 239-244): largest p, then smallest index
        float pbest = -1.0f; int ibest = INT32_MAX;
        #pragma unroll
        for (int k = 0; k < MAXV; k++) {
            const int i = tid + 1024 * k;
            if (i < a.n) { const float p = e_all[i] / fs; if (p > pbest) { pbest = p; ibest = i; } }     // ascending i within the thread
        }
        for (int m = 1; m < 64; m <<= 1) {
            const float po = __shfl_xor(pbest, m, 64); const int io = __shfl_xor(ibest, m, 64);
            if (po > pbest || (po == pbest && io < ibest)) { pbest = po; ibest = io; }
        }
        __syncthreads();
        if (lane == 0) { red_f[wave] = pbest; red_i[wave] = ibest; }
        __syncthreads();
        pbest = red_f[0]; ibest = red_i[0];
        #pragma unroll
        for (int i = 1; i < 16; i++) if (red_f[i] > pbest || (red_f[i] == pbest && red_i[i] < ibest)) { pbest = red_f[i]; ibest = red_i[i]; }
        best = ibest;
        eos_p = e_all[a.n - 1] / fs;
    }
    int tok = best;
    if (a.mode != 0) { eos_p = 0.0f; tok += a.token_base + ((step & 1) ? 1024 : 0); }     // slice start (bark.cpp:1829-1841)
    // embedding of the sampled token for the next decode step (bark.cpp:1250-1259): x = wte[tok] + wpe[n_past]
    if (embed && np_next < a.P) {
        const int t = min(max(tok, 0), a.n_in - 1);
        a.x[(size_t) slot * a.E + tid] = wte_elem(a.wte, a.wte_q, a.E, t, tid) + pe_v;
    }
    // The state is rewritten LAST, behind a barrier that every wave reaches only after its last use of `step` / `np_next` (a load cannot be sunk below
    // its use, so every wave's reads of the state are complete when it arrives here): the ordering of round 5's fix (volatile loads, kept above because
    // they also keep the requests early) no longer rests on how the compiler treats a volatile access next to the barrier intrinsic (advisor, round 5).
#ifndef BARK_DIAG_PLAIN_STATE_LOADS
    __syncthreads();
#endif
    if (tid == 0) {
        if (a.mode == 0) {
            // eos_p = probability of the LAST logit (bark.cpp:217-218,233-234)
            if ((tok == a.eos_token || eos_p >= min_eos_p) && s0.eos_step == INT32_MAX) st->eos_step = step;
            if (a.eos_trace) a.eos_trace[(size_t) slot * a.out_stride + step] = eos_p;
        }
        if (exact) st->near_tie = s0.near_tie + 1;            // samples settled by the exact path
        a.out_tokens[(size_t) slot * a.out_stride + s0.n_out] = tok;
        st->n_out = s0.n_out + 1;
        st->cur_token = tok;
        st->step = step + 1;
        st->n_past = np_next;
        st->last_eos_p = eos_p;
    }
    TRACE_END(a.tr);
#ifdef BARK_TRACE
    __syncthreads();
    if (tid == 0 && a.tr.pos) *a.tr.pos += 1;             // last kernel of a step: the next replay logs into the next segment
#endif
}
// ------------------------------------------------------------------------------------------------
// multinomial sampling on the device (gpt_multinomial_sample, bark.cpp:201-221): l /= temp; softmax;
// std::discrete_distribution.  libstdc++'s distribution normalises the probabilities once more in double, takes the
// running sums (last one forced to 1.0) and returns lower_bound(sums, u) for ONE uniform double u in [0,1) drawn with
// std::generate_canonical<double, 53> - the host draws those u from the context's std::mt19937 in the order the
// reference would (one per sample) and uploads them, so a seed selects the same random stream as in the reference.
// Fast path: float sum of the exponentials by a tree, running sums per thread range + block scan.  Against the reference's sequential
// float sum / sequential double sums the cumulative probabilities move by at most ~1.3e-7 (two float roundings per term; the common
// factor of the sum cancels in libstdc++'s renormalisation), so the pick is the reference's whenever u is further than kPickMargin =
// 1e-6 from both ends of the picked bin.  Otherwise (about two picks in a million) the EXACT path runs: one thread walks the
// exponentials out of LDS - sequential float sum (bark.cpp:191-195), p_i = e_i / sum, libstdc++'s sequential accumulate / normalise /
// partial_sum in double, last running sum pinned to 1.0, lower_bound.  (Round 4 found the need: a fine-stage pick 1.7e-11 from a bin
// boundary differed from the host path.)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void sample_multinomial_kernel(const SampleArgs a) {
    __shared__ float red_f[16];
    __shared__ double red_d[16];
    __shared__ int red_i[16];
    __shared__ int red_r[16];
    __shared__ int next_tok, next_pos;
    __shared__ float e_all[12288];                             // exponentials in index order (exact path)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int slot = blockIdx.x;
    if (a.slot_temp && !(a.slot_temp[slot] > 0.0f)) return;    // a greedy slot of a mixed batch: sample_greedy_kernel's
    if (tid == 0) red_r[0] = 0;
    const float temp = a.slot_temp ? a.slot_temp[slot] : a.temp;
    const float min_eos_p = a.slot_min_eos_p ? a.slot_min_eos_p[slot] : a.min_eos_p;
    const float * logits = a.logits + (size_t) slot * a.ld_logits;
    StepState * st = a.st + slot;
    const int step = st->step;
    const double u = a.u[(size_t) slot * a.u_stride + step];
    constexpr int CH = 12;                                      // thread t owns the contiguous ids [t*CH, t*CH+CH): up to 12288 logits
    float pv[CH];
    float mx = -INFINITY;
    #pragma unroll
    for (int k = 0; k < CH; k++) {
        const int i = tid * CH + k;
        pv[k] = i < a.n ? logits[i] / temp : -INFINITY;
        mx = fmaxf(mx, pv[k]);
    }
    mx = wave_max(mx);
    if (lane == 0) red_f[wave] = mx;
    __syncthreads();
    mx = red_f[0];
    #pragma unroll
    for (int i = 1; i < 16; i++) mx = fmaxf(mx, red_f[i]);
    float fsum = 0.0f;
    #pragma unroll
    for (int k = 0; k < CH; k++) {
        pv[k] = tid * CH + k < a.n ? (float) exp((double) (pv[k] - mx)) : 0.0f; fsum += pv[k];
        if (tid * CH + k < a.n) e_all[tid * CH + k] = pv[k];
    }
    for (int m = 1; m < 64; m <<= 1) fsum += __shfl_xor(fsum, m, 64);
    __syncthreads();
    if (lane == 0) red_f[wave] = fsum;
    __syncthreads();
    fsum = 0.0f;
    #pragma unroll
    for (int i = 0; i < 16; i++) fsum += red_f[i];
    double dsum = 0.0;
    float eos_p = 0.0f;                                         // set in the thread that owns the last id
    #pragma unroll
    for (int k = 0; k < CH; k++) {
        pv[k] = pv[k] / fsum; dsum += (double) pv[k];           // softmax probabilities (float), bark.cpp:197-199
        if (tid * CH + k == a.n - 1) eos_p = pv[k];
    }
    double wtot = wave_sum(dsum);
    if (lane == 0) red_d[wave] = wtot;
    __syncthreads();
    double total = 0.0, wave_off = 0.0;
    #pragma unroll
    for (int i = 0; i < 16; i++) { if (i < wave) wave_off += red_d[i]; total += red_d[i]; }
    // exclusive prefix of the per-thread sums inside the wave
    double incl = dsum;
    for (int m = 1; m < 64; m <<= 1) { const double o = __shfl_up(incl, m, 64); if (lane >= m) incl += o; }
    double run = (wave_off + (incl - dsum)) / total;
    int pick = INT32_MAX, risky = 0;
    #pragma unroll
    for (int k = 0; k < CH; k++) {
        const int i = tid * CH + k;
        if (i < a.n) {
            const double prev = run;
            run += (double) pv[k] / total;
            const double cp = i == a.n - 1 ? 1.0 : run;                                    // libstdc++ pins the last running sum to 1.0
            if (cp >= u && i < pick) { pick = i; risky = (cp - u < kPickMargin || u - prev < kPickMargin) ? 1 : 0; }
        }
    }
    // the smallest index wins; its thread knows how far u is from the ends of its bin
    const int mine = pick;
    for (int m = 1; m < 64; m <<= 1) pick = min(pick, __shfl_xor(pick, m, 64));
    if (lane == 0) red_i[wave] = pick;
    if (tid == ((a.n - 1) / CH)) red_f[0] = eos_p;
    __syncthreads();
    pick = red_i[0];
    #pragma unroll
    for (int i = 1; i < 16; i++) pick = min(pick, red_i[i]);
    if (mine == pick && pick != INT32_MAX) red_r[0] = risky;                               // exactly one thread holds the winning index
    __syncthreads();
    if (tid == 0) {
        int tok = pick;
        float ep = red_f[0];                                    // probability of the LAST logit (bark.cpp:217-218)
        if (red_r[0] || pick == INT32_MAX) {                    // u next to a bin boundary: the reference's arithmetic, literally
            tok = multinomial_exact(e_all, a.n, u, &ep);
            st->near_tie += 1;
        }
        if (a.mode == 0) {
            if ((tok == a.eos_token || ep >= min_eos_p) && st->eos_step == INT32_MAX) st->eos_step = step;
            if (a.eos_trace) a.eos_trace[(size_t) slot * a.out_stride + step] = ep;
        } else {
            ep = 0.0f;
            tok += a.token_base + ((step & 1) ? 1024 : 0);
        }
        a.out_tokens[(size_t) slot * a.out_stride + st->n_out] = tok;
        st->n_out += 1;
        st->cur_token = tok;
        st->step = step + 1;
        const int np = st->n_past + a.n_past_add;
        st->n_past = np;
        st->last_eos_p = ep;
        next_tok = tok; next_pos = np;
    }
    __syncthreads();
    if (a.x && next_pos < a.P) {
        const int tok = min(max(next_tok, 0), a.n_in - 1);
        const float * pe = a.wpe + (size_t) next_pos * a.E;
        float * xo = a.x + (size_t) slot * a.E;
        for (int e = tid; e < a.E; e += 1024) xo[e] = wte_elem(a.wte, a.wte_q, a.E, tok, e) + pe[e];
    }
}

// fine stage: one wave per row, multinomial over the first n_cols logits of the row; u[row] is that sample's uniform draw
__global__ __launch_bounds__(256) void sample_rows_multinomial_kernel(const float * logits, int ld, int n_rows, int n_cols, float temp,
                                                                     const double * u, int32_t * out, int out_stride, StepState * st) {
    __shared__ float es[4][1024];                               // the rows' exponentials in index order (exact path)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row = blockIdx.x * 4 + wave;
    if (row >= n_rows) return;
    const float * l = logits + (size_t) row * ld;
    constexpr int CH = 16;                                      // lane owns ids [lane*16, lane*16+16): n_cols <= 1024
    float pv[CH];
    float mx = -INFINITY;
    #pragma unroll
    for (int k = 0; k < CH; k++) { const int i = lane * CH + k; pv[k] = i < n_cols ? l[i] / temp : -INFINITY; mx = fmaxf(mx, pv[k]); }
    mx = wave_max(mx);
    float fsum = 0.0f;
    #pragma unroll
    for (int k = 0; k < CH; k++) {
        pv[k] = lane * CH + k < n_cols ? (float) exp((double) (pv[k] - mx)) : 0.0f; fsum += pv[k];
        es[wave][lane * CH + k] = pv[k];
    }
    for (int m = 1; m < 64; m <<= 1) fsum += __shfl_xor(fsum, m, 64);
    double dsum = 0.0;
    #pragma unroll
    for (int k = 0; k < CH; k++) { pv[k] = pv[k] / fsum; dsum += (double) pv[k]; }
    const double total = wave_sum(dsum);
    double incl = dsum;
    for (int m = 1; m < 64; m <<= 1) { const double o = __shfl_up(incl, m, 64); if (lane >= m) incl += o; }
    double run = (incl - dsum) / total;
    const double uu = u[row];
    int pick = INT32_MAX, risky = 0;
    #pragma unroll
    for (int k = 0; k < CH; k++) {
        const int i = lane * CH + k;
        if (i < n_cols) {
            const double prev = run;
            run += (double) pv[k] / total;
            const double cp = i == n_cols - 1 ? 1.0 : run;
            if (cp >= uu && i < pick) { pick = i; risky = (cp - uu < kPickMargin || uu - prev < kPickMargin) ? 1 : 0; }
        }
    }
    const int mine = pick;
    for (int m = 1; m < 64; m <<= 1) pick = min(pick, __shfl_xor(pick, m, 64));
    // the lane that holds the winning index knows how far u is from the ends of its bin
    const bool exact = __builtin_amdgcn_ballot_w64(pick == INT32_MAX || (mine == pick && risky)) != 0;
    if (exact) {                                                // wave-uniform; the wave's LDS row was written by this wave only
        __builtin_amdgcn_s_waitcnt(0xc07f);                     // lgkmcnt(0): the ds_writes above have landed
        if (lane == 0) {
            pick = multinomial_exact(es[wave], n_cols, uu, nullptr);
            if (st) atomicAdd(&st->near_tie, 1);
        }
    }
    if (lane == 0) out[(size_t) row * out_stride] = pick;
}
void launch_sample_rows_multinomial(hipStream_t s, const float * logits, int ld, int n_rows, int n_cols, float temp, const double * u,
                                    int32_t * out, int out_stride, StepState * st) {
    if (n_cols > 1024) kernel_fail("bark-hip: the per-row multinomial pick takes at most 1024 columns");
    hipLaunchKernelGGL(sample_rows_multinomial_kernel, dim3((n_rows + 3) / 4), dim3(256), 0, s, logits, ld, n_rows, n_cols, temp, u, out, out_stride, st);
}

void launch_sample_greedy(hipStream_t s, const SampleArgs & a) {
    if (a.x && a.E > 1024) kernel_fail("bark-hip: the sampler writes the next token's embedding with one thread per element (n_embd <= 1024)");
    if (a.slot_temp) {
        if (a.kinds & 1) hipLaunchKernelGGL(sample_greedy_kernel, dim3(a.nbatch), dim3(1024), 0, s, a.logits, a.st, a.n, a.ld_logits, a);
        if (a.kinds & 2) hipLaunchKernelGGL(sample_multinomial_kernel, dim3(a.nbatch), dim3(1024), 0, s, a);
        return;
    }
    if (a.temp > 0.0f) hipLaunchKernelGGL(sample_multinomial_kernel, dim3(a.nbatch), dim3(1024), 0, s, a);
    else hipLaunchKernelGGL(sample_greedy_kernel, dim3(a.nbatch), dim3(1024), 0, s, a.logits, a.st, a.n, a.ld_logits, a);
}

// fine stage, greedy: per-row pick of gpt_argmax_sample (bark.cpp:223-247) over the first n_cols logits.  Fast path as in sample_greedy_kernel: the
// first index whose e_i is 1.0f wins unless another logit lies within kNearTie of the maximum; then (about one row in 10^6) the wave repeats the
// reference's arithmetic literally - e_i in double precision, sequential float sum, p_i = e_i / sum, first strict maximum.
__global__ __launch_bounds__(256) void argmax_rows_kernel(const float * logits, int ld, int n_rows, int n_cols, int32_t * out,
                                                         int out_stride, StepState * st, int force_exact) {
    __shared__ float es[4][1024];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row = blockIdx.x * 4 + wave;
    if (row >= n_rows) return;
    const float * l = logits + (size_t) row * ld;
    float mx = -INFINITY;
    for (int i = lane; i < n_cols; i += 64) mx = fmaxf(mx, l[i] / 0.7f);
    mx = wave_max(mx);
    int best = INT32_MAX, close = 0;
    for (int i = lane; i < n_cols; i += 64) {
        const float d = l[i] / 0.7f - mx;
        if (d >= kTieCut && i < best) best = i;
        if (d >= kNearTie) close++;
    }
    for (int m = 1; m < 64; m <<= 1) { best = min(best, __shfl_xor(best, m, 64)); close += __shfl_xor(close, m, 64); }
    if ((close > 1 || force_exact) && n_cols <= 1024) {        // wave-uniform
        for (int i = lane; i < n_cols; i += 64) es[wave][i] = (float) exp((double) (l[i] / 0.7f - mx));
        __builtin_amdgcn_s_waitcnt(0xc07f);                     // lgkmcnt(0): the wave's own LDS row
        if (lane == 0) {
            float fs = 0.0f;
            for (int i = 0; i < n_cols; i++) fs += es[wave][i];                             // float sum in index order (bark.cpp:191-195)
            float pbest = -1.0f; int ibest = 0;
            for (int i = 0; i < n_cols; i++) { const float p = es[wave][i] / fs; if (p > pbest) { pbest = p; ibest = i; } }      // first strict maximum
            best = ibest;
            if (st) atomicAdd(&st->near_tie, 1);
        }
    }
    if (lane == 0) out[(size_t) row * out_stride] = best;
}
void launch_argmax_rows(hipStream_t s, const float * logits, int ld, int n_rows, int n_cols, int32_t * out, int out_stride,
                        StepState * st) {
    static const int force_exact = getenv("BARK_HIP_EXACT_SAMPLING") ? atoi(getenv("BARK_HIP_EXACT_SAMPLING")) : 0;      // tests: every row through the exact path
    hipLaunchKernelGGL(argmax_rows_kernel, dim3((n_rows + 3) / 4), dim3(256), 0, s, logits, ld, n_rows, n_cols, out, out_stride, st, force_exact);
}

}  // namespace barkhip
