// attention_kernels.hip - attention over the f32 KV cache: single-query decode kernels (orders C2 / C4e / C5 on VALU) and the
// multi-query prompt-pass / fine kernels (the same orders on v_mfma_f32_32x32x2_f32, one accumulator = one chain).
//   attn_ps_kernel              decode, one sequence, block_size 1024: finishes the partial scores the QKV kernel formed (the default)
//   attn_fused_kernel           decode, any block size, one workgroup per (head, slot): lock-step batches, f32 model files, and the
//                               cross-check route of attn_ps_kernel (BARK_HIP_CROSSCHECK bit 2)
//   attn_window_kernel<CAUSAL>  N queries with the scores in registers: whole windows of the fine model (false), prompt passes of the causal models
//                               at block_size 1024 (true, round 6)
//   attn_rows_kernel            N queries with the score tile in LDS: other block sizes, and the cross-check of the kernel above (bit 512)
// (the two-launch, value-sliced, wide and materialised variants of rounds 1-2 lost their A/Bs and are gone: git history)
#include "device_utils.h"

#include <algorithm>
#include <cfloat>
#include <cstdio>
#include <cstdlib>

namespace barkhip {

template <int G> DEVINL void load_k_group(float4 (&kv)[16], const float4 * kp, int P) {
    #pragma unroll
    for (int dq = 0; dq < 16; dq++) kv[dq] = kp[(size_t) dq * P + 256 * G];
}
// C2: four chains over the 16-d blocks (d-quads 4b .. 4b+3), combined as (c0 + c1) + (c2 + c3)
DEVINL float score_chain(const float4 (&kv)[16], const float * __restrict__ qh) {
    const float c0 = score_block_f4(kv, qh), c1 = score_block_f4(kv + 4, qh + 16), c2 = score_block_f4(kv + 8, qh + 32), c3 = score_block_f4(kv + 12, qh + 48);
    return ((c0 + c1) + (c2 + c3)) * 0.125f;                  // 1/sqrt(64), bark.cpp:1318
}
// ------------------------------------------------------------------------------------------------
// Fused decode attention: ONE launch, one 256-thread workgroup (4 waves, one per SIMD, up to 512
// registers each) per head.  With a 1.7 us launch floor a second launch costs more than pulling the
// head's K rows through the same CU, so scores, softmax and mix share a kernel:
//   scores : thread t owns keys t, t+256, t+512, t+768 (C2: one fmaf chain over d per key)
//   softmax: row max / double sum through LDS (two barriers)
//   mix    : wave w, 16-lane group g own chain c = 4w+g of C5; lane&15 owns 4 adjacent dims (float4 V
//            loads, so one instruction covers four keys); the 16 chains meet in LDS (tree order)
// K is loaded two 256-key groups ahead, every V row group of the live context is requested before the
// first arithmetic instruction.
// ------------------------------------------------------------------------------------------------
// VS = 2 (lock-step batches with at most half as many (head, slot) pairs as CUs): the 64 value dims of a pair are shared by two workgroups
// (blockIdx.z = half; both form all scores, lanes 0..7 of every 16-lane group mix 32 dims) - a CU pulls ~24 bytes / ns from the memory side,
// and a pair's K + V (328 KB at 640 keys) through ONE CU is what the kernel's time is made of: 246 KB per workgroup instead.  Same chains.
template <int VS>
__global__ __launch_bounds__(256) void attn_fused_kernel(const AttnDecodeArgs a) {
    __shared__ float es[1024];
    __shared__ float red_f[4];
    __shared__ double red_d[4];
    __shared__ float part[16][64];
    const int h = blockIdx.x, slot = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int P = a.P;
    const int E = a.H * 64;
    const float * __restrict__ qh = a.q + (size_t) slot * E + h * 64;            // wave-uniform: scalar loads
    const float * kc = a.kc + (size_t) slot * a.kv_slot_stride, * vc = a.vc + (size_t) slot * a.kv_slot_stride;
    const float4 * kp = reinterpret_cast<const float4 *>(kc) + (size_t) h * 16 * P + tid;
    const int chain = 4 * wave + (lane >> 4);
    const int vhalf = VS == 2 ? (int) blockIdx.z : 0;
    const bool mixer = VS == 1 || (lane & 15) < 8;             // VS = 2: lanes 8..15 of a group hold no values
    const int d4 = VS == 2 ? 8 * vhalf + (lane & 7) : (lane & 15);
    const float4 * vp = reinterpret_cast<const float4 *>(vc + (size_t) h * P * 64) + (size_t) chain * 16 + d4;   // row `chain`, dims 4*d4..
    float4 k0[16], k1[16];
    load_k_group<0>(k0, kp, P);                                // keys 0..255: always inside the cache
    const int ctx = a.st[slot].n_past + 1;
    if (ctx > 256) load_k_group<1>(k1, kp, P);
    float4 vv[64];
    #pragma unroll
    for (int g = 0; g < 4; g++) {
        if ((g == 0 || ctx > 256 * g) && mixer) {
            #pragma unroll
            for (int i = 0; i < 16; i++) vv[16 * g + i] = vp[(size_t) (16 * g + i) * 256];     // key chain + 16*(16g+i): 16 rows = 256 float4
        }
    }
    float s[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    { const float v = score_chain(k0, qh); if (tid < ctx) s[0] = v; }
    if (ctx > 512) load_k_group<2>(k0, kp, P);
    if (ctx > 256) { const float v = score_chain(k1, qh); if (tid + 256 < ctx) s[1] = v; }
    if (ctx > 768) load_k_group<3>(k1, kp, P);
    if (ctx > 512) { const float v = score_chain(k0, qh); if (tid + 512 < ctx) s[2] = v; }
    if (ctx > 768) { const float v = score_chain(k1, qh); if (tid + 768 < ctx) s[3] = v; }
    float mx = wave_max(fmaxf(fmaxf(s[0], s[1]), fmaxf(s[2], s[3])));
    if (lane == 0) red_f[wave] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red_f[0], red_f[1]), fmaxf(red_f[2], red_f[3]));
    double lsum = 0.0;
    #pragma unroll
    for (int i = 0; i < 4; i++) {
        const int j = tid + 256 * i;
        float e = 0.0f;
        if (j < ctx) { e = canon_expf(s[i] - mx); lsum += (double) e; }
        es[j] = e;
    }
    lsum = wave_sum(lsum);
    if (lane == 0) red_d[wave] = lsum;
    __syncthreads();
    const double sum = (red_d[0] + red_d[1]) + (red_d[2] + red_d[3]);
    const float inv = (float) (1.0 / sum);
    float4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
    #pragma unroll
    for (int g = 0; g < 4; g++) {
        if (g == 0 || ctx > 256 * g) {
            // probabilities fetched in one batch, masked terms dropped by a select on the result (see mix_chain below): the guarded
            // form cost one LDS round trip per key
            float pj[16];
            #pragma unroll
            for (int i = 0; i < 16; i++) pj[i] = es[chain + 16 * (16 * g + i)] * inv;      // p = e * (float)(1/sum), as ggml_soft_max scales in place
            #pragma unroll
            for (int i = 0; i < 16; i++) {
                const bool ok = chain + 16 * (16 * g + i) < ctx;
                const float4 v = vv[16 * g + i];
                const float tx = fmaf(v.x, pj[i], acc.x), ty = fmaf(v.y, pj[i], acc.y), tz = fmaf(v.z, pj[i], acc.z), tw = fmaf(v.w, pj[i], acc.w);
                acc.x = ok ? tx : acc.x; acc.y = ok ? ty : acc.y; acc.z = ok ? tz : acc.z; acc.w = ok ? tw : acc.w;
            }
        }
    }
    if (mixer) *reinterpret_cast<float4 *>(&part[chain][4 * d4]) = acc;
    __syncthreads();
    if (tid < 64 / VS) {
        const int d = VS == 2 ? 32 * vhalf + tid : tid;
        float p[16];
        #pragma unroll
        for (int c = 0; c < 16; c++) p[c] = part[c][d];
        #pragma unroll
        for (int st = 1; st < 16; st <<= 1)
            #pragma unroll
            for (int c = 0; c < 16; c += 2 * st) p[c] = p[c] + p[c + st];
        if (a.att32) a.att32[(size_t) slot * E + h * 64 + d] = p[0]; else a.att[(size_t) slot * E + h * 64 + d] = to_half(p[0]);
    }
}

// ------------------------------------------------------------------------------------------------
// Lock steps over few live slots (engine_batch.hip: kFewSlotsScores; measured and adopted in round 5): attn_fused_kernel for lock steps whose QKV product has already formed
// the partial scores of the cached keys (gemv_ln_slots_ps_kernel, kernels.hip: ps [slot][H][4][P]).  The workgroup of a (head, slot[, value
// half]) then reads 16 bytes of partial scores per key instead of the key's 256 bytes of K: 92 KB instead of 246 KB through its CU at 640
// keys with two value halves - the K stream was what the kernel's time is made of.  Scores: ((c0 + c1) + (c2 + c3)) * 0.125 over the four
// C2 blocks = score_chain's value bit for bit; the key the step appended itself is scored here from its K row in the cache (lanes 0..3 of
// wave 3 form the four blocks, as attn_ps_kernel does).  Softmax (C4), mix (C5) and the tree are attn_fused_kernel's, statement for statement.
// ------------------------------------------------------------------------------------------------
template <int VS>
__global__ __launch_bounds__(256) void attn_fused_ps_kernel(const AttnDecodeArgs a) {
    __shared__ float es[1024];
    __shared__ float red_f[4];
    __shared__ double red_d[4];
    __shared__ float part[16][64];
    __shared__ float snew;
    const int h = blockIdx.x, slot = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int P = a.P;
    const int E = a.H * 64;
    const float * kc = a.kc + (size_t) slot * a.kv_slot_stride, * vc = a.vc + (size_t) slot * a.kv_slot_stride;
    const float * __restrict__ psl = a.ps + ((size_t) slot * a.H + h) * 4 * P;          // block b of key j at b * P + j
    const int chain = 4 * wave + (lane >> 4);
    const int vhalf = VS == 2 ? (int) blockIdx.z : 0;
    const bool mixer = VS == 1 || (lane & 15) < 8;
    const int d4 = VS == 2 ? 8 * vhalf + (lane & 7) : (lane & 15);
    const float4 * vp = reinterpret_cast<const float4 *>(vc + (size_t) h * P * 64) + (size_t) chain * 16 + d4;
    // partial scores of keys tid + 256 g: the first group lies inside the buffer for every context (stale bits beyond the context are never used)
    float pq[4][4];
    #pragma unroll
    for (int b = 0; b < 4; b++) pq[0][b] = psl[b * P + tid];
    const int ctx = a.st[slot].n_past + 1;
    #pragma unroll
    for (int g = 1; g < 4; g++) {
        if (ctx > 256 * g) {
            #pragma unroll
            for (int b = 0; b < 4; b++) pq[g][b] = psl[b * P + tid + 256 * g];
        }
    }
    float4 vv[64];
    #pragma unroll
    for (int g = 0; g < 4; g++) {
        if ((g == 0 || ctx > 256 * g) && mixer) {
            #pragma unroll
            for (int i = 0; i < 16; i++) vv[16 * g + i] = vp[(size_t) (16 * g + i) * 256];
        }
    }
    // the key this step appended (position ctx - 1): its K row is in the cache, its partial scores are not
    if (wave == 3) {
        const int b = lane & 3;
        const float * __restrict__ qh = a.q + (size_t) slot * E + h * 64;
        const float4 * kp = reinterpret_cast<const float4 *>(kc) + ((size_t) h * 16 + 4 * b) * P + (ctx - 1);
        float4 kq[4];
        float qb[16];
        #pragma unroll
        for (int i = 0; i < 4; i++) kq[i] = kp[(size_t) i * P];
        #pragma unroll
        for (int i = 0; i < 16; i++) qb[i] = qh[16 * b + i];
        const float cb = score_block_f4(kq, qb);
        const float c0 = readlane_f32(cb, 0), c1 = readlane_f32(cb, 1), c2 = readlane_f32(cb, 2), c3 = readlane_f32(cb, 3);
        if (lane == 0) snew = ((c0 + c1) + (c2 + c3)) * 0.125f;  // 1/sqrt(64), bark.cpp:1318
    }
    float s[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    #pragma unroll
    for (int g = 0; g < 4; g++) {
        if (g == 0 || ctx > 256 * g) {
            const float v = ((pq[g][0] + pq[g][1]) + (pq[g][2] + pq[g][3])) * 0.125f;
            if (tid + 256 * g < ctx - 1) s[g] = v;
        }
    }
    __syncthreads();                                             // snew
    #pragma unroll
    for (int g = 0; g < 4; g++) if (tid + 256 * g == ctx - 1) s[g] = snew;
    float mx = wave_max(fmaxf(fmaxf(s[0], s[1]), fmaxf(s[2], s[3])));
    if (lane == 0) red_f[wave] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red_f[0], red_f[1]), fmaxf(red_f[2], red_f[3]));
    double lsum = 0.0;
    #pragma unroll
    for (int i = 0; i < 4; i++) {
        const int j = tid + 256 * i;
        float e = 0.0f;
        if (j < ctx) { e = canon_expf(s[i] - mx); lsum += (double) e; }
        es[j] = e;
    }
    lsum = wave_sum(lsum);
    if (lane == 0) red_d[wave] = lsum;
    __syncthreads();
    const double sum = (red_d[0] + red_d[1]) + (red_d[2] + red_d[3]);
    const float inv = (float) (1.0 / sum);
    float4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
    #pragma unroll
    for (int g = 0; g < 4; g++) {
        if (g == 0 || ctx > 256 * g) {
            float pj[16];
            #pragma unroll
            for (int i = 0; i < 16; i++) pj[i] = es[chain + 16 * (16 * g + i)] * inv;      // p = e * (float)(1/sum), as ggml_soft_max scales in place
            #pragma unroll
            for (int i = 0; i < 16; i++) {
                const bool ok = chain + 16 * (16 * g + i) < ctx;
                const float4 v = vv[16 * g + i];
                const float tx = fmaf(v.x, pj[i], acc.x), ty = fmaf(v.y, pj[i], acc.y), tz = fmaf(v.z, pj[i], acc.z), tw = fmaf(v.w, pj[i], acc.w);
                acc.x = ok ? tx : acc.x; acc.y = ok ? ty : acc.y; acc.z = ok ? tz : acc.z; acc.w = ok ? tw : acc.w;
            }
        }
    }
    if (mixer) *reinterpret_cast<float4 *>(&part[chain][4 * d4]) = acc;
    __syncthreads();
    if (tid < 64 / VS) {
        const int d = VS == 2 ? 32 * vhalf + tid : tid;
        float p[16];
        #pragma unroll
        for (int c = 0; c < 16; c++) p[c] = part[c][d];
        #pragma unroll
        for (int st = 1; st < 16; st <<= 1)
            #pragma unroll
            for (int c = 0; c < 16; c += 2 * st) p[c] = p[c] + p[c + st];
        if (a.att32) a.att32[(size_t) slot * E + h * 64 + d] = p[0]; else a.att[(size_t) slot * E + h * 64 + d] = to_half(p[0]);
    }
}

// ------------------------------------------------------------------------------------------------
// Decode attention on partial scores.  The QKV kernel of the step (gemv_ln_wg_kernel<PS>) has already formed, for every cached key,
// the four 16-d block sums of C2; this kernel adds them ((c0 + c1) + (c2 + c3)) * 0.125, scores the ONE key the step appended
// itself (from the fixed-address copy of its K row, so that those loads do not wait for the context length), and runs softmax + mix
// in the value-sliced layout: workgroup = head x 4 value dims (one d-quad), all 16 C5 chains local, no cross-workgroup traffic, no K row read.
//   thread = key: 16 bytes of partial scores + 16 bytes of V (one d-quad, read from the K-layout copy of V where the quads of
//   consecutive keys are contiguous: with the [key][64] layout a wave's load touched 64 cache lines for 1 KB), staged in LDS.  A CU pulls
//   only ~24 bytes/ns from the memory side, so the head's V is spread over 16 workgroups (10 KB each at 640 keys); the slices of a
//   head share an XCD and its L2, where all but the first find the partial scores.  (Holding the V rows in the registers of
//   the two mix waves cost those waves ~0.6 us just to ISSUE 48 loads - the memory pipeline's queue - in front of the first barrier.)
//   mix: 16 chains x DS dims walk their keys out of LDS (values and probabilities), tree over the chains, 8 outputs.
// leading parameters: what the first loads need (preloaded into SGPRs at wave launch, see gemv_kernel in kernels.hip)
// ------------------------------------------------------------------------------------------------
constexpr int SNEW_WAVE = 3;                                    // last wave of the first key group: owns keys in every context, never mixes
__global__ __launch_bounds__(1024) void attn_ps_kernel(const float * __restrict__ ps, const float * __restrict__ vt, const StepState * __restrict__ st,
                                                      const float * __restrict__ knew, const float * __restrict__ q, const int H, const int ng, const AttnDecodeArgs a) {
    TRACE_T0();
    TRACE_T1(H);
    constexpr int P = 1024, S = 16, DS = 4;                        // 16 value slices per head = the d-quads of the K-layout copy of V
    __shared__ __attribute__((aligned(16))) float vl[1024 * DS];     // V slice of the head: [key][4 dims]
    __shared__ float es[1024];
    __shared__ float red_f[16];
    __shared__ double red_d[16];
    __shared__ float part[16][DS];
    __shared__ float snew;
    const int g8 = blockIdx.x >> 3, x8 = blockIdx.x & 7;          // slices of a head get ids congruent mod 8 (same XCD), as attn_dslice_kernel
    const int h = x8 + 8 * (g8 / S), s = g8 % S;
    if (h >= H) return;                                             // padding workgroups of the last head group
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // every stream is a buffer load: uniform base + lane offset + scalar offset (device_utils.h).  The keys below 256 ng (the launch's
    // bound on the context, a preloaded argument) are requested at once; should the context be longer, the others follow once its
    // length has arrived from the device-resident state.
    const BufRsrc pr = buf_rsrc(ps + (size_t) h * 4 * P);                // [H][4][P]: block b of key j at b * P + j
    const BufRsrc vr = buf_rsrc(reinterpret_cast<const float4 *>(vt) + ((size_t) h * 16 + s) * P);      // quad s of every key: contiguous
    float4 p4 = {0.0f, 0.0f, 0.0f, 0.0f}, va = p4;
    float s_new = -INFINITY;
    if (wave < 4 * ng) {
        p4.x = buf_ld_f32(pr, (unsigned) tid * 4u, 0u); p4.y = buf_ld_f32(pr, (unsigned) tid * 4u, 4096u);
        p4.z = buf_ld_f32(pr, (unsigned) tid * 4u, 8192u); p4.w = buf_ld_f32(pr, (unsigned) tid * 4u, 12288u);
        va = buf_ld_f4(vr, (unsigned) tid * 16u, 0u);
    }
    if (wave == SNEW_WAVE) {
        // lanes 0..3 form the four C2 blocks of the newest key's score
        const int b = lane & 3;
        const BufRsrc kr = buf_rsrc(knew + h * 64), qr = buf_rsrc(q + h * 64);
        float4 kq[4], q4[4];
        #pragma unroll
        for (int i = 0; i < 4; i++) { kq[i] = buf_ld_f4(kr, (unsigned) b * 64u, (unsigned) i * 16u); q4[i] = buf_ld_f4(qr, (unsigned) b * 64u, (unsigned) i * 16u); }
        const float qb[16] = {q4[0].x, q4[0].y, q4[0].z, q4[0].w, q4[1].x, q4[1].y, q4[1].z, q4[1].w,
                              q4[2].x, q4[2].y, q4[2].z, q4[2].w, q4[3].x, q4[3].y, q4[3].z, q4[3].w};
        const float cb = score_block_f4(kq, qb);
        const float c0 = readlane_f32(cb, 0), c1 = readlane_f32(cb, 1), c2 = readlane_f32(cb, 2), c3 = readlane_f32(cb, 3);
        s_new = ((c0 + c1) + (c2 + c3)) * 0.125f;               // 1/sqrt(64), bark.cpp:1318
    }
    const int ctx = st->n_past + 1;
    if (wave >= 4 * ng && wave * 64 < ctx) {
        p4.x = buf_ld_f32(pr, (unsigned) tid * 4u, 0u); p4.y = buf_ld_f32(pr, (unsigned) tid * 4u, 4096u);
        p4.z = buf_ld_f32(pr, (unsigned) tid * 4u, 8192u); p4.w = buf_ld_f32(pr, (unsigned) tid * 4u, 12288u);
        va = buf_ld_f4(vr, (unsigned) tid * 16u, 0u);
    }
    float sc = -INFINITY;
    if (tid < ctx - 1) sc = ((p4.x + p4.y) + (p4.z + p4.w)) * 0.125f;
    *reinterpret_cast<float4 *>(vl + tid * DS) = va;
    [[maybe_unused]] unsigned long long stamp0 = 0, stamp1 = 0, stamp2 = 0;     // diagnostic build: score ready, softmax statistics ready, mix done
    TRACE_SET(stamp0, sc);
    float mx = fmaxf(wave_max(sc), s_new);
    if (lane == 0) { red_f[wave] = mx; if (wave == SNEW_WAVE) snew = s_new; }
    __syncthreads();
    mx = red_f[0];
    #pragma unroll
    for (int i = 1; i < 16; i++) mx = fmaxf(mx, red_f[i]);
    if (tid == ctx - 1) sc = snew;
    float e = 0.0f;
    if (tid < ctx) e = canon_expf(sc - mx);
    es[tid] = e;
    const double wsum = wave_sum((double) e);
    if (lane == 0) red_d[wave] = wsum;
    __syncthreads();
    TRACE_SET(stamp1, e);
    if (tid < 16 * DS) {                                        // whole waves: 64 or 128 mix threads
        double sum = 0.0;                                       // fixed order: ascending waves
        #pragma unroll
        for (int i = 0; i < 16; i++) sum += red_d[i];
        const float inv = (float) (1.0 / sum);
        // C5: chain = key mod 16 walks its keys in ascending order; keys below 16 * (ctx / 16) exist for every chain
        const int chain = tid / DS, d = tid % DS;
        const float * vp = vl + chain * DS + d, * ep = es + chain;
        float acc = 0.0f;
        const int n_full = ctx >> 4;
        int i = 0;
        for (; i + 8 <= n_full; i += 8) {
            float v[8], p[8];
            #pragma unroll
            for (int u = 0; u < 8; u++) { v[u] = vp[(i + u) * 16 * DS]; p[u] = ep[(i + u) * 16] * inv; }    // p = e * (float)(1/sum), as ggml_soft_max scales in place
            #pragma unroll
            for (int u = 0; u < 8; u++) acc = fmaf(v[u], p[u], acc);
        }
        for (; i < n_full; i++) acc = fmaf(vp[i * 16 * DS], ep[i * 16] * inv, acc);
        if (chain + 16 * n_full < ctx) acc = fmaf(vp[n_full * 16 * DS], ep[n_full * 16] * inv, acc);
        TRACE_SET(stamp2, acc);
        part[chain][d] = acc;
        __builtin_amdgcn_s_waitcnt(0xc07f);                      // lgkmcnt(0)
    }
    __syncthreads();
    if (tid < DS) {
        float p[16];
        #pragma unroll
        for (int c = 0; c < 16; c++) p[c] = part[c][tid];
        #pragma unroll
        for (int st2 = 1; st2 < 16; st2 <<= 1)
            #pragma unroll
            for (int c = 0; c < 16; c += 2 * st2) p[c] = p[c] + p[c + st2];
        const int o = h * 64 + DS * s + tid;
        if (a.att32) a.att32[o] = p[0]; else a.att[o] = to_half(p[0]);
    }
#ifdef BARK_TRACE
    trace_emit(a.tr, _tr0, _tr1, stamp2, trace_clock(), stamp0, stamp1);
#endif
}

// ------------------------------------------------------------------------------------------------
// Lock-step batches: the decode attention of all slots as TWO launches.  attn_fused_kernel is one workgroup per (head, slot) that pulls the
// head's whole K and V (328 KB at 640 keys) through ONE CU - and a CU pulls ~24 bytes/ns: 13.4 us per launch at 8 slots with 96 of 256 CUs
// busy, and at 32 slots 384 such workgroups (400+ registers: one per CU) are two rounds, the second half empty.  Split at the only place
// the orders C2 / C4 / C5 allow: the scores of different keys are independent (C2), the mix of different value dims is independent (C5).
//   attn_slots_scores_kernel   workgroup = (256-key group, head, slot), thread = key: 16 coalesced float4 of K, one C2 score -> sc [slot][head][P]
//   attn_slots_mix_kernel<DS>  workgroup = (value slice of 64 / DS dims, head, slot): every thread takes its share of the scores (max, e =
//                              canon_expf(s - max), double sum: C4, formed redundantly by the DS slices of a head), then lane
//                              (chain c, 4 dims) walks keys c, c + 16, ... with fmaf (C5) and the 16 chains meet in LDS in tree order.
// One more kernel boundary (~1.5 us) and 1.5 MB of scores, for 5 x as many workgroups of a quarter of the bytes each.  Used when there are
// more (head, slot) pairs than CUs (launch_attn_decode); with fewer pairs the single launch is faster: its time is the latency chain
// K -> scores -> exp -> mix of ONE workgroup, not the CU's bandwidth (13.5 against 15.9 us at 8 slots).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void attn_slots_scores_kernel(const AttnDecodeArgs a, float * __restrict__ sc) {
    const int g = blockIdx.x, h = blockIdx.y, slot = blockIdx.z, tid = threadIdx.x;
    const int ctx = a.st[slot].n_past + 1;
    if (g * 256 >= ctx) return;
    const int P = a.P, E = a.H * 64;
    const float * __restrict__ qh = a.q + (size_t) slot * E + h * 64;            // wave-uniform: scalar loads
    const int j = min(g * 256 + tid, ctx - 1);
    const float4 * kp = reinterpret_cast<const float4 *>(a.kc + (size_t) slot * a.kv_slot_stride) + (size_t) h * 16 * P + j;
    float4 kv[16];
    #pragma unroll
    for (int dq = 0; dq < 16; dq++) kv[dq] = kp[(size_t) dq * P];
    const float v = score_chain(kv, qh);
    if (g * 256 + tid < ctx) sc[((size_t) slot * a.H + h) * P + g * 256 + tid] = v;
}

template <int DS>
__global__ __launch_bounds__(256) void attn_slots_mix_kernel(const AttnDecodeArgs a, const float * __restrict__ sc) {
    constexpr int NT = 256, NW = NT / 64, DW = 64 / DS;             // DW value dims per workgroup; the first 16 x DW / 4 threads mix (chain, float4 lane)
    __shared__ float es[1024];
    __shared__ float red_f[NW];
    __shared__ double red_d[NW];
    __shared__ float part[16][DW];
    const int ds = blockIdx.x, h = blockIdx.y, slot = blockIdx.z, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int P = a.P, E = a.H * 64;
    const int ctx = a.st[slot].n_past + 1;
    const float * __restrict__ srow = sc + ((size_t) slot * a.H + h) * P;
    constexpr int L4 = DW / 4;                                      // float4 lanes per chain
    const bool mixer = tid < 16 * L4;
    const int chain = mixer ? tid / L4 : 0, d4 = tid % L4;
    const float4 * vp = reinterpret_cast<const float4 *>(a.vc + (size_t) slot * a.kv_slot_stride + (size_t) h * P * 64) + (size_t) chain * 16 + ds * L4 + d4;
    // the value rows of the first 256 keys are requested before the softmax (every context holds them or the loads hit allocated cache rows)
    float4 v0[16];
    if (mixer) {
        #pragma unroll
        for (int i = 0; i < 16; i++) v0[i] = vp[(size_t) i * 256];
    }
    constexpr int KPT = 1024 / NT;                                  // keys per thread in the softmax phase
    float s[KPT];
    float mx = -INFINITY;
    #pragma unroll
    for (int i = 0; i < KPT; i++) { const int j = tid + NT * i; s[i] = j < ctx ? srow[j] : -INFINITY; mx = fmaxf(mx, s[i]); }
    mx = wave_max(mx);
    if (lane == 0) red_f[wave] = mx;
    __syncthreads();
    mx = red_f[0];
    #pragma unroll
    for (int w = 1; w < NW; w++) mx = fmaxf(mx, red_f[w]);
    double lsum = 0.0;
    #pragma unroll
    for (int i = 0; i < KPT; i++) {
        const int j = tid + NT * i;
        float e = 0.0f;
        if (j < ctx) { e = canon_expf(s[i] - mx); lsum += (double) e; }
        es[j] = e;
    }
    lsum = wave_sum(lsum);
    if (lane == 0) red_d[wave] = lsum;
    __syncthreads();
    static_assert(NW == 4, "the waves' partial sums are combined as in attn_fused_kernel");
    const double sum = (red_d[0] + red_d[1]) + (red_d[2] + red_d[3]);
    const float inv = (float) (1.0 / sum);
    float4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
    auto mix16 = [&](const float4 (&vv)[16], int g) {               // keys chain + 16 (16 g + i), i < 16
        float pj[16];
        #pragma unroll
        for (int i = 0; i < 16; i++) pj[i] = es[chain + 16 * (16 * g + i)] * inv;          // p = e * (float)(1/sum)
        #pragma unroll
        for (int i = 0; i < 16; i++) {
            const bool ok = chain + 16 * (16 * g + i) < ctx;
            const float4 v = vv[i];
            const float tx = fmaf(v.x, pj[i], acc.x), ty = fmaf(v.y, pj[i], acc.y), tz = fmaf(v.z, pj[i], acc.z), tw = fmaf(v.w, pj[i], acc.w);
            acc.x = ok ? tx : acc.x; acc.y = ok ? ty : acc.y; acc.z = ok ? tz : acc.z; acc.w = ok ? tw : acc.w;
        }
    };
    if (mixer) {
        mix16(v0, 0);
        for (int g = 1; g * 256 < ctx; g++) {
            float4 vv[16];
            #pragma unroll
            for (int i = 0; i < 16; i++) vv[i] = vp[(size_t) (16 * g + i) * 256];
            mix16(vv, g);
        }
        *reinterpret_cast<float4 *>(&part[chain][4 * d4]) = acc;
    }
    __syncthreads();
    if (tid < DW) {
        float p[16];
        #pragma unroll
        for (int c = 0; c < 16; c++) p[c] = part[c][tid];
        #pragma unroll
        for (int st = 1; st < 16; st <<= 1)
            #pragma unroll
            for (int c = 0; c < 16; c += 2 * st) p[c] = p[c] + p[c + st];
        const size_t o = (size_t) slot * E + h * 64 + ds * DW + tid;
        if (a.att32) a.att32[o] = p[0]; else a.att[o] = to_half(p[0]);
    }
}

void launch_attn_decode(hipStream_t s, const AttnDecodeArgs & a) {
    if (a.ps && a.nbatch > 1) {
        // lock step at few slots: partial scores per slot from gemv_ln_slots_ps_kernel
        if (a.P != 1024 || a.knew || a.vt) kernel_fail("bark-hip: the lock-step partial-score attention takes block_size 1024 and the slots' own caches");
        static const int n_cu2 = [] { hipDeviceProp_t p; int d = 0; return (hipGetDevice(&d) == hipSuccess && hipGetDeviceProperties(&p, d) == hipSuccess) ? p.multiProcessorCount : 256; }();
        if (2 * a.H * a.nbatch <= n_cu2 && !(crosscheck_mask() & 32)) hipLaunchKernelGGL((attn_fused_ps_kernel<2>), dim3(a.H, a.nbatch, 2), dim3(256), 0, s, a);
        else                                                            hipLaunchKernelGGL((attn_fused_ps_kernel<1>), dim3(a.H, a.nbatch), dim3(256), 0, s, a);
        return;
    }
    if (a.ps) {
        if (a.nbatch != 1 || a.P != 1024) { kernel_fail("bark-hip: partial-score decode attention needs one sequence and block_size 1024"); }
        if (!a.knew) kernel_fail("bark-hip: partial-score decode attention needs the fixed-address copy of the appended K row");
        if (!a.vt) kernel_fail("bark-hip: partial-score decode attention needs the K-layout copy of V");
        hipLaunchKernelGGL(attn_ps_kernel, dim3(8 * 16 * ((a.H + 7) / 8)), dim3(1024), 0, s, a.ps, a.vt, a.st, a.knew, a.q, a.H, std::max(1, std::min(a.ng, 4)), a);
        return;
    }
    // lock-step batches with more (head, slot) pairs than CUs: scores and softmax + mix as two launches.  Measured per call at 8 / 16 / 32
    // slots (context 640; profiles/r03_attn_slots_split.txt): one workgroup per pair 13.5 / 17.5 / 33.1 us, the pair of launches 15.9 / 18.6 /
    // 29.9 us (44.3 -> 37.6 at context 900) - it pays only where attn_fused_kernel needs a second, half-empty round of workgroups
    static const int n_cu = [] { hipDeviceProp_t p; int d = 0; return (hipGetDevice(&d) == hipSuccess && hipGetDeviceProperties(&p, d) == hipSuccess) ? p.multiProcessorCount : 256; }();
    if (a.sc && a.P == 1024 && (a.H * a.nbatch > n_cu || (crosscheck_mask() & 64)) && !(crosscheck_mask() & 32)) {
        hipLaunchKernelGGL(attn_slots_scores_kernel, dim3(4, a.H, a.nbatch), dim3(256), 0, s, a, a.sc);
        hipLaunchKernelGGL((attn_slots_mix_kernel<2>), dim3(2, a.H, a.nbatch), dim3(256), 0, s, a, a.sc);
        return;
    }
    // few pairs: two workgroups per pair share its value dims (BARK_HIP_CROSSCHECK bit 32 keeps one workgroup per pair)
    if (a.nbatch > 1 && 2 * a.H * a.nbatch <= n_cu && !(crosscheck_mask() & 32)) { hipLaunchKernelGGL((attn_fused_kernel<2>), dim3(a.H, a.nbatch, 2), dim3(256), 0, s, a); return; }
    hipLaunchKernelGGL((attn_fused_kernel<1>), dim3(a.H, a.nbatch), dim3(256), 0, s, a);
}

// ------------------------------------------------------------------------------------------------
// Fused prefill / fine attention: one workgroup (8 waves) per (head, 32-query tile); the 32 x ctx score tile
// lives in LDS (row stride 1026 floats: conflict-free column reads), so scores never travel through HBM:
//   1. S = 0.125 * Q K^T   f32 MFMA, wave w takes key tiles w, w+8, ... (C2: one accumulator chain over d)
//   2. row softmax in LDS   (4 rows per wave; max, e = canon_expf(s - max), double sum)
//   3. O = P V              f32 MFMA, wave w owns chains 2w, 2w+1 of C5; p = e * inv formed at the operand read
//   4. the 16 chains meet in LDS (aliasing the score tile) in tree order
// ------------------------------------------------------------------------------------------------
constexpr int ATT_LD = 1026;
__global__ __launch_bounds__(512) void attn_rows_kernel(const AttnPrefillArgs a0) {
    extern __shared__ __attribute__((aligned(16))) float lds[];        // [32][ATT_LD] scores, then [8][32][64] partial sums
    __shared__ float rowinv[32];
    AttnPrefillArgs a = a0;
    // XCD-aware: the query tiles of one (sequence, head) hold consecutive ranks, so its K and V (512 KB at 1024 keys) stay in one L2
    const int QT = (a.N + 31) >> 5, rank = xcd_rank(blockIdx.x, QT * a.H * max(1, a.Z));
    const int h = (rank / QT) % a.H, i0 = (rank % QT) * 32;
    if (a.Z > 1 || a.seqtab) {                                          // sequence z: its own rows of q / att and its own cache
        const size_t z = rank / (QT * a.H);
        a.q += z * (size_t) a.N * a.ldq;
        if (a.att) a.att += z * (size_t) a.N * a.ld_att;
        if (a.att32) a.att32 += z * (size_t) a.N * a.ld_att;
        if (a.seqtab) {                                                 // ragged: its own length, its slot's cache, its own start position
            const SeqTab t = a.seqtab[z];
            a.kc += (size_t) t.slot * a.kv_seq_stride; a.vc += (size_t) t.slot * a.kv_seq_stride;
            a.N = t.len; a.n_past = t.pos0;
            if (i0 >= a.N) return;                                      // a tile of padding rows (the whole workgroup leaves)
        } else { a.kc += z * a.kv_seq_stride; a.vc += z * a.kv_seq_stride; }
    }
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    const int ctx = a.n_past + a.N;
    // keys this tile can see: causal rows of the tile end at n_past + i0 + 31
    const int jend = a.causal ? min(ctx, a.n_past + i0 + 32) : ctx;
    const int jend32 = (jend + 31) & ~31;
    // ---- 1. scores -------------------------------------------------------------------------------------
    {
        const int irow = min(i0 + l31, a.N - 1);
        const float4 * qp = reinterpret_cast<const float4 *>(a.q + (size_t) irow * a.ldq + h * 64);
        float4 qv[16];
        #pragma unroll
        for (int dq = 0; dq < 16; dq++) qv[dq] = qp[dq];
        auto load_k = [&](float4 (&kv)[16], int jt) {
            const int jrow = min(jt + l31, ctx - 1);
            const float4 * kp = reinterpret_cast<const float4 *>(a.kc) + (size_t) h * 16 * a.P + jrow;
            #pragma unroll
            for (int dq = 0; dq < 16; dq++) kv[dq] = kp[(size_t) dq * a.P];
        };
        auto score_tile = [&](const float4 (&kv)[16], int jt) {
            // C2: one accumulator (= one fmaf chain) per block of 16 d, combined as (c0 + c1) + (c2 + c3)
            floatx16 acc[4];
            #pragma unroll
            for (int b = 0; b < 4; b++) {
                #pragma unroll
                for (int r = 0; r < 16; r++) acc[b][r] = 0.0f;
                #pragma unroll
                for (int dq = 4 * b; dq < 4 * b + 4; dq++) {
                    acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(half ? qv[dq].y : qv[dq].x, half ? kv[dq].y : kv[dq].x, acc[b], 0, 0, 0);
                    acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(half ? qv[dq].w : qv[dq].z, half ? kv[dq].w : kv[dq].z, acc[b], 0, 0, 0);
                }
            }
            #pragma unroll
            for (int r = 0; r < 16; r++) {
                const int i = (r & 3) + 8 * (r >> 2) + 4 * half;       // accumulator row = query, column = key
                lds[i * ATT_LD + jt + l31] = ((acc[0][r] + acc[1][r]) + (acc[2][r] + acc[3][r])) * 0.125f;   // 1/sqrt(64), bark.cpp:1318
            }
        };
        // key tiles w, w+8, w+16, w+24 (32 keys each); the next tile's K rows are in flight during the MFMAs
        float4 ka[16], kb[16];
        int jt = w * 32;
        if (jt < jend) load_k(ka, jt);
        for (; jt < jend; jt += 512) {
            const bool more = jt + 256 < jend;
            if (more) load_k(kb, jt + 256);
            __builtin_amdgcn_sched_barrier(0);
            score_tile(ka, jt);
            __builtin_amdgcn_sched_barrier(0);
            if (more) {
                if (jt + 512 < jend) load_k(ka, jt + 512);
                __builtin_amdgcn_sched_barrier(0);
                score_tile(kb, jt + 256);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    __syncthreads();
    // ---- 2. softmax, rows 4w .. 4w+3 ---------------------------------------------------------------------
    #pragma unroll 1
    for (int rr = 0; rr < 4; rr++) {
        const int il = 4 * w + rr, i = i0 + il;
        float * s = lds + il * ATT_LD;
        const int valid = i < a.N ? (a.causal ? min(ctx, a.n_past + i + 1) : ctx) : 0;
        float mx = -INFINITY;
        for (int j = lane; j < valid; j += 64) mx = fmaxf(mx, s[j]);
        mx = wave_max(mx);
        // four independent exp evaluations per lane and trip
        double sum4[4] = {0.0, 0.0, 0.0, 0.0};
        for (int j = lane; j < valid; j += 256) {
            float e[4];
            #pragma unroll
            for (int u = 0; u < 4; u++) e[u] = j + 64 * u < valid ? canon_expf(s[j + 64 * u] - mx) : 0.0f;
            #pragma unroll
            for (int u = 0; u < 4; u++) if (j + 64 * u < valid) { s[j + 64 * u] = e[u]; sum4[u] += (double) e[u]; }
        }
        const double sum = wave_sum((sum4[0] + sum4[1]) + (sum4[2] + sum4[3]));
        for (int j = valid + lane; j < jend32; j += 64) s[j] = 0.0f;   // masked keys: p == 0
        if (lane == 0) rowinv[il] = valid ? (float) (1.0 / sum) : 0.0f;
    }
    __syncthreads();
    // ---- 3. mix ------------------------------------------------------------------------------------------
    const float inv = rowinv[l31];
    const float * prow = lds + l31 * ATT_LD;
    const float * vbase = a.vc + (size_t) h * a.P * 64;
    floatx16 acc[2][2];
    #pragma unroll
    for (int s = 0; s < 2; s++) for (int t = 0; t < 2; t++) for (int r = 0; r < 16; r++) acc[s][t][r] = 0.0f;
    // batches of 8 key blocks (256 keys): the V rows and probabilities of batch b+1 are requested before the 32 MFMAs
    // of batch b issue (static double buffer; per-lane key slot j = jb + 2w + 16*half, chains 2w and 2w+1)
    float va[8][2][2], vb[8][2][2];
    float2 ea[8], eb[8];
#define ATT_LOAD_BATCH(V, E, JB0)                                                                        \
    _Pragma("unroll") for (int u = 0; u < 8; u++) {                                                      \
        const int j = (JB0) + 32 * u + 2 * w + 16 * half;                                                \
        const int jc = min(j, jend32 - 2);                                                               \
        E[u] = *reinterpret_cast<const float2 *>(prow + jc);                                             \
        _Pragma("unroll") for (int s = 0; s < 2; s++) {                                                  \
            const int jr = min(j + s, ctx - 1);                                                          \
            _Pragma("unroll") for (int t = 0; t < 2; t++) V[u][s][t] = vbase[(size_t) jr * 64 + t * 32 + l31]; \
        }                                                                                                \
    }
#define ATT_MFMA_BATCH(V, E, JB0)                                                                        \
    _Pragma("unroll") for (int u = 0; u < 8; u++) {                                                      \
        const int j = (JB0) + 32 * u + 2 * w + 16 * half;                                                \
        _Pragma("unroll") for (int s = 0; s < 2; s++) {                                                  \
            const bool oks = j + s < jend;                                                               \
            const float pv = oks ? (s ? E[u].y : E[u].x) * inv : 0.0f;      /* p = e * (float)(1/sum) */ \
            _Pragma("unroll") for (int t = 0; t < 2; t++)                                                \
                acc[s][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(pv, oks ? V[u][s][t] : 0.0f, acc[s][t], 0, 0, 0); \
        }                                                                                                \
    }
    const int jstop = jend;
    if (jstop > 0) { ATT_LOAD_BATCH(va, ea, 0) }
    for (int jb = 0; jb < jstop; jb += 512) {
        const bool more = jb + 256 < jstop;
        if (more) { ATT_LOAD_BATCH(vb, eb, jb + 256) }
        __builtin_amdgcn_sched_barrier(0);
        ATT_MFMA_BATCH(va, ea, jb)
        __builtin_amdgcn_sched_barrier(0);
        if (more) {
            if (jb + 512 < jstop) { ATT_LOAD_BATCH(va, ea, jb + 512) }
            __builtin_amdgcn_sched_barrier(0);
            ATT_MFMA_BATCH(vb, eb, jb + 256)
            __builtin_amdgcn_sched_barrier(0);
        }
    }
#undef ATT_LOAD_BATCH
#undef ATT_MFMA_BATCH
    __syncthreads();                                             // every wave is done reading the score tile
    float * part = lds;                                          // [8][32][64]
    #pragma unroll
    for (int t = 0; t < 2; t++)
        #pragma unroll
        for (int r = 0; r < 16; r++) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
            part[(w * 32 + row) * 64 + t * 32 + l31] = acc[0][t][r] + acc[1][t][r];
        }
    __syncthreads();
    for (int idx = threadIdx.x; idx < 32 * 64; idx += 512) {
        const int row = idx >> 6, d = idx & 63;
        float p[8];
        #pragma unroll
        for (int q = 0; q < 8; q++) p[q] = part[(q * 32 + row) * 64 + d];
        const float v = ((p[0] + p[1]) + (p[2] + p[3])) + ((p[4] + p[5]) + (p[6] + p[7]));
        const int i = i0 + row;
        if (i < a.N) { if (a.att32) a.att32[(size_t) i * a.ld_att + h * 64 + d] = v; else a.att[(size_t) i * a.ld_att + h * 64 + d] = to_half(v); }
    }
}

// ------------------------------------------------------------------------------------------------
// Whole-window attention of the fine model (non-causal, 1024 queries x 1024 keys per window and head; orders C2 / C4 / C5 as everywhere):
// the scores never leave the registers.  One workgroup (8 waves) per (window, head, 32 queries); wave w OWNS the chains 2w and 2w + 1 of C5,
// i.e. the 128 keys j with j mod 16 in {2w, 2w + 1}, for scores AND mix:
//   1. S^T = K Q^T on the f32 matrix cores, the score tile TRANSPOSED (row = key, column = query): accumulator register r of lane
//      (half h, query q) of tile t holds key  2w + (r >> 3) + 16 (16 t + 2 (r & 7) + h)  - the K rows a lane supplies are permuted accordingly.
//      C2: one accumulator per block of 16 d, (c0 + c1) + (c2 + c3), * 0.125.
//   2. softmax in place (C4): a lane holds 64 scores of ITS query; maximum and double sum meet across the two halves by a lane swap and
//      across the waves through 2 x 1 KB of LDS; e = canon_expf(s - max), p = e * (float)(1 / sum).
//   3. O = P V: register r of the lane IS the A operand of the MFMA that adds keys (.., h = 0) and (.., h = 1) - two consecutive keys of
//      chain 2w + (r >> 3) - to that chain's accumulator: registers 8 b .. 8 b + 7 of tiles 0 .. 3 walk chain 2w + b in ascending key order.
//   4. chains 2w + (2w + 1) meet in the wave, the eight waves through LDS in C5's tree order.
// Against attn_rows_kernel (scores through a 128 KB LDS tile, one barrier per phase, each wave reading all probabilities of its chains back):
// no score traffic at all, 64 exponentials per lane with full instruction-level parallelism.  Same operations per element: same bits.
// ------------------------------------------------------------------------------------------------
#ifdef ATTNW_STAMPS      // diagnostic build (build_variant.sh attnw -DATTNW_STAMPS; tools/attnw_phases.py): in-kernel time line of attn_window_kernel
__device__ unsigned long long g_attnw_stamps[512 * 8 * 8];
DEVINL unsigned long long attnw_clock(float dep) { unsigned long long t; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) : "v"(dep) : "memory"); return t; }
__device__ unsigned long long g_attnw_cycles[512 * 8 * 8];      // the same points on the shader clock (s_memtime): cycles per phase -> the clock the CU really ran at
#define ATTNW_STAMP(i, dep) do { const unsigned long long t_ = attnw_clock(dep); const unsigned long long c_ = __builtin_readcyclecounter(); if ((threadIdx.x & 63) == 0 && blockIdx.x < 512) { g_attnw_stamps[((size_t) blockIdx.x * 8 + (threadIdx.x >> 6)) * 8 + (i)] = t_; g_attnw_cycles[((size_t) blockIdx.x * 8 + (threadIdx.x >> 6)) * 8 + (i)] = c_; } } while (0)
#else
#define ATTNW_STAMP(i, dep)
#endif
constexpr int FQ_LD = 68;                                            // floats per staged query row (64 + 4: a ds_read_b128 of 32 rows covers all banks)
// CAUSAL (round 6): the same register-resident structure for the prompt passes of the causal models (semantic prompt, coarse window prompts, the all-slots
// prompt pass of a lock-step job; any N <= 1024 rows, rows continue a cache at n_past, ragged sequences through SeqTab; block_size 1024): a query tile sees
// the keys below jend = n_past + i0 + 32, so only the first ceil(jend / 256) key tiles are requested and multiplied; inside them a score whose key lies behind
// the query's own position is -inf in front of the maximum (its exponential is +0), and the V operand of a key at or beyond jend is 0 (cache rows beyond the
// context hold whatever was there).  Orders C2 / C4e / C5 as in the whole-window form and in attn_rows_kernel, which stays as the route for other block sizes
// and as the cross-check (BARK_HIP_CROSSCHECK bit 512).
template <bool CAUSAL>
__global__ __launch_bounds__(512, 1) void attn_window_kernel(const AttnPrefillArgs a0) {
    __shared__ __attribute__((aligned(16))) float qs[32 * FQ_LD];   // the query tile
    __shared__ float red_m[8][32];
    __shared__ double red_s[8][32];
    __shared__ __attribute__((aligned(16))) float part[8 * 32 * 64];            // partial outputs of the waves (64 KB)
    AttnPrefillArgs a = a0;
    constexpr int S = 1024;
    const int QT = CAUSAL ? (a.N + 31) >> 5 : S / 32, total = QT * a.H * max(1, a.Z);
    int rank = xcd_rank(blockIdx.x, total);
    if (!CAUSAL && (total & 7) == 0) {
        // an XCD's run of consecutive ranks, dealt over up to four (window, head) units at a time: workgroups that are resident together then read the K / V of
        // several heads (4 x 512 KB: still inside the 4 MB L2) instead of all pulling one head's lines at once
        const int C = total >> 3, base = (rank / C) * C, j = rank - base;
        const int NS = (C % 4 == 0 && C / 4 >= QT) ? 4 : (C % 3 == 0 && C / 3 >= QT) ? 3 : (C % 2 == 0 && C / 2 >= QT) ? 2 : 1;
        rank = base + (j % NS) * (C / NS) + j / NS;
    }
    // CAUSAL: a head's query tiles in DESCENDING order - the last tile sees all keys, the first 32 of them, and workgroups start in rank order: the long ones
    // first, the short ones fill the tail of the launch (336 workgroups on 256 CUs at 887 rows: 54 us per layer in ascending order)
    const int hd = (rank / QT) % a.H, i0 = (CAUSAL ? QT - 1 - rank % QT : rank % QT) * 32;
    const int rot_b = __builtin_amdgcn_readfirstlane((rank % QT) & 3), rot_t = CAUSAL ? 0 : __builtin_amdgcn_readfirstlane(((rank % QT) >> 2) & 3);
    if constexpr (CAUSAL) {
        if (a.Z > 1 || a.seqtab) {                                      // sequence z: its own rows of q / att and its own cache (as attn_rows_kernel)
            const size_t z = rank / (QT * a.H);
            a.q += z * (size_t) a.N * a.ldq;
            if (a.att) a.att += z * (size_t) a.N * a.ld_att;
            if (a.att32) a.att32 += z * (size_t) a.N * a.ld_att;
            if (a.seqtab) {
                const SeqTab t = a.seqtab[z];
                a.kc += (size_t) t.slot * a.kv_seq_stride; a.vc += (size_t) t.slot * a.kv_seq_stride;
                a.N = t.len; a.n_past = t.pos0;
                if (i0 >= a.N) return;                                  // a tile of padding rows (the whole workgroup leaves)
            } else { a.kc += z * a.kv_seq_stride; a.vc += z * a.kv_seq_stride; }
        }
    } else {
        const size_t z = rank / (QT * a.H);
        a.q += z * (size_t) S * a.ldq;
        if (a.att) a.att += z * (size_t) S * a.ld_att;
        if (a.att32) a.att32 += z * (size_t) S * a.ld_att;
        a.kc += z * a.kv_seq_stride; a.vc += z * a.kv_seq_stride;
    }
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, half = lane >> 5, l31 = lane & 31;
    // CAUSAL: keys this tile can see, key tiles (of 256) it needs, and this lane's query: the number of keys it may attend to
    const int ctx = CAUSAL ? a.n_past + a.N : S;
    const int jend = CAUSAL ? min(ctx, a.n_past + i0 + 32) : S;
    const int nt = CAUSAL ? __builtin_amdgcn_readfirstlane((jend + 255) >> 8) : 4;
    const int valid_q = CAUSAL ? min(ctx, a.n_past + min(i0 + l31, a.N - 1) + 1) : S;
    ATTNW_STAMP(0, (float) tid);
    // query tile -> LDS (512 threads x one float4)
    {
        const int row = tid >> 4, c4 = tid & 15;
        const int irow = CAUSAL ? min(i0 + row, a.N - 1) : i0 + row;
        *reinterpret_cast<float4 *>(qs + row * FQ_LD + 4 * c4) = *reinterpret_cast<const float4 *>(a.q + (size_t) irow * a.ldq + hd * 64 + 4 * c4);
    }
    // the key whose K row this lane supplies to tile t (A operand: lane l31 = tile row rho = (r & 3) + 8 (r >> 2) + 4 h')
    const int hp = (l31 >> 2) & 1, rr = (l31 & 3) + 4 * (l31 >> 3);
    const int key_a0 = 2 * w + (rr >> 3) + 16 * (2 * (rr & 7) + hp);                 // + 256 t
    const float4 * kbase = reinterpret_cast<const float4 *>(a.kc) + (size_t) hd * 16 * a.P + key_a0;
    floatx16 sc[4];
    // ---- 1. scores ------------------------------------------------------------------------------------
    {
        // block n = 4 t + b (tile t, 16-d block b): four register sets, a block's K rows are requested three blocks (24 MFMAs of this wave,
        // twice that on the SIMD) before they are multiplied - one block ahead left the matrix cores waiting for memory
        float4 ks[4][4];
        // Every query tile of a head starts its walk through K at another (key tile, 16-d block): the 32 workgroups of a head run side by side on one XCD,
        // and in step they all asked its L2 for the same lines at the same moment (scores phase 19.2 -> 15.3 us when they do not, profiles/r06_attnw_phases_*).
        // Register set t holds key tile t ^ rot_t until the sets are put back in order below; block order b ^ rot_b only permutes which of c0 .. c3 is
        // formed when: (c0 + c1) + (c2 + c3) reads the same four values, and f32 addition commutes - the bits do not move.
        auto load_blk = [&](float4 (&kv)[4], int n) {
            #pragma unroll
            for (int i = 0; i < 4; i++) kv[i] = kbase[(size_t) (4 * ((n & 3) ^ rot_b) + i) * a.P + 256 * ((n >> 2) ^ rot_t)];
        };
        load_blk(ks[0], 0); load_blk(ks[1], 1); load_blk(ks[2], 2);
        __syncthreads();                                         // the query tile is in LDS (the first K rows are already on their way)
        ATTNW_STAMP(1, ks[0][0].x);
        #pragma unroll
        for (int t = 0; t < 4; t++) {
            if (CAUSAL && t >= nt) {                             // uniform: no key of this tile is visible to the query tile
                #pragma unroll
                for (int r = 0; r < 16; r++) sc[t][r] = -INFINITY;
                continue;
            }
            floatx16 acc[4];
            #pragma unroll
            for (int b = 0; b < 4; b++) {
                const int n = 4 * t + b;
                if (n + 3 < 16 && (!CAUSAL || n + 3 < 4 * nt)) load_blk(ks[(n + 3) & 3], n + 3);
                #pragma unroll
                for (int r = 0; r < 16; r++) acc[b][r] = 0.0f;
                #pragma unroll
                for (int i = 0; i < 4; i++) {
                    const float4 kv = ks[n & 3][i];
                    const float4 qv = *reinterpret_cast<const float4 *>(qs + l31 * FQ_LD + 4 * (4 * (b ^ rot_b) + i));
                    acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(half ? kv.y : kv.x, half ? qv.y : qv.x, acc[b], 0, 0, 0);
                    acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(half ? kv.w : kv.z, half ? qv.w : qv.z, acc[b], 0, 0, 0);
                }
            }
            #pragma unroll
            for (int r = 0; r < 16; r++) {
                const float v = ((acc[0][r] + acc[1][r]) + (acc[2][r] + acc[3][r])) * 0.125f;      // 1/sqrt(64), bark.cpp:1318
                // register r of lane (half, query l31) holds key 2w + (r >> 3) + 16 (16 t + 2 (r & 7) + half)
                sc[t][r] = (!CAUSAL || 2 * w + (r >> 3) + 16 * (16 * t + 2 * (r & 7) + half) < valid_q) ? v : -INFINITY;
            }
        }
    }
    // key tiles back in order: register set t holds key tile t from here on (softmax sums and the C5 chains of the mix walk the keys in ascending order)
    if (rot_t & 1) { floatx16 x0 = sc[0]; sc[0] = sc[1]; sc[1] = x0; floatx16 x1 = sc[2]; sc[2] = sc[3]; sc[3] = x1; }
    if (rot_t & 2) { floatx16 x0 = sc[0]; sc[0] = sc[2]; sc[2] = x0; floatx16 x1 = sc[1]; sc[1] = sc[3]; sc[3] = x1; }
    ATTNW_STAMP(2, sc[3][15]);
    // ---- 2. softmax of query l31 (this lane: 64 of its 1024 scores) -------------------------------------
    float mx = -INFINITY;
    #pragma unroll
    for (int t = 0; t < 4; t++)
        #pragma unroll
        for (int r = 0; r < 16; r++) mx = fmaxf(mx, sc[t][r]);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    if (half == 0) red_m[w][l31] = mx;
    __syncthreads();
    mx = red_m[0][l31];
    #pragma unroll
    for (int i = 1; i < 8; i++) mx = fmaxf(mx, red_m[i][l31]);
    ATTNW_STAMP(3, mx);
    double ls[4] = {0.0, 0.0, 0.0, 0.0};
    #pragma unroll
    for (int t = 0; t < 4; t++)
        #pragma unroll
        for (int r = 0; r < 16; r++) {
            float e = canon_expf(sc[t][r] - mx);
            if (CAUSAL) e = sc[t][r] == -INFINITY ? 0.0f : e;              // a masked key: p == 0 (said outright instead of through the routine's cut-off)
            sc[t][r] = e; ls[r & 3] += (double) e;
        }
    double lsum = (ls[0] + ls[1]) + (ls[2] + ls[3]);
    lsum += __shfl_xor(lsum, 32, 64);
    if (half == 0) red_s[w][l31] = lsum;
    __syncthreads();
    const double sum = ((red_s[0][l31] + red_s[1][l31]) + (red_s[2][l31] + red_s[3][l31])) + ((red_s[4][l31] + red_s[5][l31]) + (red_s[6][l31] + red_s[7][l31]));
    const float inv = (float) (1.0 / sum);
    ATTNW_STAMP(4, inv);
    #pragma unroll
    for (int t = 0; t < 4; t++)
        #pragma unroll
        for (int r = 0; r < 16; r++) sc[t][r] = sc[t][r] * inv;                      // p = e * (float)(1/sum), as ggml_soft_max scales in place
    // ---- 3. mix: chains 2w (b = 0) and 2w + 1 (b = 1), two 32-dim column tiles ---------------------------
    floatx16 o[2][2];
    #pragma unroll
    for (int b = 0; b < 2; b++) for (int c = 0; c < 2; c++) for (int r = 0; r < 16; r++) o[b][c][r] = 0.0f;
    // B operand: V[key of (t, r, this lane's half)][2 l31 + c] - column l31 of dim tile c is dim 2 l31 + c, so that ONE 8-byte load feeds both tiles
    const float2 * vlane = reinterpret_cast<const float2 *>(a.vc + (size_t) hd * a.P * 64 + (size_t) (2 * w + 16 * half) * 64) + l31;   // key = 2w + (r >> 3) + 16 (16 t + 2 (r & 7) + half)
    // groups of 8 registers = 16 consecutive keys of ONE chain (group g: tile g >> 1, chain 2w + (g & 1)); four register sets, a group's
    // values are requested three groups (48 MFMAs) ahead
    float2 vs[4][8];
    auto load_v = [&](float2 (&vv)[8], int g) {
        #pragma unroll
        for (int i = 0; i < 8; i++) vv[i] = vlane[(size_t) ((g & 1) + 16 * (16 * (g >> 1) + 2 * i)) * 32];
    };
    load_v(vs[0], 0); load_v(vs[1], 1); load_v(vs[2], 2);
    #pragma unroll
    for (int g = 0; g < 8; g++) {
        if (CAUSAL && g >= 2 * nt) break;                      // uniform: the tiles behind jend hold no visible key
        if (g + 3 < 8 && (!CAUSAL || g + 3 < 2 * nt)) load_v(vs[(g + 3) & 3], g + 3);
        #pragma unroll
        for (int i = 0; i < 8; i++) {
            float2 v = vs[g & 3][i];
            if (CAUSAL) {                                        // rows at or beyond jend were never written for this sequence: their p is 0, their bits anything
                const bool ok = 2 * w + (g & 1) + 16 * (16 * (g >> 1) + 2 * i + half) < jend;
                v.x = ok ? v.x : 0.0f; v.y = ok ? v.y : 0.0f;
            }
            o[g & 1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(sc[g >> 1][8 * (g & 1) + i], v.x, o[g & 1][0], 0, 0, 0);
            o[g & 1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(sc[g >> 1][8 * (g & 1) + i], v.y, o[g & 1][1], 0, 0, 0);
        }
    }
    ATTNW_STAMP(5, o[1][1][15] + o[0][0][15]);
    // ---- 4. the 16 chains meet: 2w + (2w + 1) here, the waves in LDS (tree levels xor 2, 4, 8) ------------
    #pragma unroll
    for (int c = 0; c < 2; c++)
        #pragma unroll
        for (int r = 0; r < 16; r++) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * half;         // accumulator row = query, column = dim
            part[(w * 32 + row) * 64 + 2 * l31 + c] = o[0][c][r] + o[1][c][r];
        }
    __syncthreads();
    ATTNW_STAMP(6, part[tid]);
    for (int idx = tid; idx < 32 * 64; idx += 512) {
        const int row = idx >> 6, d = idx & 63;
        float pp[8];
        #pragma unroll
        for (int q = 0; q < 8; q++) pp[q] = part[(q * 32 + row) * 64 + d];
        const float v = ((pp[0] + pp[1]) + (pp[2] + pp[3])) + ((pp[4] + pp[5]) + (pp[6] + pp[7]));
        if (CAUSAL && i0 + row >= a.N) continue;
        const size_t oi = (size_t) (i0 + row) * a.ld_att + hd * 64 + d;
        if (a.att32) a.att32[oi] = v; else a.att[oi] = to_half(v);
    }
    ATTNW_STAMP(7, part[tid ^ 1]);
}

#ifdef ATTNW_STAMPS
extern "C" __attribute__((visibility("default"))) int bark_hip_debug_attnw_stamps(unsigned long long * out, int n) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_attnw_stamps), sizeof(unsigned long long) * (size_t) std::min(n, 512 * 8 * 8)) == hipSuccess ? 0 : -1;
}
extern "C" __attribute__((visibility("default"))) int bark_hip_debug_attnw_cycles(unsigned long long * out, int n) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_attnw_cycles), sizeof(unsigned long long) * (size_t) std::min(n, 512 * 8 * 8)) == hipSuccess ? 0 : -1;
}
#endif

void launch_attn_prefill(hipStream_t s, const AttnPrefillArgs & a) {
    // whole windows of the fine model: the register-resident kernel (BARK_HIP_CROSSCHECK bit 9 (512) keeps attn_rows_kernel)
    if (!a.causal && a.N == 1024 && a.n_past == 0 && !a.seqtab && a.P >= 1024 && !(crosscheck_mask() & 512)) {
        hipLaunchKernelGGL((attn_window_kernel<false>), dim3(32 * a.H * std::max(1, a.Z)), dim3(512), 0, s, a);
        return;
    }
    // prompt passes of the causal models at block_size 1024: the same register-resident kernel with the causal mask (bit 512 keeps attn_rows_kernel here too)
    if (a.causal && a.P == 1024 && a.N >= 1 && a.n_past + a.N <= 1024 && !(crosscheck_mask() & 512)) {
        hipLaunchKernelGGL((attn_window_kernel<true>), dim3((a.N + 31) / 32 * a.H * std::max(1, a.Z)), dim3(512), 0, s, a);
        return;
    }
    hipLaunchKernelGGL(attn_rows_kernel, dim3((a.N + 31) / 32 * a.H * std::max(1, a.Z)), dim3(512), 32 * ATT_LD * sizeof(float), s, a);
}

void init_attention_attributes() {
    (void) hipFuncSetAttribute(reinterpret_cast<const void *>(attn_rows_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                               32 * ATT_LD * (int) sizeof(float));
}

}  // namespace barkhip
