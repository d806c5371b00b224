// quant_kernels.hip - linear operators whose weights are not f16: ggml block formats (q4_0, q4_1, q5_0, q5_1, q8_0 files
// written by bark_model_quantize) and plain f32 (files converted without --use-f16).  Same canonical chains as kernels.hip.
#include "device_utils.h"

#include <algorithm>
#include <cfloat>
#include <cstdio>
#include <cstdlib>

namespace barkhip {

// ------------------------------------------------------------------------------------------------
// Quantised weights (ggml block formats q4_0 - BASELINE config 4 - q4_1, q5_0, q5_1, q8_0; quant_formats.h), computed as
// ggml's vec_dot_q*_q8_* restated by the oracle's C1q order: the activation row is quantised to q8 blocks of 32
// (d = amax / 127, q = roundf(x / d), d stored as f16, s = f16(d * sum q) for the formats with a minimum), every block
// product is an exact integer sum, scaled per format (block_term), block b belongs to chain b mod 16, chains are plain float
// adds in ascending block order and meet in the C1 tree.  Weight levels are widened to int8 in registers (unpack_raw) so that
// one code path - v_dot4_i32_i8 for decode, v_mfma_i32_32x32x32_i8 for rows - serves all five formats.
// ------------------------------------------------------------------------------------------------
template <int QT> struct QTraits {
    static constexpr bool has_m = QT == QT_Q4_1 || QT == QT_Q5_1;
    static constexpr bool has_h = QT == QT_Q5_0 || QT == QT_Q5_1;
    static constexpr bool wide = QT == QT_Q8_0;                // 32 bytes of levels per block instead of 16
};
template <int QT> struct RawBlock { uint4 qs, qs2; unsigned qh; half_t d, m; };
template <int QT> DEVINL RawBlock<QT> load_raw(const QMat & q, size_t idx) {
    RawBlock<QT> r;
    if constexpr (QTraits<QT>::wide) { const uint4 * p = reinterpret_cast<const uint4 *>(q.qs) + 2 * idx; r.qs = p[0]; r.qs2 = p[1]; }
    else r.qs = reinterpret_cast<const uint4 *>(q.qs)[idx];
    if constexpr (QTraits<QT>::has_h) r.qh = q.qh[idx];
    r.d = q.d[idx];
    if constexpr (QTraits<QT>::has_m) r.m = q.m[idx];
    return r;
}
// four bits b3 b2 b1 b0 -> bit 4 of bytes 3..0
DEVINL unsigned spread_fifth_bits(unsigned b) { return ((b * 0x00204081u) & 0x01010101u) << 4; }
// per-byte v - k for bytes v < 128, k < 128, without borrows between bytes: set bit 7, subtract, flip bit 7 back
DEVINL int bytes_minus(unsigned v, unsigned k4) { return (int) (((v | 0x80808080u) - k4) ^ 0x80808080u); }
// half == 0: elements 0..15 of the block, half == 1: elements 16..31, as four dwords of int8 levels
template <int QT> DEVINL void unpack_half(const RawBlock<QT> & r, int half, int (&o)[4]) {
    if constexpr (QTraits<QT>::wide) {
        const uint4 v = half ? r.qs2 : r.qs;
        o[0] = (int) v.x; o[1] = (int) v.y; o[2] = (int) v.z; o[3] = (int) v.w;
    } else {
        const unsigned w[4] = {r.qs.x, r.qs.y, r.qs.z, r.qs.w};
        #pragma unroll
        for (int i = 0; i < 4; i++) {
            unsigned v = (w[i] >> (4 * half)) & 0x0F0F0F0Fu;
            if constexpr (QTraits<QT>::has_h) v |= spread_fifth_bits((r.qh >> (16 * half + 4 * i)) & 0xFu);
            if constexpr (QT == QT_Q4_0) o[i] = bytes_minus(v, 0x08080808u);
            else if constexpr (QT == QT_Q5_0) o[i] = bytes_minus(v, 0x10101010u);
            else o[i] = (int) v;
        }
    }
}
template <int QT> DEVINL void unpack_raw(const RawBlock<QT> & r, int (&o)[8]) {
    int lo[4], hi[4];
    unpack_half<QT>(r, 0, lo); unpack_half<QT>(r, 1, hi);
    #pragma unroll
    for (int i = 0; i < 4; i++) { o[i] = lo[i]; o[4 + i] = hi[i]; }
}
// ggml's per-block scaling (oracle: dot_q_q8)
template <int QT> DEVINL float block_term(int sumi, float dw, float mw, float dx, float sx) {
    if constexpr (QT == QT_Q4_0) return ((float) sumi * dw) * dx;
    else {
        const float dd = dw * dx;
        float t = dd * (float) sumi;
        if constexpr (QTraits<QT>::has_m) { const float ms = mw * sx; t = t + ms; }
        return t;
    }
}
DEVINL int dot_q4_q8(const int (&w)[8], const int (&q)[8]) {
    int s = 0;
    #pragma unroll
    for (int i = 0; i < 8; i++) s = __builtin_amdgcn_sdot4(w[i], q[i], s, false);
    return s;
}
// NV f32 values -> NV/4 dwords of int8 levels q = roundf(v * id); returns sum q
template <int NV> DEVINL int quantize_levels(const float (&v)[NV], float id, int (&q)[NV / 4]) {
    int sum = 0;
    #pragma unroll
    for (int i = 0; i < NV / 4; i++) {
        unsigned w = 0;
        #pragma unroll
        for (int j = 0; j < 4; j++) {
            const int qi = (int) __builtin_roundf(v[4 * i + j] * id);       // round half away from zero, as roundf on the host
            sum += qi;
            w |= ((unsigned) qi & 0xFFu) << (8 * j);
        }
        q[i] = (int) w;
    }
    return sum;
}

// decode (N = 1): x is an f32 row, optionally LayerNorm-ed first.  A 256-thread workgroup owns 16 output rows and follows the
// f16 decode GEMV of kernels.hip (gemv_ln_wg_kernel), whose structure the in-kernel time line of round 2 paid for:
//   0. what the first loads need arrives as explicit leading parameters (preloaded into SGPRs at wave launch): the row to normalise is
//      requested first, the weight blocks of every lane's chain (up to 8) right behind it, the argument struct is read behind both;
//   1. LayerNorm (ggml_norm: double sums) by wave 0 alone, the normalised f32 row published in LDS - one barrier instead of the two
//      cross-wave reductions every workgroup used to run;
//   2. q8 quantisation of the row (ggml quantize_row_q8_0 / q8_1) by the whole workgroup: four consecutive elements per thread, the
//      block maximum and level sum over the 8 threads of a block by three DPP steps, levels packed into LDS; second barrier;
//   3. wave w dots rows 4 w .. 4 w + 3: lane c of a row walks the blocks c, c + 16, ... (chain c of C1q): exact integer block sums on
//      v_dot4_i32_i8, per-format scaling, plain float adds in ascending block order, C1 tree over the 16 lanes.
//   PS (QKV product of a decode step, block_size 1024): as in the f16 kernel, copies of the q workgroups behind the main grid repeat the
//      16 q rows of one C2 block and score the cached keys against them (partial scores for attn_ps_kernel).
// Lock-step batches: grid.y = slot (own x row, state, KV cache).
// NBLK = K / 128 is a compile-time constant and every load is unconditional (lanes without a block in the last round re-read their
// neighbour's): guarded loads compile to load / s_waitcnt vmcnt(0) / branch chains, one exposed memory round trip each (the first cut
// of this kernel took 2.4 us to get the row into LDS that way, in-kernel time line profiles/r03_trace_q4_decode_step.txt).
template <int QT, int NBLK, bool LN, bool LNB, bool PS>
__global__ __launch_bounds__(256) void gemv_q_kernel(const uint8_t * __restrict__ wqs, const half_t * __restrict__ wd, const float * __restrict__ x_f32,
                                                     const float * __restrict__ ln_g, const float * __restrict__ ln_b, const StepState * __restrict__ st,
                                                     const int M, const int parity_rows, const int kpc, const LinArgs a) {
    TRACE_T0();
    TRACE_T1(M);
    [[maybe_unused]] unsigned long long stamp_a = 0, stamp_b = 0;       // diagnostic build: LayerNorm-ed row / quantised row available
    constexpr int K = NBLK * 128, nblk = K / 32;               // q8 blocks per row
    constexpr int NCH = (nblk + 15) / 16;                      // blocks per chain (rounds): the last round may be partial
    constexpr int EPT = K / 64;                                // row elements per lane of the normalising wave
    constexpr int NQ = (K / 4 + 255) / 256;                    // quantisation rounds: four elements per thread and round
    static_assert(!LN || K <= 1024, "LayerNorm-fused quantised GEMV: n_embd <= 1024");
    __shared__ __attribute__((aligned(16))) float xs[LN ? K : 4];        // LayerNorm-ed row
    __shared__ __attribute__((aligned(16))) int xq[nblk * 8];            // q8 levels, four per dword: block b = dwords 8 b .. 8 b + 7
    __shared__ float xd[nblk], xsm[nblk];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 15, rg = lane >> 4;
    const int slot = blockIdx.y;                              // lock-step batch: one sequence per grid.y (a single sequence: grid.y == 1)
    [[maybe_unused]] const int n_main = (M + 15) >> 4, n_q = K >> 4;     // PS: the QKV product, n_embd == K
    [[maybe_unused]] const bool copy = PS && (int) blockIdx.x >= n_main;
    [[maybe_unused]] const int rep = copy ? ((int) blockIdx.x - n_main) / n_q : 0;
    const int wg = copy ? ((int) blockIdx.x - n_main) % n_q : (int) blockIdx.x;
    const int m = wg * 16 + wave * 4 + rg;
    const int row_off = parity_rows ? parity_rows * (st->step & 1) : 0;
    const bool live = m < M;
    const size_t wrow = (size_t) (row_off + (live ? m : 0)) * nblk;
    const float * xrow = x_f32 + (size_t) slot * K;
    // ---- 0. requests: the row first (loads return in order), then the weights
    [[maybe_unused]] float xv[EPT], gv[EPT], bv[EPT];
    [[maybe_unused]] float4 xdir[NQ];                         // no LayerNorm: thread t quantises elements 4 (t + 256 r) .. + 3
    if constexpr (LN) {
        if (wave == 0) {
            #pragma unroll
            for (int i = 0; i < EPT; i++) xv[i] = xrow[lane + 64 * i];
            #pragma unroll
            for (int i = 0; i < EPT; i++) { gv[i] = ln_g[lane + 64 * i]; if constexpr (LNB) bv[i] = ln_b[lane + 64 * i]; else bv[i] = 0.0f; }
        }
    } else {
        #pragma unroll
        for (int r = 0; r < NQ; r++) xdir[r] = *reinterpret_cast<const float4 *>(xrow + 4 * min(tid + 256 * r, K / 4 - 1));
    }
    __builtin_amdgcn_sched_barrier(0);
    RawBlock<QT> wb[NCH];
    #pragma unroll
    for (int i = 0; i < NCH; i++) {
        const size_t bi = wrow + min(c + 16 * i, nblk - 1);
        if constexpr (QTraits<QT>::wide) { const uint4 * p = reinterpret_cast<const uint4 *>(wqs) + 2 * bi; wb[i].qs = p[0]; wb[i].qs2 = p[1]; }
        else wb[i].qs = reinterpret_cast<const uint4 *>(wqs)[bi];
        wb[i].d = wd[bi];
    }
    const int n_past_now = st ? st[slot].n_past : 0;          // on the preloaded state pointer, not through the argument struct
    __builtin_amdgcn_sched_barrier(0);                        // everything above goes out on the preloaded arguments alone
    #pragma unroll
    for (int i = 0; i < NCH; i++) {
        const size_t bi = wrow + min(c + 16 * i, nblk - 1);
        if constexpr (QTraits<QT>::has_h) wb[i].qh = a.wq.qh[bi];
        if constexpr (QTraits<QT>::has_m) wb[i].m = a.wq.m[bi];
    }
    EpiPre pre = epilogue_prefetch(a, slot, live ? m : 0, row_off);
    if (PS || a.epi == EPI_QKV) pre.n_past = n_past_now;
    // partial scores (copies only): keys rep * kpc + tid (+ 256); d-quads 4 blk .. 4 blk + 3 of head hq; requested once the row is in LDS
    [[maybe_unused]] float4 kq[2][4];
    [[maybe_unused]] const int m0 = wg * 16;
    [[maybe_unused]] const int hq = m0 >> 6, blk = (m0 >> 4) & 3;

    // ---- 1. LayerNorm by wave 0 (ggml_norm: double sums, eps on the variance; bark.cpp:1265-1274)
    if constexpr (LN) {
        if (wave == 0) {
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
                float u = xv[i] * scale;
                u = u * gv[i];
                if constexpr (LNB) u = u + bv[i];
                xs[lane + 64 * i] = u;
            }
        }
        __syncthreads();
        TRACE_SET(stamp_a, xs[0]);
    }
    if constexpr (PS) {
        if (copy) {
            const BufRsrc kr = buf_rsrc(reinterpret_cast<const float4 *>(a.kc) + ((size_t) hq * 16 + 4 * blk) * 1024 + rep * kpc);     // PS implies P == 1024
            #pragma unroll
            for (int i = 0; i < 4; i++) kq[0][i] = buf_ld_f4(kr, (unsigned) tid * 16u, (unsigned) i * 16384u);
            if (tid + 256 < kpc) {
                #pragma unroll
                for (int i = 0; i < 4; i++) kq[1][i] = buf_ld_f4(kr, (unsigned) tid * 16u + 4096u, (unsigned) i * 16384u);
            }
        }
    }
    // ---- 2. q8 quantisation: thread t owns elements 4 t' .. 4 t' + 3 (t' = t + 256 r), the 8 threads 8 b .. 8 b + 7 own block b
    #pragma unroll
    for (int r = 0; r < NQ; r++) {
        const int t4 = tid + 256 * r;
        float4 f;
        if constexpr (LN) f = *reinterpret_cast<const float4 *>(xs + 4 * min(t4, K / 4 - 1)); else f = xdir[r];
        const float v[4] = {f.x, f.y, f.z, f.w};
        float amax = fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3])));
        amax = fmaxf(amax, dpp_f32<DPP_XOR1>(amax)); amax = fmaxf(amax, dpp_f32<DPP_XOR2>(amax)); amax = fmaxf(amax, dpp_f32<DPP_HALF_MIRROR>(amax));
        const float d = amax / 127.0f;
        const float id = d != 0.0f ? 1.0f / d : 0.0f;
        int q[1];
        int sum = quantize_levels<4>(v, id, q);
        if constexpr (QTraits<QT>::has_m) {
            sum += dpp_i32<DPP_XOR1>(sum); sum += dpp_i32<DPP_XOR2>(sum); sum += dpp_i32<DPP_HALF_MIRROR>(sum);
        }
        if (4 * t4 < K) {                                      // K is a multiple of 32: whole 8-thread groups are in or out together
            xq[t4] = q[0];
            if ((t4 & 7) == 0) { xd[t4 >> 3] = (float) to_half(d); if constexpr (QTraits<QT>::has_m) xsm[t4 >> 3] = (float) to_half((float) sum * d); }
        }
    }
    __syncthreads();
    TRACE_SET(stamp_b, xd[0]);

    // ---- 3. dot
    float acc = 0.0f;
    #pragma unroll
    for (int i = 0; i < NCH; i++) {
        const int b = c + 16 * i;
        const int bl = min(b, nblk - 1);
        const int4 q0 = *reinterpret_cast<const int4 *>(xq + 8 * bl), q1 = *reinterpret_cast<const int4 *>(xq + 8 * bl + 4);
        const int q[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
        int w[8]; unpack_raw<QT>(wb[i], w);
        const int sumi = dot_q4_q8(w, q);
        const float tb = block_term<QT>(sumi, (float) wb[i].d, QTraits<QT>::has_m ? (float) wb[i].m : 0.0f, xd[bl], QTraits<QT>::has_m ? xsm[bl] : 0.0f);
        const float nxt = acc + tb;
        acc = ((i + 1) * 16 <= nblk || b < nblk) ? nxt : acc;  // a lane without a block in the last round keeps its sum
    }
    TRACE_T2(acc);
    acc = wave_xor_add16(acc);
    if (live && c == 0 && !copy) linear_epilogue_pre(a, slot, m, acc, pre);
    if constexpr (PS) {
        if (blockIdx.x == 0 && tid == 0 && pre.n_past > (((int) gridDim.x - n_main) / n_q) * kpc) const_cast<StepState *>(st)->fault = 1;      // see gemv_ln_wg_kernel
        __shared__ float qsh[16];
        if (copy) {                                              // uniform per workgroup
            if (c == 0) qsh[wave * 4 + rg] = a.bias ? acc + pre.bias : acc;       // the q value the epilogue stores
            __syncthreads();
            float qb[16];
            #pragma unroll
            for (int i = 0; i < 16; i++) qb[i] = qsh[i];
            const int j = rep * kpc + tid;
            if (j < pre.n_past) a.ps[((size_t) hq * 4 + blk) * a.P + j] = score_block_f4(kq[0], qb);
            if (tid + 256 < kpc && j + 256 < pre.n_past) a.ps[((size_t) hq * 4 + blk) * a.P + j + 256] = score_block_f4(kq[1], qb);
        }
    }
#ifdef BARK_TRACE
    trace_emit(a.tr, _tr0, _tr1, _tr2, trace_clock(), stamp_a, stamp_b);
#endif
}

// rows (N > 1): q8 quantisation of the activation rows once (optionally with the LayerNorm in front), one wave per row
struct Q8RowsArgs { const float * x; int N, K; const float * ln_g; const float * ln_b; int8_t * q; float * d; float * dT; float * s; float * sT; };
__global__ __launch_bounds__(64) void q8_rows_kernel(const Q8RowsArgs a) {
    const int lane = threadIdx.x, n = blockIdx.x;
    const int K = a.K, nblk = K >> 5;
    const float * xr = a.x + (size_t) n * K;
    float mean = 0.0f, scale = 1.0f;
    if (a.ln_g) {
        double s1 = 0.0;
        for (int e = lane; e < K; e += 64) s1 += (double) xr[e];
        s1 = wave_sum(s1);
        mean = (float) (s1 / (double) K);
        double s2 = 0.0;
        for (int e = lane; e < K; e += 64) { const float u = xr[e] - mean; s2 += (double) (u * u); }
        s2 = wave_sum(s2);
        const float var = (float) (s2 / (double) K);
        scale = 1.0f / sqrtf(var + 1e-5f);
    }
    for (int b = lane; b < nblk; b += 64) {
        float v[32];
        const float4 * xp = reinterpret_cast<const float4 *>(xr + (b << 5));
        #pragma unroll
        for (int i = 0; i < 8; i++) { const float4 f = xp[i]; v[4 * i] = f.x; v[4 * i + 1] = f.y; v[4 * i + 2] = f.z; v[4 * i + 3] = f.w; }
        if (a.ln_g) {
            #pragma unroll
            for (int j = 0; j < 32; j++) {
                float u = (v[j] - mean) * scale;
                u = u * a.ln_g[(b << 5) + j];
                if (a.ln_b) u = u + a.ln_b[(b << 5) + j];
                v[j] = u;
            }
        }
        float amax = 0.0f;
        #pragma unroll
        for (int j = 0; j < 32; j++) amax = fmaxf(amax, fabsf(v[j]));
        const float d = amax / 127.0f;
        const float id = d != 0.0f ? 1.0f / d : 0.0f;
        int q[8];
        const int sum = quantize_levels<32>(v, id, q);
        int4 * qp = reinterpret_cast<int4 *>(a.q + (size_t) n * K + (b << 5));
        qp[0] = make_int4(q[0], q[1], q[2], q[3]);
        qp[1] = make_int4(q[4], q[5], q[6], q[7]);
        const float dh = (float) to_half(d), sh = (float) to_half((float) sum * d);
        a.d[(size_t) n * nblk + b] = dh;
        a.s[(size_t) n * nblk + b] = sh;
        // block-major copies for the MFMA kernel ([K/32][1024])
        a.dT[(size_t) b * 1024 + n] = dh;
        a.sT[(size_t) b * 1024 + n] = sh;
    }
}

// rows (N > 1), v_dot4 version: NB pre-quantised activation rows per wave share each unpacked weight block.  Kept as the
// cross-check path of the MFMA kernel (BARK_HIP_CROSSCHECK bit 0); bound by L2 re-reads of the weights.
struct QRowsArgs { LinArgs lin; const int8_t * q; const float * d, * dT, * s, * sT; };
template <int QT, int NB>
__global__ __launch_bounds__(64) void gemm_q_rows_kernel(const QRowsArgs qa) {
    const LinArgs & a = qa.lin;
    const int lane = threadIdx.x;
    const int c = lane & 15, rg = lane >> 4;
    const int m = blockIdx.x * 4 + rg;
    const int n0 = blockIdx.y * NB;
    const int K = a.K, nblk = K >> 5;
    const bool live = m < a.M;
    const size_t wrow = (size_t) (live ? m : 0) * nblk;
    float acc[NB];
    #pragma unroll
    for (int i = 0; i < NB; i++) acc[i] = 0.0f;
    for (int b = c; b < nblk; b += 16) {
        const RawBlock<QT> wb = load_raw<QT>(a.wq, wrow + b);
        int w[8]; unpack_raw<QT>(wb, w);
        const float dw = (float) wb.d, mw = QTraits<QT>::has_m ? (float) wb.m : 0.0f;
        #pragma unroll
        for (int i = 0; i < NB; i++) {
            const int n = min(n0 + i, a.N - 1);
            const int4 * qp = reinterpret_cast<const int4 *>(qa.q + (size_t) n * K + (b << 5));
            const int4 q0 = qp[0], q1 = qp[1];
            const int q[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
            const int sumi = dot_q4_q8(w, q);
            const float tb = block_term<QT>(sumi, dw, mw, qa.d[(size_t) n * nblk + b], QTraits<QT>::has_m ? qa.s[(size_t) n * nblk + b] : 0.0f);
            acc[i] = acc[i] + tb;
        }
    }
    #pragma unroll
    for (int i = 0; i < NB; i++) {
        const float r = wave_xor_add16(acc[i]);
        if (live && c == 0 && n0 + i < a.N) linear_epilogue(a, n0 + i, m, r, 0);
    }
}

// rows (N > 1) on the matrix cores: v_mfma_i32_32x32x32_i8 multiplies exactly one weight block (32 levels of 32 output rows,
// widened to int8) by one q8 block of 32 activation rows - the int32 results are the exact block sums of C1q.
// Workgroup tile: 64 activation rows x 32 output rows, 8 waves; wave w owns the chains 2 w and 2 w + 1 (blocks 2 w + 16 i
// and 2 w + 1 + 16 i, ascending), scales every block sum per format (block_term) and adds it to the chain in f32.
// The weight scale (and minimum) is per lane (the lane's output row), the 16 activation scales of a lane's accumulator rows
// come from the block-major copies of d8 / s8 as float4 loads.  Chains 2 w and 2 w + 1 meet in registers (tree level xor 1),
// the eight pair sums of an output in LDS (levels xor 2, 4, 8).
constexpr int Q4G_TM = 32, Q4G_TN = 64, Q4G_LD = 33;
typedef int intx4 __attribute__((ext_vector_type(4)));
typedef int intx16 __attribute__((ext_vector_type(16)));
template <int QT>
__global__ __launch_bounds__(512) void gemm_q_mfma_kernel(const QRowsArgs qa) {
    extern __shared__ float q4g_red[];                         // [8 chain pairs][64 activation rows][33]
    const LinArgs & a = qa.lin;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = blockIdx.x * Q4G_TM, n0 = blockIdx.y * Q4G_TN;
    const int K = a.K, nblk = K >> 5;
    const int li = lane & 31, kh = lane >> 5;
    const int m = min(m0 + li, a.M - 1);
    const size_t wrow = (size_t) m * nblk;
    const int8_t * xq[2]; const float * xdT[2], * xsT[2];
    #pragma unroll
    for (int nt = 0; nt < 2; nt++) {
        const int n = min(n0 + 32 * nt + li, a.N - 1);
        xq[nt] = qa.q + (size_t) n * K + 16 * kh;
        xdT[nt] = qa.dT + n0 + 32 * nt + 4 * kh;                 // accumulator register r <-> activation row (r & 3) + 8 (r >> 2) + 4 kh
        xsT[nt] = qa.sT + n0 + 32 * nt + 4 * kh;
    }
    floatx16 acc[2][2];
    #pragma unroll
    for (int ch = 0; ch < 2; ch++)
        #pragma unroll
        for (int nt = 0; nt < 2; nt++)
            #pragma unroll
            for (int r = 0; r < 16; r++) acc[ch][nt][r] = 0.0f;
    const intx16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int b0 = 2 * wave; b0 < nblk; b0 += 16) {
        #pragma unroll
        for (int ch = 0; ch < 2; ch++) {
            const int b = b0 + ch;
            if (b < nblk) {                                    // wave-uniform
                const RawBlock<QT> wb = load_raw<QT>(a.wq, wrow + b);
                intx4 xa[2]; float4 d8[2][4], s8[2][4];
                #pragma unroll
                for (int nt = 0; nt < 2; nt++) {
                    xa[nt] = *reinterpret_cast<const intx4 *>(xq[nt] + (b << 5));
                    #pragma unroll
                    for (int g = 0; g < 4; g++) {
                        d8[nt][g] = *reinterpret_cast<const float4 *>(xdT[nt] + (size_t) b * 1024 + 8 * g);
                        if constexpr (QTraits<QT>::has_m) s8[nt][g] = *reinterpret_cast<const float4 *>(xsT[nt] + (size_t) b * 1024 + 8 * g);
                    }
                }
                // lanes 0..31 carry elements 0..15 of the block, lanes 32..63 elements 16..31
                int wlo[4], whi[4];
                unpack_half<QT>(wb, 0, wlo); unpack_half<QT>(wb, 1, whi);
                intx4 wv;
                #pragma unroll
                for (int i = 0; i < 4; i++) wv[i] = kh ? whi[i] : wlo[i];
                const float dw = (float) wb.d, mw = QTraits<QT>::has_m ? (float) wb.m : 0.0f;
                #pragma unroll
                for (int nt = 0; nt < 2; nt++) {
                    const intx16 si = __builtin_amdgcn_mfma_i32_32x32x32_i8(xa[nt], wv, zero, 0, 0, 0);
                    #pragma unroll
                    for (int g = 0; g < 4; g++) {
                        const float dd[4] = {d8[nt][g].x, d8[nt][g].y, d8[nt][g].z, d8[nt][g].w};
                        float ss[4] = {0.0f, 0.0f, 0.0f, 0.0f};
                        if constexpr (QTraits<QT>::has_m) { ss[0] = s8[nt][g].x; ss[1] = s8[nt][g].y; ss[2] = s8[nt][g].z; ss[3] = s8[nt][g].w; }
                        #pragma unroll
                        for (int j = 0; j < 4; j++) {
                            const int r = 4 * g + j;
                            const float tb = block_term<QT>(si[r], dw, mw, dd[j], ss[j]);
                            acc[ch][nt][r] = acc[ch][nt][r] + tb;
                        }
                    }
                }
            }
        }
    }
    // tree level xor 1 in registers, then the 8 pair sums per output through LDS
    float * red = q4g_red + (size_t) wave * (Q4G_TN * Q4G_LD);
    #pragma unroll
    for (int nt = 0; nt < 2; nt++)
        #pragma unroll
        for (int r = 0; r < 16; r++) {
            const int nl = 32 * nt + (r & 3) + 8 * (r >> 2) + 4 * kh;
            red[nl * Q4G_LD + li] = acc[0][nt][r] + acc[1][nt][r];
        }
    __syncthreads();
    #pragma unroll
    for (int k = 0; k < (Q4G_TM * Q4G_TN) / 512; k++) {
        const int o = tid + 512 * k;
        const int nl = o >> 5, ml = o & 31;
        float p[8];
        #pragma unroll
        for (int c = 0; c < 8; c++) p[c] = q4g_red[(size_t) c * (Q4G_TN * Q4G_LD) + nl * Q4G_LD + ml];
        const float r = ((p[0] + p[1]) + (p[2] + p[3])) + ((p[4] + p[5]) + (p[6] + p[7]));
        if (n0 + nl < a.N && m0 + ml < a.M) linear_epilogue(a, n0 + nl, m0 + ml, r, 0);
    }
}

void launch_q8_rows(hipStream_t s, const float * x, int N, int K, const float * ln_g, const float * ln_b, const Q8Scratch & o) {
    if (N > 1024) { kernel_fail("bark-hip: q8 row quantisation handles at most 1024 rows"); }
    Q8RowsArgs a{x, N, K, ln_g, ln_b, o.q, o.d, o.dT, o.s, o.sT};
    hipLaunchKernelGGL(q8_rows_kernel, dim3(N), dim3(64), 0, s, a);
}

template <int QT, int NBLK>
static void launch_gemv_q_n(hipStream_t s, const LinArgs & a) {
    // lock-step batch: grid.y walks the sequences; grid.x rounded up to a multiple of 8 so that every grid.y of a row group
    // lands on the same XCD and re-reads the weight blocks from its L2
    const int gx = (a.M + 15) / 16;
    dim3 grid(a.batched ? (gx + 7) / 8 * 8 : gx, a.batched ? a.nbatch : 1), block(256);
    const float * xr = a.x_f32;
    if (a.ln_g) {
        if constexpr (NBLK <= 8) {
            if (a.ps && a.epi == EPI_QKV && a.P == 1024 && !a.batched) {
                // copies of the q workgroups that score the cached keys: sized as in launch_gemv_n (kernels.hip)
                const int n_main = gx, n_q = a.E / 16, keys = 256 * std::max(1, std::min(a.ng, 4));
                const int fit = std::max(1, std::min(2, (256 - n_main) / n_q));
                const int n_copy = std::max((keys + 511) / 512, std::min(fit, keys / 256));
                const int kpc = ((keys + n_copy - 1) / n_copy + 127) / 128 * 128;
                const dim3 gps(n_main + n_copy * n_q);
                if (a.ln_b) hipLaunchKernelGGL((gemv_q_kernel<QT, NBLK, true, true, true>), gps, block, 0, s, a.wq.qs, a.wq.d, xr, a.ln_g, a.ln_b, a.st, a.M, a.parity_rows, kpc, a);
                else        hipLaunchKernelGGL((gemv_q_kernel<QT, NBLK, true, false, true>), gps, block, 0, s, a.wq.qs, a.wq.d, xr, a.ln_g, a.ln_b, a.st, a.M, a.parity_rows, kpc, a);
            }
            else if (a.ln_b) hipLaunchKernelGGL((gemv_q_kernel<QT, NBLK, true, true, false>), grid, block, 0, s, a.wq.qs, a.wq.d, xr, a.ln_g, a.ln_b, a.st, a.M, a.parity_rows, 0, a);
            else             hipLaunchKernelGGL((gemv_q_kernel<QT, NBLK, true, false, false>), grid, block, 0, s, a.wq.qs, a.wq.d, xr, a.ln_g, a.ln_b, a.st, a.M, a.parity_rows, 0, a);
        } else { kernel_fail("bark-hip: LayerNorm-fused quantised GEMV supports n_embd <= 1024"); }
    } else hipLaunchKernelGGL((gemv_q_kernel<QT, NBLK, false, false, false>), grid, block, 0, s, a.wq.qs, a.wq.d, xr, a.ln_g, a.ln_b, a.st, a.M, a.parity_rows, 0, a);
}

template <int QT>
static void launch_linear_qt(hipStream_t s, const LinArgs & a) {
    if (a.N == 1) {
        switch (a.K >> 7) {           // n_embd in {128, 256, 512, 768, 1024} and 4x those
            case 1:  launch_gemv_q_n<QT, 1>(s, a); break;
            case 2:  launch_gemv_q_n<QT, 2>(s, a); break;
            case 4:  launch_gemv_q_n<QT, 4>(s, a); break;
            case 6:  launch_gemv_q_n<QT, 6>(s, a); break;
            case 8:  launch_gemv_q_n<QT, 8>(s, a); break;
            case 16: launch_gemv_q_n<QT, 16>(s, a); break;
            case 24: launch_gemv_q_n<QT, 24>(s, a); break;
            case 32: launch_gemv_q_n<QT, 32>(s, a); break;
            default: kernel_fail("bark-hip: unsupported K=%d in quantised decode GEMV", a.K);
        }
        return;
    }
    const QRowsArgs qa{a, a.xq.q, a.xq.d, a.xq.dT, a.xq.s, a.xq.sT};
    if (!(crosscheck_mask() & 1)) {                                              // bit 0: v_dot4 row kernel, the cross-check path
        dim3 grid((a.M + Q4G_TM - 1) / Q4G_TM, (a.N + Q4G_TN - 1) / Q4G_TN), block(512);
        hipLaunchKernelGGL((gemm_q_mfma_kernel<QT>), grid, block, 8 * Q4G_TN * Q4G_LD * sizeof(float), s, qa);
        return;
    }
    constexpr int NB = 8;
    dim3 grid((a.M + 3) / 4, (a.N + NB - 1) / NB), block(64);
    hipLaunchKernelGGL((gemm_q_rows_kernel<QT, NB>), grid, block, 0, s, qa);
}
// the MFMA kernels' LDS tree (67.6 KB) is above the default 64 KB limit; set once at load time, outside any stream capture
template <int QT> static void allow_large_lds() {
    (void) hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_q_mfma_kernel<QT>), hipFuncAttributeMaxDynamicSharedMemorySize,
                               8 * Q4G_TN * Q4G_LD * (int) sizeof(float));
}
void init_quant_attributes() {
    allow_large_lds<QT_Q4_0>(); allow_large_lds<QT_Q4_1>(); allow_large_lds<QT_Q5_0>(); allow_large_lds<QT_Q5_1>(); allow_large_lds<QT_Q8_0>();
}

void launch_linear_q(hipStream_t s, const LinArgs & a) {
    if ((a.K & 31) != 0 || a.K > 4096) { kernel_fail("bark-hip: quantised rows must be a multiple of 32 and at most 4096 long"); }
    if (a.N == 1 && (a.K & 127) != 0) { kernel_fail("bark-hip: quantised decode GEMV needs rows that are a multiple of 128 long"); }
    if (a.batched && (a.N != 1 || a.ln_stats)) { kernel_fail("bark-hip: batched quantised products take one row per sequence and in-kernel LayerNorm statistics"); }
    if (a.N == 1 && !a.x_f32) { kernel_fail("bark-hip: quantised GEMV needs an f32 activation row"); }
    if (a.N > 1 && (!a.xq.q || a.parity_rows)) { kernel_fail("bark-hip: quantised row product needs pre-quantised rows"); }
    switch (a.wq.qt) {
        case QT_Q4_0: launch_linear_qt<QT_Q4_0>(s, a); break;
        case QT_Q4_1: launch_linear_qt<QT_Q4_1>(s, a); break;
        case QT_Q5_0: launch_linear_qt<QT_Q5_0>(s, a); break;
        case QT_Q5_1: launch_linear_qt<QT_Q5_1>(s, a); break;
        case QT_Q8_0: launch_linear_qt<QT_Q8_0>(s, a); break;
        default: kernel_fail("bark-hip: unknown weight block format %d", a.wq.qt);
    }
}

// ------------------------------------------------------------------------------------------------
// f32 weights (model files converted without --use-f16).  ggml converts the activation only when the weight type asks for
// it, so both operands are f32 here; the summation order is C1 unchanged (8-element chunks, chunk q -> chain q mod 16, fmaf).
// These are plain kernels - the format is a compatibility path, not a tuned one: decode stages the (LayerNorm-ed) row in
// LDS once per 16 output rows, rows (N > 1) re-read the weights from L2 for every eight activation rows.
// ------------------------------------------------------------------------------------------------
template <bool LN, bool LNB>
__global__ __launch_bounds__(256) void gemv_w32_kernel(const LinArgs a) {
    __shared__ __attribute__((aligned(16))) float xs[4096];
    __shared__ double red[2][4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 15, rg = lane >> 4;
    const int m = blockIdx.x * 16 + wave * 4 + rg;
    const int K = a.K, nchunk = K >> 3;
    const int row_off = a.parity_rows ? a.parity_rows * (a.st->step & 1) : 0;
    const bool live = m < a.M;
    const float * wrow = reinterpret_cast<const float *>(a.wq.qs) + (size_t) (row_off + (live ? m : 0)) * K;
    const EpiPre pre = epilogue_prefetch(a, 0, live ? m : 0, row_off);
    // ---- stage x: thread t owns elements [16 t, 16 t + 16)
    const bool mine = 16 * tid < K;
    const int k0 = mine ? 16 * tid : 0;
    float v[16];
    {
        const float4 * xp = reinterpret_cast<const float4 *>(a.x_f32 + k0);
        #pragma unroll
        for (int i = 0; i < 4; i++) { const float4 f = xp[i]; v[4 * i] = f.x; v[4 * i + 1] = f.y; v[4 * i + 2] = f.z; v[4 * i + 3] = f.w; }
    }
    if constexpr (LN) {
        double s1 = 0.0;
        if (mine) {
            #pragma unroll
            for (int j = 0; j < 16; j++) s1 += (double) v[j];
        }
        s1 = wave_sum(s1);
        if (lane == 0) red[0][wave] = s1;
        __syncthreads();
        const float mean = (float) (((red[0][0] + red[0][1]) + (red[0][2] + red[0][3])) / (double) K);
        double s2 = 0.0;
        #pragma unroll
        for (int j = 0; j < 16; j++) { const float u = v[j] - mean; v[j] = u; if (mine) s2 += (double) (u * u); }
        s2 = wave_sum(s2);
        if (lane == 0) red[1][wave] = s2;
        __syncthreads();
        const float var = (float) (((red[1][0] + red[1][1]) + (red[1][2] + red[1][3])) / (double) K);
        const float scale = 1.0f / sqrtf(var + 1e-5f);
        #pragma unroll
        for (int j = 0; j < 16; j++) {
            float u = v[j] * scale;
            u = u * a.ln_g[k0 + j];
            if constexpr (LNB) u = u + a.ln_b[k0 + j];
            v[j] = u;
        }
    }
    if (mine) {
        #pragma unroll
        for (int i = 0; i < 4; i++) *reinterpret_cast<float4 *>(xs + k0 + 4 * i) = make_float4(v[4 * i], v[4 * i + 1], v[4 * i + 2], v[4 * i + 3]);
    }
    __syncthreads();
    float acc = 0.0f;
    #pragma unroll 4
    for (int q = c; q < nchunk; q += 16) {
        const float4 w0 = *reinterpret_cast<const float4 *>(wrow + (q << 3)), w1 = *reinterpret_cast<const float4 *>(wrow + (q << 3) + 4);
        const float4 x0 = *reinterpret_cast<const float4 *>(xs + (q << 3)), x1 = *reinterpret_cast<const float4 *>(xs + (q << 3) + 4);
        acc = fmaf(w0.x, x0.x, acc); acc = fmaf(w0.y, x0.y, acc); acc = fmaf(w0.z, x0.z, acc); acc = fmaf(w0.w, x0.w, acc);
        acc = fmaf(w1.x, x1.x, acc); acc = fmaf(w1.y, x1.y, acc); acc = fmaf(w1.z, x1.z, acc); acc = fmaf(w1.w, x1.w, acc);
    }
    acc = wave_xor_add16(acc);
    if (live && c == 0) linear_epilogue_pre(a, 0, m, acc, pre);
}
// rows (N > 1): x_f32 holds N rows of length K (already LayerNorm-ed where the operator has one)
template <int NB>
__global__ __launch_bounds__(64) void gemm_w32_rows_kernel(const LinArgs a) {
    const int lane = threadIdx.x;
    const int c = lane & 15, rg = lane >> 4;
    const int m = blockIdx.x * 4 + rg;
    const int n0 = blockIdx.y * NB;
    const int K = a.K, nchunk = K >> 3;
    const bool live = m < a.M;
    const float * wrow = reinterpret_cast<const float *>(a.wq.qs) + (size_t) (live ? m : 0) * K;
    float acc[NB];
    #pragma unroll
    for (int i = 0; i < NB; i++) acc[i] = 0.0f;
    for (int q = c; q < nchunk; q += 16) {
        const float4 w0 = *reinterpret_cast<const float4 *>(wrow + (q << 3)), w1 = *reinterpret_cast<const float4 *>(wrow + (q << 3) + 4);
        #pragma unroll
        for (int i = 0; i < NB; i++) {
            const int n = min(n0 + i, a.N - 1);
            const float * xr = a.x_f32 + (size_t) n * K + (q << 3);
            const float4 x0 = *reinterpret_cast<const float4 *>(xr), x1 = *reinterpret_cast<const float4 *>(xr + 4);
            float r = acc[i];
            r = fmaf(w0.x, x0.x, r); r = fmaf(w0.y, x0.y, r); r = fmaf(w0.z, x0.z, r); r = fmaf(w0.w, x0.w, r);
            r = fmaf(w1.x, x1.x, r); r = fmaf(w1.y, x1.y, r); r = fmaf(w1.z, x1.z, r); r = fmaf(w1.w, x1.w, r);
            acc[i] = r;
        }
    }
    #pragma unroll
    for (int i = 0; i < NB; i++) {
        const float r = wave_xor_add16(acc[i]);
        if (live && c == 0 && n0 + i < a.N) linear_epilogue(a, n0 + i, m, r, 0);
    }
}
void launch_linear_w32(hipStream_t s, const LinArgs & a) {
    if ((a.K & 127) != 0 || a.K > 4096) { kernel_fail("bark-hip: unsupported K=%d in f32 linear op", a.K); }
    if (a.batched || !a.x_f32) { kernel_fail("bark-hip: f32-weight products take f32 rows, one sequence at a time"); }
    if (a.N == 1) {
        dim3 grid((a.M + 15) / 16), block(256);
        if (a.ln_g) {
            if (a.ln_b) hipLaunchKernelGGL((gemv_w32_kernel<true, true>), grid, block, 0, s, a);
            else        hipLaunchKernelGGL((gemv_w32_kernel<true, false>), grid, block, 0, s, a);
        } else hipLaunchKernelGGL((gemv_w32_kernel<false, false>), grid, block, 0, s, a);
        return;
    }
    if (a.ln_g || a.parity_rows) { kernel_fail("bark-hip: f32 row product needs LayerNorm-ed rows"); }
    constexpr int NB = 8;
    dim3 grid((a.M + 3) / 4, (a.N + NB - 1) / NB), block(64);
    hipLaunchKernelGGL((gemm_w32_rows_kernel<NB>), grid, block, 0, s, a);
}

}  // namespace barkhip
