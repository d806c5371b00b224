// kernels.hip - GPT kernels of the MI355X-native Bark engine (gfx950 / CDNA4, wave64).
//
// Every kernel follows the canonical numerics of DESIGN.md (orders C1/C2/C5, explicit fmaf, double
// accumulated LayerNorm / softmax sums, f16 rounding points of ggml's CPU backend) so that the CPU
// oracle reproduces its results bit for bit.  Built with -ffp-contract=off.
//
//   decode (N = 1)      : gemv_kernel (16 lanes per output row = the 16 chains of C1, coalesced
//                         16-byte weight loads, LayerNorm fused as prologue, bias / residual /
//                         GELU-LUT / KV-append fused as epilogue), attn_decode_kernel (one
//                         workgroup per head, thread per key for C2, wave per chain for C5).
//   prefill / fine (N>1): gemm_kernel (v_mfma_f32_32x32x2_f32: exact f32 fma chains; the 16 chains
//                         of C1 live in 8 waves x 2 accumulator sets and meet in LDS),
//                         attn_qk_kernel / softmax_rows_kernel / attn_pv_kernel.
#include "kernels.h"

#include <cfloat>
#include <cstdio>

namespace barkhip {

typedef float  floatx16 __attribute__((ext_vector_type(16)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));

#define DEVINL __device__ __forceinline__

// Sums over the 16 lanes of a DPP row in the C1/C5 tree order (partner xor 1, 2, 4, 8).  After the
// xor-1 / xor-2 quad permutes every lane of a quad holds the quad sum, so the half-row mirror (lane i
// <- 7-i) and the row mirror (lane i <- 15-i) deliver exactly the xor-4 / xor-8 partner sums; fp add is
// commutative, so every lane ends with the same bits as the butterfly.  DPP moves cost one VALU op,
// ds_bpermute-based __shfl_xor costs an LDS round trip per stage.
template <int CTRL> DEVINL float dpp_f32(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false));
}
template <int CTRL> DEVINL double dpp_f64(double v) {
    const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = (unsigned) __builtin_amdgcn_update_dpp(0, (int) (unsigned) u, CTRL, 0xF, 0xF, false);
    const unsigned hi = (unsigned) __builtin_amdgcn_update_dpp(0, (int) (unsigned) (u >> 32), CTRL, 0xF, 0xF, false);
    return __builtin_bit_cast(double, ((unsigned long long) hi << 32) | lo);
}
constexpr int DPP_XOR1 = 0xB1, DPP_XOR2 = 0x4E, DPP_HALF_MIRROR = 0x141, DPP_MIRROR = 0x140;
DEVINL float wave_xor_add16(float v) {
    v = v + dpp_f32<DPP_XOR1>(v);
    v = v + dpp_f32<DPP_XOR2>(v);
    v = v + dpp_f32<DPP_HALF_MIRROR>(v);
    v = v + dpp_f32<DPP_MIRROR>(v);
    return v;
}
DEVINL double group16_sum(double v) {
    v += dpp_f64<DPP_XOR1>(v); v += dpp_f64<DPP_XOR2>(v); v += dpp_f64<DPP_HALF_MIRROR>(v); v += dpp_f64<DPP_MIRROR>(v);
    return v;
}
DEVINL float readlane_f32(float v, int lane) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane)); }
DEVINL double readlane_f64(double v, int lane) {
    const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = (unsigned) __builtin_amdgcn_readlane((int) (unsigned) u, lane);
    const unsigned hi = (unsigned) __builtin_amdgcn_readlane((int) (unsigned) (u >> 32), lane);
    return __builtin_bit_cast(double, ((unsigned long long) hi << 32) | lo);
}
// whole-wave reductions: DPP inside each 16-lane row, then the four row results through SGPRs
DEVINL double wave_sum(double v) {
    v = group16_sum(v);
    return (readlane_f64(v, 0) + readlane_f64(v, 16)) + (readlane_f64(v, 32) + readlane_f64(v, 48));
}
DEVINL float wave_max(float v) {
    v = fmaxf(v, dpp_f32<DPP_XOR1>(v)); v = fmaxf(v, dpp_f32<DPP_XOR2>(v));
    v = fmaxf(v, dpp_f32<DPP_HALF_MIRROR>(v)); v = fmaxf(v, dpp_f32<DPP_MIRROR>(v));
    return fmaxf(fmaxf(readlane_f32(v, 0), readlane_f32(v, 16)), fmaxf(readlane_f32(v, 32), readlane_f32(v, 48)));
}
// order-preserving float <-> unsigned map, so that the row maximum can be kept with an integer atomicMax
DEVINL unsigned f32_ordered(float f) { const unsigned b = __builtin_bit_cast(unsigned, f); return (b & 0x80000000u) ? ~b : (b | 0x80000000u); }
DEVINL float f32_unordered(unsigned u) { const unsigned b = (u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u; return __builtin_bit_cast(float, b); }
DEVINL half8 ld_half8(const half_t * p) { return *reinterpret_cast<const half8 *>(p); }
// f32 -> f16, round to nearest even, of an ALREADY ROUNDED f32 value.  The empty asm keeps hipcc from
// folding the producing multiply/add into v_fma_mixlo_f16, which rounds the exact result once and
// differs from the CPU's two roundings in about one of 2^13 cases.
DEVINL half_t to_half(float v) { asm("" : "+v"(v)); return (half_t) v; }

// ggml_gelu on the CPU backend: f16 lookup table, pass-through outside (-10, 10)
DEVINL half_t gelu_lut_apply(float v, const uint16_t * lut) {
    if (v <= -10.0f) return (half_t) 0.0f;
    if (v >= 10.0f) return to_half(v);
    const half_t hv = to_half(v);                         // round to nearest even
    const uint16_t bits = __builtin_bit_cast(uint16_t, hv);
    return __builtin_bit_cast(half_t, lut[bits]);
}

// one element of an embedding row: f16 table, or a quantised table dequantised as ggml_get_rows does (level * d (+ m))
DEVINL float wte_elem(const half_t * wte, const QMat & q, int E, int tok, int e) {
    if (!q.qs) return (float) wte[(size_t) tok * E + e];
    if (q.qt == QT_F32) return reinterpret_cast<const float *>(q.qs)[(size_t) tok * E + e];
    const size_t blk = (size_t) tok * (E >> 5) + (e >> 5);
    const int j = e & 31;
    const float d = (float) q.d[blk];
    if (q.qt == QT_Q8_0) return (float) (int) reinterpret_cast<const int8_t *>(q.qs)[blk * 32 + j] * d;
    const uint8_t byte = q.qs[blk * 16 + (j & 15)];
    int lev = j < 16 ? (byte & 0x0F) : (byte >> 4);
    if (q.qh) lev |= (int) ((q.qh[blk] >> j) & 1u) << 4;
    if (q.qt == QT_Q4_0) lev -= 8;
    if (q.qt == QT_Q5_0) lev -= 16;
    const float v = (float) lev * d;
    return q.m ? v + (float) q.m[blk] : v;
}

// K cache element address: [H][16][P][4] floats; V cache: [H][P][64]
DEVINL size_t kc_index(int h, int d, int pos, int P) { return (((size_t) h * 16 + (d >> 2)) * P + pos) * 4 + (d & 3); }
DEVINL size_t vc_index(int h, int d, int pos, int P) { return ((size_t) h * P + pos) * 64 + d; }

// Operands the epilogue reads, fetched at kernel entry so that their latency overlaps the weight stream.
struct EpiPre { float bias; float res; int n_past; };
DEVINL EpiPre epilogue_prefetch(const LinArgs & a, int n, int m, int row_off) {
    EpiPre p;
    p.bias = a.bias ? a.bias[row_off + m] : 0.0f;
    p.res = a.epi == EPI_RESID ? a.res[(size_t) n * a.M + m] : 0.0f;
    p.n_past = (a.epi == EPI_QKV && a.st) ? (a.batched ? a.st[n].n_past : a.st->n_past) : 0;
    return p;
}
DEVINL void linear_epilogue_pre(const LinArgs & a, int n, int m, float dot, const EpiPre & p) {
    float v = dot;
    if (a.bias) v = v + p.bias;
    switch (a.epi) {
        case EPI_QKV: {
            const int E = a.E;
            if (m < E) { a.q[(size_t) n * E + m] = v; break; }
            // batched decode: row n is sequence slot n with its own cache and position; otherwise rows are consecutive positions
            const int pos = a.pos0 + p.n_past + (a.batched ? 0 : n);
            const size_t slot = a.batched ? (size_t) n * a.kv_slot_stride : 0;
            const int mm = m < 2 * E ? m - E : m - 2 * E;
            const int h = mm >> 6, d = mm & 63;
            if (m < 2 * E) a.kc[slot + kc_index(h, d, pos, a.P)] = v; else a.vc[slot + vc_index(h, d, pos, a.P)] = v;
            break;
        }
        case EPI_RESID: a.res[(size_t) n * a.M + m] = v + p.res; break;                          // cur + inpL (bark.cpp:1352,1388)
        case EPI_GELU: {
            const half_t g = gelu_lut_apply(v, a.lut);
            if (a.out_h32) a.out_h32[(size_t) n * a.M + m] = (float) g; else a.out_h[(size_t) n * a.M + m] = g;
            break;
        }
        default:        a.out[(size_t) n * a.ld_out + m] = v; break;
    }
}
DEVINL void linear_epilogue(const LinArgs & a, int n, int m, float dot, int row_off) {
    float v = dot;
    if (a.bias) v = v + a.bias[row_off + m];
    switch (a.epi) {
        case EPI_QKV: {
            const int E = a.E;
            if (m < E) { a.q[(size_t) n * E + m] = v; break; }
            const int pos = a.pos0 + (a.st ? a.st->n_past : 0) + n;
            const int mm = m < 2 * E ? m - E : m - 2 * E;
            const int h = mm >> 6, d = mm & 63;
            if (m < 2 * E) a.kc[kc_index(h, d, pos, a.P)] = v; else a.vc[vc_index(h, d, pos, a.P)] = v;
            break;
        }
        case EPI_RESID: { float * r = a.res + (size_t) n * a.M + m; *r = v + *r; break; }       // cur + inpL (bark.cpp:1352,1388)
        case EPI_GELU: {
            const half_t g = gelu_lut_apply(v, a.lut);
            if (a.out_h32) a.out_h32[(size_t) n * a.M + m] = (float) g; else a.out_h[(size_t) n * a.M + m] = g;
            break;
        }
        default:        a.out[(size_t) n * a.ld_out + m] = v; break;
    }
}

// ------------------------------------------------------------------------------------------------
// decode GEMV.  One wave = 4 output rows x 16 lanes; lane c of a row owns chain c of C1, i.e. the
// 16-byte chunks c, c+16, c+32, ... of that weight row: the wave's loads are four fully used
// 256-byte row segments per instruction.  x is either an f16 vector or (LN = true) an f32 row that
// every 16-lane group normalises redundantly in registers (no LDS, no barrier).
// ------------------------------------------------------------------------------------------------
// NBLK = K / 128 is a compile-time constant so that every load of a lane (weights, x, LayerNorm
// parameters) is issued up front with no control flow in between: the kernel is one memory round
// trip deep.  One wave per workgroup (4 output rows) spreads the rows over as many CUs as possible.
template <int NBLK, bool LN, bool LNB>
__global__ __launch_bounds__(64) void gemv_kernel(const LinArgs a) {
    const int lane = threadIdx.x;
    const int c = lane & 15, rg = lane >> 4;
    const int m = blockIdx.x * 4 + rg;
    constexpr int K = NBLK * 128;
    const int row_off = a.parity_rows ? a.parity_rows * (a.st->step & 1) : 0;
    const bool live = m < a.M;                              // whole 16-lane groups are live or dead together
    const half_t * wrow = a.W + (size_t) (row_off + (live ? m : 0)) * K + (c << 3);
    const EpiPre pre = epilogue_prefetch(a, 0, live ? m : 0, row_off);
    float acc = 0.0f;

    if constexpr (LN) {
        static_assert(NBLK <= 8, "LayerNorm-fused GEMV keeps the row in registers (n_embd <= 1024)");
        half8 wv[NBLK];
        float4 xa[NBLK][2], ga[NBLK][2], ba[NBLK][2];
        #pragma unroll
        for (int b = 0; b < NBLK; b++) {
            const int k0 = (b * 16 + c) << 3;
            wv[b] = ld_half8(wrow + (b << 7));
            xa[b][0] = *reinterpret_cast<const float4 *>(a.x_f32 + k0); xa[b][1] = *reinterpret_cast<const float4 *>(a.x_f32 + k0 + 4);
            ga[b][0] = *reinterpret_cast<const float4 *>(a.ln_g + k0);  ga[b][1] = *reinterpret_cast<const float4 *>(a.ln_g + k0 + 4);
            if constexpr (LNB) { ba[b][0] = *reinterpret_cast<const float4 *>(a.ln_b + k0); ba[b][1] = *reinterpret_cast<const float4 *>(a.ln_b + k0 + 4); }
        }
        // ggml_norm (+mul, +add): double sums, eps on the variance (bark.cpp:1265-1274)
        // (four partial sums per lane keep the fp64 add chains short; the order of a double sum of floats
        // changes the rounded float mean / variance with probability ~2^-29)
        float xr[NBLK][8];
        double p1[4] = {0.0, 0.0, 0.0, 0.0};
        #pragma unroll
        for (int b = 0; b < NBLK; b++) {
            xr[b][0] = xa[b][0].x; xr[b][1] = xa[b][0].y; xr[b][2] = xa[b][0].z; xr[b][3] = xa[b][0].w;
            xr[b][4] = xa[b][1].x; xr[b][5] = xa[b][1].y; xr[b][6] = xa[b][1].z; xr[b][7] = xa[b][1].w;
            // the wave's four 16-lane groups hold identical rows: group rg sums only the blocks b = rg (mod 4)
            if ((b & 3) == rg) {
                #pragma unroll
                for (int e = 0; e < 8; e++) p1[e & 3] += (double) xr[b][e];
            }
        }
        const double s1 = wave_sum((p1[0] + p1[1]) + (p1[2] + p1[3]));
        const float mean = (float) (s1 / (double) K);
        double p2[4] = {0.0, 0.0, 0.0, 0.0};
        #pragma unroll
        for (int b = 0; b < NBLK; b++) {
            #pragma unroll
            for (int e = 0; e < 8; e++) { const float v = xr[b][e] - mean; xr[b][e] = v; if ((b & 3) == rg) p2[e & 3] += (double) (v * v); }
        }
        const double s2 = wave_sum((p2[0] + p2[1]) + (p2[2] + p2[3]));
        const float var = (float) (s2 / (double) K);
        const float scale = 1.0f / sqrtf(var + 1e-5f);
        #pragma unroll
        for (int b = 0; b < NBLK; b++) {
            const float gg[8] = {ga[b][0].x, ga[b][0].y, ga[b][0].z, ga[b][0].w, ga[b][1].x, ga[b][1].y, ga[b][1].z, ga[b][1].w};
            float bb[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            if constexpr (LNB) { bb[0] = ba[b][0].x; bb[1] = ba[b][0].y; bb[2] = ba[b][0].z; bb[3] = ba[b][0].w; bb[4] = ba[b][1].x; bb[5] = ba[b][1].y; bb[6] = ba[b][1].z; bb[7] = ba[b][1].w; }
            #pragma unroll
            for (int e = 0; e < 8; e++) {
                float v = xr[b][e] * scale;
                v = v * gg[e];
                if constexpr (LNB) v = v + bb[e];
                // mul_mat converts the activation to f16 first
                acc = fmaf((float) wv[b][e], (float) to_half(v), acc);
            }
        }
    } else {
        const half_t * xrow = a.x_f16 + (c << 3);
        constexpr int G = NBLK < 8 ? NBLK : 8;                // loads in flight per lane and operand
        static_assert(NBLK % G == 0, "K/128 must be <= 8 or a multiple of 8");
        half8 wv[2][G], xv[2][G];
        #pragma unroll
        for (int i = 0; i < G; i++) { wv[0][i] = ld_half8(wrow + (i << 7)); xv[0][i] = ld_half8(xrow + (i << 7)); }
        #pragma unroll
        for (int g = 0; g < NBLK / G; g++) {
            if (g + 1 < NBLK / G) {
                #pragma unroll
                for (int i = 0; i < G; i++) {
                    wv[(g + 1) & 1][i] = ld_half8(wrow + (((g + 1) * G + i) << 7));
                    xv[(g + 1) & 1][i] = ld_half8(xrow + (((g + 1) * G + i) << 7));
                }
            }
            #pragma unroll
            for (int i = 0; i < G; i++) {
                #pragma unroll
                for (int e = 0; e < 8; e++) acc = fmaf((float) wv[g & 1][i][e], (float) xv[g & 1][i][e], acc);
            }
        }
    }
    acc = wave_xor_add16(acc);
    if (live && c == 0) linear_epilogue_pre(a, 0, m, acc, pre);
}

template <int NBLK>
static void launch_gemv_n(hipStream_t s, const LinArgs & a) {
    dim3 grid((a.M + 3) / 4), block(64);
    if (a.x_f32) {
        if constexpr (NBLK <= 8) {
            if (a.ln_b) hipLaunchKernelGGL((gemv_kernel<NBLK, true, true>), grid, block, 0, s, a);
            else        hipLaunchKernelGGL((gemv_kernel<NBLK, true, false>), grid, block, 0, s, a);
        } else { fprintf(stderr, "bark-hip: LayerNorm-fused GEMV supports n_embd <= 1024\n"); abort(); }
    } else {
        hipLaunchKernelGGL((gemv_kernel<NBLK, false, false>), grid, block, 0, s, a);
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
        } else { fprintf(stderr, "bark-hip: LayerNorm-fused GEMV supports n_embd <= 1024\n"); abort(); }
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
// whole tile (2 x 2 MFMA tiles x 2 chains = 8 accumulators); per 128-element K block it loads one
// 16-byte chunk per operand row and chain and issues 4 MFMAs (k pairs) per tile.  The 16 chains meet
// in LDS and are added in the C1 tree order.
// ------------------------------------------------------------------------------------------------
constexpr int GEMM_TM = 64, GEMM_TN = 64;
__global__ __launch_bounds__(512) void gemm_kernel(const LinArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];      // [8][64][64]
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    const int n0 = blockIdx.y * GEMM_TN, m0 = blockIdx.x * GEMM_TM;
    const int K = a.K, nblk = K >> 7;
    int nrow[2], mrow[2];
    #pragma unroll
    for (int t = 0; t < 2; t++) {
        nrow[t] = min(n0 + t * 32 + l31, a.N - 1);
        mrow[t] = min(m0 + t * 32 + l31, a.M - 1);
    }
    floatx16 acc[2][2][2];
    #pragma unroll
    for (int s = 0; s < 2; s++) for (int i = 0; i < 2; i++) for (int j = 0; j < 2; j++)
        for (int r = 0; r < 16; r++) acc[s][i][j][r] = 0.0f;

    // operands of one K block (both chains of this wave): [chain][row tile]; double buffered so that the
    // loads of block b+1 are in flight while the 32 MFMAs of block b issue
    // (buffer indices are compile-time constants: runtime-indexed register arrays would go to scratch)
    half8 xa0[2][2], wb0[2][2], xa1[2][2], wb1[2][2];
#define GEMM_LOAD_BLOCK(XA, WB, B)                                                                       \
    _Pragma("unroll") for (int s = 0; s < 2; s++) {                                                      \
        const int koff = (((B) * 16 + 2 * w + s) << 3);                                                  \
        _Pragma("unroll") for (int t = 0; t < 2; t++) {                                                  \
            XA[s][t] = ld_half8(a.x_f16 + (size_t) nrow[t] * K + koff);                                  \
            WB[s][t] = ld_half8(a.W + (size_t) mrow[t] * K + koff);                                      \
        }                                                                                                \
    }
#define GEMM_MFMA_BLOCK(XA, WB)                                                                          \
    _Pragma("unroll") for (int s = 0; s < 2; s++) {                                                      \
        uint4 xu[2], wu[2];                                                                              \
        _Pragma("unroll") for (int t = 0; t < 2; t++) { xu[t] = __builtin_bit_cast(uint4, XA[s][t]); wu[t] = __builtin_bit_cast(uint4, WB[s][t]); } \
        _Pragma("unroll") for (int kp = 0; kp < 4; kp++) {                                               \
            float av[2], bv[2];                                                                          \
            _Pragma("unroll") for (int t = 0; t < 2; t++) {                                              \
                const unsigned xr = kp == 0 ? xu[t].x : kp == 1 ? xu[t].y : kp == 2 ? xu[t].z : xu[t].w; \
                const unsigned wr = kp == 0 ? wu[t].x : kp == 1 ? wu[t].y : kp == 2 ? wu[t].z : wu[t].w; \
                av[t] = (float) __builtin_bit_cast(half_t, (unsigned short) (xr >> sh16));               \
                bv[t] = (float) __builtin_bit_cast(half_t, (unsigned short) (wr >> sh16));               \
            }                                                                                            \
            _Pragma("unroll") for (int i = 0; i < 2; i++)                                                \
                _Pragma("unroll") for (int j = 0; j < 2; j++)                                            \
                    acc[s][i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[s][i][j], 0, 0, 0); \
        }                                                                                                \
    }
    // lanes 32-63 feed the odd element of each f16 pair (the MFMA's second k slot): one per-lane shift selects it
    const unsigned sh16 = half ? 16u : 0u;
    GEMM_LOAD_BLOCK(xa0, wb0, 0)
    int b = 0;
    for (; b + 1 < nblk; b += 2) {
        GEMM_LOAD_BLOCK(xa1, wb1, b + 1)
        __builtin_amdgcn_sched_barrier(0);                   // keep the prefetch ahead of the MFMAs (hipcc sinks it otherwise)
        GEMM_MFMA_BLOCK(xa0, wb0)
        __builtin_amdgcn_sched_barrier(0);
        if (b + 2 < nblk) { GEMM_LOAD_BLOCK(xa0, wb0, b + 2) }
        __builtin_amdgcn_sched_barrier(0);
        GEMM_MFMA_BLOCK(xa1, wb1)
        __builtin_amdgcn_sched_barrier(0);
    }
    if (b < nblk) { GEMM_MFMA_BLOCK(xa0, wb0) }
#undef GEMM_LOAD_BLOCK
#undef GEMM_MFMA_BLOCK
    // chain pair (2w, 2w+1) -> LDS; accumulator register r of lane l holds row (r&3)+8(r>>2)+4*half, col l31
    float * mine = lds + (size_t) w * (GEMM_TN * GEMM_TM);
    #pragma unroll
    for (int i = 0; i < 2; i++)
        #pragma unroll
        for (int j = 0; j < 2; j++)
            #pragma unroll
            for (int r = 0; r < 16; r++) {
                const int row = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half, col = j * 32 + l31;
                mine[row * GEMM_TM + col] = acc[0][i][j][r] + acc[1][i][j][r];
            }
    __syncthreads();
    // each thread finishes 2 x 4 adjacent outputs: every epilogue operand is fetched as one 16-byte access and all
    // of them are requested before the first is used
    float4 bias4[2], res4[2];
    int nn[2], mm[2];
    #pragma unroll
    for (int r = 0; r < 2; r++) {
        const int idx4 = threadIdx.x + 512 * r;
        nn[r] = n0 + (idx4 >> 4); mm[r] = m0 + ((idx4 & 15) << 2);
        const bool ok = nn[r] < a.N && mm[r] < a.M;
        bias4[r] = (a.bias && ok) ? *reinterpret_cast<const float4 *>(a.bias + mm[r]) : float4{0.f, 0.f, 0.f, 0.f};
        res4[r] = (a.epi == EPI_RESID && ok) ? *reinterpret_cast<const float4 *>(a.res + (size_t) nn[r] * a.M + mm[r]) : float4{0.f, 0.f, 0.f, 0.f};
    }
    #pragma unroll
    for (int r = 0; r < 2; r++) {
        const int idx4 = threadIdx.x + 512 * r;
        float4 p[8];
        #pragma unroll
        for (int q = 0; q < 8; q++) p[q] = *reinterpret_cast<const float4 *>(lds + q * (GEMM_TN * GEMM_TM) + idx4 * 4);
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
                const int pos = a.pos0 + (a.st ? a.st->n_past : 0) + n;
                const int m2 = m < 2 * E ? m - E : m - 2 * E;
                const int h = m2 >> 6, d = m2 & 63;                     // d is a multiple of 4: one d-quad of the K layout
                if (m < 2 * E) *reinterpret_cast<float4 *>(a.kc + kc_index(h, d, pos, a.P)) = o;
                else           *reinterpret_cast<float4 *>(a.vc + vc_index(h, d, pos, a.P)) = o;
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
}

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

// decode (N = 1): x is an f32 row, optionally LayerNorm-ed first.  A 256-thread workgroup owns 16 output rows:
//   1. every lane requests the weight blocks of its chain (up to 8) before anything else,
//   2. the q8 quantisation of x (~8 VALU ops per element) is spread over the workgroup - two threads per block of 32,
//      block maximum / level sum through one DPP exchange - and published in LDS once for the 16 rows,
//   3. wave w dots rows 4 w .. 4 w + 3: lane c of a row walks the blocks c, c + 16, ... (chain c of C1q).
// The first version quantised inside every 16-lane group (each lane its own blocks): 1500 VALU instructions per wave
// for K = 3072, 8.0 us per launch; this one measures about half of that (quantised files).
template <int QT, bool LN, bool LNB>
__global__ __launch_bounds__(256) void gemv_q_kernel(const LinArgs a) {
    constexpr int MAXB = 8;                                    // blocks per chain: K <= 4096
    __shared__ int4 xq[128][2];                                // q8 levels of block b: elements 0..15 and 16..31
    __shared__ float xd[128], xs[128];
    __shared__ double red[2][4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 15, rg = lane >> 4;
    const int m = blockIdx.x * 16 + wave * 4 + rg;
    const int K = a.K, nblk = K >> 5;
    const int slot = a.batched ? blockIdx.y : 0;              // lock-step batch: one sequence per grid.y (own x row, state, KV cache)
    const int row_off = a.parity_rows ? a.parity_rows * (a.st->step & 1) : 0;
    const bool live = m < a.M;
    const size_t wrow = (size_t) (row_off + (live ? m : 0)) * nblk;
    RawBlock<QT> wb[MAXB];
    #pragma unroll
    for (int i = 0; i < MAXB; i++) {
        const int b = c + 16 * i;
        if (b < nblk) wb[i] = load_raw<QT>(a.wq, wrow + b);
    }
    const EpiPre pre = epilogue_prefetch(a, slot, live ? m : 0, row_off);

    // ---- x -> q8: thread t quantises elements [16 (t & 1), +16) of block t >> 1
    const int qb = tid >> 1, qh = tid & 1;
    const bool mine = qb < nblk;
    const int k0 = mine ? (qb << 5) + (qh << 4) : 0;
    float v[16];
    {
        const float4 * xp = reinterpret_cast<const float4 *>(a.x_f32 + (size_t) slot * K + k0);
        #pragma unroll
        for (int i = 0; i < 4; i++) { const float4 f = xp[i]; v[4 * i] = f.x; v[4 * i + 1] = f.y; v[4 * i + 2] = f.z; v[4 * i + 3] = f.w; }
    }
    if constexpr (LN) {
        float g[16], bb[16];
        {
            const float4 * gp = reinterpret_cast<const float4 *>(a.ln_g + k0);
            const float4 * bp = reinterpret_cast<const float4 *>((LNB ? a.ln_b : a.ln_g) + k0);
            #pragma unroll
            for (int i = 0; i < 4; i++) {
                const float4 f = gp[i]; g[4 * i] = f.x; g[4 * i + 1] = f.y; g[4 * i + 2] = f.z; g[4 * i + 3] = f.w;
                if constexpr (LNB) { const float4 h = bp[i]; bb[4 * i] = h.x; bb[4 * i + 1] = h.y; bb[4 * i + 2] = h.z; bb[4 * i + 3] = h.w; }
            }
        }
        // ggml_norm: double sums over the row (bark.cpp:1265-1274)
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
            u = u * g[j];
            if constexpr (LNB) u = u + bb[j];
            v[j] = u;
        }
    }
    {
        float amax = 0.0f;
        #pragma unroll
        for (int j = 0; j < 16; j++) amax = fmaxf(amax, fabsf(v[j]));
        amax = fmaxf(amax, dpp_f32<DPP_XOR1>(amax));           // the other half of the block sits in the neighbouring lane
        const float d = amax / 127.0f;
        const float id = d != 0.0f ? 1.0f / d : 0.0f;
        int q[4];
        int sum = quantize_levels<16>(v, id, q);
        sum += __builtin_amdgcn_update_dpp(0, sum, DPP_XOR1, 0xF, 0xF, false);
        if (mine) {
            xq[qb][qh] = make_int4(q[0], q[1], q[2], q[3]);
            if (qh == 0) { xd[qb] = (float) to_half(d); xs[qb] = (float) to_half((float) sum * d); }
        }
    }
    __syncthreads();

    float acc = 0.0f;
    #pragma unroll
    for (int i = 0; i < MAXB; i++) {
        const int b = c + 16 * i;
        if (b < nblk) {
            const int4 q0 = xq[b][0], q1 = xq[b][1];
            const int q[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
            int w[8]; unpack_raw<QT>(wb[i], w);
            const int sumi = dot_q4_q8(w, q);
            const float tb = block_term<QT>(sumi, (float) wb[i].d, QTraits<QT>::has_m ? (float) wb[i].m : 0.0f, xd[b], xs[b]);
            acc = acc + tb;
        }
    }
    acc = wave_xor_add16(acc);
    if (live && c == 0) linear_epilogue_pre(a, slot, m, acc, pre);
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
// cross-check path of the MFMA kernel (BARK_HIP_Q4_ROWS); bound by L2 re-reads of the weights.
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
    if (N > 1024) { fprintf(stderr, "bark-hip: q8 row quantisation handles at most 1024 rows\n"); abort(); }
    Q8RowsArgs a{x, N, K, ln_g, ln_b, o.q, o.d, o.dT, o.s, o.sT};
    hipLaunchKernelGGL(q8_rows_kernel, dim3(N), dim3(64), 0, s, a);
}

template <int QT>
static void launch_linear_qt(hipStream_t s, const LinArgs & a) {
    if (a.N == 1) {
        // lock-step batch: grid.y walks the sequences; grid.x rounded up to a multiple of 8 so that every grid.y of a row group
        // lands on the same XCD and re-reads the weight blocks from its L2
        const int gx = (a.M + 15) / 16;
        dim3 grid(a.batched ? (gx + 7) / 8 * 8 : gx, a.batched ? a.nbatch : 1), block(256);
        if (a.ln_g) {
            if (a.ln_b) hipLaunchKernelGGL((gemv_q_kernel<QT, true, true>), grid, block, 0, s, a);
            else        hipLaunchKernelGGL((gemv_q_kernel<QT, true, false>), grid, block, 0, s, a);
        } else hipLaunchKernelGGL((gemv_q_kernel<QT, false, false>), grid, block, 0, s, a);
        return;
    }
    static const bool force_rows = getenv("BARK_HIP_Q4_ROWS") != nullptr;        // v_dot4 row kernel, the cross-check path
    const QRowsArgs qa{a, a.xq.q, a.xq.d, a.xq.dT, a.xq.s, a.xq.sT};
    if (!force_rows) {
        static const bool attr = [] {
            (void) hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_q_mfma_kernel<QT>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       8 * Q4G_TN * Q4G_LD * (int) sizeof(float));
            return true;
        }();
        (void) attr;
        dim3 grid((a.M + Q4G_TM - 1) / Q4G_TM, (a.N + Q4G_TN - 1) / Q4G_TN), block(512);
        hipLaunchKernelGGL((gemm_q_mfma_kernel<QT>), grid, block, 8 * Q4G_TN * Q4G_LD * sizeof(float), s, qa);
        return;
    }
    constexpr int NB = 8;
    dim3 grid((a.M + 3) / 4, (a.N + NB - 1) / NB), block(64);
    hipLaunchKernelGGL((gemm_q_rows_kernel<QT, NB>), grid, block, 0, s, qa);
}
static void launch_linear_q(hipStream_t s, const LinArgs & a) {
    if ((a.K & 31) != 0 || a.K > 4096) { fprintf(stderr, "bark-hip: quantised rows must be a multiple of 32 and at most 4096 long\n"); abort(); }
    if (a.batched && (a.N != 1 || a.ln_stats)) { fprintf(stderr, "bark-hip: batched quantised products take one row per sequence and in-kernel LayerNorm statistics\n"); abort(); }
    if (a.N == 1 && !a.x_f32) { fprintf(stderr, "bark-hip: quantised GEMV needs an f32 activation row\n"); abort(); }
    if (a.N > 1 && (!a.xq.q || a.parity_rows)) { fprintf(stderr, "bark-hip: quantised row product needs pre-quantised rows\n"); abort(); }
    switch (a.wq.qt) {
        case QT_Q4_0: launch_linear_qt<QT_Q4_0>(s, a); break;
        case QT_Q4_1: launch_linear_qt<QT_Q4_1>(s, a); break;
        case QT_Q5_0: launch_linear_qt<QT_Q5_0>(s, a); break;
        case QT_Q5_1: launch_linear_qt<QT_Q5_1>(s, a); break;
        case QT_Q8_0: launch_linear_qt<QT_Q8_0>(s, a); break;
        default: fprintf(stderr, "bark-hip: unknown weight block format %d\n", a.wq.qt); abort();
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
static void launch_linear_w32(hipStream_t s, const LinArgs & a) {
    if ((a.K & 127) != 0 || a.K > 4096) { fprintf(stderr, "bark-hip: unsupported K=%d in f32 linear op\n", a.K); abort(); }
    if (a.batched || !a.x_f32) { fprintf(stderr, "bark-hip: f32-weight products take f32 rows, one sequence at a time\n"); abort(); }
    if (a.N == 1) {
        dim3 grid((a.M + 15) / 16), block(256);
        if (a.ln_g) {
            if (a.ln_b) hipLaunchKernelGGL((gemv_w32_kernel<true, true>), grid, block, 0, s, a);
            else        hipLaunchKernelGGL((gemv_w32_kernel<true, false>), grid, block, 0, s, a);
        } else hipLaunchKernelGGL((gemv_w32_kernel<false, false>), grid, block, 0, s, a);
        return;
    }
    if (a.ln_g || a.parity_rows) { fprintf(stderr, "bark-hip: f32 row product needs LayerNorm-ed rows\n"); abort(); }
    constexpr int NB = 8;
    dim3 grid((a.M + 3) / 4, (a.N + NB - 1) / NB), block(64);
    hipLaunchKernelGGL((gemm_w32_rows_kernel<NB>), grid, block, 0, s, a);
}

void launch_linear(hipStream_t s, const LinArgs & a) {
    if (a.wq.qs && a.wq.qt == QT_F32) { launch_linear_w32(s, a); return; }
    if (a.wq.qs) { launch_linear_q(s, a); return; }
    if ((a.K & 127) != 0 || a.K > 4096) { fprintf(stderr, "bark-hip: unsupported K=%d in linear op\n", a.K); abort(); }
    if (a.N > 1 && ((a.M & 3) || (a.epi == EPI_LOGITS && (a.ld_out & 3)))) { fprintf(stderr, "bark-hip: batched linear op needs M %% 4 == 0\n"); abort(); }
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
            default: fprintf(stderr, "bark-hip: unsupported K=%d in batched decode GEMV\n", a.K); abort();
        }
        return;
    }
    if (a.N == 1) {
        switch (nblk) {           // n_embd in {128, 256, 512, 768, 1024} and 4x those
            case 1:  launch_gemv_n<1>(s, a); break;
            case 2:  launch_gemv_n<2>(s, a); break;
            case 4:  launch_gemv_n<4>(s, a); break;
            case 6:  launch_gemv_n<6>(s, a); break;
            case 8:  launch_gemv_n<8>(s, a); break;
            case 16: launch_gemv_n<16>(s, a); break;
            case 24: launch_gemv_n<24>(s, a); break;
            case 32: launch_gemv_n<32>(s, a); break;
            default: fprintf(stderr, "bark-hip: unsupported K=%d in decode GEMV\n", a.K); abort();
        }
        return;
    }
    if (a.x_f32 || a.parity_rows) { fprintf(stderr, "bark-hip: batched linear op needs f16 rows\n"); abort(); }
    static const bool force_rows = getenv("BARK_HIP_GEMM_ROWS") != nullptr;      // cross-check path
    if (force_rows) {
        dim3 grid((a.M + 15) / 16, a.N), block(256);
        hipLaunchKernelGGL((gemv_rows_kernel<32>), grid, block, 0, s, a);
        return;
    }
    dim3 grid((a.M + GEMM_TM - 1) / GEMM_TM, (a.N + GEMM_TN - 1) / GEMM_TN), block(512);
    hipLaunchKernelGGL(gemm_kernel, grid, block, 8 * GEMM_TN * GEMM_TM * sizeof(float), s, a);
}

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

struct FineEmbedArgs { const half_t * wte[8]; QMat wte_q[8]; const float * wpe; int E, n_in; const int32_t * tok; int nn; float * x; };
__global__ void embed_fine_kernel(const FineEmbedArgs a) {
    const int i = blockIdx.x;
    float * out = a.x + (size_t) i * a.E;
    const float * pe = a.wpe + (size_t) i * a.E;
    for (int e = threadIdx.x; e < a.E; e += blockDim.x) {
        float v = 0.0f;                                     // ggml_set_zero(tok_emb), bark.cpp:1936-1937
        for (int cb = 0; cb <= a.nn; cb++) {
            int id = a.tok[cb * 1024 + i];
            id = min(max(id, 0), a.n_in - 1);
            v = v + wte_elem(a.wte[cb], a.wte_q[cb], a.E, id, e);
        }
        out[e] = v + pe[e];
    }
}
void launch_embed_fine(hipStream_t s, const half_t * const wte[8], const QMat * wte_q, const float * wpe, int E, int n_in,
                       const int32_t * tokens_8x1024, int nn, float * x) {
    FineEmbedArgs a; for (int i = 0; i < 8; i++) { a.wte[i] = wte[i]; a.wte_q[i] = wte_q[i]; }
    a.wpe = wpe; a.E = E; a.n_in = n_in; a.tok = tokens_8x1024; a.nn = nn; a.x = x;
    hipLaunchKernelGGL(embed_fine_kernel, dim3(1024), dim3(256), 0, s, a);
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
void launch_ln_rows(hipStream_t s, const float * x, int N, int E, const float * g, const float * b, half_t * out) {
    hipLaunchKernelGGL(ln_rows_kernel, dim3((N + 3) / 4), dim3(256), 0, s, x, N, E, g, b, out);
}

// ------------------------------------------------------------------------------------------------
// decode attention, two launches so that the key stream is spread over the whole chip:
//   attn_scores_kernel : one wave per 64 keys and head (grid P/64 x H); lane = key, C2 = one fmaf chain
//                        over d; the K cache is d-quad major, so a wave's 16-byte loads are contiguous.
//   attn_mix_kernel    : one workgroup (16 waves) per head: softmax statistics over the score row
//                        (max, e = (float) exp((double)(s - max)), double sum), then wave c owns chain c
//                        of C5 (keys c, c+16, ...), lane = d; the 16 chains meet in LDS (tree order).
// Every load is issued before the arithmetic that needs the previous one.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void attn_scores_kernel(const AttnDecodeArgs a) {
    const int h = blockIdx.y, j = blockIdx.x * 64 + threadIdx.x;
    const int P = a.P;
    const float4 * kp = reinterpret_cast<const float4 *>(a.kc) + (size_t) h * 16 * P + j;     // j < P: always inside the cache
    float4 kv[16];
    #pragma unroll
    for (int dq = 0; dq < 16; dq++) kv[dq] = kp[(size_t) dq * P];
    const int ctx = a.st->n_past + 1;
    const float * __restrict__ qh = a.q + h * 64;             // wave-uniform: scalar loads
    float acc = 0.0f;
    #pragma unroll
    for (int dq = 0; dq < 16; dq++) {
        acc = fmaf(kv[dq].x, qh[4 * dq + 0], acc);
        acc = fmaf(kv[dq].y, qh[4 * dq + 1], acc);
        acc = fmaf(kv[dq].z, qh[4 * dq + 2], acc);
        acc = fmaf(kv[dq].w, qh[4 * dq + 3], acc);
    }
    const float sc = acc * 0.125f;                                       // 1/sqrt(64), bark.cpp:1318
    if (j < ctx) a.scores[(size_t) h * P + j] = sc;
    // row maximum for the softmax, kept exactly with an integer atomic (a.hmax[h] is reset by attn_mix_kernel)
    const float wmax = wave_max(j < ctx ? sc : -INFINITY);
    if (threadIdx.x == 0 && blockIdx.x * 64 < ctx) atomicMax(a.hmax + h, f32_ordered(wmax));
}

__global__ __launch_bounds__(1024) void attn_mix_kernel(const AttnDecodeArgs a) {
    __shared__ float es[1024];
    __shared__ double red_d[16];
    __shared__ float part[16][64];
    const int h = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int P = a.P;
    const int ctx = a.st->n_past + 1;
    const float sraw = a.scores[(size_t) h * P + tid];                // tid < P; garbage beyond ctx is masked below
    const float mx = f32_unordered(a.hmax[h]);
    const float * vp = a.vc + (size_t) h * P * 64 + lane;
    float vv[64];
    #pragma unroll
    for (int g = 0; g < 4; g++) {
        if (g == 0 || ctx > 256 * g) {
            #pragma unroll
            for (int i = 0; i < 16; i++) vv[16 * g + i] = vp[(size_t) (wave + 16 * (16 * g + i)) * 64];   // row < P
        }
    }
    float e = 0.0f;
    if (tid < ctx) e = (float) exp((double) (sraw - mx));
    es[tid] = e;
    const double wsum = wave_sum((double) e);
    if (lane == 0) red_d[wave] = wsum;
    __syncthreads();
    if (tid == 0) a.hmax[h] = 0u;                                      // below every encoded float: ready for the next layer
    double sum = 0.0;
    #pragma unroll
    for (int i = 0; i < 16; i++) sum += red_d[i];
    const float inv = (float) (1.0 / sum);
    float acc = 0.0f;
    #pragma unroll
    for (int g = 0; g < 4; g++) {
        if (g == 0 || ctx > 256 * g) {
            #pragma unroll
            for (int i = 0; i < 16; i++) { const int j = wave + 16 * (16 * g + i); if (j < ctx) acc = fmaf(vv[16 * g + i], es[j] * inv, acc); }
        }
    }
    part[wave][lane] = acc;
    __syncthreads();
    if (tid < 64) {
        float p[16];
        #pragma unroll
        for (int c = 0; c < 16; c++) p[c] = part[c][tid];
        #pragma unroll
        for (int st = 1; st < 16; st <<= 1)
            #pragma unroll
            for (int c = 0; c < 16; c += 2 * st) p[c] = p[c] + p[c + st];
        if (a.att32) a.att32[h * 64 + tid] = p[0]; else a.att[h * 64 + tid] = to_half(p[0]);
    }
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
template <int G> DEVINL void load_k_group(float4 (&kv)[16], const float4 * kp, int P) {
    #pragma unroll
    for (int dq = 0; dq < 16; dq++) kv[dq] = kp[(size_t) dq * P + 256 * G];
}
DEVINL float score_chain(const float4 (&kv)[16], const float * __restrict__ qh) {
    float acc = 0.0f;
    #pragma unroll
    for (int dq = 0; dq < 16; dq++) {
        acc = fmaf(kv[dq].x, qh[4 * dq + 0], acc);
        acc = fmaf(kv[dq].y, qh[4 * dq + 1], acc);
        acc = fmaf(kv[dq].z, qh[4 * dq + 2], acc);
        acc = fmaf(kv[dq].w, qh[4 * dq + 3], acc);
    }
    return acc * 0.125f;                                      // 1/sqrt(64), bark.cpp:1318
}
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
    const int chain = 4 * wave + (lane >> 4), d4 = lane & 15;
    const float4 * vp = reinterpret_cast<const float4 *>(vc + (size_t) h * P * 64) + (size_t) chain * 16 + d4;   // row `chain`, dims 4*d4..
    float4 k0[16], k1[16];
    load_k_group<0>(k0, kp, P);                                // keys 0..255: always inside the cache
    const int ctx = a.st[slot].n_past + 1;
    if (ctx > 256) load_k_group<1>(k1, kp, P);
    float4 vv[64];
    #pragma unroll
    for (int g = 0; g < 4; g++) {
        if (g == 0 || ctx > 256 * g) {
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
        if (j < ctx) { e = (float) exp((double) (s[i] - mx)); lsum += (double) e; }
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
            #pragma unroll
            for (int i = 0; i < 16; i++) {
                const int j = chain + 16 * (16 * g + i);
                if (j < ctx) {
                    const float p = es[j] * inv;               // p = e * (float)(1/sum), as ggml_soft_max scales in place
                    const float4 v = vv[16 * g + i];
                    acc.x = fmaf(v.x, p, acc.x); acc.y = fmaf(v.y, p, acc.y); acc.z = fmaf(v.z, p, acc.z); acc.w = fmaf(v.w, p, acc.w);
                }
            }
        }
    }
    *reinterpret_cast<float4 *>(&part[chain][4 * d4]) = acc;
    __syncthreads();
    if (tid < 64) {
        float p[16];
        #pragma unroll
        for (int c = 0; c < 16; c++) p[c] = part[c][tid];
        #pragma unroll
        for (int st = 1; st < 16; st <<= 1)
            #pragma unroll
            for (int c = 0; c < 16; c += 2 * st) p[c] = p[c] + p[c + st];
        if (a.att32) a.att32[(size_t) slot * E + h * 64 + tid] = p[0]; else a.att[(size_t) slot * E + h * 64 + tid] = to_half(p[0]);
    }
}

// ------------------------------------------------------------------------------------------------
// decode attention spread over ATTN_SPLIT workgroups per head, without any cross-workgroup traffic.
// One CU streams a head's K and V rows at only ~40-65 GB/s (attn_fused_kernel: two dependent load
// rounds, 512 B per key), and 12 heads leave 244 CUs idle for the longest kernel of the step.
// Workgroup (h, s) scores ALL keys of head h (every workgroup repeats the C2 chains and the softmax
// statistics in the same order, so all of them hold identical bits) but mixes only the value dims
// [16 s, 16 s + 16) - with all 16 C5 chains, so the tree is local.  Per workgroup that is 256 + 64
// instead of 512 bytes per key, all of them requested up front (one memory round trip; the
// workgroup's four waves sit alone on their SIMDs, so ~350 VGPRs per lane are available).
// A variant that also split the keys and exchanged scores through agent-scope atomics measured
// 10.7 us vs 8.2 us fused at ctx 641: each cross-XCD hop costs ~2 us.
// ------------------------------------------------------------------------------------------------
constexpr int ATTN_SPLIT = 4;
__global__ __launch_bounds__(256) void attn_dslice_kernel(const AttnDecodeArgs a) {
    __shared__ float es[1024];
    __shared__ float red_f[4];
    __shared__ double red_d[4];
    __shared__ float part[16][16];
    const int h = blockIdx.x, s = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int P = a.P;
    const float * __restrict__ qh = a.q + h * 64;
    const float4 * kp = reinterpret_cast<const float4 *>(a.kc) + (size_t) h * 16 * P + tid;
    const int chain = tid >> 4, d = tid & 15;
    const float * vp = a.vc + ((size_t) h * P + chain) * 64 + 16 * s + d;        // key `chain`, value dim 16 s + d
    const int ctx = a.st->n_past + 1;
    float4 k0[16], k1[16], k2[16], k3[16];
    load_k_group<0>(k0, kp, P);
    if (ctx > 256) load_k_group<1>(k1, kp, P);
    if (ctx > 512) load_k_group<2>(k2, kp, P);
    if (ctx > 768) load_k_group<3>(k3, kp, P);
    float vv[64];
    #pragma unroll
    for (int g = 0; g < 4; g++) {
        if (g == 0 || ctx > 256 * g) {
            #pragma unroll
            for (int i = 0; i < 16; i++) vv[16 * g + i] = vp[(size_t) (16 * g + i) * 1024];     // key chain + 16 (16 g + i)
        }
    }
    float sv[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    { const float v = score_chain(k0, qh); if (tid < ctx) sv[0] = v; }
    if (ctx > 256) { const float v = score_chain(k1, qh); if (tid + 256 < ctx) sv[1] = v; }
    if (ctx > 512) { const float v = score_chain(k2, qh); if (tid + 512 < ctx) sv[2] = v; }
    if (ctx > 768) { const float v = score_chain(k3, qh); if (tid + 768 < ctx) sv[3] = v; }
    float mx = wave_max(fmaxf(fmaxf(sv[0], sv[1]), fmaxf(sv[2], sv[3])));
    if (lane == 0) red_f[wave] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red_f[0], red_f[1]), fmaxf(red_f[2], red_f[3]));
    double lsum = 0.0;
    #pragma unroll
    for (int i = 0; i < 4; i++) {
        const int j = tid + 256 * i;
        float e = 0.0f;
        if (j < ctx) { e = (float) exp((double) (sv[i] - mx)); lsum += (double) e; }
        es[j] = e;
    }
    lsum = wave_sum(lsum);
    if (lane == 0) red_d[wave] = lsum;
    __syncthreads();
    const double sum = (red_d[0] + red_d[1]) + (red_d[2] + red_d[3]);
    const float inv = (float) (1.0 / sum);
    float acc = 0.0f;
    #pragma unroll
    for (int g = 0; g < 4; g++) {
        if (g == 0 || ctx > 256 * g) {
            #pragma unroll
            for (int i = 0; i < 16; i++) {
                const int j = chain + 16 * (16 * g + i);
                if (j < ctx) acc = fmaf(vv[16 * g + i], es[j] * inv, acc);
            }
        }
    }
    part[chain][d] = acc;
    __syncthreads();
    if (tid < 16) {
        float p[16];
        #pragma unroll
        for (int c = 0; c < 16; c++) p[c] = part[c][tid];
        #pragma unroll
        for (int st = 1; st < 16; st <<= 1)
            #pragma unroll
            for (int c = 0; c < 16; c += 2 * st) p[c] = p[c] + p[c + st];
        const int o = h * 64 + 16 * s + tid;
        if (a.att32) a.att32[o] = p[0]; else a.att[o] = to_half(p[0]);
    }
}

void launch_attn_decode_part(hipStream_t s, const AttnDecodeArgs & a, int parts) {
    if (parts == 4) { hipLaunchKernelGGL(attn_fused_kernel, dim3(a.H, a.nbatch), dim3(256), 0, s, a); return; }
    if (parts == 5) {
        if (a.nbatch != 1 || a.P != 1024) { fprintf(stderr, "bark-hip: value-sliced decode attention needs one sequence and block_size 1024\n"); abort(); }
        hipLaunchKernelGGL(attn_dslice_kernel, dim3(a.H, ATTN_SPLIT), dim3(256), 0, s, a);
        return;
    }
    if (parts & 1) hipLaunchKernelGGL(attn_scores_kernel, dim3(a.P / 64, a.H), dim3(64), 0, s, a);
    if (parts & 2) hipLaunchKernelGGL(attn_mix_kernel, dim3(a.H), dim3(1024), 0, s, a);
}
void launch_attn_decode(hipStream_t s, const AttnDecodeArgs & a) {
    static const bool split = getenv("BARK_HIP_ATTN_SPLIT") != nullptr;      // two-launch variant kept for A/B timing
    static const bool one_wg = getenv("BARK_HIP_ATTN_ONE_WG") != nullptr;    // one workgroup per head (A/B timing)
    if (split) { launch_attn_decode_part(s, a, 3); return; }
    // several sequences in lock step already give H * nbatch workgroups; a single one is spread over H * ATTN_SPLIT
    const bool can_split = a.nbatch == 1 && a.P == 1024 && !one_wg;
    launch_attn_decode_part(s, a, can_split ? 5 : 4);
}

// ------------------------------------------------------------------------------------------------
// prefill / fine attention (materialised scores): S = scale * Q K^T on the f32 matrix cores (C2 = one
// MFMA accumulator chain over d), row softmax, O = P V on the f32 matrix cores (C5: 16 chains in
// 8 waves x 2 accumulator sets, LDS tree).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void attn_qk_kernel(const AttnPrefillArgs a) {
    // workgroup tile 128 queries x 128 keys; wave (wi, wj) owns a 64 x 64 sub-tile = 2 x 2 MFMA tiles, so every
    // 16-byte operand load feeds four MFMAs
    const int h = blockIdx.z, i0 = blockIdx.y * 128, j0 = blockIdx.x * 128;
    const int ctx = a.n_past + a.N;
    if (j0 >= ctx) return;
    if (a.causal && j0 > a.n_past + i0 + 127) return;       // tile entirely masked
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    const int wi = w >> 1, wj = w & 1;
    const float4 * qp[2]; const float4 * kp[2];
    #pragma unroll
    for (int t = 0; t < 2; t++) {
        const int irow = min(i0 + wi * 64 + t * 32 + l31, a.N - 1);
        const int jrow = min(j0 + wj * 64 + t * 32 + l31, ctx - 1);
        qp[t] = reinterpret_cast<const float4 *>(a.q + (size_t) irow * a.ldq + h * 64);
        kp[t] = reinterpret_cast<const float4 *>(a.kc) + (size_t) h * 16 * a.P + jrow;
    }
    floatx16 acc[2][2];
    #pragma unroll
    for (int i = 0; i < 2; i++) for (int j = 0; j < 2; j++) for (int r = 0; r < 16; r++) acc[i][j][r] = 0.0f;
    #pragma unroll 4
    for (int dq = 0; dq < 16; dq++) {
        float4 qv[2], kv[2];
        #pragma unroll
        for (int t = 0; t < 2; t++) { qv[t] = qp[t][dq]; kv[t] = kp[t][(size_t) dq * a.P]; }
        // MFMA k pair (d = 4dq, 4dq+1) then (4dq+2, 4dq+3): lanes 0-31 feed the even d, 32-63 the odd d
        #pragma unroll
        for (int i = 0; i < 2; i++)
            #pragma unroll
            for (int j = 0; j < 2; j++) {
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(half ? qv[i].y : qv[i].x, half ? kv[j].y : kv[j].x, acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(half ? qv[i].w : qv[i].z, half ? kv[j].w : kv[j].z, acc[i][j], 0, 0, 0);
            }
    }
    // A operand = Q (accumulator rows = queries i), B operand = K (accumulator cols = keys j: coalesced stores)
    #pragma unroll
    for (int ti = 0; ti < 2; ti++)
        #pragma unroll
        for (int tj = 0; tj < 2; tj++)
            #pragma unroll
            for (int r = 0; r < 16; r++) {
                const int i = i0 + wi * 64 + ti * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                const int j = j0 + wj * 64 + tj * 32 + l31;
                if (i < a.N && j < ctx) a.scores[((size_t) h * a.N + i) * a.P + j] = acc[ti][tj][r] * 0.125f;    // 1/sqrt(64), bark.cpp:1318
            }
}

__global__ __launch_bounds__(256) void softmax_rows_kernel(const AttnPrefillArgs a) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);     // row = h * N + i
    if (row >= a.H * a.N) return;
    const int i = row % a.N;
    const int ctx = a.n_past + a.N;
    const int valid = a.causal ? min(ctx, a.n_past + i + 1) : ctx;
    float * s = a.scores + (size_t) row * a.P;
    float mx = -INFINITY;
    for (int j = lane; j < valid; j += 64) mx = fmaxf(mx, s[j]);
    mx = wave_max(mx);
    double sum = 0.0;
    for (int j = lane; j < valid; j += 64) { const float e = (float) exp((double) (s[j] - mx)); s[j] = e; sum += (double) e; }
    sum = wave_sum(sum);
    const float inv = (float) (1.0 / sum);
    for (int j = lane; j < valid; j += 64) s[j] = s[j] * inv;
    const int ctx32 = min((ctx + 31) & ~31, a.P);
    for (int j = valid + lane; j < ctx32; j += 64) s[j] = 0.0f;          // masked keys: p == 0
}

__global__ __launch_bounds__(512) void attn_pv_kernel(const AttnPrefillArgs a) {
    __shared__ float part[8][32][64];
    const int h = blockIdx.y, i0 = blockIdx.x * 32;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    const int ctx = a.n_past + a.N;
    const int irow = min(i0 + l31, a.N - 1);
    // causal: rows of this tile see keys <= n_past + i0 + 31
    const int jend = a.causal ? min(ctx, a.n_past + i0 + 32) : ctx;
    const float * prow = a.scores + ((size_t) h * a.N + irow) * a.P;
    const float * vbase = a.vc + (size_t) h * a.P * 64;
    floatx16 acc[2][2];
    #pragma unroll
    for (int s = 0; s < 2; s++) for (int t = 0; t < 2; t++) for (int r = 0; r < 16; r++) acc[s][t][r] = 0.0f;
    const int jlim = (jend + 1) & ~1;                            // P rows are zero-filled up to a multiple of 32 keys
    #pragma unroll 2
    for (int jb = 0; jb < jend; jb += 32) {
        const int j = jb + 2 * w + 16 * half;                   // chains 2w, 2w+1: keys j, j+1 (this half-wave's k slot)
        const bool ok = j < jlim;
        const float2 p2 = ok ? *reinterpret_cast<const float2 *>(prow + j) : float2{0.0f, 0.0f};
        #pragma unroll
        for (int s = 0; s < 2; s++) {
            const bool oks = j + s < jend;
            const float pv = s ? p2.y : p2.x;
            #pragma unroll
            for (int t = 0; t < 2; t++) {
                const float vv = oks ? vbase[(size_t) (j + s) * 64 + t * 32 + l31] : 0.0f;
                acc[s][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(oks ? pv : 0.0f, vv, acc[s][t], 0, 0, 0);
            }
        }
    }
    #pragma unroll
    for (int t = 0; t < 2; t++)
        #pragma unroll
        for (int r = 0; r < 16; r++) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
            part[w][row][t * 32 + l31] = acc[0][t][r] + acc[1][t][r];
        }
    __syncthreads();
    for (int idx = threadIdx.x; idx < 32 * 64; idx += 512) {
        const int row = idx >> 6, d = idx & 63;
        float p[8];
        #pragma unroll
        for (int q = 0; q < 8; q++) p[q] = part[q][row][d];
        const float v = ((p[0] + p[1]) + (p[2] + p[3])) + ((p[4] + p[5]) + (p[6] + p[7]));
        const int i = i0 + row;
        if (i < a.N) { if (a.att32) a.att32[(size_t) i * a.ld_att + h * 64 + d] = v; else a.att[(size_t) i * a.ld_att + h * 64 + d] = to_half(v); }
    }
}

// ------------------------------------------------------------------------------------------------
// Fused prefill / fine attention: one workgroup (8 waves) per (head, 32-query tile); the 32 x ctx score tile
// lives in LDS (row stride 1026 floats: conflict-free column reads), so scores never travel through HBM:
//   1. S = 0.125 * Q K^T   f32 MFMA, wave w takes key tiles w, w+8, ... (C2: one accumulator chain over d)
//   2. row softmax in LDS   (4 rows per wave; max, e = (float) exp((double)(s - max)), double sum)
//   3. O = P V              f32 MFMA, wave w owns chains 2w, 2w+1 of C5; p = e * inv formed at the operand read
//   4. the 16 chains meet in LDS (aliasing the score tile) in tree order
// ------------------------------------------------------------------------------------------------
constexpr int ATT_LD = 1026;
__global__ __launch_bounds__(512) void attn_rows_kernel(const AttnPrefillArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];        // [32][ATT_LD] scores, then [8][32][64] partial sums
    __shared__ float rowinv[32];
    const int h = blockIdx.y, i0 = blockIdx.x * 32;
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
            floatx16 acc;
            #pragma unroll
            for (int r = 0; r < 16; r++) acc[r] = 0.0f;
            #pragma unroll
            for (int dq = 0; dq < 16; dq++) {
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(half ? qv[dq].y : qv[dq].x, half ? kv[dq].y : kv[dq].x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(half ? qv[dq].w : qv[dq].z, half ? kv[dq].w : kv[dq].z, acc, 0, 0, 0);
            }
            #pragma unroll
            for (int r = 0; r < 16; r++) {
                const int i = (r & 3) + 8 * (r >> 2) + 4 * half;       // accumulator row = query, column = key
                lds[i * ATT_LD + jt + l31] = acc[r] * 0.125f;            // 1/sqrt(64), bark.cpp:1318
            }
        };
        // key tiles w, w+8, w+16, w+24 (32 keys each); the next tile's K rows are in flight during the MFMAs
        float4 ka[16], kb[16];
        int jt = w * 32;
        if (a.dbg & 1) jt = jend;
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
        // four independent exp evaluations per lane and trip: the double-precision exp is a long dependent chain
        double sum4[4] = {0.0, 0.0, 0.0, 0.0};
        for (int j = lane; j < valid; j += 256) {
            float e[4];
            #pragma unroll
            for (int u = 0; u < 4; u++) e[u] = j + 64 * u < valid ? ((a.dbg & 2) ? s[j + 64 * u] - mx : (float) exp((double) (s[j + 64 * u] - mx))) : 0.0f;
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
    const int jstop = (a.dbg & 4) ? 0 : jend;
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

void launch_attn_prefill(hipStream_t s, const AttnPrefillArgs & a) {
    static const bool materialised = getenv("BARK_HIP_ATTN_MATERIALISED") != nullptr;   // three-kernel variant kept for A/B checks
    if (!materialised) {
        hipLaunchKernelGGL(attn_rows_kernel, dim3((a.N + 31) / 32, a.H), dim3(512), 32 * ATT_LD * sizeof(float), s, a);
        return;
    }
    const int ctx = a.n_past + a.N;
    hipLaunchKernelGGL(attn_qk_kernel, dim3((ctx + 127) / 128, (a.N + 127) / 128, a.H), dim3(256), 0, s, a);
    hipLaunchKernelGGL(softmax_rows_kernel, dim3((a.H * a.N + 3) / 4), dim3(256), 0, s, a);
    hipLaunchKernelGGL(attn_pv_kernel, dim3((a.N + 31) / 32, a.H), dim3(512), 0, s, a);
}

// ------------------------------------------------------------------------------------------------
// greedy sampling (gpt_argmax_sample, bark.cpp:223-247): l /= 0.7; softmax; first index of the
// largest probability.  p_i = e_i / sum is monotone in e_i = (float) exp((double)(l_i/0.7 - max)), so
// the winner is the first index whose e_i rounds to 1.0f, i.e. l_i/0.7 - max >= -2^-25.
// Picks whose runner-up is within kNearTie are counted in st->near_tie (the float division can
// merge neighbouring probabilities; the host re-checks those).
// ------------------------------------------------------------------------------------------------
constexpr float kTieCut = -2.98023223876953125e-08f;     // -2^-25
constexpr float kNearTie = -4.0e-7f;

__global__ __launch_bounds__(1024) void sample_greedy_kernel(const SampleArgs a) {
    __shared__ float red_f[16];
    __shared__ int red_i[16];
    __shared__ int red_c[16];
    __shared__ float red_s[16];
    __shared__ int next_tok, next_pos;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int slot = blockIdx.x;                               // sequence slot (batched decode); 0 otherwise
    const float * logits = a.logits + (size_t) slot * a.ld_logits;
    constexpr int MAXV = 12;                                   // up to 12288 logits
    float sv[MAXV];
    float mx = -INFINITY;
    #pragma unroll
    for (int k = 0; k < MAXV; k++) {
        const int i = tid + 1024 * k;
        sv[k] = i < a.n ? logits[i] / 0.7f : -INFINITY;        // gpt_argmax_sample divides by 0.7 whatever the temperature
        mx = fmaxf(mx, sv[k]);
    }
    const float last_logit = logits[a.n - 1];
    mx = wave_max(mx);
    if (lane == 0) red_f[wave] = mx;
    __syncthreads();
    mx = red_f[0];
    #pragma unroll
    for (int i = 1; i < 16; i++) mx = fmaxf(mx, red_f[i]);
    int best = INT32_MAX, close = 0;
    float sum = 0.0f;
    #pragma unroll
    for (int k = 0; k < MAXV; k++) {
        const int i = tid + 1024 * k;
        if (i < a.n) {
            const float d = sv[k] - mx;
            if (d >= kTieCut && i < best) best = i;
            if (d >= kNearTie) close++;
            if (a.mode == 0) sum += (float) exp((double) d);
        }
    }
    for (int m = 1; m < 64; m <<= 1) {
        best = min(best, __shfl_xor(best, m, 64));
        close += __shfl_xor(close, m, 64);
        sum += __shfl_xor(sum, m, 64);
    }
    if (lane == 0) { red_i[wave] = best; red_c[wave] = close; red_s[wave] = sum; }
    __syncthreads();
    if (tid == 0) {
        for (int i = 1; i < 16; i++) { best = min(best, red_i[i]); close += red_c[i]; sum += red_s[i]; }
        StepState * st = a.st + slot;
        const int step = st->step;
        int tok = best;
        float eos_p = 0.0f;
        if (a.mode == 0) {
            // eos_p = probability of the LAST logit (bark.cpp:217-218,233-234)
            eos_p = (float) exp((double) (last_logit / 0.7f - mx)) / sum;
            if ((tok == a.eos_token || eos_p >= a.min_eos_p) && st->eos_step == INT32_MAX) st->eos_step = step;
            if (a.eos_trace) a.eos_trace[(size_t) slot * a.out_stride + step] = eos_p;
        } else {
            tok += a.token_base + ((step & 1) ? 1024 : 0);   // slice start (bark.cpp:1829-1841)
        }
        if (close > 1) st->near_tie += 1;
        a.out_tokens[(size_t) slot * a.out_stride + st->n_out] = tok;
        st->n_out += 1;
        st->cur_token = tok;
        st->step = step + 1;
        const int np = st->n_past + a.n_past_add;
        st->n_past = np;
        st->last_eos_p = eos_p;
        next_tok = tok; next_pos = np;
    }
    __syncthreads();
    // embedding of the sampled token for the next decode step (bark.cpp:1250-1259): x = wte[tok] + wpe[n_past]
    if (a.x && next_pos < a.P) {
        const int tok = min(max(next_tok, 0), a.n_in - 1);
        const float * pe = a.wpe + (size_t) next_pos * a.E;
        float * xo = a.x + (size_t) slot * a.E;
        for (int e = tid; e < a.E; e += 1024) xo[e] = wte_elem(a.wte, a.wte_q, a.E, tok, e) + pe[e];
    }
}
// ------------------------------------------------------------------------------------------------
// multinomial sampling on the device (gpt_multinomial_sample, bark.cpp:201-221): l /= temp; softmax;
// std::discrete_distribution.  libstdc++'s distribution normalises the probabilities once more in double, takes the
// running sums (last one forced to 1.0) and returns lower_bound(sums, u) for ONE uniform double u in [0,1) drawn with
// std::generate_canonical<double, 53> - the host draws those u from the context's std::mt19937 in the order the
// reference would (one per sample) and uploads them, so a seed selects the same random stream as in the reference.
// The running sums are formed per thread range + block scan instead of sequentially: a pick can differ from libstdc++
// only if u falls within ~1e-16 of a boundary.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void sample_multinomial_kernel(const SampleArgs a) {
    __shared__ float red_f[16];
    __shared__ double red_d[16];
    __shared__ int red_i[16];
    __shared__ int next_tok, next_pos;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int slot = blockIdx.x;
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
        pv[k] = i < a.n ? logits[i] / a.temp : -INFINITY;
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
    for (int k = 0; k < CH; k++) { pv[k] = tid * CH + k < a.n ? (float) exp((double) (pv[k] - mx)) : 0.0f; fsum += pv[k]; }
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
    int pick = INT32_MAX;
    #pragma unroll
    for (int k = 0; k < CH; k++) {
        const int i = tid * CH + k;
        if (i < a.n) {
            run += (double) pv[k] / total;
            const double cp = i == a.n - 1 ? 1.0 : run;                                    // libstdc++ pins the last running sum to 1.0
            if (cp >= u && i < pick) pick = i;
        }
    }
    for (int m = 1; m < 64; m <<= 1) pick = min(pick, __shfl_xor(pick, m, 64));
    if (lane == 0) red_i[wave] = pick;
    if (tid == ((a.n - 1) / CH)) red_f[0] = eos_p;
    __syncthreads();
    if (tid == 0) {
        for (int i = 1; i < 16; i++) pick = min(pick, red_i[i]);
        int tok = pick;
        float ep = 0.0f;
        if (a.mode == 0) {
            ep = red_f[0];                                      // probability of the LAST logit (bark.cpp:217-218)
            if ((tok == a.eos_token || ep >= a.min_eos_p) && st->eos_step == INT32_MAX) st->eos_step = step;
            if (a.eos_trace) a.eos_trace[(size_t) slot * a.out_stride + step] = ep;
        } else {
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
                                                                     const double * u, int32_t * out, int out_stride) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
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
    for (int k = 0; k < CH; k++) { pv[k] = lane * CH + k < n_cols ? (float) exp((double) (pv[k] - mx)) : 0.0f; fsum += pv[k]; }
    for (int m = 1; m < 64; m <<= 1) fsum += __shfl_xor(fsum, m, 64);
    double dsum = 0.0;
    #pragma unroll
    for (int k = 0; k < CH; k++) { pv[k] = pv[k] / fsum; dsum += (double) pv[k]; }
    const double total = wave_sum(dsum);
    double incl = dsum;
    for (int m = 1; m < 64; m <<= 1) { const double o = __shfl_up(incl, m, 64); if (lane >= m) incl += o; }
    double run = (incl - dsum) / total;
    const double uu = u[row];
    int pick = INT32_MAX;
    #pragma unroll
    for (int k = 0; k < CH; k++) {
        const int i = lane * CH + k;
        if (i < n_cols) {
            run += (double) pv[k] / total;
            const double cp = i == n_cols - 1 ? 1.0 : run;
            if (cp >= uu && i < pick) pick = i;
        }
    }
    for (int m = 1; m < 64; m <<= 1) pick = min(pick, __shfl_xor(pick, m, 64));
    if (lane == 0) out[(size_t) row * out_stride] = pick;
}
void launch_sample_rows_multinomial(hipStream_t s, const float * logits, int ld, int n_rows, int n_cols, float temp, const double * u,
                                    int32_t * out, int out_stride) {
    hipLaunchKernelGGL(sample_rows_multinomial_kernel, dim3((n_rows + 3) / 4), dim3(256), 0, s, logits, ld, n_rows, n_cols, temp, u, out, out_stride);
}

void launch_sample_greedy(hipStream_t s, const SampleArgs & a) {
    if (a.temp > 0.0f) hipLaunchKernelGGL(sample_multinomial_kernel, dim3(a.nbatch), dim3(1024), 0, s, a);
    else hipLaunchKernelGGL(sample_greedy_kernel, dim3(a.nbatch), dim3(1024), 0, s, a);
}

__global__ __launch_bounds__(256) void argmax_rows_kernel(const float * logits, int ld, int n_rows, int n_cols, int32_t * out,
                                                         int out_stride, StepState * st) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
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
    if (lane == 0) {
        out[(size_t) row * out_stride] = best;
        if (close > 1 && st) atomicAdd(&st->near_tie, 1);
    }
}
void launch_argmax_rows(hipStream_t s, const float * logits, int ld, int n_rows, int n_cols, int32_t * out, int out_stride,
                        StepState * st) {
    hipLaunchKernelGGL(argmax_rows_kernel, dim3((n_rows + 3) / 4), dim3(256), 0, s, logits, ld, n_rows, n_cols, out, out_stride, st);
}

void init_kernel_attributes() {
    (void) hipFuncSetAttribute(reinterpret_cast<const void *>(attn_rows_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                               32 * ATT_LD * (int) sizeof(float));
    (void) hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                               8 * GEMM_TN * GEMM_TM * (int) sizeof(float));
}

}  // namespace barkhip
