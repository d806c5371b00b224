// device_utils.h - device-side helpers shared by the GPT kernel files: wave64 DPP reductions in the canonical tree order,
// the explicit f16 rounding point, ggml's GELU table lookup, KV-cache addressing, embedding-row reads and the fused
// epilogues of the linear operators (bias / residual / GELU / KV append / logits).  See DESIGN.md section 3 for the numerics.
#pragma once
#include "kernels.h"

namespace barkhip {


typedef float  floatx16 __attribute__((ext_vector_type(16)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));

#define DEVINL __device__ __forceinline__

#ifdef BARK_TRACE
// time stamps that cannot be scheduled before their operand exists (the asm consumes it)
DEVINL unsigned long long trace_clock() { unsigned long long t; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory"); return t; }
DEVINL unsigned long long trace_clock_s(int dep) { unsigned long long t; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) : "s"(dep) : "memory"); return t; }
DEVINL unsigned long long trace_clock_v(float dep) { unsigned long long t; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) : "v"(dep) : "memory"); return t; }
DEVINL void trace_emit(const TraceSink & tr, unsigned long long t0, unsigned long long t1, unsigned long long t2, unsigned long long t3,
                       unsigned long long ta = 0, unsigned long long tb = 0) {
    if (!tr.rec || (threadIdx.x & 63) != 0) return;
    const unsigned wpb = (blockDim.x + 63) >> 6;
    const unsigned i = *tr.pos * tr.per_replay + tr.base + (blockIdx.y * gridDim.x + blockIdx.x) * wpb + (threadIdx.x >> 6);
    if (i >= tr.cap) return;
    const unsigned xcc = __builtin_amdgcn_s_getreg(20 | (31 << 11)) & 15u;          // HW_REG_XCC_ID
    unsigned long long * r = tr.rec + (size_t) i * 8;
    r[0] = (unsigned long long) tr.kid | ((unsigned long long) xcc << 16);
    r[1] = (unsigned long long) blockIdx.x | ((unsigned long long) blockIdx.y << 24) | ((unsigned long long) (threadIdx.x >> 6) << 40);
    r[2] = t0; r[3] = t1; r[4] = t2; r[5] = t3; r[6] = ta; r[7] = tb;
}
#define TRACE_T0() const unsigned long long _tr0 = trace_clock()
#define TRACE_T1(dep) const unsigned long long _tr1 = trace_clock_s(dep)
#define TRACE_T2(dep) const unsigned long long _tr2 = trace_clock_v(dep)
#define TRACE_TA(dep) const unsigned long long _tra = trace_clock_v(dep)
#define TRACE_TB(dep) const unsigned long long _trb = trace_clock_v(dep)
#define TRACE_SET(var, dep) var = trace_clock_v(dep)
#define TRACE_END(tr) trace_emit(tr, _tr0, _tr1, _tr2, trace_clock())
#define TRACE_END_AB(tr) trace_emit(tr, _tr0, _tr1, _tr2, trace_clock(), _tra, _trb)
#else
#define TRACE_T0()
#define TRACE_T1(dep)
#define TRACE_T2(dep)
#define TRACE_TA(dep)
#define TRACE_TB(dep)
#define TRACE_SET(var, dep)
#define TRACE_END(tr)
#define TRACE_END_AB(tr)
#endif

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
// whole-wave integer / float reductions by DPP (every lane ends with the result of its 16-lane row; rows are combined through SGPRs)
template <int CTRL> DEVINL int dpp_i32(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, false); }
DEVINL int wave_min_i32(int v) {
    v = min(v, dpp_i32<DPP_XOR1>(v)); v = min(v, dpp_i32<DPP_XOR2>(v)); v = min(v, dpp_i32<DPP_HALF_MIRROR>(v)); v = min(v, dpp_i32<DPP_MIRROR>(v));
    return min(min(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)), min(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}
DEVINL int wave_add_i32(int v) {
    v += dpp_i32<DPP_XOR1>(v); v += dpp_i32<DPP_XOR2>(v); v += dpp_i32<DPP_HALF_MIRROR>(v); v += dpp_i32<DPP_MIRROR>(v);
    return (__builtin_amdgcn_readlane(v, 0) + __builtin_amdgcn_readlane(v, 16)) + (__builtin_amdgcn_readlane(v, 32) + __builtin_amdgcn_readlane(v, 48));
}
DEVINL float wave_add_f32(float v) {
    v = wave_xor_add16(v);
    return (readlane_f32(v, 0) + readlane_f32(v, 16)) + (readlane_f32(v, 32) + readlane_f32(v, 48));
}
// C4e: THE exponential of every attention softmax: one stated f32 routine, bit-reproducible on a CPU and on CDNA4 because
// it is made of correctly rounded single operations only (fmaf, one multiply, one subtraction, integer bit moves; the file is compiled with
// -ffp-contract=off).  Algorithm and constants: the vector expf of ARM's optimised routines, which ggml's f32 soft_max carries as ggml_v_expf
// (bark.cpp:1322,1513 -> ggml_soft_max_inplace): n = round(x log2 e) by the 1.5 x 2^23 shift, b = x - n ln2 in two steps (hi / lo), 2^n from the
// exponent bits, degree-5 polynomial in b; 1.45 + 0.5 ulp.  The argument is s - max <= 0; below n = -125 (x < -86.3) the result is defined as +0
// (no subnormal intermediate ever forms, so the denormal mode of either machine cannot matter).  Rounds 1 - 5 used (float) exp((double) x): 1360
// fp64 instructions per wave of attn_window_kernel that could not overlap the matrix cores.
DEVINL float canon_expf(float x) {
    const float r = 0x1.8p23f;
    const float z = fmaf(x, 0x1.715476p+0f, r);
    const float n = z - r;
    float b = fmaf(-n, 0x1.62e4p-1f, x);
    b = fmaf(-n, 0x1.7f7d1cp-20f, b);
    const float k = __builtin_bit_cast(float, (__builtin_bit_cast(unsigned, z) << 23) + 0x3f800000u);      // 2^n
    const float u = b * b;
    const float j = fmaf(fmaf(fmaf(0x1.0e4020p-7f, b, 0x1.573e2ep-5f), u, fmaf(0x1.555e66p-3f, b, 0x1.fffdb6p-2f)), u, 0x1.ffffecp-1f * b);
    return n < -125.0f ? 0.0f : fmaf(k, j, k);
}

// a / K in double, correctly rounded, for a compile-time row length K (LayerNorm mean and variance, ggml_norm divides the double sums by
// the float count).  Powers of two scale exactly; otherwise (K = 768) one Newton step on a correctly rounded reciprocal: q0 = RN(a y),
// r = a - K q0 exactly (fma), q = RN(q0 + r y) is the correctly rounded quotient (Markstein) - three dependent fp64 operations instead of
// the ~11 of the generic division sequence.  Checked against exact rational arithmetic in tests/test_canon_orders.py.
template <int K> DEVINL double div_by_const(double a) {
    if constexpr ((K & (K - 1)) == 0) return a * (1.0 / (double) K);
    else {
        constexpr double y = 1.0 / (double) K;                 // compile-time, correctly rounded
        const double q0 = a * y;
        const double r = __builtin_fma(-(double) K, q0, a);
        return __builtin_fma(r, y, q0);
    }
}
// order-preserving float <-> unsigned map, so that the row maximum can be kept with an integer atomicMax
DEVINL unsigned f32_ordered(float f) { const unsigned b = __builtin_bit_cast(unsigned, f); return (b & 0x80000000u) ? ~b : (b | 0x80000000u); }
DEVINL float f32_unordered(unsigned u) { const unsigned b = (u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u; return __builtin_bit_cast(float, b); }
DEVINL half8 ld_half8(const half_t * p) { return *reinterpret_cast<const half8 *>(p); }
// weight rows a decode step reads exactly once: non-temporal (A/B build -DBARK_NT_WEIGHTS; plain otherwise)
#ifdef BARK_NT_WEIGHTS
DEVINL half8 ld_half8_w(const half_t * p) { return __builtin_nontemporal_load(reinterpret_cast<const half8 *>(p)); }
#else
DEVINL half8 ld_half8_w(const half_t * p) { return *reinterpret_cast<const half8 *>(p); }
#endif
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

// XCD-aware workgroup ids.  Workgroups are dispatched round-robin over the 8 XCDs (workgroup b runs on XCD b % 8; observed, used for
// speed only) and every XCD has its own 4 MB L2: with the hardware order, the tiles that share an operand (a row panel of x, the K / V of
// one head) are spread over all eight L2s and every one of them fetches everything through the fabric - measured on the many-row
// products and the prefill attention as a common ceiling of ~3.8 TB/s of operand traffic.  xcd_rank() renumbers the workgroups so that
// one XCD holds CONSECUTIVE ranks (bijective for any grid size); panel_tile() walks the output tiles in column panels of width pw, so
// that a run of consecutive ranks covers a compact rows x pw rectangle (few distinct operand slices per XCD at any moment).
DEVINL int xcd_rank(int b, int n) {
    const int q = n >> 3, r = n & 7, x = b & 7, i = b >> 3;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
}
DEVINL void panel_tile(int l, int nrow, int ncol, int pw, int & row, int & col) {
    const int per_panel = nrow * pw, p = l / per_panel, rem = l - p * per_panel;
    const int c0 = p * pw, w = min(pw, ncol - c0);
    row = rem / w; col = c0 + (rem - row * w);
}

// K cache element address: [H][16][P][4] floats; V cache: [H][P][64]
DEVINL size_t kc_index(int h, int d, int pos, int P) { return (((size_t) h * 16 + (d >> 2)) * P + pos) * 4 + (d & 3); }
DEVINL size_t vc_index(int h, int d, int pos, int P) { return ((size_t) h * P + pos) * 64 + d; }

// Buffer loads: address = wave-uniform base (in a 128-bit descriptor held in SGPRs) + 32-bit lane offset + scalar offset, all in bytes.
// A stream of loads at constant strides then needs ONE address VGPR and one scalar move per load; the flat form the compiler picks
// for `base[lane_part + i * stride]` spends a 64-bit vector add (+ wait state) per load, ~250 issue slots for a 64-load stream.
// (The clang builtin __builtin_amdgcn_raw_buffer_load_b128 of this ROCm release is lowered to a ONE-dword load - probe
// tools/probes/bufload_probe.hip - so the LLVM intrinsics are bound by name, the way composable_kernel does.)
typedef int   int4v   __attribute__((ext_vector_type(4)));
typedef float float4v __attribute__((ext_vector_type(4)));
__device__ float4v llvm_amdgcn_raw_buffer_load_v4f32(int4v rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.v4f32");
__device__ float   llvm_amdgcn_raw_buffer_load_f32(int4v rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.f32");
struct BufRsrc { int4v d; };
DEVINL BufRsrc buf_rsrc(const void * uniform_base) {
    const unsigned long long a = (unsigned long long) uniform_base;
    BufRsrc r;
    r.d.x = (int) (unsigned) a; r.d.y = (int) (unsigned) (a >> 32);     // 48-bit base, stride 0
    r.d.z = -1;                                                          // num_records: no bound in practice
    r.d.w = 0x00020000;                                                  // raw buffer, 32-bit data format (gfx9 descriptor word 3)
    return r;
}
DEVINL float buf_ld_f32(const BufRsrc & r, unsigned lane_bytes, unsigned scalar_bytes) {
    return llvm_amdgcn_raw_buffer_load_f32(r.d, (int) lane_bytes, (int) scalar_bytes, 0);
}
DEVINL float4 buf_ld_f4(const BufRsrc & r, unsigned lane_bytes, unsigned scalar_bytes) {
    const float4v u = llvm_amdgcn_raw_buffer_load_v4f32(r.d, (int) lane_bytes, (int) scalar_bytes, 0);
    return float4{u.x, u.y, u.z, u.w};
}

// C2: one 16-d block of an attention score: kq = the block's four d-quads of the key, qb = the block's 16 q values; one fmaf chain
DEVINL float score_block_f4(const float4 * kq, const float * qb) {
    float acc = 0.0f;
    #pragma unroll
    for (int i = 0; i < 4; i++) {
        acc = fmaf(kq[i].x, qb[4 * i + 0], acc);
        acc = fmaf(kq[i].y, qb[4 * i + 1], acc);
        acc = fmaf(kq[i].z, qb[4 * i + 2], acc);
        acc = fmaf(kq[i].w, qb[4 * i + 3], acc);
    }
    return acc;
}

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
            if (m < 2 * E) { a.kc[slot + kc_index(h, d, pos, a.P)] = v; if (a.knew) a.knew[mm] = v; }
            else { a.vc[slot + vc_index(h, d, pos, a.P)] = v; if (a.vt) a.vt[kc_index(h, d, pos, a.P)] = v; }
            break;
        }
        case EPI_RESID: a.res[(size_t) n * a.M + m] = v + p.res; break;                          // cur + inpL (bark.cpp:1352,1388)
        case EPI_GELU: {
            const half_t g = gelu_lut_apply(v, a.lut);
            if (a.out_h32) a.out_h32[(size_t) n * a.M + m] = (float) g; else a.out_h[(size_t) n * a.M + m] = g;
            break;
        }
        default:        a.out[(size_t) n * a.ld_out + m] = a.out_div != 0.0f ? v / a.out_div : v; break;
    }
}
DEVINL void linear_epilogue(const LinArgs & a, int n, int m, float dot, int row_off) {
    float v = dot;
    if (a.bias) v = v + a.bias[row_off + m];
    switch (a.epi) {
        case EPI_QKV: {
            const int E = a.E;
            if (m < E) { a.q[(size_t) n * E + m] = v; break; }
            const int zq = a.seq ? n / a.seq : 0;
            int pos = a.pos0 + (a.st ? a.st->n_past : 0) + (n - zq * a.seq);
            size_t zoff = (size_t) zq * a.kv_slot_stride;
            if (a.seqtab) {
                const SeqTab t = a.seqtab[zq];
                if (n - zq * a.seq >= t.len) break;
                pos = t.pos0 + (n - zq * a.seq); zoff = (size_t) t.slot * a.kv_slot_stride;
            }
            const int mm = m < 2 * E ? m - E : m - 2 * E;
            const int h = mm >> 6, d = mm & 63;
            if (m < 2 * E) a.kc[zoff + kc_index(h, d, pos, a.P)] = v; else { a.vc[zoff + vc_index(h, d, pos, a.P)] = v; if (a.vt) a.vt[zoff + kc_index(h, d, pos, a.P)] = v; }
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


}  // namespace barkhip
