// codec_kernels.hip - EnCodec 24 kHz decoder kernels (RVQ de-embedding, SEANet conv / LSTM /
// transposed-conv stack).  Architecture restated from HF transformers modeling_encodec.py:82-450 (the
// model the reference's convert.py converts from; the reference delegates this stage to the
// un-vendored encodec.cpp, call site /root/reference/bark.cpp:2143-2167).
// Activations are TIME-MAJOR [row][C] (row = frame of the stage; the utterances of a batch back to back: utterance b owns rows
// [tm Tpre[b], tm Tpre[b + 1]) at a stage with upsampling factor tm), so that a convolution is a product on the f16 matrix cores: order
// C9m = the arithmetic of v_mfma_f32_32x32x16_f16 (oracle/mfma_f16_emu.h) over the axis kd = k * cin + ci (transposed conv: one product
// per output phase over tap * cin + ci).  Convolutions whose input channel count is not a multiple of 8 (toy models) keep order C9 (one fmaf
// chain in (ci, k) order) in a plain kernel; BARK_HIP_CROSSCHECK bit 10 (1024) sends every convolution there (oracle: set_codec_mfma(False)).
// Every kernel takes a CodecBatch: the codec of a lock-step batch costs the launches of ONE utterance.
#include "kernels.h"

namespace barkhip {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));

// utterance blockIdx.z of a batch: its frame count (times the stage's upsampling factor) and the element offsets of its input / output arrays
struct UttView { int T; size_t in_off, out_off; };
__device__ __forceinline__ UttView utt_view(const CodecBatch & cb, int T, int tmul_in, int tmul_out, int cin, int cout) {
    if (!cb.T) return UttView{T, 0, 0};
    const int z = blockIdx.z;
    return UttView{cb.T[z] * tmul_in, (size_t) cin * tmul_in * cb.Tpre[z], (size_t) cout * tmul_out * cb.Tpre[z]};
}

// first row of the utterance that owns global row `row` at upsampling factor tm, and that utterance's row count
__device__ __forceinline__ void utt_of_row(const CodecBatch & cb, int T_single, int tm, int row, int & row0, int & rows) {
    if (!cb.T) { row0 = 0; rows = T_single * tm; return; }
    int z = 0;
    while (z + 1 < cb.B && row >= cb.Tpre[z + 1] * tm) z++;
    row0 = cb.Tpre[z] * tm; rows = cb.T[z] * tm;
}

// z[frame][h] = 0 + sum_q embed_q[code] (modeling_encodec.py:440-448); codes of utterance b: [n_q][T[b]] at n_q Tpre[b]
__global__ void rvq_gather_kernel(const float * codebooks, int n_bins, int Hd, const int32_t * codes, int n_q, int T_, int rows_total, float * z, const CodecBatch cb) {
    const int row = blockIdx.x;
    if (row >= rows_total) return;
    int row0, rows;
    utt_of_row(cb, T_, 1, row, row0, rows);
    const int32_t * cz = codes + (size_t) n_q * row0;
    const int t = row - row0;
    for (int d = threadIdx.x; d < Hd; d += blockDim.x) {
        float v = 0.0f;
        for (int q = 0; q < n_q; q++) {
            int id = cz[(size_t) q * rows + t];
            id = min(max(id, 0), n_bins - 1);
            v = v + codebooks[((size_t) q * n_bins + id) * Hd + d];
        }
        z[(size_t) row * Hd + d] = v;
    }
}
void launch_rvq_gather(hipStream_t s, const float * codebooks, int n_bins, int Hd, const int32_t * codes, int n_q, int T, int rows_total, float * z, const CodecBatch & cb) {
    hipLaunchKernelGGL(rvq_gather_kernel, dim3(rows_total), dim3(128), 0, s, codebooks, n_bins, Hd, codes, n_q, T, rows_total, z, cb);
}

// see kernels.hip: keeps the compiler from fusing the producing multiply into the f16 conversion
__device__ __forceinline__ half_t to_half(float v) { asm("" : "+v"(v)); return (half_t) v; }
__device__ __forceinline__ float elu_canon(float x) { return x > 0.0f ? x : (float) expm1((double) x); }

__global__ void act_round_kernel(const float * x, size_t n, int elu, half_t * out) {
    for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t) gridDim.x * blockDim.x) {
        float v = x[i];
        if (elu) v = elu_canon(v);
        out[i] = to_half(v);
    }
}
void launch_act_round(hipStream_t s, const float * x, size_t n, int elu, half_t * out_h) {
    const int blocks = (int) ((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    hipLaunchKernelGGL(act_round_kernel, dim3(blocks), dim3(256), 0, s, x, n, elu, out_h);
}

// EncodecConv1d: causal, stride 1, pad_mode reflect (modeling_encodec.py:140-176): left pad K - 1; the reflect source of padded index
// i < left is x[left - i]; inputs shorter than the pad are zero-extended first.
// EncodecConvTranspose1d, causal, kernel 2 s: full output (T - 1) s + K trimmed by K - s on the right (modeling_encodec.py:206-233):
// output row q s + r takes frame q - 1 through kernel element r + s and frame q through element r.
// source row of tap kk for the output built from input row t of an utterance of `rows` rows; -1: the operand is zero
__device__ __forceinline__ int conv_src_row(int convT, int K, int kk, int t, int rows) {
    int j;
    if (convT) j = t - 1 + kk;                                    // tap 0: the previous frame, tap 1: this frame
    else { j = t + kk - (K - 1); j = j < 0 ? -j : j; }            // reflect on the left
    return (j >= 0 && j < rows) ? j : -1;
}

// C9m: one wave = 32 input rows x 32 output channels of one output phase; A = kernel image rows (co), B = activation rows (time), both
// 8 consecutive kd per lane = one 16-byte load; the accumulator walks kd in ascending blocks of 16 - the canonical chain.
typedef float floatx16c __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void conv_tm_mfma_kernel(const ConvTmArgs a) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, half = lane >> 5, l31 = lane & 31;
    const int row = (blockIdx.x * 4 + w) * 32 + l31;              // global input row of this lane's column
    const int co0 = blockIdx.y * 32, phase = blockIdx.z;
    const bool live = row < a.rows_in;
    int row0 = 0, rows = 0;
    if (live) utt_of_row(a.cb, a.T_single, a.tm_in, row, row0, rows);
    const int t = row - row0;
    const int nkb = a.kd16 >> 4;
    const half_t * wrow = a.W + ((size_t) phase * a.cout32 + co0 + l31) * a.kd16 + 8 * half;
    floatx16c acc;
    #pragma unroll
    for (int r = 0; r < 16; r++) acc[r] = 0.0f;
    auto load_b = [&](int kb) {
        const int kdd = 16 * kb + 8 * half;
        half8 bv;
        #pragma unroll
        for (int e = 0; e < 8; e++) bv[e] = (half_t) 0.0f;
        if (live && kdd < a.kd) {
            const int kk = kdd / a.cin, ci0 = kdd - kk * a.cin;
            const int j = conv_src_row(a.convT, a.K, kk, t, rows);
            if (j >= 0) bv = *reinterpret_cast<const half8 *>(a.xh + (size_t) (row0 + j) * a.cin + ci0);
        }
        return bv;
    };
    // four kd blocks per trip: their eight operand loads are in flight together
    int kb = 0;
    for (; kb + 4 <= nkb; kb += 4) {
        half8 av[4], bv[4];
        #pragma unroll
        for (int i = 0; i < 4; i++) { av[i] = *reinterpret_cast<const half8 *>(wrow + 16 * (kb + i)); bv[i] = load_b(kb + i); }
        #pragma unroll
        for (int i = 0; i < 4; i++) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(av[i], bv[i], acc, 0, 0, 0);
    }
    for (; kb < nkb; kb++) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(*reinterpret_cast<const half8 *>(wrow + 16 * kb), load_b(kb), acc, 0, 0, 0);
    if (!live) return;
    const size_t orow = a.convT ? (size_t) row * a.nphase + phase : (size_t) row;
    #pragma unroll
    for (int g = 0; g < 4; g++) {
        const int co = co0 + 8 * g + 4 * half;                     // accumulator registers 4 g .. 4 g + 3: channels co .. co + 3 of this lane's row
        #pragma unroll
        for (int e = 0; e < 4; e++) {
            if (co + e >= a.cout) break;
            float v = acc[4 * g + e] + a.bias[co + e];
            const size_t o = orow * a.cout + co + e;
            if (a.add) v = v + a.add[o];
            if (a.y) a.y[o] = v;
            if (a.yh_raw) a.yh_raw[o] = to_half(v);
            if (a.yh_elu) a.yh_elu[o] = to_half(elu_canon(v));
        }
    }
}

// C9: one fmaf chain per output in (ci, k) order, bias last (toy models' narrow convolutions; the cross-check route of C9m)
__global__ __launch_bounds__(256) void conv_tm_chain_kernel(const ConvTmArgs a) {
    const size_t idx = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
    const int phase = blockIdx.z;
    if (idx >= (size_t) a.rows_in * a.cout) return;
    const int row = (int) (idx / a.cout), co = (int) (idx % a.cout);
    int row0, rows;
    utt_of_row(a.cb, a.T_single, a.tm_in, row, row0, rows);
    const int t = row - row0;
    float acc = 0.0f;
    if (a.convT) {
        const int s = a.nphase, K = 2 * s;
        for (int ci = 0; ci < a.cin; ci++) {
            const float * wr = a.w32 + ((size_t) ci * a.cout + co) * K;
            if (t > 0) acc = fmaf(wr[phase + s], (float) a.xh[(size_t) (row - 1) * a.cin + ci], acc);
            acc = fmaf(wr[phase], (float) a.xh[(size_t) row * a.cin + ci], acc);
        }
    } else {
        for (int ci = 0; ci < a.cin; ci++) {
            const float * wr = a.w32 + ((size_t) co * a.cin + ci) * a.K;
            for (int k = 0; k < a.K; k++) {
                const int j = conv_src_row(0, a.K, k, t, rows);
                const float xv = j >= 0 ? (float) a.xh[(size_t) (row0 + j) * a.cin + ci] : 0.0f;
                acc = fmaf(wr[k], xv, acc);
            }
        }
    }
    float v = acc + a.bias[co];
    const size_t o = (a.convT ? (size_t) row * a.nphase + phase : (size_t) row) * a.cout + co;
    if (a.add) v = v + a.add[o];
    if (a.y) a.y[o] = v;
    if (a.yh_raw) a.yh_raw[o] = to_half(v);
    if (a.yh_elu) a.yh_elu[o] = to_half(elu_canon(v));
}

void launch_conv_tm(hipStream_t s, const ConvTmArgs & a) {
    if (a.convT && a.K != 2 * a.nphase) kernel_fail("bark-hip: transposed convolution needs kernel == 2 * stride (got %d, %d)", a.K, a.nphase);
    if (a.W && !(crosscheck_mask() & 1024)) {
        hipLaunchKernelGGL(conv_tm_mfma_kernel, dim3((a.rows_in + 127) / 128, a.cout32 / 32, a.nphase), dim3(256), 0, s, a);
        return;
    }
    const size_t n = (size_t) a.rows_in * a.cout;
    hipLaunchKernelGGL(conv_tm_chain_kernel, dim3((unsigned) ((n + 255) / 256), 1, a.nphase), dim3(256), 0, s, a);
}

// one unit (d) of one layer at one step: the four gates in the four 16-lane groups of the wave, C1 dots (lane c of a group = chain c: its
// 8-element chunk of every 128-block, ascending; tree over the 16 lanes), gate non-linearities in double precision rounded once (C9), state
// update by lane 0.  Every operand chunk of the step's dots is requested before the first multiply-add (NBLK = D / 128 is a template
// parameter: with the run-time loop each 128-block was a dependent round trip to L2, eight of them for a unit of the second layer).
template <int NBLK>
__device__ __forceinline__ void lstm_load(half8 (&v)[NBLK], const half_t * row) {
    #pragma unroll
    for (int b = 0; b < NBLK; b++) v[b] = *reinterpret_cast<const half8 *>(row + (b << 7));
}
template <int NBLK>
__device__ __forceinline__ float lstm_dot(const half8 (&w)[NBLK], const half8 (&h)[NBLK]) {
    float acc = 0.0f;
    #pragma unroll
    for (int b = 0; b < NBLK; b++)
        #pragma unroll
        for (int e = 0; e < 8; e++) acc = fmaf((float) w[b][e], (float) h[b][e], acc);
    acc = acc + __shfl_xor(acc, 1, 64); acc = acc + __shfl_xor(acc, 2, 64);
    acc = acc + __shfl_xor(acc, 4, 64); acc = acc + __shfl_xor(acc, 8, 64);
    return acc;
}
template <int NBLK>
__global__ __launch_bounds__(256) void lstm_pair_step_kernel(const LstmPairArgs a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 15, g = lane >> 4;
    constexpr int D = NBLK * 128, nb1 = D / 4;
    const int i = a.t_base ? a.t_base[0] + a.t : a.t;              // launch index: layer 1 at step i, layer 2 at step i - 1
    // utterance blockIdx.z of a batch: T frames, its rows of gi1 / h1 / h2 / out2 start at row0, its cells at D z
    const int z = blockIdx.z;
    const int T = a.cb.T ? a.cb.T[z] : (a.t_base ? a.t_base[1] : a.T);
    const size_t row0 = a.cb.T ? (size_t) a.cb.Tpre[z] : 0;
    const bool second = (int) blockIdx.x >= nb1;
    const int d = ((int) blockIdx.x - (second ? nb1 : 0)) * 4 + wave;
    const int t = second ? i - 1 : i;
    if (t < 0 || t >= T) return;
    const size_t row = (size_t) (g * D + d);
    const half_t * h1 = a.h1 + row0 * D, * h2 = a.h2 + row0 * D;
    float * cs = (second ? a.c2 : a.c1) + (size_t) z * D;
    const float cprev = t ? cs[d] : 0.0f;                          // requested with the operands, not behind the gates
    float gi, gh = 0.0f, bi, bh;
    if (!second) {
        half8 w[NBLK], h[NBLK];
        if (t) { lstm_load<NBLK>(w, a.w_hh1 + row * D + (c << 3)); lstm_load<NBLK>(h, h1 + (size_t) (t - 1) * D + (c << 3)); }
        gi = a.gi1[(row0 + t) * 4 * D + row]; bi = a.b_ih1[row]; bh = a.b_hh1[row];
        if (t) gh = lstm_dot<NBLK>(w, h);
    } else {
        half8 wi[NBLK], hi[NBLK], wh[NBLK], hh[NBLK];
        lstm_load<NBLK>(wi, a.w_ih2 + row * D + (c << 3)); lstm_load<NBLK>(hi, h1 + (size_t) t * D + (c << 3));
        if (t) { lstm_load<NBLK>(wh, a.w_hh2 + row * D + (c << 3)); lstm_load<NBLK>(hh, h2 + (size_t) (t - 1) * D + (c << 3)); }
        bi = a.b_ih2[row]; bh = a.b_hh2[row];
        gi = lstm_dot<NBLK>(wi, hi);                                // W_ih2 . f16(h1_t)
        if (t) gh = lstm_dot<NBLK>(wh, hh);
    }
    const float pre = (gi + bi) + (gh + bh);                   // (gi + b_ih) + (gh + b_hh)
    const float act = g == 2 ? (float) tanh((double) pre) : 1.0f / (1.0f + (float) exp((double) (-pre)));
    const float i_t = __shfl(act, 0, 64), f_t = __shfl(act, 16, 64), g_t = __shfl(act, 32, 64), o_t = __shfl(act, 48, 64);
    if (lane == 0) {
        const float cn = f_t * cprev + i_t * g_t;
        const float hn = o_t * (float) tanh((double) cn);
        cs[d] = cn;
        (second ? a.h2 : a.h1)[(row0 + t) * D + d] = to_half(hn);
        if (second) a.out2[(row0 + t) * D + d] = hn;                // time-major, as every activation of the codec
    }
}
void launch_lstm_pair_step(hipStream_t s, const LstmPairArgs & a) {
    const dim3 grid(2 * (a.D / 4), 1, a.cb.B), block(256);
    switch (a.D >> 7) {
        case 1: hipLaunchKernelGGL((lstm_pair_step_kernel<1>), grid, block, 0, s, a); break;
        case 2: hipLaunchKernelGGL((lstm_pair_step_kernel<2>), grid, block, 0, s, a); break;
        case 4: hipLaunchKernelGGL((lstm_pair_step_kernel<4>), grid, block, 0, s, a); break;
        case 8: hipLaunchKernelGGL((lstm_pair_step_kernel<8>), grid, block, 0, s, a); break;
        default: kernel_fail("bark-hip: the codec's LSTM width must be 128, 256, 512 or 1024 (got %d)", a.D);
    }
}

__global__ void add_int_kernel(int * p, int v) { *p += v; }      // p[0]: step base of the replayed LSTM block
void launch_add_int(hipStream_t s, int * p, int v) { hipLaunchKernelGGL(add_int_kernel, dim3(1), dim3(1), 0, s, p, v); }
__global__ void add_kernel(const float * a, const float * b, size_t n, float * out) {
    for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t) gridDim.x * blockDim.x) out[i] = a[i] + b[i];
}
void launch_add(hipStream_t s, const float * a, const float * b, size_t n, float * out) {
    const int blocks = (int) ((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    hipLaunchKernelGGL(add_kernel, dim3(blocks), dim3(256), 0, s, a, b, n, out);
}

}  // namespace barkhip
