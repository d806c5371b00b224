// codec_kernels.hip - EnCodec 24 kHz decoder kernels (RVQ de-embedding, SEANet conv / LSTM /
// transposed-conv stack).  Architecture restated from HF transformers modeling_encodec.py:82-450 (the
// model the reference's convert.py converts from; the reference delegates this stage to the
// un-vendored encodec.cpp, call site /root/reference/bark.cpp:2143-2167).
// Convolutions are exact-order direct kernels (register-blocked, one fmaf chain per output in (ci, k) order).
// Every kernel takes a CodecBatch: several utterances of different lengths run in one launch (grid.z = utterance), each on its own
// compact [C][T] arrays laid back to back - the codec of a lock-step batch costs the launches of ONE utterance.
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

__global__ void rvq_gather_kernel(const float * codebooks, int n_bins, int Hd, const int32_t * codes, int n_q, int T_, float * z, const CodecBatch cb) {
    const UttView u = utt_view(cb, T_, 1, 1, n_q, Hd);
    const int T = u.T;
    codes += u.in_off; z += u.out_off;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int d = blockIdx.y;
    if (t >= T) return;
    float v = 0.0f;                                              // quantized_out = 0 + sum_q embed_q[code] (modeling_encodec.py:440-448)
    for (int q = 0; q < n_q; q++) {
        int id = codes[(size_t) q * T + t];
        id = min(max(id, 0), n_bins - 1);
        v = v + codebooks[((size_t) q * n_bins + id) * Hd + d];
    }
    z[(size_t) d * T + t] = v;
}
void launch_rvq_gather(hipStream_t s, const float * codebooks, int n_bins, int Hd, const int32_t * codes, int n_q, int T, float * z, const CodecBatch & cb) {
    hipLaunchKernelGGL(rvq_gather_kernel, dim3((T + 127) / 128, Hd, cb.B), dim3(128), 0, s, codebooks, n_bins, Hd, codes, n_q, T, z, cb);
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
// EncodecConvTranspose1d, causal: full output (T - 1) s + K, trimmed by K - s on the right (modeling_encodec.py:206-233).
// Register-blocked kernels, one fmaf chain per output element: ci ascending, k ascending, bias last.
// Weights are kept as f32 copies of the f16 file values (exact) so that a wave-uniform weight becomes a scalar load and an
// SGPR operand of v_fma_f32; every x value loaded is reused by CO x (taps that touch it) multiply-adds.
template <int CO, int TT, int K>
__global__ __launch_bounds__(256) void conv1d_blocked_kernel(const float * __restrict__ w, const float * __restrict__ bias, int cout, int cin,
                                                            const half_t * __restrict__ xh, int T_, const float * add, float * y, const CodecBatch cb, int tmul) {
    const UttView u = utt_view(cb, T_, tmul, tmul, cin, cout);
    const int T = u.T;
    xh += u.in_off; y += u.out_off; if (add) add += u.out_off;
    const int t0 = (blockIdx.x * blockDim.x + threadIdx.x) * TT;
    const int co0 = blockIdx.y * CO;
    if (t0 >= T) return;
    float acc[CO][TT];
    #pragma unroll
    for (int c = 0; c < CO; c++)
        #pragma unroll
        for (int j = 0; j < TT; j++) acc[c][j] = 0.0f;
    const bool interior = t0 >= K - 1 && t0 + TT <= T;
    for (int ci = 0; ci < cin; ci++) {
        const half_t * xr = xh + (size_t) ci * T;
        float xv[TT + K - 1];                                   // inputs t0-(K-1) .. t0+TT-1, reflect-padded on the left
        if (interior) {
            #pragma unroll
            for (int i = 0; i < TT + K - 1; i++) xv[i] = (float) xr[t0 - (K - 1) + i];
        } else {
            #pragma unroll
            for (int i = 0; i < TT + K - 1; i++) {
                int j = t0 - (K - 1) + i;
                j = j < 0 ? -j : j;
                xv[i] = j < T ? (float) xr[j] : 0.0f;
            }
        }
        #pragma unroll
        for (int c = 0; c < CO; c++) {
            const float * wr = w + ((size_t) min(co0 + c, cout - 1) * cin + ci) * K;     // wave-uniform: scalar loads
            #pragma unroll
            for (int k = 0; k < K; k++) {
                const float wk = wr[k];
                #pragma unroll
                for (int j = 0; j < TT; j++) acc[c][j] = fmaf(wk, xv[j + k], acc[c][j]);
            }
        }
    }
    #pragma unroll
    for (int c = 0; c < CO; c++) {
        const int co = co0 + c;
        if (co >= cout) break;
        #pragma unroll
        for (int j = 0; j < TT; j++) {
            const int t = t0 + j;
            if (t >= T) break;
            float v = acc[c][j] + bias[co];
            if (add) v = v + add[(size_t) co * T + t];
            y[(size_t) co * T + t] = v;
        }
    }
}

// transposed conv with K == 2 * stride: output to = t*s + kk takes x[t-1] (tap kk+s) then x[t] (tap kk).
// Thread = time step t; the block owns CO output channels x KB phases kk, whose weights are wave-uniform.
template <int CO, int KB>
__global__ __launch_bounds__(256) void convtr1d_blocked_kernel(const float * __restrict__ w, const float * __restrict__ bias, int cin, int cout,
                                                              int stride, const half_t * __restrict__ xh, int T_, float * y, const CodecBatch cb, int tmul) {
    const UttView u = utt_view(cb, T_, tmul, tmul * stride, cin, cout);
    const int T = u.T;
    xh += u.in_off; y += u.out_off;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int nkb = stride / KB;                                  // phase groups per channel group (stride % KB == 0)
    const int co0 = (blockIdx.y / nkb) * CO, kk0 = (blockIdx.y % nkb) * KB;
    if (t >= T) return;
    const int K = 2 * stride;
    float acc[CO][KB];
    #pragma unroll
    for (int c = 0; c < CO; c++)
        #pragma unroll
        for (int q = 0; q < KB; q++) acc[c][q] = 0.0f;
    for (int ci = 0; ci < cin; ci++) {
        const half_t * xr = xh + (size_t) ci * T;
        const float xp = t > 0 ? (float) xr[t - 1] : 0.0f, xc = (float) xr[t];
        #pragma unroll
        for (int c = 0; c < CO; c++) {
            const float * wr = w + ((size_t) ci * cout + min(co0 + c, cout - 1)) * K + kk0;      // wave-uniform
            #pragma unroll
            for (int q = 0; q < KB; q++) {
                if (t > 0) acc[c][q] = fmaf(wr[q + stride], xp, acc[c][q]);
                acc[c][q] = fmaf(wr[q], xc, acc[c][q]);
            }
        }
    }
    const int Tout = T * stride;
    #pragma unroll
    for (int c = 0; c < CO; c++) {
        const int co = co0 + c;
        if (co >= cout) break;
        #pragma unroll
        for (int q = 0; q < KB; q++) y[(size_t) co * Tout + (size_t) t * stride + kk0 + q] = acc[c][q] + bias[co];
    }
}

// T: frames of the (longest) utterance at this stage = cb.Tmax * tmul for a batch
void launch_conv1d_f32w(hipStream_t s, const float * w, const float * bias, int cout, int cin, int K, const half_t * xh, int T,
                        const float * add, float * y, const CodecBatch & cb, int tmul) {
    constexpr int TT = 4;
    const int co_grp = cout >= 4 ? 4 : 1;
    dim3 grid((T + 256 * TT - 1) / (256 * TT), (cout + co_grp - 1) / co_grp, cb.B), block(256);
#define LAUNCH_CONV(CO, KK) hipLaunchKernelGGL((conv1d_blocked_kernel<CO, TT, KK>), grid, block, 0, s, w, bias, cout, cin, xh, T, add, y, cb, tmul)
    if (co_grp == 4) { if (K == 7) LAUNCH_CONV(4, 7); else if (K == 3) LAUNCH_CONV(4, 3); else if (K == 1) LAUNCH_CONV(4, 1); else kernel_fail("bark-hip: unsupported convolution kernel size %d", K); }
    else             { if (K == 7) LAUNCH_CONV(1, 7); else if (K == 3) LAUNCH_CONV(1, 3); else if (K == 1) LAUNCH_CONV(1, 1); else kernel_fail("bark-hip: unsupported convolution kernel size %d", K); }
#undef LAUNCH_CONV
}
bool conv1d_f32w_supported(int K) { return K == 7 || K == 3 || K == 1; }

void launch_convtr1d_f32w(hipStream_t s, const float * w, const float * bias, int cin, int cout, int K, int stride, const half_t * xh, int T, float * y,
                          const CodecBatch & cb, int tmul) {
    if (K != 2 * stride) kernel_fail("bark-hip: transposed convolution needs kernel == 2 * stride (got %d, %d)", K, stride);
    const int KB = stride % 4 == 0 ? 4 : (stride % 2 == 0 ? 2 : 1);
    const int CO = cout >= 2 ? 2 : 1;
    dim3 grid((T + 255) / 256, ((cout + CO - 1) / CO) * (stride / KB), cb.B), block(256);
    if (CO == 2) {
        if (KB == 4) hipLaunchKernelGGL((convtr1d_blocked_kernel<2, 4>), grid, block, 0, s, w, bias, cin, cout, stride, xh, T, y, cb, tmul);
        else if (KB == 2) hipLaunchKernelGGL((convtr1d_blocked_kernel<2, 2>), grid, block, 0, s, w, bias, cin, cout, stride, xh, T, y, cb, tmul);
        else hipLaunchKernelGGL((convtr1d_blocked_kernel<2, 1>), grid, block, 0, s, w, bias, cin, cout, stride, xh, T, y, cb, tmul);
    } else {
        if (KB == 4) hipLaunchKernelGGL((convtr1d_blocked_kernel<1, 4>), grid, block, 0, s, w, bias, cin, cout, stride, xh, T, y, cb, tmul);
        else if (KB == 2) hipLaunchKernelGGL((convtr1d_blocked_kernel<1, 2>), grid, block, 0, s, w, bias, cin, cout, stride, xh, T, y, cb, tmul);
        else hipLaunchKernelGGL((convtr1d_blocked_kernel<1, 1>), grid, block, 0, s, w, bias, cin, cout, stride, xh, T, y, cb, tmul);
    }
}

__global__ void transpose_round_kernel(const float * x, int C, int T_, half_t * xt, const CodecBatch cb) {
    const UttView u = utt_view(cb, T_, 1, 1, C, C);
    const int T = u.T;
    x += u.in_off; xt += u.out_off;                             // [C][T] -> [T][C]: the utterances' rows stay back to back
    __shared__ float tile[32][33];
    const int c0 = blockIdx.y * 32, t0 = blockIdx.x * 32;
    for (int r = threadIdx.y; r < 32; r += blockDim.y) {
        const int c = c0 + r, t = t0 + threadIdx.x;
        tile[r][threadIdx.x] = (c < C && t < T) ? x[(size_t) c * T + t] : 0.0f;
    }
    __syncthreads();
    for (int r = threadIdx.y; r < 32; r += blockDim.y) {
        const int t = t0 + r, c = c0 + threadIdx.x;
        if (t < T && c < C) xt[(size_t) t * C + c] = to_half(tile[threadIdx.x][r]);
    }
}
void launch_transpose_round(hipStream_t s, const float * x, int C, int T, half_t * xt, const CodecBatch & cb) {
    hipLaunchKernelGGL(transpose_round_kernel, dim3((T + 31) / 32, (C + 31) / 32, cb.B), dim3(32, 8), 0, s, x, C, T, xt, cb);
}

// one unit (d) of one layer at one step: the four gates in the four 16-lane groups of the wave, C1 dots, gate non-linearities in
// double precision rounded once (C9), state update by lane 0
__device__ __forceinline__ float lstm_dot(const half_t * wrow, const half_t * hrow, int nblk) {
    float acc = 0.0f;
    for (int b = 0; b < nblk; b++) {
        const half8 wv = *reinterpret_cast<const half8 *>(wrow + (b << 7));
        const half8 hv = *reinterpret_cast<const half8 *>(hrow + (b << 7));
        #pragma unroll
        for (int e = 0; e < 8; e++) acc = fmaf((float) wv[e], (float) hv[e], acc);
    }
    acc = acc + __shfl_xor(acc, 1, 64); acc = acc + __shfl_xor(acc, 2, 64);
    acc = acc + __shfl_xor(acc, 4, 64); acc = acc + __shfl_xor(acc, 8, 64);
    return acc;
}
__global__ __launch_bounds__(256) void lstm_pair_step_kernel(const LstmPairArgs a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 15, g = lane >> 4;
    const int D = a.D, nblk = D >> 7, nb1 = (D + 3) >> 2;
    const int i = a.t_base ? a.t_base[0] + a.t : a.t;              // launch index: layer 1 at step i, layer 2 at step i - 1
    // utterance blockIdx.z of a batch: T frames, its rows of gi1 / h1 / h2 start at row0, its [D][T] output at D row0, its cells at D z
    const int z = blockIdx.z;
    const int T = a.cb.T ? a.cb.T[z] : (a.t_base ? a.t_base[1] : a.T);
    const size_t row0 = a.cb.T ? (size_t) a.cb.Tpre[z] : 0;
    const bool second = (int) blockIdx.x >= nb1;
    const int d = ((int) blockIdx.x - (second ? nb1 : 0)) * 4 + wave;
    const int t = second ? i - 1 : i;
    if (d >= D || t < 0 || t >= T) return;
    const size_t row = (size_t) (g * D + d);
    const half_t * h1 = a.h1 + row0 * D, * h2 = a.h2 + row0 * D;
    float gi, gh = 0.0f, bi, bh;
    if (!second) {
        gi = a.gi1[(row0 + t) * 4 * D + row]; bi = a.b_ih1[row]; bh = a.b_hh1[row];
        if (t) gh = lstm_dot(a.w_hh1 + row * D + (c << 3), h1 + (size_t) (t - 1) * D + (c << 3), nblk);
    } else {
        bi = a.b_ih2[row]; bh = a.b_hh2[row];
        gi = lstm_dot(a.w_ih2 + row * D + (c << 3), h1 + (size_t) t * D + (c << 3), nblk);          // W_ih2 . f16(h1_t)
        if (t) gh = lstm_dot(a.w_hh2 + row * D + (c << 3), h2 + (size_t) (t - 1) * D + (c << 3), nblk);
    }
    const float pre = (gi + bi) + (gh + bh);                   // (gi + b_ih) + (gh + b_hh)
    const float act = g == 2 ? (float) tanh((double) pre) : 1.0f / (1.0f + (float) exp((double) (-pre)));
    const float i_t = __shfl(act, 0, 64), f_t = __shfl(act, 16, 64), g_t = __shfl(act, 32, 64), o_t = __shfl(act, 48, 64);
    if (lane == 0) {
        float * cs = (second ? a.c2 : a.c1) + (size_t) z * D;
        const float cprev = t ? cs[d] : 0.0f;
        const float cn = f_t * cprev + i_t * g_t;
        const float hn = o_t * (float) tanh((double) cn);
        cs[d] = cn;
        (second ? a.h2 : a.h1)[(row0 + t) * D + d] = to_half(hn);
        if (second) a.out2[row0 * D + (size_t) d * T + t] = hn;
    }
}
void launch_lstm_pair_step(hipStream_t s, const LstmPairArgs & a) {
    hipLaunchKernelGGL(lstm_pair_step_kernel, dim3(2 * ((a.D + 3) / 4), 1, a.cb.B), dim3(256), 0, s, a);
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
