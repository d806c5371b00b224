// codec_kernels.hip - EnCodec 24 kHz decoder kernels (RVQ de-embedding, SEANet conv / LSTM /
// transposed-conv stack).  Architecture restated from HF transformers modeling_encodec.py:82-450 (the
// model the reference's convert.py converts from; the reference delegates this stage to the
// un-vendored encodec.cpp, call site /root/reference/bark.cpp:2143-2167).
// Round-1 kernels are exact-order direct convolutions (one output element per thread, one fmaf chain
// in (ci, k) order); they are bit-compatible with an f32 MFMA formulation, which is the planned
// optimisation.
#include "kernels.h"

namespace barkhip {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));

__global__ void rvq_gather_kernel(const float * codebooks, int n_bins, int Hd, const int32_t * codes, int n_q, int T, float * z) {
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
void launch_rvq_gather(hipStream_t s, const float * codebooks, int n_bins, int Hd, const int32_t * codes, int n_q, int T, float * z) {
    hipLaunchKernelGGL(rvq_gather_kernel, dim3((T + 127) / 128, Hd), dim3(128), 0, s, codebooks, n_bins, Hd, codes, n_q, T, z);
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

// EncodecConv1d, causal, stride 1, pad_mode reflect (modeling_encodec.py:140-176): left pad K-1.
// Reflect source of padded index i < left is x[left - i]; inputs shorter than the pad are zero-extended first.
__global__ __launch_bounds__(256) void conv1d_kernel(const half_t * w, const float * bias, int cout, int cin, int K, const half_t * xh,
                                                    int T, const float * add, float * y) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int co = blockIdx.y;
    if (t >= T) return;
    const half_t * wr = w + (size_t) co * cin * K;
    float acc = 0.0f;
    for (int ci = 0; ci < cin; ci++) {
        const half_t * xr = xh + (size_t) ci * T;
        for (int k = 0; k < K; k++) {
            int j = t + k - (K - 1);
            j = j < 0 ? -j : j;
            const float xv = j < T ? (float) xr[j] : 0.0f;
            acc = fmaf((float) wr[ci * K + k], xv, acc);
        }
    }
    acc = acc + bias[co];
    if (add) acc = acc + add[(size_t) co * T + t];               // shortcut(x) + block(x)  (modeling_encodec.py:276-282)
    y[(size_t) co * T + t] = acc;
}
void launch_conv1d(hipStream_t s, const half_t * w, const float * bias, int cout, int cin, int K, const half_t * xh, int T,
                   const float * add, float * y) {
    hipLaunchKernelGGL(conv1d_kernel, dim3((T + 255) / 256, cout), dim3(256), 0, s, w, bias, cout, cin, K, xh, T, add, y);
}

// EncodecConvTranspose1d, causal: full output (T-1)*s + K, trimmed by K - s on the right (modeling_encodec.py:206-233)
__global__ __launch_bounds__(256) void convtr1d_kernel(const half_t * w, const float * bias, int cin, int cout, int K, int stride,
                                                      const half_t * xh, int T, float * y) {
    const int Tout = T * stride;
    const int to = blockIdx.x * blockDim.x + threadIdx.x;
    const int co = blockIdx.y;
    if (to >= Tout) return;
    int t_lo = to - (K - 1);                                      // smallest t with to - t*s <= K-1
    t_lo = t_lo <= 0 ? 0 : (t_lo + stride - 1) / stride;
    const int t_hi = min(to / stride, T - 1);
    float acc = 0.0f;
    for (int ci = 0; ci < cin; ci++) {
        const half_t * xr = xh + (size_t) ci * T;
        const half_t * wr = w + ((size_t) ci * cout + co) * K;
        for (int t = t_lo; t <= t_hi; t++) acc = fmaf((float) wr[to - t * stride], (float) xr[t], acc);
    }
    y[(size_t) co * Tout + to] = acc + bias[co];
}
void launch_convtr1d(hipStream_t s, const half_t * w, const float * bias, int cin, int cout, int K, int stride, const half_t * xh,
                     int T, float * y) {
    hipLaunchKernelGGL(convtr1d_kernel, dim3((T * stride + 255) / 256, cout), dim3(256), 0, s, w, bias, cin, cout, K, stride, xh, T, y);
}

__global__ void transpose_round_kernel(const float * x, int C, int T, half_t * xt) {
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
void launch_transpose_round(hipStream_t s, const float * x, int C, int T, half_t * xt) {
    hipLaunchKernelGGL(transpose_round_kernel, dim3((T + 31) / 32, (C + 31) / 32), dim3(32, 8), 0, s, x, C, T, xt);
}

// One wave per hidden unit d: its four gate rows (i,f,g,o) x 16 chain lanes (order C1 over K = D).
__global__ __launch_bounds__(256) void lstm_step_kernel(const LstmStepArgs a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 15, g = lane >> 4;
    const int d = blockIdx.x * 4 + wave;
    const int D = a.D, nblk = D >> 7;
    if (d >= D) return;
    float acc = 0.0f;
    if (a.hprev_h) {
        const half_t * wrow = a.w_hh + (size_t) (g * D + d) * D + (c << 3);
        const half_t * hrow = a.hprev_h + (c << 3);
        for (int b = 0; b < nblk; b++) {
            const half8 wv = *reinterpret_cast<const half8 *>(wrow + (b << 7));
            const half8 hv = *reinterpret_cast<const half8 *>(hrow + (b << 7));
            #pragma unroll
            for (int e = 0; e < 8; e++) acc = fmaf((float) wv[e], (float) hv[e], acc);
        }
    }
    acc = acc + __shfl_xor(acc, 1, 64); acc = acc + __shfl_xor(acc, 2, 64);
    acc = acc + __shfl_xor(acc, 4, 64); acc = acc + __shfl_xor(acc, 8, 64);
    // gate pre-activation (gi + b_ih) + (gh + b_hh), evaluated by lane 0 of each 16-lane group
    const float pre = (a.gi[g * D + d] + a.b_ih[g * D + d]) + (acc + a.b_hh[g * D + d]);
    const float pi = __shfl(pre, 0, 64), pf = __shfl(pre, 16, 64), pg = __shfl(pre, 32, 64), po = __shfl(pre, 48, 64);
    if (lane == 0) {
        const float i_t = 1.0f / (1.0f + (float) exp((double) (-pi)));
        const float f_t = 1.0f / (1.0f + (float) exp((double) (-pf)));
        const float g_t = (float) tanh((double) pg);
        const float o_t = 1.0f / (1.0f + (float) exp((double) (-po)));
        const float cn = f_t * a.c[d] + i_t * g_t;
        const float hn = o_t * (float) tanh((double) cn);
        a.c[d] = cn;
        a.hout_h[d] = to_half(hn);
        a.hseq[(size_t) d * a.T + a.t] = hn;
    }
}
void launch_lstm_step(hipStream_t s, const LstmStepArgs & a) {
    hipLaunchKernelGGL(lstm_step_kernel, dim3((a.D + 3) / 4), dim3(256), 0, s, a);
}

__global__ void add_kernel(const float * a, const float * b, size_t n, float * out) {
    for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t) gridDim.x * blockDim.x) out[i] = a[i] + b[i];
}
void launch_add(hipStream_t s, const float * a, const float * b, size_t n, float * out) {
    const int blocks = (int) ((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    hipLaunchKernelGGL(add_kernel, dim3(blocks), dim3(256), 0, s, a, b, n, out);
}

}  // namespace barkhip
