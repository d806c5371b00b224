// mfma_rate_probe.hip - issue rate and dependent-accumulator latency of the f32-input MFMA forms on gfx950, and the register layout
// of the multi-block forms (16x16x1 in 4 blocks, 32x32x1 in 2 blocks).  The lock-step decode products (kernels.hip) walk C1 chains
// as dependent MFMAs on one accumulator, so what matters is cycles per MFMA as a function of the number of independent accumulators
// a wave alternates between, and of the waves sharing a SIMD.
//   build: hipcc --offload-arch=gfx950 -O2 -o /tmp/mfma_rate_probe tools/probes/mfma_rate_probe.hip ; run: /tmp/mfma_rate_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx32 __attribute__((ext_vector_type(32)));

#define OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int INSTR> struct Acc;
template <> struct Acc<0> { typedef floatx16 T; static constexpr int N = 16; };   // 32x32x2
template <> struct Acc<1> { typedef floatx4 T;  static constexpr int N = 4; };    // 16x16x4
template <> struct Acc<2> { typedef floatx4 T;  static constexpr int N = 4; };    // 4x4x1 (16 blocks)
template <> struct Acc<3> { typedef floatx16 T; static constexpr int N = 16; };   // 16x16x1 (4 blocks)
template <> struct Acc<4> { typedef floatx32 T; static constexpr int N = 32; };   // 32x32x1 (2 blocks)

template <int INSTR> __device__ __forceinline__ typename Acc<INSTR>::T mfma(float a, float b, typename Acc<INSTR>::T c) {
    if constexpr (INSTR == 0) return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
    else if constexpr (INSTR == 1) return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
    else if constexpr (INSTR == 2) return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0);
    else if constexpr (INSTR == 3) return __builtin_amdgcn_mfma_f32_16x16x1f32(a, b, c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_32x32x1f32(a, b, c, 0, 0, 0);
}

// NACC independent accumulators, ITER rounds: every round issues one MFMA per accumulator (dependent on that accumulator's previous one)
template <int INSTR, int NACC>
__global__ void rate_kernel(const float * in, float * out, long long * cycles, int iters) {
    typedef typename Acc<INSTR>::T T;
    T acc[NACC];
    #pragma unroll
    for (int i = 0; i < NACC; i++)
        #pragma unroll
        for (int r = 0; r < Acc<INSTR>::N; r++) acc[i][r] = 0.0f;
    const float a = in[threadIdx.x], b = in[64 + threadIdx.x];
    __syncthreads();
    const long long t0 = clock64();
    for (int it = 0; it < iters; it++) {
        #pragma unroll
        for (int u = 0; u < 8; u++) {
            #pragma unroll
            for (int i = 0; i < NACC; i++) acc[i] = mfma<INSTR>(a, b, acc[i]);
        }
    }
    float s = 0.0f;
    #pragma unroll
    for (int i = 0; i < NACC; i++)
        #pragma unroll
        for (int r = 0; r < Acc<INSTR>::N; r++) s += acc[i][r];
    const long long t1 = clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cycles[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int INSTR, int NACC>
static double run_rate(int threads, int blocks, const float * d_in, float * d_out, long long * d_cyc) {
    const int iters = 256;
    hipLaunchKernelGGL((rate_kernel<INSTR, NACC>), dim3(blocks), dim3(threads), 0, 0, d_in, d_out, d_cyc, iters);
    OK(hipDeviceSynchronize());
    hipLaunchKernelGGL((rate_kernel<INSTR, NACC>), dim3(blocks), dim3(threads), 0, 0, d_in, d_out, d_cyc, iters);
    OK(hipDeviceSynchronize());
    const int nw = blocks * threads / 64;
    std::vector<long long> c((size_t) nw);
    OK(hipMemcpy(c.data(), d_cyc, sizeof(long long) * nw, hipMemcpyDeviceToHost));
    long long mx = 0;
    for (long long v : c) mx = v > mx ? v : mx;
    return (double) mx / ((double) iters * 8 * NACC);       // cycles (s_memtime ticks) per MFMA of one wave
}

template <int INSTR>
static void rates(const char * name, double fma_per_instr, const float * d_in, float * d_out, long long * d_cyc) {
    // waves per SIMD: 64 threads = one wave alone; 256 = one wave per SIMD; 512 / 1024 = 2 / 4 waves per SIMD
    const int th[4] = {64, 256, 512, 1024};
    for (int t = 0; t < 4; t++) {
        const double c1 = run_rate<INSTR, 1>(th[t], 1, d_in, d_out, d_cyc), c2 = run_rate<INSTR, 2>(th[t], 1, d_in, d_out, d_cyc);
        const double c4 = run_rate<INSTR, 4>(th[t], 1, d_in, d_out, d_cyc);
        double c8 = 0.0;
        if constexpr (Acc<INSTR>::N <= 16) c8 = run_rate<INSTR, 8>(th[t], 1, d_in, d_out, d_cyc);
        const double waves_per_simd = th[t] <= 256 ? 1.0 : th[t] / 256.0;
        const double best = c8 > 0.0 ? fmin(fmin(c1, c2), fmin(c4, c8)) : fmin(fmin(c1, c2), c4);
        printf("%-12s threads %4d: cycles per MFMA per wave with 1/2/4/8 accumulators %7.1f %7.1f %7.1f %7.1f | best: %.1f fma/clk/SIMD\n", name, th[t], c1, c2, c4, c8,
               fma_per_instr * waves_per_simd / best);
    }
}

// layout of the multi-block forms: A = 1 + n / 128 (n = 16 b + i resp. 32 b + i), B = 2^j: the product names (block, i, j)
template <int INSTR>
__global__ void layout_kernel(float * out) {
    typedef typename Acc<INSTR>::T T;
    T acc;
    #pragma unroll
    for (int r = 0; r < Acc<INSTR>::N; r++) acc[r] = 0.0f;
    const int l = threadIdx.x;
    constexpr int R = INSTR == 3 ? 16 : 32;                  // rows / cols per block
    const float a = 1.0f + (float) l / 128.0f;               // lane l -> n = l (hypothesis: lane = R b + i)
    const float b = ldexpf(1.0f, l % R);
    acc = mfma<INSTR>(a, b, acc);
    #pragma unroll
    for (int r = 0; r < Acc<INSTR>::N; r++) out[r * 64 + l] = acc[r];
}
template <int INSTR>
static void layout(const char * name, float * d_out) {
    constexpr int R = INSTR == 3 ? 16 : 32, NB = INSTR == 3 ? 4 : 2, NR = Acc<INSTR>::N;
    hipLaunchKernelGGL((layout_kernel<INSTR>), dim3(1), dim3(64), 0, 0, d_out);
    OK(hipDeviceSynchronize());
    std::vector<float> o((size_t) NR * 64);
    OK(hipMemcpy(o.data(), d_out, o.size() * 4, hipMemcpyDeviceToHost));
    // hypothesis: register r of lane l holds block b = r / (NR / NB), row i = (r % 4) + 8 ((r % (NR / NB)) / 4) ... for R = 32; (r % 4) + 4 (l / 16) for R = 16; col j = l % R
    int bad = 0;
    for (int r = 0; r < NR; r++)
        for (int l = 0; l < 64; l++) {
            const float v = o[(size_t) r * 64 + l];
            int e; const float m = frexpf(v, &e);             // v = m 2^e, m in [0.5, 1): v = (1 + n/128) 2^j -> j = e - 1, n = (2 m - 1) 128
            const int j = e - 1, n = (int) lrintf((2.0f * m - 1.0f) * 128.0f);
            const int b = n / R, i = n % R;
            int hb, hi, hj;
            if (R == 16) { hb = r / 4; hi = (r % 4) + 4 * (l / 16); hj = l % 16; }
            else { hb = r / 16; hi = (r % 4) + 8 * ((r % 16) / 4) + 4 * (l / 32); hj = l % 32; }
            if (b != hb || i != hi || j != hj) { if (bad < 6) printf("  %s: reg %d lane %d holds (block %d, row %d, col %d), hypothesis (%d, %d, %d)\n", name, r, l, b, i, j, hb, hi, hj); bad++; }
        }
    printf("%-12s layout: %s (A lane = %d b + i, B lane = %d b + j; D reg = %d b + (i %% 4)%s, lane = %s)\n", name, bad ? "DIFFERENT" : "as assumed", R, R, NR / NB,
           R == 16 ? "" : " + 4 (i / 8)", R == 16 ? "16 (i / 4) + j" : "32 ((i / 4) % 2) + j");
}

int main() {
    float * d_in, * d_out; long long * d_cyc;
    OK(hipMalloc(&d_in, 128 * 4)); OK(hipMalloc(&d_out, 32 * 1024 * 4)); OK(hipMalloc(&d_cyc, 64 * 8));
    std::vector<float> in(128);
    for (int i = 0; i < 128; i++) in[i] = 1.0f + (float) (i % 7) * 0.125f;
    OK(hipMemcpy(d_in, in.data(), 128 * 4, hipMemcpyHostToDevice));
    rates<0>("32x32x2", 2048, d_in, d_out, d_cyc);
    rates<1>("16x16x4", 1024, d_in, d_out, d_cyc);
    rates<2>("4x4x1_16b", 256, d_in, d_out, d_cyc);
    rates<3>("16x16x1_4b", 1024, d_in, d_out, d_cyc);
    rates<4>("32x32x1_2b", 2048, d_in, d_out, d_cyc);
    layout<3>("16x16x1_4b", d_out);
    layout<4>("32x32x1_2b", d_out);
    return 0;
}
