// mfma_chip_probe.hip - what do ALL CUs sustain on f32-input MFMAs with non-trivial data?  The exact GEMMs (kernels.hip) measure ~50 % of
// the 157 TFLOP/s peak in two unrelated formulations; is that the kernels or the chip (clock under matrix-core load)?
//   grid = 256 x wgs_per_cu workgroups of 512 threads (2 waves per SIMD), every wave issues `iters` x 64 MFMAs on 16 independent accumulator
//   pairs (the pattern of gemm16_kernel) with operands from registers (variant 0) or re-read from LDS every 4 MFMAs (variant 1)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
#define OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int VAR>
__global__ __launch_bounds__(512) void k16(const float * in, float * out, int iters) {
    __shared__ float2 lds[4096];
    const int tid = threadIdx.x;
    for (int i = tid; i < 4096; i += 512) lds[i] = float2{in[i & 1023], in[(i + 7) & 1023]};
    __syncthreads();
    floatx4 acc[16][2];
    for (int c = 0; c < 16; c++) for (int t = 0; t < 2; t++) acc[c][t] = floatx4{0.f, 0.f, 0.f, 0.f};
    float2 a0 = lds[tid], a1 = lds[tid + 512], bw = lds[tid + 1024];
    for (int it = 0; it < iters; it++) {
        #pragma unroll
        for (int c = 0; c < 16; c++) {
            if (VAR == 1) { a0 = lds[(tid + 64 * c + it) & 4095]; a1 = lds[(tid + 64 * c + 512 + it) & 4095]; bw = lds[(tid + 64 * c + 1024 + it) & 4095]; }
            acc[c][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.x, bw.x, acc[c][0], 0, 0, 0);
            acc[c][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.x, bw.x, acc[c][1], 0, 0, 0);
            acc[c][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.y, bw.y, acc[c][0], 0, 0, 0);
            acc[c][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.y, bw.y, acc[c][1], 0, 0, 0);
        }
    }
    floatx4 s = {0.f, 0.f, 0.f, 0.f};
    for (int c = 0; c < 16; c++) for (int t = 0; t < 2; t++) s = s + acc[c][t];
    out[(size_t) blockIdx.x * 512 + tid] = s[0] + s[1] + s[2] + s[3];
}
__global__ __launch_bounds__(512) void k32(const float * in, float * out, int iters) {
    const int tid = threadIdx.x;
    floatx16 acc[8];
    for (int c = 0; c < 8; c++) for (int r = 0; r < 16; r++) acc[c][r] = 0.f;
    const float a = in[tid & 1023], b = in[(tid + 77) & 1023];
    for (int it = 0; it < iters; it++) {
        #pragma unroll
        for (int u = 0; u < 4; u++)
            #pragma unroll
            for (int c = 0; c < 8; c++) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
    }
    float s = 0.f;
    for (int c = 0; c < 8; c++) for (int r = 0; r < 16; r++) s += acc[c][r];
    out[(size_t) blockIdx.x * 512 + tid] = s;
}
template <typename F> static double timeit(F launch) {
    launch(); OK(hipDeviceSynchronize());
    hipEvent_t e0, e1; OK(hipEventCreate(&e0)); OK(hipEventCreate(&e1));
    OK(hipEventRecord(e0)); for (int i = 0; i < 5; i++) launch(); OK(hipEventRecord(e1)); OK(hipEventSynchronize(e1));
    float ms; OK(hipEventElapsedTime(&ms, e0, e1)); return ms * 1000.0 / 5;
}
int main() {
    float * in, * out; OK(hipMalloc(&in, 4096)); OK(hipMalloc(&out, (size_t) 1024 * 512 * 4));
    std::vector<float> h(1024); srand(1); for (auto & v : h) v = (float) (rand() % 2000 - 1000) / 977.0f;
    OK(hipMemcpy(in, h.data(), 4096, hipMemcpyHostToDevice));
    for (int wgs : {256, 768}) {
        const int iters = 64;                                   // 64 x 64 MFMAs of 1024 MACs per wave
        const double flop16 = (double) wgs * 8 * iters * 64 * 1024 * 2, flop32 = (double) wgs * 8 * iters * 32 * 2048 * 2;
        const double t0 = timeit([&] { hipLaunchKernelGGL(k16<0>, dim3(wgs), dim3(512), 0, 0, in, out, iters); });
        const double t1 = timeit([&] { hipLaunchKernelGGL(k16<1>, dim3(wgs), dim3(512), 0, 0, in, out, iters); });
        const double t2 = timeit([&] { hipLaunchKernelGGL(k32, dim3(wgs), dim3(512), 0, 0, in, out, iters); });
        printf("%4d workgroups x 8 waves: 16x16x4 from registers %.1f us = %.1f TFLOP/s | 16x16x4 with LDS operand reads %.1f us = %.1f TFLOP/s | 32x32x2 %.1f us = %.1f TFLOP/s\n",
               wgs, t0, flop16 / t0 / 1e6, t1, flop16 / t1 / 1e6, t2, flop32 / t2 / 1e6);
    }
    return 0;
}
