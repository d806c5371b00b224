// mfma_dep_probe.hip - why does a run of dependent v_mfma_f32_16x16x1_4b_f32 in the lock-step product kernel take ~70 cycles per issue
// (SQ_WAIT_INST_ANY) when the rate probe measures 32?  Variants of one wave issuing runs of 8 dependent MFMAs:
//   acc in VGPRs / AGPRs (inline asm), operands constant / eight distinct register pairs / pairs produced by a burst of v_cvt_f32_f16 in front of the run
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef _Float16 half_t;
#define OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <bool AGPR> __device__ __forceinline__ void mfma(floatx16 & acc, float a, float b) {
    if constexpr (AGPR) asm volatile("v_mfma_f32_16x16x1_4b_f32 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
    else                asm volatile("v_mfma_f32_16x16x1_4b_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}
// MODE 0: constant operands; 1: eight distinct operand pairs (registers); 2: operands converted from packed f16 in front of every run
template <bool AGPR, int MODE>
__global__ void k(const unsigned * in, float * out, long long * cycles, int iters) {
    floatx16 acc;
    #pragma unroll
    for (int i = 0; i < 16; i++) acc[i] = 0.0f;
    unsigned wp[4], xp[4];
    #pragma unroll
    for (int i = 0; i < 4; i++) { wp[i] = in[threadIdx.x * 4 + i]; xp[i] = in[256 + threadIdx.x * 4 + i]; }
    float wf[8], xf[8];
    #pragma unroll
    for (int e = 0; e < 8; e++) { wf[e] = 1.0f + (float) (wp[e >> 1] & 7) * 0.125f + e; xf[e] = 1.0f + (float) (xp[e >> 1] & 3) * 0.25f; }
    __syncthreads();
    const long long t0 = clock64();
    for (int it = 0; it < iters; it++) {
        if constexpr (MODE == 2) {
            #pragma unroll
            for (int e = 0; e < 8; e++) {
                wf[e] = (float) __builtin_bit_cast(half_t, (unsigned short) ((e & 1) ? (wp[e >> 1] >> 16) : wp[e >> 1]));
                xf[e] = (float) __builtin_bit_cast(half_t, (unsigned short) ((e & 1) ? (xp[e >> 1] >> 16) : xp[e >> 1]));
            }
            #pragma unroll
            for (int i = 0; i < 4; i++) { wp[i] += 0x00010001u * (it & 1); xp[i] ^= (unsigned) it & 0x00010001u; }     // new bits every round
        }
        __builtin_amdgcn_sched_barrier(0);
        #pragma unroll
        for (int e = 0; e < 8; e++) mfma<AGPR>(acc, MODE == 0 ? wf[0] : wf[e], MODE == 0 ? xf[0] : xf[e]);
        __builtin_amdgcn_sched_barrier(0);
    }
    float s = 0.0f;
    #pragma unroll
    for (int i = 0; i < 16; i++) s += acc[i];
    const long long t1 = clock64();
    out[threadIdx.x] = s;
    if (threadIdx.x == 0) cycles[0] = t1 - t0;
}
template <bool AGPR, int MODE> static double run(int threads, const unsigned * d_in, float * d_out, long long * d_cyc) {
    const int iters = 512;
    for (int r = 0; r < 2; r++) { hipLaunchKernelGGL((k<AGPR, MODE>), dim3(1), dim3(threads), 0, 0, d_in, d_out, d_cyc, iters); OK(hipDeviceSynchronize()); }
    long long c; OK(hipMemcpy(&c, d_cyc, 8, hipMemcpyDeviceToHost));
    return (double) c / (iters * 8.0);
}
int main() {
    unsigned * d_in; float * d_out; long long * d_cyc;
    OK(hipMalloc(&d_in, 4096 * 4)); OK(hipMalloc(&d_out, 1024 * 4)); OK(hipMalloc(&d_cyc, 8));
    std::vector<unsigned> in(4096);
    for (int i = 0; i < 4096; i++) in[i] = 0x3C003C00u + (unsigned) (i % 13) * 0x00100010u;
    OK(hipMemcpy(d_in, in.data(), 4096 * 4, hipMemcpyHostToDevice));
    for (int th : {64, 256}) {
        printf("threads %3d cycles per MFMA: VGPR acc  const %.1f  distinct %.1f  cvt-burst %.1f | AGPR acc  const %.1f  distinct %.1f  cvt-burst %.1f\n", th,
               run<false, 0>(th, d_in, d_out, d_cyc), run<false, 1>(th, d_in, d_out, d_cyc), run<false, 2>(th, d_in, d_out, d_cyc),
               run<true, 0>(th, d_in, d_out, d_cyc), run<true, 1>(th, d_in, d_out, d_cyc), run<true, 2>(th, d_in, d_out, d_cyc));
    }
    return 0;
}
