// pull_probe.hip - what does it cost every CU to pull (a) a block of bytes ALL workgroups read (the f16 activation rows of a lock-step
// decode product: 32 slots x K, L2 resident) and (b) its private share of a weight matrix (HBM, rotating over 12 matrices so that
// nothing stays in L2 / Infinity Cache longer than in the engine)?  One 256-thread workgroup per CU, every load is a 16-byte load,
// all of a thread's loads are requested before the first is consumed.  Times are per launch inside a 48-node hipGraph (kernel
// boundaries included, as in tools/time_slots.py).
//   build: hipcc --offload-arch=gfx950 -O2 -o /tmp/pull_probe tools/probes/pull_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// every workgroup reads the same `shared_kb` KB (NS 16-byte loads per thread) and its own `priv` bytes (NP loads per thread)
template <int NS, int NP>
__global__ __launch_bounds__(256) void pull_kernel(const uint4 * __restrict__ shared, const uint4 * __restrict__ priv, float * out) {
    const int tid = threadIdx.x;
    uint4 s[NS > 0 ? NS : 1], p[NP > 0 ? NP : 1];
    const uint4 * pp = priv + (size_t) blockIdx.x * NP * 256;
    #pragma unroll
    for (int i = 0; i < NP; i++) p[i] = pp[i * 256 + tid];
    #pragma unroll
    for (int i = 0; i < NS; i++) s[i] = shared[i * 256 + tid];
    unsigned acc = 0;
    #pragma unroll
    for (int i = 0; i < NP; i++) acc += p[i].x ^ p[i].y ^ p[i].z ^ p[i].w;
    #pragma unroll
    for (int i = 0; i < NS; i++) acc += s[i].x ^ s[i].y ^ s[i].z ^ s[i].w;
    if (acc == 0x12345678u) out[blockIdx.x * 256 + tid] = 1.0f;      // never true for the data used; keeps the loads alive
}

template <int NS, int NP>
static double time_it(int blocks, const uint4 * shared, const uint4 * const * privs, float * out, hipStream_t st) {
    hipGraph_t g; hipGraphExec_t e;
    OK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < 48; i++) hipLaunchKernelGGL((pull_kernel<NS, NP>), dim3(blocks), dim3(256), 0, st, shared, privs[i % 12], out);
    OK(hipStreamEndCapture(st, &g));
    OK(hipGraphInstantiate(&e, g, nullptr, nullptr, 0));
    OK(hipGraphLaunch(e, st)); OK(hipStreamSynchronize(st));
    hipEvent_t e0, e1; OK(hipEventCreate(&e0)); OK(hipEventCreate(&e1));
    OK(hipEventRecord(e0, st));
    for (int r = 0; r < 10; r++) OK(hipGraphLaunch(e, st));
    OK(hipEventRecord(e1, st)); OK(hipEventSynchronize(e1));
    float ms; OK(hipEventElapsedTime(&ms, e0, e1));
    OK(hipGraphExecDestroy(e)); OK(hipGraphDestroy(g));
    return ms * 1000.0 / 480.0;
}

int main() {
    hipStream_t st; OK(hipStreamCreate(&st));
    const size_t priv_bytes = (size_t) 768 * 32 * 256 * 16;          // up to 32 loads per thread for 768 workgroups
    std::vector<const uint4 *> privs;
    for (int i = 0; i < 12; i++) { uint4 * p; OK(hipMalloc(&p, priv_bytes)); OK(hipMemset(p, 1 + i, priv_bytes)); privs.push_back(p); }
    uint4 * shared; OK(hipMalloc(&shared, 48 * 256 * 16)); OK(hipMemset(shared, 3, 48 * 256 * 16));
    float * out; OK(hipMalloc(&out, 768 * 256 * 4));
    const uint4 ** d = privs.data();
    printf("us per launch (256 threads per workgroup; shared KB read by every workgroup / private KB per workgroup)\n");
    printf("empty kernel, 256 wgs                       %6.2f\n", time_it<0, 0>(256, shared, d, out, st));
    printf("shared 12 KB, 256 wgs                       %6.2f\n", time_it<3, 0>(256, shared, d, out, st));
    printf("shared 48 KB, 256 wgs                       %6.2f\n", time_it<12, 0>(256, shared, d, out, st));
    printf("shared 96 KB, 256 wgs                       %6.2f\n", time_it<24, 0>(256, shared, d, out, st));
    printf("shared 192 KB, 256 wgs                      %6.2f\n", time_it<48, 0>(256, shared, d, out, st));
    printf("private 16 KB, 256 wgs (4.2 MB)             %6.2f\n", time_it<0, 4>(256, shared, d, out, st));
    printf("private 24 KB, 192 wgs (4.7 MB)             %6.2f\n", time_it<0, 6>(192, shared, d, out, st));
    printf("private 96 KB, 48 wgs (4.7 MB)              %6.2f\n", time_it<0, 24>(48, shared, d, out, st));
    printf("private 6 KB, 768 wgs (4.7 MB)              %6.2f\n", time_it<0, 2>(768, shared, d, out, st) );
    printf("shared 48 KB + private 24 KB, 192 wgs       %6.2f\n", time_it<12, 6>(192, shared, d, out, st));
    printf("shared 48 KB + private 6 KB, 768 wgs        %6.2f\n", time_it<12, 2>(768, shared, d, out, st));   // NP = 2 is 8 KB here, close enough
    printf("shared 192 KB + private 24 KB, 192 wgs      %6.2f\n", time_it<48, 6>(192, shared, d, out, st));
    printf("shared 48 KB + private 96 KB, 48 wgs        %6.2f\n", time_it<12, 24>(48, shared, d, out, st));
    return 0;
}
