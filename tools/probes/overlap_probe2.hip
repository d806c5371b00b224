// overlap_probe2.hip - second attempt at hiding the kernel boundaries of a chain of dependent decode-sized GEMVs.
//
// overlap_probe.hip (two streams + a counter per kernel, agent-scope release / acquire FENCES in every workgroup) measured 26.7 us per
// kernel against 2.99 us for plain dependent launches.  The hand-off recipe of cdna_hip_programming.md (guideline 16, form R2) needs no
// fence at all when the payload is small: every output is ONE 8-byte granule {tag = epoch, value} written with a relaxed agent-scope
// atomic store (write-through) and read with relaxed agent-scope atomic loads until the tag matches - the data is the flag.  A decode
// GEMV consumes a 768-float vector (3 KB of values = 6 KB of granules), so here:
//   * the consumer requests its weight rows first (they do not depend on the producer),
//   * its wave 0 sweeps the 768 granules of the input vector (12 per lane) until every tag equals the kernel's epoch (BOUNDED: gives up
//     after ~20 ms and raises an error flag - the probe cannot hang), then publishes the values in LDS for the other three waves,
//   * every output is stored as a granule tagged with the next kernel's epoch.
// Variants, each a hipGraph of 62 kernels, results compared bit for bit:
//   A  plain kernels, plain vectors, one stream                         (the engine's structure; 2.99 us per kernel in probe 1)
//   C  granule kernels, one stream                                      (what the granule I/O itself costs)
//   B  granule kernels, even kernels on stream 0 and odd ones on stream 1 (kernel i+1 is resident while kernel i runs)
// Build: hipcc --offload-arch=gfx950 -O2 -o tools/probes/overlap_probe2 tools/probes/overlap_probe2.hip ; run: timeout 30 tools/probes/overlap_probe2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef unsigned long long u64;

constexpr int K = 768, M = 3072, NK = 62, ROWS_PER_WG = 16, N_WG = M / ROWS_PER_WG;

// GRAN = false: x is a plain float vector, y is written plainly.
// GRAN = true : xg / yg are granule vectors; wait for tag == epoch on the input, tag the output with epoch + 1 (first == 1: input plain).
template <bool GRAN>
__global__ __launch_bounds__(256) void gemv_chain_kernel(const _Float16 * __restrict__ W, const float * xin, float * xout,
                                                         u64 * xg, u64 * yg, unsigned epoch, int first, unsigned * err) {
    __shared__ float xs[K];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 15, rg = lane >> 4;
    const int m = blockIdx.x * ROWS_PER_WG + wave * 4 + rg;
    const _Float16 * wrow = W + (size_t) m * K + (c << 3);
    half8 wv[K / 128];
    #pragma unroll
    for (int b = 0; b < K / 128; b++) wv[b] = *reinterpret_cast<const half8 *>(wrow + (b << 7));
    if (GRAN && !first) {
        if (wave == 0) {
            unsigned v[K / 64];
            unsigned spins = 0;
            for (;;) {
                bool ok = true;
                #pragma unroll
                for (int k = 0; k < K / 64; k++) {
                    const u64 g = __hip_atomic_load(xg + lane + 64 * k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    v[k] = (unsigned) g; ok &= (unsigned) (g >> 32) == epoch;
                }
                if (__all(ok)) break;
                __builtin_amdgcn_s_sleep(1);
                if (++spins > 400000u) { if (lane == 0) atomicOr(err, 1u); break; }        // ~20 ms: give up, never hang
            }
            #pragma unroll
            for (int k = 0; k < K / 64; k++) xs[lane + 64 * k] = __builtin_bit_cast(float, v[k]);
        }
    } else {
        for (int k = tid; k < K; k += 256) xs[k] = xin[k];
    }
    __syncthreads();
    float acc = 0.0f;
    #pragma unroll
    for (int b = 0; b < K / 128; b++) {
        #pragma unroll
        for (int e = 0; e < 8; e++) acc = fmaf((float) wv[b][e], xs[(b << 7) + (c << 3) + e], acc);
    }
    #pragma unroll
    for (int off = 1; off < 16; off <<= 1) acc += __shfl_xor(acc, off);
    if (c == 0) {
        const float y = acc * 0.03125f;
        if (GRAN) __hip_atomic_store(yg + m, ((u64) (epoch + 1) << 32) | __builtin_bit_cast(unsigned, y), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else xout[m] = y;
    }
}

int main() {
    const int n_layers = 12;
    std::vector<_Float16> hw((size_t) n_layers * M * K);
    unsigned seed = 1u;
    for (auto & v : hw) { seed = seed * 1664525u + 1013904223u; v = (_Float16) (((int) (seed >> 20) % 2001 - 1000) * (1.0f / 1000.0f)); }
    std::vector<float> hx(M);
    for (int i = 0; i < M; i++) hx[i] = (float) ((i * 37) % 101 - 50) * 0.01f;
    _Float16 * dW; float * dx[2]; u64 * dg[3]; unsigned * err;
    OK(hipMalloc(&dW, hw.size() * 2)); OK(hipMalloc(&dx[0], M * 4)); OK(hipMalloc(&dx[1], M * 4));
    OK(hipMalloc(&dg[0], M * 8)); OK(hipMalloc(&dg[1], M * 8)); OK(hipMalloc(&dg[2], M * 8)); OK(hipMalloc(&err, 4));
    OK(hipMemcpy(dW, hw.data(), hw.size() * 2, hipMemcpyHostToDevice));
    OK(hipMemset(err, 0, 4));
    hipStream_t s0, s1;
    OK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking)); OK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    hipEvent_t fork, join, t0, t1;
    OK(hipEventCreateWithFlags(&fork, hipEventDisableTiming)); OK(hipEventCreateWithFlags(&join, hipEventDisableTiming));
    OK(hipEventCreate(&t0)); OK(hipEventCreate(&t1));

    // plain: kernel i reads vector i & 1 and writes vector (i + 1) & 1.  Granules: kernel i waits for tag i on buffer i % 3 and writes tag
    // i + 1 into buffer (i + 1) % 3 - THREE buffers, because kernel i + 1 may finish (and write buffer (i + 2) % 3) while late workgroups of
    // kernel i still read buffer i % 3; kernel i + 2 (which writes buffer i % 3 again) sits behind kernel i in the same stream.
    auto capture = [&](int variant, hipGraphExec_t * out) -> int {          // 0 = A, 1 = C, 2 = B
        hipGraph_t g;
        OK(hipStreamBeginCapture(s0, hipStreamCaptureModeThreadLocal));
        if (variant) { for (int b = 0; b < 3; b++) OK(hipMemsetAsync(dg[b], 0, M * 8, s0)); }      // every polled word, every replay
        if (variant == 2) { OK(hipEventRecord(fork, s0)); OK(hipStreamWaitEvent(s1, fork, 0)); }
        for (int i = 0; i < NK; i++) {
            hipStream_t s = (variant == 2 && (i & 1)) ? s1 : s0;
            const _Float16 * w = dW + (size_t) (i % n_layers) * M * K;
            if (variant == 0) hipLaunchKernelGGL(gemv_chain_kernel<false>, dim3(N_WG), dim3(256), 0, s, w, dx[i & 1], dx[(i + 1) & 1], (u64 *) nullptr, (u64 *) nullptr, 0u, 0, err);
            else              hipLaunchKernelGGL(gemv_chain_kernel<true>, dim3(N_WG), dim3(256), 0, s, w, dx[0], (float *) nullptr, dg[i % 3], dg[(i + 1) % 3], (unsigned) i, i == 0 ? 1 : 0, err);
        }
        if (variant == 2) { OK(hipEventRecord(join, s1)); OK(hipStreamWaitEvent(s0, join, 0)); }
        OK(hipStreamEndCapture(s0, &g));
        OK(hipGraphInstantiate(out, g, nullptr, nullptr, 0));
        (void) hipGraphDestroy(g);
        return 0;
    };
    const char * names[3] = {"A plain, one stream      ", "C granules, one stream   ", "B granules, two streams  "};
    std::vector<float> res[3];
    const int reps = 20;
    for (int variant = 0; variant < 3; variant++) {
        hipGraphExec_t e;
        if (capture(variant, &e)) return 1;
        OK(hipMemcpy(dx[0], hx.data(), M * 4, hipMemcpyHostToDevice));
        OK(hipGraphLaunch(e, s0)); OK(hipStreamSynchronize(s0));
        res[variant].resize(M);
        if (variant == 0) OK(hipMemcpy(res[0].data(), dx[NK & 1], M * 4, hipMemcpyDeviceToHost));
        else {
            std::vector<u64> g(M);
            OK(hipMemcpy(g.data(), dg[NK % 3], M * 8, hipMemcpyDeviceToHost));
            for (int i = 0; i < M; i++) { const unsigned u = (unsigned) g[i]; memcpy(&res[variant][i], &u, 4); }
        }
        unsigned herr = 0;
        OK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
        float ms = 0;
        if (!herr) {
            OK(hipEventRecord(t0, s0));
            for (int r = 0; r < reps; r++) OK(hipGraphLaunch(e, s0));
            OK(hipEventRecord(t1, s0)); OK(hipEventSynchronize(t1));
            OK(hipEventElapsedTime(&ms, t0, t1));
            OK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
        }
        printf("%s %.2f us per kernel, results %s, wait timeouts %s\n", names[variant], ms * 1000.0f / (reps * NK),
               variant == 0 ? "(reference)" : memcmp(res[variant].data(), res[0].data(), M * 4) == 0 ? "equal to A" : "DIFFERENT from A",
               herr ? "YES" : "none");
        OK(hipMemset(err, 0, 4));
        (void) hipGraphExecDestroy(e);
    }
    return 0;
}
