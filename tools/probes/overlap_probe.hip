// overlap_probe.hip - can a chain of DEPENDENT decode-sized GEMV kernels hide its kernel boundaries?
//
// The decode step is 62 dependent kernels; the in-kernel time line (profiles/r02_trace_decode_step.txt) puts 1.3-1.6 us of every
// ~3 us kernel slot into the boundary (completion -> dispatch -> first wave -> arguments), during which nothing is in flight.
// Variant B launches consecutive kernels on TWO streams (even kernels on one, odd ones on the other, captured as two parallel
// branches of one hipGraph) so that kernel i+1 is dispatched while kernel i still runs: its workgroups request their weight rows
// at once (they do not depend on kernel i) and only then wait, on a counter in device memory that kernel i's workgroups bump when
// their outputs are written (release / acquire at agent scope), for the input vector.  In-stream order keeps at most two kernels
// in flight (kernel i+2 sits behind kernel i in the same stream), so both always fit on the chip together.
// Every wait is BOUNDED (a wave gives up after ~20 ms and raises an error flag), so the probe cannot hang the GPU.
//
//   A: the same kernels in one stream (plain dependent launches in a hipGraph)          -> us per kernel
//   B: two streams + counters                                                          -> us per kernel, results equal to A's?
//
// Build: hipcc --offload-arch=gfx950 -O2 -o tools/probes/overlap_probe tools/probes/overlap_probe.hip ; run: timeout 30 tools/probes/overlap_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

typedef _Float16 half8 __attribute__((ext_vector_type(8)));

constexpr int K = 768, M = 3072, NK = 62, ROWS_PER_WG = 16, N_WG = M / ROWS_PER_WG;     // one FC-sized product per kernel

// y[m] = 2^-5 * sum_k W[m][k] x[k] (16 lanes per row, 4 rows per wave, 4 waves per workgroup); the next kernel reads y[0 .. K)
// wait_ctr == nullptr: plain kernel.  Otherwise: weights first, then wait until *wait_ctr >= wait_value, then x.
__global__ __launch_bounds__(256) void gemv_chain_kernel(const _Float16 * __restrict__ W, const float * xin, float * xout,
                                                         const unsigned * wait_ctr, unsigned wait_value, unsigned * signal_ctr, unsigned * err) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 15, rg = lane >> 4;
    const int m = blockIdx.x * ROWS_PER_WG + wave * 4 + rg;
    const _Float16 * wrow = W + (size_t) m * K + (c << 3);
    half8 wv[K / 128];
    #pragma unroll
    for (int b = 0; b < K / 128; b++) wv[b] = *reinterpret_cast<const half8 *>(wrow + (b << 7));
    if (wait_ctr) {
        if (tid == 0) {
            unsigned spins = 0;
            while (__hip_atomic_load(const_cast<unsigned *>(wait_ctr), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < wait_value) {
                __builtin_amdgcn_s_sleep(2);
                if (++spins > 400000u) { atomicOr(err, 1u); break; }              // ~20 ms: give up, never hang
            }
        }
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");                       // every wave reads x from memory, not from a stale cache line
    }
    float acc = 0.0f;
    #pragma unroll
    for (int b = 0; b < K / 128; b++) {
        const float4 x0 = *reinterpret_cast<const float4 *>(xin + (b << 7) + (c << 3));
        const float4 x1 = *reinterpret_cast<const float4 *>(xin + (b << 7) + (c << 3) + 4);
        const float xs[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
        #pragma unroll
        for (int e = 0; e < 8; e++) acc = fmaf((float) wv[b][e], xs[e], acc);
    }
    #pragma unroll
    for (int off = 1; off < 16; off <<= 1) acc += __shfl_xor(acc, off);
    if (c == 0) xout[m] = acc * 0.03125f;
    if (signal_ctr) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");                       // every wave pushes its rows out to memory ...
        __syncthreads();                                                         // ... all 16 rows of the workgroup are out ...
        if (tid == 0) __hip_atomic_fetch_add(signal_ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);      // ... then the count
    }
}

int main() {
    const int n_layers = 12;
    std::vector<_Float16> hw((size_t) n_layers * M * K);
    unsigned seed = 1u;
    for (auto & v : hw) { seed = seed * 1664525u + 1013904223u; v = (_Float16) (((int) (seed >> 20) % 2001 - 1000) * (1.0f / 1000.0f)); }
    std::vector<float> hx(M);
    for (int i = 0; i < M; i++) hx[i] = (float) ((i * 37) % 101 - 50) * 0.01f;
    _Float16 * dW; float * dx[2]; unsigned * ctr; unsigned * err;
    OK(hipMalloc(&dW, hw.size() * 2)); OK(hipMalloc(&dx[0], M * 4)); OK(hipMalloc(&dx[1], M * 4));
    OK(hipMalloc(&ctr, (NK + 1) * 4)); OK(hipMalloc(&err, 4));
    OK(hipMemcpy(dW, hw.data(), hw.size() * 2, hipMemcpyHostToDevice));
    hipStream_t s0, s1;
    OK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking)); OK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    hipEvent_t fork, join, t0, t1;
    OK(hipEventCreateWithFlags(&fork, hipEventDisableTiming)); OK(hipEventCreateWithFlags(&join, hipEventDisableTiming));
    OK(hipEventCreate(&t0)); OK(hipEventCreate(&t1));

    // ---- graph A: one stream, plain dependent kernels ------------------------------------------------------------------
    hipGraph_t gA; hipGraphExec_t eA;
    OK(hipStreamBeginCapture(s0, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < NK; i++)
        hipLaunchKernelGGL(gemv_chain_kernel, dim3(N_WG), dim3(256), 0, s0, dW + (size_t) (i % n_layers) * M * K, dx[i & 1], dx[(i + 1) & 1],
                           (const unsigned *) nullptr, 0u, (unsigned *) nullptr, err);
    OK(hipStreamEndCapture(s0, &gA));
    OK(hipGraphInstantiate(&eA, gA, nullptr, nullptr, 0));

    // ---- graph B: even kernels on s0, odd kernels on s1, counters between them -----------------------------------------
    // kernel i waits for ctr[i] == N_WG * epoch (bumped by kernel i-1) and bumps ctr[i+1]; kernel 0 waits for nothing.
    // `epoch` is baked per replay: the graph is re-captured per replay count below (cheap; a real engine would pass a device epoch).
    auto capture_B = [&](unsigned epoch, hipGraphExec_t * out) -> int {
        hipGraph_t g;
        OK(hipStreamBeginCapture(s0, hipStreamCaptureModeThreadLocal));
        OK(hipEventRecord(fork, s0)); OK(hipStreamWaitEvent(s1, fork, 0));
        for (int i = 0; i < NK; i++) {
            hipStream_t s = (i & 1) ? s1 : s0;
            hipLaunchKernelGGL(gemv_chain_kernel, dim3(N_WG), dim3(256), 0, s, dW + (size_t) (i % n_layers) * M * K, dx[i & 1], dx[(i + 1) & 1],
                               i == 0 ? (const unsigned *) nullptr : (const unsigned *) (ctr + i), (unsigned) N_WG * epoch, ctr + i + 1, err);
        }
        OK(hipEventRecord(join, s1)); OK(hipStreamWaitEvent(s0, join, 0));
        OK(hipStreamEndCapture(s0, &g));
        OK(hipGraphInstantiate(out, g, nullptr, nullptr, 0));
        (void) hipGraphDestroy(g);
        return 0;
    };

    std::vector<float> resA(M), resB(M);
    // A: warm up, time
    OK(hipMemcpy(dx[0], hx.data(), M * 4, hipMemcpyHostToDevice));
    OK(hipMemset(err, 0, 4));
    OK(hipGraphLaunch(eA, s0)); OK(hipStreamSynchronize(s0));
    OK(hipMemcpy(resA.data(), dx[NK & 1], M * 4, hipMemcpyDeviceToHost));
    const int reps = 20;
    OK(hipEventRecord(t0, s0));
    for (int r = 0; r < reps; r++) { OK(hipMemcpyAsync(dx[0], hx.data(), M * 4, hipMemcpyHostToDevice, s0)); OK(hipGraphLaunch(eA, s0)); }
    OK(hipEventRecord(t1, s0)); OK(hipEventSynchronize(t1));
    float msA = 0; OK(hipEventElapsedTime(&msA, t0, t1));

    // B: counters start at 0; replay r uses epoch r + 1
    OK(hipMemset(ctr, 0, (NK + 1) * 4));
    float msB = 0;
    unsigned herr = 0;
    {
        hipGraphExec_t eB;
        if (capture_B(1u, &eB)) return 1;
        OK(hipMemcpy(dx[0], hx.data(), M * 4, hipMemcpyHostToDevice));
        OK(hipGraphLaunch(eB, s0)); OK(hipStreamSynchronize(s0));
        OK(hipMemcpy(resB.data(), dx[NK & 1], M * 4, hipMemcpyDeviceToHost));
        OK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
        (void) hipGraphExecDestroy(eB);
    }
    const bool same = memcmp(resA.data(), resB.data(), M * 4) == 0;
    if (!herr) {
        std::vector<hipGraphExec_t> eBs(reps);
        for (int r = 0; r < reps; r++) if (capture_B(2u + (unsigned) r, &eBs[r])) return 1;
        OK(hipEventRecord(t0, s0));
        for (int r = 0; r < reps; r++) { OK(hipMemcpyAsync(dx[0], hx.data(), M * 4, hipMemcpyHostToDevice, s0)); OK(hipGraphLaunch(eBs[r], s0)); }
        OK(hipEventRecord(t1, s0)); OK(hipEventSynchronize(t1));
        OK(hipEventElapsedTime(&msB, t0, t1));
        OK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
    }
    printf("A one stream:   %.2f us per kernel (%d kernels per graph, %d replays)\n", msA * 1000.0f / (reps * NK), NK, reps);
    printf("B two streams:  %.2f us per kernel, results %s, wait timeouts %s\n", msB * 1000.0f / (reps * NK), same ? "equal to A" : "DIFFERENT from A",
           herr ? "YES (a wait gave up: the two kernels were not co-resident)" : "none");
    return 0;
}
