// kv_stream_probe.hip - how fast can the K cache of a lock-step batch be streamed at all?  Reads, per (slot, head), the first `ctx` keys of the
// 16 d-quad streams of the K layout [slot][L][H][16][P][4] (10 KB contiguous pieces 16 KB apart, slots 75 MB apart) exactly as
// attn_slots_scores_kernel does (thread = key, 16 float4 per thread), sums them into one float per thread (no arithmetic to speak of), against the
// same number of bytes read as ONE contiguous array.  Tells whether the attention kernels (4.4 TB/s at 64 slots) or the access pattern is the limit.
// Build: hipcc --offload-arch=gfx950 -O2 -o tools/probes/kv_stream_probe tools/probes/kv_stream_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__global__ __launch_bounds__(256) void k_pattern(const float4 * kc, size_t slot_stride4, int H, int P, int ctx, float * out) {
    const int g = blockIdx.x, h = blockIdx.y, slot = blockIdx.z, tid = threadIdx.x;
    if (g * 256 >= ctx) return;
    const int j = min(g * 256 + tid, ctx - 1);
    const float4 * kp = kc + (size_t) slot * slot_stride4 + (size_t) h * 16 * P + j;
    float4 kv[16];
    #pragma unroll
    for (int dq = 0; dq < 16; dq++) kv[dq] = kp[(size_t) dq * P];
    float s = 0.0f;
    #pragma unroll
    for (int dq = 0; dq < 16; dq++) s += kv[dq].x + kv[dq].y + kv[dq].z + kv[dq].w;
    if (s == 12345.678f) out[0] = s;
}
__global__ __launch_bounds__(256) void k_linear(const float4 * p, size_t n4, float * out) {
    float s = 0.0f;
    const size_t i0 = ((size_t) blockIdx.x * 256 + threadIdx.x) * 16;
    if (i0 + 16 <= n4) {
        float4 v[16];
        #pragma unroll
        for (int i = 0; i < 16; i++) v[i] = p[(size_t) blockIdx.x * 4096 + (size_t) i * 256 + threadIdx.x];
        #pragma unroll
        for (int i = 0; i < 16; i++) s += v[i].x + v[i].y + v[i].z + v[i].w;
    }
    if (s == 12345.678f) out[0] = s;
}

int main() {
    const int S = 64, L = 12, H = 12, P = 1024, ctx = 640;
    const size_t layer4 = (size_t) H * 16 * P, slot4 = layer4 * L;            // float4 units
    float4 * kc; float * out;
    if (hipMalloc(&kc, slot4 * S * sizeof(float4)) != hipSuccess || hipMalloc(&out, 4) != hipSuccess) { printf("no memory\n"); return 1; }
    (void) hipMemset(kc, 0, slot4 * S * sizeof(float4));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const double bytes = (double) S * H * 16.0 * ctx * 16.0;                      // per layer
    for (int rep = 0; rep < 2; rep++) {
        hipEventRecord(e0);
        for (int it = 0; it < 20; it++) for (int l = 0; l < L; l++)
            hipLaunchKernelGGL(k_pattern, dim3(4, H, S), dim3(256), 0, 0, kc + (size_t) l * layer4, slot4, H, P, ctx, out);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep) printf("K layout pattern (64 slots, ctx 640): %.1f MB per launch, %.2f us per launch, %.2f TB/s\n", bytes / 1e6, ms * 1000 / 240, bytes / (ms / 240 * 1e-3) / 1e12);
    }
    const size_t n4 = (size_t) (bytes / 16);
    const int blocks = (int) (n4 / 4096);
    for (int rep = 0; rep < 2; rep++) {
        hipEventRecord(e0);
        for (int it = 0; it < 240; it++) hipLaunchKernelGGL(k_linear, dim3(blocks), dim3(256), 0, 0, kc + (size_t) (it % 12) * (slot4 * 4), n4, out);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep) printf("contiguous array of the same size:     %.1f MB per launch, %.2f us per launch, %.2f TB/s\n", blocks * 65536.0 / 1e6, ms * 1000 / 240, blocks * 65536.0 / (ms / 240 * 1e-3) / 1e12);
    }
    return 0;
}
