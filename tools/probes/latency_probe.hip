// probe: launch floor inside a hipGraph, dependent-load latency (L2 / MALL / HBM), effective shader clock
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <numeric>
#include <random>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
__global__ void empty_k(int * p) { if (p && threadIdx.x == 9999) *p = 1; }
__global__ void chase(const unsigned * next, unsigned start, int hops, unsigned * out, long long * cyc, long long * wall) {
    unsigned i = start;
    long long c0 = clock64(), w0 = wall_clock64();
    for (int h = 0; h < hops; h++) i = next[i];
    long long c1 = clock64(), w1 = wall_clock64();
    out[0] = i; cyc[0] = c1 - c0; wall[0] = w1 - w0;
}
__global__ void stream_k(const uint4 * src, uint4 * dst, size_t n) {
    size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
    uint4 acc = {0, 0, 0, 0};
    for (; i < n; i += (size_t) gridDim.x * blockDim.x) { uint4 v = src[i]; acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w; }
    if (acc.x == 0x12345678) dst[0] = acc;
}
int main() {
    hipStream_t s; CK(hipStreamCreate(&s));
    int * d; CK(hipMalloc(&d, 4));
    // 1. graph of 64 empty kernels
    for (int blocks : {1, 192, 1024}) {
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < 64; i++) empty_k<<<blocks, 64, 0, s>>>(d);
        CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int i = 0; i < 5; i++) CK(hipGraphLaunch(ge, s));
        CK(hipEventRecord(e0, s));
        for (int i = 0; i < 50; i++) CK(hipGraphLaunch(ge, s));
        CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("graph of 64 empty kernels (%d blocks x 64): %.2f us per kernel\n", blocks, ms * 1000 / 50 / 64);
    }
    // 2. pointer chase
    for (size_t bytes : {(size_t) 256 << 10, (size_t) 16 << 20, (size_t) 128 << 20, (size_t) 1 << 30}) {
        size_t n = bytes / 4;
        std::vector<unsigned> nx(n);
        // random cycle with 64-element (256 B) granularity
        size_t lines = n / 64; std::vector<unsigned> perm(lines); std::iota(perm.begin(), perm.end(), 0u);
        std::mt19937 rng(1); std::shuffle(perm.begin(), perm.end(), rng);
        for (size_t i = 0; i < lines; i++) nx[(size_t) perm[i] * 64] = perm[(i + 1) % lines] * 64;
        unsigned * dn; unsigned * dout; long long * dc, * dw;
        CK(hipMalloc(&dn, bytes)); CK(hipMalloc(&dout, 4)); CK(hipMalloc(&dc, 8)); CK(hipMalloc(&dw, 8));
        CK(hipMemcpy(dn, nx.data(), bytes, hipMemcpyHostToDevice));
        for (int rep = 0; rep < 2; rep++) {
            chase<<<1, 1, 0, s>>>(dn, perm[0] * 64, 2000, dout, dc, dw);
            CK(hipStreamSynchronize(s));
        }
        long long c, w; CK(hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(&w, dw, 8, hipMemcpyDeviceToHost));
        printf("chase over %6zu KB: %.0f shader cycles/hop, %.1f ns/hop (wall clock 100MHz), => shader clock %.2f GHz\n", bytes >> 10,
               c / 2000.0, w * 10.0 / 2000.0, (double) c / (w * 10.0));
        hipFree(dn);
    }
    // 3. streaming read bandwidth
    {
        size_t bytes = (size_t) 2 << 30; uint4 * src, * dst; CK(hipMalloc(&src, bytes)); CK(hipMalloc(&dst, 64));
        CK(hipMemset(src, 1, bytes));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        stream_k<<<2048, 256, 0, s>>>(src, dst, bytes / 16);
        CK(hipEventRecord(e0, s));
        for (int i = 0; i < 10; i++) stream_k<<<2048, 256, 0, s>>>(src, dst, bytes / 16);
        CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("streaming read: %.2f TB/s\n", bytes * 10.0 / (ms * 1e-3) / 1e12);
    }
    return 0;
}
