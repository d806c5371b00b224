// probe: what one XCD can do on its own - the numbers behind DESIGN.md section 5 (single-XCD persistent decode kernel):
//   1. which XCCs a CU-masked stream really runs on (HW_REG_XCC_ID per workgroup),
//   2. the cost of a barrier across the workgroups of one XCD (L2-coherent atomics) vs across the whole chip,
//   3. the weight-streaming bandwidth of one XCD's 32 CUs vs all 256.
// Every spin loop is bounded; build: hipcc --offload-arch=gfx950 -O3 xcd_probe.hip -o xcd_probe
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <vector>
#include <set>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__device__ __forceinline__ unsigned xcc_id() { unsigned v; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v)); return v & 0xF; }

__global__ void where_k(unsigned * out) { if (threadIdx.x == 0) out[blockIdx.x] = xcc_id(); }

// G workgroups meet `iters` times at a counter.  mode 0: relaxed agent-scope atomics only; mode 1: release add + acquire
// poll (what publishing data between the phases of a persistent kernel needs); mode 2: mode 0 + every workgroup stores a
// value (sc1 write-through) before arriving and reads its neighbour's value (sc1 load) after the meeting, checked.
__global__ void barrier_k(unsigned * cnt, unsigned * data, int iters, int mode, unsigned * fail) {
    const unsigned G = gridDim.x;
    for (int it = 0; it < iters; it++) {
        if (__hip_atomic_load(fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;      // a timed-out meeting ends the run quickly
        if (mode == 2 && threadIdx.x == 0) __hip_atomic_store(data + blockIdx.x, (unsigned) (it * 977 + blockIdx.x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // modes 3, 4: 1 - 4 KB per workgroup, double-buffered by iteration parity so that a slow reader never races the next write.
        // mode 3: plain stores + release add; mode 4: sc1 (write-through) stores + relaxed add; both: relaxed polls, one acquire
        // fence after the meeting, plain loads.
        unsigned * dbuf = data + (size_t) (it & 1) * 512 * 1024;
        if (mode == 3) dbuf[(size_t) blockIdx.x * 1024 + threadIdx.x] = (unsigned) (it * 977 + blockIdx.x) + threadIdx.x;
        if (mode == 4) __hip_atomic_store(dbuf + (size_t) blockIdx.x * 1024 + threadIdx.x, (unsigned) (it * 977 + blockIdx.x) + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        if (threadIdx.x == 0) {
            const unsigned target = (unsigned) (it + 1) * G;
            if (mode == 1 || mode == 3) __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            else           __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            int spins = 0;
            while (true) {
                const unsigned v = mode == 1 ? __hip_atomic_load(cnt, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT)
                                             : __hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if ((int) (v - target) >= 0) break;
                if (++spins > (1 << 20)) { atomicAdd(fail, 1u); break; }
            }
        }
        __syncthreads();
        if (mode >= 3) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            const unsigned nb = (blockIdx.x + 1 + (unsigned) it % (G - 1)) % G;
            const unsigned v = dbuf[(size_t) nb * 1024 + threadIdx.x];
            if (v != (unsigned) (it * 977 + nb) + threadIdx.x) atomicAdd(fail + 1, 1u);
        }
        if (mode == 2 && threadIdx.x == 0) {
            const unsigned nb = (blockIdx.x + 1) % G;
            const unsigned v = __hip_atomic_load(data + nb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (v != (unsigned) (it * 977 + nb)) atomicAdd(fail + 1, 1u);
        }
    }
}

__global__ void stream_k(const uint4 * src, uint4 * dst, size_t n) {
    size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
    uint4 acc = {0, 0, 0, 0};
    const size_t stride = (size_t) gridDim.x * blockDim.x;
    for (; i + 3 * stride < n; i += 4 * stride) {
        const uint4 a = src[i], b = src[i + stride], c = src[i + 2 * stride], d = src[i + 3 * stride];
        acc.x ^= a.x ^ b.x ^ c.x ^ d.x; acc.y ^= a.y ^ b.y ^ c.y ^ d.y; acc.z ^= a.z ^ b.z ^ c.z ^ d.z; acc.w ^= a.w ^ b.w ^ c.w ^ d.w;
    }
    for (; i < n; i += stride) { const uint4 v = src[i]; acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w; }
    if (acc.x == 0x12345678 && acc.y == 0x9abcdef0) dst[0] = acc;
}

int main() {
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    printf("%s: %d CUs\n", prop.gcnArchName, prop.multiProcessorCount);
    const int n_cu = prop.multiProcessorCount, n_words = (n_cu + 31) / 32;
    hipStream_t s_all; CK(hipStreamCreate(&s_all));
    unsigned * d_where, * d_cnt, * d_data, * d_fail;
    CK(hipMalloc(&d_where, 4096 * 4)); CK(hipMalloc(&d_cnt, 64)); CK(hipMalloc(&d_data, (size_t) 1024 * 1024 * 4)); CK(hipMalloc(&d_fail, 64));
    struct MaskCase { const char * name; std::vector<uint32_t> mask; int n_set; };
    std::vector<MaskCase> cases;
    {   // bit i -> XCC (i mod 8) if the driver interleaves the mask over the XCCs; bit i -> XCC (i / 32) if it is blocked
        MaskCase a{"bits i % 8 == 0", std::vector<uint32_t>(n_words, 0), 0}, b{"bits 0..31", std::vector<uint32_t>(n_words, 0), 0};
        for (int i = 0; i < n_cu; i += 8) { a.mask[i / 32] |= 1u << (i % 32); a.n_set++; }
        for (int i = 0; i < 32; i++) { b.mask[i / 32] |= 1u << (i % 32); b.n_set++; }
        cases.push_back(a); cases.push_back(b);
    }
    hipStream_t s_xcd = nullptr; int xcd_cus = 0;
    for (auto & mc : cases) {
        hipStream_t s;
        if (hipExtStreamCreateWithCUMask(&s, (uint32_t) mc.mask.size(), mc.mask.data()) != hipSuccess) { printf("CU mask '%s': stream creation failed\n", mc.name); continue; }
        CK(hipMemsetAsync(d_where, 0xFF, 4096 * 4, s));
        where_k<<<1024, 64, 0, s>>>(d_where);
        CK(hipStreamSynchronize(s));
        std::vector<unsigned> h(1024); CK(hipMemcpy(h.data(), d_where, 1024 * 4, hipMemcpyDeviceToHost));
        std::set<unsigned> seen(h.begin(), h.end());
        printf("CU mask '%s' (%d CUs): workgroups ran on XCCs {", mc.name, mc.n_set);
        for (unsigned x : seen) printf(" %u", x);
        printf(" }\n");
        if (seen.size() == 1 && !s_xcd) { s_xcd = s; xcd_cus = mc.n_set; }
    }
    {
        where_k<<<1024, 64, 0, s_all>>>(d_where);
        CK(hipStreamSynchronize(s_all));
        std::vector<unsigned> h(16); CK(hipMemcpy(h.data(), d_where, 64, hipMemcpyDeviceToHost));
        printf("unmasked stream, workgroups 0..15 on XCCs:"); for (unsigned x : h) printf(" %u", x); printf("\n");
    }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto time_barrier = [&](hipStream_t s, int G, int threads, int mode, const char * what) -> int {
        const int iters = 2000;
        CK(hipMemsetAsync(d_cnt, 0, 64, s)); CK(hipMemsetAsync(d_fail, 0, 64, s));
        barrier_k<<<G, threads, 0, s>>>(d_cnt, d_data, 10, mode, d_fail);
        CK(hipMemsetAsync(d_cnt, 0, 64, s));
        CK(hipEventRecord(e0, s));
        barrier_k<<<G, threads, 0, s>>>(d_cnt, d_data, iters, mode, d_fail);
        CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        unsigned f[2]; CK(hipMemcpy(f, d_fail, 8, hipMemcpyDeviceToHost));
        printf("barrier %-28s G=%3d x %4d threads, mode %d: %.3f us per barrier (timeouts %u, stale reads %u)\n", what, G, threads, mode, ms * 1000 / iters, f[0], f[1]);
        return 0;
    };
    for (int mode = 2; mode < 5; mode++) {
        if (s_xcd) { if (time_barrier(s_xcd, xcd_cus, 256, mode, "one XCD (CU-masked stream)")) return 1; }
        for (int G : {32, 64, 128}) if (time_barrier(s_all, G, 256, mode, "whole chip")) return 1;
    }
    for (int G : {32, 64, 128}) if (time_barrier(s_all, G, 1024, 4, "whole chip, 1024-thread WGs")) return 1;
    if (s_xcd) { if (time_barrier(s_xcd, xcd_cus, 1024, 2, "one XCD, 1024-thread WGs")) return 1; }
    // bandwidth: 192 MB of "weights" (fits the 256 MB Infinity Cache like bark-small's 185 MB), read repeatedly
    const size_t bytes = 192u << 20, n = bytes / 16;
    uint4 * src, * dst; CK(hipMalloc(&src, bytes)); CK(hipMalloc(&dst, 64));
    CK(hipMemset(src, 1, bytes));
    auto time_stream = [&](hipStream_t s, int G, int threads, const char * what) -> int {
        for (int i = 0; i < 3; i++) stream_k<<<G, threads, 0, s>>>(src, dst, n);
        CK(hipEventRecord(e0, s));
        const int reps = 10;
        for (int i = 0; i < reps; i++) stream_k<<<G, threads, 0, s>>>(src, dst, n);
        CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("stream 192 MB, %-28s G=%4d x %4d: %.1f us, %.0f GB/s\n", what, G, threads, ms * 1000 / reps, bytes / (ms / reps * 1e-3) / 1e9);
        return 0;
    };
    if (s_xcd) {
        for (int th : {256, 512, 1024}) if (time_stream(s_xcd, xcd_cus, th, "one XCD")) return 1;
        if (time_stream(s_xcd, xcd_cus * 2, 1024, "one XCD, 2 WGs per CU")) return 1;
    }
    for (int G : {32, 64, 128}) if (time_stream(s_all, G, 1024, "whole chip")) return 1;
    if (time_stream(s_all, 256, 1024, "whole chip")) return 1;
    if (time_stream(s_all, 1024, 1024, "whole chip")) return 1;
    return 0;
}
