// A stand-in for <hip/hip_runtime.h> that lets the kernels of bark.cpp_amd/csrc run on the HOST, work-item for work-item (tests/test_simt_emulation.py):
// test infrastructure only - nothing of the product includes it.  One workgroup at a time, one fiber per work-item; `__shared__` variables are
// function-local statics (shared by the work-items of the running workgroup), __syncthreads() is a barrier that leaving work-items drop out of, and
// the wave-wide operations the kernels use (DPP row permutations, readlane, shuffles, matrix-core instructions) are rendezvous of the 64 work-items of a wave.  Floating point:
// fmaf is the hardware FMA (-mfma), f32 <-> f16 conversions are IEEE round-to-nearest-even as on the device, -ffp-contract=off as in the product.
#pragma once
#include <algorithm>
#include <chrono>
#include <mutex>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <functional>
#include <memory>
#include <thread>
#include <vector>

#include "../../oracle/mfma_f16_emu.h"

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __shared__ static
#define __launch_bounds__(...)

struct dim3 { unsigned x = 1, y = 1, z = 1; dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {} };
struct uint3_sim { unsigned x, y, z; };
inline thread_local uint3_sim threadIdx{0, 0, 0}, blockIdx{0, 0, 0}, blockDim{1, 1, 1}, gridDim{1, 1, 1};

struct float4 { float x, y, z, w; };
struct float2 { float x, y; };
struct int4 { int x, y, z, w; };
struct uint4 { unsigned x, y, z, w; };
struct uint2 { unsigned x, y; };
inline int4 make_int4(int a, int b, int c, int d) { return int4{a, b, c, d}; }
inline uint4 make_uint4(unsigned a, unsigned b, unsigned c, unsigned d) { return uint4{a, b, c, d}; }
inline float4 make_float4(float a, float b, float c, float d) { return float4{a, b, c, d}; }
inline float2 make_float2(float a, float b) { return float2{a, b}; }
inline uint2 make_uint2(unsigned a, unsigned b) { return uint2{a, b}; }

// atomics: one work-item runs at a time
inline int atomicAdd(int * p, int v) { const int o = *p; *p = o + v; return o; }
inline unsigned atomicAdd(unsigned * p, unsigned v) { const unsigned o = *p; *p = o + v; return o; }
inline float atomicAdd(float * p, float v) { const float o = *p; *p = o + v; return o; }
inline unsigned atomicMax(unsigned * p, unsigned v) { const unsigned o = *p; if (v > o) *p = v; return o; }
inline int atomicMax(int * p, int v) { const int o = *p; if (v > o) *p = v; return o; }
inline int min(int a, int b) { return a < b ? a : b; }
inline int max(int a, int b) { return a > b ? a : b; }
inline unsigned min(unsigned a, unsigned b) { return a < b ? a : b; }
inline unsigned max(unsigned a, unsigned b) { return a > b ? a : b; }

// ---- the host API the engine uses: memory is host memory, a stream executes at once unless it is capturing, a graph is the list of what was captured ----
struct sim_stream { std::vector<std::function<void()>> * capture = nullptr; };
typedef sim_stream * hipStream_t;
typedef std::vector<std::function<void()>> sim_graph;
typedef sim_graph * hipGraph_t;
typedef sim_graph * hipGraphExec_t;
struct sim_event { std::chrono::steady_clock::time_point t; };
typedef sim_event * hipEvent_t;
typedef int hipError_t;
constexpr hipError_t hipSuccess = 0, hipErrorInvalidValue = 1;
enum hipMemcpyKind { hipMemcpyHostToHost, hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };
enum hipStreamCaptureMode { hipStreamCaptureModeGlobal, hipStreamCaptureModeThreadLocal, hipStreamCaptureModeRelaxed };
constexpr unsigned hipStreamNonBlocking = 1, hipHostMallocDefault = 0;
struct hipDeviceProp_t { char name[256] = "host emulation (tests/simt)"; char gcnArchName[256] = "gfx950:emulated"; int multiProcessorCount = 256; size_t totalGlobalMem = (size_t) 1 << 40; };
constexpr int hipFuncAttributeMaxDynamicSharedMemorySize = 0;
inline const char * hipGetErrorString(hipError_t) { return "emulated HIP runtime"; }
inline hipError_t hipGetLastError() { return hipSuccess; }
inline hipError_t hipGetDeviceCount(int * n) { *n = 1; return hipSuccess; }
inline hipError_t hipGetDevice(int * d) { *d = 0; return hipSuccess; }
inline hipError_t hipSetDevice(int d) { return d == 0 ? hipSuccess : hipErrorInvalidValue; }
inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
inline hipError_t hipGetDeviceProperties(hipDeviceProp_t * p, int) { *p = hipDeviceProp_t(); return hipSuccess; }
inline hipError_t hipFuncSetAttribute(const void *, int, int) { return hipSuccess; }
inline hipError_t hipDeviceGetStreamPriorityRange(int * least, int * greatest) { *least = 0; *greatest = -1; return hipSuccess; }
namespace sim {
// every allocation is registered, so that a request (the weight-prefetch experiment) can be checked against what is actually allocated
inline std::mutex & alloc_mutex() { static std::mutex m; return m; }
inline std::vector<std::pair<const char *, size_t>> & allocs() { static std::vector<std::pair<const char *, size_t>> v; return v; }
inline bool inside_an_allocation(const void * p, size_t n) {
    std::lock_guard<std::mutex> g(alloc_mutex());
    for (auto & a : allocs()) if ((const char *) p >= a.first && (const char *) p + n <= a.first + a.second) return true;
    return false;
}
inline long & prefetch_requests() { static long n = 0; return n; }      // weight-prefetch requests seen (and checked) so far
struct PrefetchReport { ~PrefetchReport() { if (getenv("BARK_SIM_VERBOSE")) fprintf(stderr, "sim: %ld prefetch requests, every one inside an allocation\n", prefetch_requests()); } };
inline PrefetchReport prefetch_report;
inline void enqueue(hipStream_t s, std::function<void()> f) { if (s && s->capture) s->capture->push_back(std::move(f)); else f(); }
}
inline hipError_t hipMalloc(void ** p, size_t n) {
    *p = aligned_alloc(256, (n + 255) / 256 * 256);
    if (!*p) return hipErrorInvalidValue;
    memset(*p, 0xA5, n);                                      // device memory starts with arbitrary contents
    std::lock_guard<std::mutex> g(sim::alloc_mutex()); sim::allocs().emplace_back((const char *) *p, n);
    return hipSuccess;
}
template <typename T> hipError_t hipMalloc(T ** p, size_t n) { return hipMalloc(reinterpret_cast<void **>(p), n); }
inline hipError_t hipFree(void * p) {
    if (!p) return hipSuccess;
    { std::lock_guard<std::mutex> g(sim::alloc_mutex()); auto & v = sim::allocs(); for (size_t i = 0; i < v.size(); i++) if (v[i].first == p) { v.erase(v.begin() + (long) i); break; } }
    free(p); return hipSuccess;
}
template <typename T> hipError_t hipHostMalloc(T ** p, size_t n, unsigned = 0) { *p = static_cast<T *>(malloc(n)); return *p ? hipSuccess : hipErrorInvalidValue; }
inline hipError_t hipHostFree(void * p) { free(p); return hipSuccess; }
inline hipError_t hipMemcpy(void * d, const void * s, size_t n, hipMemcpyKind) { memmove(d, s, n); return hipSuccess; }
inline hipError_t hipMemcpyAsync(void * d, const void * s, size_t n, hipMemcpyKind, hipStream_t st) { sim::enqueue(st, [=] { memmove(d, s, n); }); return hipSuccess; }
inline hipError_t hipMemsetAsync(void * d, int v, size_t n, hipStream_t st) { sim::enqueue(st, [=] { memset(d, v, n); }); return hipSuccess; }
inline hipError_t hipStreamCreateWithFlags(hipStream_t * s, unsigned) { *s = new sim_stream(); return hipSuccess; }
inline hipError_t hipStreamCreateWithPriority(hipStream_t * s, unsigned, int) { *s = new sim_stream(); return hipSuccess; }
inline hipError_t hipExtStreamCreateWithCUMask(hipStream_t * s, unsigned, const unsigned *) { *s = new sim_stream(); return hipSuccess; }
inline hipError_t hipStreamDestroy(hipStream_t s) { delete s; return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
inline hipError_t hipStreamBeginCapture(hipStream_t s, hipStreamCaptureMode) { if (!s || s->capture) return hipErrorInvalidValue; s->capture = new sim_graph(); return hipSuccess; }
inline hipError_t hipStreamEndCapture(hipStream_t s, hipGraph_t * g) { if (!s || !s->capture) { *g = nullptr; return hipErrorInvalidValue; } *g = s->capture; s->capture = nullptr; return hipSuccess; }
inline hipError_t hipGraphInstantiate(hipGraphExec_t * e, hipGraph_t g, void *, void *, size_t) { *e = new sim_graph(*g); return hipSuccess; }
inline hipError_t hipGraphDestroy(hipGraph_t g) { delete g; return hipSuccess; }
inline hipError_t hipGraphExecDestroy(hipGraphExec_t e) { delete e; return hipSuccess; }
inline hipError_t hipGraphLaunch(hipGraphExec_t e, hipStream_t s) { for (auto & f : *e) sim::enqueue(s, f); return hipSuccess; }
inline hipError_t hipEventCreate(hipEvent_t * e) { *e = new sim_event(); return hipSuccess; }
inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t) { e->t = std::chrono::steady_clock::now(); return hipSuccess; }
inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
inline hipError_t hipEventElapsedTime(float * ms, hipEvent_t a, hipEvent_t b) { *ms = std::chrono::duration<float, std::milli>(b->t - a->t).count(); return hipSuccess; }

// ---- the running workgroup: one FIBER per work-item on the calling host thread --------------------------------------------------------------
// Work-items are cooperative fibers (own stacks, a twenty-instruction context switch): a work-item runs until it reaches a barrier or a wave-wide
// rendezvous, where the scheduler moves on to the next one; a phase completes when every work-item that has not left yet has arrived.  Nothing runs in
// parallel, so `__shared__` statics need no protection inside a launch; launches from several host threads (a job's tail on its second stream) are
// serialised by one mutex.  A sweep in which nobody can move is a deadlock of the kernel under test and aborts with a message.
extern "C" void sim_switch(void ** save_sp, void * load_sp);
asm(".text\n.weak sim_switch\n.type sim_switch,@function\nsim_switch:\n"
    "  pushq %rbp\n  pushq %rbx\n  pushq %r12\n  pushq %r13\n  pushq %r14\n  pushq %r15\n"
    "  movq %rsp, (%rdi)\n  movq %rsi, %rsp\n"
    "  popq %r15\n  popq %r14\n  popq %r13\n  popq %r12\n  popq %rbx\n  popq %rbp\n  ret\n"
    ".size sim_switch, .-sim_switch\n");
namespace sim {
struct Bar { int expected = 0, arrived = 0; unsigned phase = 0; };
struct Fiber { void * sp = nullptr; int state = 0; Bar * on = nullptr; unsigned wait_phase = 0; unsigned tid = 0; };      // state: 0 runnable, 1 waiting, 2 done
struct Wave { Bar bar; uint64_t slot[64]; };
struct Group {
    Bar bar; Wave waves[16]; Fiber fibers[1024]; int n = 0;
    void * sched_sp = nullptr; int cur = -1;
    const std::function<void()> * body = nullptr;
    uint3_sim bidx{0, 0, 0}, bdim{1, 1, 1}, gdim{1, 1, 1};
};
inline thread_local Group * group = nullptr;
inline thread_local int lane_id = 0;
inline Wave & wave() { return group->waves[threadIdx.x >> 6]; }
inline void yield_to_scheduler() { Group * g = group; sim_switch(&g->fibers[g->cur].sp, g->sched_sp); }
inline void leave(Bar & b) { if (--b.expected > 0 && b.arrived == b.expected) { b.arrived = 0; b.phase++; } }
inline void wait(Bar & b) {
    if (++b.arrived == b.expected) { b.arrived = 0; b.phase++; return; }        // the last one to arrive goes straight on
    Fiber & f = group->fibers[group->cur];
    f.on = &b; f.wait_phase = b.phase; f.state = 1;
    yield_to_scheduler();
}
inline void fiber_entry() {
    Group * g = group;
    (*g->body)();
    Fiber & f = g->fibers[g->cur];
    // a work-item that is done no longer takes part in barriers and rendezvous (early returns are uniform per wave / workgroup in these kernels)
    leave(g->waves[f.tid >> 6].bar); leave(g->bar);
    f.state = 2;
    yield_to_scheduler();
    abort();                                                   // never resumed
}
// every work-item of the wave deposits `v`, then reads the deposit of work-item `from`
inline uint64_t exchange(uint64_t v, int from) {
    Wave & w = wave();
    w.slot[lane_id] = v;
    wait(w.bar);
    const uint64_t r = w.slot[from & 63];
    wait(w.bar);
    return r;
}
inline int dpp_source(int lane, int ctrl) {
    const int row = lane & ~15, i = lane & 15;
    if (ctrl >= 0 && ctrl <= 0xFF) return (lane & ~3) + ((ctrl >> (2 * (lane & 3))) & 3);       // quad_perm
    if (ctrl == 0x140) return row + (15 - i);                                                       // row_mirror
    if (ctrl == 0x141) return row + (i & 8) + (7 - (i & 7));                                        // row_half_mirror
    fprintf(stderr, "sim: unsupported DPP control 0x%x\n", ctrl); abort();
}
inline std::mutex & launch_mutex() { static std::mutex m; return m; }
constexpr size_t kStack = 256 * 1024;
inline char * stacks() { static char * p = static_cast<char *>(aligned_alloc(4096, 1024 * kStack)); return p; }
// One workgroup after the other (blocks of up to 1024 work-items, 1-D blocks as all kernels here use).
template <typename F> void launch(dim3 grid, int block, F && body_in) {
    std::lock_guard<std::mutex> lock(launch_mutex());
    static Group * g = new Group();                           // under the mutex: one workgroup at a time in the whole process
    const std::function<void()> body = body_in;
    if (block < 1 || block > 1024) { fprintf(stderr, "sim: block of %d work-items\n", block); abort(); }
    for (unsigned bz = 0; bz < grid.z; bz++) for (unsigned by = 0; by < grid.y; by++) for (unsigned bx = 0; bx < grid.x; bx++) {
        g->n = block; g->body = &body; g->bar = Bar(); g->bar.expected = block;
        for (int w = 0; w < 16; w++) { g->waves[w].bar = Bar(); g->waves[w].bar.expected = std::max(0, std::min(64, block - 64 * w)); }
        for (int t = 0; t < block; t++) {
            Fiber & f = g->fibers[t];
            f = Fiber(); f.tid = (unsigned) t;
            void ** top = reinterpret_cast<void **>(stacks() + (size_t) (t + 1) * kStack);       // 16-byte aligned
            top[-2] = reinterpret_cast<void *>(&fiber_entry);                                    // the `ret` of the first switch lands there with rsp = 8 mod 16
            for (int i = 3; i <= 8; i++) top[-i] = nullptr;                                       // rbp, rbx, r12 .. r15
            f.sp = top - 8;
        }
        Group * outer = group;
        group = g;
        int done = 0;
        while (done < block) {
            bool moved = false;
            for (int t = 0; t < block; t++) {
                Fiber & f = g->fibers[t];
                if (f.state == 1 && f.on->phase != f.wait_phase) f.state = 0;
                if (f.state != 0) continue;
                g->cur = t; lane_id = t & 63;
                threadIdx = {(unsigned) t, 0, 0}; blockIdx = {bx, by, bz}; blockDim = {(unsigned) block, 1, 1}; gridDim = {grid.x, grid.y, grid.z};
                sim_switch(&g->sched_sp, f.sp);
                moved = true;
                if (f.state == 2) done++;
            }
            if (!moved) { fprintf(stderr, "sim: deadlock in workgroup (%u, %u, %u): no work-item can move\n", bx, by, bz); abort(); }
        }
        group = outer;
    }
}
}  // namespace sim

inline void __syncthreads() { sim::wait(sim::group->bar); }
inline int sim_update_dpp(int, int src, int ctrl, int, int, bool) { return (int) (uint32_t) sim::exchange((uint32_t) src, sim::dpp_source(sim::lane_id, ctrl)); }
inline int sim_readlane(int v, int lane) { return (int) (uint32_t) sim::exchange((uint32_t) v, lane); }
#define __builtin_amdgcn_update_dpp(old, src, ctrl, rm, bm, bc) sim_update_dpp(old, src, ctrl, rm, bm, bc)
#define __builtin_amdgcn_readlane(v, lane) sim_readlane(v, lane)
#define __builtin_amdgcn_readfirstlane(v) sim_readlane(v, 0)
#define __builtin_amdgcn_sched_barrier(x) ((void) 0)
// s_waitcnt as the kernels use it - "the wave's own LDS writes have landed" before one lane reads what the other lanes wrote - relies on the lanes of a
// wave running in lock step; fibers do not, so it is a rendezvous of the wave here (every use sits in wave-uniform code; a divergent one would deadlock and abort)
#define __builtin_amdgcn_s_waitcnt(x) sim::wait(sim::wave().bar)
inline float __shfl_xor(float v, int mask, int = 64) { uint32_t u; memcpy(&u, &v, 4); u = (uint32_t) sim::exchange(u, sim::lane_id ^ mask); memcpy(&v, &u, 4); return v; }
inline double __shfl_xor(double v, int mask, int = 64) { uint64_t u; memcpy(&u, &v, 8); u = sim::exchange(u, sim::lane_id ^ mask); memcpy(&v, &u, 8); return v; }
inline int __shfl_xor(int v, int mask, int = 64) { return (int) (uint32_t) sim::exchange((uint32_t) v, sim::lane_id ^ mask); }
inline float __shfl(float v, int lane, int = 64) { uint32_t u; memcpy(&u, &v, 4); u = (uint32_t) sim::exchange(u, lane); memcpy(&v, &u, 4); return v; }
inline int __shfl(int v, int lane, int = 64) { return (int) (uint32_t) sim::exchange((uint32_t) v, lane); }
// __shfl_up: lanes below `delta` keep their own value
inline double __shfl_up(double v, unsigned delta, int = 64) { uint64_t u; memcpy(&u, &v, 8); const int from = sim::lane_id >= (int) delta ? sim::lane_id - (int) delta : sim::lane_id; u = sim::exchange(u, from); memcpy(&v, &u, 8); return v; }
inline float __shfl_up(float v, unsigned delta, int = 64) { uint32_t u; memcpy(&u, &v, 4); const int from = sim::lane_id >= (int) delta ? sim::lane_id - (int) delta : sim::lane_id; u = (uint32_t) sim::exchange(u, from); memcpy(&v, &u, 4); return v; }
inline int __shfl_up(int v, unsigned delta, int = 64) { const int from = sim::lane_id >= (int) delta ? sim::lane_id - (int) delta : sim::lane_id; return (int) (uint32_t) sim::exchange((uint32_t) v, from); }
// v_dot4_i32_i8: four signed byte products added to c (no clamp)
inline int sim_sdot4(int a, int b, int c) { for (int i = 0; i < 4; i++) c += (int) (int8_t) (a >> (8 * i)) * (int) (int8_t) (b >> (8 * i)); return c; }
#define __builtin_amdgcn_sdot4(a, b, c, clamp) sim_sdot4(a, b, c)
// v_mfma_i32_32x32x32_i8: lane l holds the 16 levels k = 16 (l / 32) .. + 15 of A row / B column l % 32; D as for the 32 x 32 f32 forms; exact integer sums
typedef int sim_intx4 __attribute__((ext_vector_type(4)));
typedef int sim_intx16 __attribute__((ext_vector_type(16)));
inline sim_intx16 sim_mfma_i32_32x32x32_i8(sim_intx4 a, sim_intx4 b, sim_intx16 acc) {
    struct Side { int8_t a[64][16], b[64][16]; };
    static Side side[16];
    sim::Wave & w = sim::wave();
    Side & sd = side[threadIdx.x >> 6];
    memcpy(sd.a[sim::lane_id], &a, 16); memcpy(sd.b[sim::lane_id], &b, 16);
    sim::wait(w.bar);
    const int half = sim::lane_id >> 5, col = sim::lane_id & 31;
    for (int r = 0; r < 16; r++) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
        int v = acc[r];
        for (int hk = 0; hk < 2; hk++) for (int k = 0; k < 16; k++) v += (int) sd.a[32 * hk + row][k] * (int) sd.b[32 * hk + col][k];
        acc[r] = v;
    }
    sim::wait(w.bar);
    return acc;
}
#define __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc, x, y, z) sim_mfma_i32_32x32x32_i8(a, b, acc)

// The f32 matrix-core instructions as wave-wide rendezvous, with the arithmetic the device probes established (tools/probes/mfma*_probe.hip,
// profiles/r02_mfma16x16x4_probe.txt, r03_mfma_*_probe.txt): every output element is one fmaf chain over k in ascending order.
typedef float sim_floatx16 __attribute__((ext_vector_type(16)));
namespace sim {
// deposit this work-item's two f32 operands; after the rendezvous every work-item reads the deposits it needs (a in the high, b in the low word)
inline void deposit2(float a, float b) {
    Wave & w = wave();
    uint32_t ua, ub; memcpy(&ua, &a, 4); memcpy(&ub, &b, 4);
    w.slot[lane_id] = ((uint64_t) ua << 32) | ub;
    wait(w.bar);
}
inline float dep_a(const Wave & w, int l) { const uint32_t x = (uint32_t) (w.slot[l] >> 32); float f; memcpy(&f, &x, 4); return f; }
inline float dep_b(const Wave & w, int l) { const uint32_t x = (uint32_t) w.slot[l]; float f; memcpy(&f, &x, 4); return f; }
}
// v_mfma_f32_32x32x2_f32: lane l holds A[l % 32][l / 32] and B[l / 32][l % 32]; D register r of lane l = element ((r & 3) + 8 (r >> 2) + 4 (l / 32), l % 32)
inline sim_floatx16 sim_mfma_32x32x2(float a, float b, sim_floatx16 acc) {
    sim::deposit2(a, b);
    const sim::Wave & w = sim::wave();
    const int half = sim::lane_id >> 5, col = sim::lane_id & 31;
    const float b0 = sim::dep_b(w, col), b1 = sim::dep_b(w, 32 + col);
    for (int r = 0; r < 16; r++) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
        float v = acc[r];
        v = fmaf(sim::dep_a(w, row), b0, v);
        v = fmaf(sim::dep_a(w, 32 + row), b1, v);
        acc[r] = v;
    }
    sim::wait(sim::wave().bar);
    return acc;
}
// v_mfma_f32_16x16x1_4b_f32: four independent 16 x 16 x 1 blocks; lane 16 g + r holds A_g[r] and B_g[r]; D register 4 b + v of lane 16 g + j = element (4 g + v, j) of block b
inline sim_floatx16 sim_mfma_16x16x1_4b(float a, float b, sim_floatx16 acc) {
    sim::deposit2(a, b);
    const sim::Wave & w = sim::wave();
    const int g = sim::lane_id >> 4, j = sim::lane_id & 15;
    for (int blk = 0; blk < 4; blk++) {
        const float bj = sim::dep_b(w, 16 * blk + j);
        for (int v = 0; v < 4; v++) acc[4 * blk + v] = fmaf(sim::dep_a(w, 16 * blk + 4 * g + v), bj, acc[4 * blk + v]);
    }
    sim::wait(sim::wave().bar);
    return acc;
}
// v_mfma_f32_32x32x16_f16: lane l holds the 8 halves k = 8 (l / 32) .. + 7 of A row / B column l % 32; D as for 32x32x2.  The arithmetic is the oracle's
// restatement of the instruction (oracle/mfma_f16_emu.h, established bit for bit on the device): two dependent groups of 8 products per element.
typedef _Float16 sim_half8 __attribute__((ext_vector_type(8)));
inline sim_floatx16 sim_mfma_32x32x16_f16(sim_half8 a, sim_half8 b, sim_floatx16 acc) {
    static_assert(sizeof(sim_half8) == 16, "eight packed halves");
    sim::Wave & w = sim::wave();
    // rendezvous of two 16-byte operands per lane: two rounds through the 64-bit slots would do; a per-wave side buffer is simpler
    static thread_local int dummy = 0; (void) dummy;
    struct Side { uint16_t a[64][8], b[64][8]; };
    static Side side[16];                                     // one per wave of the running workgroup (at most 1024 work-items)
    Side & sd = side[threadIdx.x >> 6];
    memcpy(sd.a[sim::lane_id], &a, 16); memcpy(sd.b[sim::lane_id], &b, 16);
    sim::wait(w.bar);
    const int half = sim::lane_id >> 5, col = sim::lane_id & 31;
    for (int r = 0; r < 16; r++) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
        float v = acc[r];
        v = mfma_emu::group8(sd.a[row], sd.b[col], 8, v);             // k = 0 .. 7: lanes 0 .. 31
        v = mfma_emu::group8(sd.a[32 + row], sd.b[32 + col], 8, v);   // k = 8 .. 15: lanes 32 .. 63
        acc[r] = v;
    }
    sim::wait(w.bar);
    return acc;
}
#define __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, x, y, z) sim_mfma_32x32x16_f16(a, b, acc)
inline unsigned long long sim_ballot(bool p) {
    unsigned long long m = 0;
    sim::Wave & w = sim::wave();
    w.slot[sim::lane_id] = p ? 1 : 0;
    sim::wait(w.bar);
    for (int l = 0; l < 64; l++) m |= (unsigned long long) (w.slot[l] & 1) << l;
    sim::wait(w.bar);
    return m;
}
#define __builtin_amdgcn_ballot_w64(p) sim_ballot(p)
#define __builtin_amdgcn_exp2f(x) exp2f(x)
#define __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, x, y, z) sim_mfma_32x32x2(a, b, acc)
#define __builtin_amdgcn_mfma_f32_16x16x1f32(a, b, acc, x, y, z) sim_mfma_16x16x1_4b(a, b, acc)

// a launch: executed at once, or recorded while the stream captures (arguments by value, as a kernel node keeps them)
#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) \
    sim::enqueue(stream, [=]() { sim::launch(dim3(grid), (int) dim3(block).x, [&]() { kernel(__VA_ARGS__); }); })
