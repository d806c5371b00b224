// persist_layer_probe.hip - go / no-go probe for a PERSISTENT decode layer with run-ahead weight staging (round-5 review item 2).
//
// Question: the single-utterance decode step is 62 dependent kernels x ~3.0 us (profiles/r05_trace_decode_step.txt: 1.4 us boundary + 1.2 - 1.6 us
// until the first operands arrive).  A persistent kernel removes the boundaries and can hold layer l + 1's weights in LDS while layer l computes
// (bark-small: 14.2 MB per layer = 55 KB per CU), but every operator boundary of a decode layer is an ALL-TO-ALL edge: each of the 256 workgroups
// needs the whole activation vector its predecessors produced (768 f32 of x, 768 f16 of att, 3072 f16 of h).  What does one such edge cost inside
// one launch on this chip, in Bark's geometry?  The persistent layer pays per phase:  edge (publish -> every workgroup has gathered the vector in
// LDS) + the product out of LDS.  It wins only if that is clearly below the ~3.0 us a launch costs today (the review's bar: 2.7 us per phase).
//
// Method (MI355X_MICROARCH.md, "Persistent kernels: synchronisation and hand-off price list" - rows allgather / handoff-1to1): 8-byte {payload, tag}
// granules, written by ONE agent-scope (sc1) store each, polled with agent-scope relaxed loads; tag = phase number; two granule buffers by phase
// parity (a producer cannot lap a consumer by two phases: it needs every other workgroup's next vector first).  256 workgroups x 256 threads, one per
// CU.  Phase p: (1) every workgroup publishes its V / 256 share of the vector, (2) every workgroup gathers all V granules into LDS (all four waves
// sweep; unready granules are polled again; bounded: a spin that exceeds its budget raises a flag and the kernel unwinds), (3) a product of ROWS rows
// x V elements out of LDS-resident f16 weights (16 lanes per row, C1-like chunk chains, 16-lane shuffle reduction), whose results are the next
// phase's payload - so phases are truly dependent.  Optional: every workgroup streams its 55 KB share of the NEXT layer's weights from a 190 MB
// buffer while the five phases of a layer run (14 x 16-byte loads per thread issued at the layer's start, stored to the other half of LDS at its end).
//
// Output: us per phase (host-side events over one launch of many phases) per configuration, and workgroup 0's in-kernel split (wait for the
// vector / product).  Build: hipcc --offload-arch=gfx950 -O3 -o persist_layer_probe persist_layer_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

typedef unsigned long long u64;
typedef unsigned uint4v __attribute__((ext_vector_type(4)));
constexpr int NWG = 256, NT = 256;
constexpr int SPIN_BUDGET = 1 << 22;

struct Args {
    u64 * gran[2];              // [2][Vmax] granule buffers by phase parity
    const uint4v * weights;      // 190 MB stream source
    size_t wslab_u4;            // uint4 per layer (14.2 MB / 16)
    int * fail;                 // set when a spin gave up
    u64 * stamps;               // workgroup 0: [phase][2] wait / product cycles
    float * sink;
    int V;                      // granules per vector
    int rows;                   // rows of the product per workgroup
    int phases;
    int stream;                 // 1: stream next layer's weights during every group of 5 phases
};

__global__ __launch_bounds__(NT, 1) void persist_probe(const Args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    // LDS: [0, 55 KB) weights of this layer, [55, 110 KB) staging of the next, then the gathered vector (V x 4 bytes)
    _Float16 * w_lds = reinterpret_cast<_Float16 *>(lds_raw);
    uint4v * w_next = reinterpret_cast<uint4v *>(lds_raw + 56320);
    float * vec = reinterpret_cast<float *>(lds_raw + 2 * 56320);
    const int tid = threadIdx.x, wg = blockIdx.x, lane = tid & 63;
    const int V = a.V, share = V / NWG;                       // granules this workgroup publishes per phase
    // something finite in the LDS weights
    for (int i = tid; i < 56320 / 2; i += NT) w_lds[i] = (_Float16) (0.001f * (float) ((i * 7 + wg) & 63) - 0.03f);
    __syncthreads();
    float mine[8];
    #pragma unroll
    for (int i = 0; i < 8; i++) mine[i] = 0.01f * (float) (wg + i);
    uint4v stream_regs[14];
    unsigned failed = 0;
    for (int p = 0; p < a.phases && !failed; p++) {
        u64 * g = a.gran[p & 1];
        const unsigned tag = (unsigned) p + 1u;
        if (a.stream && p % 5 == 0) {
            // run-ahead: this workgroup's 55 KB of the next layer (3520 uint4 = 13.75 per thread -> 14 loads)
            const size_t layer = (size_t) ((p / 5 + 1) % 12);
            const uint4v * src = a.weights + layer * a.wslab_u4 + (size_t) wg * 3520;
            #pragma unroll
            for (int i = 0; i < 14; i++) { const int k = tid + NT * i; stream_regs[i] = __builtin_nontemporal_load(src + (k < 3520 ? k : 3519)); }
        }
        // (1) publish: `share` granules of this workgroup, one 8-byte agent-scope store each
        if (tid < share) {
            const u64 v = ((u64) tag << 32) | (u64) __builtin_bit_cast(unsigned, mine[tid & 7]);
            __hip_atomic_store(g + (size_t) wg * share + tid, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        const u64 t0 = wall_clock64();
        // (2) gather all V granules into LDS: every thread requests ALL its granules (up to 8) at once, then polls again the ones that were not there yet
        {
            u64 v[8];
            #pragma unroll
            for (int k = 0; k < 8; k++) if (tid + NT * k < V) v[k] = __hip_atomic_load(g + tid + NT * k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            #pragma unroll
            for (int k = 0; k < 8; k++) {
                if (tid + NT * k < V) {
                    int spins = 0;
                    while ((unsigned) (v[k] >> 32) != tag) {
                        if (++spins > SPIN_BUDGET) { failed = 1; break; }
                        __builtin_amdgcn_s_sleep(1);
                        v[k] = __hip_atomic_load(g + tid + NT * k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    vec[tid + NT * k] = __builtin_bit_cast(float, (unsigned) v[k]);
                }
            }
        }
        failed = __syncthreads_or((int) failed);
        if (failed) { if (tid == 0) atomicExch(a.fail, p + 1); break; }
        const u64 t1 = wall_clock64();
        // (3) product out of LDS: 16 lanes per row, 16 rows per pass; lane l of a row takes chunks l, l + 16, ... of 8 elements
        const int row = tid >> 4, l16 = tid & 15;
        float out = 0.0f;
        if (row < a.rows) {
            const _Float16 * wr = w_lds + ((size_t) row * V) % (56320 / 2 - V);
            float acc = 0.0f;
            for (int c = l16; c < V / 8; c += 16) {
                #pragma unroll
                for (int e = 0; e < 8; e++) acc = fmaf((float) wr[8 * c + e], vec[8 * c + e], acc);
            }
            acc += __shfl_xor(acc, 1, 64); acc += __shfl_xor(acc, 2, 64); acc += __shfl_xor(acc, 4, 64); acc += __shfl_xor(acc, 8, 64);
            out = acc;
        }
        // the rows' results become the next payload (keeps the phases dependent); lanes 0 of the first `share` rows hold them
        #pragma unroll
        for (int i = 0; i < 8; i++) mine[i] = 0.5f * mine[i] + 1e-3f * __shfl(out, (i % 4) * 16, 64);
        if (a.stream && p % 5 == 4) {
            #pragma unroll
            for (int i = 0; i < 14; i++) { const int k = tid + NT * i; if (k < 3520) w_next[k] = stream_regs[i]; }
        }
        __syncthreads();
        if (wg == 0 && tid == 0 && a.stamps) { a.stamps[2 * p] = t1 - t0; a.stamps[2 * p + 1] = wall_clock64() - t1; }
    }
    if (lane == 0) a.sink[wg * 4 + (tid >> 6)] = mine[0] + (a.stream ? __builtin_bit_cast(float, w_next[tid].x) * 0.0f : 0.0f);
}

int main() {
    hipDeviceProp_t prop; HIP_OK(hipGetDeviceProperties(&prop, 0));
    printf("device %s, %d CUs; 256 workgroups x 256 threads, one per CU; wall_clock64 at %d kHz\n", prop.gcnArchName, prop.multiProcessorCount, 100000);
    const int Vmax = 4096, phases = 1200;
    Args a{};
    HIP_OK(hipMalloc(&a.gran[0], sizeof(u64) * Vmax)); HIP_OK(hipMalloc(&a.gran[1], sizeof(u64) * Vmax));
    const size_t slab = 14155776 / 16;                       // uint4 per layer (bark-small: 7.08 M f16 weights)
    uint4v * w; HIP_OK(hipMalloc(&w, slab * 16 * 13)); HIP_OK(hipMemset(w, 1, slab * 16 * 13));     // 13: the shares of 256 workgroups (3520 x 16 B each) overhang a 14.16 MB slab by 262 KB
    a.weights = w; a.wslab_u4 = slab;
    HIP_OK(hipMalloc(&a.fail, 4)); HIP_OK(hipMalloc(&a.stamps, sizeof(u64) * 2 * phases)); HIP_OK(hipMalloc(&a.sink, 4 * NWG * 4));
    const int lds = 2 * 56320 + Vmax * 4;
    HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void *>(persist_probe), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipEvent_t e0, e1; HIP_OK(hipEventCreate(&e0)); HIP_OK(hipEventCreate(&e1));
    struct Cfg { const char * name; int V, rows, stream; };
    const Cfg cfgs[] = {
        {"edge only:  256 granules (2 KB), no product", 256, 0, 0},
        {"att  (768 f16 = 384 granules, 3 KB) -> proj 3 rows", 512, 3, 0},                 // 512: the next multiple of 256 (share = 2)
        {"x    (768 f32 = 768 granules, 6 KB) -> qkv 9 rows", 768, 9, 0},
        {"x    (768 f32 = 768 granules, 6 KB) -> fc 12 rows", 768, 12, 0},
        {"h    (3072 f16 = 1536 granules, 12 KB) -> mproj 3 rows", 1536, 3, 0},
        {"x -> fc 12 rows, next layer's 55 KB per CU streaming underneath", 768, 12, 1},
        {"h -> mproj 3 rows, next layer's 55 KB per CU streaming underneath", 1536, 3, 1},
    };
    for (const Cfg & c : cfgs) {
        a.V = c.V; a.rows = c.rows; a.stream = c.stream; a.phases = phases;
        double best = 1e30; int fail = 0;
        std::vector<u64> st((size_t) 2 * phases);
        for (int rep = 0; rep < 4; rep++) {
            HIP_OK(hipMemset(a.gran[0], 0, sizeof(u64) * Vmax)); HIP_OK(hipMemset(a.gran[1], 0, sizeof(u64) * Vmax)); HIP_OK(hipMemset(a.fail, 0, 4));
            HIP_OK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(persist_probe, dim3(NWG), dim3(NT), lds, 0, a);
            HIP_OK(hipEventRecord(e1, 0));
            HIP_OK(hipEventSynchronize(e1));
            float ms; HIP_OK(hipEventElapsedTime(&ms, e0, e1));
            HIP_OK(hipMemcpy(&fail, a.fail, 4, hipMemcpyDeviceToHost));
            if (fail) break;
            if (rep && ms * 1000.0 / phases < best) { best = ms * 1000.0 / phases; HIP_OK(hipMemcpy(st.data(), a.stamps, sizeof(u64) * 2 * phases, hipMemcpyDeviceToHost)); }
        }
        if (fail) { printf("%-70s  a spin gave up in phase %d (not all 256 workgroups resident?)\n", c.name, fail); continue; }
        double wait = 0, prod = 0;
        for (int p = 100; p < phases; p++) { wait += (double) st[2 * p]; prod += (double) st[2 * p + 1]; }
        wait /= (phases - 100) * 100.0; prod /= (phases - 100) * 100.0;              // 100 MHz counter -> us
        printf("%-70s  %.2f us per phase   (workgroup 0: %.2f us publish -> vector gathered, %.2f us product + barrier)\n", c.name, best, wait, prod);
    }
    printf("today, per kernel of the launch chain (profiles/r05_trace_decode_step.txt, production build): 188 us / 62 = 3.03 us; go only below 2.7 us per phase\n");
    return 0;
}
