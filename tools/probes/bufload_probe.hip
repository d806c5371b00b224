// Probe: raw buffer loads (uniform descriptor + lane offset + scalar offset) against plain global loads on the same addresses.
// Finding (ROCm 7.2 clang, gfx950): __builtin_amdgcn_raw_buffer_load_b128 / _b64 are lowered to ONE buffer_load_dword whose value
// is replicated into every component; the LLVM intrinsic llvm.amdgcn.raw.buffer.load.v4f32 bound by name (the composable_kernel
// idiom, used in csrc/device_utils.h) emits buffer_load_dwordx4 and matches.  _b32 is fine either way.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef int int4v __attribute__((ext_vector_type(4)));
typedef float float4v __attribute__((ext_vector_type(4)));
__device__ float4v llvm_raw_buffer_load_v4f32(int4v rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.v4f32");
__device__ float   llvm_raw_buffer_load_f32(int4v rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.f32");
__device__ inline int4v rsrc_of(const void * p) {
    const unsigned long long a = (unsigned long long) p;
    int4v r; r.x = (int) (unsigned) a; r.y = (int) (unsigned) (a >> 32); r.z = -1; r.w = 0x00020000;
    return r;
}
// out[tid] = {16 strided dwords via buffer loads, the same via pointers}
__global__ void k_b32(const float * p, float2 * out, int h) {
    const int4v r = rsrc_of(p + (size_t) h * 65536);
    const unsigned tid = threadIdx.x;
    float a = 0.f, b = 0.f;
    #pragma unroll
    for (int i = 0; i < 16; i++) { a += llvm_raw_buffer_load_f32(r, (int) (tid * 4u), (int) (i * 4096u), 0); b += p[(size_t) h * 65536 + tid + i * 1024]; }
    out[tid] = float2{a, b};
}
__global__ void k_named_x4(const float * p, float4 * out, int h) {
    const float4v q = llvm_raw_buffer_load_v4f32(rsrc_of(p + (size_t) h * 65536), (int) (threadIdx.x * 16u), 16384, 0);
    out[threadIdx.x] = float4{q.x, q.y, q.z, q.w};
}
__global__ void k_builtin_x4(const float * p, float4 * out, int h) {
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void *) (p + (size_t) h * 65536), 0, 0xFFFFFFFFu, 0x00020000);
    const auto q = __builtin_amdgcn_raw_buffer_load_b128(r, (int) (threadIdx.x * 16u), 16384, 0);
    out[threadIdx.x] = float4{__builtin_bit_cast(float, q[0]), __builtin_bit_cast(float, q[1]), __builtin_bit_cast(float, q[2]), __builtin_bit_cast(float, q[3])};
}
int main() {
    const size_t n = 1 << 20;
    std::vector<float> h(n);
    for (size_t i = 0; i < n; i++) h[i] = (float) (i % 9973) * 0.001f;
    float * d; float2 * o2; float4 * o4a, * o4b;
    (void) hipMalloc(&d, n * 4); (void) hipMalloc(&o2, 1024 * 8); (void) hipMalloc(&o4a, 1024 * 16); (void) hipMalloc(&o4b, 1024 * 16);
    (void) hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_b32, dim3(1), dim3(1024), 0, 0, d, o2, 3);
    hipLaunchKernelGGL(k_named_x4, dim3(1), dim3(1024), 0, 0, d, o4a, 3);
    hipLaunchKernelGGL(k_builtin_x4, dim3(1), dim3(1024), 0, 0, d, o4b, 3);
    std::vector<float2> r2(1024); std::vector<float4> ra(1024), rb(1024);
    (void) hipMemcpy(r2.data(), o2, 1024 * 8, hipMemcpyDeviceToHost);
    (void) hipMemcpy(ra.data(), o4a, 1024 * 16, hipMemcpyDeviceToHost);
    (void) hipMemcpy(rb.data(), o4b, 1024 * 16, hipMemcpyDeviceToHost);
    int bad32 = 0, bad_named = 0, bad_builtin = 0;
    for (int t = 0; t < 1024; t++) {
        const float * want = h.data() + (size_t) 3 * 65536 + 4 * (t + 1024);
        bad32 += r2[t].x != r2[t].y;
        bad_named += !(ra[t].x == want[0] && ra[t].y == want[1] && ra[t].z == want[2] && ra[t].w == want[3]);
        bad_builtin += !(rb[t].x == want[0] && rb[t].y == want[1] && rb[t].z == want[2] && rb[t].w == want[3]);
    }
    printf("mismatching threads of 1024: dword loads %d, named dwordx4 intrinsic %d, clang b128 builtin %d (thread 5 got %g %g %g %g, wanted %g %g %g %g)\n",
           bad32, bad_named, bad_builtin, rb[5].x, rb[5].y, rb[5].z, rb[5].w, h[3 * 65536 + 4 * 1029], h[3 * 65536 + 4 * 1029 + 1], h[3 * 65536 + 4 * 1029 + 2], h[3 * 65536 + 4 * 1029 + 3]);
    return (bad32 || bad_named) ? 1 : 0;
}
