// mfma_f16_order_probe.hip - raw input / output pairs of the f16 matrix-core instructions, for an OFFLINE search of the arithmetic they
// implement (tools/mfma_f16_order.py drives it and analyses the dump with exact rational arithmetic on the CPU):
//   can v_mfma_f32_32x32x16_f16 be restated bit for bit on a CPU (exact products, a fixed grouping / alignment / rounding)?  If so the fine
//   model's products can move to the f16 matrix cores under a canonical order of their own.
// The binary is a dumb executor: it reads n trials {A: 64 lanes x 8 f16, B: 64 lanes x 8 f16, C: 64 lanes x 16 f32} from argv[1], runs
// ONE instruction per trial and variant and writes D (64 lanes x 16 f32) per trial and variant to argv[2]:
//   variant 0  v_mfma_f32_32x32x16_f16   (8 halves per lane, 16 accumulator registers)
//   variant 1  v_mfma_f32_16x16x32_f16   (8 halves per lane, 4 accumulator registers: C / D registers 0..3)
//   variant 2  v_mfma_f32_32x32x8_f16    (halves 0..3 of the lane, 16 registers)
//   variant 3  two chained 32x32x16 issues: D = mfma(A, B, mfma(A', B', C)) with A' = the trial's A with halves reversed lane-wise (checks
//              that the accumulator input of a second issue behaves exactly as C does)
// Build: hipcc --offload-arch=gfx950 -O2 -o tools/probes/mfma_f16_order_probe tools/probes/mfma_f16_order_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

__global__ void probe_kernel(const unsigned short * A, const unsigned short * B, const float * C, float * D, int n) {
    const int t = blockIdx.x, lane = threadIdx.x;
    half8 a, b, ar;
    for (int e = 0; e < 8; e++) {
        a[e] = __builtin_bit_cast(_Float16, A[((size_t) t * 64 + lane) * 8 + e]);
        b[e] = __builtin_bit_cast(_Float16, B[((size_t) t * 64 + lane) * 8 + e]);
    }
    for (int e = 0; e < 8; e++) ar[e] = a[7 - e];
    floatx16 c;
    for (int v = 0; v < 16; v++) c[v] = C[((size_t) t * 64 + lane) * 16 + v];
    float * d0 = D + (((size_t) 0 * n + t) * 64 + lane) * 16;
    float * d1 = D + (((size_t) 1 * n + t) * 64 + lane) * 16;
    float * d2 = D + (((size_t) 2 * n + t) * 64 + lane) * 16;
    float * d3 = D + (((size_t) 3 * n + t) * 64 + lane) * 16;
    {
        const floatx16 r = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
        for (int v = 0; v < 16; v++) d0[v] = r[v];
    }
    {
        floatx4 c4; for (int v = 0; v < 4; v++) c4[v] = c[v];
        const floatx4 r = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c4, 0, 0, 0);
        for (int v = 0; v < 16; v++) d1[v] = v < 4 ? r[v] : 0.0f;
    }
    {
        half4 a4, b4; for (int e = 0; e < 4; e++) { a4[e] = a[e]; b4[e] = b[e]; }
        const floatx16 r = __builtin_amdgcn_mfma_f32_32x32x8f16(a4, b4, c, 0, 0, 0);
        for (int v = 0; v < 16; v++) d2[v] = r[v];
    }
    {
        floatx16 r = __builtin_amdgcn_mfma_f32_32x32x16_f16(ar, b, c, 0, 0, 0);
        r = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, r, 0, 0, 0);
        for (int v = 0; v < 16; v++) d3[v] = r[v];
    }
}

int main(int argc, char ** argv) {
    if (argc < 3) { fprintf(stderr, "usage: %s in.bin out.bin\n", argv[0]); return 2; }
    FILE * f = fopen(argv[1], "rb");
    if (!f) { perror("in"); return 2; }
    int n = 0;
    if (fread(&n, 4, 1, f) != 1 || n <= 0 || n > (1 << 20)) { fprintf(stderr, "bad trial count\n"); return 2; }
    std::vector<unsigned short> A((size_t) n * 512), B((size_t) n * 512);
    std::vector<float> C((size_t) n * 1024), D((size_t) 4 * n * 1024);
    if (fread(A.data(), 2, A.size(), f) != A.size() || fread(B.data(), 2, B.size(), f) != B.size() || fread(C.data(), 4, C.size(), f) != C.size()) {
        fprintf(stderr, "short input\n"); return 2;
    }
    fclose(f);
    unsigned short * dA, * dB; float * dC, * dD;
    if (hipMalloc(&dA, A.size() * 2) != hipSuccess || hipMalloc(&dB, B.size() * 2) != hipSuccess || hipMalloc(&dC, C.size() * 4) != hipSuccess ||
        hipMalloc(&dD, D.size() * 4) != hipSuccess) { fprintf(stderr, "no device memory\n"); return 1; }
    (void) hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice);
    (void) hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice);
    (void) hipMemcpy(dC, C.data(), C.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe_kernel, dim3(n), dim3(64), 0, 0, dA, dB, dC, dD, n);
    if (hipDeviceSynchronize() != hipSuccess) { fprintf(stderr, "launch failed\n"); return 1; }
    (void) hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
    f = fopen(argv[2], "wb");
    if (!f) { perror("out"); return 2; }
    fwrite(D.data(), 4, D.size(), f);
    fclose(f);
    printf("mfma_f16_order_probe: %d trials x 4 variants written\n", n);
    return 0;
}
