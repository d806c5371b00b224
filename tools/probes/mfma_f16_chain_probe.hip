// mfma_f16_chain_probe.hip - soak test companion of mfma_f16_order_probe.hip: chains of KB dependent v_mfma_f32_32x32x16_f16 issues (the
// accumulation of one output tile over K = 16 KB) on caller-supplied operands, raw lanes in, raw lanes out.  tools/mfma_f16_order.py soak
// compares the dump with the oracle's CPU restatement (oracle/mfma_f16_emu.h) element by element.
// Input file: int32 n, int32 KB, A [n][KB][64][8] f16 bits, B [n][KB][64][8] f16 bits, C [n][64][16] f32.  Output: D [n][64][16] f32.
// Build: hipcc --offload-arch=gfx950 -O2 -o tools/probes/mfma_f16_chain_probe tools/probes/mfma_f16_chain_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

__global__ void chain_kernel(const uint4 * A, const uint4 * B, const float * C, float * D, int KB) {
    const int t = blockIdx.x, lane = threadIdx.x;
    floatx16 acc;
    for (int v = 0; v < 16; v++) acc[v] = C[((size_t) t * 64 + lane) * 16 + v];
    for (int kb = 0; kb < KB; kb++) {
        const uint4 av = A[((size_t) t * KB + kb) * 64 + lane], bv = B[((size_t) t * KB + kb) * 64 + lane];
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, av), __builtin_bit_cast(half8, bv), acc, 0, 0, 0);
    }
    for (int v = 0; v < 16; v++) D[((size_t) t * 64 + lane) * 16 + v] = acc[v];
}

int main(int argc, char ** argv) {
    if (argc < 3) { fprintf(stderr, "usage: %s in.bin out.bin\n", argv[0]); return 2; }
    FILE * f = fopen(argv[1], "rb");
    if (!f) { perror("in"); return 2; }
    int n = 0, KB = 0;
    if (fread(&n, 4, 1, f) != 1 || fread(&KB, 4, 1, f) != 1 || n <= 0 || KB <= 0 || (size_t) n * KB > (1u << 24)) { fprintf(stderr, "bad header\n"); return 2; }
    std::vector<unsigned short> A((size_t) n * KB * 512), B((size_t) n * KB * 512);
    std::vector<float> C((size_t) n * 1024), D((size_t) n * 1024);
    if (fread(A.data(), 2, A.size(), f) != A.size() || fread(B.data(), 2, B.size(), f) != B.size() || fread(C.data(), 4, C.size(), f) != C.size()) {
        fprintf(stderr, "short input\n"); return 2;
    }
    fclose(f);
    void * dA, * dB; float * dC, * dD;
    if (hipMalloc(&dA, A.size() * 2) != hipSuccess || hipMalloc(&dB, B.size() * 2) != hipSuccess || hipMalloc(&dC, C.size() * 4) != hipSuccess ||
        hipMalloc(&dD, D.size() * 4) != hipSuccess) { fprintf(stderr, "no device memory\n"); return 1; }
    (void) hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice);
    (void) hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice);
    (void) hipMemcpy(dC, C.data(), C.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(chain_kernel, dim3(n), dim3(64), 0, 0, (const uint4 *) dA, (const uint4 *) dB, dC, dD, KB);
    if (hipDeviceSynchronize() != hipSuccess) { fprintf(stderr, "launch failed\n"); return 1; }
    (void) hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
    f = fopen(argv[2], "wb");
    if (!f) { perror("out"); return 2; }
    fwrite(D.data(), 4, D.size(), f);
    fclose(f);
    printf("mfma_f16_chain_probe: %d trials, chains of %d issues\n", n, KB);
    return 0;
}
