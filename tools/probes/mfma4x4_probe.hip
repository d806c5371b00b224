// mfma4x4_probe.hip - what v_mfma_f32_4x4x1_16b_f32 computes, checked bit for bit against host fmaf:
//   (1) operand / result layout: sixteen independent 4x4x1 blocks, block = lane / 4; which of (A index, B index) lives in the
//       result REGISTER and which in the LANE;
//   (2) that one issue is exactly D = fmaf(A, B, C) per element (one rounding), and that a sequence of issues on the same
//       accumulator is the sequential fmaf chain - the property C1 needs;
//   values are products of f16-representable numbers with cancelling accumulators, the case where fused and unfused differ.
// Build and run on the GPU box:  hipcc --offload-arch=gfx950 -O2 -o /tmp/mfma4x4_probe tools/probes/mfma4x4_probe.hip && /tmp/mfma4x4_probe
// Output: one line "layout=<rows_in_regs|rows_in_lanes|unknown> exact_chain=<0|1>".
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

typedef float floatx4 __attribute__((ext_vector_type(4)));

constexpr int STEPS = 96;

__global__ void probe_kernel(const float * A, const float * B, const float * C, float * D) {
    // A, B: [STEPS][64] one value per lane and step; C: [4][64]; D: [4][64]
    const int lane = threadIdx.x;
    floatx4 acc;
    for (int v = 0; v < 4; v++) acc[v] = C[v * 64 + lane];
    for (int s = 0; s < STEPS; s++)
        acc = __builtin_amdgcn_mfma_f32_4x4x1f32(A[s * 64 + lane], B[s * 64 + lane], acc, 0, 0, 0);
    for (int v = 0; v < 4; v++) D[v * 64 + lane] = acc[v];
}

static float f16ish(unsigned & seed) {          // a value with an 11-bit significand and a moderate exponent, either sign
    seed = seed * 1664525u + 1013904223u;
    const int mant = 1024 + (int) ((seed >> 8) & 1023);
    const int ex = (int) ((seed >> 20) & 15) - 10;
    const float v = ldexpf((float) mant, ex - 10);
    return (seed >> 30) & 1 ? -v : v;
}

int main() {
    static float A[STEPS * 64], B[STEPS * 64], C[4 * 64], D[4 * 64];
    unsigned seed = 12345u;
    for (auto & v : A) v = f16ish(seed);
    for (auto & v : B) v = f16ish(seed);
    for (auto & v : C) v = f16ish(seed) * 1.0009765625f;           // not f16-representable: rounding matters from the first step
    float *dA, *dB, *dC, *dD;
    hipMalloc(&dA, sizeof(A)); hipMalloc(&dB, sizeof(B)); hipMalloc(&dC, sizeof(C)); hipMalloc(&dD, sizeof(D));
    hipMemcpy(dA, A, sizeof(A), hipMemcpyHostToDevice); hipMemcpy(dB, B, sizeof(B), hipMemcpyHostToDevice); hipMemcpy(dC, C, sizeof(C), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe_kernel, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD);
    if (hipDeviceSynchronize() != hipSuccess) { printf("layout=unknown exact_chain=0 (launch failed)\n"); return 1; }
    hipMemcpy(D, dD, sizeof(D), hipMemcpyDeviceToHost);
    // hypothesis R ("rows in registers"): D[v][4b + j] = chain over s of fmaf(A[s][4b + v], B[s][4b + j], .)
    // hypothesis L ("rows in lanes"):     D[v][4b + j] = chain over s of fmaf(A[s][4b + j], B[s][4b + v], .)
    int okR = 1, okL = 1, okR_unfused = 1;
    for (int b = 0; b < 16; b++) for (int v = 0; v < 4; v++) for (int j = 0; j < 4; j++) {
        float r = C[v * 64 + 4 * b + j], l = r, u = r;
        for (int s = 0; s < STEPS; s++) {
            r = fmaf(A[s * 64 + 4 * b + v], B[s * 64 + 4 * b + j], r);
            l = fmaf(A[s * 64 + 4 * b + j], B[s * 64 + 4 * b + v], l);
            volatile float p = A[s * 64 + 4 * b + v] * B[s * 64 + 4 * b + j]; u = u + p;
        }
        const float got = D[v * 64 + 4 * b + j];
        if (memcmp(&got, &r, 4)) okR = 0;
        if (memcmp(&got, &l, 4)) okL = 0;
        if (memcmp(&got, &u, 4)) okR_unfused = 0;
    }
    printf("layout=%s exact_chain=%d unfused_also_matches=%d\n", okR ? "rows_in_regs" : okL ? "rows_in_lanes" : "unknown", (okR || okL) ? 1 : 0, okR_unfused);
    return (okR || okL) ? 0 : 2;
}
