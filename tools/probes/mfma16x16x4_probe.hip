// mfma16x16x4_probe.hip - what v_mfma_f32_16x16x4_f32 computes, checked bit for bit against host fmaf (companion of
// mfma4x4_probe.hip; written for the next formulation of the lock-step products):
//   D[i][j] = C[i][j] + sum over k = 0..3 of A[i][k] B[k][j], 16 x 16 outputs, 4 result registers per lane.
//   hypotheses checked: operand layout (A: lane l holds row l % 16 of k = l / 16; B: lane l holds column l % 16 of k = l / 16;
//   D register v of lane l: row 4 (l / 16) + v, column l % 16) and the ORDER of the four products inside one instruction -
//   ascending k as one fmaf chain (what C1 needs), descending, or a pairwise tree - over a sequence of dependent issues.
// Values are NOT f16-representable (24-bit significands, mixed signs), so every rounding matters and the orders tell apart.
// Build: hipcc --offload-arch=gfx950 -O2 -o tools/probes/mfma16x16x4_probe tools/probes/mfma16x16x4_probe.hip ; run it on the device.
// Output: "layout=<ok|unknown> k_order=<ascending_fmaf_chain|descending_fmaf_chain|pairwise|unknown>".
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

typedef float floatx4 __attribute__((ext_vector_type(4)));
constexpr int STEPS = 48;

__global__ void probe_kernel(const float * A, const float * B, const float * C, float * D) {
    const int lane = threadIdx.x;
    floatx4 acc;
    for (int v = 0; v < 4; v++) acc[v] = C[v * 64 + lane];
    for (int s = 0; s < STEPS; s++)
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[s * 64 + lane], B[s * 64 + lane], acc, 0, 0, 0);
    for (int v = 0; v < 4; v++) D[v * 64 + lane] = acc[v];
}

static float rnd(unsigned & seed) {
    seed = seed * 1664525u + 1013904223u;
    const int mant = 8388608 + (int) ((seed >> 7) & 8388607);          // full 24-bit significand
    const int ex = (int) ((seed >> 3) & 7) - 4;
    const float v = ldexpf((float) mant, ex - 23);
    return (seed >> 31) ? -v : v;
}

int main() {
    static float A[STEPS * 64], B[STEPS * 64], C[4 * 64], D[4 * 64];
    unsigned seed = 4242u;
    for (auto & v : A) v = rnd(seed);
    for (auto & v : B) v = rnd(seed);
    for (auto & v : C) v = rnd(seed);
    float *dA, *dB, *dC, *dD;
    if (hipMalloc(&dA, sizeof(A)) != hipSuccess || hipMalloc(&dB, sizeof(B)) != hipSuccess || hipMalloc(&dC, sizeof(C)) != hipSuccess ||
        hipMalloc(&dD, sizeof(D)) != hipSuccess) { printf("layout=unknown k_order=unknown (no device memory)\n"); return 1; }
    (void) hipMemcpy(dA, A, sizeof(A), hipMemcpyHostToDevice); (void) hipMemcpy(dB, B, sizeof(B), hipMemcpyHostToDevice);
    (void) hipMemcpy(dC, C, sizeof(C), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe_kernel, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD);
    if (hipDeviceSynchronize() != hipSuccess) { printf("layout=unknown k_order=unknown (launch failed)\n"); return 1; }
    (void) hipMemcpy(D, dD, sizeof(D), hipMemcpyDeviceToHost);
    // element (i, j): a_k = A[s][16 k + i], b_k = B[s][16 k + j]; result in register v = i % 4 of lane 16 (i / 4) + j
    int ok_asc = 1, ok_desc = 1, ok_pair = 1;
    for (int i = 0; i < 16; i++) for (int j = 0; j < 16; j++) {
        float asc = C[(i % 4) * 64 + 16 * (i / 4) + j], desc = asc, pair = asc;
        for (int s = 0; s < STEPS; s++) {
            float a[4], b[4];
            for (int k = 0; k < 4; k++) { a[k] = A[s * 64 + 16 * k + i]; b[k] = B[s * 64 + 16 * k + j]; }
            for (int k = 0; k < 4; k++) asc = fmaf(a[k], b[k], asc);
            for (int k = 3; k >= 0; k--) desc = fmaf(a[k], b[k], desc);
            { const float p01 = fmaf(a[1], b[1], a[0] * b[0]), p23 = fmaf(a[3], b[3], a[2] * b[2]); pair = pair + (p01 + p23); }
        }
        const float got = D[(i % 4) * 64 + 16 * (i / 4) + j];
        if (memcmp(&got, &asc, 4)) ok_asc = 0;
        if (memcmp(&got, &desc, 4)) ok_desc = 0;
        if (memcmp(&got, &pair, 4)) ok_pair = 0;
    }
    printf("layout=%s k_order=%s\n", (ok_asc || ok_desc || ok_pair) ? "ok" : "unknown",
           ok_asc ? "ascending_fmaf_chain" : ok_desc ? "descending_fmaf_chain" : ok_pair ? "pairwise" : "unknown");
    return ok_asc ? 0 : 2;
}
