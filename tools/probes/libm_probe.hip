// probe: do (float)f((double)x) agree between device libm and host libm?
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
#include <random>
__global__ void k(const float * x, float * e, float * t, float * m, float * sg, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x; if (i >= n) return;
    e[i] = (float) exp((double) x[i]);
    t[i] = (float) tanh((double) x[i]);
    m[i] = (float) expm1((double) x[i]);
    sg[i] = 1.0f / (1.0f + (float) exp((double) (-x[i])));
}
int main() {
    const int n = 1 << 22;
    std::vector<float> x(n), e(n), t(n), m(n), sg(n);
    std::mt19937 rng(1); std::uniform_real_distribution<float> d(-8.f, 8.f);
    for (auto & v : x) v = d(rng);
    float *dx, *de, *dt, *dm, *ds;
    hipMalloc(&dx, n * 4); hipMalloc(&de, n * 4); hipMalloc(&dt, n * 4); hipMalloc(&dm, n * 4); hipMalloc(&ds, n * 4);
    hipMemcpy(dx, x.data(), n * 4, hipMemcpyHostToDevice);
    k<<<n / 256, 256>>>(dx, de, dt, dm, ds, n);
    hipMemcpy(e.data(), de, n * 4, hipMemcpyDeviceToHost); hipMemcpy(t.data(), dt, n * 4, hipMemcpyDeviceToHost);
    hipMemcpy(m.data(), dm, n * 4, hipMemcpyDeviceToHost); hipMemcpy(sg.data(), ds, n * 4, hipMemcpyDeviceToHost);
    long be = 0, bt = 0, bm = 0, bs = 0;
    for (int i = 0; i < n; i++) {
        be += e[i] != (float) exp((double) x[i]);
        bt += t[i] != (float) tanh((double) x[i]);
        bm += m[i] != (float) expm1((double) x[i]);
        bs += sg[i] != 1.0f / (1.0f + (float) exp((double) (-x[i])));
    }
    printf("n=%d mismatches: exp %ld tanh %ld expm1 %ld sigmoid %ld\n", n, be, bt, bm, bs);
    return 0;
}
