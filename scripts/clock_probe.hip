// GPU box micro-benchmark: what clock does the chip hold during a chain of short, latency-bound launches (the LSTM
// sweep pattern) versus during one long MFMA-dense kernel?  clock = d(s_memtime) / d(s_memrealtime) * 100 MHz.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k_mfma(float* out, unsigned long long* stamps, int n, int slot) {
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    f32x4 a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0};
    float x = threadIdx.x * 1e-3f, y = blockIdx.x * 1e-3f;
    for (int i = 0; i < n; i += 2) {
        a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(y, x, a1, 0, 0, 0);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0[0] + a1[1];
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (blockIdx.x == 0 && threadIdx.x == 0) { stamps[2 * slot] = t1 - t0; stamps[2 * slot + 1] = r1 - r0; }
}
int main() {
    float* out; unsigned long long* st; hipMalloc(&out, 1 << 22); hipMalloc(&st, 16 * 4096);
    hipStream_t s; hipStreamCreate(&s);
    unsigned long long h[2 * 2048];
    for (int wgs : {128, 256}) {
        // (a) chain of short launches, 128 MFMAs per wave each
        for (int i = 0; i < 2000; ++i) k_mfma<<<wgs, 256, 0, s>>>(out, st, 128, i % 2000);
        hipStreamSynchronize(s);
        auto t0 = std::chrono::high_resolution_clock::now();
        for (int i = 0; i < 2000; ++i) k_mfma<<<wgs, 256, 0, s>>>(out, st, 128, i);
        hipStreamSynchronize(s);
        double us = std::chrono::duration<double, std::micro>(std::chrono::high_resolution_clock::now() - t0).count() / 2000;
        hipMemcpy(h, st, sizeof(h), hipMemcpyDeviceToHost);
        double ct = 0, cr = 0; for (int i = 1000; i < 2000; ++i) { ct += h[2 * i]; cr += h[2 * i + 1]; }
        printf("WGs %d short chain: %.2f us per launch, in-kernel %.0f cycles (%.2f us), clock %.2f GHz\n", wgs, us, ct / 1000, cr / 1000 / 100.0, ct / cr * 0.1);
        // (b) one long kernel
        k_mfma<<<wgs, 256, 0, s>>>(out, st, 4000000, 0);
        hipStreamSynchronize(s);
        hipMemcpy(h, st, 16, hipMemcpyDeviceToHost);
        printf("WGs %d long kernel: %.0f cycles per MFMA, clock %.2f GHz\n", wgs, (double)h[0] / 4000000, (double)h[0] / h[1] * 0.1);
    }
    return 0;
}
