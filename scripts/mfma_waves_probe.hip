// GPU box: does the f32 MFMA rate depend on how many waves share a SIMD, and what does a workgroup barrier every 64 MFMAs cost?
// One workgroup per CU of 256 / 512 / 768 threads, every wave the same register-only loop of v_mfma_f32_32x32x2_f32 on four
// accumulators; BAR = 1: a bare s_barrier after every 64 MFMAs of a wave.  hipcc --offload-arch=gfx950 -O3 -o /tmp/mwp scripts/mfma_waves_probe.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int THREADS, int BAR, int NACC>
__global__ __launch_bounds__(THREADS) void k(float* out, int n) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a[4], b[4];
    for (int i = 0; i < 4; ++i) { a[i] = (float)((threadIdx.x * 7 + i * 13 + blockIdx.x) % 97) / 97.f - 0.5f; b[i] = (float)((threadIdx.x * 11 + i * 5 + blockIdx.x * 3) % 89) / 89.f - 0.5f; }
    for (int it = 0; it < n; ++it) {
#pragma unroll
        for (int k4 = 0; k4 < 64 / NACC; ++k4)
#pragma unroll
            for (int q = 0; q < NACC; ++q) acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(k4 + q) & 3], b[(k4 + 2 * q) & 3], acc[q], 0, 0, 0);
        if (BAR) asm volatile("s_barrier" ::: "memory");
        if ((it & 15) == 15) for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] *= 0.001f;
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * THREADS + threadIdx.x] = s;
}
template <int THREADS, int BAR, int NACC>
void run(float* out) {
    const int n = 20000 * 256 / THREADS;
    for (int rep = 0; rep < 2; ++rep) {
        auto t0 = std::chrono::high_resolution_clock::now();
        k<THREADS, BAR, NACC><<<256, THREADS>>>(out, n);
        hipDeviceSynchronize();
        double sec = std::chrono::duration<double>(std::chrono::high_resolution_clock::now() - t0).count();
        if (rep) printf("waves/SIMD %d  barrier %d  accumulators %d: %.1f TFLOP/s\n", THREADS / 256, BAR, NACC, 256.0 * (THREADS / 64) * n * 64.0 * 4096 / sec / 1e12);
    }
}
int main() {
    float* out; hipMalloc(&out, 1 << 22);
    run<256, 0, 4>(out); run<512, 0, 4>(out); run<768, 0, 4>(out);
    run<256, 1, 4>(out); run<512, 1, 4>(out); run<768, 1, 4>(out);
    run<256, 0, 2>(out); run<768, 0, 2>(out); run<768, 1, 2>(out); run<768, 0, 1>(out);
    return 0;
}
