// GPU box: what this particular device sustains on a pure-register f32 MFMA loop (the boxes of the pool differ by more than 10 %): the
// yardstick for kernel timings taken in the same call.  ~0.3 s.  hipcc --offload-arch=gfx950 -O3 -o /tmp/clock_calib scripts/clock_calib.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void k32(float* out, unsigned long long* st, int n) {
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a[4], b[4];
    for (int i = 0; i < 4; ++i) { a[i] = (float)((threadIdx.x * 7 + i * 13 + blockIdx.x) % 97) / 97.f - 0.5f; b[i] = (float)((threadIdx.x * 11 + i * 5 + blockIdx.x * 3) % 89) / 89.f - 0.5f; }
    for (int it = 0; it < n; ++it) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[k], b[k], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[k], b[(k + 1) & 3], acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(k + 1) & 3], b[k], acc[2], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(k + 2) & 3], b[(k + 3) & 3], acc[3], 0, 0, 0);
        }
        if ((it & 63) == 63) for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] *= 0.001f;
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { st[2 * blockIdx.x] = t1 - t0; st[2 * blockIdx.x + 1] = r1 - r0; }
}
int main() {
    float* out; unsigned long long* st; hipMalloc(&out, 1 << 22); hipMalloc(&st, 16 * 1024);
    static unsigned long long h[2 * 256];
    for (int rep = 0; rep < 2; ++rep) {
        const int n = 600000;
        auto t0 = std::chrono::high_resolution_clock::now();
        k32<<<256, 256>>>(out, st, n);
        hipDeviceSynchronize();
        double sec = std::chrono::duration<double>(std::chrono::high_resolution_clock::now() - t0).count();
        hipMemcpy(h, st, 16 * 256, hipMemcpyDeviceToHost);
        double ct = 0, cr = 0; for (int i = 0; i < 256; ++i) { ct += h[2 * i]; cr += h[2 * i + 1]; }
        printf("calib: pure f32 MFMA 32x32x2 loop: %.1f TFLOP/s, in-kernel clock %.2f GHz\n", 256.0 * 4 * n * 16.0 * 4096 / sec / 1e12, ct / cr * 0.1);
    }
    return 0;
}
