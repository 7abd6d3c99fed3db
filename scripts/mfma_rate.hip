// GPU box: what the device sustains on register-only loops of the 16-bit MFMA shapes (bf16 / f16, 32x32x16 and 16x16x32), with one and two
// waves per SIMD and with every accumulator taking 1 or 3 back-to-back MFMAs (the bf16x6 / fp16x3 kernels chain 3 on one accumulator).
// hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_rate scripts/mfma_rate.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int KIND, int CHAIN, int NACC>
__global__ __launch_bounds__(512) void rate(float* out, unsigned long long* st, int n) {
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    f32x16 acc[NACC];
    f32x4 acc4[NACC];
    for (int i = 0; i < NACC; ++i) { for (int r = 0; r < 16; ++r) acc[i][r] = 0.f; for (int r = 0; r < 4; ++r) acc4[i][r] = 0.f; }
    bf16x8 ab[4]; f16x8 ah[4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 8; ++j) {
        const float v = (float)((threadIdx.x * 7 + i * 13 + j * 3 + blockIdx.x) % 97) / 97.f - 0.5f;
        ab[i][j] = (__bf16)v; ah[i][j] = (_Float16)v;
    }
    for (int it = 0; it < n; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i)
#pragma unroll
            for (int c = 0; c < CHAIN; ++c) {
                if (KIND == 0) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab[(i + c) & 3], ab[(i + 2 * c + 1) & 3], acc[i], 0, 0, 0);
                if (KIND == 1) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[(i + c) & 3], ah[(i + 2 * c + 1) & 3], acc[i], 0, 0, 0);
                if (KIND == 2) acc4[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab[(i + c) & 3], ab[(i + 2 * c + 1) & 3], acc4[i], 0, 0, 0);
                if (KIND == 3) acc4[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[(i + c) & 3], ah[(i + 2 * c + 1) & 3], acc4[i], 0, 0, 0);
            }
        if ((it & 63) == 63) for (int i = 0; i < NACC; ++i) { for (int r = 0; r < 16; ++r) acc[i][r] *= 0.001f; for (int r = 0; r < 4; ++r) acc4[i][r] *= 0.001f; }
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) { for (int r = 0; r < 16; ++r) s += acc[i][r]; for (int r = 0; r < 4; ++r) s += acc4[i][r]; }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { st[2 * blockIdx.x] = t1 - t0; st[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int KIND, int CHAIN, int NACC>
void run(const char* what, int threads, float* out, unsigned long long* st) {
    static unsigned long long h[2 * 256];
    const int n = 40000;
    const double flop_per = KIND < 2 ? 2.0 * 32 * 32 * 16 : 2.0 * 16 * 16 * 32;
    for (int rep = 0; rep < 2; ++rep) {
        auto t0 = std::chrono::high_resolution_clock::now();
        rate<KIND, CHAIN, NACC><<<256, threads>>>(out, st, n);
        hipDeviceSynchronize();
        const double sec = std::chrono::duration<double>(std::chrono::high_resolution_clock::now() - t0).count();
        hipMemcpy(h, st, 16 * 256, hipMemcpyDeviceToHost);
        double ct = 0, cr = 0; for (int i = 0; i < 256; ++i) { ct += h[2 * i]; cr += h[2 * i + 1]; }
        if (rep) printf("%-22s %d waves/SIMD, %d accumulators x chain %d: %7.1f TFLOP/s, in-kernel clock %.2f GHz\n", what, threads / 256, NACC, CHAIN,
                        256.0 * (threads / 64) * n * NACC * CHAIN * flop_per / sec / 1e12, ct / cr * 0.1);
    }
}

int main() {
    float* out; unsigned long long* st; hipMalloc(&out, 1 << 22); hipMalloc(&st, 16 * 1024);
    run<0, 1, 4>("bf16 32x32x16", 256, out, st);
    run<0, 1, 4>("bf16 32x32x16", 512, out, st);
    run<0, 3, 4>("bf16 32x32x16", 256, out, st);
    run<0, 3, 4>("bf16 32x32x16", 512, out, st);
    run<0, 3, 8>("bf16 32x32x16", 512, out, st);
    run<1, 1, 4>("f16  32x32x16", 256, out, st);
    run<1, 3, 4>("f16  32x32x16", 512, out, st);
    run<1, 3, 8>("f16  32x32x16", 512, out, st);
    run<1, 3, 8>("f16  32x32x16", 256, out, st);
    run<2, 1, 8>("bf16 16x16x32", 256, out, st);
    run<2, 3, 8>("bf16 16x16x32", 512, out, st);
    run<3, 1, 8>("f16  16x16x32", 256, out, st);
    run<3, 3, 8>("f16  16x16x32", 512, out, st);
    return 0;
}
