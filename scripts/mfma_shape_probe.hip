// GPU box micro-benchmark: which f32 MFMA shape sustains the most FLOP/s once the chip has settled its clock under load?
// Pure register loops on random operands (MI355X_MICROARCH.md, DVFS give-back item 7: for bf16 the 16x16 shape held a
// higher clock than 32x32 at equal cycles per FLOP).  One wave per SIMD (grid = CUs x 4 waves) and two.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__device__ __forceinline__ float rnd(unsigned s) { s = s * 747796405u + 2891336453u; s = ((s >> ((s >> 28) + 4)) ^ s) * 277803737u; return ((s >> 9) & 0xffff) * (1.0f / 32768.0f) - 1.0f; }

__global__ __launch_bounds__(256) void k32(float* out, unsigned long long* st, int n) {
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a[4], b[4];
    for (int i = 0; i < 4; ++i) { a[i] = rnd(threadIdx.x * 8 + i + blockIdx.x * 977); b[i] = rnd(threadIdx.x * 8 + 4 + i + blockIdx.x * 131); }
    for (int it = 0; it < n; ++it) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[k], b[k], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[k], b[(k + 1) & 3], acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(k + 1) & 3], b[k], acc[2], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(k + 2) & 3], b[(k + 3) & 3], acc[3], 0, 0, 0);
        }
        // keep the sums bounded without leaving the matrix pipe idle for long
        if ((it & 63) == 63) for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] *= 0.001f;
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { st[2 * blockIdx.x] = t1 - t0; st[2 * blockIdx.x + 1] = r1 - r0; }
}

__global__ __launch_bounds__(256) void k16(float* out, unsigned long long* st, int n) {
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    f32x4 acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float a[4], b[4];
    for (int i = 0; i < 4; ++i) { a[i] = rnd(threadIdx.x * 8 + i + blockIdx.x * 977); b[i] = rnd(threadIdx.x * 8 + 4 + i + blockIdx.x * 131); }
    for (int it = 0; it < n; ++it) {
        // same FLOPs per iteration as k32: 16 x (32x32x2) = 64 x (16x16x4)... here 4 k-steps x 16 tiles
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i * 4 + j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[(i + k) & 3], b[(j + k) & 3], acc[i * 4 + j], 0, 0, 0);
        if ((it & 63) == 63) for (int i = 0; i < 16; ++i) acc[i] *= 0.001f;
    }
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { st[2 * blockIdx.x] = t1 - t0; st[2 * blockIdx.x + 1] = r1 - r0; }
}

int main() {
    float* out; unsigned long long* st; hipMalloc(&out, 1 << 24); hipMalloc(&st, 16 * 8192);
    hipStream_t s; hipStreamCreate(&s);
    static unsigned long long h[2 * 2048];
    for (int rep = 0; rep < 2; ++rep)
        for (int wgs : {256, 512}) {
            for (int shape = 0; shape < 2; ++shape) {
                const int n = 2500000;          // ~1 s of work: the clock settles
                auto t0 = std::chrono::high_resolution_clock::now();
                if (shape == 0) k32<<<wgs, 256, 0, s>>>(out, st, n); else k16<<<wgs, 256, 0, s>>>(out, st, n);
                hipStreamSynchronize(s);
                double sec = std::chrono::duration<double>(std::chrono::high_resolution_clock::now() - t0).count();
                hipMemcpy(h, st, 16 * wgs, hipMemcpyDeviceToHost);
                double ct = 0, cr = 0; for (int i = 0; i < wgs; ++i) { ct += h[2 * i]; cr += h[2 * i + 1]; }
                // FLOPs: k32: 16 MFMA x 4096 per iteration per wave; k16: 64 MFMA x 2048... = 131072 per iteration per wave (x2 for FMA)
                double flop = (double)wgs * 4 * n * (shape == 0 ? 16.0 * 32 * 32 * 2 * 2 : 64.0 * 16 * 16 * 4 * 2);
                printf("%s  WGs %3d: %.3f s  %.1f TFLOP/s  in-kernel clock %.2f GHz  cycles/iter/wave %.0f\n", shape == 0 ? "32x32x2" : "16x16x4", wgs, sec,
                       flop / sec / 1e12, ct / cr * 0.1, ct / wgs / n);
            }
        }
    return 0;
}
