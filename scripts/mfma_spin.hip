// Probe (GPU box): a low-priority kernel that only issues MFMAs of one shape, to co-run beside a persistent LSTM sweep:
// does the LENGTH of the co-runner's MFMA (64 / 32 / 8 cycles of the matrix pipe, which is not pre-emptive) set the sweep's slowdown?
// Built by scripts/sweep_corun.py into a shared object; spin_launch(kind, blocks, threads, iters, stream).
#include <hip/hip_runtime.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int KIND>
__device__ __forceinline__ void spin_body(float* out, int iters) {
    const float a = (float)threadIdx.x * 1e-9f, b = 1.0f;
    if (KIND == 0) {          // 32x32x2: 16 passes = 64 cycles
        f32x16 c0 = {0}, c1 = {0};
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c1, 0, 0, 0);
            }
        }
        if (c0[0] + c1[3] == 123.456f) out[0] = c0[1];
    } else if (KIND == 1) {   // 16x16x4: 8 passes = 32 cycles
        f32x4 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c1, 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c2, 0, 0, 0);
                c3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c3, 0, 0, 0);
            }
        }
        if (c0[0] + c1[3] + c2[1] + c3[2] == 123.456f) out[0] = c0[1];
    } else if (KIND == 3) {   // no MFMA at all: dependent v_fma chains (does the sweep get dispatched beside a busy low-priority kernel?)
        float c0 = a, c1 = b, c2 = a + 1, c3 = b + 1;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 64; ++u) { c0 = c0 * 1.0001f + a; c1 = c1 * 1.0001f + a; c2 = c2 * 1.0001f + a; c3 = c3 * 1.0001f + a; }
        }
        if (c0 + c1 + c2 + c3 == 123.456f) out[0] = c0;
    } else if (KIND == 4) {   // 32x32x2 with a pause after every 16: the pipe is free 20 % of the time
        f32x16 c0 = {0}, c1 = {0};
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c1, 0, 0, 0);
            }
            __builtin_amdgcn_s_sleep(4);      // 4 x 64 cycles
        }
        if (c0[0] + c1[3] == 123.456f) out[0] = c0[1];
    } else if (KIND == 5) {   // 16x16x4 with the same pauses: 32 x 32 cycles, then 4 x 64 cycles free
        f32x4 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c1, 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c2, 0, 0, 0);
                c3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c3, 0, 0, 0);
            }
            __builtin_amdgcn_s_sleep(4);
        }
        if (c0[0] + c1[3] + c2[1] + c3[2] == 123.456f) out[0] = c0[1];
    } else if (KIND == 6) {   // 4x4x1 with the same pauses
        f32x4 c[8] = {};
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 128; ++u) c[u & 7] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c[u & 7], 0, 0, 0);
            __builtin_amdgcn_s_sleep(4);
        }
        float s = 0;
        for (int u = 0; u < 8; ++u) s += c[u][0];
        if (s == 123.456f) out[0] = s;
    } else {                  // 4x4x1 (16 blocks): 2 passes = 8 cycles
        f32x4 c[8] = {};
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 16; ++u) c[u & 7] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c[u & 7], 0, 0, 0);
        }
        float s = 0;
        for (int u = 0; u < 8; ++u) s += c[u][0];
        if (s == 123.456f) out[0] = s;
    }
}

__global__ void spin0(float* out, int iters) { spin_body<0>(out, iters); }
__global__ void spin1(float* out, int iters) { spin_body<1>(out, iters); }
__global__ void spin2(float* out, int iters) { spin_body<2>(out, iters); }
__global__ void spin3(float* out, int iters) { spin_body<3>(out, iters); }
__global__ void spin4(float* out, int iters) { spin_body<4>(out, iters); }
__global__ void spin5(float* out, int iters) { spin_body<5>(out, iters); }
__global__ void spin6(float* out, int iters) { spin_body<6>(out, iters); }

extern "C" int spin_launch(int kind, int blocks, int threads, int iters, float* out, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    if (kind == 0) spin0<<<blocks, threads, 0, s>>>(out, iters);
    else if (kind == 1) spin1<<<blocks, threads, 0, s>>>(out, iters);
    else if (kind == 3) spin3<<<blocks, threads, 0, s>>>(out, iters);
    else if (kind == 4) spin4<<<blocks, threads, 0, s>>>(out, iters);
    else if (kind == 5) spin5<<<blocks, threads, 0, s>>>(out, iters);
    else if (kind == 6) spin6<<<blocks, threads, 0, s>>>(out, iters);
    else spin2<<<blocks, threads, 0, s>>>(out, iters);
    return (int)hipGetLastError();
}
