// GPU box: what does one non-MFMA instruction cost beside v_mfma_f32_32x32x2_f32 (16 passes = 64 cycles; the f32 MFMA peak IS the
// vector f32 peak)?  Register-only loop, 3 waves per SIMD (1 where said), KIND instructions of one kind per MFMA:
//   0 nothing | 1 v_fma_f32 | 2 v_pk_add_f32 | 3 v_add_u32 (integer) | 4 s_nop 1 | 5 ds_read_b128, never waited inside the loop
//   6 ds_read_b32 | 7 v_xor_b32 | 8 buffer_load_dwordx4 ... lds (1 KiB LDS-DMA from a 64 KB hot region) | 9 v_mov_b32
// hipcc --offload-arch=gfx950 -O3 -o /tmp/mvp scripts/mfma_valu_probe.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int THREADS, int KIND, int NV, int PER>
__global__ __launch_bounds__(THREADS) void k(float* out, const float* src, int n) {
    __shared__ __attribute__((aligned(16))) float lds[16384];
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a = threadIdx.x * 0.001f, b = blockIdx.x * 0.002f;
    float v[8]; f32x2 pv[4]; unsigned iv[8];
    for (int i = 0; i < 8; ++i) { v[i] = threadIdx.x * 0.01f + i; iv[i] = threadIdx.x + i; }
    for (int i = 0; i < 4; ++i) pv[i] = f32x2{v[i], v[i + 4]};
    for (int i = threadIdx.x; i < 16384; i += THREADS) lds[i] = i;
    __syncthreads();
    const unsigned la = (unsigned)(size_t)(lds + (threadIdx.x & 63) * 4);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, 1 << 16, 0x00020000);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float4 t[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    for (int it = 0; it < n; ++it) {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            acc[q & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[q & 3], 0, 0, 0);
            if (q % PER == 0) {
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                if (KIND == 1) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[i & 7]) : "v"(a));
                if (KIND == 2) asm volatile("v_pk_add_f32 %0, %0, %1 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "+v"(pv[i & 3]) : "v"(pv[(i + 1) & 3]));
                if (KIND == 3) asm volatile("v_add_u32 %0, %0, %1" : "+v"(iv[i & 7]) : "v"(iv[(i + 1) & 7]));
                if (KIND == 4) asm volatile("s_nop 1");
                if (KIND == 5) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(t[i & 3]) : "v"(la), "i"((q * 1024) & 0xFFFF));
                if (KIND == 6) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(t[i & 3].x) : "v"(la), "i"((q * 1024) & 0xFFFF));
                if (KIND == 7) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(iv[i & 7]) : "v"(iv[(i + 1) & 7]));
                if (KIND == 8) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(lds + wave * 256), 16, (threadIdx.x & 63) * 16 + ((q * 1024) & 0xFFFF), 0, 0, 0);
                if (KIND == 9) asm volatile("v_mov_b32 %0, %1" : "=v"(iv[i & 7]) : "v"(iv[(i + 1) & 7]));
            }
            }
        }
        if (KIND == 5 || KIND == 6) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (KIND == 8) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    for (int i = 0; i < 8; ++i) s += v[i] + iv[i];
    for (int i = 0; i < 4; ++i) s += pv[i].x + pv[i].y + t[i].x;
    out[blockIdx.x * THREADS + threadIdx.x] = s;
}
static const char* NAMES[] = {"nothing", "v_fma_f32", "v_pk_add_f32", "v_add_u32", "s_nop 1", "ds_read_b128", "ds_read_b32", "v_xor_b32", "LDS-DMA 1 KiB", "v_mov_b32"};
template <int THREADS, int KIND, int NV, int PER>
void run(float* out, const float* src, double base) {
    const int n = 40000 * 256 / THREADS;
    for (int rep = 0; rep < 2; ++rep) {
        auto t0 = std::chrono::high_resolution_clock::now();
        k<THREADS, KIND, NV, PER><<<256, THREADS>>>(out, src, n);
        hipError_t e = hipDeviceSynchronize(); hipError_t e2 = hipGetLastError();
        if (e != hipSuccess || e2 != hipSuccess) { printf("launch failed: %s / %s\n", hipGetErrorString(e), hipGetErrorString(e2)); return; }
        double sec = std::chrono::duration<double>(std::chrono::high_resolution_clock::now() - t0).count();
        double tf = 256.0 * (THREADS / 64) * n * 16.0 * 4096 / sec / 1e12;
        if (rep) printf("waves/SIMD %d  %d x %-14s per %d MFMA: %6.1f TFLOP/s  -> %5.1f cycles per instruction (64-cycle MFMAs at %.1f TF)\n", THREADS / 256, NV, NAMES[KIND], PER, tf,
                        NV ? (base / tf - 1.0) * 64.0 * PER / NV : 0.0, base);
    }
}
int main() {
    float* out; float* src; hipMalloc(&out, 1 << 22); hipMalloc(&src, 1 << 20); hipMemset(src, 0, 1 << 20);
    const double B = 154.7;
    run<768, 0, 0, 1>(out, src, B);
    run<768, 1, 2, 1>(out, src, B); run<768, 2, 2, 1>(out, src, B); run<768, 3, 2, 1>(out, src, B); run<768, 4, 2, 1>(out, src, B); run<768, 7, 2, 1>(out, src, B); run<768, 9, 2, 1>(out, src, B);
    run<768, 5, 1, 1>(out, src, B); run<768, 5, 1, 2>(out, src, B); run<768, 6, 1, 1>(out, src, B); run<768, 6, 2, 1>(out, src, B);
    run<768, 8, 1, 4>(out, src, B); run<768, 8, 1, 8>(out, src, B); run<768, 8, 1, 2>(out, src, B);
    run<256, 2, 2, 1>(out, src, B); run<256, 5, 1, 1>(out, src, B); run<256, 8, 1, 4>(out, src, B);
    return 0;
}
