// GPU box: which LDS read of conv3x3_wgrad_wino2d_kernel's double step conflicts?  Each kernel issues ONE of its read patterns (same
// per-lane addresses as the kernel: x / dy planes [64 channels][8 positions][4 floats], position of slot m of channel c = m ^ ((c >> 1) & 7))
// 4 x 4096 times from 8 waves; rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS gives cycles per instruction.
//   k0 ds_read_b128 of the x piece (adC)            k1 ds_read_b32 of the left neighbour's last element (adP + 12)
//   k2 ds_read_b32 of the right neighbour's first   k3 j = 0 form of k1: lanes 0-31 from the extras (16-byte stride), 32-63 from the plane
//   k4 ds_read_b128 with channel rows 144 bytes apart instead of 128 (a padded layout, for comparison)
//   k5 the b32 reads on that padded layout
// hipcc --offload-arch=gfx950 -O3 -o /tmp/ldsp scripts/lds_conflict_probe.hip && rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS --kernel-trace ... -- /tmp/ldsp
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int KIND>
__global__ __launch_bounds__(512) void k(float* out, int n) {
    __shared__ __attribute__((aligned(256))) float lds[20480];
    for (int i = threadIdx.x; i < 20480; i += 512) lds[i] = i;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 31, lk = lane >> 5, cih = wave & 1;
    const int cB = cih * 32 + li, sw = (cB >> 1) & 7;
    unsigned ad[4];
    for (int j = 0; j < 4; ++j) {
        const int m = 2 * j + lk;
        if (KIND == 0) ad[j] = (cB * 32 + 4 * (m ^ sw)) * 4;
        if (KIND == 1) ad[j] = (cB * 32 + 4 * (((m - 1) & 7) ^ sw)) * 4 + 12;
        if (KIND == 2) ad[j] = (cB * 32 + 4 * (((m + 1) & 7) ^ sw)) * 4;
        if (KIND == 3) ad[j] = lk ? (cB * 32 + 4 * (((m - 1) & 7) ^ sw)) * 4 + 12 : (16384 + cB * 4) * 4 + 12;
        if (KIND == 4) ad[j] = (cB * 36 + 4 * m) * 4;
        if (KIND == 5) ad[j] = (cB * 36 + 4 * ((m + 7) & 7)) * 4 + 12;
    }
    const unsigned base = (unsigned)(size_t)lds;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < n; ++it) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (KIND == 0 || KIND == 4) {
                f32x4 t;
                asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(t) : "v"(base + ad[j]));
                acc += t;
            } else {
                float t;
                asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(t) : "v"(base + ad[j]));
                acc[0] += t;
            }
        }
    }
    out[blockIdx.x * 512 + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
}
int main() {
    float* out;
    hipMalloc(&out, 256 * 512 * 4);
    const int n = 4096;
    k<0><<<256, 512>>>(out, n); k<1><<<256, 512>>>(out, n); k<2><<<256, 512>>>(out, n);
    k<3><<<256, 512>>>(out, n); k<4><<<256, 512>>>(out, n); k<5><<<256, 512>>>(out, n);
    hipDeviceSynchronize();
    printf("done\n");
    return 0;
}
