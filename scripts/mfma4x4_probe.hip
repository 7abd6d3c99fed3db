// GPU box probe: operand/accumulator layout of v_mfma_f32_4x4x1_16b_f32 (16 blocks of D[4x4] += A[4x1] * B[1x4]).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const float* a, const float* b, float* d) {
    const int l = threadIdx.x;
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_4x4x1f32(a[l], b[l], c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) d[l * 4 + r] = c[r];
}
int main() {
    float ha[64], hb[64], hd[256], *a, *b, *d;
    for (int l = 0; l < 64; ++l) { ha[l] = 1.f + l; hb[l] = 100.f + l; }
    hipMalloc(&a, 256); hipMalloc(&b, 256); hipMalloc(&d, 1024);
    hipMemcpy(a, ha, 256, hipMemcpyHostToDevice); hipMemcpy(b, hb, 256, hipMemcpyHostToDevice);
    k<<<1, 64>>>(a, b, d); hipMemcpy(hd, d, 1024, hipMemcpyDeviceToHost);
    // hypothesis: lane 4*blk + j, register r holds A[4*blk + r] * B[4*blk + j]
    int bad = 0;
    for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) { const int blk = l >> 2, j = l & 3; const float want = ha[4 * blk + r] * hb[4 * blk + j]; if (hd[l * 4 + r] != want) ++bad; }
    printf("hypothesis D[lane 4b+j][reg r] = A[4b+r]*B[4b+j]: %s (%d mismatches)\n", bad ? "WRONG" : "ok", bad);
    int bad2 = 0;
    for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) { const int blk = l >> 2, j = l & 3; const float want = ha[4 * blk + j] * hb[4 * blk + r]; if (hd[l * 4 + r] != want) ++bad2; }
    printf("alternative D[lane 4b+j][reg r] = A[4b+j]*B[4b+r]: %s (%d mismatches)\n", bad2 ? "WRONG" : "ok", bad2);
    printf("lane0: %g %g %g %g  lane1: %g %g %g %g  lane4: %g %g %g %g\n", hd[0], hd[1], hd[2], hd[3], hd[4], hd[5], hd[6], hd[7], hd[16], hd[17], hd[18], hd[19]);
    return 0;
}
