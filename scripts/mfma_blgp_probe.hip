// GPU box probe: the B-matrix lane-group pattern (blgp) of v_mfma_f32_4x4x1_16b_f32.
// Hypothesis: blgp = 4 + g replaces every 16-lane group of the B operand by group g:  B'[lane] = B[16g + (lane & 15)], so
// D[lane 4b+j][reg r] = A[4b+r] * B[16g + ((4b+j) & 15)];  blgp 1 / 2: lanes 0-31 / 32-63 to both halves; 3: rotate by 16 lanes.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int BLGP>
__global__ void k(const float* a, const float* b, float* d) {
    const int l = threadIdx.x;
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_4x4x1f32(a[l], b[l], c, 0, 0, BLGP);
    for (int r = 0; r < 4; ++r) d[l * 4 + r] = c[r];
}
float ha[64], hb[64], hd[256], *a, *b, *d;
template <int BLGP>
void run() {
    k<BLGP><<<1, 64>>>(a, b, d); hipMemcpy(hd, d, 1024, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) {
        const int blk = l >> 2;
        int src = l;
        if (BLGP >= 4) src = 16 * (BLGP - 4) + (l & 15);
        else if (BLGP == 1) src = l & 31;
        else if (BLGP == 2) src = 32 + (l & 31);
        else if (BLGP == 3) src = (l + 16) & 63;
        if (hd[l * 4 + r] != ha[4 * blk + r] * hb[src]) ++bad;
    }
    printf("blgp %d: %s (%d mismatches)  lane0 %g lane17 %g lane40 %g lane63 %g\n", BLGP, bad ? "WRONG" : "ok", bad, hd[0], hd[17 * 4], hd[40 * 4], hd[63 * 4]);
}
int main() {
    for (int l = 0; l < 64; ++l) { ha[l] = 1.f + l; hb[l] = 100.f + l; }
    hipMalloc(&a, 256); hipMalloc(&b, 256); hipMalloc(&d, 1024);
    hipMemcpy(a, ha, 256, hipMemcpyHostToDevice); hipMemcpy(b, hb, 256, hipMemcpyHostToDevice);
    run<0>(); run<1>(); run<2>(); run<3>(); run<4>(); run<5>(); run<6>(); run<7>();
    return 0;
}
