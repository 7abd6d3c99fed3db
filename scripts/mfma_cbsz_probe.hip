// GPU box probe: the A-matrix broadcast controls (cbsz / abid) of v_mfma_f32_4x4x1_16b_f32.
// Hypothesis: with cbsz = n the 16 blocks form groups of 2^n consecutive blocks and every block of a group takes its A
// operand from the group's block number abid:  D[lane 4b+j][reg r] = A[4*((b & ~(2^n - 1)) + abid) + r] * B[4b+j].
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int CBSZ, int ABID>
__global__ void k(const float* a, const float* b, float* d) {
    const int l = threadIdx.x;
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_4x4x1f32(a[l], b[l], c, CBSZ, ABID, 0);
    for (int r = 0; r < 4; ++r) d[l * 4 + r] = c[r];
}
float ha[64], hb[64], hd[256], *a, *b, *d;
template <int CBSZ, int ABID>
void run() {
    k<CBSZ, ABID><<<1, 64>>>(a, b, d); hipMemcpy(hd, d, 1024, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) {
        const int blk = l >> 2, j = l & 3, src = (blk & ~((1 << CBSZ) - 1)) + ABID;
        if (hd[l * 4 + r] != ha[4 * src + r] * hb[4 * blk + j]) ++bad;
    }
    printf("cbsz %d abid %2d: %s (%d mismatches)  lane0 %g lane4 %g lane32 %g lane60 %g\n", CBSZ, ABID, bad ? "WRONG" : "ok", bad, hd[0], hd[16], hd[128], hd[240]);
}
int main() {
    for (int l = 0; l < 64; ++l) { ha[l] = 1.f + l; hb[l] = 100.f + l; }
    hipMalloc(&a, 256); hipMalloc(&b, 256); hipMalloc(&d, 1024);
    hipMemcpy(a, ha, 256, hipMemcpyHostToDevice); hipMemcpy(b, hb, 256, hipMemcpyHostToDevice);
    run<0, 0>(); run<4, 0>(); run<4, 5>(); run<4, 15>(); run<3, 0>(); run<3, 3>(); run<3, 7>(); run<1, 1>(); run<2, 2>();
    return 0;
}
