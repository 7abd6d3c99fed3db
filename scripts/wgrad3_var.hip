// Diagnostic build (never shipped): conv3x3_wgrad_wino3_kernel WITHOUT stamps, parts cut at compile time (-DW3_CUT=bits, see
// conv_wino.hip; results are wrong, only the time matters), timed with events next to a pure-MFMA calibration loop of the same launch
// geometry.  hipcc --offload-arch=gfx950 -O3 -DW3_CUT=3 -o /tmp/w3v scripts/wgrad3_var.hip
#include "../vistaocr_amd/csrc/conv_wino.hip"
#include <vector>
#include <cstdio>
#include <cstdlib>
void vocr_set_error(const char*, ...) {}
int vocr_conv3x3_pack_weights(const float*, float*, float*, int, int, void*) { return 0; }
void vocr_internal_wgrad_reduce(const float*, float*, int, int, int, hipStream_t) {}
__global__ __launch_bounds__(768) void calib(float* out, int n) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a = threadIdx.x * 0.001f, b = blockIdx.x * 0.002f;
    for (int it = 0; it < n; ++it)
#pragma unroll
        for (int q = 0; q < 64; ++q) acc[q & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[q & 3], 0, 0, 0);
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 768 + threadIdx.x] = s;
}
int main(int argc, char** argv) {
    const int N = 32, Cin = argc > 1 ? atoi(argv[1]) : 256, Cout = argc > 2 ? atoi(argv[2]) : 256, H = argc > 3 ? atoi(argv[3]) : 7, W = argc > 4 ? atoi(argv[4]) : 294;
    float *x, *dy, *dw; void* ws;
    hipMalloc(&x, (size_t)N * Cin * H * W * 4); hipMalloc(&dy, (size_t)N * Cout * H * W * 4); hipMalloc(&dw, (size_t)Cin * 9 * Cout * 4);
    hipMalloc(&ws, vocr_conv3x3_wgrad_wino_workspace_bytes(N, Cin, H, W, Cout));
    std::vector<float> hx((size_t)N * Cin * H * W), hy((size_t)N * Cout * H * W);
    for (size_t i = 0; i < hx.size(); ++i) hx[i] = (float)((i * 2654435761u) >> 8 & 0xFFFF) / 65536.f - 0.5f;
    for (size_t i = 0; i < hy.size(); ++i) hy[i] = (float)((i * 40503u + 17) >> 4 & 0xFFFF) / 65536.f - 0.5f;
    hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dy, hy.data(), hy.size() * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1); float ms = 0, best = 1e9, cal = 1e9;
    const int nseg = (N * H * ((W + 3) / 4 + 1) + 15) / 16, tiles = ((Cin + 63) / 64) * ((Cout + 63) / 64), spl = (256 + tiles - 1) / tiles, sps = (nseg + spl - 1) / spl;
    for (int rep = 0; rep < 30; ++rep) {
        hipEventRecord(e0, nullptr);
        int rc = vocr_conv3x3_wgrad_wino(x, dy, dw, ws, N, Cin, H, W, Cout, nullptr);
        hipEventRecord(e1, nullptr); hipDeviceSynchronize(); hipEventElapsedTime(&ms, e0, e1);
        if (rc) { printf("rc %d\n", rc); return 1; }
        if (rep >= 10 && ms < best) best = ms;
    }
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0, nullptr);
        calib<<<256, 768>>>((float*)ws, sps);
        hipEventRecord(e1, nullptr); hipDeviceSynchronize(); hipEventElapsedTime(&ms, e0, e1);
        if (ms < cal) cal = ms;
    }
    printf("W3_CUT=%2d  wgrad %d->%d @ %dx%dx%d: %.1f us; the same MFMAs bare: %.1f us -> %.3f\n", W3_CUT, Cin, Cout, N, H, W, best * 1e3, cal * 1e3, cal / best);
    return 0;
}
