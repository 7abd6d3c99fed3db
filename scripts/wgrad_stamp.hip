// Diagnostic build (never shipped): the weight-gradient kernel of conv.hip with s_memtime stamps: k-loop vs everything else
// per segment for an MFMA wave, and issue / store / barrier for a loader wave (VOCR_WGRAD_MODE=0|1|2).
#define VOCR_CONV_STAMPS 1
#include "../vistaocr_amd/csrc/conv.hip"
#include <vector>
#include <cstdio>
#include <cstdlib>
void vocr_set_error(const char*, ...) {}
int main() {
    const int N = 32, Cin = 256, Cout = 256, H = 7, W = 294;
    float *x, *dy, *dw; void* ws; unsigned long long* dbg;
    hipMalloc(&x, (size_t)N * Cin * H * W * 4); hipMalloc(&dy, (size_t)N * Cout * H * W * 4); hipMalloc(&dw, (size_t)Cin * 9 * Cout * 4);
    hipMalloc(&ws, vocr_conv3x3_wgrad_workspace_bytes(N, Cin, H, W, Cout));
    hipMalloc(&dbg, 8 * 8 * 4096); hipMemset(dbg, 0, 8 * 8 * 4096);
    hipMemset(x, 0x3c, (size_t)N * Cin * H * W * 4); hipMemset(dy, 0x3c, (size_t)N * Cout * H * W * 4);
    hipMemcpyToSymbol(HIP_SYMBOL(g_stamp_out), &dbg, sizeof(dbg));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1); float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0, nullptr);
        int rc = vocr_conv3x3_wgrad(x, dy, dw, ws, N, Cin, H, W, Cout, nullptr);
        hipEventRecord(e1, nullptr); hipDeviceSynchronize(); hipEventElapsedTime(&ms, e0, e1);
        if (rc) { printf("rc %d\n", rc); return 1; }
    }
    std::vector<unsigned long long> h(8 * 64);
    hipMemcpy(h.data(), dbg, 8 * 64 * 8, hipMemcpyDeviceToHost);
    printf("wgrad 256->256 @ 32x7x294: %.1f us (incl. reduce)\n", ms * 1e3);
    for (int z = 0; z < 2; ++z) {
        const double n = (double)h[z * 8 + 2];
        printf("  split %d MFMA wave 0: %.0f segments; per segment: k-loop %.0f cycles, barrier(s)+staging %.0f\n", z, n, h[z * 8] / n, h[z * 8 + 1] / n);
        const double m = (double)h[(8 + z) * 8 + 3];
        if (m > 0) printf("  split %d loader wave 4: per segment: address arithmetic + DMA issue %.0f, DMA landing wait %.0f, barrier %.0f\n", z,
                          h[(8 + z) * 8] / m, h[(8 + z) * 8 + 1] / m, h[(8 + z) * 8 + 2] / m);
    }
    return 0;
}
