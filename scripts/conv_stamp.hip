// Diagnostic build (never shipped): the forward conv kernel of conv.hip with s_memtime stamps around the phases of one
// K-chunk, to see where a workgroup's time goes.  usage: conv_stamp <N>
#define VOCR_CONV_STAMPS 1
#include "../vistaocr_amd/csrc/conv.hip"
#include <vector>
#include <cstdio>
#include <cstdlib>
void vocr_set_error(const char*, ...) {}
int main(int argc, char** argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 1, Cin = 256, Cout = 256, H = 7, W = 294;
    float *x, *wp, *y; unsigned long long* dbg;
    hipMalloc(&x, (size_t)N * Cin * H * W * 4); hipMalloc(&wp, (size_t)Cin * 9 * Cout * 4); hipMalloc(&y, (size_t)N * Cout * H * W * 4);
    hipMalloc(&dbg, 8 * 8 * 4096); hipMemset(dbg, 0, 8 * 8 * 4096);
    hipMemset(x, 0x3c, (size_t)N * Cin * H * W * 4); hipMemset(wp, 0x3c, (size_t)Cin * 9 * Cout * 4);
    g_stamp_out = nullptr;
    hipMemcpyToSymbol(HIP_SYMBOL(g_stamp_out), &dbg, sizeof(dbg));
    for (int rep = 0; rep < 3; ++rep) {
        vocr_conv3x3_fwd(x, wp, nullptr, y, N, Cin, H, W, Cout, nullptr);
        hipDeviceSynchronize();
    }
    std::vector<unsigned long long> h(8 * 4096);
    hipMemcpy(h.data(), dbg, 8 * 8 * 4096, hipMemcpyDeviceToHost);
    const int wgs = N * 65;
    double s[8] = {0};
    int n = 0;
    for (int b = 0; b < wgs && b < 4096; ++b) { for (int k = 0; k < 8; ++k) s[k] += h[b * 8 + k]; ++n; }
    printf("N=%d WGs=%d: per workgroup (wave 0) cycles: total %.0f | k-loop (MFMA) %.0f | barriers %.0f | patch store %.0f | DMA+load issue %.0f | prologue %.0f | epilogue %.0f\n", N, wgs,
           s[0] / n, s[1] / n, s[2] / n, s[3] / n, s[4] / n, s[5] / n, s[6] / n);
    return 0;
}
