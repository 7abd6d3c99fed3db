// Diagnostic build (never shipped): the 4-row forward chain sweep (lstm_fwd_chain4v; 16 < B <= 32: lstm_fwd_chain4w unless VOCR_LSTM_SWEEP_FWD=chain4) with s_memtime stamps around the phases
// of a time step.  usage: lstm_stamp4 [B]
#define VOCR_LSTM_STAMPS 1
#include "../vistaocr_amd/csrc/lstm.hip"
#include <vector>
#include <cstdio>
#include <cstdlib>
void vocr_set_error(const char*, ...) {}
extern "C" int vocr_colsum(const float*, float*, int, int, void*, void*) { return 0; }
int main(int argc, char** argv) {
    const int T = 294, B = argc > 1 ? atoi(argv[1]) : 32, H = 512;
    float *xproj, *wf, *wr, *y, *gates, *cell; int32_t* lens; void* ws; unsigned long long* dbg;
    hipMalloc(&xproj, (size_t)2 * T * B * 4 * H * 4); hipMalloc(&wf, (size_t)4 * H * H * 4); hipMalloc(&wr, (size_t)4 * H * H * 4);
    hipMalloc(&y, (size_t)T * B * 2 * H * 4); hipMalloc(&gates, (size_t)2 * T * B * 4 * H * 4); hipMalloc(&cell, (size_t)2 * T * B * H * 4);
    hipMalloc(&lens, B * 4); hipMalloc(&ws, vocr_lstm_workspace_bytes(T, B, H) + 256); hipMalloc(&dbg, 512 * 2 * 8 * 8);
    hipMemset(xproj, 0, (size_t)2 * T * B * 4 * H * 4); hipMemset(wf, 0, (size_t)4 * H * H * 4); hipMemset(wr, 0, (size_t)4 * H * H * 4);
    hipMemset(dbg, 0, 512 * 2 * 8 * 8);
    std::vector<int32_t> hl(B, T); hipMemcpy(lens, hl.data(), B * 4, hipMemcpyHostToDevice);
    hipMemcpyToSymbol(HIP_SYMBOL(g_lstm_stamp_out), &dbg, sizeof(dbg));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0, nullptr);
        int rc = vocr_lstm_fwd(xproj, wf, wr, lens, y, gates, cell, ws, T, B, H, nullptr, nullptr);
        hipEventRecord(e1, nullptr); hipDeviceSynchronize(); hipEventElapsedTime(&ms, e0, e1);
        if (rc) { printf("rc %d\n", rc); return 1; }
    }
    const bool four = getenv("VOCR_LSTM_SWEEP_FWD") && !strcmp(getenv("VOCR_LSTM_SWEEP_FWD"), "chain4");      // lstm_fwd_chain4v (two workgroups per CU) also above B = 16
    const bool eight = B > 16 && !four;             // lstm_fwd_chain4w: 8 waves, stamps of waves 0 and 7
    const int nwg = eight ? 256 : B > 16 ? 512 : 256;
    std::vector<unsigned long long> h(512 * 2 * 8);
    hipMemcpy(h.data(), dbg, 512 * 2 * 8 * 8, hipMemcpyDeviceToHost);
    const char* names[8] = {"h slice arrived (poll loop)", "MFMA + partial tile to LDS", "barrier A", "reduce + activation", "barrier B", "cell update + stores issued",
                            "-", "loop top"};
    printf("B = %d: sweep %.3f ms = %.2f us per step\n", B, ms, ms * 1e3 / T);
    for (int w = 0; w < 2; ++w) {
        double s[8] = {0}, tot = 0;
        for (int b = 0; b < nwg; ++b) for (int k = 0; k < 8; ++k) s[k] += (double)h[(b * 2 + w) * 8 + k] / nwg / T;
        for (int k = 0; k < 8; ++k) if (k != 6) tot += s[k];
        printf("wave %d, s_memtime ticks per step (total %.1f => one tick = %.2f ns):\n", w ? (eight ? 7 : 3) : 0, tot, ms * 1e6 / T / tot);
        for (int k = 0; k < 8; ++k) printf("   %-38s %8.1f  %5.1f %%\n", names[k], s[k], 100 * s[k] / tot);
    }
    if (eight) {           // ---- backward sweep (lstm_bwd_chain4w): stamps land behind the forward ones
        float *dyb, *wtf, *wtr, *dgt;
        hipMalloc(&dyb, (size_t)T * B * 2 * H * 4); hipMalloc(&wtf, (size_t)4 * H * H * 4); hipMalloc(&wtr, (size_t)4 * H * H * 4); hipMalloc(&dgt, (size_t)2 * T * B * 4 * H * 4);
        hipMemset(dyb, 0, (size_t)T * B * 2 * H * 4); hipMemset(wtf, 0, (size_t)4 * H * H * 4); hipMemset(wtr, 0, (size_t)4 * H * H * 4);
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0, nullptr);
            int rc = vocr_lstm_bwd(dyb, wtf, wtr, lens, gates, cell, dgt, ws, T, B, H, nullptr, nullptr);
            hipEventRecord(e1, nullptr); hipDeviceSynchronize(); hipEventElapsedTime(&ms, e0, e1);
            if (rc) { printf("bwd rc %d\n", rc); return 1; }
        }
        std::vector<unsigned long long> hb(512 * 2 * 8);
        hipMemcpy(hb.data(), dbg, 512 * 2 * 8 * 8, hipMemcpyDeviceToHost);
        const char* bn[8] = {"partial blocks arrived (polls; waves 0, 1)", "sum + cell gradient + dgates out", "barrier", "LDS fragments + MFMA + partial stores issued",
                             "-", "-", "-", "resets + loop top"};
        printf("backward sweep %.3f ms = %.2f us per step\n", ms, ms * 1e3 / T);
        for (int w = 0; w < 2; ++w) {
            double s2[8] = {0}, tot = 0;
            for (int b = 0; b < 256; ++b) for (int k = 0; k < 8; ++k) s2[k] += (double)hb[((256 + b) * 2 + w) * 8 + k] / 256 / T;
            for (int k = 0; k < 8; ++k) tot += s2[k];
            printf("wave %d, s_memtime ticks per step (total %.1f):\n", w ? 7 : 0, tot);
            for (int k = 0; k < 8; ++k) printf("   %-46s %8.1f  %5.1f %%\n", bn[k], s2[k], 100 * s2[k] / tot);
        }
    }
    if (nwg == 512) {      // two workgroups per CU: do the two chains of an XCD (slot = bit 3 of the block index) progress alike?
        for (int slot = 0; slot < 2; ++slot) {
            double tot = 0, poll = 0; int n = 0;
            for (int b = 0; b < nwg; ++b) if (((b >> 3) & 1) == slot) {
                for (int k = 0; k < 8; ++k) if (k != 6) tot += (double)h[(b * 2) * 8 + k];
                poll += (double)h[(b * 2) * 8 + 0]; ++n;
            }
            printf("slot %d workgroups: wave 0 total %.0f ticks per sweep (%.1f per step), of which polling %.1f per step\n", slot, tot / n, tot / n / T, poll / n / T);
        }
    }
    return 0;
}
