// Bidirectional LSTM recurrence on a packed (length-sorted, zero-padded) batch — nn.LSTM semantics
// (reference src/models/cnnlstm.py:148-149,288-290): gate order i,f,g,o; the reverse direction starts at
// each sequence's own last frame; outputs past a sequence's length are zero.
//
// The time-parallel half (x W_ih^T + biases for all T) is a vocr_gemm call made by the host; this file is
// the sequential half.  One launch per time step covers BOTH directions:
//   forward step : gates[b, 4 units x 4 gates] = h_{t-1}[b,:] . W_hh^T  — a [B x H] x [H x 16] product per
//                  workgroup on v_mfma_f32_16x16x4_f32, K split over the 4 waves (one per SIMD) and reduced
//                  through LDS, then the cell update for the 4 owned units is fused in the same kernel.
//   backward step: dh_{t}[b, 16 units] = dG_{t+1}[b,:] . W_hh  (K = 4H, one gate block per wave), fused with
//                  the gate-gradient computation of the 16 owned units.
// h / dh ping-pong between two small global buffers (the only cross-workgroup traffic, L2 resident).
#include "vocr_common.h"

namespace {

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + __expf(-x)); }

// grid.x = 2 * (H/4); 256 threads
__global__ __launch_bounds__(256) void lstm_fwd_step_kernel(const float* __restrict__ xproj, const float* __restrict__ whh_f,
                                                            const float* __restrict__ whh_r,
                                                            const int32_t* __restrict__ lens, float* __restrict__ y,
                                                            float* __restrict__ gates, float* __restrict__ cell,
                                                            const float* __restrict__ h_prev, float* __restrict__ h_next,
                                                            float* __restrict__ cbuf, int T, int B, int H, int step) {
    __shared__ float red[4][64][17];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ublocks = H >> 2;
    const int dir = blockIdx.x / ublocks, unit0 = (blockIdx.x % ublocks) * 4;
    const int t = dir == 0 ? step : T - 1 - step;
    const int RT = (B + 15) >> 4;
    const int lr = lane & 15, q = lane >> 4;
    const int KQ = H >> 4;                       // k values per lane: wave covers H/4, lane-quarter covers H/16
    const int kbase = wave * (H >> 2) + q * KQ;

    const float* hp = h_prev + (long)dir * B * H;
    const float* whh = dir ? whh_r : whh_f;
    const float* wrow = whh + ((long)(lr >> 2) * H + unit0 + (lr & 3)) * H + kbase;

    f32x4 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const float* ha[4];
    bool hv[4];
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) {
        const int b = rt * 16 + lr;
        hv[rt] = rt < RT && b < B;
        ha[rt] = hp + (long)(hv[rt] ? b : 0) * H + kbase;
    }
    if (step > 0) {   // h_{-1} = 0: the first step has no recurrent term
#pragma unroll 4
        for (int s = 0; s < KQ; ++s) {
            const float bw = wrow[s];
#pragma unroll
            for (int rt = 0; rt < 4; ++rt) {
                if (rt < RT) {
                    const float a = hv[rt] ? ha[rt][s] : 0.f;
                    acc[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bw, acc[rt], 0, 0, 0);
                }
            }
        }
    }
#pragma unroll
    for (int rt = 0; rt < 4; ++rt)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[wave][rt * 16 + q * 4 + r][lr] = acc[rt][r];
    __syncthreads();

    if (tid < B * 4) {
        const int b = tid >> 2, u = tid & 3, unit = unit0 + u;
        const bool active = t < lens[b];
        const long gbase = (((long)dir * T + t) * B + b) * 4 * H + unit;
        const long cidx = ((long)dir * B + b) * H + unit;
        const long sidx = (((long)dir * T + t) * B + b) * H + unit;
        float pre[4];
#pragma unroll
        for (int g = 0; g < 4; ++g)
            pre[g] = ((red[0][b][g * 4 + u] + red[1][b][g * 4 + u]) + (red[2][b][g * 4 + u] + red[3][b][g * 4 + u])) +
                     xproj[gbase + (long)g * H];
        const float hprev = step > 0 ? hp[(long)b * H + unit] : 0.f;
        float* hn = h_next + (long)dir * B * H + (long)b * H + unit;
        float* yo = y + ((long)t * B + b) * 2 * H + dir * H + unit;
        if (active) {
            const float ig = sigmoidf_(pre[0]), fg = sigmoidf_(pre[1]), gg = tanhf(pre[2]), og = sigmoidf_(pre[3]);
            const float cprev = step > 0 ? cbuf[cidx] : 0.f;
            const float c = fg * cprev + ig * gg;
            const float h = og * tanhf(c);
            gates[gbase] = ig;
            gates[gbase + H] = fg;
            gates[gbase + 2l * H] = gg;
            gates[gbase + 3l * H] = og;
            cell[sidx] = c;
            cbuf[cidx] = c;
            *hn = h;
            *yo = h;
        } else {
            gates[gbase] = 0.f;
            gates[gbase + H] = 0.f;
            gates[gbase + 2l * H] = 0.f;
            gates[gbase + 3l * H] = 0.f;
            cell[sidx] = 0.f;
            if (step == 0) cbuf[cidx] = 0.f;
            *hn = hprev;
            *yo = 0.f;
        }
    }
}

// grid.x = 2 * (H/16); 256 threads.  Backward iteration `step` visits t = T-1-step (forward dir) and
// t = step (reverse dir); dG of the previously visited time feeds dh through W_hh.
__global__ __launch_bounds__(256) void lstm_bwd_step_kernel(const float* __restrict__ dy, const float* __restrict__ whh_f,
                                                            const float* __restrict__ whh_r,
                                                            const int32_t* __restrict__ lens, const float* __restrict__ gates,
                                                            const float* __restrict__ cell, float* __restrict__ dgates,
                                                            float* __restrict__ dcbuf, int T, int B, int H, int step) {
    __shared__ float red[4][64][17];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ublocks = H >> 4;
    const int dir = blockIdx.x / ublocks, unit0 = (blockIdx.x % ublocks) * 16;
    const int t = dir == 0 ? T - 1 - step : step;
    const int tv = dir == 0 ? t + 1 : t - 1;      // time whose dG was produced by the previous iteration
    const int RT = (B + 15) >> 4;
    const int lr = lane & 15, q = lane >> 4;
    const int KQ = H >> 2;                        // wave = one gate block of H rows; lane-quarter covers H/4
    const int nbase = wave * H + q * KQ;

    f32x4 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};

    if (step > 0) {
        const float* dgp = dgates + (((long)dir * T + tv) * B) * 4 * H + nbase;
        const float* whh = dir ? whh_r : whh_f;
        const float* wp = whh + (long)nbase * H + unit0 + lr;
        const float* ga[4];
        bool gv[4];
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) {
            const int b = rt * 16 + lr;
            gv[rt] = rt < RT && b < B;
            ga[rt] = dgp + (long)(gv[rt] ? b : 0) * 4 * H;
        }
#pragma unroll 4
        for (int s = 0; s < KQ; ++s) {
            const float bw = wp[(long)s * H];
#pragma unroll
            for (int rt = 0; rt < 4; ++rt) {
                if (rt < RT) {
                    const float a = gv[rt] ? ga[rt][s] : 0.f;
                    acc[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bw, acc[rt], 0, 0, 0);
                }
            }
        }
    }
#pragma unroll
    for (int rt = 0; rt < 4; ++rt)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[wave][rt * 16 + q * 4 + r][lr] = acc[rt][r];
    __syncthreads();

    for (int cidx_l = tid; cidx_l < B * 16; cidx_l += 256) {
        const int b = cidx_l >> 4, j = cidx_l & 15, unit = unit0 + j;
        const int len = lens[b];
        const long gbase = (((long)dir * T + t) * B + b) * 4 * H + unit;
        const long cb = ((long)dir * B + b) * H + unit;
        if (t >= len) {
            dgates[gbase] = 0.f;
            dgates[gbase + H] = 0.f;
            dgates[gbase + 2l * H] = 0.f;
            dgates[gbase + 3l * H] = 0.f;
            dcbuf[cb] = 0.f;
            continue;
        }
        const float dh = dy[((long)t * B + b) * 2 * H + dir * H + unit] +
                         ((red[0][b][j] + red[1][b][j]) + (red[2][b][j] + red[3][b][j]));
        const float ig = gates[gbase], fg = gates[gbase + H], gg = gates[gbase + 2l * H], og = gates[gbase + 3l * H];
        const long sidx = (((long)dir * T + t) * B + b) * H + unit;
        const float c = cell[sidx];
        const int tp = dir == 0 ? t - 1 : t + 1;   // previous step of the recurrence
        const float cprev = (tp >= 0 && tp < len) ? cell[(((long)dir * T + tp) * B + b) * H + unit] : 0.f;
        const float tc = tanhf(c);
        const float dcar = step > 0 ? dcbuf[cb] : 0.f;
        const float dc = dcar + dh * og * (1.f - tc * tc);
        dgates[gbase] = dc * gg * ig * (1.f - ig);
        dgates[gbase + H] = dc * cprev * fg * (1.f - fg);
        dgates[gbase + 2l * H] = dc * ig * (1.f - gg * gg);
        dgates[gbase + 3l * H] = dh * tc * og * (1.f - og);
        dcbuf[cb] = dc * fg;
    }
}

}  // namespace

extern "C" size_t vocr_lstm_workspace_bytes(int t, int b, int h) {
    if (t <= 0 || b <= 0 || h <= 0) return 0;
    return (size_t)6 * 2 * b * h * sizeof(float);   // h ping-pong (2 x [2][B][H]) + c ([2][B][H]) + spare
}

extern "C" int vocr_lstm_fwd(const float* xproj, const float* whh_fwd, const float* whh_rev, const int32_t* lens, float* y,
                             float* gates, float* cell, void* workspace, int t, int b, int h, void* stream) {
    VOCR_CHECK_ARG(xproj && whh_fwd && whh_rev && lens && y && gates && cell && workspace, "vocr_lstm_fwd: null pointer");
    VOCR_CHECK_ARG(t > 0 && b > 0 && b <= 64 && h > 0 && h % 16 == 0, "vocr_lstm_fwd: need 1<=B<=64 and H%%16==0 (B=%d H=%d)", b, h);
    hipStream_t s = (hipStream_t)stream;
    float* ws = (float*)workspace;
    const size_t st = (size_t)2 * b * h;
    float* hb[2] = {ws, ws + st};
    float* cb = ws + 2 * st;
    for (int step = 0; step < t; ++step) {
        lstm_fwd_step_kernel<<<2 * (h / 4), 256, 0, s>>>(xproj, whh_fwd, whh_rev, lens, y, gates, cell, hb[step & 1], hb[(step + 1) & 1], cb,
                                                         t, b, h, step);
    }
    VOCR_CHECK_LAUNCH("vocr_lstm_fwd");
    return VOCR_OK;
}

extern "C" int vocr_lstm_bwd(const float* dy, const float* whh_fwd, const float* whh_rev, const int32_t* lens,
                             const float* gates, const float* cell, float* dgates, void* workspace, int t, int b, int h,
                             void* stream) {
    VOCR_CHECK_ARG(dy && whh_fwd && whh_rev && lens && gates && cell && dgates && workspace, "vocr_lstm_bwd: null pointer");
    VOCR_CHECK_ARG(t > 0 && b > 0 && b <= 64 && h > 0 && h % 16 == 0, "vocr_lstm_bwd: need 1<=B<=64 and H%%16==0 (B=%d H=%d)", b, h);
    hipStream_t s = (hipStream_t)stream;
    float* dcb = (float*)workspace;
    for (int step = 0; step < t; ++step) {
        lstm_bwd_step_kernel<<<2 * (h / 16), 256, 0, s>>>(dy, whh_fwd, whh_rev, lens, gates, cell, dgates, dcb, t, b, h, step);
    }
    VOCR_CHECK_LAUNCH("vocr_lstm_bwd");
    return VOCR_OK;
}
