// The direct 3x3 convolution of ONE 32-channel x 32-position MFMA tile straight from global memory (no LDS staging), K split over
// the waves of the workgroup: the building block of the "tail pieces" that replace the last partial round of workgroup tiles of
// the staged kernels (conv.hip: conv3x3_dma_kernel, conv_wino.hip: conv3x3_wino_kernel).  wpack is the DIRECT pack
// [(ci*9 + kh*3 + kw)][co].  See conv.hip for the measurements behind it.
#pragma once
#include "vocr_common.h"

namespace {

// one MFMA N-tile of 32 lane positions: `rows` sub-rows of `pw` patch columns (ow outputs each) starting at image row h, column w0
struct SegInfo { long base; int n; int h; int w0; int valid; int rows; int pw; int ow; };

// Lane position li (both lane halves) computes output pixel (pn, ph, pwc) if pix_ok; `live` is wave-uniform (false: nothing to do).
template <int TAIL_WAVES>
__device__ __forceinline__ void conv3x3_tail_piece_px(float* __restrict__ red_, int pn, int ph, int pwc, bool pix_ok, bool live, int co_base,
                                                      const float* __restrict__ in, const float* __restrict__ wpack, const float* __restrict__ bias,
                                                      float* __restrict__ out, const float* __restrict__ zero_page, int Cin, int H, int W, int Cout) {
    float (*red)[16][64] = (float (*)[16][64])red_;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lk = lane >> 5;
    (void)li;
    if (!live || co_base >= Cout) return;                               // whole workgroup: no barrier is skipped by part of it
    const long HW = (long)H * W;
    // the nine taps of this lane's pixel: offsets inside a channel plane (clamped) and 0/1 masks
    int toff[9];
    float tm[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int hh = ph + t / 3 - 1, ww = pwc + t % 3 - 1;
        tm[t] = (pix_ok && hh >= 0 && hh < H && ww >= 0 && ww < W) ? 1.f : 0.f;
        toff[t] = min(max(hh, 0), H - 1) * W + min(max(ww, 0), W - 1);
    }
    const float* xin = in + (long)pn * Cin * HW;
    const int co = co_base + (lane & 31);
    const bool co_ok = co < Cout;
    const int nsteps = (Cin + 2 * TAIL_WAVES - 1) / (2 * TAIL_WAVES);
    constexpr int DEPTH = 3;
    float a[DEPTH][9], b[DEPTH][9];
    auto loads = [&](int s, float (&av)[9], float (&bv)[9]) {
        const int ci = 2 * (TAIL_WAVES * s + wave) + lk;
        const bool ok = ci < Cin && s < nsteps;
        const float* wrow = (ok && co_ok) ? wpack + (long)ci * 9 * Cout + co : zero_page;
        const long wstride = (ok && co_ok) ? Cout : 0;
        const float* xc = xin + (long)min(ci, Cin - 1) * HW;
        const float cm = ok ? 1.f : 0.f;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            av[t] = wrow[t * wstride];
            bv[t] = xc[toff[t]] * (tm[t] * cm);
        }
    };
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) loads(d, a[d], b[d]);
    for (int s0 = 0; s0 < nsteps; s0 += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            if (s0 + d < nsteps) {                                                       // wave-uniform
#pragma unroll
                for (int t = 0; t < 9; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[d][t], b[d][t], acc, 0, 0, 0);
                loads(s0 + d + DEPTH, a[d], b[d]);
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) red[wave][r][lane] = acc[r];
    __syncthreads();
    // thread (wave, lane) finishes 16 / TAIL_WAVES registers of lane's column
#pragma unroll
    for (int q = 0; q < 16 / TAIL_WAVES; ++q) {
        const int r = (16 / TAIL_WAVES) * wave + q;
        float sum = 0.f;
#pragma unroll
        for (int w8 = 0; w8 < TAIL_WAVES; ++w8) sum += red[w8][r][lane];
        const int oc = co_base + (r & 3) + 8 * (r >> 2) + 4 * lk;
        if (pix_ok && oc < Cout)
            out[(long)pn * Cout * HW + (long)oc * HW + (long)ph * W + pwc] = sum + (bias ? bias[oc] : 0.f);
    }
}

// the same for one segment of conv.hip's geometry: lane position li = sub-row li / pw, column li % pw
template <int TAIL_WAVES>
__device__ __forceinline__ void conv3x3_tail_piece_at(float* __restrict__ red_, const SegInfo& sg, int co_base, const float* __restrict__ in,
                                                      const float* __restrict__ wpack, const float* __restrict__ bias, float* __restrict__ out,
                                                      const float* __restrict__ zero_page, int Cin, int H, int W, int Cout) {
    const int li = threadIdx.x & 31;
    const int rr = li / sg.pw, cc = li - rr * sg.pw;
    conv3x3_tail_piece_px<TAIL_WAVES>(red_, sg.n, sg.h + rr, sg.w0 + cc, rr < sg.rows && cc < sg.ow, sg.valid != 0, co_base, in, wpack, bias, out,
                                      zero_page, Cin, H, W, Cout);
}

}  // namespace
