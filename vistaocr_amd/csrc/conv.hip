// 3x3 "same" convolution (nn.Conv2d(k=3,pad=1), reference src/models/cnnlstm.py:118,264) as an implicit
// GEMM on the f32 MFMA (v_mfma_f32_32x32x2_f32): forward, data-gradient (same kernel, transposed/flipped
// weight pack) and weight-gradient.
//
// Work decomposition (NCHW, W contiguous): an image row is cut into 32-pixel SEGMENTS — one MFMA N-tile.
// forward/dgrad: M = output channels, N = pixels, K = (ci,kh,kw).  A workgroup (4 waves, one per SIMD)
//   owns CO_T channels x NSEG segments; per 8-input-channel K-chunk it stages in LDS the packed weights
//   Wt[72][CO_T] and, per segment, the halo patch P[8][3][34]; B fragments are read straight out of the
//   patch (lane j -> P[ci][kh][j+kw]), so im2col is never materialised.  Global loads for chunk c+1 are
//   issued before the 144 MFMAs of chunk c and land in registers.
// wgrad: M = co, N = ci, K = pixels; 9 taps share the dy fragment; each wave keeps 9 32x32 accumulators;
//   the pixel range is split over workgroups into slabs that a second kernel sums in a fixed order
//   (bitwise reproducible, no float atomics).
#include "vocr_common.h"

namespace {

constexpr int SEGW = 32;         // pixels per segment
constexpr int CI_C = 8;          // input channels per K-chunk
constexpr int KC = CI_C * 9;     // 72
constexpr int PROW = SEGW + 2;   // 34
constexpr int PCI = 3 * PROW;    // 102
constexpr int PSEG = CI_C * PCI; // 816

__global__ void pack_weights_kernel(const float* __restrict__ w, float* __restrict__ pf, float* __restrict__ pd,
                                    int cout, int cin) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int total = cout * cin * 9;
    if (i >= total) return;
    const int tap = i % 9, ci = (i / 9) % cin, co = i / (9 * cin);
    const float v = w[i];
    if (pf) pf[(long)(ci * 9 + tap) * cout + co] = v;
    if (pd) pd[(long)(co * 9 + (8 - tap)) * cin + ci] = v;   // (2-kh)*3+(2-kw) = 8 - tap
}

struct SegInfo { long base; int h; int w0; int valid; };

template <int CO_T>
__global__ __launch_bounds__(256) void conv3x3_kernel(const float* __restrict__ in, const float* __restrict__ wpack,
                                                      const float* __restrict__ bias, float* __restrict__ out, int N,
                                                      int Cin, int H, int W, int Cout, int SW, int nseg_total) {
    constexpr int WAVES_CO = CO_T / 64;
    constexpr int WAVES_PX = 4 / WAVES_CO;
    constexpr int NSEG = WAVES_PX * 2;
    constexpr int EA = KC * CO_T / 256;                    // 36 or 18
    constexpr int EP = (NSEG * PSEG + 255) / 256;          // 13 or 26
    __shared__ float Wt[KC * CO_T];
    __shared__ float P[NSEG * PSEG];
    __shared__ SegInfo segs[NSEG];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lk = lane >> 5;
    const int co0 = blockIdx.y * CO_T;
    const int seg0 = blockIdx.x * NSEG;
    const long HW = (long)H * W;
    if (tid < NSEG) {
        const int g = seg0 + tid;
        SegInfo s;
        s.valid = g < nseg_total;
        const int gg = s.valid ? g : 0;
        const int n = gg / (H * SW), rem = gg % (H * SW);
        s.h = rem / SW;
        s.w0 = (rem % SW) * SEGW;
        s.base = (long)n * Cin * HW;
        segs[tid] = s;
    }
    __syncthreads();

    const int wco = (wave / WAVES_PX) * 64;
    const int wsg = (wave % WAVES_PX) * 2;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    float ra[EA], rp[EP];
    const int Ktot = Cin * 9;

    auto load_chunk = [&](int ci0) {
#pragma unroll
        for (int e = 0; e < EA; ++e) {
            const int idx = tid + 256 * e;
            const int kk = idx / CO_T, co = idx % CO_T;
            const int gk = ci0 * 9 + kk, gco = co0 + co;
            ra[e] = (gk < Ktot && gco < Cout) ? wpack[(long)gk * Cout + gco] : 0.f;
        }
#pragma unroll
        for (int e = 0; e < EP; ++e) {
            const int idx = tid + 256 * e;
            float v = 0.f;
            if (idx < NSEG * PSEG) {
                const int sg = idx / PSEG, r1 = idx % PSEG;
                const int ci = r1 / PCI, r2 = r1 % PCI;
                const int kh = r2 / PROW, col = r2 % PROW;
                const SegInfo s = segs[sg];
                const int hh = s.h + kh - 1, ww = s.w0 + col - 1, gci = ci0 + ci;
                if (s.valid && gci < Cin && hh >= 0 && hh < H && ww >= 0 && ww < W)
                    v = in[s.base + (long)gci * HW + (long)hh * W + ww];
            }
            rp[e] = v;
        }
    };
    auto store_chunk = [&]() {
#pragma unroll
        for (int e = 0; e < EA; ++e) Wt[tid + 256 * e] = ra[e];
#pragma unroll
        for (int e = 0; e < EP; ++e) {
            const int idx = tid + 256 * e;
            if (idx < NSEG * PSEG) P[idx] = rp[e];
        }
    };

    const int nchunks = (Cin + CI_C - 1) / CI_C;
    load_chunk(0);
    store_chunk();
    __syncthreads();
    for (int c = 0; c < nchunks; ++c) {
        if (c + 1 < nchunks) load_chunk((c + 1) * CI_C);
        const float* wa = Wt + wco + li;
        const float* pb0 = P + (wsg + 0) * PSEG + li;
        const float* pb1 = P + (wsg + 1) * PSEG + li;
#pragma unroll
        for (int ks = 0; ks < KC / 2; ++ks) {
            const int k0 = 2 * ks, k1 = 2 * ks + 1;
            const int o0 = (k0 / 9) * PCI + ((k0 % 9) / 3) * PROW + (k0 % 3);
            const int o1 = (k1 / 9) * PCI + ((k1 % 9) / 3) * PROW + (k1 % 3);
            const int off = lk ? o1 : o0;
            const int kr = lk ? k1 : k0;
            const float a0 = wa[kr * CO_T], a1 = wa[kr * CO_T + 32];
            const float b0 = pb0[off], b1 = pb1[off];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
        __syncthreads();
        if (c + 1 < nchunks) {
            store_chunk();
            __syncthreads();
        }
    }

#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const SegInfo s = segs[wsg + j];
        const int ww = s.w0 + li;
        if (!s.valid || ww >= W) continue;
        const long obase = (s.base / Cin) * Cout + (long)s.h * W + ww;   // n*Cout*HW + h*W + w
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co0 + wco + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
                if (co < Cout) out[obase + (long)co * HW] = acc[i][j][r] + (bias ? bias[co] : 0.f);
            }
    }
}

// ---------------------------------------------------------------- weight gradient
constexpr int WG_DYP = SEGW + 1;    // 33: odd pitch -> conflict-free reads across channels
constexpr int WG_XCI = 3 * PROW + 1;  // 103

__global__ __launch_bounds__(256) void conv3x3_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                            float* __restrict__ slab, int N, int Cin, int H, int W,
                                                            int Cout, int SW, int nseg_total, int segs_per_split) {
    __shared__ float dyT[64 * WG_DYP];
    __shared__ float xp[64 * WG_XCI];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lk = lane >> 5;
    const int ci0 = blockIdx.x * 64, co0 = blockIdx.y * 64, split = blockIdx.z;
    const int wco = (wave >> 1) * 32, wci = (wave & 1) * 32;
    const long HW = (long)H * W;

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    constexpr int EDY = 64 * SEGW / 256;               // 8
    constexpr int EX = (64 * 3 * PROW + 255) / 256;    // 26
    float rdy[EDY], rx[EX];

    const int sbeg = split * segs_per_split;
    const int send = min(nseg_total, sbeg + segs_per_split);

    auto load_seg = [&](int g) {
        const int n = g / (H * SW), rem = g % (H * SW);
        const int h = rem / SW, w0 = (rem % SW) * SEGW;
#pragma unroll
        for (int e = 0; e < EDY; ++e) {
            const int idx = tid + 256 * e;
            const int co = idx / SEGW, px = idx % SEGW;
            const int gco = co0 + co, ww = w0 + px;
            rdy[e] = (gco < Cout && ww < W) ? dy[((long)n * Cout + gco) * HW + (long)h * W + ww] : 0.f;
        }
#pragma unroll
        for (int e = 0; e < EX; ++e) {
            const int idx = tid + 256 * e;
            float v = 0.f;
            if (idx < 64 * 3 * PROW) {
                const int ci = idx / (3 * PROW), r2 = idx % (3 * PROW);
                const int kh = r2 / PROW, col = r2 % PROW;
                const int gci = ci0 + ci, hh = h + kh - 1, ww = w0 + col - 1;
                if (gci < Cin && hh >= 0 && hh < H && ww >= 0 && ww < W) v = x[((long)n * Cin + gci) * HW + (long)hh * W + ww];
            }
            rx[e] = v;
        }
    };
    auto store_seg = [&]() {
#pragma unroll
        for (int e = 0; e < EDY; ++e) {
            const int idx = tid + 256 * e;
            dyT[(idx / SEGW) * WG_DYP + (idx % SEGW)] = rdy[e];
        }
#pragma unroll
        for (int e = 0; e < EX; ++e) {
            const int idx = tid + 256 * e;
            if (idx < 64 * 3 * PROW) xp[(idx / (3 * PROW)) * WG_XCI + (idx % (3 * PROW))] = rx[e];
        }
    };

    if (sbeg < send) {
        load_seg(sbeg);
        store_seg();
    }
    __syncthreads();
    for (int g = sbeg; g < send; ++g) {
        if (g + 1 < send) load_seg(g + 1);
        const float* ap = dyT + (wco + li) * WG_DYP + lk;
        const float* bp = xp + (wci + li) * WG_XCI + lk;
#pragma unroll
        for (int ks = 0; ks < SEGW / 2; ++ks) {
            const float a = ap[2 * ks];
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const float b = bp[(t / 3) * PROW + (t % 3) + 2 * ks];
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
            }
        }
        __syncthreads();
        if (g + 1 < send) {
            store_seg();
            __syncthreads();
        }
    }
    // slab[split][tap][co][ci]  (ci contiguous -> coalesced stores)
    const long plane = (long)Cout * Cin;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = co0 + wco + (r & 3) + 8 * (r >> 2) + 4 * lk;
            const int ci = ci0 + wci + li;
            if (co < Cout && ci < Cin) slab[((long)split * 9 + t) * plane + (long)co * Cin + ci] = acc[t][r];
        }
}

__global__ void wgrad_reduce_kernel(const float* __restrict__ slab, float* __restrict__ dw, int Cout, int Cin, int splits) {
    const long plane = (long)Cout * Cin;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;   // index over [tap][co][ci]
    if (i >= 9 * plane) return;
    const int t = (int)(i / plane);
    const long r = i % plane;
    float s = 0.f;
    for (int sp = 0; sp < splits; ++sp) s += slab[((long)sp * 9 + t) * plane + r];
    dw[r * 9 + t] = s;
}

// per-channel sum over (n, h*w): grid (C, chunks); double accumulation inside a workgroup, float atomics across
__global__ __launch_bounds__(256) void channel_sum_kernel(const float* __restrict__ x, float* __restrict__ out, int N, int C,
                                                          long HW, int nchunk) {
    __shared__ double red[4];
    const int c = blockIdx.x, j = blockIdx.y;
    const long total = (long)N * HW;
    const long per = (total + nchunk - 1) / nchunk;
    const long beg = j * per, end = min(total, beg + per);
    double s = 0.0;
    for (long e = beg + threadIdx.x; e < end; e += 256) {
        const long n = e / HW, r = e - n * HW;
        s += (double)x[(n * C + c) * HW + r];
    }
    s = wave_sum_d(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float v = (float)((red[0] + red[1]) + (red[2] + red[3]));
        if (nchunk > 1) atomicAdd(out + c, v); else out[c] = v;
    }
}

int wgrad_splits(int n, int cin, int h, int w, int cout, int* segs_per_split) {
    const int SW = vocr_cdiv(w, SEGW);
    const long nseg = (long)n * h * SW;
    const int tiles = vocr_cdiv(cin, 64) * vocr_cdiv(cout, 64);
    long s = (768 + tiles - 1) / tiles;
    if (s > nseg) s = nseg;
    if (s < 1) s = 1;
    const int sps = (int)((nseg + s - 1) / s);
    *segs_per_split = sps;
    return (int)((nseg + sps - 1) / sps);
}

}  // namespace

extern "C" int vocr_conv3x3_pack_weights(const float* w, float* wpack_fwd, float* wpack_dgrad, int cout, int cin, void* stream) {
    VOCR_CHECK_ARG(w && (wpack_fwd || wpack_dgrad) && cout > 0 && cin > 0, "vocr_conv3x3_pack_weights: bad argument");
    const int total = cout * cin * 9;
    pack_weights_kernel<<<vocr_cdiv(total, 256), 256, 0, (hipStream_t)stream>>>(w, wpack_fwd, wpack_dgrad, cout, cin);
    VOCR_CHECK_LAUNCH("vocr_conv3x3_pack_weights");
    return VOCR_OK;
}

extern "C" int vocr_conv3x3_fwd(const float* x, const float* wpack, const float* bias, float* y, int n, int cin, int h,
                                int w, int cout, void* stream) {
    VOCR_CHECK_ARG(x && wpack && y, "vocr_conv3x3_fwd: null pointer");
    VOCR_CHECK_ARG(n > 0 && cin > 0 && h > 0 && w > 0 && cout > 0, "vocr_conv3x3_fwd: bad shape");
    const int SW = vocr_cdiv(w, SEGW);
    const long nseg = (long)n * h * SW;
    VOCR_CHECK_ARG(nseg < (1l << 30), "vocr_conv3x3_fwd: too many segments");
    hipStream_t s = (hipStream_t)stream;
    if (cout > 64) {
        dim3 grid(vocr_cdiv(nseg, 4), vocr_cdiv(cout, 128));
        conv3x3_kernel<128><<<grid, 256, 0, s>>>(x, wpack, bias, y, n, cin, h, w, cout, SW, (int)nseg);
    } else {
        dim3 grid(vocr_cdiv(nseg, 8), 1);
        conv3x3_kernel<64><<<grid, 256, 0, s>>>(x, wpack, bias, y, n, cin, h, w, cout, SW, (int)nseg);
    }
    VOCR_CHECK_LAUNCH("vocr_conv3x3_fwd");
    return VOCR_OK;
}

extern "C" size_t vocr_conv3x3_wgrad_workspace_bytes(int n, int cin, int h, int w, int cout) {
    if (n <= 0 || cin <= 0 || h <= 0 || w <= 0 || cout <= 0) return 0;
    int sps;
    const int splits = wgrad_splits(n, cin, h, w, cout, &sps);
    return (size_t)splits * 9 * cout * cin * sizeof(float);
}

extern "C" int vocr_conv3x3_wgrad(const float* x, const float* dy, float* dw, void* workspace, int n, int cin, int h,
                                  int w, int cout, void* stream) {
    VOCR_CHECK_ARG(x && dy && dw && workspace, "vocr_conv3x3_wgrad: null pointer");
    VOCR_CHECK_ARG(n > 0 && cin > 0 && h > 0 && w > 0 && cout > 0, "vocr_conv3x3_wgrad: bad shape");
    const int SW = vocr_cdiv(w, SEGW);
    const long nseg = (long)n * h * SW;
    int sps;
    const int splits = wgrad_splits(n, cin, h, w, cout, &sps);
    hipStream_t s = (hipStream_t)stream;
    dim3 grid(vocr_cdiv(cin, 64), vocr_cdiv(cout, 64), splits);
    conv3x3_wgrad_kernel<<<grid, 256, 0, s>>>(x, dy, (float*)workspace, n, cin, h, w, cout, SW, (int)nseg, sps);
    VOCR_CHECK_LAUNCH("vocr_conv3x3_wgrad");
    const long total = 9l * cout * cin;
    wgrad_reduce_kernel<<<vocr_cdiv(total, 256), 256, 0, s>>>((const float*)workspace, dw, cout, cin, splits);
    VOCR_CHECK_LAUNCH("vocr_conv3x3_wgrad(reduce)");
    return VOCR_OK;
}

extern "C" int vocr_channel_sum(const float* x, float* out, int n, int c, int hw, void* stream) {
    VOCR_CHECK_ARG(x && out && n > 0 && c > 0 && hw > 0, "vocr_channel_sum: bad argument");
    hipStream_t s = (hipStream_t)stream;
    long nchunk = ((long)n * hw + 16383) / 16384;
    if (nchunk > 64) nchunk = 64;
    if (nchunk < 1) nchunk = 1;
    if (nchunk > 1 && hipMemsetAsync(out, 0, (size_t)c * sizeof(float), s) != hipSuccess) {
        vocr_set_error("vocr_channel_sum: memset failed");
        return VOCR_ELAUNCH;
    }
    channel_sum_kernel<<<dim3(c, (unsigned)nchunk), 256, 0, s>>>(x, out, n, c, hw, (int)nchunk);
    VOCR_CHECK_LAUNCH("vocr_channel_sum");
    return VOCR_OK;
}
