// 3x3 "same" convolution with the minimal-filtering transform F(2,3) ALONG THE ROW (Winograd 1-D), on the f32 MFMA.
// Replaces the same reference op as conv.hip (nn.Conv2d(k=3,pad=1), src/models/cnnlstm.py:118,264) for the layers with
// Cin >= 4; forward and data-gradient (same kernel, transposed / flipped weight pack).
//
// Two neighbouring outputs of a row need 4 multiplications per (ci, kh) instead of 6:
//   d0..d3 = input columns 2t-1 .. 2t+2            v0 = d0 - d2   v1 = d1 + d2   v2 = d2 - d1   v3 = d1 - d3
//   g0..g2 = the taps of filter row kh             u0 = g0   u1 = (g0+g1+g2)/2   u2 = (g0-g1+g2)/2   u3 = g2
//   m_x = sum over (ci, kh) of u_x * v_x           y(2t) = m0 + m1 + m2          y(2t+1) = m1 - m2 - m3
// i.e. four independent contractions over K = (ci, kh) whose N index is the column PAIR t: 12 MACs per pair and input
// channel instead of 18, so the matrix pipe does 2/3 of the direct kernel's work.  The input transform costs four vector
// adds per B fragment (the raw halo row sits in LDS split into even / odd columns, so d0..d3 are two conflict-free
// ds_read2), the output transform six adds per pair in the epilogue, the filter transform is part of the weight pack.
// Everything else follows conv.hip's LDS-DMA kernel: a workgroup (4 waves) owns CO_T output channels x NSEG segments
// of 32 pairs (64 pixels) of one image row each, K advances in half-chunks of 4 input channels (48 weight rows of the pack
// = one contiguous block that goes global -> LDS by DMA), two LDS buffers, workgroups in XCD-sliced order.
// Rounding: the transforms are sums of two or three fp32 values and one multiplication by 0.5 (exact); results differ
// from the direct kernel by the usual few ulp of a different summation order (tests/test_ops_gpu.py compares both with
// F.conv2d).
#include "vocr_common.h"
#include "conv_tail.h"

namespace {

constexpr int TS = 32;              // column pairs per segment
constexpr int CI_H = 4;             // input channels per half-chunk
constexpr int WR = CI_H * 12;       // weight rows (c, kh, x) per half-chunk
constexpr int PRW = 68;             // floats per staged halo row: E[0..32] at 0, O[0..32] at 34
constexpr int POFF = 34;            // offset of the odd columns inside a row
constexpr int PSEG = CI_H * 3 * PRW;   // floats of halo per segment and half-chunk

__device__ __attribute__((aligned(16))) float g_wino_zero_page[64];

__device__ __forceinline__ int xcd_slice_order(int b, int nwg) {
    const int x = b & 7, q = nwg >> 3, r = nwg & 7;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (b >> 3);
}

// pf[(ci*12 + kh*4 + x)][co], pd[(co*12 + kh*4 + x)][ci] (data gradient: taps flipped, channels transposed)
__global__ void wino_pack_kernel(const float* __restrict__ w, float* __restrict__ pf, float* __restrict__ pd, int cout, int cin) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int total = cout * cin * 3;
    if (i >= total) return;
    const int kh = i % 3, ci = (i / 3) % cin, co = i / (3 * cin);
    const float* g = w + ((long)(co * cin + ci) * 3 + kh) * 3;
    const float g0 = g[0], g1 = g[1], g2 = g[2];
    if (pf) {
        float* o = pf + (long)(ci * 12 + kh * 4) * cout + co;
        o[0] = g0;
        o[cout] = 0.5f * ((g0 + g2) + g1);
        o[2 * cout] = 0.5f * ((g0 + g2) - g1);
        o[3 * cout] = g2;
    }
    if (pd) {       // flipped filter: row 2-kh, taps (g2, g1, g0)
        float* o = pd + (long)(co * 12 + (2 - kh) * 4) * cin + ci;
        o[0] = g2;
        o[cin] = 0.5f * ((g2 + g0) + g1);
        o[2 * cin] = 0.5f * ((g2 + g0) - g1);
        o[3 * cin] = g0;
    }
}

struct WGeom { int nsr, per_img, nseg; };       // segments per image row, per image, in all

template <int CO_T>
__global__ __launch_bounds__(256, 2) void conv3x3_wino_kernel(const float* __restrict__ in, const float* __restrict__ wpack,
                                                           const float* __restrict__ bias, float* __restrict__ out,
                                                           const float* __restrict__ zero_page, int N, int Cin, int H, int W,
                                                           int Cout, WGeom geo, int co_tiles, const float* __restrict__ wdirect,
                                                           int n_tail, int first_tail_tile) {
    constexpr int WAVES_CO = CO_T / 64;                      // a wave owns 64 output channels (two MFMA row blocks) ...
    constexpr int NSEG = 4 / WAVES_CO;                       // ... of one segment
    constexpr int TM = 2;
    constexpr int WBUF = WR * CO_T;                          // floats per weight buffer
    constexpr int PBUF = NSEG * PSEG;                        // floats per halo buffer
    constexpr int ROWS_W = NSEG * 3;                         // halo rows a wave stages per half-chunk (NSEG*12 rows / 4 waves)
    constexpr int LPR = CO_T / 4;                            // lanes per weight row in one DMA
    constexpr int RPI = 64 / LPR;                            // weight rows per DMA instruction (2 or 4)
    constexpr int NDMA = WR / RPI;                           // DMA instructions per half-chunk (24 or 12)
    constexpr int DPW = NDMA / 4;                            // ... per wave
    // ONE LDS object: Wt[2][48][CO_T] | P[2][NSEG][4][3][68] | 64 dummy floats | NSEG x 8 ints of segment geometry
    __shared__ __attribute__((aligned(16))) float lds[2 * WBUF + 2 * PBUF + 64 + NSEG * 8];
    float* const Wt = lds;
    float* const P = lds + 2 * WBUF;
    constexpr int DUMMY = 2 * PBUF;
    int* const segw = (int*)(lds + 2 * WBUF + 2 * PBUF + 64);

    if ((int)blockIdx.x < n_tail) {
        // the last partial round of workgroup tiles, cut into 32-channel x 32-pixel pieces computed by the DIRECT form straight
        // from global memory (conv_tail.h; conv.hip explains why): a tile is (CO_T/32) channel blocks x 2*NSEG pixel blocks
        constexpr int COSUB = CO_T / 32, PPW = COSUB * 2 * NSEG;
        const int piece = blockIdx.x, vt = first_tail_tile + piece / PPW, sub = piece % PPW;
        const int g = (vt / co_tiles) * NSEG + (sub / COSUB) / 2;
        SegInfo sgi;
        sgi.valid = g < geo.nseg;
        const int gg = sgi.valid ? g : 0;
        sgi.n = gg / geo.per_img;
        const int loc = gg - sgi.n * geo.per_img;
        sgi.h = loc / geo.nsr;
        sgi.w0 = (loc - sgi.h * geo.nsr) * 2 * TS + ((sub / COSUB) & 1) * 32;
        sgi.rows = 1; sgi.pw = 34; sgi.ow = min(32, W - sgi.w0); sgi.base = 0;
        if (sgi.ow <= 0) sgi.valid = 0;
        conv3x3_tail_piece_at<4>(lds, sgi, (vt % co_tiles) * CO_T + (sub % COSUB) * 32, in, wdirect, bias, out, zero_page, Cin, H, W, Cout);
        return;
    }
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lk = lane >> 5;
    const int v = xcd_slice_order(blockIdx.x - n_tail, gridDim.x - n_tail);
    const int co0 = (v % co_tiles) * CO_T;
    const int seg0 = (v / co_tiles) * NSEG;
    const long HW = (long)H * W;
    if (tid < NSEG) {
        const int g = seg0 + tid;
        const int valid = g < geo.nseg;
        const int gg = valid ? g : 0;
        const int n = gg / geo.per_img, loc = gg - n * geo.per_img, h = loc / geo.nsr, w0 = (loc - h * geo.nsr) * 2 * TS;
        int* o = segw + tid * 8;
        o[0] = n; o[1] = h; o[2] = w0; o[3] = valid;
    }
    __syncthreads();

    const int wco = (wave / NSEG) * 64;
    const int wsg = wave % NSEG;
    f32x16 acc[TM][4];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int x = 0; x < 4; ++x)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][x][r] = 0.f;

    // ---- halo loader: this wave stages rows [wave*ROWS_W, (wave+1)*ROWS_W) of the (segment, c, kh) rows of a half-chunk;
    // a row is 66 columns w0-1 .. w0+64: lanes 0..63 take the first 64, the last two of all the wave's rows share one load
    constexpr int RPS = 12;                                  // rows per segment and half-chunk
    int r_seg[ROWS_W], r_c[ROWS_W], r_kh[ROWS_W];
#pragma unroll
    for (int j = 0; j < ROWS_W; ++j) {
        const int r = wave * ROWS_W + j;
        r_seg[j] = r / RPS; r_c[j] = (r % RPS) / 3; r_kh[j] = r % 3;
    }
    // (with ROWS_W = 6 or 12 a wave's rows belong to ONE segment when NSEG*3 divides 12, i.e. always here)
    const int st_seg = (wave * ROWS_W) / RPS;
    const int* sgs = segw + st_seg * 8;
    const int s_n = sgs[0], s_h = sgs[1], s_w0 = sgs[2], s_ok = sgs[3];
    const float* p_base = in + (long)s_n * Cin * HW;
    const int mcol = s_w0 - 1 + lane;                                          // main lanes: column of lane
    const float m_ok = (s_ok && mcol >= 0 && mcol < W) ? 1.f : 0.f;
    const int m_off = min(max(mcol, 0), W - 1);
    const int m_lds = (lane & 1) * POFF + (lane >> 1);                         // even columns -> E, odd -> O
    // halo items: lane -> (row j = lane >> 1, column 64 + (lane & 1))
    const int hj = min(lane >> 1, ROWS_W - 1), hcol = s_w0 + 63 + (lane & 1);
    const bool h_lane = lane < 2 * ROWS_W;
    const float h_okc = (s_ok && h_lane && hcol < W) ? 1.f : 0.f;
    const int h_off = min(hcol, W - 1);
    const int h_c = ((wave * ROWS_W + hj) % RPS) / 3, h_kh = (wave * ROWS_W + hj) % 3;
    const int h_lds = h_lane ? (st_seg * PSEG + h_c * 3 * PRW + h_kh * PRW + (lane & 1) * POFF + 32) : -1;
    float rp[ROWS_W + 1];
    auto load_patch = [&](int ci0) {
#pragma unroll
        for (int j = 0; j < ROWS_W; ++j) {
            const int hh = min(max(s_h + r_kh[j] - 1, 0), H - 1);
            const float* cb = p_base + (long)min(ci0 + r_c[j], Cin - 1) * HW + (long)hh * W;      // wave-uniform
            rp[j] = cb[m_off];
        }
        {
            const int hh = min(max(s_h + h_kh - 1, 0), H - 1);
            rp[ROWS_W] = p_base[(long)min(ci0 + h_c, Cin - 1) * HW + (long)hh * W + h_off];
        }
    };
    auto store_patch = [&](int ci0, int buf) {
#pragma unroll
        for (int j = 0; j < ROWS_W; ++j) {
            const int hh = s_h + r_kh[j] - 1;
            const float rm = ((ci0 + r_c[j]) < Cin && hh >= 0 && hh < H) ? 1.f : 0.f;               // wave-uniform
            P[buf * PBUF + r_seg[j] * PSEG + r_c[j] * 3 * PRW + r_kh[j] * PRW + m_lds] = rp[j] * (m_ok * rm);
        }
        {
            const int hh = s_h + h_kh - 1;
            const float rm = ((ci0 + h_c) < Cin && hh >= 0 && hh < H) ? 1.f : 0.f;
            P[h_lane ? buf * PBUF + h_lds : DUMMY + lane] = rp[ROWS_W] * (h_okc * rm);
        }
    };
    // ---- weight DMA: instruction q of a half-chunk moves rows [q*RPI, (q+1)*RPI) x CO_T floats = 1 KiB
    const int Ktot = Cin * 12;
    const int drow = lane / LPR, dcol = (lane % LPR) * 4;
    const bool dcol_ok = co0 + dcol < Cout;
    auto dma_weights = [&](int ci0, int buf) {
#pragma unroll
        for (int d = 0; d < DPW; ++d) {
            const int q = wave + 4 * d;                                               // wave-uniform
            const int gk = ci0 * 12 + q * RPI + drow;
            const float* src = (gk < Ktot && dcol_ok) ? wpack + (long)gk * Cout + co0 + dcol : zero_page;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(Wt + buf * WBUF + q * 256), 16, 0, 0);
        }
    };
    // ---- K loop of one half-chunk: 6 steps (channel pair cp: lanes 0-31 take channel cp, lanes 32-63 channel cp + 2; row kh),
    // each 4 transform points x TM row blocks = 8 MFMAs
    auto kloop = [&](int buf) {
        const float* wa = Wt + buf * WBUF + wco + li + lk * (2 * 12) * CO_T;
        const float* pb = P + buf * PBUF + wsg * PSEG + li + lk * 2 * 3 * PRW;
        float a[4][TM], e0, e1, o0, o1;
        auto reads = [&](int s, float (&aa)[4][TM], float& E0, float& E1, float& O0, float& O1) {
            const int cp = s / 3, kh = s % 3;                                            // compile-time after unrolling
            const float* pr = pb + (cp * 3 + kh) * PRW;
            E0 = pr[0]; E1 = pr[1]; O0 = pr[POFF]; O1 = pr[POFF + 1];
#pragma unroll
            for (int x = 0; x < 4; ++x)
#pragma unroll
                for (int i = 0; i < TM; ++i) aa[x][i] = wa[((cp * 3 + kh) * 4 + x) * CO_T + 32 * i];
        };
        reads(0, a, e0, e1, o0, o1);
#pragma unroll
        for (int s = 0; s < 6; ++s) {
            float an[4][TM], ne0 = 0.f, ne1 = 0.f, no0 = 0.f, no1 = 0.f;
#pragma unroll
            for (int x = 0; x < 4; ++x)
#pragma unroll
                for (int i = 0; i < TM; ++i) an[x][i] = 0.f;
            if (s + 1 < 6) reads(s + 1, an, ne0, ne1, no0, no1);
            const float vv[4] = {e0 - e1, o0 + e1, e1 - o0, o0 - o1};
            __builtin_amdgcn_sched_barrier(0);      // next step's ds_reads are issued BEFORE this step's MFMAs
#pragma unroll
            for (int x = 0; x < 4; ++x)
#pragma unroll
                for (int i = 0; i < TM; ++i) acc[i][x] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[x][i], vv[x], acc[i][x], 0, 0, 0);
#pragma unroll
            for (int x = 0; x < 4; ++x)
#pragma unroll
                for (int i = 0; i < TM; ++i) a[x][i] = an[x][i];
            e0 = ne0; e1 = ne1; o0 = no0; o1 = no1;
        }
    };

    const int nh = ((Cin + 2 * CI_H - 1) / (2 * CI_H)) * 2;          // half-chunks, padded to an even count (zero weights)
    dma_weights(0, 0);
    load_patch(0);
    store_patch(0, 0);
    for (int h = 0; h < nh; h += 2) {
        __syncthreads();                            // buffer 0 complete (DMA drained: vmcnt(0)), buffer 1 free
        dma_weights((h + 1) * CI_H, 1);
        load_patch((h + 1) * CI_H);
        kloop(0);
        store_patch((h + 1) * CI_H, 1);
        __syncthreads();                            // buffer 1 complete, buffer 0 free
        dma_weights((h + 2) * CI_H, 0);             // past the last channel: zero page / masked rows, never used
        load_patch((h + 2) * CI_H);
        kloop(1);
        store_patch((h + 2) * CI_H, 0);
    }

    // ---- output transform and stores: lane li = column pair, y(2t) = m0 + m1 + m2, y(2t+1) = m1 - m2 - m3
    const int* sg = segw + wsg * 8;
    if (!sg[3]) return;
    const int px = sg[2] + 2 * li;
    if (px >= W) return;
    const bool two = px + 1 < W;
    float* obase = out + (long)sg[0] * Cout * HW + (long)sg[1] * W + px;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = co0 + wco + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
            if (co < Cout) {
                const float b = bias ? bias[co] : 0.f;
                const float m0 = acc[i][0][r], m1 = acc[i][1][r], m2 = acc[i][2][r], m3 = acc[i][3][r];
                float* o = obase + (long)co * HW;
                o[0] = ((m0 + m1) + m2) + b;
                if (two) o[1] = ((m1 - m2) - m3) + b;
            }
        }
}

const float* wino_zero_page_ptr() {
    static const float* zp[64] = {nullptr};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    if (!zp[dev]) {
        void* p = nullptr;
        if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_wino_zero_page)) != hipSuccess) return nullptr;
        zp[dev] = (const float*)p;
    }
    return zp[dev];
}

}  // namespace

extern "C" int vocr_conv3x3_wino_supported(int cin, int cout) { return cin >= 4 && cout % 4 == 0 ? 1 : 0; }

// a pack = 12 transformed rows per contraction channel, followed by the direct pack's 9 rows per channel (for the tail pieces)
extern "C" size_t vocr_conv3x3_wino_pack_floats(int cout, int cin) { return (size_t)cout * cin * 21; }

extern "C" int vocr_conv3x3_wino_pack_weights(const float* w, float* wpack_fwd, float* wpack_dgrad, int cout, int cin, void* stream) {
    VOCR_CHECK_ARG(w && (wpack_fwd || wpack_dgrad), "vocr_conv3x3_wino_pack_weights: null pointer");
    VOCR_CHECK_ARG(cout > 0 && cin > 0, "vocr_conv3x3_wino_pack_weights: bad shape");
    const int total = cout * cin * 3;
    wino_pack_kernel<<<vocr_cdiv(total, 256), 256, 0, (hipStream_t)stream>>>(w, wpack_fwd, wpack_dgrad, cout, cin);
    VOCR_CHECK_LAUNCH("vocr_conv3x3_wino_pack_weights");
    // the direct pack behind the transformed rows
    return vocr_conv3x3_pack_weights(w, wpack_fwd ? wpack_fwd + (size_t)cout * cin * 12 : nullptr,
                                     wpack_dgrad ? wpack_dgrad + (size_t)cout * cin * 12 : nullptr, cout, cin, stream);
}

extern "C" int vocr_conv3x3_wino_fwd(const float* x, const float* wpack, const float* bias, float* y, int n, int cin, int h,
                                     int w, int cout, void* stream) {
    VOCR_CHECK_ARG(x && wpack && y, "vocr_conv3x3_wino_fwd: null pointer");
    VOCR_CHECK_ARG(n > 0 && cin > 0 && h > 0 && w > 0 && cout > 0, "vocr_conv3x3_wino_fwd: bad shape");
    VOCR_CHECK_ARG(cout % 4 == 0 && ((((uintptr_t)wpack) & 15) == 0), "vocr_conv3x3_wino_fwd: needs Cout %% 4 == 0 and a 16-byte aligned pack");
    VOCR_CHECK_ARG((long)n * (cin > cout ? cin : cout) * h * w < (1l << 31), "vocr_conv3x3_wino_fwd: tensor exceeds 2^31 elements");
    WGeom geo;
    geo.nsr = vocr_cdiv(vocr_cdiv(w, 2), TS);
    geo.per_img = h * geo.nsr;
    geo.nseg = n * geo.per_img;
    const float* zp = wino_zero_page_ptr();
    VOCR_CHECK_ARG(zp != nullptr, "vocr_conv3x3_wino_fwd: no device zero page");
    hipStream_t s = (hipStream_t)stream;
    const float* wdirect = wpack + (size_t)cin * 12 * cout;
    // VOCR_CONV_TAIL: 1 (default) the last partial round of tiles is cut into direct-form pieces that lead the launch, 0 whole tiles only
    static const int tail_mode = getenv("VOCR_CONV_TAIL") ? atoi(getenv("VOCR_CONV_TAIL")) : 1;
    int ncu = 256;
    {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ncu = prop.multiProcessorCount;
    }
#define VOCR_WINO_LAUNCH(CO_T, NSEG, CO_TILES)                                                                              \
    do {                                                                                                                    \
        const int tiles = vocr_cdiv(geo.nseg, NSEG) * (CO_TILES), rem = tiles % ncu;                                        \
        /* a piece is 1/16 of a tile in the direct form (1.5x the multiplications) and latency-bound when K is short:   */  \
        /* measured worth it up to a quarter round of tiles, up to half a round from 128 input channels on               */  \
        const bool cut = tail_mode == 1 && tiles > ncu && rem > 0 && (rem <= ncu / 4 || (rem <= ncu / 2 && cin >= 128));   \
        const int n_main = cut ? tiles - rem : tiles, n_tail = cut ? rem * (CO_T / 32) * 2 * NSEG : 0;                      \
        conv3x3_wino_kernel<CO_T><<<dim3(n_tail + n_main), 256, 0, s>>>(x, wpack, bias, y, zp, n, cin, h, w, cout, geo, (CO_TILES), wdirect, n_tail, n_main); \
    } while (0)
    if (cout > 64) VOCR_WINO_LAUNCH(128, 2, vocr_cdiv(cout, 128));
    else VOCR_WINO_LAUNCH(64, 4, 1);
#undef VOCR_WINO_LAUNCH
    VOCR_CHECK_LAUNCH("vocr_conv3x3_wino_fwd");
    return VOCR_OK;
}
