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
// x4 != 0: the four transform points of a (channel, filter row) are CONTIGUOUS per output channel - pf[(ci*3 + kh)][co][x] - so that a
// lane's four A operands of a k-step are one 16-byte LDS read (conv3x3_wino8_kernel<.., true>)
__global__ void wino_pack_kernel(const float* __restrict__ w, float* __restrict__ pf, float* __restrict__ pd, int cout, int cin, int x4) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int total = cout * cin * 3;
    if (i >= total) return;
    const int kh = i % 3, ci = (i / 3) % cin, co = i / (3 * cin);
    const float* g = w + ((long)(co * cin + ci) * 3 + kh) * 3;
    const float g0 = g[0], g1 = g[1], g2 = g[2];
    if (pf) {
        float* o = x4 ? pf + ((long)(ci * 3 + kh) * cout + co) * 4 : pf + (long)(ci * 12 + kh * 4) * cout + co;
        const long st = x4 ? 1 : cout;
        o[0] = g0;
        o[st] = 0.5f * ((g0 + g2) + g1);
        o[2 * st] = 0.5f * ((g0 + g2) - g1);
        o[3 * st] = g2;
    }
    if (pd) {       // flipped filter: row 2-kh, taps (g2, g1, g0)
        float* o = x4 ? pd + ((long)(co * 3 + (2 - kh)) * cin + ci) * 4 : pd + (long)(co * 12 + (2 - kh) * 4) * cin + ci;
        const long st = x4 ? 1 : cin;
        o[0] = g2;
        o[st] = 0.5f * ((g2 + g0) + g1);
        o[2 * st] = 0.5f * ((g2 + g0) - g1);
        o[3 * st] = g0;
    }
}

// VOCR_CONV_WINO2 (default 1): conv3x3_wino2_body, which reads the x-fastest pack; 0: round 3's kernels (VOCR_CONV_PACK4 then picks the
// pack for the VOCR_CONV_WINO8=1 experiments)
static int wino2_mode() {
    static const int v = VOCR_EXPERIMENT_INT("VOCR_CONV_WINO2", 1);
    return v;
}
// VOCR_CONV_WINO4 (default 1): F(4,3) along the row (conv3x3_wino4_body) for the forward / data-gradient launches with >= 64 output
// channels: own pack (18 transformed + 9 direct rows per channel); 0: F(2,3) everywhere (round 3 / early round 4)
static int wino4_mode() {
    static const int v = VOCR_EXPERIMENT_INT("VOCR_CONV_WINO4", 1);
    return v;
}
// which packs / launches take the F(4,3) kernel: convolutions with at least 64 OUTPUT channels (the forward of a layer with Cout >= 64, the
// data gradient of a layer with Cin >= 64).  Round 4 measured the 64-channel launches slower than F(2,3) (one 8-wave workgroup of eight
// segments) and kept them on F(2,3); with two 4-wave workgroups per CU (conv3x3_wino4x2_kernel_64) they are faster: 64 -> 64 at 30x600
// 255 -> 247 us forward, 249 -> 240 data gradient, the 64-channel data gradient of 64 -> 128 at 15x420 179 -> 162 (same-box step
// 15.27 / 15.28 / 15.29 -> 15.24 / 15.24 / 15.26 ms)
static int wino4_min_cout() {
    static const int v = VOCR_EXPERIMENT_INT("VOCR_CONV_WINO4_MINCO", 64);       // experiments: 128 = rounds 4 - 5's rule
    return v;
}
static bool wino4_for(int cout) { return wino4_mode() != 0 && cout >= wino4_min_cout() && cout % 4 == 0; }
static int wino_pack_x4() {
    static const int v = wino2_mode() ? 1 : (VOCR_EXPERIMENT_INT("VOCR_CONV_PACK4", 0));
    return v;
}

typedef unsigned int u32x4w __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
// Forward / data-gradient geometry: the column pairs of all image rows form ONE stream - a row contributes its T = ceil(W/2) pairs and
// one gap slot (the pair after the last one of a row must not see the next row's first columns) - and a segment is any 32 consecutive
// slots, so only 1 of T+1 lane positions is idle (whole 32-pair segments per row idled 8 % of them at W = 294, 6 % at 420 / 600).
// Slot q -> image n, row h, pair k (k == T: the gap).
struct WGeom { int nseg; int T, S, slots_img, nslot; };
struct WSlot { int n, h, k, valid; };
__device__ __forceinline__ WSlot wslot(int q, const WGeom& g) {
    WSlot s;
    s.valid = q < g.nslot;
    const int qq = s.valid ? q : 0;
    s.n = qq / g.slots_img;
    const int r = qq - s.n * g.slots_img;
    s.h = r / g.S;
    s.k = r - s.h * g.S;
    return s;
}

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
        // piece pixel block pb: 16 slots = 32 lane positions (lane position li = slot li >> 1, pixel parity li & 1)
        const int pb = sub / COSUB, q0 = ((vt / co_tiles) * NSEG + pb / 2) * TS + (pb & 1) * 16;
        const int pli = threadIdx.x & 31;
        const WSlot ps = wslot(q0 + (pli >> 1), geo);
        const int pcol = 2 * ps.k + (pli & 1);
        conv3x3_tail_piece_px<4>(lds, ps.n, ps.h, pcol, ps.valid && ps.k < geo.T && pcol < W, q0 < geo.nslot,
                                 (vt % co_tiles) * CO_T + (sub % COSUB) * 32, in, wdirect, bias, out, zero_page, Cin, H, W, Cout);
        return;
    }
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lk = lane >> 5;
    const int v = xcd_slice_order(blockIdx.x - n_tail, gridDim.x - n_tail);
    const int co0 = (v % co_tiles) * CO_T;
    const int seg0 = (v / co_tiles) * NSEG;
    const long HW = (long)H * W;

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
    // main lanes: entry e = lane >> 1 of the segment's 33 (entry e serves pair e as d0/d1 and pair e-1 as d2/d3), E or O by lane & 1
    const WSlot ms = wslot((seg0 + st_seg) * TS + (lane >> 1), geo);
    const int mcol = 2 * ms.k - 1 + (lane & 1);
    const float m_ok = (ms.valid && mcol >= 0 && mcol < W) ? 1.f : 0.f;
    const int m_base = ms.n * Cin * (int)HW + min(max(mcol, 0), W - 1);
    int m_row[3];
    float m_rok[3];
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
        const int hh = ms.h + kh - 1;
        m_row[kh] = min(max(hh, 0), H - 1) * W;
        m_rok[kh] = (hh >= 0 && hh < H) ? m_ok : 0.f;
    }
    const int m_lds = (lane & 1) * POFF + (lane >> 1);                         // even raw index -> E, odd -> O
    // halo items: lane -> (row j = lane >> 1 of this wave's rows, E or O of entry 32 = the first slot of the next segment)
    const int hj = min(lane >> 1, ROWS_W - 1);
    const bool h_lane = lane < 2 * ROWS_W;
    const WSlot hs = wslot((seg0 + st_seg) * TS + 32, geo);
    const int hcol = 2 * hs.k - 1 + (lane & 1);
    const int h_c = ((wave * ROWS_W + hj) % RPS) / 3, h_kh = (wave * ROWS_W + hj) % 3;
    const int h_hh = hs.h + h_kh - 1;
    const float h_okc = (hs.valid && h_lane && hcol >= 0 && hcol < W && h_hh >= 0 && h_hh < H) ? 1.f : 0.f;
    const int h_off = hs.n * Cin * (int)HW + min(max(h_hh, 0), H - 1) * W + min(max(hcol, 0), W - 1);
    const int h_lds = h_lane ? (st_seg * PSEG + h_c * 3 * PRW + h_kh * PRW + (lane & 1) * POFF + 32) : -1;
    float rp[ROWS_W + 1];
    auto load_patch = [&](int ci0) {
#pragma unroll
        for (int j = 0; j < ROWS_W; ++j) {
            const float* cb = in + (long)min(ci0 + r_c[j], Cin - 1) * HW;                          // wave-uniform
            rp[j] = cb[m_base + m_row[r_kh[j]]];
        }
        rp[ROWS_W] = in[(long)min(ci0 + h_c, Cin - 1) * HW + h_off];
    };
    auto store_patch = [&](int ci0, int buf) {
#pragma unroll
        for (int j = 0; j < ROWS_W; ++j) {
            const float rm = (ci0 + r_c[j]) < Cin ? 1.f : 0.f;                                       // wave-uniform
            P[buf * PBUF + r_seg[j] * PSEG + r_c[j] * 3 * PRW + r_kh[j] * PRW + m_lds] = rp[j] * (m_rok[r_kh[j]] * rm);
        }
        P[h_lane ? buf * PBUF + h_lds : DUMMY + lane] = rp[ROWS_W] * ((ci0 + h_c) < Cin ? h_okc : 0.f);
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
    const WSlot os = wslot((seg0 + wsg) * TS + li, geo);
    const int px = 2 * os.k;
    if (!os.valid || os.k >= geo.T || px >= W) return;
    const bool two = px + 1 < W;
    float* obase = out + (long)os.n * Cout * HW + (long)os.h * W + px;
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

// conv3x3_wino_kernel with buffer-addressed loaders (see the comment at its loaders); everything else is the same.
template <int CO_T>
__device__ __forceinline__ void conv3x3_wino2_body(const float* __restrict__ in, const float* __restrict__ wpack,
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
        // piece pixel block pb: 16 slots = 32 lane positions (lane position li = slot li >> 1, pixel parity li & 1)
        const int pb = sub / COSUB, q0 = ((vt / co_tiles) * NSEG + pb / 2) * TS + (pb & 1) * 16;
        const int pli = threadIdx.x & 31;
        const WSlot ps = wslot(q0 + (pli >> 1), geo);
        const int pcol = 2 * ps.k + (pli & 1);
        conv3x3_tail_piece_px<4>(lds, ps.n, ps.h, pcol, ps.valid && ps.k < geo.T && pcol < W, q0 < geo.nslot,
                                 (vt % co_tiles) * CO_T + (sub % COSUB) * 32, in, wdirect, bias, out, zero_page, Cin, H, W, Cout);
        return;
    }
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lk = lane >> 5;
    const int v = xcd_slice_order(blockIdx.x - n_tail, gridDim.x - n_tail);
    const int co0 = (v % co_tiles) * CO_T;
    const int seg0 = (v / co_tiles) * NSEG;
    const long HW = (long)H * W;

    const int wco = (wave / NSEG) * 64;
    const int wsg = wave % NSEG;
    f32x16 acc[TM][4];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int x = 0; x < 4; ++x)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][x][r] = 0.f;

    // ---- loaders (round 4).  Nothing vector hides behind an f32 MFMA on this chip (scripts/mfma_valu_probe.hip: every VALU instruction
    // costs 4 - 5 cycles of MFMA time), and conv3x3_wino_kernel spent ~90 VALU instructions per half-chunk of 48 MFMAs on 64-bit pointer
    // arithmetic, clamps and 0/1 masks.  Here every load goes through a buffer resource: the per-lane part of an address is a kernel
    // constant in one register (out-of-image positions: an offset beyond the buffer's range, which reads 0.0), the per-half-chunk part
    // (channel) is the instruction's SCALAR offset, and a channel past the end selects a resource of length 0 - no vector arithmetic
    // and no masks between two half-chunks.  (Tensors < 2^29 elements: byte offsets stay below 2^31, FAR + offset never wraps.)
    constexpr unsigned FAR = 0x80000000u;
    const __amdgpu_buffer_rsrc_t in_rs = __builtin_amdgcn_make_buffer_rsrc((void*)in, 0, (int)((long)N * Cin * HW * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t null_rs = __builtin_amdgcn_make_buffer_rsrc((void*)in, 0, 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t w_rs = __builtin_amdgcn_make_buffer_rsrc((void*)wpack, 0, Cin * 12 * Cout * 4, 0x00020000);
    // halo: this wave stages rows [wave*ROWS_W, (wave+1)*ROWS_W) of the (segment, c, kh) rows of a half-chunk (row r lives at P + r * PRW);
    // a row is 66 columns w0-1 .. w0+64: lanes 0..63 take the first 64, the last two of all the wave's rows share one load.
    // ROWS_W is a multiple of 3, so row j of a wave has filter row j % 3 (compile time) and channel (wave*ROWS_W % 12 + j) / 3.
    constexpr int RPS = 12;                                  // rows per segment and half-chunk
    const int st_seg = (wave * ROWS_W) / RPS;
    const int c_first = (wave * ROWS_W) % RPS;               // wave-uniform: row j -> channel (c_first + j) / 3
    // main lanes: entry e = lane >> 1 of the segment's 33 (entry e serves pair e as d0/d1 and pair e-1 as d2/d3), E or O by lane & 1
    const WSlot ms = wslot((seg0 + st_seg) * TS + (lane >> 1), geo);
    const int mcol = 2 * ms.k - 1 + (lane & 1);
    const bool m_ok = ms.valid && mcol >= 0 && mcol < W;
    unsigned m_vo[3];
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
        const int hh = ms.h + kh - 1;
        m_vo[kh] = (m_ok && hh >= 0 && hh < H) ? (unsigned)(ms.n * Cin * (int)HW + hh * W + mcol) * 4u : FAR;
    }
    const int m_lds = wave * ROWS_W * PRW + (lane & 1) * POFF + (lane >> 1);   // even raw index -> E, odd -> O
    // halo items: lane -> (row j = lane >> 1 of this wave's rows, E or O of entry 32 = the first slot of the next segment)
    const int hj = min(lane >> 1, ROWS_W - 1);
    const bool h_lane = lane < 2 * ROWS_W;
    const WSlot hs = wslot((seg0 + st_seg) * TS + 32, geo);
    const int hcol = 2 * hs.k - 1 + (lane & 1);
    const int h_c = ((wave * ROWS_W + hj) % RPS) / 3, h_kh = (wave * ROWS_W + hj) % 3;
    const int h_hh = hs.h + h_kh - 1;
    const unsigned h_vo = (hs.valid && h_lane && hcol >= 0 && hcol < W && h_hh >= 0 && h_hh < H)
                              ? (unsigned)(hs.n * Cin * (int)HW + h_c * (int)HW + h_hh * W + hcol) * 4u : FAR;
    const int h_lds = h_lane ? ((wave * ROWS_W + hj) * PRW + (lane & 1) * POFF + 32) : -1;
    float rp[ROWS_W + 1];
    auto load_patch = [&](int ci0) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < ROWS_W; ++j) {
            const int ch = ci0 + (c_first + j) / 3;                                                  // wave-uniform
            rp[j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(ch < Cin ? in_rs : null_rs, m_vo[j % 3], ch * (int)HW * 4, 0));
        }
        // the halo item's channel is per lane: only the last half-chunk of a channel count that is not a multiple of 4 has to look
        const unsigned hv = (ci0 + CI_H <= Cin || ci0 + h_c < Cin) ? h_vo : FAR;
        rp[ROWS_W] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(ci0 < Cin ? in_rs : null_rs, hv, ci0 * (int)HW * 4, 0));
    };
    auto store_patch = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < ROWS_W; ++j) P[buf * PBUF + j * PRW + m_lds] = rp[j];
        P[h_lane ? buf * PBUF + h_lds : DUMMY + lane] = rp[ROWS_W];
    };
    // ---- weight DMA.  The pack is x-fastest, pf[(ci*3 + kh)][co][4 transform points]: a half-chunk is 12 rows of CO_T x 16 bytes, an
    // instruction moves 64 output channels of one row (lane = channel); rows past the pack's end are out of the resource's range
    constexpr int IPR = CO_T / 64;                           // instructions per row
    unsigned w_vo[DPW];
#pragma unroll
    for (int d = 0; d < DPW; ++d) {
        const int q = wave + 4 * d, row = q / IPR, co = co0 + (q % IPR) * 64 + lane;
        w_vo[d] = co < Cout ? (unsigned)(row * Cout + co) * 16u : FAR;
    }
    auto dma_weights = [&](int ci0, int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int d = 0; d < DPW; ++d)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rs, (__attribute__((address_space(3))) void*)(Wt + buf * WBUF + (wave + 4 * d) * 256), 16, w_vo[d],
                                                     ci0 * 3 * Cout * 16, 0, 0);
    };
    // ---- K loop of one half-chunk: 6 steps (channel pair cp: lanes 0-31 take channel cp, lanes 32-63 channel cp + 2; row kh),
    // each 4 transform points x TM row blocks = 8 MFMAs.  Per step: two 16-byte reads of the weights (the four transform points of a
    // channel block), two 8-byte-pair reads of the raw row, three VALU instructions for the input transform (one of them packed).
    auto kloop = [&](int buf) __attribute__((always_inline)) {
        const float* wa = Wt + buf * WBUF + ((lk * 2 * 3) * CO_T + wco + li) * 4;
        const float* pb = P + buf * PBUF + wsg * PSEG + li + lk * 2 * 3 * PRW;
        f32x4 a[TM];
        f32x2 ev, ov;
        auto reads = [&](int s, f32x4 (&aa)[TM], f32x2& E, f32x2& O) __attribute__((always_inline)) {
            const int cp = s / 3, kh = s % 3;                                            // compile-time after unrolling
            const float* pr = pb + (cp * 3 + kh) * PRW;
            E = f32x2{pr[0], pr[1]};
            O = f32x2{pr[POFF], pr[POFF + 1]};
#pragma unroll
            for (int i = 0; i < TM; ++i) aa[i] = *(const f32x4*)(wa + ((cp * 3 + kh) * CO_T + 32 * i) * 4);
        };
        reads(0, a, ev, ov);
#pragma unroll
        for (int s = 0; s < 6; ++s) {
            f32x4 an[TM];
            f32x2 ne = ev, no = ov;
#pragma unroll
            for (int i = 0; i < TM; ++i) an[i] = a[i];
            if (s + 1 < 6) reads(s + 1, an, ne, no);
            // v0 = e0 - e1   v1 = o0 + e1   v2 = e1 - o0   v3 = o0 - o1
            f32x2 v12;
            asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[0,1] neg_hi:[1,0]\n\ts_nop 1" : "=v"(v12) : "v"(ov), "v"(ev));
            const float vv[4] = {ev[0] - ev[1], v12[0], v12[1], ov[0] - ov[1]};
            __builtin_amdgcn_sched_barrier(0);      // next step's ds_reads are issued BEFORE this step's MFMAs
#pragma unroll
            for (int x = 0; x < 4; ++x)
#pragma unroll
                for (int i = 0; i < TM; ++i) acc[i][x] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][x], vv[x], acc[i][x], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = an[i];
            ev = ne; ov = no;
        }
    };

    const int nh = ((Cin + 2 * CI_H - 1) / (2 * CI_H)) * 2;          // half-chunks, padded to an even count (zero weights)
    dma_weights(0, 0);
    load_patch(0);
    store_patch(0);
    for (int h = 0; h < nh; h += 2) {
        __syncthreads();                            // buffer 0 complete (DMA drained: vmcnt(0)), buffer 1 free
        dma_weights((h + 1) * CI_H, 1);
        load_patch((h + 1) * CI_H);
        kloop(0);
        store_patch(1);
        __syncthreads();                            // buffer 1 complete, buffer 0 free
        dma_weights((h + 2) * CI_H, 0);             // past the last channel: zero page / masked rows, never used
        load_patch((h + 2) * CI_H);
        kloop(1);
        store_patch(0);
    }

    // ---- output transform and stores: lane li = column pair, y(2t) = m0 + m1 + m2, y(2t+1) = m1 - m2 - m3
    const WSlot os = wslot((seg0 + wsg) * TS + li, geo);
    const int px = 2 * os.k;
    if (!os.valid || os.k >= geo.T || px >= W) return;
    const bool two = px + 1 < W;
    float* obase = out + (long)os.n * Cout * HW + (long)os.h * W + px;
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


// (device-only builtins inside a TEMPLATE kernel make the host pass drop its launch stub, hence the body as a device function)
__global__ __launch_bounds__(256, 2) void conv3x3_wino2_kernel_128(const float* __restrict__ in, const float* __restrict__ wpack, const float* __restrict__ bias,
                                                                   float* __restrict__ out, const float* __restrict__ zero_page, int N, int Cin, int H, int W, int Cout,
                                                                   WGeom geo, int co_tiles, const float* __restrict__ wdirect, int n_tail, int first_tail_tile) {
    conv3x3_wino2_body<128>(in, wpack, bias, out, zero_page, N, Cin, H, W, Cout, geo, co_tiles, wdirect, n_tail, first_tail_tile);
}
__global__ __launch_bounds__(256, 2) void conv3x3_wino2_kernel_64(const float* __restrict__ in, const float* __restrict__ wpack, const float* __restrict__ bias,
                                                                  float* __restrict__ out, const float* __restrict__ zero_page, int N, int Cin, int H, int W, int Cout,
                                                                  WGeom geo, int co_tiles, const float* __restrict__ wdirect, int n_tail, int first_tail_tile) {
    conv3x3_wino2_body<64>(in, wpack, bias, out, zero_page, N, Cin, H, W, Cout, geo, co_tiles, wdirect, n_tail, first_tail_tile);
}

// ---------------------------------------------------------------- forward / data gradient with F(4,3) ALONG THE ROW (round 4, VOCR_CONV_WINO4=1)
// Four neighbouring outputs of a row from six multiplications per (ci, kh) instead of the eight of two F(2,3) pairs: the MFMA does 1/2
// of the direct kernel's work instead of 2/3.
//   d0..d5 = input columns 4q-1 .. 4q+4
//   v0 = 4 d0 - 5 d2 + d4        v1 = (d4 - 4 d2) + (d3 - 4 d1)      v2 = (d4 - 4 d2) - (d3 - 4 d1)
//   v3 = (d4 - d2) + 2 (d3 - d1) v4 = (d4 - d2) - 2 (d3 - d1)        v5 = 4 d1 - 5 d3 + d5
//   g0..g2 = the taps of filter row kh:
//   u0 = g0/4   u1 = -(g0+g1+g2)/6   u2 = -(g0-g1+g2)/6   u3 = g0/24 + g1/12 + g2/6   u4 = g0/24 - g1/12 + g2/6   u5 = g2
//   m_x = sum over (ci, kh) of u_x v_x
//   y(4q) = m0+m1+m2+m3+m4   y(4q+1) = (m1-m2) + 2 (m3-m4)   y(4q+2) = (m1+m2) + 4 (m3+m4)   y(4q+3) = (m1-m2) + 8 (m3-m4) + m5
// Rounding: measured 2.7 - 3x the F(2,3) error on 64 .. 256-channel sums of non-negative activations (1.4e-6 of the result's scale
// against 5e-7, direct fp32 7e-7: scripts/README.md); the full-size oracle test decides whether it may be the default.
// Six accumulator tiles per 32 output channels and 32 quads: 96 registers - one wave per SIMD with BIG register tiles (the budget of
// §5: the input transform's 12 VALU instructions are paid once per quad row and serve TM x 6 MFMAs).  Workgroup = 4 waves = one
// per SIMD, one workgroup per CU; a wave owns ALL CO_T output channels (TM = CO_T / 32 row blocks) of TN segments of 32 quads =
// 128 pixels: CO_T = 128: TM 4, TN 1 (4 segments per workgroup), CO_T = 64: TM 2, TN 2 (8 segments).  The raw halo rows of a wave's
// segments are staged by that wave alone (no other wave reads them); the weights of a stage (4 input channels: 12 filter rows of
// CO_T x 6 transformed values, as [row][co][4] + [row][co][2] so that a fragment is one 16-byte and one 8-byte read) arrive by
// LDS-DMA for all four waves, two stages.
#ifndef W43_CUT            // diagnostic builds: 1 no DMA / halo loads in the loop (wrong results)
#define W43_CUT 0
#endif
#ifndef W43_PACKED         // 0: the scalar twelve-instruction input transform (A/B builds)
#define W43_PACKED 1
#endif
constexpr int Q4_TS = 32;                   // quads per segment
constexpr int Q4_PRW = 132;                 // floats per staged halo row: raw index r = 4 e + c <-> column 4 k_e - 1 + c, r < 130
__global__ void wino4_pack_kernel(const float* __restrict__ w, float* __restrict__ pf, float* __restrict__ pd, int cout, int cin) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int total = cout * cin * 3;
    if (i >= total) return;
    const int kh = i % 3, ci = (i / 3) % cin, co = i / (3 * cin);
    const float* g = w + ((long)(co * cin + ci) * 3 + kh) * 3;
    auto put = [](float* p4, float* p2, float g0, float g1, float g2) {
        p4[0] = 0.25f * g0;
        p4[1] = -((g0 + g2) + g1) * (1.f / 6.f);
        p4[2] = -((g0 + g2) - g1) * (1.f / 6.f);
        p4[3] = (g0 * (1.f / 24.f) + g2 * (1.f / 6.f)) + g1 * (1.f / 12.f);
        p2[0] = (g0 * (1.f / 24.f) + g2 * (1.f / 6.f)) - g1 * (1.f / 12.f);
        p2[1] = g2;
    };
    if (pf) {           // pf: [(ci*3 + kh)][co][4] | [(ci*3 + kh)][co][2]
        float* p4 = pf + ((long)(ci * 3 + kh) * cout + co) * 4;
        float* p2 = pf + (long)cin * 3 * cout * 4 + ((long)(ci * 3 + kh) * cout + co) * 2;
        put(p4, p2, g[0], g[1], g[2]);
    }
    if (pd) {           // data gradient: filter row 2-kh, taps (g2, g1, g0), channels transposed
        float* p4 = pd + ((long)(co * 3 + (2 - kh)) * cin + ci) * 4;
        float* p2 = pd + (long)cout * 3 * cin * 4 + ((long)(co * 3 + (2 - kh)) * cin + ci) * 2;
        put(p4, p2, g[2], g[1], g[0]);
    }
}

// F(4,3) input transform of one quad's six raw columns e0..e5 as SIX packed instructions (the scalar form is twelve):
//   (ta, tc) = e4 + (-4, -1) e2     (tb, td) = e3 + (-4, -1) e1     (v1, v2) = ta +- tb     (v3, v4) = tc +- 2 td
//   (v0, v5) = (4 e0 + e4, 4 e1 + e5) - 5 (e2, e3)
// op_sel / op_sel_hi pick the half of a register pair per result half, so no value is moved; ka = (-4, -1), kb = (2, -2), kc = (4, -5)
// live in registers.  The results are MFMA operands right away: the block ends in the two wait states (see w3_xform_first).
__device__ __forceinline__ void w43_input_xform(f32x2 e01, f32x2 e23, f32x2 e45, f32x2 ka, f32x2 kb, f32x2 kc, f32x2& v12, f32x2& v34,
                                                f32x2& v05) {
    f32x2 p2;                  // v05 first holds (4 e0 + e4, 4 e1 + e5), v34 first (ta, tc)
    asm("v_pk_fma_f32 %2, %4, %9, %6 op_sel:[0,0,0] op_sel_hi:[1,0,1]\n\t"
        "v_pk_fma_f32 %1, %5, %7, %6 op_sel:[0,0,0] op_sel_hi:[0,1,0]\n\t"
        "v_pk_fma_f32 %3, %4, %7, %5 op_sel:[1,0,1] op_sel_hi:[1,1,1]\n\t"
        "v_pk_fma_f32 %2, %5, %9, %2 op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t"
        "v_pk_add_f32 %0, %1, %3 op_sel:[0,0] op_sel_hi:[0,0] neg_hi:[0,1]\n\t"
        "v_pk_fma_f32 %1, %3, %8, %1 op_sel:[1,0,1] op_sel_hi:[1,1,1]\n\t"
        "s_nop 1"
        : "=&v"(v12), "=&v"(v34), "=&v"(v05), "=&v"(p2)
        : "v"(e01), "v"(e23), "v"(e45), "v"(ka), "v"(kb), "v"(kc));
}

template <int CO_T, int NW, int CS>
__device__ __forceinline__ void conv3x3_wino4_body(const float* __restrict__ in, const float* __restrict__ wpack, const float* __restrict__ bias,
                                                   float* __restrict__ out, const float* __restrict__ zero_page, int N, int Cin, int H, int W, int Cout,
                                                   WGeom geo, int co_tiles, const float* __restrict__ wdirect, int n_tail, int first_tail_tile) {
    constexpr int TM = 2, TN = 1;                            // a wave: 64 output channels x one segment (12 accumulator tiles = 192 registers,
                                                             // all of them AGPRs; 24 tiles made the compiler shuttle them through VGPRs)
    constexpr int WAVES_CO = CO_T / 64;
    constexpr int NSEG = NW / WAVES_CO;                      // segments per workgroup (NW waves: 4 = one per SIMD, 8 = two)
    constexpr int RS = 3 * CS;                               // (channel, filter row) rows of a stage: CS = 4 or 2 input channels
    constexpr int KS = RS / 2;                               // k-steps of a stage (lanes 0-31: the first CS/2 channels, lanes 32-63 the others)
    constexpr int PSEG = RS * Q4_PRW;                        // floats of halo per segment and stage
    constexpr int ROWS_W = NSEG * RS / NW;                   // halo rows a wave stages per stage (a multiple of 3)
    constexpr int W4 = RS * CO_T * 4, W2 = RS * CO_T * 2;    // floats of a weight stage
    constexpr int WBUF = W4 + W2;
    constexpr int PBUF = NSEG * PSEG;
    constexpr int NDMA = WBUF / 256;                         // 36 or 18 instructions of 1 KiB per stage
    constexpr int DPW = (NDMA + NW - 1) / NW;                // per wave (instructions past NDMA are dummies)
    __shared__ __attribute__((aligned(16))) float lds[2 * WBUF + 2 * PBUF + 256 + 64];
    float* const Wt = lds;
    float* const P = lds + 2 * WBUF;
    constexpr int DMA_DUMMY = 2 * WBUF + 2 * PBUF, ST_DUMMY = DMA_DUMMY + 256;

    if ((int)blockIdx.x < n_tail) {
        // the last partial round of tiles as direct-form pieces of 32 channels x 32 pixels (conv_tail.h): a tile is (CO_T/32) channel
        // blocks x 4*NSEG pixel blocks; lane position li of a piece = quad li >> 2 of its 8, pixel li & 3
        constexpr int COSUB = CO_T / 32, PPW = COSUB * 4 * NSEG;
        const int piece = blockIdx.x, vt = first_tail_tile + piece / PPW, sub = piece % PPW;
        const int pb = sub / COSUB, q0 = ((vt / co_tiles) * NSEG + pb / 4) * Q4_TS + (pb & 3) * 8;
        const int pli = threadIdx.x & 31;
        const WSlot ps = wslot(q0 + (pli >> 2), geo);
        const int pcol = 4 * ps.k + (pli & 3);
        conv3x3_tail_piece_px<NW>(lds, ps.n, ps.h, pcol, ps.valid && ps.k < geo.T && pcol < W, q0 < geo.nslot,
                                 (vt % co_tiles) * CO_T + (sub % COSUB) * 32, in, wdirect, bias, out, zero_page, Cin, H, W, Cout);
        return;
    }
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lk = lane >> 5;
    const int v = xcd_slice_order(blockIdx.x - n_tail, gridDim.x - n_tail);
    const int co0 = (v % co_tiles) * CO_T;
    const int wco = (wave / NSEG) * 64, wsg = wave % NSEG;    // this wave's channels and segment
    const int sega = (v / co_tiles) * NSEG;                  // the workgroup's first segment
    const int seg0 = sega + wsg;
    const int iHW = H * W;

    f32x16 acc[TM][TN][6];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int n = 0; n < TN; ++n)
#pragma unroll
            for (int x = 0; x < 6; ++x)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][n][x][r] = 0.f;

    constexpr unsigned FAR = 0x80000000u;
    const __amdgpu_buffer_rsrc_t in_rs = __builtin_amdgcn_make_buffer_rsrc((void*)in, 0, (int)((long)N * Cin * iHW * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t null_rs = __builtin_amdgcn_make_buffer_rsrc((void*)in, 0, 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t w_rs = __builtin_amdgcn_make_buffer_rsrc((void*)wpack, 0, Cin * 18 * Cout * 4, 0x00020000);
    // ---- halo: per segment 12 rows (c, kh) of 130 raw values, row r of the workgroup's NSEG * 12 at P + r * PRW.  This wave stages rows
    // [wave * ROWS_W, + ROWS_W) (one segment's, ROWS_W a multiple of 3: row j has filter row j % 3): the 128 columns of a row go
    // global -> LDS by two 4-byte LDS-DMAs (lane = raw index r, quad r >> 2, column 4 k - 1 + (r & 3); nothing passes through registers),
    // raw 128 / 129 (the first two columns of the NEXT segment's first quad) of the wave's rows by one load (lanes 0 .. 2 ROWS_W - 1)
    const int st_seg = (wave * ROWS_W) / RS, c_first = (wave * ROWS_W) % RS;
    unsigned m_vo[2][3];
#pragma unroll
    for (int hlf = 0; hlf < 2; ++hlf) {
        const int r = lane + 64 * hlf;
        const WSlot ms = wslot((sega + st_seg) * Q4_TS + (r >> 2), geo);
        const int col = 4 * ms.k - 1 + (r & 3);
        const bool ok = ms.valid && col >= 0 && col < W;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            const int hh = ms.h + kh - 1;
            m_vo[hlf][kh] = (ok && hh >= 0 && hh < H) ? (unsigned)(ms.n * Cin * iHW + hh * W + col) * 4u : FAR;
        }
    }
    const int hj = min(lane >> 1, ROWS_W - 1), h_c = (c_first + hj) / 3, h_kh = (c_first + hj) % 3;
    const bool h_lane = lane < 2 * ROWS_W;
    unsigned h_vo;
    {
        const WSlot hs = wslot((sega + st_seg) * Q4_TS + 32, geo);
        const int col = 4 * hs.k - 1 + (lane & 1), hh = hs.h + h_kh - 1;
        h_vo = (hs.valid && h_lane && col >= 0 && col < W && hh >= 0 && hh < H) ? (unsigned)(hs.n * Cin * iHW + h_c * iHW + hh * W + col) * 4u : FAR;
    }
    const int p_rows = wave * ROWS_W * Q4_PRW;               // this wave's rows inside a halo stage
    float rh;
    auto load_patch = [&](int ci0, int buf) __attribute__((always_inline)) {
        float* pp = P + buf * PBUF + p_rows;
#pragma unroll
        for (int j = 0; j < ROWS_W; ++j) {
            const int ch = ci0 + (c_first + j) / 3;                                              // wave-uniform
#pragma unroll
            for (int hlf = 0; hlf < 2; ++hlf)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(ch < Cin ? in_rs : null_rs, (__attribute__((address_space(3))) void*)(pp + j * Q4_PRW + 64 * hlf), 4,
                                                         m_vo[hlf][j % 3], ch * iHW * 4, 0, 0);
        }
        const unsigned hv = (ci0 + CS <= Cin || ci0 + h_c < Cin) ? h_vo : FAR;
        rh = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(ci0 < Cin ? in_rs : null_rs, hv, ci0 * iHW * 4, 0));
    };
    auto store_patch = [&](int buf) __attribute__((always_inline)) {
        float* hd = h_lane ? P + buf * PBUF + p_rows + hj * Q4_PRW + 128 + (lane & 1) : lds + ST_DUMMY + lane;
        *hd = rh;
    };
    // ---- weight DMA: instruction q < NDMA of a stage; q < W4/256: row q / (CO_T/64) of the [row][co][4] part, 64 channels; else the
    // [row][co][2] part: 128 channels (two per lane... one lane = 16 bytes = two channels) of row (q - W4/256) / (CO_T/128 or 1)
    unsigned w_vo[DPW];
#pragma unroll
    for (int d = 0; d < DPW; ++d) {
        const int q = wave + NW * d;
        if (q < W4 / 256) {
            const int row = q / (CO_T / 64), co = co0 + (q % (CO_T / 64)) * 64 + lane;
            w_vo[d] = co < Cout ? (unsigned)(row * Cout + co) * 16u : FAR;
        } else if (q < NDMA) {
            constexpr int IPR2 = CO_T / 128 > 0 ? CO_T / 128 : 1;      // instructions per row of the 2-value part (CO_T = 64: one covers TWO rows)
            const int q2 = q - W4 / 256;
            int row, co;
            if (CO_T >= 128) { row = q2 / IPR2; co = co0 + (q2 % IPR2) * 128 + 2 * lane; }
            else { row = 2 * q2 + (lane >> 5); co = co0 + 2 * (lane & 31); }
            w_vo[d] = co < Cout ? (unsigned)(Cin * 3 * Cout * 4 + (row * Cout + co) * 2) * 4u : FAR;
        } else {
            w_vo[d] = FAR;
        }
    }
    auto dma_weights = [&](int ci0, int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int d = 0; d < DPW; ++d) {
            const int q = wave + NW * d;                      // wave-uniform
            float* dst = q < NDMA ? Wt + buf * WBUF + q * 256 : lds + DMA_DUMMY;
            // the scalar offset advances by 3 rows per channel: 16 bytes per (row, channel) in the first part, 8 in the second
            const int so = q < W4 / 256 ? ci0 * 3 * Cout * 16 : ci0 * 3 * Cout * 8;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rs, (__attribute__((address_space(3))) void*)dst, 16, w_vo[d], so, 0, 0);
        }
    };
    // ---- K loop of one stage: 6 steps (channel pair cp: lanes 0-31 channel cp, lanes 32-63 channel cp + 2; filter row kh), each
    // 6 transform points x TM row blocks x TN segments MFMAs
    // the packed input transform's constants: pinned in registers (as literals the compiler would re-materialise them per use)
    f32x2 xka = {-4.f, -1.f}, xkb = {2.f, -2.f}, xkc = {4.f, -5.f};
    asm volatile("" : "+v"(xka), "+v"(xkb), "+v"(xkc));
    auto kloop = [&](int buf) __attribute__((always_inline)) {
        const float* wa4 = Wt + buf * WBUF + ((lk * KS) * CO_T + wco + li) * 4;
        const float* wa2 = Wt + buf * WBUF + W4 + ((lk * KS) * CO_T + wco + li) * 2;
        const float* pb = P + buf * PBUF + wsg * PSEG + lk * KS * Q4_PRW + 4 * li;
        f32x4 d4[TN], a4[TM];
        f32x2 d2[TN], a2[TM];
        auto reads = [&](int s, f32x4 (&A4)[TM], f32x2 (&A2)[TM], f32x4 (&D4)[TN], f32x2 (&D2)[TN]) __attribute__((always_inline)) {
            const int row = s;                                                           // (channel s / 3 [+ CS/2 for lanes 32-63], filter row s % 3)
#pragma unroll
            for (int n = 0; n < TN; ++n) {
                D4[n] = *(const f32x4*)(pb + n * PSEG + row * Q4_PRW);
                D2[n] = *(const f32x2*)(pb + n * PSEG + row * Q4_PRW + 4);
            }
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                A4[i] = *(const f32x4*)(wa4 + (row * CO_T + 32 * i) * 4);
                A2[i] = *(const f32x2*)(wa2 + (row * CO_T + 32 * i) * 2);
            }
        };
        // one wave per SIMD: nobody else covers an LDS latency, so the fragments of step s + 1 are in flight under the MFMAs of step s;
        // with two waves per SIMD (two workgroups per CU, or the 8-wave workgroup) the registers that costs are worth more than the prefetch
        constexpr bool PF = NW == 4 && CS != 2;        // the 8-wave workgroup also has two waves per SIMD
        reads(0, a4, a2, d4, d2);
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            f32x4 nd4[TN], na4[TM];
            f32x2 nd2[TN], na2[TM];
#pragma unroll
            for (int i = 0; i < TM; ++i) { na4[i] = a4[i]; na2[i] = a2[i]; }
#pragma unroll
            for (int n = 0; n < TN; ++n) { nd4[n] = d4[n]; nd2[n] = d2[n]; }
            if (PF && s + 1 < KS) reads(s + 1, na4, na2, nd4, nd2);
            float vv[TN][6];
#pragma unroll
            for (int n = 0; n < TN; ++n) {
                if (W43_PACKED) {
                    f32x2 v12, v34, v05;
                    w43_input_xform((f32x2){d4[n][0], d4[n][1]}, (f32x2){d4[n][2], d4[n][3]}, d2[n], xka, xkb, xkc, v12, v34, v05);
                    vv[n][0] = v05[0]; vv[n][1] = v12[0]; vv[n][2] = v12[1]; vv[n][3] = v34[0]; vv[n][4] = v34[1]; vv[n][5] = v05[1];
                } else {
                    const float e0 = d4[n][0], e1 = d4[n][1], e2 = d4[n][2], e3 = d4[n][3], e4 = d2[n][0], e5 = d2[n][1];
                    const float ta = __builtin_fmaf(-4.f, e2, e4), tb = __builtin_fmaf(-4.f, e1, e3);
                    const float tc = e4 - e2, td = e3 - e1;
                    vv[n][0] = __builtin_fmaf(4.f, e0, __builtin_fmaf(-5.f, e2, e4));
                    vv[n][1] = ta + tb;
                    vv[n][2] = ta - tb;
                    vv[n][3] = __builtin_fmaf(2.f, td, tc);
                    vv[n][4] = __builtin_fmaf(-2.f, td, tc);
                    vv[n][5] = __builtin_fmaf(4.f, e1, __builtin_fmaf(-5.f, e3, e5));
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int x = 0; x < 6; ++x)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int n = 0; n < TN; ++n)
                        acc[i][n][x] = __builtin_amdgcn_mfma_f32_32x32x2f32(x < 4 ? a4[i][x] : a2[i][x - 4], vv[n][x], acc[i][n][x], 0, 0, 0);
            if (PF) {
#pragma unroll
                for (int i = 0; i < TM; ++i) { a4[i] = na4[i]; a2[i] = na2[i]; }
#pragma unroll
                for (int n = 0; n < TN; ++n) { d4[n] = nd4[n]; d2[n] = nd2[n]; }
            } else if (s + 1 < KS) {
                reads(s + 1, a4, a2, d4, d2);       // issued behind this step's MFMAs: the SIMD's other wave covers their latency
            }
        }
    };

    const int nh = ((Cin + 2 * CS - 1) / (2 * CS)) * 2;     // stages of CS channels, padded to an even count (zero weights)
    dma_weights(0, 0);
    load_patch(0, 0);
    store_patch(0);
    for (int h = 0; h < nh; h += 2) {
        __syncthreads();                            // buffer 0 complete (DMA drained: vmcnt(0)), buffer 1 free
        if (!(W43_CUT & 1)) { dma_weights((h + 1) * CS, 1); load_patch((h + 1) * CS, 1); }
        kloop(0);
        if (!(W43_CUT & 1)) store_patch(1);
        __syncthreads();                            // buffer 1 complete, buffer 0 free
        if (!(W43_CUT & 1)) { dma_weights((h + 2) * CS, 0); load_patch((h + 2) * CS, 0); }      // past the last channel: out of range / null resource, never used
        kloop(1);
        if (!(W43_CUT & 1)) store_patch(0);
    }

    // ---- output transform and stores: lane li = quad, 4 pixels 4k .. 4k+3
#pragma unroll
    for (int n = 0; n < TN; ++n) {
        const WSlot os = wslot((seg0 + n) * Q4_TS + li, geo);
        const int px = 4 * os.k;
        if (!os.valid || os.k >= geo.T || px >= W) continue;
        const int nvalid = min(4, W - px);
        float* obase = out + (long)os.n * Cout * iHW + (long)os.h * W + px;
        const bool vec = nvalid == 4 && ((W & 3) == 0) && ((((uintptr_t)out) & 15) == 0);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co0 + wco + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
                if (co < Cout) {
                    const float b = bias ? bias[co] : 0.f;
                    const float m0 = acc[i][n][0][r], m1 = acc[i][n][1][r], m2 = acc[i][n][2][r], m3 = acc[i][n][3][r], m4 = acc[i][n][4][r], m5 = acc[i][n][5][r];
                    const float s12 = m1 + m2, d12 = m1 - m2, s34 = m3 + m4, d34 = m3 - m4;
                    f32x4 y;
                    y[0] = ((m0 + s12) + s34) + b;
                    y[1] = __builtin_fmaf(2.f, d34, d12) + b;
                    y[2] = __builtin_fmaf(4.f, s34, s12) + b;
                    y[3] = (__builtin_fmaf(8.f, d34, d12) + m5) + b;
                    float* o = obase + (long)co * iHW;
                    if (vec) *(f32x4*)o = y;
                    else {
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (e < nvalid) o[e] = y[e];
                    }
                }
            }
    }
}
#define VOCR_WINO4_KERNEL(NAME, CO_T, NW, CS, WGS)                                                                                                \
    __global__ __launch_bounds__(64 * NW, WGS) void NAME(const float* __restrict__ in, const float* __restrict__ wpack,                              \
                                                        const float* __restrict__ bias, float* __restrict__ out,                                    \
                                                        const float* __restrict__ zero_page, int N, int Cin, int H, int W, int Cout, WGeom geo,       \
                                                        int co_tiles, const float* __restrict__ wdirect, int n_tail, int first_tail_tile) {          \
        conv3x3_wino4_body<CO_T, NW, CS>(in, wpack, bias, out, zero_page, N, Cin, H, W, Cout, geo, co_tiles, wdirect, n_tail, first_tail_tile);     \
    }
VOCR_WINO4_KERNEL(conv3x3_wino4_kernel_128, 128, 4, 4, 1)
VOCR_WINO4_KERNEL(conv3x3_wino4_kernel_64, 64, 4, 4, 1)
VOCR_WINO4_KERNEL(conv3x3_wino4w8_kernel_128, 128, 8, 4, 1)
VOCR_WINO4_KERNEL(conv3x3_wino4w8_kernel_64, 64, 8, 4, 1)
// two 4-wave workgroups per CU (256 registers per lane, 61 KB of LDS each: stages of TWO channels), independent barriers
VOCR_WINO4_KERNEL(conv3x3_wino4x2_kernel_128, 128, 4, 2, 2)
VOCR_WINO4_KERNEL(conv3x3_wino4x2_kernel_64, 64, 4, 2, 2)
#undef VOCR_WINO4_KERNEL

// ---------------------------------------------------------------- weight gradient, F(3,2) along the row
// dw[co][ci][kh][kw] = sum over pixels of dy[co][h][w] * x[ci][h+kh-1][w+kw-1].  For a column PAIR (2p, 2p+1) the three taps kw
// are the outputs of a 2-tap filter g = (dy[2p], dy[2p+1]) over d0..d3 = x columns 2p-1 .. 2p+2 - the transposed form of the
// forward transform: 4 multiplications per pair, channel pair and kh instead of 6:
//   a0 = g0   a1 = (g0+g1)/2   a2 = (g0-g1)/2   a3 = g1          b0 = d0 - d2   b1 = d1 + d2   b2 = d2 - d1   b3 = d3 - d1
//   M_x += a_x * b_x  over all pairs                              dw(kw=0) = M0+M1+M2   dw(1) = M1-M2   dw(2) = M1+M2+M3
// The output transform is linear, so it is applied once to the accumulators at the end.
// (Round 3's two kernels of this contraction - 64-pixel row segments staged through registers, then by LDS-DMA - were the default
// until round 4 and are gone: HISTORY.md describes them; the piece-stream kernel below serves the channel counts the row-pair kernel
// does not take.)

// ---------------------------------------------------------------- weight gradient F(3,2), piece stream, twelve waves (round 4)
// What changed against round 3's segment kernels, and why:
//   * THE BUDGET.  On this chip the f32 MFMA peak IS the vector f32 peak, and nothing vector hides behind a v_mfma_f32_32x32x2_f32:
//     beside a stream of them (64 cycles each) every plain VALU instruction costs 4 - 5 cycles of MFMA time, a packed one
//     (v_pk_add_f32) the same 5, a ds_read 2, an LDS-DMA 7 - 14 (scripts/mfma_valu_probe.hip, three waves per SIMD).  The segment
//     kernel spent 40 VALU instructions per 12 MFMAs; this one 7 per 8 (+ 4 LDS reads):
//       - both transforms as PACKED adds on register pairs that the 16-byte fragment reads deliver aligned: (g0+g1, g0-g1) is one
//         v_pk_add_f32, (d1+d2, d2-d1) one, (c2-c0, c1-c3) of the two pairs of a piece one; the factor 1/2 of the filter-side
//         transform moved into the output transform (exact: a power of two);
//       - every fragment address is a per-lane kernel constant in a register (32 of them), the second LDS buffer is the
//         instruction's immediate offset (the loop is unrolled over the two buffers), so the k-loop has no address arithmetic;
//       - a DMA's source offset is one add-and-shift of a per-lane constant and the segment's slot offset plus one select.
//   * FRAGMENT READS.  There every fragment element was a ds_read_b32 whose 32 lanes (= channels) sit 16 bytes x pieces apart: the
//     channel XOR moves a lane by whole 16-byte pieces, so lanes c, c+8, c+16, c+24 always share a bank - a 4-way conflict on all
//     14 reads of a k-step (SQ_LDS_BANK_CONFLICT 6.7e7 per launch, 19x the forward kernel's).  Here a lane reads WHOLE 16-byte
//     pieces with ds_read_b128 from rows of exactly 16 pieces (256 bytes = one bank row) with the piece position XORed by
//     (channel & 15): the 16 lanes of every ds_read_b128 lane group then cover the 64 banks exactly once.  One dy piece = pixels
//     4k .. 4k+3 = two column pairs = two k-steps; lane half lk takes slot 2j + lk of the segment (the MFMA only needs A and B to
//     agree on k).  The x piece of the same slot holds columns 4k .. 4k+3; the columns 4k-1 and 4k+4 the two pairs also need are
//     the last / first element of the neighbouring slots' pieces (two ds_read_b32).
//   * PIECE STREAM.  A row contributes np = ceil(W/4) slots plus ONE all-zero slot (dy = 0, x = 0: it is the zero padding right of
//     the row's last column AND left of the next row's first one); a segment is any 16 consecutive slots of the stream of all rows
//     of all images.  Whole 64-pixel segments per row idled 6-8 % of the MFMA work at W = 600 / 420 / 294; this form < 1.5 %.
//   * TWELVE WAVES.  wave = (filter row kh, output-channel half, input-channel half): 4 accumulator tiles = 64 registers instead of
//     192, three waves per SIMD.
// LDS: dy[2][64 co][16 pos][4] | 2 x { x [3 kh][64 ci][16 pos][4] | extras [6][64 ci][4] } (extras: slot -1 and slot 16 of the
// segment for the three filter rows - the neighbours of its first and last slot); position of logical slot s of channel c =
// s ^ (c & 15).  A DMA instruction fills 1 KiB = 4 channels x 16 positions; instruction id -> wave so that id & 3 == wave & 3: a lane
// then only ever moves ONE logical slot of a segment, (lane & 15) ^ (lane >> 4) ^ 4 (wave & 3), and tracks its (image, row, piece)
// incrementally.
#ifndef W3_CUT            // diagnostic builds (scripts/wgrad3_var.hip): 1 no DMA in the loop, 2 no fragment reads, 4 no transforms
#define W3_CUT 0
#endif
constexpr int G3_DY = 64 * 64;                   // floats of dy per buffer
constexpr int G3_X = 3 * 64 * 64;                // floats of x per buffer
constexpr int G3_XE = 6 * 256;                   // floats of extras per buffer
constexpr int G3_XB = G3_X + G3_XE;              // x + extras per buffer: 13824 floats = 55296 bytes (< 64 KiB: an immediate offset)
constexpr int G3_X0 = 2 * G3_DY;                 // first x buffer
struct W3Geom { int np, S, slots_img, nseg, adv; };
// The transforms of one dy piece g = (g0..g3) and one x piece c = (c0..c3) as packed adds.  Inline assembly because the compiler only
// folds some of the half selections / negations into the instruction's modifiers (the others cost a v_xor or v_mov each).  The
// compiler's hazard recogniser does not look into inline assembly, and a VALU result needs two wait states before an MFMA may read
// it as A or B: the blocks end in the s_nop that guarantees them for their last result.
//   a01 = (g0+g1, g0-g1)   b01 = (c0+c1, c1-c0)   bx = (c2-c0, c1-c3)
__device__ __forceinline__ void w3_xform_first(f32x2 g01, f32x2 c01, f32x2 c23, f32x2& a01, f32x2& b01, f32x2& bx) {
    asm("v_pk_add_f32 %0, %3, %3 op_sel:[0,1] op_sel_hi:[0,1] neg_hi:[0,1]\n\t"
        "v_pk_add_f32 %1, %4, %4 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]\n\t"
        "v_pk_add_f32 %2, %5, %4 neg_lo:[0,1] neg_hi:[1,0]\n\t"
        "s_nop 1"
        : "=&v"(a01), "=&v"(b01), "=&v"(bx) : "v"(g01), "v"(c01), "v"(c23));
}
//   a23 = (g2+g3, g2-g3)   b23 = (c2+c3, c3-c2)
__device__ __forceinline__ void w3_xform_second(f32x2 g23, f32x2 c23, f32x2& a23, f32x2& b23) {
    asm("v_pk_add_f32 %0, %2, %2 op_sel:[0,1] op_sel_hi:[0,1] neg_hi:[0,1]\n\t"
        "v_pk_add_f32 %1, %3, %3 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]\n\t"
        "s_nop 1"
        : "=&v"(a23), "=&v"(b23) : "v"(g23), "v"(c23));
}

__global__ __launch_bounds__(768) void conv3x3_wgrad_wino3_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                                  float* __restrict__ slab, int N, int Cin, int H, int W, int Cout,
                                                                  W3Geom geo, int segs_per_split) {
    __shared__ __attribute__((aligned(256))) float lds[2 * G3_DY + 2 * G3_XB];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lk = lane >> 5;
    int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    if ((gridDim.z & 7) == 0) {             // the (ci, co) tiles of a split on one XCD (see conv3x3_wgrad_kernel)
        const int nxy = gridDim.x * gridDim.y;
        const int b = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
        const int k = b & 7, slot = b >> 3;
        bz = k + 8 * (slot / nxy);
        const int xy = slot - (slot / nxy) * nxy;
        bx = xy % gridDim.x;
        by = xy / gridDim.x;
    }
    const int ci0 = bx * 64, co0 = by * 64, split = bz;
    const int iHW = H * W;
    // consumer role
    const int kh = wave >> 2, wco = ((wave >> 1) & 1) * 32, wci = (wave & 1) * 32;
    // DMA role
    const int wq = wave >> 2, wr = wave & 3;

    f32x16 acc[4];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;

    const int sbeg = split * segs_per_split;
    const int send = min(geo.nseg, sbeg + segs_per_split);
    const __amdgpu_buffer_rsrc_t dyrs = __builtin_amdgcn_make_buffer_rsrc((void*)dy, 0, (int)((long)N * Cout * iHW * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, (int)((long)N * Cin * iHW * 4), 0x00020000);
    constexpr unsigned OOB = 0xFFFFFFF0u;

    // ---- slot states: (image n, row h, piece k) of the stream position this lane moves; k < 0: in front of the stream
    struct Slot { int n, h, k; };
    auto decode = [&](int pos) __attribute__((always_inline)) {
        Slot s;
        if (pos < 0) { s.n = 0; s.h = 0; s.k = pos; return s; }
        s.n = pos / geo.slots_img;
        const int r = pos - s.n * geo.slots_img;
        s.h = r / geo.S;
        s.k = r - s.h * geo.S;
        return s;
    };
    // + 16 slots, branch-free (geo.adv = ceil(16 / S) carries, 1 from W = 57 on)
    auto advance = [&](Slot& s) __attribute__((always_inline)) {
        s.k += 16;
        for (int it = 0; it < geo.adv; ++it) {
            const int c = s.k >= geo.S ? 1 : 0;
            s.k -= c ? geo.S : 0;
            s.h += c;
            const int d = s.h >= H ? 1 : 0;
            s.h -= d ? H : 0;
            s.n += d;
        }
    };
    const int my_ls = ((lane & 15) ^ (lane >> 4)) ^ (4 * wr);
    Slot ms = decode(sbeg * 16 + my_ls);
    // the wave's extra instruction (ids 64 .. 69 = which * 3 + filter row): waves 4..7 take ids 64 + wr, waves 8, 9 ids 68 + wr
    const int xt = wq == 1 ? wr : (wq == 2 && wr < 2 ? 4 + wr : -1);            // wave-uniform
    const int x_which = xt >= 3 ? 1 : 0, x_kh = xt >= 3 ? xt - 3 : xt;
    Slot es = decode(sbeg * 16 + (x_which ? 16 : -1));                          // wave-uniform: lives in scalar registers

    // ---- DMA maps.  Instruction t (0..5) of a wave is u = wq + 3t: u < 4 a dy instruction (output channels 16u + 4wr + lane/16),
    // u < 16 an x instruction (i = 4(u-4) + wr: filter row (u-4) / 4, input channels 4(i % 16) + lane/16), u >= 16 the wave's extra.
    // The loop below is compiled once per wq (a wave-uniform switch around it), so kind and filter row of every instruction are
    // compile-time constants; per lane and instruction one register: the channel's (and filter row's) BYTE offset, or - channel past
    // the end - a value that keeps every sum out of the buffer's range (tensors are < 2^31 bytes, so sums neither wrap nor come back).
    constexpr unsigned FAR = 0x80000000u;
    unsigned d_off[6];
#pragma unroll
    for (int t = 0; t < 6; ++t) {
        const int u = wq + 3 * t;
        if (u < 4) {
            const int co = co0 + 16 * u + 4 * wr + (lane >> 4);
            d_off[t] = co < Cout ? (unsigned)(co * iHW) * 4u : FAR;
        } else if (u < 16) {
            const int i = 4 * (u - 4) + wr, k3 = i >> 4, ci = ci0 + 4 * (i & 15) + (lane >> 4);
            d_off[t] = ci < Cin ? (unsigned)(ci * iHW + (k3 - 1) * W) * 4u : FAR;
        } else {
            const int ci = ci0 + lane;
            d_off[t] = (xt >= 0 && ci < Cin) ? (unsigned)(ci * iHW + (x_kh - 1) * W) * 4u : FAR;
        }
    }
    auto dma = [&](const __amdgpu_buffer_rsrc_t& rs, float* dst, unsigned vo) __attribute__((always_inline)) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)dst, 16, vo, 0, 0, 0);
    };
    auto issue = [&](auto wq_tag, int buf) __attribute__((always_inline)) {
        constexpr int WQ = decltype(wq_tag)::value;
        float* const bdy = lds + buf * G3_DY;
        float* const bx_ = lds + G3_X0 + buf * G3_XB;
        const bool mok = ms.k >= 0 && ms.k < geo.np && ms.n < N;
        const int pix = ms.h * W + 4 * ms.k;
        // the slot's byte offset per operand (and filter row: rows outside the image are out of range)
        const unsigned dyb = mok ? (unsigned)(ms.n * Cout * iHW + pix) * 4u : FAR;
        const unsigned xmid = mok ? (unsigned)(ms.n * Cin * iHW + pix) * 4u : FAR;
        const unsigned xb3[3] = {ms.h >= 1 ? xmid : FAR, xmid, ms.h + 1 < H ? xmid : FAR};
#pragma unroll
        for (int t = 0; t < 6; ++t) {
            const int u = WQ + 3 * t;                                           // compile time
            if (u < 4) dma(dyrs, bdy + (4 * u + wr) * 256, dyb + d_off[t]);
            else if (u < 16) dma(xrs, bx_ + (4 * (u - 4) + wr) * 256, xb3[(u - 4) >> 2] + d_off[t]);
            else if (xt >= 0) {
                const int row = es.h + x_kh - 1;                                // all wave-uniform
                const bool eok = es.k >= 0 && es.k < geo.np && es.n < N && row >= 0 && row < H;
                dma(xrs, bx_ + G3_X + xt * 256, (eok ? (unsigned)(es.n * Cin * iHW + es.h * W + 4 * es.k) * 4u : FAR) + d_off[t]);
            }
        }
    };
    // a row's last piece carries the next row's first columns when W % 4 != 0: zero them once the DMA has landed
    auto patch = [&](auto wq_tag, int buf) __attribute__((always_inline)) {
        constexpr int WQ = decltype(wq_tag)::value;
        if ((W & 3) == 0) return;
        float* const bdy = lds + buf * G3_DY;
        float* const bx_ = lds + G3_X0 + buf * G3_XB;
        const int nv = W - 4 * (geo.np - 1);                                    // valid elements of a row's last piece (1..3)
        if (ms.k == geo.np - 1) {
#pragma unroll
            for (int t = 0; t < 6; ++t) {
                const int u = WQ + 3 * t;
                if (u < 16) {
                    float* o = (u < 4 ? bdy + (4 * u + wr) * 256 : bx_ + (4 * (u - 4) + wr) * 256) + lane * 4;
#pragma unroll
                    for (int e = 1; e < 4; ++e)
                        if (e >= nv) o[e] = 0.f;
                }
            }
        }
        if (xt >= 0 && es.k == geo.np - 1) {
            float* o = bx_ + G3_X + xt * 256 + lane * 4;
#pragma unroll
            for (int e = 1; e < 4; ++e)
                if (e >= nv) o[e] = 0.f;
        }
    };

    // ---- fragment addresses (bytes from the start of buffer 0 of the operand; kernel constants per lane, one register each)
    const int cA = wco + li, cB = wci + li, sA = cA & 15, sB = cB & 15;
    const char* const ldsb = (const char*)lds;
    int adA[8], adC[8], adP[8], adN[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int m = 2 * j + lk, p = m - 1, q = m + 1;
        adA[j] = (cA * 64 + 4 * (m ^ sA)) * 4;
        adC[j] = (G3_X0 + kh * 4096 + cB * 64 + 4 * (m ^ sB)) * 4;
        adP[j] = (G3_X0 + (p >= 0 ? kh * 4096 + cB * 64 + 4 * (p ^ sB) : G3_X + kh * 256 + cB * 4)) * 4;
        adN[j] = (G3_X0 + (q <= 15 ? kh * 4096 + cB * 64 + 4 * (q ^ sB) : G3_X + (3 + kh) * 256 + cB * 4)) * 4;
    }
    // one segment out of buffer CUR (compile time: the buffer is the reads' immediate offset)
    auto segment = [&](auto wq_tag, auto cur_tag, bool more) __attribute__((always_inline)) {
        constexpr int CUR = decltype(cur_tag)::value;
        constexpr int OA = CUR * G3_DY * 4, OX = CUR * G3_XB * 4;
        if (more && !(W3_CUT & 1)) {
            advance(ms);
            advance(es);
            issue(wq_tag, CUR ^ 1);
        }
        // (the neighbours' pieces are read whole although only one element of each is used: a 4-byte read across channel lanes
        // that sit a multiple of 16 bytes apart is a 4-way bank conflict, the 16-byte read is conflict-free and costs the same issue slot)
        f32x4 rg, rc, ng, nc, rp4, rn4, np4, nn4;
        rg = *(const f32x4*)(ldsb + adA[0] + OA); rc = *(const f32x4*)(ldsb + adC[0] + OX);
        rp4 = *(const f32x4*)(ldsb + adP[0] + OX); rn4 = *(const f32x4*)(ldsb + adN[0] + OX);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            ng = rg; nc = rc; np4 = rp4; nn4 = rn4;
            if (j + 1 < 8 && !(W3_CUT & 2)) {
                ng = *(const f32x4*)(ldsb + adA[j + 1] + OA);
                nc = *(const f32x4*)(ldsb + adC[j + 1] + OX);
                np4 = *(const f32x4*)(ldsb + adP[j + 1] + OX);
                nn4 = *(const f32x4*)(ldsb + adN[j + 1] + OX);
            }
            __builtin_amdgcn_sched_barrier(0);      // the next double-step's ds_reads stay above this one's MFMAs
            // the x piece c0..c3 = columns 4k .. 4k+3, p = column 4k-1, n = column 4k+4; the dy piece g0..g3
            // pair (4k, 4k+1):   a = g0, g0+g1, g0-g1, g1   b = p-c1, c0+c1, c1-c0, c2-c0      (the 1/2 of a1, a2: output transform)
            // pair (4k+2, 4k+3): a = g2, g2+g3, g2-g3, g3   b = c1-c3, c2+c3, c3-c2, n-c2
            const f32x2 g01 = {rg[0], rg[1]}, g23 = {rg[2], rg[3]}, c01 = {rc[0], rc[1]}, c23 = {rc[2], rc[3]};
            f32x2 a0, a1, b0, b1, bx2;
            float e0, e1;
            const float rp = rp4[3], rn = rn4[0];
            if (W3_CUT & 4) { a0 = g01; a1 = g23; b0 = c01; b1 = c23; bx2 = c01; e0 = rp; e1 = rn; }
            else {
                w3_xform_first(g01, c01, c23, a0, b0, bx2);
                w3_xform_second(g23, c23, a1, b1);
                e0 = rp - rc[1];
                e1 = rn - rc[2];
            }
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(rg[0], e0, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[0], b0[0], acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[1], b0[1], acc[2], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(rg[1], bx2[0], acc[3], 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(rg[2], bx2[1], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[0], b1[0], acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[1], b1[1], acc[2], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(rg[3], e1, acc[3], 0, 0, 0);
            rg = ng; rc = nc; rp4 = np4; rn4 = nn4;
        }
        if (more && !(W3_CUT & 1)) {
            __builtin_amdgcn_s_waitcnt(0x0F70);     // the next segment's DMAs have landed
            patch(wq_tag, CUR ^ 1);
        }
        __syncthreads();                            // next buffer complete, this one free
    };

    auto run = [&](auto wq_tag) __attribute__((always_inline)) {
        if (sbeg < send) {
            issue(wq_tag, 0);
            __builtin_amdgcn_s_waitcnt(0x0F70);
            patch(wq_tag, 0);
        }
        __syncthreads();
        for (int g = sbeg; g < send; g += 2) {
            segment(wq_tag, std::integral_constant<int, 0>{}, g + 1 < send);
            if (g + 1 < send) segment(wq_tag, std::integral_constant<int, 1>{}, g + 2 < send);
        }
    };
    if (wq == 0) run(std::integral_constant<int, 0>{});
    else if (wq == 1) run(std::integral_constant<int, 1>{});
    else run(std::integral_constant<int, 2>{});
    // output transform with the filter-side 1/2:  dw(kw=0) = M0 + (M1+M2)/2   dw(1) = (M1-M2)/2   dw(2) = (M1+M2)/2 + M3
    const long plane = (long)Cout * Cin;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int co = co0 + wco + (r & 3) + 8 * (r >> 2) + 4 * lk;
        const int ci = ci0 + wci + li;
        if (co < Cout && ci < Cin) {
            const float m0 = acc[0][r], m1 = 0.5f * acc[1][r], m2 = 0.5f * acc[2][r], m3 = acc[3][r];
            float* o = slab + ((long)split * 9 + kh * 3) * plane + (long)co * Cin + ci;
            o[0] = (m0 + m1) + m2;
            o[plane] = m1 - m2;
            o[2 * plane] = (m1 + m2) + m3;
        }
    }
}

// ---------------------------------------------------------------- weight gradient F(3,2) x F(3,2): rows in PAIRS, eight waves (round 4)
// conv3x3_wgrad_wino3_kernel applies the minimal-filtering transform along the row only and runs the three filter rows as
// three separate contractions: per output-row pair 3 x 2 = 6 row products.  The same transform ACROSS rows needs 4: with
// r = 2p, the x rows r-1 .. r+2 and the dy rows r, r+1 of a pair p
//     X0 = x(r-1) - x(r+1)   X1 = x(r) + x(r+1)      X2 = x(r+1) - x(r)     X3 = x(r+2) - x(r)
//     D0 = d(r)              D1 = d(r) + d(r+1)      D2 = d(r) - d(r+1)     D3 = d(r+1)
//     V_v = (row-wise F(3,2) contraction of D_v with X_v, exactly the one-row kernel's)      v = 0..3
//     dw(kh=0) = V0 + (V1+V2)/2      dw(1) = (V1-V2)/2      dw(2) = (V1+V2)/2 + V3
// so the matrix pipe does 4/6 of the one-row kernel's work (4/9 of the direct contraction), x 8/7 and 16/15 where the height
// is odd (the pair's missing row is zero).  The row combinations are one packed add per register pair on top of that kernel's
// transforms; a wave = (v, input-channel half) owns 64 co x 32 ci x 4 row-wise points = 8 accumulator tiles, so the x-side
// work (the larger half) is done once per 16 MFMAs: 13 - 17 VALU and 8 - 10 LDS reads per 16 MFMAs.
// Piece stream as there, over row PAIRS (np = ceil(W/4) slots + one all-zero slot per pair), in segments of 8 slots.
// LDS (floats; every region's second buffer 8192 floats = 32 KiB further, the reads' immediate offset):
//     x   [4 rows][64 ci][8 pos][4]            0 .. 8192     (row plane 8 KiB)
//     xe  [2: slot -1 / slot 8][4 rows][64 ci][4]   16384 .. 18432
//     dy  [2 rows][64 co][8 pos][4]            18432 .. 22528
// position of logical slot m of channel c = m ^ ((c >> 1) & 7): the 16 lanes of a ds_read_b128 group (16 consecutive channels,
// 128 bytes apart) cover the 64 banks exactly once.  A DMA instruction fills 1 KiB = 8 channels x 8 positions; wave (wq, pw)
// moves channel group 2 wq + pw of all six planes, so a lane only ever moves ONE logical slot, (lane & 7) ^ (lane >> 4 | 4 pw).
// The slab holds [split][v][kw] planes; wgrad_reduce2d_kernel adds the splits and applies the row-pair output transform.
#ifndef W4_CUT            // diagnostic builds: 1 no DMA in the loop (wrong results)
#define W4_CUT 0
#endif
constexpr int G4_PLANE = 64 * 8 * 4;
constexpr int G4_BUF = 8192;
constexpr int G4_X0 = 0;
constexpr int G4_XE0 = 16384;
constexpr int G4_DY0 = 18432;
constexpr int G4_LDS = 30720;                    // 120 KiB
constexpr int G4_TAB = 64 * 8 * 8;               // + the slot table: [64 segments][8 slots][8 ints] = 16 KiB

// One block per operand and double-step: the row combination X_v = xa +- xb (D_v = da +- db) and the row-wise transform on top of it,
// all packed adds; the block's results may be MFMA operands right away, so it ends in the two wait states (see w3_xform_first).
//   c = ca +- cb      b01 = (c0+c1, c1-c0)   bx = (c2-c0, c1-c3)   b23 = (c2+c3, c3-c2)
template <bool MINUS>
__device__ __forceinline__ void w4_xform_b(f32x2 ca01, f32x2 ca23, f32x2 cb01, f32x2 cb23, f32x2& c01, f32x2& c23, f32x2& b01, f32x2& bx,
                                           f32x2& b23) {
    if (MINUS)
        asm("v_pk_add_f32 %0, %5, %7 neg_lo:[0,1] neg_hi:[0,1]\n\t"
            "v_pk_add_f32 %1, %6, %8 neg_lo:[0,1] neg_hi:[0,1]\n\t"
            "v_pk_add_f32 %2, %0, %0 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]\n\t"
            "v_pk_add_f32 %3, %1, %0 neg_lo:[0,1] neg_hi:[1,0]\n\t"
            "v_pk_add_f32 %4, %1, %1 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]\n\t"
            "s_nop 1"
            : "=&v"(c01), "=&v"(c23), "=&v"(b01), "=&v"(bx), "=&v"(b23) : "v"(ca01), "v"(ca23), "v"(cb01), "v"(cb23));
    else
        asm("v_pk_add_f32 %0, %5, %7\n\t"
            "v_pk_add_f32 %1, %6, %8\n\t"
            "v_pk_add_f32 %2, %0, %0 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]\n\t"
            "v_pk_add_f32 %3, %1, %0 neg_lo:[0,1] neg_hi:[1,0]\n\t"
            "v_pk_add_f32 %4, %1, %1 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]\n\t"
            "s_nop 1"
            : "=&v"(c01), "=&v"(c23), "=&v"(b01), "=&v"(bx), "=&v"(b23) : "v"(ca01), "v"(ca23), "v"(cb01), "v"(cb23));
}
//   a01 = (g0+g1, g0-g1)   a23 = (g2+g3, g2-g3)
__device__ __forceinline__ void w4_xform_a(f32x2 g01, f32x2 g23, f32x2& a01, f32x2& a23) {
    asm("v_pk_add_f32 %0, %2, %2 op_sel:[0,1] op_sel_hi:[0,1] neg_hi:[0,1]\n\t"
        "v_pk_add_f32 %1, %3, %3 op_sel:[0,1] op_sel_hi:[0,1] neg_hi:[0,1]\n\t"
        "s_nop 1"
        : "=&v"(a01), "=&v"(a23) : "v"(g01), "v"(g23));
}
//   g = ga +- gb, then the same
template <bool MINUS>
__device__ __forceinline__ void w4_xform_a2(f32x2 ga01, f32x2 ga23, f32x2 gb01, f32x2 gb23, f32x2& g01, f32x2& g23, f32x2& a01, f32x2& a23) {
    if (MINUS)
        asm("v_pk_add_f32 %0, %4, %6 neg_lo:[0,1] neg_hi:[0,1]\n\t"
            "v_pk_add_f32 %1, %5, %7 neg_lo:[0,1] neg_hi:[0,1]\n\t"
            "v_pk_add_f32 %2, %0, %0 op_sel:[0,1] op_sel_hi:[0,1] neg_hi:[0,1]\n\t"
            "v_pk_add_f32 %3, %1, %1 op_sel:[0,1] op_sel_hi:[0,1] neg_hi:[0,1]\n\t"
            "s_nop 1"
            : "=&v"(g01), "=&v"(g23), "=&v"(a01), "=&v"(a23) : "v"(ga01), "v"(ga23), "v"(gb01), "v"(gb23));
    else
        asm("v_pk_add_f32 %0, %4, %6\n\t"
            "v_pk_add_f32 %1, %5, %7\n\t"
            "v_pk_add_f32 %2, %0, %0 op_sel:[0,1] op_sel_hi:[0,1] neg_hi:[0,1]\n\t"
            "v_pk_add_f32 %3, %1, %1 op_sel:[0,1] op_sel_hi:[0,1] neg_hi:[0,1]\n\t"
            "s_nop 1"
            : "=&v"(g01), "=&v"(g23), "=&v"(a01), "=&v"(a23) : "v"(ga01), "v"(ga23), "v"(gb01), "v"(gb23));
}

__global__ __launch_bounds__(512) void conv3x3_wgrad_wino2d_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                                   float* __restrict__ slab, int N, int Cin, int H, int W, int Cout,
                                                                   W3Geom geo, int segs_per_split) {
    __shared__ __attribute__((aligned(256))) float lds[G4_LDS + G4_TAB];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lk = lane >> 5;
    int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    if ((gridDim.z & 7) == 0) {             // the (ci, co) tiles of a split on one XCD (see conv3x3_wgrad_kernel)
        const int nxy = gridDim.x * gridDim.y;
        const int b = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
        const int k = b & 7, slot = b >> 3;
        bz = k + 8 * (slot / nxy);
        const int xy = slot - (slot / nxy) * nxy;
        bx = xy % gridDim.x;
        by = xy / gridDim.x;
    }
    const int ci0 = bx * 64, co0 = by * 64, split = bz;
    const int iHW = H * W, HP = geo.slots_img / geo.S;
    // consumer role: transform point across rows, input-channel half
    const int v = wave >> 1, cih = wave & 1;
    // DMA role: channel group of the six planes; one extras instruction (which neighbour, x row)
    const int pw = wave & 1, grp = wave;
    const int e_which = wave >> 2, e_row = wave & 3;

    f32x16 acc[2][4];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[s][q][r] = 0.f;

    const int sbeg = split * segs_per_split;
    const int send = min(geo.nseg, sbeg + segs_per_split);
    const int nloc = send - sbeg;
    const __amdgpu_buffer_rsrc_t dyrs = __builtin_amdgcn_make_buffer_rsrc((void*)dy, 0, (int)((long)N * Cout * iHW * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, (int)((long)N * Cin * iHW * 4), 0x00020000);
    constexpr unsigned FAR = 0x80000000u;
    const unsigned W4 = (unsigned)W * 4u;

    // ---- slot table.  What a DMA needs of a stream slot - the byte offsets of its piece in the two dy rows and the four x rows of its
    // row pair (FAR: outside the image or the stream) and whether it is a row's last piece - does not depend on the channel: it is
    // computed ONCE per workgroup (one thread per (segment, slot): two divisions) into a ring of 64 segments in LDS, half a ring at a
    // time, instead of by every lane of every wave for every segment (45 of a segment's 115 vector instructions per wave before).
    int4* const tab = (int4*)(lds + G4_LDS);
    auto fill = [&](int q0, int nq) __attribute__((always_inline)) {      // relative segments q0 .. q0 + nq - 1, nq * 8 <= 512
        if (tid < nq * 8) {
            const int q = q0 + (tid >> 3), slot = tid & 7;
            const unsigned pos = (unsigned)(sbeg + q) * 8u + slot;
            const int n = pos / (unsigned)geo.slots_img;
            const int r = pos - n * geo.slots_img;
            const int hp = r / geo.S, k = r - hp * geo.S;
            const bool ok = k < geo.np && n < N;
            const int row = 2 * hp;
            const unsigned pix = (unsigned)(row * W + 4 * k);
            const unsigned a0 = ok ? ((unsigned)(n * Cout * iHW) + pix) * 4u : FAR;
            const unsigned b1 = ok ? ((unsigned)(n * Cin * iHW) + pix) * 4u : FAR;
            int4 t0, t1;
            t0.x = a0;
            t0.y = (ok && row + 1 < H) ? a0 + W4 : FAR;
            t0.z = (ok && row >= 1) ? b1 - W4 : FAR;
            t0.w = b1;
            t1.x = (ok && row + 1 < H) ? b1 + W4 : FAR;
            t1.y = (ok && row + 2 < H) ? b1 + 2u * W4 : FAR;
            t1.z = k == geo.np - 1 ? 1 : 0;
            t1.w = 0;
            tab[((q & 63) * 8 + slot) * 2] = t0;
            tab[((q & 63) * 8 + slot) * 2 + 1] = t1;
        }
    };
    const int my_ls = (lane & 7) ^ ((lane >> 4) | (4 * pw));
    struct Entry { int4 t0, t1; };
    auto entry = [&](int q) __attribute__((always_inline)) {
        Entry e;
        e.t0 = tab[((q & 63) * 8 + my_ls) * 2];
        e.t1 = tab[((q & 63) * 8 + my_ls) * 2 + 1];
        return e;
    };
    // the extras' slot (wave-uniform, scalar registers): slot -1 / slot 8 of the segment
    struct Slot { int n, hp, k; };
    auto decode = [&](int pos) __attribute__((always_inline)) {
        Slot s;
        if (pos < 0) { s.n = 0; s.hp = 0; s.k = pos; return s; }
        s.n = pos / geo.slots_img;
        const int r = pos - s.n * geo.slots_img;
        s.hp = r / geo.S;
        s.k = r - s.hp * geo.S;
        return s;
    };
    auto advance = [&](Slot& s) __attribute__((always_inline)) {      // + 8 slots, branch-free (geo.adv = ceil(8 / S) carries)
        s.k += 8;
        for (int it = 0; it < geo.adv; ++it) {
            const int c = s.k >= geo.S ? 1 : 0;
            s.k -= c ? geo.S : 0;
            s.hp += c;
            const int d = s.hp >= HP ? 1 : 0;
            s.hp -= d ? HP : 0;
            s.n += d;
        }
    };
    Slot es = decode(sbeg * 8 + (e_which ? 8 : -1));
    unsigned offA, offB, offE;
    {
        const int co = co0 + 8 * grp + (lane >> 3), ci = ci0 + 8 * grp + (lane >> 3), cie = ci0 + lane;
        offA = co < Cout ? (unsigned)(co * iHW) * 4u : FAR;
        offB = ci < Cin ? (unsigned)(ci * iHW) * 4u : FAR;
        offE = cie < Cin ? (unsigned)(cie * iHW) * 4u : FAR;
    }
    auto dma = [&](const __amdgpu_buffer_rsrc_t& rs, float* dst, unsigned vo) __attribute__((always_inline)) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)dst, 16, vo, 0, 0, 0);
    };
    auto issue = [&](int buf, const Entry& e) __attribute__((always_inline)) {
        float* const bxm = lds + G4_X0 + buf * G4_BUF + grp * 256;
        float* const bxe = lds + G4_XE0 + buf * G4_BUF + e_which * 1024 + e_row * 256;
        float* const bdy = lds + G4_DY0 + buf * G4_BUF + grp * 256;
        dma(dyrs, bdy, (unsigned)e.t0.x + offA);
        dma(dyrs, bdy + G4_PLANE, (unsigned)e.t0.y + offA);
        dma(xrs, bxm, (unsigned)e.t0.z + offB);
        dma(xrs, bxm + G4_PLANE, (unsigned)e.t0.w + offB);
        dma(xrs, bxm + 2 * G4_PLANE, (unsigned)e.t1.x + offB);
        dma(xrs, bxm + 3 * G4_PLANE, (unsigned)e.t1.y + offB);
        const int row = 2 * es.hp + e_row - 1;                         // wave-uniform
        const bool eok = es.k >= 0 && es.k < geo.np && es.n < N && row >= 0 && row < H;
        dma(xrs, bxe, (eok ? (unsigned)(es.n * Cin * iHW + row * W + 4 * es.k) * 4u : FAR) + offE);
    };
    // a row's last piece carries the next row's first columns when W % 4 != 0: zero them once the DMA has landed
    auto patch = [&](int buf, int last_piece) __attribute__((always_inline)) {
        if ((W & 3) == 0) return;
        const int nv = W - 4 * (geo.np - 1);                            // valid elements of a row's last piece (1..3)
        if (last_piece) {
            float* const bxm = lds + G4_X0 + buf * G4_BUF + grp * 256 + lane * 4;
            float* const bdy = lds + G4_DY0 + buf * G4_BUF + grp * 256 + lane * 4;
#pragma unroll
            for (int e = 1; e < 4; ++e)
                if (e >= nv) {
                    bdy[e] = 0.f; bdy[G4_PLANE + e] = 0.f;
                    bxm[e] = 0.f; bxm[G4_PLANE + e] = 0.f; bxm[2 * G4_PLANE + e] = 0.f; bxm[3 * G4_PLANE + e] = 0.f;
                }
        }
        if (es.k == geo.np - 1) {
            float* o = lds + G4_XE0 + buf * G4_BUF + e_which * 1024 + e_row * 256 + lane * 4;
#pragma unroll
            for (int e = 1; e < 4; ++e)
                if (e >= nv) o[e] = 0.f;
        }
    };

    // ---- fragment addresses (bytes; kernel constants per lane)
    const int cB = cih * 32 + li, swB = (cB >> 1) & 7;
    const char* const ldsb = (const char*)lds;
    int adA[2][4], adC[4], adP[4], adN[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int m = 2 * j + lk;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int cA = 32 * s + li, swA = (cA >> 1) & 7;
            adA[s][j] = (G4_DY0 + cA * 32 + 4 * (m ^ swA)) * 4;
        }
        adC[j] = (G4_X0 + cB * 32 + 4 * (m ^ swB)) * 4;
        adP[j] = (G4_X0 + cB * 32 + 4 * (((m - 1) & 7) ^ swB)) * 4;          // j = 0, lk = 0: not used (adP0 below)
        adN[j] = (G4_X0 + cB * 32 + 4 * (((m + 1) & 7) ^ swB)) * 4;          // j = 3, lk = 1: not used (adN3 below)
    }
    auto ld = [&](int addr, int imm) __attribute__((always_inline)) { return *(const f32x4*)(ldsb + addr + imm); };
    auto ldf = [&](int addr, int imm) __attribute__((always_inline)) { return *(const float*)(ldsb + addr + imm); };

    auto run = [&](auto v_tag) __attribute__((always_inline)) {
        constexpr int V = decltype(v_tag)::value;
        constexpr int RA = (V == 0 ? 0 : V == 1 ? 1 : V == 2 ? 2 : 3), RB = (V == 0 ? 2 : V == 1 ? 2 : 1);     // X_v = x[RA] +- x[RB]
        constexpr bool PLUS = V == 1;
        // the first slot's left neighbour / the last slot's right neighbour: extras for one lane half, the main plane for the other
        const int adP0a = lk ? adP[0] + RA * 8192 : (G4_XE0 + RA * 256 + cB * 4) * 4;
        const int adP0b = lk ? adP[0] + RB * 8192 : (G4_XE0 + RB * 256 + cB * 4) * 4;
        const int adN3a = lk ? (G4_XE0 + 1024 + RA * 256 + cB * 4) * 4 : adN[3] + RA * 8192;
        const int adN3b = lk ? (G4_XE0 + 1024 + RB * 256 + cB * 4) * 4 : adN[3] + RB * 8192;

        auto segment = [&](auto cur_tag, int q, bool more) __attribute__((always_inline)) {
            constexpr int OB = decltype(cur_tag)::value * G4_BUF * 4;
            // raw fragments of one double-step: x rows RA / RB (piece, left neighbour, right neighbour), dy rows of the two co sub-tiles
            // (the neighbours' single elements as 4-byte reads: two lanes per bank, a quarter of a 16-byte read's LDS cycles and registers)
            struct Raw { f32x4 ca, cb, g0[2], g1[2]; float pa, pb, na, nb; };
            auto load = [&](int j, Raw& r) __attribute__((always_inline)) {
                r.ca = ld(adC[j], OB + RA * 8192); r.cb = ld(adC[j], OB + RB * 8192);
                r.pa = j == 0 ? ldf(adP0a, OB + 12) : ldf(adP[j], OB + RA * 8192 + 12);
                r.pb = j == 0 ? ldf(adP0b, OB + 12) : ldf(adP[j], OB + RB * 8192 + 12);
                r.na = j == 3 ? ldf(adN3a, OB) : ldf(adN[j], OB + RA * 8192);
                r.nb = j == 3 ? ldf(adN3b, OB) : ldf(adN[j], OB + RB * 8192);
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    if (V != 3) r.g0[s] = ld(adA[s][j], OB);
                    if (V != 0) r.g1[s] = ld(adA[s][j], OB + 8192);
                }
            };
            Raw cur, nxt;
            load(0, cur);                               // first: their latency runs under the DMA issue below
            if (q > 0 && (q & 31) == 0) fill(q + 32, 32);      // the ring's other half: segments q + 32 .. q + 63 (first read in segment q + 30)
            int last_piece = 0;
            if (more && !(W4_CUT & 1)) {
                const Entry e = entry(q + 1);
                advance(es);
                issue(decltype(cur_tag)::value ^ 1, e);
                last_piece = e.t1.z;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (j + 1 < 4) load(j + 1, nxt);
                __builtin_amdgcn_sched_barrier(0);      // the next double-step's ds_reads stay above this one's MFMAs
                // X_v = x[RA] +- x[RB]: the piece c0..c3 (columns 4k .. 4k+3), p = column 4k-1, n = column 4k+4; row-wise transform:
                // pair (4k, 4k+1): b = p-c1, c0+c1, c1-c0, c2-c0      pair (4k+2, 4k+3): b = c1-c3, c2+c3, c3-c2, n-c2
                f32x2 c01, c23, b0, bx2, b1;
                w4_xform_b<!PLUS>((f32x2){cur.ca[0], cur.ca[1]}, (f32x2){cur.ca[2], cur.ca[3]}, (f32x2){cur.cb[0], cur.cb[1]},
                                  (f32x2){cur.cb[2], cur.cb[3]}, c01, c23, b0, bx2, b1);
                const float p = PLUS ? cur.pa + cur.pb : cur.pa - cur.pb;
                const float n = PLUS ? cur.na + cur.nb : cur.na - cur.nb;
                const float e0 = p - c01[1], e1 = n - c23[0];
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    // D_v, then the dy piece g0..g3: a = g0, g0+g1, g0-g1, g1 | g2, g2+g3, g2-g3, g3   (the 1/2 of a1, a2: output transform)
                    f32x2 g01, g23, a0, a1;
                    if (V == 0 || V == 3) {
                        const f32x4 g = V == 0 ? cur.g0[s] : cur.g1[s];
                        g01 = (f32x2){g[0], g[1]}; g23 = (f32x2){g[2], g[3]};
                        w4_xform_a(g01, g23, a0, a1);
                    } else {
                        w4_xform_a2<V == 2>((f32x2){cur.g0[s][0], cur.g0[s][1]}, (f32x2){cur.g0[s][2], cur.g0[s][3]},
                                            (f32x2){cur.g1[s][0], cur.g1[s][1]}, (f32x2){cur.g1[s][2], cur.g1[s][3]}, g01, g23, a0, a1);
                    }
                    acc[s][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(g01[0], e0, acc[s][0], 0, 0, 0);
                    acc[s][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[0], b0[0], acc[s][1], 0, 0, 0);
                    acc[s][2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[1], b0[1], acc[s][2], 0, 0, 0);
                    acc[s][3] = __builtin_amdgcn_mfma_f32_32x32x2f32(g01[1], bx2[0], acc[s][3], 0, 0, 0);
                    acc[s][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(g23[0], bx2[1], acc[s][0], 0, 0, 0);
                    acc[s][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[0], b1[0], acc[s][1], 0, 0, 0);
                    acc[s][2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[1], b1[1], acc[s][2], 0, 0, 0);
                    acc[s][3] = __builtin_amdgcn_mfma_f32_32x32x2f32(g23[1], e1, acc[s][3], 0, 0, 0);
                }
                cur = nxt;
            }
            if (more) {
                __builtin_amdgcn_s_waitcnt(0x0F70);     // the next segment's DMAs have landed
                patch(decltype(cur_tag)::value ^ 1, last_piece);
            }
            __syncthreads();                            // next buffer complete, this one free
        };
        for (int q = 0; q < nloc; q += 2) {
            segment(std::integral_constant<int, 0>{}, q, q + 1 < nloc);
            if (q + 1 < nloc) segment(std::integral_constant<int, 1>{}, q + 1, q + 2 < nloc);
        }
    };

    fill(0, 64);
    __syncthreads();
    if (nloc > 0) {
        const Entry e = entry(0);
        issue(0, e);
        __builtin_amdgcn_s_waitcnt(0x0F70);
        patch(0, e.t1.z);
    }
    __syncthreads();
    if (v == 0) run(std::integral_constant<int, 0>{});
    else if (v == 1) run(std::integral_constant<int, 1>{});
    else if (v == 2) run(std::integral_constant<int, 2>{});
    else run(std::integral_constant<int, 3>{});
    // row-wise output transform with the filter-side 1/2 (and the row-pair transform's 1/2 for V1, V2):
    //   V(kw=0) = M0 + (M1+M2)/2   V(1) = (M1-M2)/2   V(2) = (M1+M2)/2 + M3
    const long plane = (long)Cout * Cin;
    const float sc = (v == 1 || v == 2) ? 0.5f : 1.0f;
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = co0 + 32 * s + (r & 3) + 8 * (r >> 2) + 4 * lk;
            const int ci = ci0 + cB;
            if (co < Cout && ci < Cin) {
                const float m0 = sc * acc[s][0][r], m1 = 0.5f * sc * acc[s][1][r], m2 = 0.5f * sc * acc[s][2][r], m3 = sc * acc[s][3][r];
                float* o = slab + ((long)split * 12 + v * 3) * plane + (long)co * Cin + ci;
                o[0] = (m0 + m1) + m2;
                o[plane] = m1 - m2;
                o[2 * plane] = (m1 + m2) + m3;
            }
        }
}

// adds the splits of conv3x3_wgrad_wino2d_kernel's slab [split][v][kw][co][ci] in a fixed order and applies the row-pair output
// transform: dw(kh=0) = V0 + (V1 + V2), dw(1) = V1 - V2, dw(2) = (V1 + V2) + V3 (the halves are already in V1, V2)
template <int G>
__global__ __launch_bounds__(64 * G) void wgrad_reduce2d_kernel(const float* __restrict__ slab, float* __restrict__ dw, int Cout, int Cin,
                                                                int splits) {
    __shared__ f32x4 red[G][4][64];
    const long plane = (long)Cout * Cin;
    const long per_kw = plane / 4;                  // float4s of a plane (Cout * Cin % 4 == 0: the caller checks)
    const int lane = threadIdx.x & 63, g = threadIdx.x >> 6;
    const long o = (long)blockIdx.x * 64 + lane;    // (kw, float4 of the plane)
    const bool ok = o < 3 * per_kw;
    const int kw = ok ? (int)(o / per_kw) : 0;
    const long r4 = ok ? o - (long)kw * per_kw : 0;
    f32x4 s[4];
#pragma unroll
    for (int vv = 0; vv < 4; ++vv) s[vv] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (ok)
        for (int sp = g; sp < splits; sp += G)
#pragma unroll
            for (int vv = 0; vv < 4; ++vv) s[vv] += *(const f32x4*)(slab + ((long)sp * 12 + vv * 3 + kw) * plane + r4 * 4);
#pragma unroll
    for (int vv = 0; vv < 4; ++vv) red[g][vv][lane] = s[vv];
    __syncthreads();
    if (g == 0 && ok) {
        f32x4 t[4];
#pragma unroll
        for (int vv = 0; vv < 4; ++vv) {
            f32x4 u[G];
#pragma unroll
            for (int i = 0; i < G; ++i) u[i] = red[i][vv][lane];
#pragma unroll
            for (int w = 1; w < G; w *= 2)            // fixed balanced tree
#pragma unroll
                for (int i = 0; i + w < G; i += 2 * w) u[i] += u[i + w];
            t[vv] = u[0];
        }
        const f32x4 s12 = t[1] + t[2];
        const f32x4 k0 = t[0] + s12, k1 = t[1] - t[2], k2 = s12 + t[3];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float* d = dw + (r4 * 4 + e) * 9 + kw;
            d[0] = k0[e];
            d[3] = k1[e];
            d[6] = k2[e];
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

extern "C" int vocr_conv3x3_wino_supported(int cin, int cout) { return cin >= 1 && cout % 4 == 0 ? 1 : 0; }

// a pack = 12 transformed rows per contraction channel, followed by the direct pack's 9 rows per channel (for the tail pieces)
extern "C" size_t vocr_conv3x3_wino_pack_floats(int cout, int cin) { return (size_t)cout * cin * (wino4_for(cout) ? 27 : 21); }

extern "C" int vocr_conv3x3_wino_pack_weights(const float* w, float* wpack_fwd, float* wpack_dgrad, int cout, int cin, void* stream) {
    VOCR_CHECK_ARG(w && (wpack_fwd || wpack_dgrad), "vocr_conv3x3_wino_pack_weights: null pointer");
    VOCR_CHECK_ARG(cout > 0 && cin > 0, "vocr_conv3x3_wino_pack_weights: bad shape");
    const int total = cout * cin * 3;
    // the forward pack serves a convolution with `cout` outputs, the data-gradient pack one with `cin` outputs: each in the format of ITS kernel
    const bool f4f = wpack_fwd && wino4_for(cout), f4d = wpack_dgrad && wino4_for(cin);
    hipStream_t s = (hipStream_t)stream;
    if (f4f || f4d) wino4_pack_kernel<<<vocr_cdiv(total, 256), 256, 0, s>>>(w, f4f ? wpack_fwd : nullptr, f4d ? wpack_dgrad : nullptr, cout, cin);
    if ((wpack_fwd && !f4f) || (wpack_dgrad && !f4d))
        wino_pack_kernel<<<vocr_cdiv(total, 256), 256, 0, s>>>(w, f4f ? nullptr : wpack_fwd, f4d ? nullptr : wpack_dgrad, cout, cin, wino_pack_x4());
    VOCR_CHECK_LAUNCH("vocr_conv3x3_wino_pack_weights");
    // the direct pack behind the transformed rows
    int rc = VOCR_OK;
    if (wpack_fwd) rc = vocr_conv3x3_pack_weights(w, wpack_fwd + (size_t)cout * cin * (f4f ? 18 : 12), nullptr, cout, cin, stream);
    if (rc == VOCR_OK && wpack_dgrad) rc = vocr_conv3x3_pack_weights(w, nullptr, wpack_dgrad + (size_t)cout * cin * (f4d ? 18 : 12), cout, cin, stream);
    return rc;
}

extern "C" int vocr_conv3x3_wino_fwd(const float* x, const float* wpack, const float* bias, float* y, int n, int cin, int h,
                                     int w, int cout, void* stream) {
    VOCR_CHECK_ARG(x && wpack && y, "vocr_conv3x3_wino_fwd: null pointer");
    VOCR_CHECK_ARG(n > 0 && cin > 0 && h > 0 && w > 0 && cout > 0, "vocr_conv3x3_wino_fwd: bad shape");
    VOCR_CHECK_ARG(cout % 4 == 0 && ((((uintptr_t)wpack) & 15) == 0), "vocr_conv3x3_wino_fwd: needs Cout %% 4 == 0 and a 16-byte aligned pack");
    VOCR_CHECK_ARG((long)n * (cin > cout ? cin : cout) * h * w < (1l << 31), "vocr_conv3x3_wino_fwd: tensor exceeds 2^31 elements");
    const bool f43 = wino4_for(cout);
    VOCR_CHECK_ARG(!f43 || (long)n * (cin > cout ? cin : cout) * h * w < (1l << 29), "vocr_conv3x3_wino_fwd: the F(4,3) kernel needs tensors below 2^29 elements");
    WGeom geo;
    geo.T = vocr_cdiv(w, f43 ? 4 : 2);
    geo.S = geo.T + 1;
    geo.slots_img = h * geo.S;
    geo.nslot = n * geo.slots_img;
    geo.nseg = vocr_cdiv(geo.nslot, TS);
    const float* zp = wino_zero_page_ptr();
    VOCR_CHECK_ARG(zp != nullptr, "vocr_conv3x3_wino_fwd: no device zero page");
    hipStream_t s = (hipStream_t)stream;
    const float* wdirect = wpack + (size_t)cin * (f43 ? 18 : 12) * cout;
    // VOCR_CONV_TAIL: 1 (default) the last partial round of tiles is cut into direct-form pieces that lead the launch, 0 whole tiles only
    static const int tail_mode = VOCR_EXPERIMENT_INT("VOCR_CONV_TAIL", 1);
    int ncu = 256;
    {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ncu = prop.multiProcessorCount;
    }
#define VOCR_WINO_LAUNCH(KERNEL, THREADS, CO_T, NSEG, CO_TILES)                                                             \
    do {                                                                                                                    \
        const int tiles = vocr_cdiv(geo.nseg, NSEG) * (CO_TILES), rem = tiles % ncu;                                        \
        /* a piece is 1/16 of a tile in the direct form (1.5x the multiplications) and latency-bound when K is short:   */  \
        /* measured worth it up to a quarter round of tiles, up to half a round from 128 input channels on               */  \
        const bool cut = tail_mode == 1 && tiles > ncu && rem > 0 && (rem <= ncu / 4 || (rem <= ncu / 2 && cin >= 128));   \
        const int n_main = cut ? tiles - rem : tiles, n_tail = cut ? rem * (CO_T / 32) * 2 * NSEG : 0;                      \
        KERNEL<<<dim3(n_tail + n_main), THREADS, 0, s>>>(x, wpack, bias, y, zp, n, cin, h, w, cout, geo, (CO_TILES), wdirect, n_tail, n_main); \
    } while (0)
    const int wino2 = wino2_mode();
    if (f43) {
        // one 4-wave workgroup per CU; a piece of the tail is 1/(CO_T/32 * 4 NSEG) of a tile
        // eight waves (two per SIMD, 128 channels x 4 segments per workgroup) unless that leaves CUs without a workgroup: then four waves
        // (x 2 segments: twice the tiles).  VOCR_CONV_WINO4=3 / 4: always four / always eight (experiments)
        const int t8 = vocr_cdiv(geo.nseg, 4) * vocr_cdiv(cout, 128);
        // two 4-wave workgroups per CU (stages of two channels, independent barriers: 1 - 3 % faster than one 8-wave workgroup) when there
        // are at least two tiles per CU; VOCR_CONV_WINO4=5 / 6: always / never (experiments)
        const bool x2 = wino4_mode() != 6 && wino4_mode() != 3 && wino4_mode() != 4 &&
                        (wino4_mode() == 5 || (cout > 64 ? vocr_cdiv(geo.nseg, 2) * vocr_cdiv(cout, 128) : vocr_cdiv(geo.nseg, 4)) >= 2 * ncu);
        const int nw = (wino4_mode() == 3 || x2) ? 4 : wino4_mode() == 4 ? 8 : (t8 >= ncu ? 8 : 4);
        const int tiles4 = cout > 64 ? vocr_cdiv(geo.nseg, nw / 2) * vocr_cdiv(cout, 128) : vocr_cdiv(geo.nseg, nw);
        const int slots4 = x2 ? 2 * ncu : ncu;              // workgroups resident at a time
        const int rem4 = tiles4 % slots4;
        const bool cut4 = tail_mode == 1 && tiles4 > slots4 && rem4 > 0 && rem4 <= slots4 / 2;
        const int n_main4 = cut4 ? tiles4 - rem4 : tiles4;
        // pieces per tile as conv3x3_wino4_body decodes them: COSUB x 4 x NSEG = 4 x 4 x nw/2 (128 channels) = 2 x 4 x nw (64) = 8 nw
        // (round 4 launched 16 nw: the second half mapped beyond the tiles and returned at once, each holding 61 - 126 KB of LDS)
        const int n_tail4 = cut4 ? rem4 * 8 * nw : 0;
        if (cout > 64) {
            if (x2) conv3x3_wino4x2_kernel_128<<<dim3(n_tail4 + n_main4), 256, 0, s>>>(x, wpack, bias, y, zp, n, cin, h, w, cout, geo, vocr_cdiv(cout, 128), wdirect, n_tail4, n_main4);
            else if (nw == 8) conv3x3_wino4w8_kernel_128<<<dim3(n_tail4 + n_main4), 512, 0, s>>>(x, wpack, bias, y, zp, n, cin, h, w, cout, geo, vocr_cdiv(cout, 128), wdirect, n_tail4, n_main4);
            else conv3x3_wino4_kernel_128<<<dim3(n_tail4 + n_main4), 256, 0, s>>>(x, wpack, bias, y, zp, n, cin, h, w, cout, geo, vocr_cdiv(cout, 128), wdirect, n_tail4, n_main4);
        } else {
            if (x2) conv3x3_wino4x2_kernel_64<<<dim3(n_tail4 + n_main4), 256, 0, s>>>(x, wpack, bias, y, zp, n, cin, h, w, cout, geo, 1, wdirect, n_tail4, n_main4);
            else if (nw == 8) conv3x3_wino4w8_kernel_64<<<dim3(n_tail4 + n_main4), 512, 0, s>>>(x, wpack, bias, y, zp, n, cin, h, w, cout, geo, 1, wdirect, n_tail4, n_main4);
            else conv3x3_wino4_kernel_64<<<dim3(n_tail4 + n_main4), 256, 0, s>>>(x, wpack, bias, y, zp, n, cin, h, w, cout, geo, 1, wdirect, n_tail4, n_main4);
        }
    } else if (wino2 && (long)n * (cin > cout ? cin : cout) * h * w < (1l << 29)) {
        if (cout > 64) VOCR_WINO_LAUNCH(conv3x3_wino2_kernel_128, 256, 128, 2, vocr_cdiv(cout, 128));
        else VOCR_WINO_LAUNCH(conv3x3_wino2_kernel_64, 256, 64, 4, 1);
    } else {
        if (cout > 64) VOCR_WINO_LAUNCH(conv3x3_wino_kernel<128>, 256, 128, 2, vocr_cdiv(cout, 128));
        else VOCR_WINO_LAUNCH(conv3x3_wino_kernel<64>, 256, 64, 4, 1);
    }
#undef VOCR_WINO_LAUNCH
    VOCR_CHECK_LAUNCH("vocr_conv3x3_wino_fwd");
    return VOCR_OK;
}

void vocr_internal_wgrad_reduce(const float* slab, float* dw, int cout, int cin, int splits, hipStream_t s);     // conv.hip

namespace {
// piece-stream geometry of conv3x3_wgrad_wino3_kernel
int wgrad_wino3_splits(int n, int cin, int h, int w, int cout, W3Geom* geo, int* segs_per_split) {
    geo->np = vocr_cdiv(w, 4);
    geo->S = geo->np + 1;
    geo->slots_img = h * geo->S;
    const long nslot = (long)n * geo->slots_img;
    geo->nseg = (int)((nslot + 15) / 16);
    geo->adv = vocr_cdiv(16, geo->S);
    const int tiles = vocr_cdiv(cin, 64) * vocr_cdiv(cout, 64);
    long s = (256 + tiles - 1) / tiles;                      // one workgroup per CU
    if (s > geo->nseg) s = geo->nseg;
    if (s < 1) s = 1;
    const int sps = (int)((geo->nseg + s - 1) / s);
    *segs_per_split = sps;
    return (geo->nseg + sps - 1) / sps;
}
// piece-stream geometry of conv3x3_wgrad_wino2d_kernel: row PAIRS, 8-slot segments
int wgrad_wino2d_splits(int n, int cin, int h, int w, int cout, W3Geom* geo, int* segs_per_split) {
    geo->np = vocr_cdiv(w, 4);
    geo->S = geo->np + 1;
    geo->slots_img = vocr_cdiv(h, 2) * geo->S;
    const long nslot = (long)n * geo->slots_img;
    geo->nseg = (int)((nslot + 7) / 8);
    geo->adv = vocr_cdiv(8, geo->S);
    const int tiles = vocr_cdiv(cin, 64) * vocr_cdiv(cout, 64);
    // one workgroup per CU for the whole launch (two, three or four rounds of shorter workgroups - smaller tails beside the
    // data-gradient kernel, more slab traffic - measured 15.77-15.91 ms per step against 15.64-15.68)
    long s = (256 + tiles - 1) / tiles;
    if (s > geo->nseg) s = geo->nseg;
    if (s < 1) s = 1;
    const int sps = (int)((geo->nseg + s - 1) / s);
    *segs_per_split = sps;
    return (geo->nseg + sps - 1) / sps;
}
// VOCR_WGRAD_WINO_DMA (experiments): 3 (default) row pairs (F(3,2) across rows too) / eight waves where Cin x Cout is a multiple of 4,
// 2 the one-row piece stream / twelve waves everywhere
int wgrad_wino_mode() {
    static const int m = VOCR_EXPERIMENT_INT("VOCR_WGRAD_WINO_DMA", 3);
    return m == 2 ? 2 : 3;
}
}  // namespace

extern "C" size_t vocr_conv3x3_wgrad_wino_workspace_bytes(int n, int cin, int h, int w, int cout) {
    if (n <= 0 || cin <= 0 || h <= 0 || w <= 0 || cout <= 0) return 0;
    int sps;
    W3Geom g3;
    if (wgrad_wino_mode() == 3 && (cout * cin) % 4 == 0) return (size_t)wgrad_wino2d_splits(n, cin, h, w, cout, &g3, &sps) * 12 * cout * cin * sizeof(float);
    return (size_t)wgrad_wino3_splits(n, cin, h, w, cout, &g3, &sps) * 9 * cout * cin * sizeof(float);
}

extern "C" int vocr_conv3x3_wgrad_wino(const float* x, const float* dy, float* dw, void* workspace, int n, int cin, int h, int w,
                                       int cout, void* stream) {
    VOCR_CHECK_ARG(x && dy && dw && workspace, "vocr_conv3x3_wgrad_wino: null pointer");
    VOCR_CHECK_ARG(n > 0 && cin >= 4 && h > 0 && w > 0 && cout > 0, "vocr_conv3x3_wgrad_wino: bad shape (needs cin >= 4)");
    VOCR_CHECK_ARG((long)n * (cin > cout ? cin : cout) * h * w < (1l << 29), "vocr_conv3x3_wgrad_wino: tensor exceeds 2^29 elements (32-bit byte offsets)");
    hipStream_t s = (hipStream_t)stream;
    int sps, splits;
    if (wgrad_wino_mode() == 3 && (cout * cin) % 4 == 0) {
        W3Geom g4;
        splits = wgrad_wino2d_splits(n, cin, h, w, cout, &g4, &sps);
        dim3 grid(vocr_cdiv(cin, 64), vocr_cdiv(cout, 64), splits);
        conv3x3_wgrad_wino2d_kernel<<<grid, 512, 0, s>>>(x, dy, (float*)workspace, n, cin, h, w, cout, g4, sps);
        VOCR_CHECK_LAUNCH("vocr_conv3x3_wgrad_wino(row pairs)");
        const long total4 = 3l * cout * cin / 4;
        const int wgs = (int)((total4 + 63) / 64);
        if (wgs < 64 && splits >= 64) wgrad_reduce2d_kernel<16><<<wgs, 1024, 0, s>>>((const float*)workspace, dw, cout, cin, splits);
        else wgrad_reduce2d_kernel<4><<<wgs, 256, 0, s>>>((const float*)workspace, dw, cout, cin, splits);
        VOCR_CHECK_LAUNCH("vocr_conv3x3_wgrad_wino(row pairs, reduce)");
        return VOCR_OK;
    }
    W3Geom g3;
    splits = wgrad_wino3_splits(n, cin, h, w, cout, &g3, &sps);
    dim3 grid(vocr_cdiv(cin, 64), vocr_cdiv(cout, 64), splits);
    conv3x3_wgrad_wino3_kernel<<<grid, 768, 0, s>>>(x, dy, (float*)workspace, n, cin, h, w, cout, g3, sps);
    VOCR_CHECK_LAUNCH("vocr_conv3x3_wgrad_wino");
    vocr_internal_wgrad_reduce((const float*)workspace, dw, cout, cin, splits, s);
    VOCR_CHECK_LAUNCH("vocr_conv3x3_wgrad_wino(reduce)");
    return VOCR_OK;
}
