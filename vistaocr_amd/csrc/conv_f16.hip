// 3x3 "same" convolution with fp16 operands on the matrix cores and fp32 accumulation (BASELINE.json config 5:
// "fp16 conv MFMA with fp32 CTC accumulate").  Tensors stay fp32 in HBM; activations are rounded to fp16 while they are
// staged into LDS, weights are pre-packed to fp16 once per forward.  v_mfma_f32_32x32x16_f16 runs at 16x the f32 MFMA
// rate, so this kernel is bound by staging (global -> LDS), not by the MFMA pipe.
//
// Implicit GEMM: M = output channels, N = pixels (32-pixel row segments), K = (tap, 16 input channels).
// One MFMA k-step = one tap x 16 channels: lane-half h takes channels 8h..8h+7, so
//   A fragment = WtH[tap][h][co][8 halfs]     (16 B, consecutive lanes = consecutive co   -> conflict-free ds_read_b128)
//   B fragment = PH[seg][kh][col + kw][h][8]  (16 B, consecutive lanes = consecutive cols -> conflict-free)
// The halo patch is written by lanes = columns: a lane gathers its column's 8 channels (8 coalesced loads across
// the wave), converts and stores one 16-B piece.  forward and data-gradient share the kernel (different weight pack).
#include "vocr_common.h"

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));

constexpr int SEGW = 32;
constexpr int PROW = SEGW + 2;     // 34 columns incl. halo
constexpr int CI_C = 16;           // input channels per K-chunk (= one MFMA k-step per tap)

__device__ __attribute__((aligned(16))) float g_zero_page16[64];

// wpack16[chunk][tap][h][co][8]: fp16(w[co][16*chunk + 8h + j][kh][kw]) for the forward pack,
// and with roles swapped / taps flipped for the data-gradient pack:  dgrad[chunk over co][tap'][h][ci][8] = w[co=16c+8h+j][ci][2-kh][2-kw]
__global__ void pack_weights_f16_kernel(const float* __restrict__ w, _Float16* __restrict__ pf, _Float16* __restrict__ pd,
                                        int cout, int cin) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int cin_p = (cin + 15) / 16 * 16, cout_p = (cout + 15) / 16 * 16;
    const long nf = 9l * cin_p * cout;     // forward pack elements
    const long nd = 9l * cout_p * cin;     // dgrad pack elements
    if (i < nf && pf) {
        const int j = i & 7;
        long r = i >> 3;
        const int co = r % cout; r /= cout;
        const int h = r & 1; r >>= 1;
        const int tap = r % 9;
        const int chunk = r / 9;
        const int ci = chunk * 16 + h * 8 + j;
        pf[i] = ci < cin ? (_Float16)w[((long)co * cin + ci) * 9 + tap] : (_Float16)0.f;
    }
    if (i < nd && pd) {
        const int j = i & 7;
        long r = i >> 3;
        const int ci = r % cin; r /= cin;
        const int h = r & 1; r >>= 1;
        const int tap = r % 9;
        const int chunk = r / 9;
        const int co = chunk * 16 + h * 8 + j;
        pd[i] = co < cout ? (_Float16)w[((long)co * cin + ci) * 9 + (8 - tap)] : (_Float16)0.f;
    }
}

struct SegInfo16 { long base; int h; int w0; int valid; };

constexpr int CO_T = 128;
constexpr int NSEG = 4;

__global__ __launch_bounds__(256) void conv3x3_f16_kernel(const float* __restrict__ in, const _Float16* __restrict__ wpack,
                                                          const float* __restrict__ bias, float* __restrict__ out,
                                                          const float* __restrict__ zero_page, int N, int Cin, int H, int W,
                                                          int Cout, int SW, int nseg_total) {
    __shared__ __attribute__((aligned(16))) _Float16 WtH[9 * 2 * CO_T * 8];          // 36,864 B
    __shared__ __attribute__((aligned(16))) _Float16 PH[NSEG * 3 * PROW * 2 * 8];     // 13,056 B
    __shared__ SegInfo16 segs[NSEG];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lk = lane >> 5;
    const int co0 = blockIdx.y * CO_T;
    const int seg0 = blockIdx.x * NSEG;
    const long HW = (long)H * W;
    if (tid < NSEG) {
        const int g = seg0 + tid;
        SegInfo16 s;
        s.valid = g < nseg_total;
        const int gg = s.valid ? g : 0;
        const int n = gg / (H * SW), rem = gg % (H * SW);
        s.h = rem / SW;
        s.w0 = (rem % SW) * SEGW;
        s.base = (long)n * Cin * HW;
        segs[tid] = s;
    }
    __syncthreads();
    const int wco = (wave >> 1) * 64, wsg = (wave & 1) * 2;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // staging state: this wave stages the halo of segment `wave`; lane = column
    const SegInfo16 sg = segs[wave];
    const int ww = sg.w0 - 1 + lane;
    const bool colok = sg.valid && lane < PROW && ww >= 0 && ww < W;
    const int loff = colok ? ww : 0;
    const float* pbase = in + sg.base;
    f32x4 rw[9];            // 9 x 16 B of packed fp16 weights per thread
    float rp[48];           // 3 rows x 16 channels of this lane's column

    auto load_chunk = [&](int chunk) {
#pragma unroll
        for (int e = 0; e < 9; ++e) {
            const int p = tid + 256 * e;                    // piece index over [tap*2+h][co]
            const int th = p / CO_T, co = p % CO_T;
            const _Float16* src = wpack + (((long)chunk * 18 + th) * Cout + co0 + co) * 8;
            rw[e] = *(const f32x4*)((co0 + co < Cout) ? (const void*)src : (const void*)zero_page);
        }
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            const int hh = sg.h + kh - 1;
            const bool rowok = hh >= 0 && hh < H;
            const int hhc = min(max(hh, 0), H - 1);
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                const int ci = chunk * 16 + c;
                const float* src = pbase + ((long)min(ci, Cin - 1) * HW + (long)hhc * W + loff);
                rp[kh * 16 + c] = *((rowok && colok && ci < Cin) ? src : zero_page + lane);
            }
        }
    };
    auto store_chunk = [&]() {
#pragma unroll
        for (int e = 0; e < 9; ++e) *(f32x4*)(WtH + (long)(tid + 256 * e) * 8) = rw[e];
        if (lane < PROW) {
#pragma unroll
            for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    half8 v;
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = (_Float16)rp[kh * 16 + h * 8 + j];
                    *(half8*)(PH + (((wave * 3 + kh) * PROW + lane) * 2 + h) * 8) = v;
                }
        }
    };

    const int nchunks = (Cin + CI_C - 1) / CI_C;
    load_chunk(0);
    store_chunk();
    __syncthreads();
    for (int c = 0; c < nchunks; ++c) {
        if (c + 1 < nchunks) load_chunk(c + 1);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int kh = tap / 3, kw = tap % 3;
            const half8 a0 = *(const half8*)(WtH + ((tap * 2 + lk) * CO_T + wco + li) * 8);
            const half8 a1 = *(const half8*)(WtH + ((tap * 2 + lk) * CO_T + wco + 32 + li) * 8);
            const half8 b0 = *(const half8*)(PH + ((((wsg + 0) * 3 + kh) * PROW + li + kw) * 2 + lk) * 8);
            const half8 b1 = *(const half8*)(PH + ((((wsg + 1) * 3 + kh) * PROW + li + kw) * 2 + lk) * 8);
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b1, acc[1][1], 0, 0, 0);
        }
        __syncthreads();
        if (c + 1 < nchunks) {
            store_chunk();
            __syncthreads();
        }
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const SegInfo16 s = segs[wsg + j];
        const int wx = s.w0 + li;
        if (!s.valid || wx >= W) continue;
        const long obase = (s.base / Cin) * Cout + (long)s.h * W + wx;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co0 + wco + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
                if (co < Cout) out[obase + (long)co * HW] = acc[i][j][r] + (bias ? bias[co] : 0.f);
            }
    }
}

const float* zero_page16_ptr() {
    static const float* zp[64] = {nullptr};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    if (!zp[dev]) {
        void* p = nullptr;
        if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_zero_page16)) != hipSuccess) return nullptr;
        zp[dev] = (const float*)p;
    }
    return zp[dev];
}

}  // namespace

extern "C" size_t vocr_conv3x3_f16_pack_bytes(int cout, int cin, int dgrad) {
    if (cout <= 0 || cin <= 0) return 0;
    const long cin_p = (cin + 15) / 16 * 16, cout_p = (cout + 15) / 16 * 16;
    return (size_t)(dgrad ? 9l * cout_p * cin : 9l * cin_p * cout) * sizeof(_Float16);
}

extern "C" int vocr_conv3x3_f16_pack_weights(const float* w, void* wpack_fwd, void* wpack_dgrad, int cout, int cin, void* stream) {
    VOCR_CHECK_ARG(w && (wpack_fwd || wpack_dgrad) && cout > 0 && cin > 0, "vocr_conv3x3_f16_pack_weights: bad argument");
    const long cin_p = (cin + 15) / 16 * 16, cout_p = (cout + 15) / 16 * 16;
    const long n = 9l * (cin_p * cout > cout_p * cin ? cin_p * cout : cout_p * cin);
    pack_weights_f16_kernel<<<vocr_cdiv(n, 256), 256, 0, (hipStream_t)stream>>>(w, (_Float16*)wpack_fwd, (_Float16*)wpack_dgrad, cout, cin);
    VOCR_CHECK_LAUNCH("vocr_conv3x3_f16_pack_weights");
    return VOCR_OK;
}

extern "C" int vocr_conv3x3_f16_fwd(const float* x, const void* wpack, const float* bias, float* y, int n, int cin, int h, int w,
                                    int cout, void* stream) {
    VOCR_CHECK_ARG(x && wpack && y, "vocr_conv3x3_f16_fwd: null pointer");
    VOCR_CHECK_ARG(n > 0 && cin > 0 && h > 0 && w > 0 && cout > 0, "vocr_conv3x3_f16_fwd: bad shape");
    VOCR_CHECK_ARG((((uintptr_t)wpack) & 15) == 0, "vocr_conv3x3_f16_fwd: weight pack must be 16-byte aligned");
    const int SW = vocr_cdiv(w, SEGW);
    const long nseg = (long)n * h * SW;
    VOCR_CHECK_ARG(nseg < (1l << 30), "vocr_conv3x3_f16_fwd: too many segments");
    const float* zp = zero_page16_ptr();
    VOCR_CHECK_ARG(zp != nullptr, "vocr_conv3x3_f16_fwd: no device zero page");
    dim3 grid(vocr_cdiv(nseg, NSEG), vocr_cdiv(cout, CO_T));
    conv3x3_f16_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(x, (const _Float16*)wpack, bias, y, zp, n, cin, h, w, cout, SW, (int)nseg);
    VOCR_CHECK_LAUNCH("vocr_conv3x3_f16_fwd");
    return VOCR_OK;
}

// =====================================================================================================================
// Round 5: the fp16-operand convolution on channel-blocked NHWC fp16 activations ("h16").  The kernel above gathers fp32 NCHW columns through
// registers (48 dword loads and 24 conversions per lane and 16-channel chunk) and is staging-bound at 250-330 TFLOP/s - slower than
// the fp32 minimal-filtering kernels it was meant to beat.  Here the activation is converted ONCE per use into a channel-blocked fp16
// copy [N][C/16][H][W][16] (vocr_f32_to_f16_layouts: 4 B read + 2 B written per element, HBM-bound), where the 8 channels a lane half of
// v_mfma_f32_32x32x16_f16 needs for one pixel are 16 contiguous bytes: every operand then reaches LDS by DMA
// (buffer_load_dwordx4 ... lds) with a per-lane source address computed once per launch, the channel chunk as the instruction's
// scalar offset and out-of-image pieces as out-of-range offsets that deliver zeros - no register staging, no conversion, no VALU
// in the loop beside the MFMAs and their fragment reads.
//   workgroup = 4 waves; wave tile = 64 output channels x 4 segments of 32 pixels (8 accumulator tiles, 128 registers);
//   WCO = 2: 128 channels x 8 segments per workgroup, WCO = 1 (64-channel layers): 64 x 16.
//   LDS stage (one 16-channel chunk = one MFMA k-step per tap), two stages:
//     weights [tap][half][co][8 halfs]           (the existing fp16 pack: A fragment = one conflict-free ds_read_b128)
//     input   [patch row][half][36 pixels][8]    (B fragment of segment j, tap (kh, kw) = row j + kh, piece li + kw: 16 consecutive
//                                                 lanes = 256 contiguous bytes; the segments of a workgroup are a strip of rows and share halo rows)
//   Source layout: with plain [N][H][W][C] a DMA instruction's 64 pieces touched 64 different 128-byte lines and used 16 bytes of each
//   (fill-bound at 600 TFLOP/s on 256 -> 256, 236 on 64 -> 64); channel-blocked, a patch row of a chunk is one contiguous run: 645 / 341.
//   one barrier per chunk (72 MFMAs of 32 cycles per wave between barriers); the DMAs of chunk c + 1 fly under the MFMAs of chunk c.
// Output: fp32 NCHW (+ bias), a row of 32 pixels per channel = one 128-byte store, as before.  Forward and data gradient share the
// kernel (the data-gradient pack has the taps flipped and the channel roles swapped).

namespace {

#define LDS16(p) ((__attribute__((address_space(3))) void*)(p))

// (a __device__ template behind plain kernels: a device builtin inside a TEMPLATE __global__ makes the host pass drop the launch stub)
// WCO = waves along the output channels (2: 128 channels per workgroup, 1: 64), SPW = segments per wave (2 or 3).  Eight waves = two per
// SIMD: one wave's fragment reads and DMA issues are covered by the other's MFMAs.
template <int WCO, int SPW, int NCB>
__device__ __forceinline__ void conv3x3_h16_body(const _Float16* __restrict__ x16, const _Float16* __restrict__ wpack,
                                                 const float* __restrict__ bias, float* __restrict__ out, int N, int Cin, int H,
                                                 int W, int Cout, int SW, int nseg_total, unsigned x_bytes, unsigned w_bytes) {
    constexpr int CO_T = 64 * WCO;
    constexpr int WSG = 8 / WCO;                      // waves along the segments
    constexpr int NSEG = SPW * WSG;
    constexpr int NR = NSEG / NCB;                    // rows of the tile; NCB = its 32-pixel column blocks (segment j = row j % NR, block j / NR)
    static_assert(NR * NCB == NSEG, "tile = rows x column blocks");
    constexpr int WP = 9 * 2 * CO_T;                  // 16-byte pieces of a stage's weights
    constexpr int PP = 32 * NCB + 4;                  // pieces per patch row: 32 NCB + 2 pixels + 2 never-read ones
    constexpr int IP = (NR + 2) * 2 * PP;             // ... and of its input patch: the tile's rows + one above and below
    constexpr int IPA = (IP + 63) / 64 * 64;          // ... rounded up to whole waves of pieces (a DMA instruction writes 64)
    constexpr int SP = WP + IPA;
    constexpr int NQ = (WP + 511) / 512, NJ = (IPA + 511) / 512;
    constexpr unsigned OOB = 0x80000000u;
    static_assert(WP % 64 == 0, "whole waves of DMA pieces");
    static_assert(2 * SP * 16 <= 160 * 1024, "two stages in LDS");
    __shared__ __attribute__((aligned(16))) _Float16 lds[2 * SP * 8];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lk = lane >> 5;
    const int co0 = blockIdx.y * CO_T;
    const long HW = (long)H * W;
    // The workgroup's NSEG segments are a STRIP: rows h0 .. h0 + NSEG - 1 of one image at one 32-pixel column block, so that they share
    // their halo rows - the patch is NSEG + 2 rows instead of 3 NSEG (raster-order segments fetched every input row three times: 27.6 of
    // the 64.6 KB a 128 x 8 workgroup fills per 16-channel chunk, and the fill is what bounds the kernel)
    const int HB = (H + NR - 1) / NR;                 // tile rows per image; SW = tile columns (N * HB * SW tiles)
    const int tile = blockIdx.x;
    const int img = tile / (HB * SW), trem = tile % (HB * SW);
    const int h0 = (trem / SW) * NR, w0 = (trem % SW) * SEGW * NCB;

    // ---- DMA source offsets, once per launch.  Piece P = tid + 512 j of a stage lands at LDS byte 16 P (the DMA writes a wave's 64
    // pieces contiguously); a chunk advances every source by a scalar offset
    unsigned wo[NQ], xo[NJ];
#pragma unroll
    for (int j = 0; j < NQ; ++j) {
        const int p = tid + 512 * j;
        const int th = p / CO_T, co = p % CO_T;       // th = tap*2 + half
        wo[j] = (p < WP && co0 + co < Cout) ? (unsigned)(((long)th * Cout + co0 + co) * 16) : OOB;
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int p = tid + 512 * j;
        unsigned o = OOB;
        if (p < IP) {
            const int prow = p / (2 * PP), r2 = p % (2 * PP);
            const int half = r2 / PP, px = r2 % PP;
            const int hh = h0 - 1 + prow, ww = w0 - 1 + px;
            if (px < 32 * NCB + 2 && hh >= 0 && hh < H && ww >= 0 && ww < W)
                o = (unsigned)(((((long)img * (Cin / CI_C)) * H + hh) * W + ww) * 32 + half * 16);        // [N][C/16][H][W][16]: chunk 0
        }
        xo[j] = o;
    }
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)x16, 0, x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)wpack, 0, w_bytes, 0x00020000);
    const int wchunk_bytes = 9 * 2 * Cout * 16;
    const int xchunk_bytes = H * W * 32;              // one 16-channel block of one image

    // DMA d (0 .. NQ + NJ - 1) of a stage's fill; issued one per tap between the MFMAs of the chunk before
    auto issue_one = [&](int d, int chunk, int stage) {
        _Float16* base = lds + (long)stage * SP * 8;
        if (d < NQ) {
            if (512 * d + 64 * wave < WP)             // wave-uniform: whole 64-piece groups only
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, LDS16(base + (long)(512 * d + 64 * wave) * 8), 16, wo[d < NQ ? d : 0], chunk * wchunk_bytes, 0, 0);
        } else {
            const int j = d - NQ;
            if (512 * j + 64 * wave < IPA)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, LDS16(base + (long)(WP + 512 * j + 64 * wave) * 8), 16, xo[j < NJ ? j : 0], chunk * xchunk_bytes, 0, 0);
        }
    };
    constexpr int ND = NQ + NJ;
#ifndef H16_SPREAD
#define H16_SPREAD 5
#endif

    const int wco = (wave / WSG) * 64, wsg = (wave % WSG) * SPW;
    f32x16 acc[2][SPW];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < SPW; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nchunks = Cin / CI_C;
#pragma unroll
    for (int d = 0; d < ND; ++d) issue_one(d, 0, 0);
    __builtin_amdgcn_s_waitcnt(0x0F70);               // vmcnt(0)
    __syncthreads();
    for (int c = 0; c < nchunks; ++c) {
        const bool more = c + 1 < nchunks;
        const _Float16* wl = lds + (long)(c & 1) * SP * 8;
        const _Float16* il = wl + (long)WP * 8;
        // fragments of tap t + 1 are read while the MFMAs of tap t run (two register sets): with the reads in front of their own MFMAs a
        // tap was LDS latency + 4-6 MFMAs per wave and the two waves of a SIMD could not cover each other (30 % MFMA-busy)
        half8 fa[2][2], fb[2][SPW];
        auto load_frags = [&](int tap, int set) {
            const int kh = tap / 3, kw = tap % 3;
#pragma unroll
            for (int i = 0; i < 2; ++i) fa[set][i] = *(const half8*)(wl + ((tap * 2 + lk) * CO_T + wco + 32 * i + li) * 8);
#pragma unroll
            for (int j = 0; j < SPW; ++j)
                fb[set][j] = *(const half8*)(il + ((((wsg + j) % NR + kh) * 2 + lk) * PP + 32 * ((wsg + j) / NR) + li + kw) * 8);
        };
        load_frags(0, 0);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            if (tap + 1 < 9) load_frags(tap + 1, (tap + 1) & 1);
            if (more && tap < H16_SPREAD) {           // the next stage's fill: H16_SPREAD taps share its DMAs
                constexpr int PER = (ND + H16_SPREAD - 1) / H16_SPREAD;
#pragma unroll
                for (int d = 0; d < PER; ++d)
                    if (tap * PER + d < ND) issue_one(tap * PER + d, c + 1, (c + 1) & 1);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < SPW; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[tap & 1][i], fb[tap & 1][j], acc[i][j], 0, 0, 0);
        }
        __builtin_amdgcn_s_waitcnt(0x0F70);           // this wave's DMAs of chunk c + 1 have landed ...
        __syncthreads();                              // ... everybody's have, and everybody is done reading stage c & 1
    }
#pragma unroll
    for (int j = 0; j < SPW; ++j) {
        const int hrow = h0 + (wsg + j) % NR, wx = w0 + 32 * ((wsg + j) / NR) + li;
        if (hrow >= H || wx >= W) continue;
        const long obase = (long)img * Cout * HW + (long)hrow * W + wx;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co0 + wco + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
                if (co < Cout) out[obase + (long)co * HW] = acc[i][j][r] + (bias ? bias[co] : 0.f);
            }
    }
}

#define VOCR_H16_KERNEL(NAME, WCO, SPW, NCB)                                                                                              \
    __global__ __launch_bounds__(512) void NAME(const _Float16* __restrict__ x16, const _Float16* __restrict__ wpack,                     \
                                                const float* __restrict__ bias, float* __restrict__ out, int N, int Cin, int H, int W,    \
                                                int Cout, int SW, int nseg_total, unsigned x_bytes, unsigned w_bytes) {                   \
        conv3x3_h16_body<WCO, SPW, NCB>(x16, wpack, bias, out, N, Cin, H, W, Cout, SW, nseg_total, x_bytes, w_bytes);                      \
    }
VOCR_H16_KERNEL(conv3x3_h16_kernel_128x2, 2, 2, 1)    // 128 channels x  8 rows x 32 pixels
VOCR_H16_KERNEL(conv3x3_h16_kernel_128x3, 2, 3, 1)    // 128 channels x 12 rows x 32 pixels
VOCR_H16_KERNEL(conv3x3_h16_kernel_128x4, 2, 4, 1)    // 128 channels x 16 rows x 32 pixels
VOCR_H16_KERNEL(conv3x3_h16_kernel_128x4c2, 2, 4, 2)  // 128 channels x  8 rows x 64 pixels
VOCR_H16_KERNEL(conv3x3_h16_kernel_64x2, 1, 2, 1)     //  64 channels x 16 rows x 32 pixels
#undef VOCR_H16_KERNEL

}  // namespace

extern "C" int vocr_conv3x3_h16_supported(int cin, int cout) { return cin > 0 && cout > 0 && cin % 16 == 0 ? 1 : 0; }

// Which forward / data-gradient kernel a launch takes: 1 = 64x2 (cout <= 64: 16 rows x 32 pixels), 2 = 128x2 (8 rows x 32), 3 = 128x3
// (12 x 32), 4 = 128x4 (16 x 32), 5 = 128x4c2 (8 rows x 64).  Above 64 output channels: the tile shape that needs the least time in
// whole rounds of one workgroup per CU; cost of a tile = its segments, a 16-segment tile at 0.8 of two 8-segment ones (it fills 58
// instead of 2 x 48 KB per chunk, and the fill is what bounds the kernel); VOCR_H16_TILE (experiments build): force a shape
static int h16_fwd_pick(int n, int h, int w, int cout) {
    if (cout <= 64) return 1;
    int ncu = 256;
    {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ncu = prop.multiProcessorCount;
    }
    auto tiles_of = [&](int rows, int blocks) { return (long)n * vocr_cdiv(h, rows) * vocr_cdiv(w, SEGW * blocks); };
    const int ct = vocr_cdiv(cout, 128);
    static const int force = VOCR_EXPERIMENT_INT("VOCR_H16_TILE", 0);
    const double c2 = (double)vocr_cdiv(tiles_of(8, 1) * ct, ncu) * 2, c3 = (double)vocr_cdiv(tiles_of(12, 1) * ct, ncu) * 3,
                 c4 = (double)vocr_cdiv(tiles_of(16, 1) * ct, ncu) * 4 * 0.8, c42 = (double)vocr_cdiv(tiles_of(8, 2) * ct, ncu) * 4 * 0.8;
    int pick = 2;
    double best = c2;
    if (c3 < best) { best = c3; pick = 3; }
    if (c4 < best) { best = c4; pick = 4; }
    if (c42 < best) { best = c42; pick = 5; }
    if (force) pick = force;
    return pick;
}

extern "C" int vocr_conv3x3_h16_plan(int n, int cin, int h, int w, int cout) {
    if (n <= 0 || h <= 0 || w <= 0 || !vocr_conv3x3_h16_supported(cin, cout)) return 0;
    return h16_fwd_pick(n, h, w, cout);
}

extern "C" int vocr_conv3x3_h16_fwd(const void* x16, const void* wpack, const float* bias, float* y, int n, int cin, int h, int w,
                                    int cout, void* stream) {
    VOCR_CHECK_ARG(x16 && wpack && y, "vocr_conv3x3_h16_fwd: null pointer");
    VOCR_CHECK_ARG(n > 0 && cin > 0 && h > 0 && w > 0 && cout > 0 && cin % 16 == 0, "vocr_conv3x3_h16_fwd: bad shape (Cin %% 16 == 0)");
    VOCR_CHECK_ARG(((((uintptr_t)wpack) | ((uintptr_t)x16)) & 15) == 0, "vocr_conv3x3_h16_fwd: 16-byte alignment");
    const long xb = (long)n * h * w * cin * 2, wb = 9l * cin * cout * 2;
    VOCR_CHECK_ARG(xb < (1l << 31) && wb < (1l << 31), "vocr_conv3x3_h16_fwd: tensor too large for 32-bit buffer offsets");
    hipStream_t s = (hipStream_t)stream;
    const _Float16* xp = (const _Float16*)x16;
    const _Float16* wp = (const _Float16*)wpack;
    // a workgroup's tile = rows x column blocks of 32 pixels; tiles = n * ceil(h / rows) * ceil(w / (32 blocks))
    auto tiles_of = [&](int rows, int blocks) { return (long)n * vocr_cdiv(h, rows) * vocr_cdiv(w, SEGW * blocks); };
    VOCR_CHECK_ARG(tiles_of(8, 1) < (1l << 30), "vocr_conv3x3_h16_fwd: too many tiles");
#define VOCR_H16_LAUNCH(K, ROWS, BLOCKS, CT)                                                                                            \
    K<<<dim3((unsigned)tiles_of(ROWS, BLOCKS), CT), 512, 0, s>>>(xp, wp, bias, y, n, cin, h, w, cout, vocr_cdiv(w, SEGW * BLOCKS),          \
                                                                (int)tiles_of(ROWS, BLOCKS), (unsigned)xb, (unsigned)wb)
    const int ct = vocr_cdiv(cout, 128);
    switch (h16_fwd_pick(n, h, w, cout)) {
        case 1: VOCR_H16_LAUNCH(conv3x3_h16_kernel_64x2, 16, 1, 1); break;
        case 3: VOCR_H16_LAUNCH(conv3x3_h16_kernel_128x3, 12, 1, ct); break;
        case 4: VOCR_H16_LAUNCH(conv3x3_h16_kernel_128x4, 16, 1, ct); break;
        case 5: VOCR_H16_LAUNCH(conv3x3_h16_kernel_128x4c2, 8, 2, ct); break;
        default: VOCR_H16_LAUNCH(conv3x3_h16_kernel_128x2, 8, 1, ct); break;
    }
#undef VOCR_H16_LAUNCH
    VOCR_CHECK_LAUNCH("vocr_conv3x3_h16_fwd");
    return VOCR_OK;
}

// =====================================================================================================================
// Round 5: the fp16-operand WEIGHT GRADIENT.  dW[co][ci][kh][kw] = sum over pixels of dy[co][p] x[ci][p + (kh-1, kw-1)]: M = co,
// N = ci, K = pixels, one accumulator tile per tap (9 x 16 registers per 32 x 32 block of (co, ci)).  v_mfma_f32_32x32x16_f16 wants 8
// CONSECUTIVE k per lane half, i.e. 8 consecutive pixels of one channel: both operands are read from channel-major fp16 copies
// [N][C][H][WP] whose rows are padded to WP = ceil8(W) + 8 with zeros (vocr_f32_to_f16_layouts writes them in the same pass as the NHWC
// copy).  The zero tail makes every 16-byte piece that sticks out of a row read zeros from memory (the left halo of a row start is the
// previous row's tail), so all operands reach LDS by DMA with nothing to patch; rows above / below the image are out-of-range offsets.
//   The kw = 1 fragment of x is an aligned 16-byte read; kw = 0 / 2 start one pixel (2 bytes) earlier / later: the wave reads the two
//   neighbouring dwords as well and funnel-shifts (5 v_alignbit_b32 per row of taps - free beside 32-cycle MFMAs).
//   workgroup = 8 waves = (co tiles of 32) x (ci tiles of 32) x (k subsets) over a tile of min(Cout, 128) x min(Cin, 64); a work UNIT is
//   64 pixels of one image row (<= 4 k-steps of 16 pixels); the units are cut into as many contiguous ranges ("splits") as there are
//   CUs per tile; every wave writes its 9 accumulator tiles to a slab and wgrad_h16_reduce_kernel adds the slabs in a fixed order
//   (bitwise reproducible, no atomics).
namespace {

typedef unsigned u32x4h __attribute__((ext_vector_type(4)));

struct WgH16Args {
    const _Float16* dy;            // [N][Co][H][WP]
    const _Float16* x;             // [N][Ci][H][WP]
    float* slab;                   // [tile][split][ksub][tap][CTco][CTci]
    int N, Ci, Co, H, W, WP;
    int spr;                       // 64-pixel slabs per row
    int units;                     // N * H * spr
    int splits;
    unsigned dy_bytes, x_bytes;
};

template <int CW, int IW>
__device__ __forceinline__ void wgrad_h16_body(const WgH16Args g) {
    constexpr int KS = 8 / (CW * IW);
    constexpr int CTco = 32 * CW, CTci = 32 * IW;
    constexpr int DYP = CTco * 9;                     // dy pieces of a stage: [co][8 data + 1 pad]
    constexpr int XP = CTci * 33;                     // x pieces: [ci][kh][10 data + 1 pad]
    constexpr int DYPA = (DYP + 63) / 64 * 64, XPA = (XP + 63) / 64 * 64;
    constexpr int SP = DYPA + XPA;
    constexpr int NDY = (DYPA + 511) / 512, NX = (XPA + 511) / 512;
    constexpr unsigned OOB = 0x80000000u;
    __shared__ __attribute__((aligned(16))) _Float16 lds[2 * SP * 8];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lk = lane >> 5;
    const int ks = wave % KS, iw = (wave / KS) % IW, cw = wave / (KS * IW);
    const int tiles_ci = g.Ci / CTci;
    const int tile = blockIdx.x / g.splits, split = blockIdx.x % g.splits;
    const int co0 = (tile / tiles_ci) * CTco, ci0 = (tile % tiles_ci) * CTci;
    const int u0 = (int)((long)g.units * split / g.splits), u1 = (int)((long)g.units * (split + 1) / g.splits);
    const long rowb = (long)g.WP * 2;                 // bytes per row

    // per-lane constant parts of the DMA sources (unit-independent): dy piece -> (co row, piece q), x piece -> (ci, kh, piece q)
    unsigned dyo[NDY], xo[NX];
    int xkh[NX], xq[NX];
#pragma unroll
    for (int j = 0; j < NDY; ++j) {
        const int p = tid + 512 * j;
        const int row = p / 9, q = p % 9;
        dyo[j] = (p < DYP && q < 8) ? (unsigned)(((long)(co0 + row) * g.H) * rowb + q * 16) : OOB;
    }
#pragma unroll
    for (int j = 0; j < NX; ++j) {
        const int p = tid + 512 * j;
        const int ci = p / 33, r = p % 33;
        xkh[j] = r / 11;
        xq[j] = r % 11;
        // relative to a base one row and one piece IN FRONT of the tensor (the buffer's range check looks at this per-lane part alone, so
        // it has to be non-negative; the unit's part below is the instruction's scalar offset)
        xo[j] = (p < XP && xq[j] < 10) ? (unsigned)(((long)(ci0 + ci) * g.H + xkh[j]) * rowb + xq[j] * 16) : OOB;
    }
    const unsigned xpad = (unsigned)(rowb + 16);
    const __amdgpu_buffer_rsrc_t rdy = __builtin_amdgcn_make_buffer_rsrc((void*)g.dy, 0, g.dy_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)g.x - xpad), 0, g.x_bytes + xpad, 0x00020000);

    auto issue = [&](int u, int stage) {
        const int r = u / g.spr, sx = u % g.spr;      // image row index n*H + h, slab in the row
        const int n = r / g.H, h = r % g.H;
        _Float16* base = lds + (long)stage * SP * 8;
        // scalar parts: dy (n*Co*H + h) rows + 64 sx pixels; x (n*Ci*H + h) rows + 64 sx pixels
        const unsigned sdy = (unsigned)(((long)n * g.Co * g.H + h) * rowb + sx * 128);
        const unsigned sxx = (unsigned)(((long)n * g.Ci * g.H + h) * rowb + sx * 128);
#pragma unroll
        for (int j = 0; j < NDY; ++j)
            if (512 * j + 64 * wave < DYPA)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rdy, LDS16(base + (long)(512 * j + 64 * wave) * 8), 16, dyo[j], sdy, 0, 0);
#pragma unroll
        for (int j = 0; j < NX; ++j)
            if (512 * j + 64 * wave < XPA) {
                // rows above / below the image and pieces that start at or behind the row's padded end deliver zeros
                // (and the left halo of the tensor's very first row, which would lie in front of the allocation)
                const bool dead = (xkh[j] == 0 && h == 0) || (xkh[j] == 2 && h == g.H - 1) || (64 * sx + 8 * (xq[j] - 1) >= g.WP) ||
                                  (xo[j] != OOB && xo[j] + sxx < xpad);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, LDS16(base + (long)(DYPA + 512 * j + 64 * wave) * 8), 16, dead ? OOB : xo[j], sxx, 0, 0);
            }
    };

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    if (u0 < u1) issue(u0, 0);
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __syncthreads();
    for (int u = u0; u < u1; ++u) {
        const int st = (u - u0) & 1;
        if (u + 1 < u1) issue(u + 1, st ^ 1);
        const _Float16* dl = lds + (long)st * SP * 8;
        const _Float16* xl = dl + (long)DYPA * 8;
        const int sx = u % g.spr;
        const int nk = min(4, (g.W - 64 * sx + 15) >> 4);          // 16-pixel k-steps with a real pixel in this slab
        for (int s = ks; s < nk; s += KS) {
            const half8 a = *(const half8*)(dl + ((cw * 32 + li) * 9 + 2 * s + lk) * 8);
#pragma unroll
            for (int kh = 0; kh < 3; ++kh) {
                const unsigned* row = (const unsigned*)(xl + (((iw * 32 + li) * 3 + kh) * 11) * 8);
                const int q = 1 + 2 * s + lk;             // the aligned piece: pixels 64 sx + 16 s + 8 lk .. + 7
                const u32x4h d = *(const u32x4h*)(row + 4 * q);
                const unsigned d0 = row[4 * q - 1], d5 = row[4 * q + 4];
                const unsigned a01 = __builtin_amdgcn_alignbit(d[0], d0, 16), a12 = __builtin_amdgcn_alignbit(d[1], d[0], 16),
                               a23 = __builtin_amdgcn_alignbit(d[2], d[1], 16), a34 = __builtin_amdgcn_alignbit(d[3], d[2], 16),
                               a45 = __builtin_amdgcn_alignbit(d5, d[3], 16);
                const u32x4h b0 = {a01, a12, a23, a34}, b2 = {a12, a23, a34, a45};
                acc[kh * 3 + 0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, __builtin_bit_cast(half8, b0), acc[kh * 3 + 0], 0, 0, 0);
                acc[kh * 3 + 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, __builtin_bit_cast(half8, d), acc[kh * 3 + 1], 0, 0, 0);
                acc[kh * 3 + 2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, __builtin_bit_cast(half8, b2), acc[kh * 3 + 2], 0, 0, 0);
            }
        }
        __builtin_amdgcn_s_waitcnt(0x0F70);
        __syncthreads();
    }
    // slab[tile][split][ks][tap][co in tile][ci in tile]
    float* sl = g.slab + ((((long)tile * g.splits + split) * KS + ks) * 9) * CTco * CTci;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = cw * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
            sl[((long)t * CTco + co) * CTci + iw * 32 + li] = acc[t][r];
        }
}


__global__ __launch_bounds__(512) void conv3x3_wgrad_h16_kernel_128x64(const WgH16Args g) { wgrad_h16_body<4, 2>(g); }
__global__ __launch_bounds__(512) void conv3x3_wgrad_h16_kernel_64x64(const WgH16Args g) { wgrad_h16_body<2, 2>(g); }

// dw[co][ci][tap] = sum over (split, ksub) of slab[tile][split][ksub][tap][co'][ci'], in a fixed order: a workgroup owns 64 consecutive
// (tap, co, ci) outputs (ci fastest: one part's 64 values are 256 contiguous bytes); wave w adds the parts w, w + 4, w + 8, ... with eight
// loads in flight, then the four waves' sums are added in wave order.  (One thread per output walking all <= 512 parts took 350 us on
// the 64 -> 64 layer, on the side stream's critical path at the end of the step.)
__global__ __launch_bounds__(256) void wgrad_h16_reduce_kernel(const float* __restrict__ slab, float* __restrict__ dw, int Co, int Ci, int CTco,
                                                               int CTci, int parts) {
    __shared__ float red[4][64];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long i = (long)blockIdx.x * 64 + lane;                      // (tap, co, ci), ci fastest; Ci % 64 == 0: a workgroup stays in one row
    const long total = 9l * Co * Ci;
    const bool ok = i < total;
    const long ic = ok ? i : 0;
    const int ci = (int)(ic % Ci), co = (int)((ic / Ci) % Co), tap = (int)(ic / ((long)Ci * Co));
    const int tile = (co / CTco) * (Ci / CTci) + ci / CTci;
    const float* p = slab + (((long)tile * parts) * 9 + tap) * CTco * CTci + (long)(co % CTco) * CTci + ci % CTci;
    const long stride = 9l * CTco * CTci;
    float v = 0.f;
    int k = wave;
    for (; k + 28 < parts; k += 32) {
        float t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) t[u] = __builtin_nontemporal_load(p + (long)(k + 4 * u) * stride);
#pragma unroll
        for (int u = 0; u < 8; ++u) v += t[u];
    }
    for (; k < parts; k += 4) v += __builtin_nontemporal_load(p + (long)k * stride);
    red[wave][lane] = v;
    __syncthreads();
    if (wave == 0 && ok) dw[((long)co * Ci + ci) * 9 + tap] = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
}

// fp32 [N][C][H][W] -> fp16 [N][H][W][C] (nhwc, may be NULL) and / or fp16 [N][C][H][WP] with zero-padded rows (nchwp, may be NULL;
// WP = ceil8(W) + 8) in one pass over the input: one workgroup = one image row x 64 pixels x 64 channels through LDS
__global__ __launch_bounds__(256) void f32_to_f16_layouts_kernel(const float* __restrict__ x, _Float16* __restrict__ nhwc, _Float16* __restrict__ nchwp,
                                                                 int C, int H, int W, int WP) {
    __shared__ float t[64][65];
    const int tid = threadIdx.x;
    const int w0 = blockIdx.x * 64, nh = blockIdx.y, c0 = blockIdx.z * 64;
    const int n = nh / H, h = nh % H;
    const int px = tid & 63, cs = tid >> 6;
    const float* src = x + (((long)n * C + c0) * H + h) * W + w0;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int c = cs + 4 * k;
        t[c][px] = (c0 + c < C && w0 + px < W) ? src[(long)c * H * W + px] : 0.f;
    }
    __syncthreads();
    if (nhwc) {
        // channel-blocked "NHWC": [N][C/16][H][W][16] - a pixel's 16 channels of one block are 32 contiguous bytes and consecutive
        // pixels follow each other, so the conv kernel's 16-byte DMA pieces use whole cache lines (plain [N][H][W][C] left 16 of every
        // 128 bytes of a line per instruction: the forward kernel ran at 600 TFLOP/s, fill-bound)
        const int cbn = min(64, C - c0) >> 4;         // 16-channel blocks in this channel tile
        for (int p = tid; p < cbn * 128; p += 256) {
            const int cb = p >> 7, q = (p >> 1) & 63, half = p & 1;
            if (w0 + q >= W) continue;
            half8 v;
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = (_Float16)t[cb * 16 + half * 8 + k][q];
            *(half8*)(nhwc + ((((long)n * (C >> 4) + (c0 >> 4) + cb) * H + h) * W + w0 + q) * 16 + half * 8) = v;
        }
    }
    if (nchwp) {
        // thread = (channel, 16-pixel group): two 16-byte pieces of the channel's padded row; pixels >= W are zeros (loaded as such)
        const int c = tid >> 2, gq = tid & 3;
        if (c0 + c < C) {
            _Float16* dst = nchwp + (((long)n * C + c0 + c) * H + h) * WP + w0 + 16 * gq;
#pragma unroll
            for (int hpc = 0; hpc < 2; ++hpc) {
                if (w0 + 16 * gq + 8 * hpc >= WP) continue;
                half8 v;
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = (_Float16)t[c][16 * gq + 8 * hpc + k];
                *(half8*)(dst + 8 * hpc) = v;
            }
        }
    }
}

}  // namespace

static inline int h16_wp(int w) { return (w + 7) / 8 * 8 + 8; }

extern "C" int vocr_f16_padded_row(int w) { return w > 0 ? h16_wp(w) : 0; }

extern "C" int vocr_f32_to_f16_layouts(const float* x, void* nhwc, void* nchwp, int n, int c, int h, int w, void* stream) {
    VOCR_CHECK_ARG(x && (nhwc || nchwp) && n > 0 && c > 0 && h > 0 && w > 0, "vocr_f32_to_f16_layouts: bad argument");
    VOCR_CHECK_ARG(c % 8 == 0 && (!nhwc || c % 16 == 0) && ((((uintptr_t)nhwc) | ((uintptr_t)nchwp)) & 15) == 0,
                   "vocr_f32_to_f16_layouts: C %% 8 == 0 (%% 16 for the channel-blocked copy) and 16-byte aligned outputs");
    VOCR_CHECK_ARG((long)n * h <= 65535 && vocr_cdiv(c, 64) <= 65535, "vocr_f32_to_f16_layouts: too many rows");
    const int wp = h16_wp(w);
    f32_to_f16_layouts_kernel<<<dim3(vocr_cdiv(nchwp ? wp : w, 64), n * h, vocr_cdiv(c, 64)), 256, 0, (hipStream_t)stream>>>(
        x, (_Float16*)nhwc, (_Float16*)nchwp, c, h, w, wp);
    VOCR_CHECK_LAUNCH("vocr_f32_to_f16_layouts");
    return VOCR_OK;
}

namespace {
struct WgH16Plan { bool ok; int ctco, ctci, tiles, splits, parts; };
WgH16Plan wgrad_h16_plan(int n, int cin, int h, int w, int cout) {
    WgH16Plan p = {false, 0, 0, 0, 0, 0};
    if (cin % 64 != 0 || cout % 64 != 0 || (cout > 64 && cout % 128 != 0)) return p;
    p.ctco = cout >= 128 ? 128 : 64;
    p.ctci = 64;
    p.tiles = (cout / p.ctco) * (cin / p.ctci);
    int ncu = 256;
    {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ncu = prop.multiProcessorCount;
    }
    const long units = (long)n * h * vocr_cdiv(w, 64);
    long s = ncu / p.tiles;
    if (s < 1) s = 1;
    if (s > units) s = units;
    p.splits = (int)s;
    p.parts = p.splits * (8 / ((p.ctco / 32) * (p.ctci / 32)));
    p.ok = true;
    return p;
}
}  // namespace

extern "C" int vocr_conv3x3_wgrad_h16_supported(int cin, int cout) { return wgrad_h16_plan(1, cin, 1, 64, cout).ok ? 1 : 0; }

extern "C" size_t vocr_conv3x3_wgrad_h16_workspace_bytes(int n, int cin, int h, int w, int cout) {
    if (n <= 0 || h <= 0 || w <= 0) return 0;
    const WgH16Plan p = wgrad_h16_plan(n, cin, h, w, cout);
    return p.ok ? (size_t)p.tiles * p.parts * 9 * p.ctco * p.ctci * sizeof(float) : 0;
}

extern "C" int vocr_conv3x3_wgrad_h16(const void* x16p, const void* dy16p, float* dw, void* workspace, int n, int cin, int h, int w, int cout,
                                      void* stream) {
    VOCR_CHECK_ARG(x16p && dy16p && dw && workspace && n > 0 && h > 0 && w > 0, "vocr_conv3x3_wgrad_h16: bad argument");
    const WgH16Plan p = wgrad_h16_plan(n, cin, h, w, cout);
    VOCR_CHECK_ARG(p.ok, "vocr_conv3x3_wgrad_h16: Cin %% 64 == 0 and Cout in {64, multiples of 128} (ask vocr_conv3x3_wgrad_h16_supported)");
    VOCR_CHECK_ARG(((((uintptr_t)x16p) | ((uintptr_t)dy16p)) & 15) == 0, "vocr_conv3x3_wgrad_h16: 16-byte alignment");
    const int wp = h16_wp(w);
    const long xb = (long)n * cin * h * wp * 2, db = (long)n * cout * h * wp * 2;
    VOCR_CHECK_ARG(xb < (1l << 31) && db < (1l << 31), "vocr_conv3x3_wgrad_h16: tensor too large for 32-bit buffer offsets");
    WgH16Args g;
    g.dy = (const _Float16*)dy16p; g.x = (const _Float16*)x16p; g.slab = (float*)workspace;
    g.N = n; g.Ci = cin; g.Co = cout; g.H = h; g.W = w; g.WP = wp;
    g.spr = vocr_cdiv(w, 64);
    g.units = n * h * g.spr;
    g.splits = p.splits;
    g.dy_bytes = (unsigned)db; g.x_bytes = (unsigned)xb;
    hipStream_t s = (hipStream_t)stream;
    if (p.ctco == 128) conv3x3_wgrad_h16_kernel_128x64<<<p.tiles * p.splits, 512, 0, s>>>(g);
    else conv3x3_wgrad_h16_kernel_64x64<<<p.tiles * p.splits, 512, 0, s>>>(g);
    VOCR_CHECK_LAUNCH("vocr_conv3x3_wgrad_h16");
    wgrad_h16_reduce_kernel<<<vocr_cdiv(9l * cout * cin, 64), 256, 0, s>>>((const float*)workspace, dw, cout, cin, p.ctco, p.ctci, p.parts);
    VOCR_CHECK_LAUNCH("vocr_conv3x3_wgrad_h16(reduce)");
    return VOCR_OK;
}
