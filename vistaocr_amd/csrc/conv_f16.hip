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
