// fp32 GEMM on the bf16 matrix pipe by exact operand splitting ("bf16x6"): every fp32 operand element a is written as a0 + a1 + a2 with
// a0 = bf16(a), a1 = bf16(a - a0), a2 = bf16(a - a0 - a1) - EXACT (3 x 8 significant bits cover fp32's 24; the residuals are computed in
// fp32 without rounding) - and a*b is accumulated from the six partial products of order <= 2^-16 (a0b0, a0b1, a1b0, a0b2, a2b0, a1b1),
// each of them exact in the fp32 accumulator's input (8 x 8 bits); the three dropped ones (a1b2, a2b1, a2b2) are <= 2^-23 |a b|, one unit
// roundoff of the fp32 product the f32 MFMA would have formed.  v_mfma_f32_32x32x16_bf16 runs at 16x the rate of v_mfma_f32_32x32x2_f32, so six
// of them per product are 2.67x the f32 matrix peak (417 TFLOP/s of fp32-equivalent work on an MI355X).
//
// Operands are kept as three bf16 planes in MFMA-FRAGMENT order (written by the split kernels below, one pass over the fp32 tensor):
//     plane[p][row tile rt][k16 step kk][lane][8]   lane (r = lane & 31, h = lane >> 5) holds X[32 rt + r][16 kk + 8 h + 0..7]
// so that a fragment is 1 KB contiguous in memory AND in LDS: it reaches LDS by one DMA instruction of a wave and comes back with one
// conflict-free ds_read_b128 per lane.  Rows are padded to 256, K to 32 (zeros).  The source may be K-contiguous ([row][k], vocr_gemm_x6_split_rk) or
// K-strided ([k][row], _split_kr: the transposed operands of the weight gradients); the product kernel is the same for every transpose form.
//
// gemm_x6_kernel<NCT, NP>: workgroup tile 256 x 32 NCT - 256 x 128 (eight waves as 4 x 2, 64 x 64 wave tiles) or 256 x 256 (2 x 4 waves, 128 x 64 wave tiles,
// 128 accumulator registers) -, stages of one k16 step of all planes (NP (8 + NCT) fragments of 1 KB, moved by LDS-DMA, NFRAG / 8 instructions per wave), a
// ring of 144 KB / stage slots: the DMAs of stage s + NS - 1 are issued one at a time between the MFMA groups of stage s, one barrier per stage.  Whole rounds
// of one workgroup per CU run the full K; what is left of the last round is cut along K into slabs that gemm_x6_reduce_kernel adds in a fixed order.  Tiles are
// dealt to the XCDs in contiguous ranges of 4 x 8-tile blocks (x6_tile_of / x6_xcd_chunked).  A launch may carry a second set of views for the rows past a
// cut (two products of one shape: the recurrent weight gradients of both directions).  The kernel is bound by the board's power limit.
// NP = 3 is bf16x6 as above; NP = 2 is the opt-in fp16x3 split further down (two fp16 planes with per-row scales, three products).
#include "vocr_common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4x __attribute__((ext_vector_type(4)));
#define X6_LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define X6_H(v) __builtin_bit_cast(f16x8, v)

constexpr int X6_BM = 256;
constexpr int X6_LDS = 144 * 1024;                  // the ring: 4 x 36 KB (256 x 128 tiles) or 3 x 48 KB (256 x 256)

__device__ __forceinline__ void split3(float a, __bf16& b0, __bf16& b1, __bf16& b2) {
    b0 = (__bf16)a;
    const float r1 = a - (float)b0;
    b1 = (__bf16)r1;
    const float r2 = r1 - (float)b1;
    b2 = (__bf16)r2;
}

// A source matrix may come in two pieces (the two direction planes of the LSTM's gate tensors, the two directions' weight matrices): along K
// (axis 0: k < seg from x, the rest from x2 at k - seg) or along the rows (axis 1: row < seg from x, the rest from x2 at row - seg); seg <= 0: one
// piece; seg is a multiple of 8, so a lane's eight k lie in one piece.
struct X6Src {
    const float* x;
    const float* x2;
    const float* mask;       // K-contiguous source only: element-wise factor with x's addressing (the inter-layer dropout mask), or nullptr
    long ld;
    int seg, axis;
};

// K-contiguous source X[row][k]: one wave per fragment (rt, kk), lane = (row, k half)
__global__ __launch_bounds__(256) void x6_split_rk_kernel(const X6Src src, int rows, int K, bf16x8* __restrict__ planes, int RT, int KK) {
    const long frag = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (frag >= (long)RT * KK) return;
    const int lane = threadIdx.x & 63, rt = (int)(frag / KK), kk = (int)(frag % KK);
    const int row = 32 * rt + (lane & 31), k0 = 16 * kk + 8 * (lane >> 5);
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = 0.f;
    if (row < rows && k0 < K) {
        const bool second = src.seg > 0 && (src.axis == 0 ? k0 : row) >= src.seg;
        const int r_ = second && src.axis == 1 ? row - src.seg : row, k_ = second && src.axis == 0 ? k0 - src.seg : k0;
        const long e = (long)r_ * src.ld + k_;
        const float* p = (second ? src.x2 : src.x) + e;
        if (k0 + 8 <= K && ((((uintptr_t)p) & 15) == 0)) {
            const f32x4 q0 = *(const f32x4*)p, q1 = *(const f32x4*)(p + 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) { v[j] = q0[j]; v[4 + j] = q1[j]; }
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) if (k0 + j < K) v[j] = p[j];
        }
        if (src.mask) {
#pragma unroll
            for (int j = 0; j < 8; ++j) if (k0 + j < K) v[j] *= src.mask[e + j];
        }
    }
    bf16x8 o0, o1, o2;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        __bf16 b0, b1, b2;
        split3(v[j], b0, b1, b2);
        o0[j] = b0; o1[j] = b1; o2[j] = b2;
    }
    const long pstride = (long)RT * KK * 64;
    bf16x8* o = planes + frag * 64 + lane;
    o[0] = o0;
    o[pstride] = o1;
    o[2 * pstride] = o2;
}

// K-strided source X[k][row] (K along the memory rows): lane (row, k half) gathers its eight k
__global__ __launch_bounds__(256) void x6_split_kr_kernel(const X6Src src, int rows, int K, bf16x8* __restrict__ planes, int RT, int KK) {
    const long frag = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (frag >= (long)RT * KK) return;
    // consecutive fragments of a workgroup share kk and take consecutive row tiles: 128 consecutive floats of each source row
    const int lane = threadIdx.x & 63, kk = (int)(frag / RT), rt = (int)(frag % RT);
    const int row = 32 * rt + (lane & 31), k0 = 16 * kk + 8 * (lane >> 5);
    const bool second = src.seg > 0 && (src.axis == 0 ? k0 : row) >= src.seg;
    const int r_ = second && src.axis == 1 ? row - src.seg : row, k_ = second && src.axis == 0 ? k0 - src.seg : k0;
    const float* base = (second ? src.x2 : src.x) + (long)k_ * src.ld + r_;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (row < rows && k0 + j < K) ? base[(long)j * src.ld] : 0.f;
    bf16x8 o0, o1, o2;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        __bf16 b0, b1, b2;
        split3(v[j], b0, b1, b2);
        o0[j] = b0; o1[j] = b1; o2[j] = b2;
    }
    const long pstride = (long)RT * KK * 64;
    bf16x8* o = planes + ((long)rt * KK + kk) * 64 + lane;
    o[0] = o0;
    o[pstride] = o1;
    o[2 * pstride] = o2;
}

// ---------------------------------------------------------------- "fp16x3": two fp16 planes per operand, three products (opt-in; see vocr.h)
// a * s = h0 + h1 (+ at most 2^-23 |a s|) with h0 = fp16(a s), h1 = fp16(a s - h0) and s a power of two PER ROW that brings the row's largest
// magnitude into [2^14, 2^15): fp16's 11 + 11 significant bits (+ the sign of the residual) hold 22 - 23 of fp32's 24, elements more than 2^17 below
// their row's maximum lose low bits to fp16's exponent range (absolute error <= 2^-40 of the row's maximum).  a b ~ h0 g1 + h1 g0 + h0 g0: the dropped
// h1 g1 is <= 2^-24 |a b|.  Half the MFMAs and 2/3 of the operand bytes of bf16x6, a NORM-WISE fp32-grade error bound instead of a per-product one.
// The plane set carries the rows' maxima (fp32, behind the planes); the product kernel undoes both scales in its epilogue (exact: powers of two).
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ unsigned h3_scale_exp(float amax) {
    const unsigned e = (__float_as_uint(amax) >> 23) & 0xffu;
    const unsigned se = 268u - e;                          // amax * 2^(se - 127) in [2^14, 2^15)
    return se > 253u ? 253u : se;                          // (all-zero / denormal rows: any finite scale)
}
__device__ __forceinline__ float h3_scale(float amax) { return __uint_as_float(h3_scale_exp(amax) << 23); }
__device__ __forceinline__ float h3_inv_scale(float amax) { return __uint_as_float((254u - h3_scale_exp(amax)) << 23); }

__device__ __forceinline__ void h3_load8(const X6Src& src, int rows, int K, int row, int k0, float* v) {
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = 0.f;
    if (row < rows && k0 < K) {
        const bool second = src.seg > 0 && (src.axis == 0 ? k0 : row) >= src.seg;
        const int r_ = second && src.axis == 1 ? row - src.seg : row, k_ = second && src.axis == 0 ? k0 - src.seg : k0;
        const long e = (long)r_ * src.ld + k_;
        const float* p = (second ? src.x2 : src.x) + e;
        if (k0 + 8 <= K && ((((uintptr_t)p) & 15) == 0)) {
            const f32x4 q0 = *(const f32x4*)p, q1 = *(const f32x4*)(p + 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) { v[j] = q0[j]; v[4 + j] = q1[j]; }
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) if (k0 + j < K) v[j] = p[j];
        }
        if (src.mask) {
#pragma unroll
            for (int j = 0; j < 8; ++j) if (k0 + j < K) v[j] *= src.mask[e + j];
        }
    }
}
__device__ __forceinline__ void h3_store(const float* v, float s, f16x8* o, long pstride) {
    f16x8 o0, o1;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float xs = v[j] * s;
        const _Float16 h0 = (_Float16)xs;
        o0[j] = h0;
        o1[j] = (_Float16)(xs - (float)h0);
    }
    o[0] = o0;
    o[pstride] = o1;
}

// The rows' maxima by a pass of their own (order-free: integer maximum of the magnitudes' bit patterns, so the result does not depend on the order of the
// atomics), then the planes one wave per fragment as in the bf16 split (the second read comes out of L2 / the memory-side cache).  A caller that knows a
// bound of every element (an LSTM output is inside (-1, 1)) passes it instead and saves the first pass.
// K-contiguous source: one wave per (row, 2048 consecutive k)
__global__ __launch_bounds__(256) void h3_rowmax_kernel(const X6Src src, int rows, int K, unsigned* __restrict__ amax_bits, int chunks) {
    const long w = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (w >= (long)rows * chunks) return;
    const int lane = threadIdx.x & 63, row = (int)(w / chunks), c = (int)(w % chunks);
    float v[8], m = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        h3_load8(src, rows, K, row, 2048 * c + 8 * (64 * i + lane), v);
#pragma unroll
        for (int j = 0; j < 8; ++j) m = fmaxf(m, fabsf(v[j]));
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if (lane == 0) atomicMax(amax_bits + row, __float_as_uint(m));
}
__global__ __launch_bounds__(256) void h3_split_rk_kernel(const X6Src src, int rows, int K, f16x8* __restrict__ planes, const float* __restrict__ amax, int RT,
                                                          int KK, float bound) {
    const long frag = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (frag >= (long)RT * KK) return;
    const int lane = threadIdx.x & 63, rt = (int)(frag / KK), kk = (int)(frag % KK);
    const int row = 32 * rt + (lane & 31);
    float v[8];
    h3_load8(src, rows, K, row, 16 * kk + 8 * (lane >> 5), v);
    h3_store(v, h3_scale(bound > 0.f ? bound : amax[row]), planes + frag * 64 + lane, (long)RT * KK * 64);
}
__global__ void h3_fill_kernel(float* __restrict__ p, int n, int n_set, float v) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = i < n_set ? v : 0.f;
}

// K-strided source X[k][row]: the maxima over k by a pass of their own (order-free: integer max of the magnitudes' bit patterns), then the planes
__global__ __launch_bounds__(256) void h3_colmax_kernel(const X6Src src, int rows, int K, unsigned* __restrict__ amax_bits) {
    const int row = blockIdx.x * 256 + threadIdx.x, k0 = blockIdx.y * 64;
    if (row >= rows) return;
    float m = 0.f;
#pragma unroll 8
    for (int j = 0; j < 64; ++j) {
        const int k = k0 + j;
        if (k >= K) break;
        const bool second = src.seg > 0 && (src.axis == 0 ? k : row) >= src.seg;
        const int r_ = second && src.axis == 1 ? row - src.seg : row, k_ = second && src.axis == 0 ? k - src.seg : k;
        m = fmaxf(m, fabsf((second ? src.x2 : src.x)[(long)k_ * src.ld + r_]));
    }
    atomicMax(amax_bits + row, __float_as_uint(m));
}
__global__ __launch_bounds__(256) void h3_split_kr_kernel(const X6Src src, int rows, int K, f16x8* __restrict__ planes, const float* __restrict__ amax, int RT,
                                                          int KK, float bound) {
    const long frag = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (frag >= (long)RT * KK) return;
    const int lane = threadIdx.x & 63, kk = (int)(frag / RT), rt = (int)(frag % RT);
    const int row = 32 * rt + (lane & 31), k0 = 16 * kk + 8 * (lane >> 5);
    const bool second = src.seg > 0 && (src.axis == 0 ? k0 : row) >= src.seg;
    const int r_ = second && src.axis == 1 ? row - src.seg : row, k_ = second && src.axis == 0 ? k0 - src.seg : k0;
    const float* base = (second ? src.x2 : src.x) + (long)k_ * src.ld + r_;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (row < rows && k0 + j < K) ? base[(long)j * src.ld] : 0.f;
    h3_store(v, h3_scale(bound > 0.f ? bound : amax[row]), planes + ((long)rt * KK + kk) * 64 + lane, (long)RT * KK * 64);
}

struct X6Args {
    const void* a;           // planes of A: [3][RTa][KKa][64][8] bf16
    const void* b;           // planes of B: [3][RTb][KKb][64][8] bf16
    float* c[2];             // two outputs, both with ldc: cut along the columns (col >= csplit -> c[1] at col - csplit) or the rows (rsplit)
    const float* bias[2];
    float* slab;             // K cut: [split][tile of this launch][256][128] partial sums
    int M, N, nkk;           // rows, columns, k16 steps of the product
    int RTa, RTb, KKa, KKb;  // padded row tiles and k16 steps of the two plane sets (their strides)
    int a_rt0, a_kk0, b_rt0, b_kk0;      // where the product's operands start inside the plane sets (views: shifted rows / k windows)
    int alt_mt, a_kk0_alt, b_rt0_alt, b_kk0_alt;     // row tiles >= alt_mt (256-row tiles) read these views instead: two products of one shape in one launch
    int csplit, rsplit, ldc;
    int mtiles, ntiles, tile0, ntile_launch, ksplit, stages_per_split;
    int relu;
    const float* sa;         // fp16x3 plane sets: the rows' maxima of A and B (behind their planes), indexed by plane-set row
    const float* sb;
};

// Which tile a workgroup takes.  The dispatcher deals workgroups to the eight XCDs in turn (blockIdx % 8 - up to a rotation), each with its own
// L2: XCD x takes a CONTIGUOUS range of the launch's (K split, tile) sequence, and the tile sequence runs through panels of four row tiles
// column by column, so that the 32 workgroups an XCD runs at a time cover 4 row tiles x 8 column tiles and their k16 stages of A and B come
// out of that L2 (consecutive tiles on consecutive XCDs made every XCD stream all of A: 8 x 231 MB through the fabric for the data gradient).
constexpr int X6_PANEL = 4;
__device__ __forceinline__ void x6_tile_of(int t, int mtiles, int ntiles, int& mt, int& nt) {
    const int panel = t / (X6_PANEL * ntiles), within = t - panel * X6_PANEL * ntiles;
    const int rows = min(X6_PANEL, mtiles - panel * X6_PANEL);
    nt = within / rows;
    mt = panel * X6_PANEL + within - nt * rows;
}
__device__ __forceinline__ int x6_xcd_chunked(int i, int n) {
    const int xcd = i & 7, j = i >> 3, c = n >> 3, rr = n & 7;
    return xcd * c + min(xcd, rr) + j;
}

// NCT = column tiles of 32 per workgroup tile: 4 (256 x 128: eight waves as 4 x 2 with 64 x 64 wave tiles, 36-KB stages, four ring slots) or 8
// (256 x 256: 2 x 4 waves with 128 x 64 wave tiles - 128 accumulator registers -, 48-KB stages, three slots: half the DMA instructions per MFMA)
// NP = planes per operand: 3 (bf16x6: six products per k16 step and tile pair) or 2 (fp16x3: three)
template <int NCT, int NP>
__global__ __launch_bounds__(512) void gemm_x6_kernel(const X6Args g) {
    constexpr int BN = 32 * NCT;
    constexpr int TM = NCT == 4 ? 2 : 4, TN = 2;
    constexpr int NA = 8 * NP;                                // A fragments of a stage
    constexpr int NFRAG = NA + NP * NCT;                      // fragments (KB) of a stage: 36 / 48 (bf16x6), 24 / 32 (fp16x3)
    constexpr int NS = NP == 3 ? (NCT == 4 ? 4 : 3) : (NCT == 4 ? 6 : 4);       // ring slots: 144 KB (128 for fp16x3 on the wide tile)
    constexpr int STAGE = NFRAG * 1024;
    constexpr int NDMA_MAX = (NFRAG + 7) / 8;                 // DMA instructions of a wave per stage: NFRAG / 8, one more on the first NFRAG % 8 waves
    constexpr int NDMA_LO = NFRAG / 8, NDMA_ODD = NFRAG % 8;
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = NCT == 4 ? (wave & 3) : (wave & 1), wn = NCT == 4 ? (wave >> 2) : (wave >> 1);
    const int v = x6_xcd_chunked(blockIdx.x, gridDim.x);
    const int ltile = v % g.ntile_launch, split = v / g.ntile_launch;
    int mt, nt;
    x6_tile_of(g.tile0 + ltile, g.mtiles, g.ntiles, mt, nt);
    const int s_beg = split * g.stages_per_split, s_end = min(g.nkk, s_beg + g.stages_per_split);      // K16 stages
    if (s_beg >= s_end) return;

    const long pa = (long)g.RTa * g.KKa * 1024, pb = (long)g.RTb * g.KKb * 1024;       // plane strides in bytes
    const bool alt = mt >= g.alt_mt;
    const int a_kk0 = alt ? g.a_kk0_alt : g.a_kk0, b_kk0 = alt ? g.b_kk0_alt : g.b_kk0, b_rt0 = alt ? g.b_rt0_alt : g.b_rt0;
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)g.a, 0, (int)(NP * pa), 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)g.b, 0, (int)(NP * pb), 0x00020000);
    const int voff = lane * 16;
    // A stage = one k16 step: 24 A fragments (f = plane * 8 + row tile) + 3 NCT B fragments (24 + plane * NCT + column tile) of 1 KB.  Wave w moves
    // fragments w, w + 8, ...: a wave-uniform number of instructions per stage.  A stage past the end is "moved" from out of range (zeros into a slot
    // nobody reads), so that the counts never change.
    auto dma_one = [&](int s, int slot, int i) {
        const int f = wave + 8 * i;
        unsigned char* dst = lds + slot * STAGE + f * 1024;
        const bool live = s < s_end;
#if defined(X6_CUT) && (X6_CUT & 16)
        s = s_beg;                                            // diagnostic: every stage re-reads the first one (L2 hits)
#endif
        if (f < NA) {
            const long off = (f >> 3) * pa + ((long)(g.a_rt0 + 8 * mt + (f & 7)) * g.KKa + a_kk0 + s) * 1024;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, X6_LDS_PTR(dst), 16, live ? voff : -16, live ? (int)off : 0, 0, 0);
        } else {
            const int fb = f - NA;
            const long off = (fb / NCT) * pb + ((long)(b_rt0 + NCT * nt + (fb % NCT)) * g.KKb + b_kk0 + s) * 1024;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, X6_LDS_PTR(dst), 16, live ? voff : -16, live ? (int)off : 0, 0, 0);
        }
    };
    auto dma_nth = [&](int s, int slot, int i) {              // the i-th DMA of this wave's stage, if it has one
        if (8 * i + 7 < NFRAG || wave + 8 * i < NFRAG) dma_one(s, slot, i);
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // prologue: NS - 1 stages in flight
#pragma unroll
    for (int i = 0; i < NS - 1; ++i)
#pragma unroll
        for (int d = 0; d < NDMA_MAX; ++d) dma_nth(s_beg + i, i, d);
#ifndef X6_CUT
#define X6_CUT 0              // diagnostic builds (scripts/x6_bench.py), WRONG results: 1 no DMA in the loop, 2 no MFMA, 4 no fragment reads, 8 no barrier, 16 every DMA from the first stage (L2 hits), 32 no wait for the DMAs
#endif
    int slot = 0;
    for (int s = s_beg; s < s_end; ++s) {
        // stage s has landed for this wave when only the NS - 2 younger stages' DMAs are outstanding
        if (!(X6_CUT & 1) && !(X6_CUT & 32)) {
            static_assert((NS - 2) * (NDMA_LO + (NDMA_ODD ? 1 : 0)) < 16, "vmcnt immediate");
            if (NDMA_ODD && wave < NDMA_ODD) __builtin_amdgcn_s_waitcnt(0x0F70 | ((NS - 2) * (NDMA_LO + 1)));
            else __builtin_amdgcn_s_waitcnt(0x0F70 | ((NS - 2) * NDMA_LO));
        }
        if (!(X6_CUT & 8)) __builtin_amdgcn_s_barrier();     // ... for everybody; and everybody is done reading the slot of stage s - 1
        const unsigned char* base = lds + slot * STAGE + lane * 16;
        bf16x8 af[TM][NP], bf[TN][NP];                        // (fp16x3: the same 16 bytes, re-typed at the MFMA)
        // fragments in the order the MFMAs want them
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
            for (int p = 0; p < NP; ++p) bf[tn][p] = *(const bf16x8*)(base + (NA + p * NCT + TN * wn + tn) * 1024);
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int p = 0; p < NP; ++p) af[tm][p] = *(const bf16x8*)(base + (p * 8 + TM * wm + tm) * 1024);
        __builtin_amdgcn_sched_barrier(0);
        const int snext = s + NS - 1, slot_next = slot == 0 ? NS - 1 : slot - 1;      // the slot of stage s - 1
        int d = 0;
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) {
                if (!(X6_CUT & 2)) {
                    // the small partial products first
                    if constexpr (NP == 3) {
                        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[tm][1], bf[tn][1], acc[tm][tn], 0, 0, 0);
                        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[tm][0], bf[tn][2], acc[tm][tn], 0, 0, 0);
                        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[tm][2], bf[tn][0], acc[tm][tn], 0, 0, 0);
                    } else {
                        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(X6_H(af[tm][0]), X6_H(bf[tn][1]), acc[tm][tn], 0, 0, 0);
                        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(X6_H(af[tm][1]), X6_H(bf[tn][0]), acc[tm][tn], 0, 0, 0);
                    }
                }
                // the DMAs of stage s + NS - 1 one at a time between the MFMA groups
                if (d < NDMA_MAX && !(X6_CUT & 1)) dma_nth(snext, slot_next, d);
                ++d;
                if (!(X6_CUT & 2)) {
                    if constexpr (NP == 3) {
                        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[tm][0], bf[tn][1], acc[tm][tn], 0, 0, 0);
                        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[tm][1], bf[tn][0], acc[tm][tn], 0, 0, 0);
                        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[tm][0], bf[tn][0], acc[tm][tn], 0, 0, 0);
                    } else {
                        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(X6_H(af[tm][0]), X6_H(bf[tn][0]), acc[tm][tn], 0, 0, 0);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
        for (int dd = TM * TN; dd < NDMA_MAX; ++dd)
            if (!(X6_CUT & 1)) dma_nth(snext, slot_next, dd);
        slot = slot == NS - 1 ? 0 : slot + 1;
    }

    // ---- C/D map of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
    const int li = lane & 31, lk = lane >> 5;
    const bool slabbed = g.slab != nullptr;
    float* const sl = slabbed ? g.slab + ((long)split * g.ntile_launch + ltile) * (X6_BM * BN) : nullptr;
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
        const int lcol = 32 * (TN * wn + tn) + li, col = BN * nt + lcol;
        if (col >= g.N) continue;
        const int cw = col >= g.csplit ? 1 : 0;
        const float bvc = (!slabbed && g.bias[cw]) ? g.bias[cw][col - cw * g.csplit] : 0.f;
        float ib = 1.f;
        if constexpr (NP == 2) ib = h3_inv_scale(g.sb[32 * b_rt0 + col]);
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int lrow = 32 * (TM * wm + tm) + (r & 3) + 8 * (r >> 2) + 4 * lk, row = X6_BM * mt + lrow;
                if (row < g.M) {
                    float av = acc[tm][tn][r];
                    if constexpr (NP == 2) av = (av * h3_inv_scale(g.sa[32 * g.a_rt0 + row])) * ib;      // both rows' scales undone (exact)
                    if (slabbed) {
                        sl[lrow * BN + lcol] = av;
                    } else {
                        const int rw = row >= g.rsplit ? 1 : 0, which = cw | rw;
                        float v = av + bvc;
                        if (g.relu) v = fmaxf(v, 0.f);
                        g.c[which][(long)(row - rw * g.rsplit) * g.ldc + (col - cw * g.csplit)] = v;
                    }
                }
            }
    }
}

// the tiles of a K-cut launch: C = sum over the splits' slabs in split order (+ bias)(relu)
template <int NCT>
__global__ __launch_bounds__(256) void gemm_x6_reduce_kernel(const X6Args g) {
    constexpr int BN = 32 * NCT, RPW = 256 / BN * 4;                             // rows per workgroup: 8 (BN = 128) or 4
    const int parts = X6_BM / RPW;
    const int ltile = blockIdx.x / parts, part = blockIdx.x % parts;
    int mt, nt;
    x6_tile_of(g.tile0 + ltile, g.mtiles, g.ntiles, mt, nt);
    const int lcol = threadIdx.x % BN, col = BN * nt + lcol;
    if (col >= g.N) return;
    const int cw = col >= g.csplit ? 1 : 0;
    const float bv = g.bias[cw] ? g.bias[cw][col - cw * g.csplit] : 0.f;
    for (int rr = threadIdx.x / BN; rr < RPW; rr += 256 / BN) {
        const int lrow = RPW * part + rr, row = X6_BM * mt + lrow;
        if (row >= g.M) continue;
        float v = 0.f;
        for (int s = 0; s < g.ksplit; ++s) v += g.slab[((long)s * g.ntile_launch + ltile) * (X6_BM * BN) + lrow * BN + lcol];
        v += bv;
        if (g.relu) v = fmaxf(v, 0.f);
        const int rw = row >= g.rsplit ? 1 : 0, which = cw | rw;
        g.c[which][(long)(row - rw * g.rsplit) * g.ldc + (col - cw * g.csplit)] = v;
    }
}

int x6_cu_count() {
    static int cus[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    if (!cus[dev]) {
        hipDeviceProp_t prop;
        cus[dev] = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
    }
    return cus[dev];
}

}  // namespace

static inline int x6_rt(int rows) { return vocr_cdiv(rows, 256) * 8; }
static inline int x6_kk(int k) { return vocr_cdiv(k, 32) * 2; }

// bytes of a plane set: NP planes of [RT][KK] 1-KB fragments; fp16x3 sets carry their rows' maxima (fp32) behind the planes
static inline size_t x6_planes_bytes(int np, int rows, int k) {
    if (rows <= 0 || k <= 0) return 0;
    return (size_t)np * x6_rt(rows) * x6_kk(k) * 1024 + (np == 2 ? (size_t)x6_rt(rows) * 32 * sizeof(float) : 0);
}
extern "C" size_t vocr_gemm_x6_planes_bytes(int rows, int k) { return x6_planes_bytes(3, rows, k); }
extern "C" size_t vocr_gemm_h3_planes_bytes(int rows, int k) { return x6_planes_bytes(2, rows, k); }

template <int NP>
static int x6_split_impl(const char* name, const float* x, const float* x2, int seg, int seg_axis, const float* mask, long ld, int rows, int k,
                         int k_contiguous, void* planes, void* stream, float bound = 0.f) {
    VOCR_CHECK_ARG(x && planes, "%s: null pointer", name);
    VOCR_CHECK_ARG(rows > 0 && k > 0 && ld > 0, "%s: bad shape (%d x %d, ld %ld)", name, rows, k, ld);
    VOCR_CHECK_ARG((((uintptr_t)planes) & 15) == 0, "%s: the planes must be 16-byte aligned", name);
    VOCR_CHECK_ARG(seg <= 0 || (x2 && seg % 8 == 0 && (seg_axis == 0 || seg_axis == 1)), "%s: a second piece needs x2, seg %% 8 == 0, axis 0 / 1", name);
    VOCR_CHECK_ARG(!mask || k_contiguous, "%s: a mask only with a K-contiguous source", name);
    const int RT = x6_rt(rows), KK = x6_kk(k);
    VOCR_CHECK_ARG((long)NP * RT * KK * 1024 < (1l << 31), "%s: operand too large for 32-bit fragment offsets", name);
    const long frags = (long)RT * KK;
    const X6Src src = {x, x2, mask, ld, seg, seg_axis};
    hipStream_t s = (hipStream_t)stream;
    if (NP == 3) {
        if (k_contiguous) x6_split_rk_kernel<<<(unsigned)vocr_cdiv(frags, 4), 256, 0, s>>>(src, rows, k, (bf16x8*)planes, RT, KK);
        else x6_split_kr_kernel<<<(unsigned)vocr_cdiv(frags, 4), 256, 0, s>>>(src, rows, k, (bf16x8*)planes, RT, KK);
    } else {
        float* amax = (float*)((unsigned char*)planes + (size_t)NP * frags * 1024);
        if (bound > 0.f) {
            // every row's maximum := the caller's bound (the padding rows': 0)
            h3_fill_kernel<<<vocr_cdiv(RT * 32, 256), 256, 0, s>>>(amax, RT * 32, rows, bound);
        } else {
            if (hipMemsetAsync(amax, 0, (size_t)RT * 32 * sizeof(float), s) != hipSuccess) {
                vocr_set_error("%s: hipMemsetAsync failed", name);
                return VOCR_ELAUNCH;
            }
            if (k_contiguous) {
                const int chunks = vocr_cdiv(k, 2048);
                h3_rowmax_kernel<<<(unsigned)vocr_cdiv((long)rows * chunks, 4), 256, 0, s>>>(src, rows, k, (unsigned*)amax, chunks);
            } else {
                h3_colmax_kernel<<<dim3((unsigned)vocr_cdiv(rows, 256), (unsigned)vocr_cdiv(k, 64)), 256, 0, s>>>(src, rows, k, (unsigned*)amax);
            }
        }
        if (k_contiguous) h3_split_rk_kernel<<<(unsigned)vocr_cdiv(frags, 4), 256, 0, s>>>(src, rows, k, (f16x8*)planes, amax, RT, KK, bound);
        else h3_split_kr_kernel<<<(unsigned)vocr_cdiv(frags, 4), 256, 0, s>>>(src, rows, k, (f16x8*)planes, amax, RT, KK, bound);
    }
    VOCR_CHECK_LAUNCH(name);
    return VOCR_OK;
}
extern "C" int vocr_gemm_x6_split(const float* x, const float* x2, int seg, int seg_axis, const float* mask, long ld, int rows, int k, int k_contiguous,
                                  void* planes, void* stream) {
    return x6_split_impl<3>("vocr_gemm_x6_split", x, x2, seg, seg_axis, mask, ld, rows, k, k_contiguous, planes, stream);
}
extern "C" int vocr_gemm_h3_split(const float* x, const float* x2, int seg, int seg_axis, const float* mask, long ld, int rows, int k, int k_contiguous,
                                  float bound, void* planes, void* stream) {
    VOCR_CHECK_ARG(bound >= 0.f && bound < 1e30f, "vocr_gemm_h3_split: bound must be 0 (none) or a finite positive number");
    return x6_split_impl<2>("vocr_gemm_h3_split", x, x2, seg, seg_axis, mask, ld, rows, k, k_contiguous, planes, stream, bound);
}

extern "C" size_t vocr_gemm_x6_workspace_bytes(int m, int n, int k) {
    if (m <= 0 || n <= 0 || k <= 0) return 0;
    // at most one round of workgroups writes slabs: CU count tiles of 256 x 256
    return (size_t)(x6_cu_count() + 8) * X6_BM * 256 * sizeof(float);
}

namespace {
struct X6Plan { int nct, ntiles, tiles, full, rem, ks; double cost; };
// whole rounds of one workgroup per CU run their tiles over the full K; what is left of the last round - when it would leave at least half of
// the chip idle - is cut along K into slabs, so that it takes 1 / ks of a round (294 x 8 tiles of the data gradient: 1 + 1/6 rounds instead of 2).
// Cost in units of one 256 x 128 tile over the full K; the narrow tile pays ~12 % for twice the DMA instructions per MFMA (measured).
X6Plan x6_plan(int nct, int m, int n, int nkk, int ncu, bool can_cut) {
    X6Plan p;
    p.nct = nct;
    p.ntiles = vocr_cdiv(n, 32 * nct);
    p.tiles = vocr_cdiv(m, X6_BM) * p.ntiles;
    p.full = (p.tiles / ncu) * ncu;
    p.rem = p.tiles - p.full;
    p.ks = 1;
    if (p.rem > 0 && p.rem * 2 <= ncu && nkk >= 32 && can_cut) {
        p.ks = ncu / p.rem;
        if (p.ks > 16) p.ks = 16;
        if (p.ks > nkk / 8) p.ks = nkk / 8;
        if (p.ks < 1) p.ks = 1;
    }
    const double per_tile = nct == 8 ? 2.0 : 1.12;
    p.cost = (p.full / ncu + (p.rem > 0 ? 1.0 / p.ks + (p.ks > 1 ? 0.03 : 0.0) : 0.0)) * per_tile;
    return p;
}

template <int NCT, int NP>
void x6_launch(X6Args g, const X6Plan& p, void* workspace, hipStream_t s) {
    g.ntiles = p.ntiles;
    const int main_tiles = p.ks > 1 ? p.full : p.tiles;
    if (main_tiles > 0) {
        g.tile0 = 0; g.ntile_launch = main_tiles; g.ksplit = 1; g.stages_per_split = g.nkk; g.slab = nullptr;
        gemm_x6_kernel<NCT, NP><<<main_tiles, 512, X6_LDS, s>>>(g);
    }
    if (p.ks > 1) {
        g.tile0 = p.full; g.ntile_launch = p.rem;
        g.stages_per_split = vocr_cdiv(g.nkk, p.ks);
        g.ksplit = vocr_cdiv(g.nkk, g.stages_per_split);
        g.slab = (float*)workspace;
        gemm_x6_kernel<NCT, NP><<<p.rem * g.ksplit, 512, X6_LDS, s>>>(g);
        gemm_x6_reduce_kernel<NCT><<<p.rem * (NCT == 4 ? 32 : 64), 256, 0, s>>>(g);
    }
}
}  // namespace

// C[m][n] = A . B^T (+ bias)(relu) from split planes.  A: rows a_row0 .. + m and k16 steps a_kk0 .. + k/16 of a plane set written for
// (a_rows, a_k); B likewise (rows = output columns).  Two outputs: columns >= csplit go to c1 at col - csplit, or rows >= rsplit to c1 at
// row - rsplit (0: no cut); a_row0 / b_row0 multiples of 32, k a multiple of 16.
struct X6AltViews { int a_kk0, b_row0, b_kk0; };          // the views of the rows >= rsplit of a two-view product
template <int NP>
static int x6_gemm_impl(const char* name, const void* a_planes, int a_rows, int a_k, int a_row0, int a_kk0, const void* b_planes, int b_rows, int b_k,
                        int b_row0, int b_kk0, int m, int n, int k, float* c0, float* c1, int csplit, int rsplit, int ldc, const float* bias0,
                        const float* bias1, int relu, void* workspace, void* stream, const X6AltViews* alt = nullptr) {
    VOCR_CHECK_ARG(a_planes && b_planes && c0, "%s: null pointer", name);
    VOCR_CHECK_ARG(m > 0 && n > 0 && k > 0 && ldc > 0 && k % 16 == 0, "%s: bad shape (k %% 16 == 0)", name);
    VOCR_CHECK_ARG(a_row0 % 32 == 0 && b_row0 % 32 == 0 && a_row0 >= 0 && b_row0 >= 0 && a_kk0 >= 0 && b_kk0 >= 0, "%s: views start on tile boundaries", name);
    VOCR_CHECK_ARG((csplit <= 0 && rsplit <= 0) || c1, "%s: a second output needs c1", name);
    VOCR_CHECK_ARG(!(csplit > 0 && rsplit > 0), "%s: one cut, along the columns or along the rows", name);
    X6Args g;
    g.a = a_planes; g.b = b_planes;
    g.c[0] = c0; g.c[1] = c1 ? c1 : c0;
    g.bias[0] = bias0; g.bias[1] = bias1;
    g.M = m; g.N = n; g.nkk = k / 16;
    g.RTa = x6_rt(a_rows); g.KKa = x6_kk(a_k); g.RTb = x6_rt(b_rows); g.KKb = x6_kk(b_k);
    g.a_rt0 = a_row0 / 32; g.a_kk0 = a_kk0; g.b_rt0 = b_row0 / 32; g.b_kk0 = b_kk0;
    g.alt_mt = 1 << 30; g.a_kk0_alt = a_kk0; g.b_rt0_alt = g.b_rt0; g.b_kk0_alt = b_kk0;
    if (alt) {
        VOCR_CHECK_ARG(rsplit > 0 && rsplit % X6_BM == 0 && alt->b_row0 % 32 == 0 && alt->b_row0 >= 0 && alt->a_kk0 >= 0 && alt->b_kk0 >= 0,
                       "%s: a second view set needs a row cut on a 256-row tile boundary and views on tile boundaries", name);
        g.alt_mt = rsplit / X6_BM; g.a_kk0_alt = alt->a_kk0; g.b_rt0_alt = alt->b_row0 / 32; g.b_kk0_alt = alt->b_kk0;
    }
    g.mtiles = vocr_cdiv(m, X6_BM);
    g.csplit = csplit > 0 ? csplit : (1 << 30);
    g.rsplit = rsplit > 0 ? rsplit : (1 << 30);
    g.ldc = ldc;
    g.relu = relu;
    // fp16x3: the rows' maxima behind the planes
    g.sa = NP == 2 ? (const float*)((const unsigned char*)a_planes + (size_t)NP * g.RTa * g.KKa * 1024) : nullptr;
    g.sb = NP == 2 ? (const float*)((const unsigned char*)b_planes + (size_t)NP * g.RTb * g.KKb * 1024) : nullptr;
    const int ncu = x6_cu_count();
    const X6Plan p4 = x6_plan(4, m, n, g.nkk, ncu, workspace != nullptr), p8 = x6_plan(8, m, n, g.nkk, ncu, workspace != nullptr);
    // the wide tile reads whole blocks of 8 column tiles: they must exist in B's plane set (rows padded to 256: always, unless a view starts late)
    const bool wide_ok = g.b_rt0 + 8 * p8.ntiles <= g.RTb && g.b_rt0_alt + 8 * p8.ntiles <= g.RTb;
    const X6Plan& p = (wide_ok && p8.cost < p4.cost) ? p8 : p4;
    // a tile reads whole 256-row / (32 NCT)-column blocks of fragments: they must exist in the plane sets (zero padding or later rows)
    VOCR_CHECK_ARG(g.a_rt0 + 8 * g.mtiles <= g.RTa && g.b_rt0 + p.nct * p.ntiles <= g.RTb && g.a_kk0 + g.nkk <= g.KKa && g.b_kk0 + g.nkk <= g.KKb,
                   "%s: the view leaves its plane set", name);
    VOCR_CHECK_ARG(g.b_rt0_alt + p.nct * p.ntiles <= g.RTb && g.a_kk0_alt + g.nkk <= g.KKa && g.b_kk0_alt + g.nkk <= g.KKb,
                   "%s: the second view leaves its plane set", name);
    static bool lds_ok = false;
    if (!lds_ok) {
        if (hipFuncSetAttribute((const void*)gemm_x6_kernel<4, NP>, hipFuncAttributeMaxDynamicSharedMemorySize, X6_LDS) != hipSuccess ||
            hipFuncSetAttribute((const void*)gemm_x6_kernel<8, NP>, hipFuncAttributeMaxDynamicSharedMemorySize, X6_LDS) != hipSuccess) {
            vocr_set_error("%s: hipFuncSetAttribute failed", name);
            return VOCR_ELAUNCH;
        }
        lds_ok = true;
    }
    if (p.nct == 8) x6_launch<8, NP>(g, p, workspace, (hipStream_t)stream);
    else x6_launch<4, NP>(g, p, workspace, (hipStream_t)stream);
    VOCR_CHECK_LAUNCH(name);
    return VOCR_OK;
}

extern "C" int vocr_gemm_x6(const void* a_planes, int a_rows, int a_k, int a_row0, int a_kk0, const void* b_planes, int b_rows, int b_k, int b_row0,
                            int b_kk0, int m, int n, int k, float* c0, float* c1, int csplit, int rsplit, int ldc, const float* bias0,
                            const float* bias1, int relu, void* workspace, void* stream) {
    return x6_gemm_impl<3>("vocr_gemm_x6", a_planes, a_rows, a_k, a_row0, a_kk0, b_planes, b_rows, b_k, b_row0, b_kk0, m, n, k, c0, c1, csplit, rsplit, ldc,
                           bias0, bias1, relu, workspace, stream);
}
// the same product from fp16x3 plane sets (vocr_gemm_h3_split)
extern "C" int vocr_gemm_h3(const void* a_planes, int a_rows, int a_k, int a_row0, int a_kk0, const void* b_planes, int b_rows, int b_k, int b_row0,
                            int b_kk0, int m, int n, int k, float* c0, float* c1, int csplit, int rsplit, int ldc, const float* bias0,
                            const float* bias1, int relu, void* workspace, void* stream) {
    return x6_gemm_impl<2>("vocr_gemm_h3", a_planes, a_rows, a_k, a_row0, a_kk0, b_planes, b_rows, b_k, b_row0, b_kk0, m, n, k, c0, c1, csplit, rsplit, ldc,
                           bias0, bias1, relu, workspace, stream);
}

// Two products of one shape from ONE pair of plane sets in one launch: rows < rsplit (a multiple of 256) of C = A[a_row0 ..][a_kk0 ..] . B[b_row0 ..][b_kk0 ..]^T
// go to c0, rows >= rsplit read the views (a_kk0_2, b_row0_2, b_kk0_2) and go to c1 at row - rsplit: the recurrent weight gradients of both directions
// (each a time-shifted k window of the gate gradients' and the outputs' transposed planes) - half the K-cut slabs of two launches.
extern "C" int vocr_gemm_x6_two_views(const void* a_planes, int a_rows, int a_k, int a_row0, const void* b_planes, int b_rows, int b_k, int m, int n, int k,
                                      int rsplit, int a_kk0, int b_row0, int b_kk0, int a_kk0_2, int b_row0_2, int b_kk0_2, float* c0, float* c1, int ldc,
                                      void* workspace, void* stream) {
    const X6AltViews alt = {a_kk0_2, b_row0_2, b_kk0_2};
    return x6_gemm_impl<3>("vocr_gemm_x6_two_views", a_planes, a_rows, a_k, a_row0, a_kk0, b_planes, b_rows, b_k, b_row0, b_kk0, m, n, k, c0, c1, 0, rsplit, ldc,
                           nullptr, nullptr, 0, workspace, stream, &alt);
}
extern "C" int vocr_gemm_h3_two_views(const void* a_planes, int a_rows, int a_k, int a_row0, const void* b_planes, int b_rows, int b_k, int m, int n, int k,
                                      int rsplit, int a_kk0, int b_row0, int b_kk0, int a_kk0_2, int b_row0_2, int b_kk0_2, float* c0, float* c1, int ldc,
                                      void* workspace, void* stream) {
    const X6AltViews alt = {a_kk0_2, b_row0_2, b_kk0_2};
    return x6_gemm_impl<2>("vocr_gemm_h3_two_views", a_planes, a_rows, a_k, a_row0, a_kk0, b_planes, b_rows, b_k, b_row0, b_kk0, m, n, k, c0, c1, 0, rsplit, ldc,
                           nullptr, nullptr, 0, workspace, stream, &alt);
}
