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
#include "conv_tail.h"

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

// Segment geometry.  A SEGMENT is one MFMA N-tile: 32 lane positions over a 34-wide halo patch row.  An image row of
// W pixels gives FS = W/32 full segments (one sub-row: 34 patch columns, 32 outputs); the RW = W%32 pixels left over at
// the right edge of RR consecutive image rows share ONE remainder segment: its 34 patch columns are RR sub-rows of
// RW+2 columns (left halo, RW pixels, zero right pad), RR = floor(34/(RW+2)), so lane position q is pixel (q/(RW+2),
// q%(RW+2)) and the B operand of tap kw is still patch[q + kw].  The MFMA loops do not know the difference; only the
// loaders' per-lane offsets and the store masks do.  (A remainder used to cost a whole segment per row: 8 % of the MFMA
// work at W = 294, 6 % at W = 420.)
struct SegGeom { int FS, RW, RR, per_img, nseg; };

__host__ __device__ inline SegGeom seg_geom(int N, int H, int W) {
    SegGeom g;
    g.FS = W / SEGW;
    g.RW = W % SEGW;
    g.RR = g.RW ? (PROW / (g.RW + 2)) : 1;
    if (g.RR > H) g.RR = H;
    g.per_img = H * g.FS + (g.RW ? (H + g.RR - 1) / g.RR : 0);
    g.nseg = N * g.per_img;
    return g;
}

__device__ __forceinline__ SegInfo seg_decode(int gidx, const SegGeom& g, int H, long chw) {
    SegInfo s;
    s.valid = gidx < g.nseg;
    const int gg = s.valid ? gidx : 0;
    s.n = gg / g.per_img;
    const int loc = gg - s.n * g.per_img;
    const int nfull = H * g.FS;
    if (loc < nfull) {
        s.h = loc / g.FS;
        s.w0 = (loc - s.h * g.FS) * SEGW;
        s.rows = 1; s.pw = PROW; s.ow = SEGW;
    } else {
        s.h = (loc - nfull) * g.RR;
        s.w0 = g.FS * SEGW;
        s.rows = min(g.RR, H - s.h); s.pw = g.RW + 2; s.ow = g.RW;
    }
    s.base = (long)s.n * chw;
    return s;
}

// XCD-aware workgroup order (cdna_hip_programming.md T1, bijective form): workgroups are dealt round-robin to the 8 XCDs,
// so linear id b runs on XCD b % 8; giving XCD x the x-th contiguous slice of the work list keeps the rows h-1, h, h+1 of an
// image (and both output-channel tiles of a pixel tile) in ONE XCD's L2 instead of three.  Speed only, never correctness.
__device__ __forceinline__ int xcd_slice_order(int b, int nwg) {
    const int x = b & 7, q = nwg >> 3, r = nwg & 7;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (b >> 3);
}

// Out-of-range operands are read from this zero page instead of being selected after the load: the ADDRESS is
// selected, the load itself stays unconditional, so hipcc keeps all of a chunk's loads in flight together
// (a select on the loaded value gets turned into one exec-masked branch + s_waitcnt vmcnt(0) per load).
__device__ __attribute__((aligned(16))) float g_zero_page[64];

// Loader layout (index arithmetic is wave-uniform or hoisted, so the chunk loop issues almost no VALU):
//   patch : wave w stages the halo rows of its own segment(s): row = (ci, kh), lanes 0..33 = the 34 columns
//   weights: 16-byte loads, a thread keeps one float4 column and walks rows with a constant stride
//   K order inside a chunk: MFMA step ks multiplies k = ks (lanes 0-31) and k = ks + 36 (lanes 32-63), i.e.
//   channels ci and ci+4 of the chunk, so both A and B fragment addresses are lane-constant base + immediate.
// SPWV = 32-pixel segments per wave (2: wave tile 64co x 64px, 1: 64co x 32px).  The smaller tile halves the work of
// a workgroup: at batch 32 the layers have only 4-7 full-size workgroups per CU and the last partial round costs 20 %
// (measured 88 TF at 4.4 WG/CU vs 111 TF at 16 WG/CU); twice as many half-size workgroups let the hardware
// dispatcher balance the tail.
#ifdef VOCR_CONV_STAMPS        // diagnostic build only (scripts/conv_stamp.hip): phase stamps of one workgroup, never in libvocr.so
__device__ unsigned long long* g_stamp_out;
#define VOCR_STAMP(t) do { asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory"); } while (0)
#endif

template <int CO_T, int SPWV, bool VECW, int WCO = 64>
__global__ __launch_bounds__(256) void conv3x3_kernel(const float* __restrict__ in, const float* __restrict__ wpack,
                                                      const float* __restrict__ bias, float* __restrict__ out,
                                                      const float* __restrict__ zero_page, int N, int Cin, int H, int W,
                                                      int Cout, SegGeom geo, int co_tiles) {
    constexpr int WAVES_CO = CO_T / WCO;                   // WCO = output channels per wave: 64 (two 32x32 accumulators) or 32
    constexpr int WAVES_PX = 4 / WAVES_CO;
    constexpr int TM = WCO / 32;
    constexpr int NSEG = WAVES_PX * SPWV;
    constexpr int RPW = NSEG * 6;                          // halo rows staged per wave (NSEG*24 rows / 4 waves): 12, 24 or 48
    constexpr int SST = RPW >= 24 ? RPW / 24 : 1;          // segments a wave stages rows for
    constexpr int ROWS = RPW >= 24 ? 24 : RPW;             // rows per staged segment
    constexpr int C4 = CO_T / 4;                           // float4 columns of the weight tile
    constexpr int RPP = 256 / C4;                          // weight rows per pass: 8 or 16
    constexpr int EA = (KC + RPP - 1) / RPP;               // 9 or 5 passes
    // Every staging store below is UNCONDITIONAL (weight rows padded to EA*RPP, halo lanes >= 34 land in a dummy tail of P).
    // With a store under `if (r < KC)` / `if (lane < PROW)`, LLVM's sink pass moved the global LOADS that only feed it into
    // that block, i.e. behind the end-of-chunk barrier: every chunk then waited out a full memory round trip with the matrix
    // pipe idle (a lone workgroup took 107 us per tile for 62 us of MFMA work).
    __shared__ __attribute__((aligned(16))) float Wt[EA * RPP * CO_T];
    __shared__ float P[NSEG * PSEG + 64];
    __shared__ SegInfo segs[NSEG];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lk = lane >> 5;
    // work item v = (pixel tile, output-channel tile), channel tile fastest; XCD x takes the x-th contiguous slice
    const int v = xcd_slice_order(blockIdx.x, gridDim.x);
    const int co0 = (v % co_tiles) * CO_T;
    const int seg0 = (v / co_tiles) * NSEG;
    const long HW = (long)H * W;
    if (tid < NSEG) segs[tid] = seg_decode(seg0 + tid, geo, H, (long)Cin * HW);
    __syncthreads();

    const int wco = (wave / WAVES_PX) * WCO;
    const int wsg = (wave % WAVES_PX) * SPWV;

    f32x16 acc[TM][SPWV];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < SPWV; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // ---- hoisted loader state
    const int wc4 = tid % C4, wr0 = tid / C4;
    const int wcol = co0 + wc4 * 4;
    const int wcol_c = VECW ? min(wcol, Cout - 4) : 0;      // clamped: loads are unconditional, results are selected
    const int Ktot = Cin * 9;
    f32x4 ra[EA];
    float rp[SST * ROWS];
    // this wave stages rows [wave*RPW, (wave+1)*RPW) of the NSEG*24 (segment, ci, kh) halo rows
    const int st_seg0 = (wave * RPW) / 24;
    const int st_ci0 = ((wave * RPW) % 24) / 3;
    // halo loads: channel base = wave-uniform pointer (scalar), position = one 32-bit lane offset per staged row;
    // out-of-range rows / columns / channels read a clamped in-range element and are zeroed by a 0/1 factor at the LDS
    // store, so a load is a single instruction and the loop carries almost no address arithmetic.
    int p_off[SST][3];
    float p_m[SST][3];
    const float* p_base[SST];
#pragma unroll
    for (int q = 0; q < SST; ++q) {
        const SegInfo sg = segs[st_seg0 + q];
        const int rr = lane / sg.pw, cc = lane - rr * sg.pw;          // patch column `lane` = sub-row rr, column cc
        const int ww = sg.w0 - 1 + cc;
        const bool colok = sg.valid && lane < PROW && rr < sg.rows && ww >= 0 && ww < W;
        const int loff = min(max(ww, 0), W - 1);
        p_base[q] = in + sg.base;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            const int hh = sg.h + rr + kh - 1;
            p_m[q][kh] = (colok && hh >= 0 && hh < H) ? 1.f : 0.f;
            p_off[q][kh] = min(max(hh, 0), H - 1) * W + loff;
        }
    }

    auto load_weights = [&](int ci0, int e) {
        const int r = wr0 + e * RPP;
        const int gk = ci0 * 9 + r;
        const bool rok = r < KC && gk < Ktot;
        const int gkc = min(gk, Ktot - 1);
        f32x4 v;
        if (VECW) {
            const f32x4* src = (const f32x4*)(wpack + (long)gkc * Cout + wcol_c);
            v = *((rok && wcol < Cout) ? src : (const f32x4*)zero_page);
        } else {
#pragma unroll
            for (int x = 0; x < 4; ++x) {
                const float* src = wpack + (long)gkc * Cout + min(wcol + x, Cout - 1);
                v[x] = *((rok && wcol + x < Cout) ? src : zero_page);
            }
        }
        ra[e] = v;
    };
    auto load_halo_row = [&](int ci0, int q, int j) {
        const int ci = st_ci0 + j / 3, kh = j % 3;
        const float* cb = p_base[q] + (long)min(ci0 + ci, Cin - 1) * HW;      // wave-uniform
        rp[q * ROWS + j] = cb[p_off[q][kh]];
    };
    // the loads of one chunk, cut into KC/2 = 36 slices that are issued between the k-steps of the previous chunk
    constexpr int NLOAD = SST * ROWS + EA;
    auto load_slice = [&](int ci0, int ks) {
#pragma unroll
        for (int i = 0; i < (NLOAD + KC / 2 - 1) / (KC / 2); ++i) {
            const int id = ks + i * (KC / 2);
            if (id < SST * ROWS) load_halo_row(ci0, id / ROWS, id % ROWS);
            else if (id < NLOAD) load_weights(ci0, id - SST * ROWS);
        }
    };
    auto load_chunk = [&](int ci0) {
#pragma unroll
        for (int ks = 0; ks < KC / 2; ++ks) load_slice(ci0, ks);
    };
    auto store_chunk = [&](int ci0) {
#pragma unroll
        for (int e = 0; e < EA; ++e) {
            const int r = wr0 + e * RPP;
            *(f32x4*)(Wt + r * CO_T + wc4 * 4) = ra[e];                                // rows >= KC: padding, never read
        }
#pragma unroll
        for (int q = 0; q < SST; ++q)
#pragma unroll
            for (int j = 0; j < ROWS; ++j) {
                const int ci = st_ci0 + j / 3, kh = j % 3;
                const float cm = (ci0 + ci) < Cin ? 1.f : 0.f;                         // wave-uniform
                const int dst = lane < PROW ? (st_seg0 + q) * PSEG + ci * PCI + kh * PROW + lane : NSEG * PSEG + lane;
                P[dst] = rp[q * ROWS + j] * (p_m[q][kh] * cm);
            }
    };

    const int nchunks = (Cin + CI_C - 1) / CI_C;
#ifdef VOCR_CONV_STAMPS
    unsigned long long st_begin, st0, st1, st2, st3, st4, sum_k = 0, sum_b1 = 0, sum_st = 0, sum_b2 = 0, st_pro;
    VOCR_STAMP(st_begin);
#endif
    load_chunk(0);
    store_chunk(0);
    __syncthreads();
#ifdef VOCR_CONV_STAMPS
    VOCR_STAMP(st_pro);
#endif
    const float* wa = Wt + wco + li + lk * (KC / 2) * CO_T;
    const float* pb = P + wsg * PSEG + li + lk * (CI_C / 2) * PCI;
    for (int c = 0; c < nchunks; ++c) {
        const int cnext = min(c + 1, nchunks - 1) * CI_C;     // branch-free prefetch (the last chunk re-loads itself, unused)
        // fragment reads run one k-step ahead of the MFMAs that consume them (hipcc otherwise emits
        // read -> lgkmcnt(0) -> MFMAs per step and exposes the LDS latency)
#ifdef VOCR_CONV_STAMPS
        VOCR_STAMP(st0);
#endif
        float a[TM], b[SPWV];
#pragma unroll
        for (int i = 0; i < TM; ++i) a[i] = wa[32 * i];
#pragma unroll
        for (int j = 0; j < SPWV; ++j) b[j] = pb[j * PSEG];
#pragma unroll
        for (int ks = 0; ks < KC / 2; ++ks) {
            load_slice(cnext, ks);                  // next chunk's global loads ride between this chunk's MFMAs
            float an[TM], bn[SPWV];
#pragma unroll
            for (int i = 0; i < TM; ++i) an[i] = 0.f;
#pragma unroll
            for (int j = 0; j < SPWV; ++j) bn[j] = 0.f;
            if (ks + 1 < KC / 2) {
                const int kn = ks + 1;
                const int offn = (kn / 9) * PCI + ((kn % 9) / 3) * PROW + (kn % 3);     // compile-time after unrolling
#pragma unroll
                for (int i = 0; i < TM; ++i) an[i] = wa[kn * CO_T + 32 * i];
#pragma unroll
                for (int j = 0; j < SPWV; ++j) bn[j] = pb[j * PSEG + offn];
            }
            __builtin_amdgcn_sched_barrier(0);      // next step's ds_reads are issued BEFORE this step's MFMAs
#pragma unroll
            for (int j = 0; j < SPWV; ++j)
#pragma unroll
                for (int i = 0; i < TM; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = an[i];
#pragma unroll
            for (int j = 0; j < SPWV; ++j) b[j] = bn[j];
        }
#ifdef VOCR_CONV_STAMPS
        VOCR_STAMP(st1);
#endif
        __syncthreads();
#ifdef VOCR_CONV_STAMPS
        VOCR_STAMP(st2);
#endif
        // unconditional (the last iteration stores its own re-loaded chunk, which nobody reads): under `if (c + 1 < nchunks)`
        // the compiler sank the chunk's last weight load into that block, behind the barrier
        store_chunk(cnext);
#ifdef VOCR_CONV_STAMPS
        VOCR_STAMP(st3);
#endif
        __syncthreads();
#ifdef VOCR_CONV_STAMPS
        VOCR_STAMP(st4);
        sum_k += st1 - st0; sum_b1 += st2 - st1; sum_st += st3 - st2; sum_b2 += st4 - st3;
#endif
    }
#ifdef VOCR_CONV_STAMPS
    unsigned long long st_loop_end;
    VOCR_STAMP(st_loop_end);
#endif

#pragma unroll
    for (int j = 0; j < SPWV; ++j) {
        const SegInfo s = segs[wsg + j];
        const int rr = li / s.pw, cc = li - rr * s.pw;                   // lane position -> (sub-row, column)
        if (!s.valid || rr >= s.rows || cc >= s.ow) continue;
        const long obase = (long)s.n * Cout * HW + (long)(s.h + rr) * W + s.w0 + cc;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co0 + wco + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
                if (co < Cout) out[obase + (long)co * HW] = acc[i][j][r] + (bias ? bias[co] : 0.f);
            }
    }
#ifdef VOCR_CONV_STAMPS
    unsigned long long st_end;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    VOCR_STAMP(st_end);
    if (tid == 0 && g_stamp_out && blockIdx.x < 4096) {
        unsigned long long* o = g_stamp_out + (size_t)blockIdx.x * 8;
        o[0] = st_end - st_begin; o[1] = sum_k; o[2] = sum_b1; o[3] = sum_st; o[4] = sum_b2; o[5] = st_pro - st_begin; o[6] = st_end - st_loop_end;
    }
#endif
}

// ---------------------------------------------------------------- last partial round of the forward / data-gradient kernel
// A launch of the kernel above takes ceil(workgroups / 256 CUs) rounds of equal length (scripts/conv_occ.py: a clean
// staircase), so the 32 workgroups left over from 2080 cost the 7x294 layers a ninth round with 7/8 of the chip idle
// (-8.7 % over the forward stack with the partial round simply dropped).  Those leftover workgroup tiles are computed here
// instead, cut into their 32-channel x 32-pixel MFMA tiles (8 or 16 per workgroup tile -> 256+ pieces = every CU busy for
// one short round).  One piece = one workgroup of 8 waves that split K: wave w and lane half lk take input channels
// 2*(8*s + w) + lk, s = 0, 1, ...; operands go global -> register -> MFMA directly (the weight pack row and the halo
// row are both 32 consecutive floats: two coalesced 128-byte reads per MFMA, all L2 hits), three steps of 9 taps in
// flight; the eight partial tiles are added through LDS in a fixed order (deterministic; the summation order differs
// from the main kernel's single accumulator, like any other tiling change).
// One piece, computed by a workgroup of TAIL_WAVES waves; red = TAIL_WAVES x 16 x 64 floats of LDS.  Called from the kernel of
// its own below (8 waves) and from conv3x3_dma_kernel (4 waves: there the pieces are the FIRST workgroups of the launch, so
// they run beside the first full tiles instead of in a round of their own after the last).
template <int TAIL_WAVES>
__device__ __forceinline__ void conv3x3_tail_piece(float* __restrict__ red_, int piece, const float* __restrict__ in,
                                                   const float* __restrict__ wpack, const float* __restrict__ bias, float* __restrict__ out,
                                                   const float* __restrict__ zero_page, int Cin, int H, int W, int Cout,
                                                   const SegGeom& geo, int co_tiles, int co_t, int nseg_wg, int first_tile) {
    const int cosub = co_t / 32, ppw = cosub * nseg_wg;
    const int v = first_tile + piece / ppw, sub = piece % ppw;
    const int co_base = (v % co_tiles) * co_t + (sub % cosub) * 32;
    const SegInfo sg = seg_decode((v / co_tiles) * nseg_wg + sub / cosub, geo, H, 0);
    conv3x3_tail_piece_at<TAIL_WAVES>(red_, sg, co_base, in, wpack, bias, out, zero_page, Cin, H, W, Cout);
}

__global__ __launch_bounds__(512) void conv3x3_tail_kernel(const float* __restrict__ in, const float* __restrict__ wpack,
                                                           const float* __restrict__ bias, float* __restrict__ out,
                                                           const float* __restrict__ zero_page, int Cin, int H, int W, int Cout,
                                                           SegGeom geo, int co_tiles, int co_t, int nseg_wg, int first_tile) {
    __shared__ float red[8 * 16 * 64];
    conv3x3_tail_piece<8>(red, blockIdx.x, in, wpack, bias, out, zero_page, Cin, H, W, Cout, geo, co_tiles, co_t, nseg_wg, first_tile);
}

// ---------------------------------------------------------------- forward / data gradient, LDS-DMA form
// Same implicit GEMM, same tiles and segments as conv3x3_kernel; what changes is how the operands reach LDS.  Phase stamps of
// conv3x3_kernel (scripts/conv_stamp.hip) showed a workgroup spending 14-18 % of its life in store_chunk - 37 KB of packed
// weights per 8-channel chunk going global -> VGPR -> ds_write_b128, the LDS write port being the limit - between two
// barriers with its matrix pipe idle.  Here the K loop runs over HALF chunks of 4 input channels (36 K rows), ping-pong
// between two LDS buffers of half the size (same LDS footprint, same occupancy):
//   * the weights of half-chunk h+1 are written into the idle buffer by LDS-DMA (global_load_lds_dwordx4: no VGPRs, no
//     ds_write instructions; the 36 x CO_T tile is 36 consecutive rows of wpack, one 1-KiB DMA = 2 or 4 rows) while the MFMAs
//     of half-chunk h read the other buffer;
//   * the halo rows of h+1 still pass through registers (they need the border masks) but are loaded at the top of
//     half-chunk h and stored at its end into the idle buffer: 3-24 ds_write_b32 per wave instead of a store phase;
//   * ONE barrier per half-chunk (__syncthreads drains the DMA: vmcnt(0)), none between store and use.
// K order inside a half-chunk: MFMA step ks multiplies k = ks (lanes 0-31) and k = ks + 18 (lanes 32-63), i.e. channels
// c and c + 2 of the four.  Cin is padded to a multiple of 8 with zero weights (rows past Cin*9 come from the zero page).
// Needs Cout % 4 == 0 and a 16-byte aligned weight pack (the DMA moves 16 bytes per lane).
constexpr int CI_H = 4;
constexpr int KH_ROWS = CI_H * 9;      // 36
constexpr int PSEGH = CI_H * PCI;      // 408 floats of halo patch per segment and half-chunk

template <int CO_T, int SPWV, int WCO>
__global__ __launch_bounds__(256) void conv3x3_dma_kernel(const float* __restrict__ in, const float* __restrict__ wpack,
                                                          const float* __restrict__ bias, float* __restrict__ out,
                                                          const float* __restrict__ zero_page, int N, int Cin, int H, int W,
                                                          int Cout, SegGeom geo, int co_tiles, int n_tail, int first_tail_tile) {
    constexpr int WAVES_CO = CO_T / WCO;
    constexpr int WAVES_PX = 4 / WAVES_CO;
    constexpr int TM = WCO / 32;
    constexpr int NSEG = WAVES_PX * SPWV;
    constexpr int WBUF = KH_ROWS * CO_T;                     // floats per weight buffer
    constexpr int PBUF = NSEG * PSEGH;                       // floats per halo buffer
    constexpr int ROWS_W = NSEG * 3;                         // halo rows a wave stages per half-chunk (NSEG*12 rows / 4 waves)
    constexpr int SSTH = ROWS_W >= 12 ? ROWS_W / 12 : 1;     // segments a wave stages rows for
    constexpr int ROWSH = ROWS_W >= 12 ? 12 : ROWS_W;        // rows per staged segment
    constexpr int LPR = CO_T / 4;                            // lanes per weight row in one DMA (32 or 16)
    constexpr int RPI = 64 / LPR;                            // weight rows per DMA instruction (2 or 4)
    constexpr int NDMA = KH_ROWS / RPI;                      // DMA instructions per half-chunk (18 or 9)
    constexpr int DPW = (NDMA + 3) / 4;                      // ... per wave
    // ONE LDS object (a second one beside an LDS-DMA target makes hipcc drain vmcnt before every ds_read):
    // Wt[2][36][CO_T] | P[2][NSEG][4][3][34] | 64 dummy floats (halo lanes >= 34) | NSEG x 8 ints of segment geometry
    __shared__ __attribute__((aligned(16))) float lds[2 * WBUF + 2 * PBUF + 64 + NSEG * 8];
    float* const Wt = lds;
    float* const P = lds + 2 * WBUF;
    constexpr int DUMMY = 2 * PBUF;                          // offset into P of the dummy tail
    int* const segw = (int*)(lds + 2 * WBUF + 2 * PBUF + 64);

    if ((int)blockIdx.x < n_tail) {                          // pieces of the last partial round of tiles (conv3x3_tail_piece)
        conv3x3_tail_piece<4>(lds, blockIdx.x, in, wpack, bias, out, zero_page, Cin, H, W, Cout, geo, co_tiles, CO_T, NSEG, first_tail_tile);
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
        const SegInfo s = seg_decode(seg0 + tid, geo, H, 0);
        int* o = segw + tid * 8;
        o[0] = s.n; o[1] = s.h; o[2] = s.w0; o[3] = s.valid; o[4] = s.rows; o[5] = s.pw; o[6] = s.ow;
    }
    __syncthreads();

    const int wco = (wave / WAVES_PX) * WCO;
    const int wsg = (wave % WAVES_PX) * SPWV;
    f32x16 acc[TM][SPWV];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < SPWV; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // ---- halo loader state: this wave stages rows [wave*ROWS_W, (wave+1)*ROWS_W) of the (segment, ci, kh) rows
    const int st_seg0 = (wave * ROWS_W) / 12;
    const int st_ci0 = ((wave * ROWS_W) % 12) / 3;
    int p_off[SSTH][3];
    float p_m[SSTH][3];
    const float* p_base[SSTH];
#pragma unroll
    for (int q = 0; q < SSTH; ++q) {
        const int* sg = segw + (st_seg0 + q) * 8;
        const int pw = sg[5];
        const int rr = lane / pw, cc = lane - rr * pw;
        const int ww = sg[2] - 1 + cc;
        const bool colok = sg[3] && lane < PROW && rr < sg[4] && ww >= 0 && ww < W;
        const int loff = min(max(ww, 0), W - 1);
        p_base[q] = in + (long)sg[0] * Cin * HW;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            const int hh = sg[1] + rr + kh - 1;
            p_m[q][kh] = (colok && hh >= 0 && hh < H) ? 1.f : 0.f;
            p_off[q][kh] = min(max(hh, 0), H - 1) * W + loff;
        }
    }
    float rp[SSTH * ROWSH];
    auto load_patch = [&](int ci0) {
#pragma unroll
        for (int q = 0; q < SSTH; ++q)
#pragma unroll
            for (int j = 0; j < ROWSH; ++j) {
                const int ci = st_ci0 + j / 3, kh = j % 3;
                const float* cb = p_base[q] + (long)min(ci0 + ci, Cin - 1) * HW;      // wave-uniform
                rp[q * ROWSH + j] = cb[p_off[q][kh]];
            }
    };
    auto store_patch = [&](int ci0, int buf) {
#pragma unroll
        for (int q = 0; q < SSTH; ++q)
#pragma unroll
            for (int j = 0; j < ROWSH; ++j) {
                const int ci = st_ci0 + j / 3, kh = j % 3;
                const float cm = (ci0 + ci) < Cin ? 1.f : 0.f;                         // wave-uniform
                const int dst = lane < PROW ? buf * PBUF + (st_seg0 + q) * PSEGH + ci * PCI + kh * PROW + lane : DUMMY + lane;
                P[dst] = rp[q * ROWSH + j] * (p_m[q][kh] * cm);
            }
    };
    // ---- weight DMA: instruction q of a half-chunk moves rows [q*RPI, (q+1)*RPI) x CO_T floats = 1 KiB
    const int Ktot = Cin * 9;
    const int drow = lane / LPR, dcol = (lane % LPR) * 4;
    const bool dcol_ok = co0 + dcol < Cout;
    auto dma_weights = [&](int ci0, int buf) {
#pragma unroll
        for (int d = 0; d < DPW; ++d) {
            const int q = wave + 4 * d;                                               // wave-uniform
            if (q < NDMA) {
                const int gk = ci0 * 9 + q * RPI + drow;
                const float* src = (gk < Ktot && dcol_ok) ? wpack + (long)gk * Cout + co0 + dcol : zero_page;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)(Wt + buf * WBUF + q * 256), 16, 0, 0);
            }
        }
    };
    auto kloop = [&](int buf) {
        const float* wa = Wt + buf * WBUF + wco + li + lk * (KH_ROWS / 2) * CO_T;
        const float* pb = P + buf * PBUF + wsg * PSEGH + li + lk * (CI_H / 2) * PCI;
        float a[TM], b[SPWV];
#pragma unroll
        for (int i = 0; i < TM; ++i) a[i] = wa[32 * i];
#pragma unroll
        for (int j = 0; j < SPWV; ++j) b[j] = pb[j * PSEGH];
#pragma unroll
        for (int ks = 0; ks < KH_ROWS / 2; ++ks) {
            float an[TM], bn[SPWV];
#pragma unroll
            for (int i = 0; i < TM; ++i) an[i] = 0.f;
#pragma unroll
            for (int j = 0; j < SPWV; ++j) bn[j] = 0.f;
            if (ks + 1 < KH_ROWS / 2) {
                const int kn = ks + 1;
                const int offn = (kn / 9) * PCI + ((kn % 9) / 3) * PROW + (kn % 3);     // compile-time after unrolling
#pragma unroll
                for (int i = 0; i < TM; ++i) an[i] = wa[kn * CO_T + 32 * i];
#pragma unroll
                for (int j = 0; j < SPWV; ++j) bn[j] = pb[j * PSEGH + offn];
            }
            __builtin_amdgcn_sched_barrier(0);      // next step's ds_reads are issued BEFORE this step's MFMAs
#pragma unroll
            for (int j = 0; j < SPWV; ++j)
#pragma unroll
                for (int i = 0; i < TM; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = an[i];
#pragma unroll
            for (int j = 0; j < SPWV; ++j) b[j] = bn[j];
        }
    };

    const int nh = ((Cin + 2 * CI_H - 1) / (2 * CI_H)) * 2;          // half-chunks, padded to an even count (zero weights)
#ifdef VOCR_CONV_STAMPS
    unsigned long long st_begin, t0, t1, t2, t3, t4, sum_bar = 0, sum_issue = 0, sum_k = 0, sum_st = 0, st_pro;
    VOCR_STAMP(st_begin);
#define VOCR_PHASE(tA, tB, acc_) do { VOCR_STAMP(tB); acc_ += tB - tA; } while (0)
#else
#define VOCR_PHASE(tA, tB, acc_) do { } while (0)
#endif
    dma_weights(0, 0);
    load_patch(0);
    store_patch(0, 0);
#ifdef VOCR_CONV_STAMPS
    VOCR_STAMP(st_pro);
    t4 = st_pro;
#endif
    for (int h = 0; h < nh; h += 2) {
        __syncthreads();                            // buffer 0 complete (DMA drained: vmcnt(0)), buffer 1 free
        VOCR_PHASE(t4, t0, sum_bar);
        // the next half-chunk's DMAs and halo loads are issued here in one go: spread between the k-steps they measured
        // slower (733 vs 707 us on the 256->256 layer: an LDS-DMA instruction holds the in-order wave for 100-450 cycles)
        dma_weights((h + 1) * CI_H, 1);
        load_patch((h + 1) * CI_H);
        VOCR_PHASE(t0, t1, sum_issue);
        kloop(0);
        VOCR_PHASE(t1, t2, sum_k);
        store_patch((h + 1) * CI_H, 1);
        VOCR_PHASE(t2, t3, sum_st);
        __syncthreads();                            // buffer 1 complete, buffer 0 free
        VOCR_PHASE(t3, t0, sum_bar);
        dma_weights((h + 2) * CI_H, 0);             // past the last channel: zero page / masked rows, never used
        load_patch((h + 2) * CI_H);
        VOCR_PHASE(t0, t1, sum_issue);
        kloop(1);
        VOCR_PHASE(t1, t2, sum_k);
        store_patch((h + 2) * CI_H, 0);
        VOCR_PHASE(t2, t4, sum_st);
    }
#ifdef VOCR_CONV_STAMPS
    unsigned long long st_loop_end;
    VOCR_STAMP(st_loop_end);
#endif

    // bias of this lane's 16*TM output channels, gathered with all loads in flight (a `bias ? bias[co] : 0` inside the store
    // loop became 32 load -> wait -> store round trips)
    float bv[TM][16];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) bv[i][r] = 0.f;
    if (bias) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) bv[i][r] = bias[min(co0 + wco + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk, Cout - 1)];
    }
#pragma unroll
    for (int j = 0; j < SPWV; ++j) {
        const int* sg = segw + (wsg + j) * 8;
        const int pw = sg[5];
        const int rr = li / pw, cc = li - rr * pw;
        if (!sg[3] || rr >= sg[4] || cc >= sg[6]) continue;
        const long obase = (long)sg[0] * Cout * HW + (long)(sg[1] + rr) * W + sg[2] + cc;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co0 + wco + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
                if (co < Cout) out[obase + (long)co * HW] = acc[i][j][r] + bv[i][r];
            }
    }
#ifdef VOCR_CONV_STAMPS
    unsigned long long st_end;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    VOCR_STAMP(st_end);
    if (tid == 0 && g_stamp_out && blockIdx.x < 4096) {
        unsigned long long* o = g_stamp_out + (size_t)blockIdx.x * 8;
        o[0] = st_end - st_begin; o[1] = sum_k; o[2] = sum_bar; o[3] = sum_st; o[4] = sum_issue; o[5] = st_pro - st_begin; o[6] = st_end - st_loop_end;
    }
#endif
}

// ---------------------------------------------------------------- weight gradient
constexpr int WG_DYP = SEGW + 1;    // 33: odd pitch -> conflict-free reads across channels
constexpr int WG_XCI = 3 * PROW + 1;  // 103

#ifndef VOCR_WGRAD_XCD
#define VOCR_WGRAD_XCD 1
#endif
// TM = 32-channel dy fragments per wave: the workgroup tile is 64*TM output channels x 64 input channels x 9 taps; with TM = 2 a
// halo fragment feeds two MFMAs and the x rows are staged once for 128 output channels (18 accumulators = 288 registers: one
// wave per SIMD only, i.e. MODE 0 / 1).
template <int MODE, int TM = 1>        // MODE 0: one LDS buffer, 1: two buffers, 2: two buffers + four loader waves (512 threads)
__global__ __launch_bounds__(MODE == 2 ? 512 : 256) void conv3x3_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                            float* __restrict__ slab,
                                                            const float* __restrict__ zero_page, int N, int Cin, int H,
                                                            int W, int Cout, SegGeom geo, int segs_per_split) {
    // MODE 1: two LDS buffers; the 56 loads of segment g+1 are issued in ONE burst in front of the k-steps of segment g and
    // stored into the idle buffer behind them (one barrier per segment).  Measured SLOWER than MODE 0 (769 vs 723 us on the
    // 256->256 layer): what a segment pays outside its 144 MFMAs is not load latency but VMEM ISSUE - a CU takes one wave
    // load per ~18 cycles once its queue is full, 224 dword loads per segment - and an in-order wave cannot issue MFMAs
    // while it is stuck there, wherever the burst is placed.
    // MODE 2: the issue moves to four LOADER waves (one per SIMD, waves 4-7): they fetch and stage segment g+1 while the
    // four MFMA waves run nothing but fragment reads and MFMAs on segment g; one barrier per segment joins them.
    constexpr bool DB = MODE >= 1, PC = MODE == 2;
    constexpr int NB = DB ? 2 : 1;
    constexpr int COT = 64 * TM;                        // output channels per workgroup
    constexpr int XPB = 64 * WG_XCI + 64;               // + dummy tail for lanes >= 34 (unconditional stores, see conv3x3_kernel)
    __shared__ float dyT[NB * COT * WG_DYP];
    __shared__ float xp[NB * XPB];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave_all = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool loader = !PC || wave_all >= 4, mfma_wave = !PC || wave_all < 4;
    const int wave = wave_all & 3;                      // role-local wave index
    const int li = lane & 31, lk = lane >> 5;
    // Workgroup -> (channel tile, split): the 16 (ci, co) tiles of one split read the same pixels of x and dy, so they are
    // placed on ONE XCD (linear id b runs on XCD b % 8): a line then comes from HBM / Infinity Cache once and from that
    // XCD's L2 for the other tiles.  (The loads of a segment are bound by line fetches: ~64 outstanding per CU x latency.)
    int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    if (VOCR_WGRAD_XCD && (gridDim.z & 7) == 0) {
        const int nxy = gridDim.x * gridDim.y;
        const int b = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
        const int k = b & 7, slot = b >> 3;
        bz = k + 8 * (slot / nxy);
        const int xy = slot - (slot / nxy) * nxy;
        bx = xy % gridDim.x;
        by = xy / gridDim.x;
    }
    static_assert(TM == 1 || MODE != 2, "18 accumulators do not fit two waves per SIMD");
    const int ci0 = bx * 64, co0 = by * COT, split = bz;
    const int wco = (wave >> 1) * 32 * TM, wci = (wave & 1) * 32;
    const long HW = (long)H * W;

    constexpr int EDY = COT * SEGW / 256;              // 8 per 64 channels
    float rdy[EDY], rx[48];

    const int nseg_total = geo.nseg;
    const int sbeg = split * segs_per_split;
    const int send = min(nseg_total, sbeg + segs_per_split);
    const int dpx = tid & 31, dco = (tid & 255) >> 5;   // dy loader: 32 lane positions x 8 channels per pass

    // Loader: 32-bit element offsets from the tensor bases (host guarantees < 2^31 elements); everything that does not
    // depend on the segment is hoisted; the per-segment part is three scalars.  The 56 loads of segment g+1 are issued in
    // 16 slices BETWEEN the k-steps of segment g (one basic block, no branch), so their address arithmetic shares issue
    // slots with the MFMAs instead of running ahead of them (it used to cost ~30 % of the loop: 370 VALU + 600 SALU
    // before the first MFMA).
    // channel bases are wave-uniform pointers (scalar registers), the per-segment position is one 32-bit lane offset
    // per row: a load is ONE instruction (saddr + voffset).  Out-of-range rows/columns/channels load a clamped in-range
    // element and are zeroed by a 0/1 factor when they are written to LDS.
    const float* dy_cb[EDY];
    float dy_m[EDY];
#pragma unroll
    for (int e = 0; e < EDY; ++e) {
        const int co = co0 + dco + 8 * e;
        dy_m[e] = co < Cout ? 1.f : 0.f;
        dy_cb[e] = dy + (long)min(co, Cout - 1) * HW;
    }
    const float* x_cb[16];
    float x_m[16];
#pragma unroll
    for (int ci = 0; ci < 16; ++ci) {
        const int gci = ci0 + wave * 16 + ci;
        x_m[ci] = gci < Cin ? 1.f : 0.f;
        x_cb[ci] = x + (long)min(gci, Cin - 1) * HW;
    }
    struct SegPos { int dy_off; int x_off[3]; float dy_ok; float row_ok[3]; };
    auto seg_pos = [&](int g) {
        SegPos p;
#ifdef VOCR_WGRAD_SAMESEG       // diagnostic: every segment reads the same lines (is the loader bound by line fetches or by issue?)
        g = sbeg;
#endif
        const SegInfo sg = seg_decode(g, geo, H, 0);
        const int n = sg.n, h = sg.h, w0 = sg.w0;
        // dy sits at its lane POSITION (sub-row dr, column dc of the segment; gaps between sub-rows hold zeros), so that
        // k-index q of the MFMA pairs dy[q] with patch[q + kw] exactly as in a full segment
        const int dr = dpx / sg.pw, dc = dpx - dr * sg.pw;
        const bool dok = dr < sg.rows && dc < sg.ow;
        p.dy_ok = dok ? 1.f : 0.f;
        p.dy_off = n * Cout * (int)HW + min(h + dr, H - 1) * W + min(w0 + dc, W - 1);
        const int rr = lane / sg.pw, cc = lane - rr * sg.pw;
        const int ww = w0 - 1 + cc;
        const bool colok = lane < PROW && rr < sg.rows && ww >= 0 && ww < W;
        const int loff = min(max(ww, 0), W - 1);
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            const int hh = h + rr + kh - 1;
            p.row_ok[kh] = (colok && hh >= 0 && hh < H) ? 1.f : 0.f;
            p.x_off[kh] = n * Cin * (int)HW + min(max(hh, 0), H - 1) * W + loff;
        }
        return p;
    };
    // slice q (0..15) of the loads of one segment: dy pass q/2 (even q) and x rows [3q, 3q+3)
    auto load_slice_into = [&](const SegPos& p, int q, float* rdy, float* rx) {
        if ((q & 1) == 0) rdy[q >> 1] = dy_cb[q >> 1][p.dy_off];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int r = 3 * q + j;
            rx[r] = x_cb[r / 3][p.x_off[r % 3]];
        }
    };
    auto load_slice = [&](const SegPos& p, int q) { load_slice_into(p, q, rdy, rx); };
    auto store_seg_from = [&](const SegPos& p, int buf, const float* rdy, const float* rx) {
        float* const dyTb = dyT + buf * (COT * WG_DYP);
        float* const xpb = xp + buf * XPB;
#pragma unroll
        for (int e = 0; e < EDY; ++e) dyTb[(dco + 8 * e) * WG_DYP + dpx] = rdy[e] * (dy_m[e] * p.dy_ok);
        if (DB) {               // unconditional stores (lanes >= 34 hit the dummy tail): the loads stay where they were issued
#pragma unroll
            for (int ci = 0; ci < 16; ++ci)
#pragma unroll
                for (int kh = 0; kh < 3; ++kh)
                    xpb[lane < PROW ? (wave * 16 + ci) * WG_XCI + kh * PROW + lane : 64 * WG_XCI + lane] = rx[ci * 3 + kh] * (x_m[ci] * p.row_ok[kh]);
            return;
        }
        // NOTE: with the stores under `if (lane < PROW)` LLVM sinks the 48 x loads of load_slice into this block, i.e. they
        // are issued in one burst behind the end-of-segment barrier.  That is the FASTER arrangement here (725 vs 780 us on
        // the 256->256 layer with the loads kept between the k-steps): this kernel runs one wave per SIMD, so every VMEM
        // issue slot inside the MFMA loop is a bubble nobody fills, while the burst's latency is paid once per segment.
        if (lane < PROW) {
#pragma unroll
            for (int ci = 0; ci < 16; ++ci)
#pragma unroll
                for (int kh = 0; kh < 3; ++kh)
                    xpb[(wave * 16 + ci) * WG_XCI + kh * PROW + lane] = rx[ci * 3 + kh] * (x_m[ci] * p.row_ok[kh]);
        }
    };
    auto store_seg = [&](const SegPos& p, int buf) { store_seg_from(p, buf, rdy, rx); };

    if (PC && !mfma_wave) {
        // Loader role (waves 4-7), kept apart from the MFMA role's code so that the accumulators are not live here.
        // What was measured on the way (s_memtime stamps, scripts/wgrad_stamp.hip; 256->256 layer; wait of the MFMA waves per
        // segment beside 8900 cycles of MFMAs): 56 dword loads per loader wave 3450; 34 dword loads (this form) 2550; 10
        // 16-byte loads 3900; 34 loads that all hit one L1 line 1530; 34 LDS READS instead of the loads 2800; the whole
        // address arithmetic and all LDS stores but no loads 70; scalar-only addressing (buffer loads, no VALU) 3700;
        // all 16 tiles of a split on one XCD -1 %; LDS-DMA gather (global_load_lds_dword, 35 per wave) 10800.  So neither
        // instruction count, cache, VALU nor LDS-store issue is the limit: beside a wave that keeps the matrix pipe full,
        // data RETURNING to the partner wave (VMEM or LDS, into VGPRs or as DMA) arrives about once per 300 cycles per
        // SIMD; the loader is bound by the number of registers it has to receive (34 x 300 > 8900), the MFMA wave's
        // own fragment reads are not.  What helps is fewer returned registers: a halo row is 34 floats = 34 of 64 lanes,
        // so two rows' first 32 columns share one load (lanes 0-31: channel c, lanes 32-63: channel c + 8) and the two
        // right-hand halo columns of all 48 rows take two more: 26 instead of 48 x loads.
        // Software pipeline over two register sets: the loads of segment g+2 are in flight while segment g+1 is written
        // to LDS and while this wave waits at the barrier.
        const int nsegs = send - sbeg;
        const int half = lane >> 5, q32 = lane & 31;
        int cbv[8];
        float xmv[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const int gci = ci0 + wave * 16 + c + 8 * half;
            cbv[c] = min(gci, Cin - 1) * (int)HW;
            xmv[c] = gci < Cin ? 1.f : 0.f;
        }
        struct LPos { int r0, r1, r2, b0, h0, h1; float k0, k1, k2, dk, hk0, hk1; };
        auto lpos = [&](int g) {
            LPos p;
#ifdef VOCR_WGRAD_SAMESEG
            g = sbeg;
#endif
            const SegInfo sg = seg_decode(g, geo, H, 0);
            const int n = sg.n, h = sg.h, w0 = sg.w0;
            const int img = n * Cin * (int)HW;
            const int dr = dpx / sg.pw, dc = dpx - dr * sg.pw;
            p.dk = (dr < sg.rows && dc < sg.ow) ? 1.f : 0.f;
            p.b0 = n * Cout * (int)HW + min(h + dr, H - 1) * W + min(w0 + dc, W - 1);
            const int rr = q32 / sg.pw, cc = q32 - rr * sg.pw, ww = w0 - 1 + cc;
            const bool colok = rr < sg.rows && ww >= 0 && ww < W;
            const int loff = min(max(ww, 0), W - 1);
            int ro[3], ho[2];
            float rk[3], hk[2];
#pragma unroll
            for (int kh = 0; kh < 3; ++kh) {
                const int hh = h + rr + kh - 1;
                rk[kh] = (colok && hh >= 0 && hh < H) ? 1.f : 0.f;
                ro[kh] = img + min(max(hh, 0), H - 1) * W + loff;
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int t = lane + 64 * j, row = min(t >> 1, 47), ci = row / 3, kh = row - 3 * ci, qq = 32 + (t & 1);
                const int rr2 = qq / sg.pw, cc2 = qq - rr2 * sg.pw, ww2 = w0 - 1 + cc2, hh = h + rr2 + kh - 1;
                const int gci = ci0 + wave * 16 + ci;
                hk[j] = (t < 96 && gci < Cin && rr2 < sg.rows && ww2 >= 0 && ww2 < W && hh >= 0 && hh < H) ? 1.f : 0.f;
                ho[j] = img + min(gci, Cin - 1) * (int)HW + min(max(hh, 0), H - 1) * W + min(max(ww2, 0), W - 1);
            }
            p.r0 = ro[0]; p.r1 = ro[1]; p.r2 = ro[2]; p.k0 = rk[0]; p.k1 = rk[1]; p.k2 = rk[2];
            p.h0 = ho[0]; p.h1 = ho[1]; p.hk0 = hk[0]; p.hk1 = hk[1];
            return p;
        };
        // register set layout (34 floats): [0,8) dy, [8,32) x rows (r = 3*c + kh), [32,34) right-hand halo
        auto lload = [&](const LPos& p, float* rs) {
#pragma unroll
            for (int e = 0; e < EDY; ++e) rs[e] = dy_cb[e][p.b0];
#pragma unroll
            for (int r = 0; r < 24; ++r) rs[8 + r] = x[(r % 3 == 0 ? p.r0 : r % 3 == 1 ? p.r1 : p.r2) + cbv[r / 3]];
            rs[32] = x[p.h0];
            rs[33] = x[p.h1];
        };
        auto lstore = [&](const LPos& p, int buf, const float* rs) {
            float* const dyTb = dyT + buf * (COT * WG_DYP);
            float* const xpb = xp + buf * XPB;
#pragma unroll
            for (int e = 0; e < EDY; ++e) dyTb[(dco + 8 * e) * WG_DYP + dpx] = rs[e] * (dy_m[e] * p.dk);
#pragma unroll
            for (int r = 0; r < 24; ++r)
                xpb[(wave * 16 + r / 3 + 8 * half) * WG_XCI + (r % 3) * PROW + q32] = rs[8 + r] * (xmv[r / 3] * (r % 3 == 0 ? p.k0 : r % 3 == 1 ? p.k1 : p.k2));
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int t = lane + 64 * j, row = min(t >> 1, 47), ci = row / 3, kh = row - 3 * ci;
                xpb[t < 96 ? (wave * 16 + ci) * WG_XCI + kh * PROW + 32 + (t & 1) : 64 * WG_XCI + lane] = rs[32 + j] * (j == 0 ? p.hk0 : p.hk1);
            }
        };
        float rsA[34], rsB[34];
        LPos pa = lpos(sbeg), pb = lpos(min(sbeg + 1, send - 1));
        if (nsegs > 0) {
            lload(pa, rsA);
            lload(pb, rsB);
            lstore(pa, 0, rsA);
        }
        __syncthreads();
        // invariant at the top of iteration i (segment sbeg+i being multiplied from buffer i&1): set B holds segment i+1
        for (int i = 0; i < nsegs; i += 2) {
            pa = lpos(min(sbeg + i + 2, send - 1));
            lload(pa, rsA);                                                  // segment i+2 -> set A
            if (i + 1 < nsegs) lstore(pb, 1, rsB);                          // segment i+1 -> buffer 1
            __syncthreads();
            if (i + 1 >= nsegs) break;
            pb = lpos(min(sbeg + i + 3, send - 1));
            lload(pb, rsB);                                                  // segment i+3 -> set B
            if (i + 2 < nsegs) lstore(pa, 0, rsA);                          // segment i+2 -> buffer 0
            __syncthreads();
        }
        return;
    }
    f32x16 acc[TM][9];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][t][r] = 0.f;

    if (sbeg < send && loader) {
        const SegPos p0 = seg_pos(sbeg);
#pragma unroll
        for (int q = 0; q < 16; ++q) load_slice(p0, q);
        store_seg(p0, 0);
    }
    __syncthreads();
#ifdef VOCR_CONV_STAMPS
    unsigned long long m0, m1, m2, sm_k = 0, sm_bar = 0;
#endif
    for (int g = sbeg; g < send; ++g) {
#ifdef VOCR_CONV_STAMPS
        VOCR_STAMP(m0);
#endif
        const int cur = DB ? ((g - sbeg) & 1) : 0;
        const SegPos pn = seg_pos(min(g + 1, send - 1));      // branch-free: the last iteration re-loads its own segment (unused)
        const float* ap = dyT + cur * (COT * WG_DYP) + (wco + li) * WG_DYP + lk;
        const float* bp = xp + cur * XPB + (wci + li) * WG_XCI + lk;
        if (DB && !PC) {
#pragma unroll
            for (int q = 0; q < 16; ++q) load_slice(pn, q);
            __builtin_amdgcn_sched_barrier(0);
        }
        // fragment reads run one k-step ahead of their MFMAs (one wave per SIMD: nothing else hides the LDS latency)
        float a[TM], b[9];
#pragma unroll
        for (int i = 0; i < TM; ++i) a[i] = ap[32 * i * WG_DYP];
#pragma unroll
        for (int t = 0; t < 9; ++t) b[t] = bp[(t / 3) * PROW + (t % 3)];
#pragma unroll
        for (int ks = 0; ks < SEGW / 2; ++ks) {
            if (!DB) load_slice(pn, ks);
            float an[TM], bn[9];
#pragma unroll
            for (int i = 0; i < TM; ++i) an[i] = 0.f;
#pragma unroll
            for (int t = 0; t < 9; ++t) bn[t] = 0.f;
            if (ks + 1 < SEGW / 2) {
#pragma unroll
                for (int i = 0; i < TM; ++i) an[i] = ap[32 * i * WG_DYP + 2 * (ks + 1)];
#pragma unroll
                for (int t = 0; t < 9; ++t) bn[t] = bp[(t / 3) * PROW + (t % 3) + 2 * (ks + 1)];
            }
            __builtin_amdgcn_sched_barrier(0);      // loads of the next segment and next step's ds_reads stay above the MFMAs
#pragma unroll
            for (int t = 0; t < 9; ++t)
#pragma unroll
                for (int i = 0; i < TM; ++i) acc[i][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[t], acc[i][t], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = an[i];
#pragma unroll
            for (int t = 0; t < 9; ++t) b[t] = bn[t];
        }
#ifdef VOCR_CONV_STAMPS
        VOCR_STAMP(m1);
#endif
        if (!DB) __syncthreads();
        if (!PC) store_seg(pn, cur ^ (DB ? 1 : 0));
        __syncthreads();
#ifdef VOCR_CONV_STAMPS
        VOCR_STAMP(m2);
        sm_k += m1 - m0; sm_bar += m2 - m1;
#endif
    }
#ifdef VOCR_CONV_STAMPS
    if (tid == 0 && g_stamp_out && blockIdx.x == 0 && blockIdx.y == 0) {
        unsigned long long* o = g_stamp_out + (size_t)blockIdx.z * 8;
        o[0] = sm_k; o[1] = sm_bar; o[2] = send - sbeg;
    }
#endif
    // slab[split][tap][co][ci]  (ci contiguous -> coalesced stores)
    const long plane = (long)Cout * Cin;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co0 + wco + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * lk;
                const int ci = ci0 + wci + li;
                if (co < Cout && ci < Cin) slab[((long)split * 9 + t) * plane + (long)co * Cin + ci] = acc[i][t][r];
            }
}

// Weight gradient for Cin <= 3 (the first layer: grey or RGB lines).  The generic kernel would spend a 32-wide
// MFMA N-tile on 1-3 input channels; here N = (ci, tap) <= 27 columns, so all nine taps of all channels share ONE
// accumulator per wave and the kernel is bound by streaming dy once.  2 waves = 64 output channels.
__global__ __launch_bounds__(128) void conv3x3_wgrad_smallcin_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                                     float* __restrict__ slab, const float* __restrict__ zero_page,
                                                                     int N, int Cin, int H, int W, int Cout, int SW,
                                                                     int nseg_total, int segs_per_split) {
    __shared__ float dyT[64 * WG_DYP];
    __shared__ float xp[3 * 3 * PROW + 2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lk = lane >> 5;
    const int co0 = blockIdx.y * 64, split = blockIdx.z;
    const long HW = (long)H * W;
    const int jci = li / 9, jtap = li % 9;
    const bool jok = li < Cin * 9;
    const int joff = jok ? jci * 3 * PROW + (jtap / 3) * PROW + (jtap % 3) : 0;

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float rdy[16], rx[3];
    const int sbeg = split * segs_per_split;
    const int send = min(nseg_total, sbeg + segs_per_split);
    const int dpx = tid & 31, dco = tid >> 5;           // 32 pixels x 4 channels per pass, 16 passes

    auto load_seg = [&](int g) {
        const int n = g / (H * SW), rem = g % (H * SW);
        const int h = rem / SW, w0 = (rem % SW) * SEGW;
        const float* dyb = dy + (long)n * Cout * HW + (long)h * W;
        const bool pxok = w0 + dpx < W;
        const int pxc = min(w0 + dpx, W - 1);
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int co = co0 + dco + 4 * e;
            const float* src = dyb + ((long)min(co, Cout - 1) * HW + pxc);
            rdy[e] = *((pxok && co < Cout) ? src : zero_page + lane);
        }
        // halo rows: 3*Cin*3 <= 27 rows of 34 columns; thread t < 102 loads column t%34 of row kh = t/34 for each channel
        const int kh = tid / PROW, col = tid % PROW;
        const int hh = h + kh - 1, ww = w0 - 1 + col;
        const bool ok = tid < 3 * PROW && hh >= 0 && hh < H && ww >= 0 && ww < W;
        const float* xb = x + (long)n * Cin * HW + (long)min(max(hh, 0), H - 1) * W + min(max(ww, 0), W - 1);
#pragma unroll
        for (int ci = 0; ci < 3; ++ci) rx[ci] = *((ok && ci < Cin) ? xb + (long)min(ci, Cin - 1) * HW : zero_page + lane);
    };
    auto store_seg = [&]() {
#pragma unroll
        for (int e = 0; e < 16; ++e) dyT[(dco + 4 * e) * WG_DYP + dpx] = rdy[e];
        if (tid < 3 * PROW) {
#pragma unroll
            for (int ci = 0; ci < 3; ++ci) xp[ci * 3 * PROW + tid] = rx[ci];
        }
    };
    if (sbeg < send) {
        load_seg(sbeg);
        store_seg();
    }
    __syncthreads();
    for (int g = sbeg; g < send; ++g) {
        if (g + 1 < send) load_seg(g + 1);
        const float* ap = dyT + (wave * 32 + li) * WG_DYP + lk;
        const float* bp = xp + joff + lk;
#pragma unroll
        for (int ks = 0; ks < SEGW / 2; ++ks) {
            const float b = jok ? bp[2 * ks] : 0.f;
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[2 * ks], b, acc, 0, 0, 0);
        }
        __syncthreads();
        if (g + 1 < send) {
            store_seg();
            __syncthreads();
        }
    }
    // slab[split][tap][co][ci]
    const long plane = (long)Cout * Cin;
    if (jok) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = co0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
            if (co < Cout) slab[((long)split * 9 + jtap) * plane + (long)co * Cin + jci] = acc[r];
        }
    }
}

// Weight gradient with fp16 operands on v_mfma_f32_32x32x16_f16 (config 5), fp32 accumulation: M = co, N = ci, K = pixels.
// One 32-pixel segment = 2 MFMA k-steps per tap; lane-half h owns pixels 8h..8h+7 of a k-step, so both fragments are
// 16-byte LDS reads: dy rows dyH[co][px] and — because a tap shifts the pixel window by kw — three pre-shifted copies of
// the halo rows, xH[kw][ci][kh][px] (row pitch 40 halfs = 80 B: conflict-free ds_read_b128).
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
constexpr int WH_P = 40;     // halfs per LDS row (32 pixels + pad)

__global__ __launch_bounds__(256) void conv3x3_wgrad_f16_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                                float* __restrict__ slab, int N, int Cin, int H, int W,
                                                                int Cout, int SW, int nseg_total, int segs_per_split) {
    __shared__ __attribute__((aligned(16))) _Float16 dyH[64 * WH_P];
    __shared__ __attribute__((aligned(16))) _Float16 xH[3 * 64 * 3 * WH_P];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lk = lane >> 5;
    // Workgroup -> (channel tile, split): the 16 (ci, co) tiles of one split read the same pixels of x and dy, so they are
    // placed on ONE XCD (linear id b runs on XCD b % 8): a line then comes from HBM / Infinity Cache once and from that
    // XCD's L2 for the other tiles.  (The loads of a segment are bound by line fetches: ~64 outstanding per CU x latency.)
    int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    if (VOCR_WGRAD_XCD && (gridDim.z & 7) == 0) {
        const int nxy = gridDim.x * gridDim.y;
        const int b = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
        const int k = b & 7, slot = b >> 3;
        bz = k + 8 * (slot / nxy);
        const int xy = slot - (slot / nxy) * nxy;
        bx = xy % gridDim.x;
        by = xy / gridDim.x;
    }
    const int ci0 = bx * 64, co0 = by * 64, split = bz;
    const int wco = (wave >> 1) * 32, wci = (wave & 1) * 32;
    const long HW = (long)H * W;

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    float rdy[8], rx[48];
    const int sbeg = split * segs_per_split;
    const int send = min(nseg_total, sbeg + segs_per_split);
    const int dpx = tid & 31, dco = tid >> 5;

    const float* dy_cb[8];
    float dy_m[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int co = co0 + dco + 8 * e;
        dy_m[e] = co < Cout ? 1.f : 0.f;
        dy_cb[e] = dy + (long)min(co, Cout - 1) * HW;
    }
    const float* x_cb[16];
    float x_m[16];
#pragma unroll
    for (int ci = 0; ci < 16; ++ci) {
        const int gci = ci0 + wave * 16 + ci;
        x_m[ci] = gci < Cin ? 1.f : 0.f;
        x_cb[ci] = x + (long)min(gci, Cin - 1) * HW;
    }
    struct SegPos { int dy_off; int x_off[3]; float dy_ok; float row_ok[3]; };
    auto seg_pos = [&](int g) {
        SegPos p;
        const int n = g / (H * SW), rem = g % (H * SW);
        const int h = rem / SW, w0 = (rem % SW) * SEGW;
        p.dy_ok = w0 + dpx < W ? 1.f : 0.f;
        p.dy_off = n * Cout * (int)HW + h * W + min(w0 + dpx, W - 1);
        const int ww = w0 - 1 + lane;
        const bool colok = lane < PROW && ww >= 0 && ww < W;
        const int loff = min(max(ww, 0), W - 1);
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            const int hh = h + kh - 1;
            p.row_ok[kh] = (colok && hh >= 0 && hh < H) ? 1.f : 0.f;
            p.x_off[kh] = n * Cin * (int)HW + min(max(hh, 0), H - 1) * W + loff;
        }
        return p;
    };
    auto load_seg = [&](const SegPos& p) {
#pragma unroll
        for (int e = 0; e < 8; ++e) rdy[e] = dy_cb[e][p.dy_off];
#pragma unroll
        for (int r = 0; r < 48; ++r) rx[r] = x_cb[r / 3][p.x_off[r % 3]];
    };
    auto store_seg = [&](const SegPos& p) {
#pragma unroll
        for (int e = 0; e < 8; ++e) dyH[(dco + 8 * e) * WH_P + dpx] = (_Float16)(rdy[e] * (dy_m[e] * p.dy_ok));
        if (lane < PROW) {
#pragma unroll
            for (int ci = 0; ci < 16; ++ci)
#pragma unroll
                for (int kh = 0; kh < 3; ++kh) {
                    const _Float16 v = (_Float16)(rx[ci * 3 + kh] * (x_m[ci] * p.row_ok[kh]));
#pragma unroll
                    for (int kw = 0; kw < 3; ++kw) {
                        const int ppos = lane - kw;          // copy kw holds column (p + kw) at position p
                        if (ppos >= 0 && ppos < SEGW) xH[((kw * 64 + wave * 16 + ci) * 3 + kh) * WH_P + ppos] = v;
                    }
                }
        }
    };
    if (sbeg < send) {
        const SegPos p0 = seg_pos(sbeg);
        load_seg(p0);
        store_seg(p0);
    }
    __syncthreads();
    for (int g = sbeg; g < send; ++g) {
        const SegPos pn = seg_pos(min(g + 1, send - 1));
        load_seg(pn);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const half8_t a = *(const half8_t*)(dyH + (wco + li) * WH_P + 16 * ks + 8 * lk);
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int kh = t / 3, kw = t % 3;
                const half8_t b = *(const half8_t*)(xH + ((kw * 64 + wci + li) * 3 + kh) * WH_P + 16 * ks + 8 * lk);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[t], 0, 0, 0);
            }
        }
        __syncthreads();
        store_seg(pn);
        __syncthreads();
    }
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

__global__ void wgrad_reduce_scalar_kernel(const float* __restrict__ slab, float* __restrict__ dw, int Cout, int Cin, int splits) {
    const long plane = (long)Cout * Cin;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;   // index over [tap][co][ci]
    if (i >= 9 * plane) return;
    const int t = (int)(i / plane);
    const long r = i % plane;
    float s = 0.f;
    for (int sp = 0; sp < splits; ++sp) s += slab[((long)sp * 9 + t) * plane + r];
    dw[r * 9 + t] = s;
}

// Same reduction with four times the loads in flight: a workgroup sums 64 float4 columns, its four waves take the slabs
// sp = wave, wave+4, ... and the partial sums are combined in a fixed order ((w0+w1)+(w2+w3)): bitwise reproducible.
// Needs Cout*Cin % 4 == 0 (a float4 then never straddles two taps).
// G = waves per workgroup = groups of slabs summed side by side.  G = 16 (round 4) for small weight tensors: the first layer's 576 values
// in 1024 slabs made 3 workgroups of 4 waves with 256 dependent-latency loads each - 68 us at the very end of the step, in front of the
// optimiser; 16 waves with 64 loads each: ~10.
template <int G>
__global__ __launch_bounds__(64 * G) void wgrad_reduce_kernel(const float* __restrict__ slab, float* __restrict__ dw, int Cout, int Cin,
                                                              int splits) {
    __shared__ f32x4 red[G][64];
    const long plane = (long)Cout * Cin;
    const long total4 = 9 * plane / 4;
    const int lane = threadIdx.x & 63, g = threadIdx.x >> 6;
    const long o = (long)blockIdx.x * 64 + lane;
    f32x4 s = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (o < total4)
        for (int sp = g; sp < splits; sp += G) s += *(const f32x4*)(slab + (long)sp * 9 * plane + o * 4);
    red[g][lane] = s;
    __syncthreads();
    if (g == 0 && o < total4) {
        f32x4 v[G];
#pragma unroll
        for (int i = 0; i < G; ++i) v[i] = red[i][lane];
#pragma unroll
        for (int w = 1; w < G; w *= 2)                // fixed balanced tree: ((w0+w1)+(w2+w3)) ...
#pragma unroll
            for (int i = 0; i + w < G; i += 2 * w) v[i] += v[i + w];
        const long i = o * 4;
        const int t = (int)(i / plane);
        const long r = i - (long)t * plane;
#pragma unroll
        for (int e = 0; e < 4; ++e) dw[(r + e) * 9 + t] = v[0][e];
    }
}

// per-channel sum over (n, h*w): grid (C, chunks); double accumulation inside a workgroup; with more than one chunk the
// partial sums go to part[chunk][C] and channel_sum_final_kernel adds them in chunk order (reproducible, no atomics)
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
        out[(long)j * C + c] = (float)((red[0] + red[1]) + (red[2] + red[3]));
    }
}

__global__ void channel_sum_final_kernel(const float* __restrict__ part, float* __restrict__ out, int C, int nchunk) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float v = part[c];
    for (int j = 1; j < nchunk; ++j) v += part[(long)j * C + c];
    out[c] = v;
}

const float* zero_page_ptr() {
    static const float* zp[64] = {nullptr};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    if (!zp[dev]) {
        void* p = nullptr;
        if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_zero_page)) != hipSuccess) return nullptr;
        zp[dev] = (const float*)p;
    }
    return zp[dev];
}

void launch_wgrad_reduce(const float* slab, float* dw, int cout, int cin, int splits, hipStream_t s) {
    const long total = 9l * cout * cin;
    if (((long)cout * cin) % 4 == 0 && (((uintptr_t)slab) & 15) == 0) {
        const unsigned wgs = (unsigned)((total / 4 + 63) / 64);
        if (wgs < 64 && splits >= 64) wgrad_reduce_kernel<16><<<wgs, 1024, 0, s>>>(slab, dw, cout, cin, splits);
        else wgrad_reduce_kernel<4><<<wgs, 256, 0, s>>>(slab, dw, cout, cin, splits);
    }
    else
        wgrad_reduce_scalar_kernel<<<vocr_cdiv(total, 256), 256, 0, s>>>(slab, dw, cout, cin, splits);
}

// f16 = the fp16-operand kernel, which still enumerates one segment per 32-pixel piece of a row
// output channels per workgroup of the generic f32 kernel.  VOCR_WGRAD_TM=2 (experiment): 128, two dy fragments per wave -
// measured 1117 vs 680 us on the 256->256 layer: 288 accumulator registers plus the staging registers spill.
int wgrad_cot(int cin, int cout, bool f16 = false) {
    static const int tm = VOCR_EXPERIMENT_INT("VOCR_WGRAD_TM", 1);
    return (!f16 && cin > 3 && cout >= 128 && tm == 2) ? 128 : 64;
}

int wgrad_splits(int n, int cin, int h, int w, int cout, int* segs_per_split, bool f16 = false) {
    const int SW = vocr_cdiv(w, SEGW);
    const long nseg = (cin <= 3 || f16) ? (long)n * h * SW : (long)seg_geom(n, h, w).nseg;
    const int tiles = (cin <= 3 ? 1 : vocr_cdiv(cin, 64)) * vocr_cdiv(cout, wgrad_cot(cin, cout, f16));
    // the generic kernel holds one workgroup per CU (9 accumulators per wave): 256 slabs = one full round and a
    // 3x smaller slab than 768; the small-Cin kernel is light and streams, give it more
    // (capping the registers for two workgroups per CU spills 26 dwords per lane into the loop: 78-90 TF instead of 98-107)
    long s = ((cin <= 3 ? 1024 : 256) + tiles - 1) / tiles;
    if (s > nseg) s = nseg;
    if (s < 1) s = 1;
    const int sps = (int)((nseg + s - 1) / s);
    *segs_per_split = sps;
    return (int)((nseg + sps - 1) / sps);
}

}  // namespace

// used by conv_wino.hip (same slab layout [split][tap][co][ci])
void vocr_internal_wgrad_reduce(const float* slab, float* dw, int cout, int cin, int splits, hipStream_t s) {
    launch_wgrad_reduce(slab, dw, cout, cin, splits, s);
}


extern "C" int vocr_conv3x3_pack_weights(const float* w, float* wpack_fwd, float* wpack_dgrad, int cout, int cin, void* stream) {
    VOCR_CHECK_ARG(w && (wpack_fwd || wpack_dgrad) && cout > 0 && cin > 0, "vocr_conv3x3_pack_weights: bad argument");
    const int total = cout * cin * 9;
    pack_weights_kernel<<<vocr_cdiv(total, 256), 256, 0, (hipStream_t)stream>>>(w, wpack_fwd, wpack_dgrad, cout, cin);
    VOCR_CHECK_LAUNCH("vocr_conv3x3_pack_weights");
    return VOCR_OK;
}

static int conv_cu_count() {
    static int cus[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    if (!cus[dev]) {
        hipDeviceProp_t prop;
        cus[dev] = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
    }
    return cus[dev];
}

extern "C" int vocr_conv3x3_fwd(const float* x, const float* wpack, const float* bias, float* y, int n, int cin, int h,
                                int w, int cout, void* stream) {
    VOCR_CHECK_ARG(x && wpack && y, "vocr_conv3x3_fwd: null pointer");
    VOCR_CHECK_ARG(n > 0 && cin > 0 && h > 0 && w > 0 && cout > 0, "vocr_conv3x3_fwd: bad shape");
    VOCR_CHECK_ARG((long)n * h * vocr_cdiv(w, SEGW) < (1l << 30), "vocr_conv3x3_fwd: too many segments");
    const SegGeom geo = seg_geom(n, h, w);
    const long nseg = geo.nseg;
    hipStream_t s = (hipStream_t)stream;
    // (Cin <= 3, the first layer: a vector-ALU kernel - one thread per pixel, 64 accumulators, a tap's weights as scalar loads,
    // one coalesced store per channel - was tried instead of padding K to 8 channels for the MFMA path: 97 vs 79 us, removed.)
    const bool vec = (cout % 4 == 0) && ((((uintptr_t)wpack) & 15) == 0);
    const float* zp = zero_page_ptr();
    VOCR_CHECK_ARG(zp != nullptr, "vocr_conv3x3_fwd: no device zero page");
    // full-size workgroups (128co x 4 segments or 64co x 8 segments); fewer than 12 of them per CU -> half-size (measured:
    // 6 was the break-even while every launch ran alone; with the data-gradient launches sharing the chip with the
    // weight-gradient kernel the half-size ones win up to 9 per CU, +0.4 % on the step)
    const int co_tiles = cout > 64 ? vocr_cdiv(cout, 128) : 1;
    const long full_wgs = (long)vocr_cdiv(nseg, cout > 64 ? 4 : 8) * co_tiles;
    static const int tile_mode = VOCR_EXPERIMENT_INT("VOCR_CONV_TILE", 0);     // experiments: 1 half, 2 full
    const bool small = tile_mode == 1 ? true : tile_mode == 2 ? false : full_wgs < 12l * 256;
    const bool tiny = tile_mode == 3;
    static const int lds_pad = VOCR_EXPERIMENT_INT("VOCR_CONV_LDS_PAD", 0);   // experiments: extra LDS = fewer workgroups per CU        // experiments: 32 output channels per wave (one accumulator), half the tile again
#define VOCR_CONV(CO_T, SPWV, NSEG, WCO)                                                                                     \
    do {                                                                                                                    \
        dim3 grid(vocr_cdiv(nseg, NSEG) * co_tiles);                                                                        \
        if (vec) conv3x3_kernel<CO_T, SPWV, true, WCO><<<grid, 256, lds_pad, s>>>(x, wpack, bias, y, zp, n, cin, h, w, cout, geo, co_tiles);  \
        else conv3x3_kernel<CO_T, SPWV, false, WCO><<<grid, 256, lds_pad, s>>>(x, wpack, bias, y, zp, n, cin, h, w, cout, geo, co_tiles);     \
    } while (0)
    // VOCR_CONV_DMA=0: the register-staged kernel.  (A producer/consumer variant - two extra loader waves per workgroup, the
    // four MFMA waves doing nothing but fragment reads and MFMAs between barriers - was built and measured SLOWER, 907-1020
    // vs 708 us on the 256->256 layer: with double buffering the loaders can only run one half-chunk ahead, so their
    // load -> land -> store latency (an LDS-DMA takes ~1.1 us from issue to landed) sits on the critical path of every
    // half-chunk instead of hiding behind the issuing wave's own MFMAs; a third buffer does not fit three workgroups per CU.)
    static const int use_dma = VOCR_EXPERIMENT_INT("VOCR_CONV_DMA", 1);
    // VOCR_CONV_TAIL: 1 (default) the last partial round of tiles is cut into pieces that lead the same launch, 3 the pieces
    // run as conv3x3_tail_kernel behind the launch, 0 one launch of whole tiles as before; any other value = 1.  (The
    // upper-bound experiment that dropped the partial round - wrong results, -8.7 % on the forward stack - is no longer in
    // the shipped library.)
    static const int tail_env = VOCR_EXPERIMENT_INT("VOCR_CONV_TAIL", 1);
    static const int tail_mode = (tail_env == 0 || tail_env == 3) ? tail_env : 1;
    const int ncu = conv_cu_count();
    if (use_dma && vec && !tiny) {
        // LDS-DMA form (weights by global_load_lds into ping-pong half-chunk buffers)
#define VOCR_CONV_DMA_LAUNCH(CO_T, SPWV, NSEG)                                                                              \
        do {                                                                                                                \
            const int tiles = vocr_cdiv(nseg, NSEG) * co_tiles, rem = tiles % ncu;                                         \
            const bool cut = tail_mode && tiles > ncu && rem > 0 && rem <= ncu / 2;                                        \
            const int n_main = cut ? tiles - rem : tiles;                                                                   \
            const int n_tail = (cut && tail_mode == 1) ? rem * (CO_T / 32) * NSEG : 0;                                       \
            conv3x3_dma_kernel<CO_T, SPWV, 64><<<dim3(n_tail + n_main), 256, lds_pad, s>>>(x, wpack, bias, y, zp, n, cin, h, w, cout, geo, co_tiles, n_tail, n_main);   \
            if (cut && tail_mode == 3)                                                                                      \
                conv3x3_tail_kernel<<<dim3(rem * (CO_T / 32) * NSEG), 512, 0, s>>>(x, wpack, bias, y, zp, cin, h, w, cout, geo, co_tiles, CO_T, NSEG, n_main); \
        } while (0)
        if (cout > 64) {
            if (small) VOCR_CONV_DMA_LAUNCH(128, 1, 2); else VOCR_CONV_DMA_LAUNCH(128, 2, 4);
        } else {
            if (small) VOCR_CONV_DMA_LAUNCH(64, 1, 4); else VOCR_CONV_DMA_LAUNCH(64, 2, 8);
        }
#undef VOCR_CONV_DMA_LAUNCH
    } else if (cout > 64) {
        if (tiny) VOCR_CONV(128, 1, 1, 32); else if (small) VOCR_CONV(128, 1, 2, 64); else VOCR_CONV(128, 2, 4, 64);
    } else {
        if (tiny) VOCR_CONV(64, 1, 2, 32); else if (small) VOCR_CONV(64, 1, 4, 64); else VOCR_CONV(64, 2, 8, 64);
    }
#undef VOCR_CONV
    VOCR_CHECK_LAUNCH("vocr_conv3x3_fwd");
    return VOCR_OK;
}

extern "C" size_t vocr_conv3x3_wgrad_workspace_bytes(int n, int cin, int h, int w, int cout) {
    if (n <= 0 || cin <= 0 || h <= 0 || w <= 0 || cout <= 0) return 0;
    int sps;
    const int a = wgrad_splits(n, cin, h, w, cout, &sps), b = wgrad_splits(n, cin, h, w, cout, &sps, true);   // f32 / fp16-operand kernels
    return (size_t)(a > b ? a : b) * 9 * cout * cin * sizeof(float);
}

extern "C" int vocr_conv3x3_wgrad(const float* x, const float* dy, float* dw, void* workspace, int n, int cin, int h,
                                  int w, int cout, void* stream) {
    VOCR_CHECK_ARG(x && dy && dw && workspace, "vocr_conv3x3_wgrad: null pointer");
    VOCR_CHECK_ARG(n > 0 && cin > 0 && h > 0 && w > 0 && cout > 0, "vocr_conv3x3_wgrad: bad shape");
    VOCR_CHECK_ARG((long)n * (cin > cout ? cin : cout) * h * w < (1l << 31), "vocr_conv3x3_wgrad: tensor exceeds 2^31 elements");
    const int SW = vocr_cdiv(w, SEGW);
    const long nseg = (long)n * h * SW;
    int sps;
    const int splits = wgrad_splits(n, cin, h, w, cout, &sps);
    hipStream_t s = (hipStream_t)stream;
    const float* zp = zero_page_ptr();
    VOCR_CHECK_ARG(zp != nullptr, "vocr_conv3x3_wgrad: no device zero page");
    if (cin <= 3) {
        dim3 grid(1, vocr_cdiv(cout, 64), splits);
        conv3x3_wgrad_smallcin_kernel<<<grid, 128, 0, s>>>(x, dy, (float*)workspace, zp, n, cin, h, w, cout, SW, (int)nseg, sps);
    } else {
        const int cot = wgrad_cot(cin, cout);
        dim3 grid(vocr_cdiv(cin, 64), vocr_cdiv(cout, cot), splits);
        static const int mode = VOCR_EXPERIMENT_INT("VOCR_WGRAD_MODE", 2);
        if (cot == 128) conv3x3_wgrad_kernel<0, 2><<<grid, 256, 0, s>>>(x, dy, (float*)workspace, zp, n, cin, h, w, cout, seg_geom(n, h, w), sps);
        else if (mode == 2) conv3x3_wgrad_kernel<2><<<grid, 512, 0, s>>>(x, dy, (float*)workspace, zp, n, cin, h, w, cout, seg_geom(n, h, w), sps);
        else if (mode == 1) conv3x3_wgrad_kernel<1><<<grid, 256, 0, s>>>(x, dy, (float*)workspace, zp, n, cin, h, w, cout, seg_geom(n, h, w), sps);
        else conv3x3_wgrad_kernel<0><<<grid, 256, 0, s>>>(x, dy, (float*)workspace, zp, n, cin, h, w, cout, seg_geom(n, h, w), sps);
    }
    VOCR_CHECK_LAUNCH("vocr_conv3x3_wgrad");
    launch_wgrad_reduce((const float*)workspace, dw, cout, cin, splits, s);
    VOCR_CHECK_LAUNCH("vocr_conv3x3_wgrad(reduce)");
    return VOCR_OK;
}

extern "C" int vocr_conv3x3_wgrad_f16(const float* x, const float* dy, float* dw, void* workspace, int n, int cin, int h,
                                      int w, int cout, void* stream) {
    VOCR_CHECK_ARG(x && dy && dw && workspace, "vocr_conv3x3_wgrad_f16: null pointer");
    VOCR_CHECK_ARG(n > 0 && cin > 0 && h > 0 && w > 0 && cout > 0, "vocr_conv3x3_wgrad_f16: bad shape");
    VOCR_CHECK_ARG((long)n * (cin > cout ? cin : cout) * h * w < (1l << 31), "vocr_conv3x3_wgrad_f16: tensor exceeds 2^31 elements");
    if (cin <= 3) return vocr_conv3x3_wgrad(x, dy, dw, workspace, n, cin, h, w, cout, stream);   // first layer: memory-bound f32 path
    const int SW = vocr_cdiv(w, SEGW);
    const long nseg = (long)n * h * SW;
    int sps;
    const int splits = wgrad_splits(n, cin, h, w, cout, &sps, true);
    hipStream_t s = (hipStream_t)stream;
    dim3 grid(vocr_cdiv(cin, 64), vocr_cdiv(cout, 64), splits);
    conv3x3_wgrad_f16_kernel<<<grid, 256, 0, s>>>(x, dy, (float*)workspace, n, cin, h, w, cout, SW, (int)nseg, sps);
    VOCR_CHECK_LAUNCH("vocr_conv3x3_wgrad_f16");
    launch_wgrad_reduce((const float*)workspace, dw, cout, cin, splits, s);
    VOCR_CHECK_LAUNCH("vocr_conv3x3_wgrad_f16(reduce)");
    return VOCR_OK;
}

static inline int channel_sum_chunks(int n, int hw) {
    long nchunk = ((long)n * hw + 16383) / 16384;
    if (nchunk > 64) nchunk = 64;
    return (int)(nchunk < 1 ? 1 : nchunk);
}

extern "C" size_t vocr_channel_sum_workspace_bytes(int n, int c, int hw) {
    if (n <= 0 || c <= 0 || hw <= 0) return 0;
    const int nchunk = channel_sum_chunks(n, hw);
    return nchunk > 1 ? (size_t)nchunk * c * sizeof(float) : 0;
}

extern "C" int vocr_channel_sum(const float* x, float* out, int n, int c, int hw, void* workspace, void* stream) {
    VOCR_CHECK_ARG(x && out && n > 0 && c > 0 && hw > 0, "vocr_channel_sum: bad argument");
    hipStream_t s = (hipStream_t)stream;
    const int nchunk = workspace ? channel_sum_chunks(n, hw) : 1;
    channel_sum_kernel<<<dim3(c, (unsigned)nchunk), 256, 0, s>>>(x, nchunk > 1 ? (float*)workspace : out, n, c, hw, nchunk);
    VOCR_CHECK_LAUNCH("vocr_channel_sum");
    if (nchunk > 1) {
        channel_sum_final_kernel<<<vocr_cdiv(c, 64), 64, 0, s>>>((const float*)workspace, out, c, nchunk);
        VOCR_CHECK_LAUNCH("vocr_channel_sum(final)");
    }
    return VOCR_OK;
}

// =====================================================================================================================
// Round 5: the ONE-input-channel convolution (the first layer of every grey-line configuration, and configs[4]'s rapid_ds stage) as
// plain vector arithmetic.  576 FLOP per output pixel against 256 bytes written: the layer is bound by streaming its OUTPUT once
// (147 MB at batch 32 = ~30 us at HBM speed); the MFMA kernels spent 78 us (Cin padded to 4 in the F(2,3) kernel) and 375 us (Cin
// padded to 16 in the fp16 kernel) on it.  A thread owns 4 consecutive pixels: its 3 x 6 input window stays in registers, the filter
// taps are wave-uniform (scalar loads), one 16-byte store per output channel = 1 KB contiguous per wave.
// round_f16: operands rounded to fp16 first (the fp16-operand configuration's arithmetic), fp32 accumulation either way.
namespace {

__global__ __launch_bounds__(256) void conv3x3_c1_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                             const float* __restrict__ bias, float* __restrict__ y, int H, int W, int Cout,
                                                             int round_f16) {
    // a thread's 4 pixels never cross a row: quads are counted per row (QW = ceil(W / 4)) and flattened over the image's rows, so a
    // 600-pixel row does not leave 106 of 256 threads idle
    const int n = blockIdx.y;
    const int QW = (W + 3) >> 2;
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= H * QW) return;
    const int h = q / QW, w0 = (q % QW) * 4;
    const float* xi = x + (long)n * H * W;
    float win[3][6];
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
        const int hh = h + kh - 1;
        const bool rok = hh >= 0 && hh < H;
        const float* row = xi + (long)(rok ? hh : 0) * W;
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const int ww = w0 - 1 + j;
            float v = (rok && ww >= 0 && ww < W) ? row[ww] : 0.f;
            if (round_f16) v = (float)(_Float16)v;
            win[kh][j] = v;
        }
    }
    float* yo = y + ((long)n * Cout * H + h) * W + w0;
    const long cstride = (long)H * W;
    const bool full = w0 + 3 < W && ((((uintptr_t)yo) | ((uintptr_t)(cstride * 4))) & 15) == 0;
    for (int co = 0; co < Cout; ++co) {
        float k[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            k[t] = w[co * 9 + t];                     // wave-uniform: scalar loads
            if (round_f16) k[t] = (float)(_Float16)k[t];
        }
        const float b = bias ? bias[co] : 0.f;
        float o[4];
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            float a = 0.f;
#pragma unroll
            for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) a = __builtin_fmaf(win[kh][p + kw], k[kh * 3 + kw], a);
            o[p] = a + b;
        }
        float* d = yo + co * cstride;
        if (full) {
            *(f32x4*)d = (f32x4){o[0], o[1], o[2], o[3]};
        } else {
#pragma unroll
            for (int p = 0; p < 4; ++p)
                if (w0 + p < W) d[p] = o[p];
        }
    }
}

// dw[co][tap] = sum over (n, h, w) of dy[n][co][h][w] x[n][h + kh - 1][w + kw - 1].  Workgroup = (group of 4 output channels, split of
// the image rows); a thread keeps 4 x 9 accumulators over its pixel lane of every row of the split, dy streams through once (one
// coalesced load per channel and row strip), the nine x neighbours come from the cache; block tree in LDS, then the splits are added
// in a fixed order by the second kernel (bitwise reproducible).
// The bias gradient (sum of dy per channel) rides along as a tenth accumulator: part[split][co][10].
__global__ __launch_bounds__(256) void conv3x3_c1_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                               float* __restrict__ part, int N, int H, int W, int Cout, int rows_per_split) {
    __shared__ float red[40][257];
    const int tid = threadIdx.x;
    const int cg = blockIdx.x, split = blockIdx.y;
    const int co0 = cg * 4;
    const int nco = min(4, Cout - co0);
    const long HW = (long)H * W;
    const int r0 = split * rows_per_split, r1 = min(N * H, r0 + rows_per_split);
    float acc[4][10];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int t = 0; t < 10; ++t) acc[c][t] = 0.f;
    // the split's pixels as one flat range of QUADS (4 consecutive pixels of a row; rows hold QW = ceil(W / 4) of them): a thread reads
    // its 3 x 6 input window with three 16-byte loads + the two edge columns, and one 16-byte load of dy per channel
    const int QW = (W + 3) >> 2;
    const bool vec = (W & 3) == 0 && ((((uintptr_t)x) | ((uintptr_t)dy)) & 15) == 0;
    for (long q = (long)r0 * QW + tid; q < (long)r1 * QW; q += 256) {
        const int r = (int)(q / QW), w0 = (int)(q % QW) * 4;
        const int n = r / H, h = r % H;
        const float* xi = x + (long)n * HW;
        const float* dyb = dy + ((long)n * Cout + co0) * HW + (long)h * W + w0;
        float win[3][6];
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            const int hh = h + kh - 1;
            const bool rok = hh >= 0 && hh < H;
            const float* row = xi + (long)(rok ? hh : 0) * W;
            if (vec) {
                const f32x4 v = rok ? *(const f32x4*)(row + w0) : (f32x4){0.f, 0.f, 0.f, 0.f};
                win[kh][1] = v[0]; win[kh][2] = v[1]; win[kh][3] = v[2]; win[kh][4] = v[3];
                win[kh][0] = (rok && w0 > 0) ? row[w0 - 1] : 0.f;
                win[kh][5] = (rok && w0 + 4 < W) ? row[w0 + 4] : 0.f;
            } else {
#pragma unroll
                for (int j = 0; j < 6; ++j) {
                    const int ww = w0 - 1 + j;
                    win[kh][j] = (rok && ww >= 0 && ww < W) ? row[ww] : 0.f;
                }
            }
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if (c < nco) {
                float g[4];
                if (vec) {
                    const f32x4 v = __builtin_nontemporal_load((const f32x4*)(dyb + c * HW));
                    g[0] = v[0]; g[1] = v[1]; g[2] = v[2]; g[3] = v[3];
                } else {
#pragma unroll
                    for (int p = 0; p < 4; ++p) g[p] = w0 + p < W ? dyb[c * HW + p] : 0.f;
                }
#pragma unroll
                for (int p = 0; p < 4; ++p) {
#pragma unroll
                    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                        for (int kw = 0; kw < 3; ++kw) acc[c][kh * 3 + kw] = __builtin_fmaf(g[p], win[kh][p + kw], acc[c][kh * 3 + kw]);
                    acc[c][9] += g[p];
                }
            }
        }
    }
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int t = 0; t < 10; ++t) red[c * 10 + t][tid] = acc[c][t];
    __syncthreads();
    // 40 rows of 256 partial sums: wave w reduces rows w, w + 4, ... with a fixed shuffle tree (a shuffle tree per accumulator in every
    // wave and twice as many, shorter splits measured 100 / 188 us against 69 / 116)
    const int wave = tid >> 6, lane = tid & 63;
    for (int row = wave; row < 40; row += 4) {
        float v = (red[row][lane] + red[row][lane + 64]) + (red[row][lane + 128] + red[row][lane + 192]);
        v = wave_sum(v);
        const int c = row / 10, t = row % 10;
        if (lane == 0 && c < nco) part[((long)split * Cout + co0 + c) * 10 + t] = v;
    }
}

// one wave per output: lane l adds the splits l, l + 64, ... in order, then a fixed shuffle tree (a thread per output walking all <= 1024
// splits was 60 us at the very end of the step)
__global__ __launch_bounds__(256) void conv3x3_c1_wgrad_final_kernel(const float* __restrict__ part, float* __restrict__ dw, float* __restrict__ dbias,
                                                                     int cout, int splits) {
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;       // (co, t): t < 9 taps, t = 9 the bias gradient
    const int n_out = cout * 10;
    if (i >= n_out) return;
    float v = 0.f;
    for (int s = lane; s < splits; s += 64) v += part[(long)s * n_out + i];
    v = wave_sum(v);
    if (lane == 0) {
        const int co = i / 10, t = i % 10;
        if (t < 9) dw[co * 9 + t] = v;
        else if (dbias) dbias[co] = v;
    }
}

inline int c1_wgrad_splits(int n, int h, int cout) {
    const int groups = (cout + 3) / 4;
    int s = (4 * 256 + groups - 1) / groups;          // ~4 workgroups per CU in total
    if (s > n * h) s = n * h;
    return s < 1 ? 1 : s;
}

}  // namespace

extern "C" int vocr_conv3x3_c1_fwd(const float* x, const float* w, const float* bias, float* y, int n, int h, int wd, int cout, int round_f16,
                                   void* stream) {
    VOCR_CHECK_ARG(x && w && y && n > 0 && h > 0 && wd > 0 && cout > 0, "vocr_conv3x3_c1_fwd: bad argument");
    VOCR_CHECK_ARG(n <= 65535, "vocr_conv3x3_c1_fwd: grid too large");
    conv3x3_c1_fwd_kernel<<<dim3(vocr_cdiv((long)h * ((wd + 3) / 4), 256), n), 256, 0, (hipStream_t)stream>>>(x, w, bias, y, h, wd, cout, round_f16);
    VOCR_CHECK_LAUNCH("vocr_conv3x3_c1_fwd");
    return VOCR_OK;
}

extern "C" size_t vocr_conv3x3_c1_wgrad_workspace_bytes(int n, int h, int cout) {
    if (n <= 0 || h <= 0 || cout <= 0) return 0;
    return (size_t)c1_wgrad_splits(n, h, cout) * cout * 10 * sizeof(float);
}

extern "C" int vocr_conv3x3_c1_wgrad(const float* x, const float* dy, float* dw, float* dbias, void* workspace, int n, int h, int wd, int cout,
                                     void* stream) {
    VOCR_CHECK_ARG(x && dy && dw && workspace && n > 0 && h > 0 && wd > 0 && cout > 0, "vocr_conv3x3_c1_wgrad: bad argument");
    const int splits = c1_wgrad_splits(n, h, cout);
    const int rps = vocr_cdiv((long)n * h, splits);
    hipStream_t s = (hipStream_t)stream;
    conv3x3_c1_wgrad_kernel<<<dim3(vocr_cdiv(cout, 4), splits), 256, 0, s>>>(x, dy, (float*)workspace, n, h, wd, cout, rps);
    VOCR_CHECK_LAUNCH("vocr_conv3x3_c1_wgrad");
    conv3x3_c1_wgrad_final_kernel<<<vocr_cdiv(cout * 10, 4), 256, 0, s>>>((const float*)workspace, dw, dbias, cout, splits);
    VOCR_CHECK_LAUNCH("vocr_conv3x3_c1_wgrad(final)");
    return VOCR_OK;
}
