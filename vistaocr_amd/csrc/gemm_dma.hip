// fp32 MFMA GEMM for the LARGE dense products of the step (LSTM input projections, their data and weight gradients):
// C[M,N] = op(A)[M,K] * op(B)[K,N] (+bias)(relu), up to two products per launch (two independent ones of one shape, or two K
// segments summed into one C).  gemm.hip keeps the small and the odd-shaped ones.
//
// gfx950 mapping (what differs from gemm.hip, whose main loop reached 0.74 MFMA-busy on these shapes):
//   * ONE workgroup per CU, persistent over a PANEL: 128 output columns x a contiguous range of 32-row tiles chosen so that
//     (products x column panels x row groups x K splits) == the CU count - the launch has no partial last round (a 9408 x 2048
//     product is 1184 tiles of 128 x 128 = 4.6 rounds of 256 CUs; as 16 panels x 16 row groups of 18-19 row tiles it is 1.0).
//     A group walks its rows in chunks of <= 8 row tiles; the weight panel is re-streamed per chunk (L2 hits).
//   * operands reach LDS by DMA only (buffer_load ... lds, 16 B per lane, nothing returns to a register, no ds_write): a ring
//     of THREE K-tiles of 32 (48 KB each: 256 x 32 of A, 128 x 32 of B), the 12 DMAs of a wave for tile t+2 are issued one at a
//     time between the MFMA groups of tile t (an issue costs ~60-100 cycles, a v_mfma_f32_32x32x2_f32 holds the pipe for 64),
//     one barrier per K-tile.  Out-of-range rows / columns / k are loads outside the buffer's range: they deliver zeros.
//   * a K-contiguous operand ([row][k] in memory) is stored row-major in LDS, 8 pieces of 16 B per row with the piece position
//     XOR-swizzled by the row (the DMA's source address is per lane, so the swizzle costs nothing to write) and read back with
//     ONE conflict-free ds_read_b128 per 32-row tile and 8 k: lane half h takes k = 8j + 4h + 0..3 - the MFMA only needs A and B
//     to agree on which k a lane half holds.  An M/N-contiguous operand ([k][row]) is stored K-major and read with ds_read_b32.
//   * a wave owns 32 columns x up to 256 rows (8 accumulator tiles, 128 VGPRs), one wave per SIMD; the A fragments are shared
//     by the four waves through LDS.
//   * epilogue through LDS (the ring slot just consumed): 16-byte row-contiguous stores, 32 per wave and chunk, edge rows and
//     columns dropped by the store's range check.
//   * long-K products with few tiles (weight gradients) are cut along K into slabs that splitk_reduce_kernel (gemm.hip) adds
//     in a fixed order: bitwise reproducible, no atomics.
#include "vocr_common.h"
#include "gemm_dma.h"
#include <stdlib.h>
#include <type_traits>

namespace {

constexpr int DK = 32;                       // K-tile
constexpr int DM = 256, DN = 128;            // rows per chunk (8 row tiles of 32), columns per panel
constexpr int A_FL = DM * DK, B_FL = DN * DK;
constexpr int STAGE_FL = A_FL + B_FL;        // 12288 floats = 48 KB
constexpr int NSTAGE = 3;
constexpr unsigned OOB = 0xFFFFFFF0u;
constexpr int EP = 36;                       // epilogue scratch pitch (floats): 16-byte aligned rows

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

struct DmaGemmArgs {
    const float* a[2];
    const float* b[2];
    float* c[2];
    const float* bias[2];
    unsigned a_bytes, b_bytes, c_bytes;      // buffer ranges (bytes from each base pointer)
    int M, N, K;                             // one product (and one K segment)
    int lda, ldb, ldc;
    int nprob, nseg;                         // nprob independent products (a[i], b[i] -> c[i]); nseg K segments (a[s] b[s] summed into c[0])
    int panels, groups, ksplit, kps;         // column panels, row groups, K splits and k per split (multiple of 32)
    int relu;
    int dbg;                                 // experiments (VOCR_GEMM_DBG, wrong results): 1 no DMA in the loop, 2 no barrier in the loop
    float* slab;                             // split-K: [tile = (prob * groups + group) * panels + panel][split][256][128]
};

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

template <bool A_KC, bool B_KC>
__global__ __launch_bounds__(512) void gemm_dma_kernel(const DmaGemmArgs g) {
    __shared__ __attribute__((aligned(16))) float lds[NSTAGE * STAGE_FL];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);          // 0 .. 7: two waves per SIMD
    const int cw = wave & 3, kh = wave >> 2;                            // column group (32 columns) and which half of a chunk's row tiles
    const int m = lane & 31, h = lane >> 5;

    // ---- which panel: XCD-sliced order (consecutive work items = the column panels of one row group = one L2 shares the A rows)
    const int nwg = gridDim.x;
    int lin;
    {
        const int x = blockIdx.x & 7, q = nwg >> 3, r = nwg & 7;
        lin = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (blockIdx.x >> 3);
    }
    const int panel = lin % g.panels;
    int t_ = lin / g.panels;
    const int group = t_ % g.groups;
    t_ /= g.groups;
    const int split = t_ % g.ksplit;
    const int prob = t_ / g.ksplit;
    const int n0 = panel * DN;

    // rows of this group: the 32-row tiles are dealt evenly over the groups; a group walks them in chunks of <= 8 tiles, and inside a chunk the first ceil(n/2) tiles belong to waves 0-3, the rest to waves 4-7
    const int mt = (g.M + 31) >> 5;
    const int tb_ = mt / g.groups, te_ = mt % g.groups;
    // (K split into slabs: whole 8-tile groups, because the slab reduce maps slab tile (group, panel) to rows group * 256)
    const int gtiles = g.ksplit > 1 ? min(8, mt - 8 * group) : tb_ + (group < te_ ? 1 : 0);
    const int gstart = g.ksplit > 1 ? 8 * group : group * tb_ + min(group, te_);
    const int nch = (gtiles + 7) >> 3;
    // ... in nch chunks of EVEN size (19 tiles = 7 + 6 + 6, not 8 + 8 + 3): every K-tile step moves the 16 KB weight tile and pays its barrier
    // whatever the chunk holds, so a 1 - 3 tile last chunk ran at a fraction of the matrix rate
    const int cbase = nch > 0 ? gtiles / nch : 0, cext = nch > 0 ? gtiles % nch : 0;
    const int kbeg = split * g.kps, kend = min(g.K, kbeg + g.kps);
    const int kts = (kend - kbeg + DK - 1) / DK;          // K-tiles of one segment
    const int KT = kts * g.nseg;
    const int total = nch * KT;
    if (total <= 0) return;

    const __amdgpu_buffer_rsrc_t ra0 = __builtin_amdgcn_make_buffer_rsrc((void*)g.a[g.nseg > 1 ? 0 : prob], 0, (int)g.a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t ra1 = __builtin_amdgcn_make_buffer_rsrc((void*)g.a[1], 0, (int)g.a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb0 = __builtin_amdgcn_make_buffer_rsrc((void*)g.b[g.nseg > 1 ? 0 : prob], 0, (int)g.b_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb1 = __builtin_amdgcn_make_buffer_rsrc((void*)g.b[1], 0, (int)g.b_bytes, 0x00020000);

    // ---- DMA maps (per-lane constants): 48 instructions of 1 KB per K-tile, six per wave (four of A, two of B).  K-contiguous
    // operand: instruction e of wave w covers rows 64e + 8w + lane/8, piece position lane%8 holds k-piece kb = pos ^ ((row >> 1) & 7).
    // M/N-contiguous: one instruction = 256 consecutive rows of one k (A: k = w + 8e) or 128 rows of two k (B: k = 2(w + 8e) + lane/32).
    const int kc_row = 8 * wave + (lane >> 3);
    const int kc_k4 = 4 * ((lane & 7) ^ ((4 * wave + (lane >> 4)) & 7));

    struct Step { int slot, seg, k0, row0, ntile; };
    struct Cursor { int c, kt, slot; };                    // (chunk, K-tile of the chunk, ring slot) of a step; no divisions in the loop
    auto advance = [&](Cursor& cu) {
        cu.slot = cu.slot == NSTAGE - 1 ? 0 : cu.slot + 1;
        if (++cu.kt == KT) { cu.kt = 0; ++cu.c; }
    };
    auto step_at = [&](const Cursor& cu) {
        Step st;
        st.seg = cu.kt >= kts ? 1 : 0;
        st.k0 = kbeg + (cu.kt - st.seg * kts) * DK;
        st.ntile = cbase + (cu.c < cext ? 1 : 0);
        st.row0 = 32 * (gstart + cu.c * cbase + min(cu.c, cext));
        st.slot = cu.slot;
        return st;
    };
    // The six DMA byte offsets of a step (0 .. 3: A, 4, 5: B; OOB = out of range = zeros) are computed one at a time BETWEEN the MFMAs
    // of the first K-block of the step that issues them (each is ~3 vector instructions and fits behind one MFMA); the issue itself
    // is then `s_mov m0` + `buffer_load ... lds`.  (Computed next to each load, the dependent v_cmp -> s_and -> v_cndmask -> load chain
    // held the in-order wave ~100 cycles per DMA; computed in one batch at the head of the step, the matrix pipe drained.)
    struct Prep { unsigned a_off, b_off; int a_lim, b_lim; unsigned a_step, b_step; };
    auto prep_head = [&](const Step& st, bool live) {
        Prep pr;
        const int dead = live ? 0 : (1 << 30);
        if (A_KC) {
            pr.a_off = (unsigned)((st.row0 + kc_row) * g.lda + st.k0 + kc_k4) * 4u;
            pr.a_lim = (st.k0 + kc_k4 < kend ? min(g.M - st.row0, 32 * st.ntile) - kc_row : 0) - dead;       // instruction e in range iff 64 e < lim
            pr.a_step = (unsigned)(64 * g.lda) * 4u;
        } else {
            pr.a_off = (unsigned)((st.k0 + wave) * g.lda + st.row0 + 4 * lane) * 4u;
            pr.a_lim = (4 * lane < min(g.M - st.row0, 32 * st.ntile) ? kend - st.k0 - wave : 0) - dead;       // iff 8 e < lim
            pr.a_step = (unsigned)(8 * g.lda) * 4u;
        }
        if (B_KC) {
            pr.b_off = (unsigned)((n0 + kc_row) * g.ldb + st.k0 + kc_k4) * 4u;
            pr.b_lim = (st.k0 + kc_k4 < kend ? g.N - n0 - kc_row : 0) - dead;                                  // iff 64 e < lim
            pr.b_step = (unsigned)(64 * g.ldb) * 4u;
        } else {
            pr.b_off = (unsigned)((st.k0 + 2 * wave + (lane >> 5)) * g.ldb + n0 + 4 * (lane & 31)) * 4u;
            pr.b_lim = (n0 + 4 * (lane & 31) < g.N ? kend - st.k0 - 2 * wave - (lane >> 5) : 0) - dead;        // iff 16 e < lim
            pr.b_step = (unsigned)(16 * g.ldb) * 4u;
        }
        return pr;
    };
    auto prep_one = [&](const Prep& pr, unsigned (&vo)[6], int d) {       // three vector instructions
        if (d < 4) vo[d] = ((A_KC ? 64 : 8) * d < pr.a_lim) ? pr.a_off + (unsigned)d * pr.a_step : OOB;
        else vo[d] = ((B_KC ? 64 : 16) * (d - 4) < pr.b_lim) ? pr.b_off + (unsigned)(d - 4) * pr.b_step : OOB;
    };
    auto issue = [&](const __amdgpu_buffer_rsrc_t& ra, const __amdgpu_buffer_rsrc_t& rb, float* slot, const unsigned (&vo)[6], int d) {
        if (d < 4) __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, LDS_PTR(slot + (wave + 8 * d) * 256), 16, vo[d], 0, 0, 0);
        else __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, LDS_PTR(slot + A_FL + (wave + 8 * (d - 4)) * 256), 16, vo[d], 0, 0, 0);
    };

    // ---- fragment addressing
    const int sw = (m >> 1) & 7;
    const int a_kc_base = m * 32;                          // floats: row m, + tile * 1024 + ((2j + h) ^ sw) * 4
    const int b_kc_base = (32 * cw + m) * 32;
    const int a_mc_base = 4 * h * 256 + m;                 // floats: k = 8j + 4h + s -> + (8j + s) * 256 + tile * 32
    const int b_nc_base = 4 * h * 128 + 32 * cw + m;

    // epilogue constants: this lane stores columns col4 .. +3 of rows srow + 8 * pass of every 32 x 32 tile
    const int col4 = 4 * (lane & 7), srow = lane >> 3;
    const int gcol = n0 + 32 * cw + col4;
    const bool slabbed = g.slab != nullptr;
    f32x4 bv = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (!slabbed && g.bias[prob] && gcol < g.N) bv = *(const f32x4*)(g.bias[prob] + gcol);      // before any DMA: nothing else returns to a register
    float* const outp = slabbed ? g.slab + (((long)prob * g.groups + group) * g.panels + panel) * g.ksplit * (DM * DN) + (long)split * (DM * DN) : g.c[prob];
    const int ld_out = slabbed ? DN : g.ldc;
    const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc((void*)outp, 0, (int)(slabbed ? (unsigned)(DM * DN * 4) : g.c_bytes), 0x00020000);
    const int ncols = slabbed ? DN : g.N;
    const int ocol = slabbed ? 32 * cw + col4 : gcol;

    // prologue: two K-tiles in flight
    Cursor cnext = {0, 0, 0};                              // the step whose DMAs are issued next
    for (int i = 0; i < 2; ++i) {
        const Step s0 = step_at(cnext);
        unsigned vo[6];
        const Prep pr = prep_head(s0, i < total);
#pragma unroll
        for (int d = 0; d < 6; ++d) prep_one(pr, vo, d);
#pragma unroll
        for (int d = 0; d < 6; ++d) issue(s0.seg ? ra1 : ra0, s0.seg ? rb1 : rb0, lds + s0.slot * STAGE_FL, vo, d);
        if (i < total) advance(cnext);
    }
    int issued = total > 1 ? 2 : 1;                        // steps whose (real) DMAs have been issued
    Cursor ccur = {0, 0, 0};
    int since_store = 1;                                   // 0: the previous chunk's 16 stores are the youngest outstanding operations
    // cold start: the first K-tile has landed (all but the 6 DMAs of the second are done), for every wave
    __builtin_amdgcn_s_waitcnt(0x0F70 | 6);
    asm volatile("s_barrier" ::: "memory");

    // One chunk = NT of its row tiles for this wave (compile-time: the accumulators live in AGPRs for the whole K loop and no branch sits
    // between the MFMAs; 0 = this wave only moves data and keeps the barriers) x all K-tiles.  The DMA ring runs on across chunk
    // boundaries.  A step (one K-tile) is four blocks of 8 k:
    //   blocks 0 .. 2   MFMAs; the fragments of the next block are read while a block's MFMAs run; the offsets of the 6 DMAs this wave
    //                   will issue are computed between the MFMAs of block 0
    //   after block 2   this wave's DMAs of step +1 have landed (vmcnt), then the step's ONE barrier: every wave's DMAs of step +1 have
    //                   landed and every wave is past step -1 entirely, so its ring slot may be refilled
    //   block 3         MFMAs; the DMAs of step +2 (into the slot of step -1); the fragments of step +1's first block are read here, so
    //                   no barrier and no LDS latency sits between two steps
    // Two waves share a SIMD: while one waits (barrier, DMA issue, LDS latency, epilogue) the other one's MFMAs keep the pipe busy.
    // (One wave per SIMD with all 8 tiles: 0.79 MFMA-busy; a barrier at the head of every step with the fragment reads behind it cost
    // another ~2000 of a step's 8192 MFMA cycles.)
    auto run_chunk = [&](auto nt_c, int toff) {
        constexpr int NT = decltype(nt_c)::value;
        constexpr int NA = NT > 0 ? NT : 1;
        f32x16 acc[NA];
#pragma unroll
        for (int i = 0; i < NA; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        Step st = step_at(ccur);
        f32x4 af[2][NA], bf[2];
        auto read_frags = [&](const float* As, int j, int buf) {
            if (NT == 0) return;
            const float* const Bs = As + A_FL;
            if (A_KC) {
#pragma unroll
                for (int tm = 0; tm < NT; ++tm) af[buf][tm] = *(const f32x4*)(As + a_kc_base + (toff + tm) * 1024 + (((2 * j + h) ^ sw) << 2));
            } else {
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int tm = 0; tm < NT; ++tm) af[buf][tm][s] = As[a_mc_base + (8 * j + s) * 256 + (toff + tm) * 32];
            }
            if (B_KC) {
                bf[buf] = *(const f32x4*)(Bs + b_kc_base + (((2 * j + h) ^ sw) << 2));
            } else {
#pragma unroll
                for (int s = 0; s < 4; ++s) bf[buf][s] = Bs[b_nc_base + (8 * j + s) * 128];
            }
        };
        read_frags(lds + st.slot * STAGE_FL, 0, 0);        // (once per chunk with its latency exposed; later steps prefetch)
        for (int kt = 0; kt < KT; ++kt) {
            const bool more = issued < total;
            const Step nx = step_at(cnext);
            unsigned vo[6];
            Prep pr;
            const __amdgpu_buffer_rsrc_t rxa = nx.seg ? ra1 : ra0, rxb = nx.seg ? rb1 : rb0;
            float* const nslot = lds + nx.slot * STAGE_FL;
            const float* const As = lds + st.slot * STAGE_FL;
            const int s_next = st.slot == NSTAGE - 1 ? 0 : st.slot + 1;
            if (NT < 2) {                                   // too few MFMAs in block 0 to hide the offsets behind
                pr = prep_head(nx, more && !(g.dbg & 1));
#pragma unroll
                for (int d = 0; d < 6; ++d) prep_one(pr, vo, d);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int cur = j & 1;
                if (j == 3) {
                    // This wave's DMAs of the next step have landed: nothing younger is outstanding except, in the first step after an
                    // epilogue, that chunk's 16 stores (issued behind those DMAs).
                    if (since_store == 0) __builtin_amdgcn_s_waitcnt(0x0F70 | (16 & 15) | ((16 >> 4) << 14));       // vmcnt(16)
                    else __builtin_amdgcn_s_waitcnt(0x0F70);                                                        // vmcnt(0)
                    // Bare barrier: __syncthreads() carries a workgroup-scope release, which the compiler implements by draining EVERY
                    // outstanding LDS-DMA.  The asm is opaque (no LDS access moves across it); what it orders is spelled out above.
                    if (!(g.dbg & 2)) asm volatile("s_barrier" ::: "memory");
                    since_store = 1;
                }
                const float* const nAs = j == 3 ? lds + s_next * STAGE_FL : As;      // where the next block's fragments come from
                if (j < 3 || kt + 1 < KT) read_frags(nAs, j == 3 ? 0 : j + 1, cur ^ 1);
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    if (j == 3) {                           // 2, 2, 1, 1 DMAs in front of the block's four MFMA groups
                        const int d0 = s < 2 ? 2 * s : 2 + s, d1 = s < 2 ? d0 + 2 : d0 + 1;
#pragma unroll
                        for (int d = d0; d < d1; ++d) issue(rxa, rxb, nslot, vo, d);
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int tm = 0; tm < NT; ++tm) {
                        acc[tm] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cur][tm][s], bf[cur][s], acc[tm], 0, 0, 0);
                        if (j == 0 && NT >= 2) {            // the offsets' common part behind the first MFMA, one DMA offset behind each of the next 6
                            const int idx = s * NT + tm;
                            // (the empty asm pins each value HERE: LLVM's sink pass otherwise moves the whole computation down to its
                            // use, in front of block 3's MFMAs - sched_barrier only binds the machine scheduler)
                            if (idx == 0) {
                                pr = prep_head(nx, more && !(g.dbg & 1));
                                asm volatile("" : "+v"(pr.a_off), "+v"(pr.b_off), "+v"(pr.a_lim), "+v"(pr.b_lim));
                            }
                            if (idx >= 1 && idx <= 6) {
                                prep_one(pr, vo, idx - 1);
                                asm volatile("" : "+v"(vo[idx - 1]));
                            }
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
            if (more) { advance(cnext); ++issued; }
            if (kt + 1 < KT) {
                Cursor t = {ccur.c, kt + 1, s_next};
                st = step_at(t);
            }
        }
        // ---- epilogue through the slot just consumed (wave-private 32 x 36 scratch): 16-byte row-contiguous stores, ALWAYS 16 per wave
        // (out-of-range ones are dropped by the range check) because the wait count of the next step relies on the number.
        asm volatile("s_barrier" ::: "memory");            // every wave is past its fragment reads of the last step: the slot becomes scratch
        float* const sc = lds + st.slot * STAGE_FL + wave * (32 * EP);
#pragma unroll
        for (int tm = 0; tm < 4; ++tm) {
            if (tm < NT) {
#pragma unroll
                for (int r = 0; r < 16; ++r) sc[((r & 3) + 8 * (r >> 2) + 4 * h) * EP + m] = acc[tm < NT ? tm : 0][r];
                __builtin_amdgcn_s_waitcnt(0xC07F);                               // lgkmcnt(0): same wave wrote, same wave reads
            }
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const int lrow = srow + 8 * p;
                u32x4 raw = (u32x4){0u, 0u, 0u, 0u};
                unsigned vo = OOB;
                if (tm < NT) {
                    f32x4 v = *(const f32x4*)(sc + lrow * EP + col4);
                    v += bv;
                    if (g.relu && !slabbed) {
#pragma unroll
                        for (int k = 0; k < 4; ++k) v[k] = fmaxf(v[k], 0.f);
                    }
#pragma unroll
                    for (int k = 0; k < 4; ++k) raw[k] = __float_as_uint(v[k]);
                    const int trow = 32 * (toff + tm) + lrow;
                    const int orow = slabbed ? trow : st.row0 + trow;
                    const bool ok = ocol < ncols && (slabbed || orow < g.M);
                    vo = ok ? (unsigned)(orow * ld_out + ocol) * 4u : OOB;
                }
                __builtin_amdgcn_raw_buffer_store_b128(raw, rc, vo, 0, 0);
            }
            if (tm < NT) __builtin_amdgcn_s_waitcnt(0xC07F);                      // the reads are done before the next tile overwrites the scratch
        }
        since_store = 0;
        ccur.slot = st.slot == NSTAGE - 1 ? 0 : st.slot + 1;
        ++ccur.c;
    };

    for (int c = 0; c < nch; ++c) {
        const int nt = cbase + (c < cext ? 1 : 0), na = (nt + 1) >> 1;
        const int mine = kh ? nt - na : na, toff = kh ? na : 0;
        switch (mine) {
            case 0: run_chunk(std::integral_constant<int, 0>{}, toff); break;
            case 1: run_chunk(std::integral_constant<int, 1>{}, toff); break;
            case 2: run_chunk(std::integral_constant<int, 2>{}, toff); break;
            case 3: run_chunk(std::integral_constant<int, 3>{}, toff); break;
            default: run_chunk(std::integral_constant<int, 4>{}, toff); break;
        }
    }
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------------- host side
namespace vocr_dma_gemm {

static int cu_count() {
    static int cus[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    if (!cus[dev]) {
        hipDeviceProp_t prop;
        cus[dev] = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
    }
    return cus[dev];
}

// Shapes this kernel takes: 16-byte aligned operands with leading dimensions, K and the contiguous extent multiples of 4, at least two
// column panels and enough rows per workgroup to amortise the weight panel, buffers below 2 GiB.
Plan plan(int transa, int transb, int m, int n, int k, int lda, int ldb, int ldc, int nprob, int nseg, bool no_split) {
    Plan p = {false, 0, 0, 1, 0, 0};
    static const int on = VOCR_EXPERIMENT_INT("VOCR_GEMM_DMA", 1);
    if (!on) return p;
    if (m < 256 || n < 128 || k < 64) return p;
    if ((lda | ldb | ldc | k | n) & 3) return p;
    if (transa && (m & 3)) return p;
    const long a_ext = transa ? (long)k * lda : (long)m * lda, b_ext = transb ? (long)n * ldb : (long)k * ldb, c_ext = (long)m * ldc;
    if (a_ext * 4 >= (1l << 31) || b_ext * 4 >= (1l << 31) || c_ext * 4 >= (1l << 31)) return p;
    const int ncu = cu_count();
    p.panels = vocr_cdiv(n, DN);
    const int mt = vocr_cdiv(m, 32);
    const int units = nprob * p.panels;
    int groups = ncu / units;
    if (groups < 1) groups = 1;
    if (groups > mt) groups = mt;
    p.ksplit = 1;
    p.kps = vocr_cdiv(k, DK) * DK;
    if (mt / groups < 8 && (long)k * nseg >= 2048 && !no_split) {
        // few rows per workgroup but a long K (weight gradients; the data gradient of a narrow layer): whole 256-row groups, K cut into
        // slabs.  With two K segments every slab covers the same k range of BOTH segments.
        groups = vocr_cdiv(mt, 8);
        int ks = ncu / (units * groups);
        if (ks > k / 256) ks = k / 256;
        if (ks < 1) ks = 1;
        p.kps = vocr_cdiv(vocr_cdiv(k, ks), DK) * DK;
        p.ksplit = vocr_cdiv(k, p.kps);
        if (p.ksplit == 1) { groups = ncu / units < 1 ? 1 : (ncu / units > mt ? mt : ncu / units); }      // K too short to cut after all
    }
    if (n < 256 && p.ksplit == 1) return p;            // a single column panel only pays when K is cut (else gemm.hip's tiles do better)
    if (mt / groups < 2) return p;                     // too few rows per workgroup: gemm.hip's tiles do better
    p.groups = groups;
    p.slab_bytes = p.ksplit > 1 ? (size_t)nprob * p.groups * p.panels * p.ksplit * DM * DN * sizeof(float) : 0;
    p.ok = true;
    return p;
}

int launch(const Plan& p, int transa, int transb, int m, int n, int k, const float* const a[2], int lda, const float* const b[2], int ldb,
           float* const c[2], int ldc, const float* const bias[2], int relu, int nprob, int nseg, float* slab, hipStream_t s) {
    DmaGemmArgs g;
    for (int i = 0; i < 2; ++i) {
        g.a[i] = a[i] ? a[i] : a[0];
        g.b[i] = b[i] ? b[i] : b[0];
        g.c[i] = c[i] ? c[i] : c[0];
        g.bias[i] = bias[i];
    }
    g.a_bytes = (unsigned)(((transa ? (long)(k - 1) * lda + m : (long)(m - 1) * lda + k)) * 4);
    g.b_bytes = (unsigned)(((transb ? (long)(n - 1) * ldb + k : (long)(k - 1) * ldb + n)) * 4);
    g.c_bytes = (unsigned)(((long)(m - 1) * ldc + n) * 4);
    g.M = m; g.N = n; g.K = k;
    g.lda = lda; g.ldb = ldb; g.ldc = ldc;
    g.nprob = nprob; g.nseg = nseg;
    g.panels = p.panels; g.groups = p.groups; g.ksplit = p.ksplit; g.kps = p.kps;
    g.relu = relu;
#ifdef VOCR_GEMM_DIAG      // diagnostic builds only (scripts/gemm_dbg.py): parts of the loop switched off, WRONG results - never in libvocr.so
    static const int dbg = getenv("VOCR_GEMM_DBG") ? atoi(getenv("VOCR_GEMM_DBG")) : 0;
    g.dbg = dbg;
#else
    g.dbg = 0;
#endif
    g.slab = p.ksplit > 1 ? slab : nullptr;
    const dim3 grid(nprob * p.panels * p.groups * p.ksplit);
    if (!transa && transb) gemm_dma_kernel<true, true><<<grid, 512, 0, s>>>(g);
    else if (!transa && !transb) gemm_dma_kernel<true, false><<<grid, 512, 0, s>>>(g);
    else if (transa && !transb) gemm_dma_kernel<false, false><<<grid, 512, 0, s>>>(g);
    else gemm_dma_kernel<false, true><<<grid, 512, 0, s>>>(g);
    return 0;
}

}  // namespace vocr_dma_gemm
