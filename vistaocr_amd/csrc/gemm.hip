// fp32 MFMA GEMM for the dense layers of the path (bridge Linear, LSTM input projections, prob Linear and
// their backward).  C[M,N] = op(A)[M,K] * op(B)[K,N] (+bias)(relu)(+=).
//
// gfx950 mapping: v_mfma_f32_32x32x2_f32 (exact f32, 64 FLOP/clk/SIMD = the chip's 157 TF f32 peak).
// One workgroup = 4 waves (one per SIMD); LDS holds K-major tiles As[k][m], Bs[k][n] so a fragment read
// is 32 consecutive dwords per lane-half (conflict-free ds_read_b32); global->LDS goes through registers
// and is issued one K-tile ahead of the MFMAs that consume it.
#include "vocr_common.h"
#include "gemm_dma.h"

namespace {

constexpr int BK = 32;

// Out-of-range operands are read from a zero page by ADDRESS select (the load stays unconditional, so the whole
// K-tile's loads are in flight together; a select on the loaded value turns into a branch + vmcnt(0) per load).
__device__ __attribute__((aligned(16))) float g_gemm_zero_page[64];

// VEC: operands are fetched with 16-byte loads along their contiguous dimension (needs 16-B aligned bases,
// leading dimensions and M/N/K multiples of 4); otherwise 4-byte loads.
template <int BM, int BN, int WM, int WN, bool A_KCONTIG, bool B_NCONTIG, bool VEC>
__global__ __launch_bounds__(256) void gemm_f32_kernel(
    int M, int N, int K, const float* __restrict__ A, long sam, long sak, const float* __restrict__ B, long sbk,
    long sbn, float* __restrict__ C, int ldc, const float* __restrict__ bias, int relu, int accumulate,
    int k_per_split, const float* __restrict__ zp, float* __restrict__ slab, int n_whole, int pieces_per_tile, int tiles_n) {
    // LDS tiles are K-major: As[k][m], Bs[k][n].  Pitch ≡ 2 (mod 32) keeps the transposing scalar stores of a
    // k-contiguous operand at most 2-way conflicted; a multiple of 4 keeps 16-byte stores of an m-contiguous one aligned.
    constexpr int PA = BM + ((VEC && !A_KCONTIG) ? 4 : 2);
    constexpr int PB = BN + ((VEC && B_NCONTIG) ? 4 : 2);
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int WAVES_N = BN / WN;
    constexpr int EA = BM * BK / 256, EB = BN * BK / 256;      // floats per thread per tile
    __shared__ __attribute__((aligned(16))) float lds[2 * BK * (PA + PB)];
    float* const As0 = lds;
    float* const Bs0 = lds + 2 * BK * PA;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    // 1-D ragged grid: the first n_whole workgroups each compute one whole output tile (in XCD-sliced order: an XCD works on
    // a contiguous run of tiles, so the A row panel they share sits in ONE L2); the remaining tiles - fewer than one round
    // of resident workgroups - are cut along K into pieces_per_tile short pieces that are dispatched LAST and fill the tail
    // of the launch; a piece stores its partial tile into its own slab and splitk_reduce_kernel adds a tile's slabs in
    // piece order (n_whole = 0: classic split-K of every tile, used for the long-K weight gradients).
    int tile, kbeg, kend, piece = -1;
    if ((int)blockIdx.x < n_whole) {
        const int x = blockIdx.x & 7, q = n_whole >> 3, r = n_whole & 7;
        tile = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (blockIdx.x >> 3);
        kbeg = 0;
        kend = K;
    } else {
        piece = blockIdx.x - n_whole;
        tile = n_whole + piece / pieces_per_tile;
        kbeg = (piece % pieces_per_tile) * k_per_split;
        kend = min(K, kbeg + k_per_split);
    }
    const int m0 = (tile / tiles_n) * BM, n0 = (tile % tiles_n) * BN;
    const int wm0 = (wave / WAVES_N) * WM, wn0 = (wave % WAVES_N) * WN;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    float ra[EA], rb[EB];

    // element maps: (row r along M or N, k) per thread and pass
    auto load_tiles = [&](int k0) {
        if (VEC) {
#pragma unroll
            for (int e = 0; e < EA / 4; ++e) {
                const float* src;
                bool ok;
                if (A_KCONTIG) { const int m = (tid >> 3) + 32 * e, k = (tid & 7) * 4; ok = m0 + m < M && k0 + k < kend; src = A + (long)(m0 + m) * sam + (k0 + k); }
                else           { const int m = (tid % (BM / 4)) * 4, k = tid / (BM / 4) + (1024 / BM) * e; ok = m0 + m < M && k0 + k < kend; src = A + (long)(k0 + k) * sak + (m0 + m); }
                const f32x4 v = *(const f32x4*)(ok ? src : zp);
                ra[4 * e] = v[0]; ra[4 * e + 1] = v[1]; ra[4 * e + 2] = v[2]; ra[4 * e + 3] = v[3];
            }
#pragma unroll
            for (int e = 0; e < EB / 4; ++e) {
                const float* src;
                bool ok;
                if (!B_NCONTIG) { const int n = (tid >> 3) + 32 * e, k = (tid & 7) * 4; ok = n0 + n < N && k0 + k < kend; src = B + (long)(n0 + n) * sbn + (k0 + k); }
                else            { const int n = (tid % (BN / 4)) * 4, k = tid / (BN / 4) + (1024 / BN) * e; ok = n0 + n < N && k0 + k < kend; src = B + (long)(k0 + k) * sbk + (n0 + n); }
                const f32x4 v = *(const f32x4*)(ok ? src : zp);
                rb[4 * e] = v[0]; rb[4 * e + 1] = v[1]; rb[4 * e + 2] = v[2]; rb[4 * e + 3] = v[3];
            }
        } else {
#pragma unroll
            for (int e = 0; e < EA; ++e) {
                int m, k;
                if (A_KCONTIG) { k = tid & 31; m = (tid >> 5) + 8 * e; }
                else           { m = tid % BM; k = tid / BM + (256 / BM) * e; }
                const bool ok = m0 + m < M && k0 + k < kend;
                const float* src = A + (long)(m0 + m) * sam + (long)(k0 + k) * sak;
                ra[e] = *(ok ? src : zp);
            }
#pragma unroll
            for (int e = 0; e < EB; ++e) {
                int n, k;
                if (B_NCONTIG) { n = tid % BN; k = tid / BN + (256 / BN) * e; }
                else           { k = tid & 31; n = (tid >> 5) + 8 * e; }
                const bool ok = n0 + n < N && k0 + k < kend;
                const float* src = B + (long)(k0 + k) * sbk + (long)(n0 + n) * sbn;
                rb[e] = *(ok ? src : zp);
            }
        }
    };
    auto store_tiles = [&](int buf) {
        float* as = As0 + buf * BK * PA;
        float* bs = Bs0 + buf * BK * PB;
        if (VEC) {
#pragma unroll
            for (int e = 0; e < EA / 4; ++e) {
                if (A_KCONTIG) {
                    const int m = (tid >> 3) + 32 * e, k = (tid & 7) * 4;
#pragma unroll
                    for (int x = 0; x < 4; ++x) as[(k + x) * PA + m] = ra[4 * e + x];
                } else {
                    const int m = (tid % (BM / 4)) * 4, k = tid / (BM / 4) + (1024 / BM) * e;
                    *(f32x4*)(as + k * PA + m) = (f32x4){ra[4 * e], ra[4 * e + 1], ra[4 * e + 2], ra[4 * e + 3]};
                }
            }
#pragma unroll
            for (int e = 0; e < EB / 4; ++e) {
                if (!B_NCONTIG) {
                    const int n = (tid >> 3) + 32 * e, k = (tid & 7) * 4;
#pragma unroll
                    for (int x = 0; x < 4; ++x) bs[(k + x) * PB + n] = rb[4 * e + x];
                } else {
                    const int n = (tid % (BN / 4)) * 4, k = tid / (BN / 4) + (1024 / BN) * e;
                    *(f32x4*)(bs + k * PB + n) = (f32x4){rb[4 * e], rb[4 * e + 1], rb[4 * e + 2], rb[4 * e + 3]};
                }
            }
        } else {
#pragma unroll
            for (int e = 0; e < EA; ++e) {
                int m, k;
                if (A_KCONTIG) { k = tid & 31; m = (tid >> 5) + 8 * e; }
                else           { m = tid % BM; k = tid / BM + (256 / BM) * e; }
                as[k * PA + m] = ra[e];
            }
#pragma unroll
            for (int e = 0; e < EB; ++e) {
                int n, k;
                if (B_NCONTIG) { n = tid % BN; k = tid / BN + (256 / BN) * e; }
                else           { k = tid & 31; n = (tid >> 5) + 8 * e; }
                bs[k * PB + n] = rb[e];
            }
        }
    };

    const int nk = (kend - kbeg + BK - 1) / BK;
    if (nk > 0) {
        load_tiles(kbeg);
        store_tiles(0);
    }
    __syncthreads();
    const int li = lane & 31, lk = lane >> 5;
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) load_tiles(kbeg + (kt + 1) * BK);
        const float* as = As0 + cur * BK * PA + wm0 + li + lk * PA;
        const float* bs = Bs0 + cur * BK * PB + wn0 + li + lk * PB;
#pragma unroll
        for (int ks = 0; ks < BK / 2; ++ks) {
            float af[TM], bf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i] = as[ks * 2 * PA + i * 32];
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[j] = bs[ks * 2 * PB + j * 32];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < nk) store_tiles(cur ^ 1);
        __syncthreads();
    }

    // epilogue: C/D map of 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    float* const sl = piece >= 0 ? slab + (long)piece * BM * BN : nullptr;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int ln = wn0 + j * 32 + li, gn = n0 + ln;
            if (piece >= 0) {                       // partial tile: tile-local [BM][BN] slab, every element (the reduce masks the edges)
#pragma unroll
                for (int r = 0; r < 16; ++r) sl[(wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk) * BN + ln] = acc[i][j][r];
                continue;
            }
            if (gn >= N) continue;
            const float bv = bias ? bias[gn] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int gm = m0 + wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
                if (gm >= M) continue;
                float v = acc[i][j][r] + bv;
                float* cp = C + (long)gm * ldc + gn;
                if (accumulate) v += *cp;
                if (relu) v = fmaxf(v, 0.f);
                *cp = v;
            }
        }
}

template <int BM, int BN, int WM, int WN, bool VEC>
void launch_cfg(int ta, int tb, dim3 grid, hipStream_t s, int M, int N, int K, const float* A, long sam, long sak,
                const float* B, long sbk, long sbn, float* C, int ldc, const float* bias, int relu, int acc, int kps,
                const float* zp, float* slab, int n_whole, int ppt, int tiles_n) {
    if (!ta && !tb)
        gemm_f32_kernel<BM, BN, WM, WN, true, true, VEC><<<grid, 256, 0, s>>>(M, N, K, A, sam, sak, B, sbk, sbn, C, ldc, bias, relu, acc, kps, zp, slab, n_whole, ppt, tiles_n);
    else if (!ta && tb)
        gemm_f32_kernel<BM, BN, WM, WN, true, false, VEC><<<grid, 256, 0, s>>>(M, N, K, A, sam, sak, B, sbk, sbn, C, ldc, bias, relu, acc, kps, zp, slab, n_whole, ppt, tiles_n);
    else if (ta && !tb)
        gemm_f32_kernel<BM, BN, WM, WN, false, true, VEC><<<grid, 256, 0, s>>>(M, N, K, A, sam, sak, B, sbk, sbn, C, ldc, bias, relu, acc, kps, zp, slab, n_whole, ppt, tiles_n);
    else
        gemm_f32_kernel<BM, BN, WM, WN, false, false, VEC><<<grid, 256, 0, s>>>(M, N, K, A, sam, sak, B, sbk, sbn, C, ldc, bias, relu, acc, kps, zp, slab, n_whole, ppt, tiles_n);
}

const float* gemm_zero_page() {
    static const float* zp[64] = {nullptr};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    if (!zp[dev]) {
        void* p = nullptr;
        if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_gemm_zero_page)) != hipSuccess) return nullptr;
        zp[dev] = (const float*)p;
    }
    return zp[dev];
}

// The tiles that were cut along K: C tile (+)= sum over its pieces' slabs, added in piece order (bitwise reproducible;
// float atomics into C were neither reproducible nor faster: 1.3 TB/s), then bias / ReLU.  grid.x = cut tiles x
// (BM*BN/1024) blocks: one 16-byte column group per thread, four pieces' loads in flight.  (One block per tile with a
// serial piece loop left each thread with a single load in flight: 120-345 us for 14-16 tiles on a busy chip.)
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ slab, float* __restrict__ C, int M, int N,
                                                            int ldc, int BM, int BN, int first_tile, int pieces_per_tile,
                                                            int tiles_n, const float* __restrict__ bias, int relu, int accumulate) {
    const int cols4 = BN >> 2;
    const int sub = (BM * cols4) >> 8;                       // blocks per tile
    const int t = blockIdx.x / sub, e = (blockIdx.x - t * sub) * 256 + threadIdx.x;
    const int tile = first_tile + t;
    const int m0 = (tile / tiles_n) * BM, n0 = (tile % tiles_n) * BN;
    const long tsz = (long)BM * BN;
    const int r = e / cols4, c = (e - r * cols4) << 2;
    const float* sp = slab + (long)t * pieces_per_tile * tsz + r * BN + c;
    f32x4 v = *(const f32x4*)sp;
    int z = 1;
    for (; z + 4 <= pieces_per_tile; z += 4) {
        const f32x4 a = *(const f32x4*)(sp + (long)z * tsz), b = *(const f32x4*)(sp + (long)(z + 1) * tsz);
        const f32x4 c2 = *(const f32x4*)(sp + (long)(z + 2) * tsz), d = *(const f32x4*)(sp + (long)(z + 3) * tsz);
        v += a; v += b; v += c2; v += d;                    // same order as a serial loop
    }
    for (; z < pieces_per_tile; ++z) v += *(const f32x4*)(sp + (long)z * tsz);
    const int gm = m0 + r;
    if (gm >= M) return;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int gn = n0 + c + k;
        if (gn >= N) continue;
        float x = v[k] + (bias ? bias[gn] : 0.f);
        float* cp = C + (long)gm * ldc + gn;
        if (accumulate) x += *cp;
        if (relu) x = fmaxf(x, 0.f);
        *cp = x;
    }
}

// partial[split][n] = sum of rows [r0, r1) of x[:, n]; grid (N/64 column tiles, row splits).  With one split the sum goes
// straight to out; otherwise colsum_final_kernel adds the partial rows in split order (no float atomics: reproducible).
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ x, float* __restrict__ out, int M, int N,
                                                     int rows_per_split) {
    __shared__ float red[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = blockIdx.x * 64 + lane;
    const int r0 = blockIdx.y * rows_per_split, r1 = min(M, r0 + rows_per_split);
    float s0 = 0.f, s1 = 0.f;
    if (n < N) {
        int m = r0 + wave;
        for (; m + 4 < r1; m += 8) { s0 += x[(long)m * N + n]; s1 += x[(long)(m + 4) * N + n]; }
        if (m < r1) s0 += x[(long)m * N + n];
    }
    red[wave][lane] = s0 + s1;
    __syncthreads();
    if (wave == 0 && n < N) out[(long)blockIdx.y * N + n] = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
}

__global__ void colsum_final_kernel(const float* __restrict__ part, float* __restrict__ out, int N, int splits) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    float v = part[n];
    for (int z = 1; z < splits; ++z) v += part[(long)z * N + n];
    out[n] = v;
}

// elementwise glue: 16-byte main loop (count4 = count/4 vectors) + scalar tail, grid-stride
__global__ void relu_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ out, float* __restrict__ dz, size_t n, int vec) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t n4 = vec ? n / 4 : 0;
    for (size_t v = i; v < n4; v += stride) {
        const f32x4 d = ((const f32x4*)dy)[v], o = ((const f32x4*)out)[v];
        f32x4 r;
#pragma unroll
        for (int k = 0; k < 4; ++k) r[k] = o[k] > 0.f ? d[k] : 0.f;
        ((f32x4*)dz)[v] = r;
    }
    for (size_t e = n4 * 4 + i; e < n; e += stride) dz[e] = out[e] > 0.f ? dy[e] : 0.f;
}

__global__ void mul_kernel(const float* __restrict__ x, const float* __restrict__ m, float* __restrict__ o, size_t n, int vec) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t n4 = vec ? n / 4 : 0;
    for (size_t v = i; v < n4; v += stride) ((f32x4*)o)[v] = ((const f32x4*)x)[v] * ((const f32x4*)m)[v];
    for (size_t e = n4 * 4 + i; e < n; e += stride) o[e] = x[e] * m[e];
}

__global__ void add_kernel(const float* __restrict__ x, const float* __restrict__ y, float* __restrict__ o, size_t n, int vec) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t n4 = vec ? n / 4 : 0;
    for (size_t v = i; v < n4; v += stride) ((f32x4*)o)[v] = ((const f32x4*)x)[v] + ((const f32x4*)y)[v];
    for (size_t e = n4 * 4 + i; e < n; e += stride) o[e] = x[e] + y[e];
}

__global__ void scale_dev_kernel(const float* __restrict__ x, const float* __restrict__ sc, float* __restrict__ o, size_t n, int vec) {
    const float a = sc[0];
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t n4 = vec ? n / 4 : 0;
    for (size_t v = i; v < n4; v += stride) ((f32x4*)o)[v] = ((const f32x4*)x)[v] * a;
    for (size_t e = n4 * 4 + i; e < n; e += stride) o[e] = x[e] * a;
}

__device__ __forceinline__ uint32_t mix32(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;   // splitmix64 finaliser: counter-based, one draw per element
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    z = z ^ (z >> 31);
    return (uint32_t)(z >> 32);
}

__global__ void dropout_kernel(const float* __restrict__ x, float* __restrict__ o, float* __restrict__ mask, size_t n,
                               float p, float scale, uint64_t seed, int vec) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t n4 = vec ? n / 4 : 0;
    auto draw = [&](size_t e) {
        const float u = (float)(mix32(seed * 0x9e3779b97f4a7c15ull + e) >> 8) * (1.0f / 16777216.0f);
        return (u >= p) ? scale : 0.f;
    };
    for (size_t v = i; v < n4; v += stride) {
        const f32x4 xv = ((const f32x4*)x)[v];
        f32x4 mk;
#pragma unroll
        for (int k = 0; k < 4; ++k) mk[k] = draw(4 * v + k);
        ((f32x4*)mask)[v] = mk;
        ((f32x4*)o)[v] = xv * mk;
    }
    for (size_t e = n4 * 4 + i; e < n; e += stride) {
        const float mk = draw(e);
        mask[e] = mk;
        o[e] = x[e] * mk;
    }
}

__global__ void dropout_mask_kernel(float* __restrict__ mask, size_t n, float p, float scale, uint64_t seed) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += stride) {
        const float u = (float)(mix32(seed * 0x9e3779b97f4a7c15ull + e) >> 8) * (1.0f / 16777216.0f);
        mask[e] = (u >= p) ? scale : 0.f;
    }
}

// bchw -> [w][b][c*h]: one workgroup transposes a 32(w) x 64(ch) tile of one image through LDS so both the read (along w) and the
// write (along c*h) are coalesced.  8.3 KB of LDS (round 5: a 64 x 64 tile, 16.6 KB): in the step the backward permute runs on the main
// stream while the panel GEMM of the layer-0 LSTM weight gradients holds every CU with 144 of its 160 KB - the 16.6-KB tile did not fit
// beside it and the 30-us pass took 340 us in front of the CNN backward
__global__ void bchw_to_wbch_kernel(const float* __restrict__ x, float* __restrict__ out, int B, int CH, int W, int fwd) {
    __shared__ float tile[32][65];                            // [w][ch]
    const int b = blockIdx.z, ch0 = blockIdx.y * 64, w0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;   // 64 channels x 4 columns
    const int wx = threadIdx.x & 31, wy = threadIdx.x >> 5;   // 32 columns x 8 channels
    if (fwd) {
        for (int r = wy; r < 64; r += 8) {
            const int ch = ch0 + r, w = w0 + wx;
            tile[wx][r] = (ch < CH && w < W) ? x[((long)b * CH + ch) * W + w] : 0.f;
        }
        __syncthreads();
        for (int r = ty; r < 32; r += 4) {
            const int w = w0 + r, ch = ch0 + tx;
            if (w < W && ch < CH) out[((long)w * B + b) * CH + ch] = tile[r][tx];
        }
    } else {
        for (int r = ty; r < 32; r += 4) {
            const int w = w0 + r, ch = ch0 + tx;
            tile[r][tx] = (w < W && ch < CH) ? x[((long)w * B + b) * CH + ch] : 0.f;
        }
        __syncthreads();
        for (int r = wy; r < 64; r += 8) {
            const int ch = ch0 + r, w = w0 + wx;
            if (ch < CH && w < W) out[((long)b * CH + ch) * W + w] = tile[wx][r];
        }
    }
}

}  // namespace

namespace {
struct GemmPlan { bool big, half; int bm, bn, tiles_m, tiles_n, n_whole, ppt, kps, pieces; };

int gemm_cu_count() {
    static int cus[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    if (!cus[dev]) {
        hipDeviceProp_t prop;
        cus[dev] = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
    }
    return cus[dev];
}

// Tile shape and K cuts.  max_pieces_128 = how many 128x128 slabs the workspace can hold (0: none, K is never cut).
GemmPlan gemm_plan(int m, int n, int k, bool has_epilogue, long max_pieces_128, bool allow_tail_fill) {
    GemmPlan p;
    const long tiles128 = (long)vocr_cdiv(m, 128) * vocr_cdiv(n, 128);
    const bool can_split = !has_epilogue && k >= 1024 && max_pieces_128 > 1;
    // 128x128 tiles (32 FLOP per LDS-staged byte) whenever they can fill the chip, possibly with split-K
    p.big = m >= 96 && n >= 96 && (tiles128 >= 192 || (can_split && tiles128 * (k / 512) >= 128));
    // 128x64 tiles when there are fewer than 6 full tiles per CU and no split-K: twice as many half-size workgroups let
    // the dispatcher balance the last partial round (measured 104 TF at 4.6 tiles/CU vs 118 TF at exactly 4)
    p.half = p.big && tiles128 < 6 * 256 && !(can_split && tiles128 < 384) && n >= 64;
    // ... unless K is long and the last partial round can be cut into K pieces (below): whole 128x128 tiles + tail pieces measured
    // 390 vs 411 us on the 9408x2048x1024 projection and 354 vs 378 us on its data gradient (scripts/gemm_bench.py); with
    // K = 128 the half-size tiles stay ahead (73 vs 80 us)
    static const int half_mode = VOCR_EXPERIMENT_INT("VOCR_GEMM_HALF", -1);      // experiments: 0 never, 1 always as above
    if (half_mode == 0 || (half_mode < 0 && allow_tail_fill && max_pieces_128 > 1 && k >= 512 && tiles128 >= 2l * gemm_cu_count())) p.half = false;
    p.bm = p.big ? 128 : 64;
    p.bn = p.big ? (p.half ? 64 : 128) : 64;
    p.tiles_m = vocr_cdiv(m, p.bm);
    p.tiles_n = vocr_cdiv(n, p.bn);
    const long tiles = (long)p.tiles_m * p.tiles_n;
    const long max_pieces = max_pieces_128 * (128l * 128) / ((long)p.bm * p.bn);
    const int kt = vocr_cdiv(k, BK);                          // K tiles of one output tile
    p.n_whole = (int)tiles;
    p.ppt = 1;
    p.kps = kt * BK;
    if (can_split && tiles < 384) {                           // few tiles, long K: cut every tile (weight gradients)
        int splits = (int)((512 + tiles - 1) / tiles);
        const int maxs = k / 512;
        if (splits > maxs) splits = maxs;
        if ((long)splits * tiles > max_pieces) splits = (int)(max_pieces / tiles);
        if (splits > 1) {
            p.kps = vocr_cdiv(kt, splits) * BK;
            p.ppt = vocr_cdiv(k, p.kps);
            p.n_whole = 0;
        }
    } else if (allow_tail_fill && max_pieces > 1 && kt >= 4) {
        // many tiles: whole tiles for every full round of resident workgroups, the last partial round cut into pieces
        const int occ = p.bm == 64 ? 4 : (p.bn == 64 ? 3 : 2);                  // workgroups per CU (LDS: 34 / 50 / 66.5 KB)
        const long slots = (long)gemm_cu_count() * occ;
        const long left = tiles % slots;
        if (tiles > slots && left > 0 && left * 10 < slots * 9) {
            long ppt = slots / left;
            if (ppt > kt / 2) ppt = kt / 2;                                      // a piece covers at least two K tiles
            if (ppt * left > max_pieces) ppt = max_pieces / left;
            if (ppt > 1) {
                p.kps = vocr_cdiv(kt, (int)ppt) * BK;
                p.ppt = vocr_cdiv(k, p.kps);
                p.n_whole = (int)(tiles - left);
            }
        }
    }
    if (p.ppt <= 1) { p.n_whole = (int)tiles; p.ppt = 1; p.kps = kt * BK; }
    p.pieces = p.ppt > 1 ? (int)(tiles - p.n_whole) * p.ppt : 0;
    return p;
}

bool gemm_tail_fill_enabled() {
    // (with half-size tiles the K pieces bought nothing - 9 tiles per CU balance the last round by themselves; they pay with whole
    // 128x128 tiles, see gemm_plan)
    static const int on = VOCR_EXPERIMENT_INT("VOCR_GEMM_TAILFILL", 1);
    return on != 0;
}
}  // namespace

extern "C" size_t vocr_gemm_workspace_bytes(int m, int n, int k, int has_bias_or_relu) {
    if (m <= 0 || n <= 0 || k <= 0) return 0;
    const GemmPlan p = gemm_plan(m, n, k, has_bias_or_relu != 0, 1l << 40, gemm_tail_fill_enabled());
    size_t bytes = (size_t)p.pieces * p.bm * p.bn * sizeof(float);
    // the DMA-staged kernel's K slabs (whatever the transposition flags and leading dimensions turn out to be: densest case)
    for (int ta = 0; ta < 2; ++ta) {
        const vocr_dma_gemm::Plan d = vocr_dma_gemm::plan(ta, 0, m, n, k, ta ? m : k, n, n, 1, 1, false);
        if (d.ok && d.slab_bytes > bytes) bytes = d.slab_bytes;
    }
    return bytes;
}

// Two products of one shape in one launch: slabs for both
extern "C" size_t vocr_gemm_pair_workspace_bytes(int m, int n, int k, int mode) {
    if (m <= 0 || n <= 0 || k <= 0) return 0;
    size_t bytes = vocr_gemm_workspace_bytes(m, n, k, 0);
    for (int ta = 0; ta < 2; ++ta) {
        const vocr_dma_gemm::Plan d = vocr_dma_gemm::plan(ta, 0, m, n, k, ta ? m : k, n, n, mode ? 1 : 2, mode ? 2 : 1, false);
        if (d.ok && d.slab_bytes > bytes) bytes = d.slab_bytes;
    }
    return bytes;
}

static thread_local bool g_gemm_tiles_only = false;      // set by vocr_gemm_pair's co-scheduling hint around its two vocr_gemm calls

extern "C" int vocr_gemm(int transa, int transb, int m, int n, int k, const float* a, int lda, const float* b, int ldb,
                         float* c, int ldc, const float* bias, int relu, int accumulate, void* workspace,
                         size_t workspace_bytes, void* stream) {
    VOCR_CHECK_ARG(m > 0 && n > 0 && k > 0, "vocr_gemm: bad shape m=%d n=%d k=%d", m, n, k);
    VOCR_CHECK_ARG(a && b && c, "vocr_gemm: null pointer");
    VOCR_CHECK_ARG(lda >= (transa ? m : k) && ldb >= (transb ? k : n) && ldc >= n, "vocr_gemm: bad leading dimension");
    hipStream_t s = (hipStream_t)stream;
    // large, aligned products: the DMA-staged panel kernel (gemm_dma.hip); `accumulate` only through its slab reduce
    if (!g_gemm_tiles_only && ((((uintptr_t)a) | ((uintptr_t)b) | ((uintptr_t)c) | ((uintptr_t)bias)) & 15) == 0) {
        const vocr_dma_gemm::Plan d = vocr_dma_gemm::plan(transa, transb, m, n, k, lda, ldb, ldc, 1, 1, false);
        const bool ws_ok = d.ok && (d.ksplit == 1 || (workspace && (((uintptr_t)workspace) & 15) == 0 && workspace_bytes >= d.slab_bytes));
        if (ws_ok && (!accumulate || d.ksplit > 1)) {
            const float* aa[2] = {a, nullptr};
            const float* bb[2] = {b, nullptr};
            float* cc[2] = {c, nullptr};
            const float* bs[2] = {d.ksplit > 1 ? nullptr : bias, nullptr};
            vocr_dma_gemm::launch(d, transa, transb, m, n, k, aa, lda, bb, ldb, cc, ldc, bs, d.ksplit > 1 ? 0 : relu, 1, 1, (float*)workspace, s);
            VOCR_CHECK_LAUNCH("vocr_gemm(dma)");
            if (d.ksplit > 1) {
                const int tiles = d.groups * d.panels;
                splitk_reduce_kernel<<<tiles * 32, 256, 0, s>>>((const float*)workspace, c, m, n, ldc, 256, 128, 0, d.ksplit, d.panels, bias, relu, accumulate);
                VOCR_CHECK_LAUNCH("vocr_gemm(dma, split-K reduce)");
            }
            return VOCR_OK;
        }
    }
    const long sam = transa ? 1 : lda, sak = transa ? lda : 1;
    const long sbk = transb ? 1 : ldb, sbn = transb ? ldb : 1;
    // K cuts need room for one tile-sized slab per piece; with less workspace there are fewer pieces, with none K is never cut
    const long max_pieces_128 = (workspace && (((uintptr_t)workspace) & 15) == 0) ? (long)(workspace_bytes / (128 * 128 * sizeof(float))) : 0;
    const GemmPlan p = gemm_plan(m, n, k, bias != nullptr || relu != 0, max_pieces_128, gemm_tail_fill_enabled());
    const bool big = p.big, half = p.half;
    const int kps = p.kps;
    float* slab = p.pieces > 0 ? (float*)workspace : nullptr;
    dim3 grid(p.n_whole + p.pieces);
    const float* zp = gemm_zero_page();
    VOCR_CHECK_ARG(zp != nullptr, "vocr_gemm: no device zero page");
    const bool vec = ((((uintptr_t)a) | ((uintptr_t)b)) & 15) == 0 && lda % 4 == 0 && ldb % 4 == 0 && m % 4 == 0 && n % 4 == 0 &&
                     k % 4 == 0;
#define VOCR_GEMM_ARGS transa, transb, grid, s, m, n, k, a, sam, sak, b, sbk, sbn, c, ldc, bias, relu, accumulate, kps, zp, slab, p.n_whole, p.ppt, p.tiles_n
    if (big && half) {
        if (vec) launch_cfg<128, 64, 64, 32, true>(VOCR_GEMM_ARGS);
        else launch_cfg<128, 64, 64, 32, false>(VOCR_GEMM_ARGS);
    } else if (big) {
        if (vec) launch_cfg<128, 128, 64, 64, true>(VOCR_GEMM_ARGS);
        else launch_cfg<128, 128, 64, 64, false>(VOCR_GEMM_ARGS);
    } else {
        if (vec) launch_cfg<64, 64, 32, 32, true>(VOCR_GEMM_ARGS);
        else launch_cfg<64, 64, 32, 32, false>(VOCR_GEMM_ARGS);
    }
#undef VOCR_GEMM_ARGS
    VOCR_CHECK_LAUNCH("vocr_gemm");
    if (p.pieces > 0) {
        const int cut_tiles = p.pieces / p.ppt;
        splitk_reduce_kernel<<<cut_tiles * ((p.bm * p.bn) >> 10), 256, 0, s>>>(slab, c, m, n, ldc, p.bm, p.bn, p.n_whole, p.ppt, p.tiles_n, bias, relu, accumulate);
        VOCR_CHECK_LAUNCH("vocr_gemm(split-K reduce)");
    }
    return VOCR_OK;
}

extern "C" int vocr_gemm_pair(int mode, int transa, int transb, int m, int n, int k, const float* a0, const float* a1, int lda,
                              const float* b0, const float* b1, int ldb, float* c0, float* c1, int ldc, const float* bias0,
                              const float* bias1, int relu, void* workspace, size_t workspace_bytes, void* stream) {
    const bool co_sched = (mode & 4) != 0;           // hint: prefer the tile kernel, whose workgroups can share a CU with a persistent LSTM sweep
    mode &= ~4;
    VOCR_CHECK_ARG(mode == 0 || mode == 1, "vocr_gemm_pair: mode must be 0 (two products) or 1 (two K segments summed), optionally | 4");
    VOCR_CHECK_ARG(m > 0 && n > 0 && k > 0, "vocr_gemm_pair: bad shape m=%d n=%d k=%d", m, n, k);
    VOCR_CHECK_ARG(a0 && a1 && b0 && b1 && c0 && (mode == 1 || c1), "vocr_gemm_pair: null pointer");
    VOCR_CHECK_ARG(mode == 0 || (c1 == nullptr && bias1 == nullptr), "vocr_gemm_pair: mode 1 has one output and one bias");
    VOCR_CHECK_ARG(lda >= (transa ? m : k) && ldb >= (transb ? k : n) && ldc >= n, "vocr_gemm_pair: bad leading dimension");
    hipStream_t s = (hipStream_t)stream;
    const uintptr_t al = ((uintptr_t)a0) | ((uintptr_t)a1) | ((uintptr_t)b0) | ((uintptr_t)b1) | ((uintptr_t)c0) | ((uintptr_t)c1) |
                         ((uintptr_t)bias0) | ((uintptr_t)bias1);
    if ((al & 15) == 0 && !co_sched) {
        const int nprob = mode ? 1 : 2, nseg = mode ? 2 : 1;
        const vocr_dma_gemm::Plan d = vocr_dma_gemm::plan(transa, transb, m, n, k, lda, ldb, ldc, nprob, nseg, false);
        const bool ws_ok = d.ok && (d.ksplit == 1 || (workspace && (((uintptr_t)workspace) & 15) == 0 && workspace_bytes >= d.slab_bytes));
        if (ws_ok) {
            const float* aa[2] = {a0, a1};
            const float* bb[2] = {b0, b1};
            float* cc[2] = {c0, c1};
            const bool split = d.ksplit > 1;
            const float* bs[2] = {split ? nullptr : bias0, split ? nullptr : bias1};
            vocr_dma_gemm::launch(d, transa, transb, m, n, k, aa, lda, bb, ldb, cc, ldc, bs, split ? 0 : relu, nprob, nseg, (float*)workspace, s);
            VOCR_CHECK_LAUNCH("vocr_gemm_pair(dma)");
            if (split) {
                const int tiles = d.groups * d.panels;
                for (int p = 0; p < nprob; ++p) {
                    const float* sl = (const float*)workspace + (size_t)p * tiles * d.ksplit * 256 * 128;
                    splitk_reduce_kernel<<<tiles * 32, 256, 0, s>>>(sl, p ? c1 : c0, m, n, ldc, 256, 128, 0, d.ksplit, d.panels, p ? bias1 : bias0, relu, 0);
                    VOCR_CHECK_LAUNCH("vocr_gemm_pair(dma, split-K reduce)");
                }
            }
            return VOCR_OK;
        }
    }
    // shapes the panel kernel does not take (or the co-scheduling hint): two calls of the tile kernel (mode 1: the second one
    // accumulates, the ReLU comes last)
    struct TileOnly { TileOnly(bool on) { g_gemm_tiles_only = on; } ~TileOnly() { g_gemm_tiles_only = false; } } tile_only(co_sched);
    if (mode == 0) {
        int rc = vocr_gemm(transa, transb, m, n, k, a0, lda, b0, ldb, c0, ldc, bias0, relu, 0, workspace, workspace_bytes, stream);
        if (rc != VOCR_OK) return rc;
        return vocr_gemm(transa, transb, m, n, k, a1, lda, b1, ldb, c1, ldc, bias1, relu, 0, workspace, workspace_bytes, stream);
    }
    int rc = vocr_gemm(transa, transb, m, n, k, a0, lda, b0, ldb, c0, ldc, bias0, 0, 0, workspace, workspace_bytes, stream);
    if (rc != VOCR_OK) return rc;
    return vocr_gemm(transa, transb, m, n, k, a1, lda, b1, ldb, c0, ldc, nullptr, relu, 1, workspace, workspace_bytes, stream);
}

namespace {
int colsum_splits(int m, int n, int* rows_per_split) {
    const int ctiles = vocr_cdiv(n, 64);
    int splits = vocr_cdiv(1024, ctiles);
    if (splits > 64) splits = 64;
    if (splits > vocr_cdiv(m, 32)) splits = vocr_cdiv(m, 32);
    if (splits < 1) splits = 1;
    const int rps = vocr_cdiv(m, splits);
    *rows_per_split = rps;
    return vocr_cdiv(m, rps);
}
}  // namespace

extern "C" size_t vocr_colsum_workspace_bytes(int m, int n) {
    if (m <= 0 || n <= 0) return 0;
    int rps;
    const int splits = colsum_splits(m, n, &rps);
    return splits > 1 ? (size_t)splits * n * sizeof(float) : 0;
}

extern "C" int vocr_colsum(const float* x, float* out, int m, int n, void* workspace, void* stream) {
    VOCR_CHECK_ARG(x && out && m > 0 && n > 0, "vocr_colsum: bad argument");
    hipStream_t s = (hipStream_t)stream;
    int rps;
    int splits = colsum_splits(m, n, &rps);
    if (!workspace) { splits = 1; rps = m; }       // no room for partial rows: one workgroup per column tile
    const int ctiles = vocr_cdiv(n, 64);
    colsum_kernel<<<dim3(ctiles, splits), 256, 0, s>>>(x, splits > 1 ? (float*)workspace : out, m, n, rps);
    VOCR_CHECK_LAUNCH("vocr_colsum");
    if (splits > 1) {
        colsum_final_kernel<<<vocr_cdiv(n, 256), 256, 0, s>>>((const float*)workspace, out, n, splits);
        VOCR_CHECK_LAUNCH("vocr_colsum(final)");
    }
    return VOCR_OK;
}

static inline int ew_grid(size_t count) {
    size_t g = (count / 4 + 255) / 256;
    return (int)(g > 2048 ? 2048 : (g ? g : 1));
}
static inline int ew_vec(const void* a, const void* b, const void* c) {
    return ((((uintptr_t)a) | ((uintptr_t)b) | ((uintptr_t)c)) & 15) == 0;
}

extern "C" int vocr_relu_bwd(const float* dy, const float* out, float* dz, size_t count, void* stream) {
    VOCR_CHECK_ARG(dy && out && dz, "vocr_relu_bwd: null pointer");
    if (count == 0) return VOCR_OK;
    relu_bwd_kernel<<<ew_grid(count), 256, 0, (hipStream_t)stream>>>(dy, out, dz, count, ew_vec(dy, out, dz));
    VOCR_CHECK_LAUNCH("vocr_relu_bwd");
    return VOCR_OK;
}

extern "C" int vocr_mul(const float* x, const float* mask, float* out, size_t count, void* stream) {
    VOCR_CHECK_ARG(x && mask && out, "vocr_mul: null pointer");
    if (count == 0) return VOCR_OK;
    mul_kernel<<<ew_grid(count), 256, 0, (hipStream_t)stream>>>(x, mask, out, count, ew_vec(x, mask, out));
    VOCR_CHECK_LAUNCH("vocr_mul");
    return VOCR_OK;
}

extern "C" int vocr_add(const float* x, const float* y, float* out, size_t count, void* stream) {
    VOCR_CHECK_ARG(x && y && out, "vocr_add: null pointer");
    if (count == 0) return VOCR_OK;
    add_kernel<<<ew_grid(count), 256, 0, (hipStream_t)stream>>>(x, y, out, count, ew_vec(x, y, out));
    VOCR_CHECK_LAUNCH("vocr_add");
    return VOCR_OK;
}

extern "C" int vocr_scale_dev(const float* x, const float* scalar, float* out, size_t count, void* stream) {
    VOCR_CHECK_ARG(x && scalar && out, "vocr_scale_dev: null pointer");
    if (count == 0) return VOCR_OK;
    scale_dev_kernel<<<ew_grid(count), 256, 0, (hipStream_t)stream>>>(x, scalar, out, count, ew_vec(x, out, out));
    VOCR_CHECK_LAUNCH("vocr_scale_dev");
    return VOCR_OK;
}

extern "C" int vocr_dropout_mask(float* mask, size_t count, float p, uint64_t seed, void* stream) {
    VOCR_CHECK_ARG(mask, "vocr_dropout_mask: null pointer");
    VOCR_CHECK_ARG(p >= 0.f && p < 1.f, "vocr_dropout_mask: p must be in [0,1)");
    if (count == 0) return VOCR_OK;
    dropout_mask_kernel<<<ew_grid(count), 256, 0, (hipStream_t)stream>>>(mask, count, p, 1.0f / (1.0f - p), seed);
    VOCR_CHECK_LAUNCH("vocr_dropout_mask");
    return VOCR_OK;
}

extern "C" int vocr_dropout_fwd(const float* x, float* out, float* mask, size_t count, float p, uint64_t seed, void* stream) {
    VOCR_CHECK_ARG(x && mask && out, "vocr_dropout_fwd: null pointer");
    VOCR_CHECK_ARG(p >= 0.f && p < 1.f, "vocr_dropout_fwd: p must be in [0,1)");
    if (count == 0) return VOCR_OK;
    dropout_kernel<<<ew_grid(count), 256, 0, (hipStream_t)stream>>>(x, out, mask, count, p, 1.0f / (1.0f - p), seed, ew_vec(x, out, mask));
    VOCR_CHECK_LAUNCH("vocr_dropout_fwd");
    return VOCR_OK;
}

extern "C" int vocr_bchw_to_wbch(const float* x, float* out, int b, int c, int h, int w, void* stream) {
    VOCR_CHECK_ARG(x && out && b > 0 && c > 0 && h > 0 && w > 0, "vocr_bchw_to_wbch: bad argument");
    dim3 grid(vocr_cdiv(w, 32), vocr_cdiv(c * h, 64), b);
    bchw_to_wbch_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(x, out, b, c * h, w, 1);
    VOCR_CHECK_LAUNCH("vocr_bchw_to_wbch");
    return VOCR_OK;
}

extern "C" int vocr_wbch_to_bchw(const float* x, float* out, int b, int c, int h, int w, void* stream) {
    VOCR_CHECK_ARG(x && out && b > 0 && c > 0 && h > 0 && w > 0, "vocr_wbch_to_bchw: bad argument");
    dim3 grid(vocr_cdiv(w, 32), vocr_cdiv(c * h, 64), b);
    bchw_to_wbch_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(x, out, b, c * h, w, 0);
    VOCR_CHECK_LAUNCH("vocr_wbch_to_bchw");
    return VOCR_OK;
}
