// Bidirectional LSTM recurrence on a packed (length-sorted, zero-padded) batch — nn.LSTM semantics
// (reference src/models/cnnlstm.py:148-149,288-290): gate order i,f,g,o; the reverse direction starts at
// each sequence's own last frame; outputs past a sequence's length are zero.
//
// The time-parallel half (x W_ih^T + biases for all T) is a vocr_gemm call made by the host; this file is
// the sequential half.  One launch per time step covers BOTH directions:
//   forward step : gates[b, 4 units x 4 gates] = h_{t-1}[b,:] . W_hh^T  — a [B x H] x [H x 16] product per
//                  workgroup on v_mfma_f32_16x16x4_f32, K split over the 4 waves (one per SIMD) and reduced
//                  through LDS, then the cell update for the 4 owned units is fused in the same kernel.
//   backward step: dh_{t}[b, 16 units] = dG_{t+1}[b,:] . W_hh  (K = 4H, one gate block per wave), fused with
//                  the gate-gradient computation of the 16 owned units.
// h / dh ping-pong between two small global buffers (the only cross-workgroup traffic, L2 resident).
#include "vocr_common.h"
#include <string.h>
#include <stdlib.h>
#include <type_traits>

namespace {

#ifdef VOCR_LSTM_STAMPS        // diagnostic build only (scripts/lstm_stamp.hip): phase stamps of the chain sweeps, never in libvocr.so
__device__ unsigned long long* g_lstm_stamp_out;
#define LSTM_STAMP(k) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); st_acc[k] += now_ - st_last; st_last = now_; } while (0)
#define LSTM_STAMP_DECL unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_last = __builtin_amdgcn_s_memtime()
#else
#define LSTM_STAMP(k) do { } while (0)
#define LSTM_STAMP_DECL do { } while (0)
#endif

// v_exp_f32 and v_rcp_f32 (1 ulp each): an IEEE division is ten dependent instructions on the latency-bound path of every time step
__device__ __forceinline__ float sigmoidf_(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
// tanh(c) of the BACKWARD sweeps' cell section: 1 - 2 / (1 + e^(2x)) on the hardware exp2 / rcp; below |x| = 0.1, where that form
// cancels, the odd polynomial x (1 - x^2/3 + 2 x^4/15): relative error <= 2e-6 everywhere.  Same-box A/B against the library's tanhf
// (-DVOCR_LSTM_LIB_TANH): backward sweep 0.622 vs 0.643 ms; in the FORWARD sweeps the same substitution measured 2 - 3 % slower
// (0.66 vs 0.645 ms), so they keep the library function
__device__ __forceinline__ float tanhf_(float x) {
#ifdef VOCR_LSTM_LIB_TANH
    return tanhf(x);
#else
    const float e = __expf(2.0f * x);
    const float big = 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + e);
    const float x2 = x * x;
    const float small = x * __builtin_fmaf(x2, __builtin_fmaf(x2, 0.13333334f, -0.33333334f), 1.0f);
    return __builtin_fabsf(x) < 0.1f ? small : big;
#endif
}

// grid.x = 2 * (H/4); 256 threads
__global__ __launch_bounds__(256) void lstm_fwd_step_kernel(const float* __restrict__ xproj, const float* __restrict__ whh_f,
                                                            const float* __restrict__ whh_r,
                                                            const int32_t* __restrict__ lens, float* __restrict__ y,
                                                            float* __restrict__ gates, float* __restrict__ cell,
                                                            int T, int B, int H, int step) {
    __shared__ float red[4][64][17];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ublocks = H >> 2;
    const int dir = blockIdx.x / ublocks, unit0 = (blockIdx.x % ublocks) * 4;
    const int t = dir == 0 ? step : T - 1 - step;
    const int RT = (B + 15) >> 4;
    const int lr = lane & 15, q = lane >> 4;
    const int KQ = H >> 4;                       // k values per lane: wave covers H/4, lane-quarter covers H/16
    const int kbase = wave * (H >> 2) + q * KQ;

    // h_{t-1} is read back from y (zeros outside a sequence), c_{t-1} from cell: no separate state buffers
    const int tprev = dir == 0 ? t - 1 : t + 1;
    const float* hp = y + (long)(step > 0 ? tprev : 0) * B * 2 * H + dir * H;       // row stride 2H
    const float* whh = dir ? whh_r : whh_f;
    const float* wrow = whh + ((long)(lr >> 2) * H + unit0 + (lr & 3)) * H + kbase;

    f32x4 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const float* ha[4];
    bool hv[4];
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) {
        const int b = rt * 16 + lr;
        hv[rt] = rt < RT && b < B;
        ha[rt] = hp + (long)(hv[rt] ? b : 0) * 2 * H + kbase;
    }
    if (step > 0) {   // h_{-1} = 0: the first step has no recurrent term
#pragma unroll 4
        for (int s = 0; s < KQ; ++s) {
            const float bw = wrow[s];
#pragma unroll
            for (int rt = 0; rt < 4; ++rt) {
                if (rt < RT) {
                    const float a = hv[rt] ? ha[rt][s] : 0.f;
                    acc[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bw, acc[rt], 0, 0, 0);
                }
            }
        }
    }
#pragma unroll
    for (int rt = 0; rt < 4; ++rt)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[wave][rt * 16 + q * 4 + r][lr] = acc[rt][r];
    __syncthreads();

    if (tid < B * 4) {
        const int b = tid >> 2, u = tid & 3, unit = unit0 + u;
        const bool active = t < lens[b];
        const long gbase = (((long)dir * T + t) * B + b) * 4 * H + unit;
        const long sidx = (((long)dir * T + t) * B + b) * H + unit;
        float pre[4];
#pragma unroll
        for (int g = 0; g < 4; ++g)
            pre[g] = ((red[0][b][g * 4 + u] + red[1][b][g * 4 + u]) + (red[2][b][g * 4 + u] + red[3][b][g * 4 + u])) +
                     xproj[gbase + (long)g * H];
        float* yo = y + ((long)t * B + b) * 2 * H + dir * H + unit;
        f32x4* go = (f32x4*)(gates + sidx * 4);           // gates[dir][t][b][unit][i,f,g,o]
        if (active) {
            const float ig = sigmoidf_(pre[0]), fg = sigmoidf_(pre[1]), gg = tanhf(pre[2]), og = sigmoidf_(pre[3]);
            const float cprev = step > 0 ? cell[(((long)dir * T + tprev) * B + b) * H + unit] : 0.f;
            const float c = fg * cprev + ig * gg;
            const float h = og * tanhf(c);
            *go = (f32x4){ig, fg, gg, og};
            cell[sidx] = c;
            *yo = h;
        } else {
            *go = (f32x4){0.f, 0.f, 0.f, 0.f};
            cell[sidx] = 0.f;
            *yo = 0.f;
        }
    }
}

// grid.x = 2 * (H/16); 256 threads.  Backward iteration `step` visits t = T-1-step (forward dir) and
// t = step (reverse dir); dG of the previously visited time feeds dh through W_hh.
__global__ __launch_bounds__(256) void lstm_bwd_step_kernel(const float* __restrict__ dy, const float* __restrict__ whh_f,
                                                            const float* __restrict__ whh_r,
                                                            const int32_t* __restrict__ lens, const float* __restrict__ gates,
                                                            const float* __restrict__ cell, float* __restrict__ dgates,
                                                            float* __restrict__ dcbuf, int T, int B, int H, int step) {
    __shared__ float red[4][64][17];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ublocks = H >> 4;
    const int dir = blockIdx.x / ublocks, unit0 = (blockIdx.x % ublocks) * 16;
    const int t = dir == 0 ? T - 1 - step : step;
    const int tv = dir == 0 ? t + 1 : t - 1;      // time whose dG was produced by the previous iteration
    const int RT = (B + 15) >> 4;
    const int lr = lane & 15, q = lane >> 4;
    const int KQ = H >> 2;                        // wave = one gate block of H rows; lane-quarter covers H/4
    const int nbase = wave * H + q * KQ;

    f32x4 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};

    if (step > 0) {
        const float* dgp = dgates + (((long)dir * T + tv) * B) * 4 * H + nbase;
        const float* whht = dir ? whh_r : whh_f;          // transposed weights: whhT[unit][n], n over 4H
        const float* wp = whht + (long)(unit0 + lr) * 4 * H + nbase;
        const float* ga[4];
        bool gv[4];
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) {
            const int b = rt * 16 + lr;
            gv[rt] = rt < RT && b < B;
            ga[rt] = dgp + (long)(gv[rt] ? b : 0) * 4 * H;
        }
#pragma unroll 4
        for (int s = 0; s < KQ; ++s) {
            const float bw = wp[s];
#pragma unroll
            for (int rt = 0; rt < 4; ++rt) {
                if (rt < RT) {
                    const float a = gv[rt] ? ga[rt][s] : 0.f;
                    acc[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bw, acc[rt], 0, 0, 0);
                }
            }
        }
    }
#pragma unroll
    for (int rt = 0; rt < 4; ++rt)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[wave][rt * 16 + q * 4 + r][lr] = acc[rt][r];
    __syncthreads();

    for (int cidx_l = tid; cidx_l < B * 16; cidx_l += 256) {
        const int b = cidx_l >> 4, j = cidx_l & 15, unit = unit0 + j;
        const int len = lens[b];
        const long gbase = (((long)dir * T + t) * B + b) * 4 * H + unit;
        const long cb = ((long)dir * B + b) * H + unit;
        if (t >= len) {
            dgates[gbase] = 0.f;
            dgates[gbase + H] = 0.f;
            dgates[gbase + 2l * H] = 0.f;
            dgates[gbase + 3l * H] = 0.f;
            dcbuf[cb] = 0.f;
            continue;
        }
        const float dh = dy[((long)t * B + b) * 2 * H + dir * H + unit] +
                         ((red[0][b][j] + red[1][b][j]) + (red[2][b][j] + red[3][b][j]));
        const long sidx = (((long)dir * T + t) * B + b) * H + unit;
        const f32x4 gv = *(const f32x4*)(gates + sidx * 4);
        const float ig = gv[0], fg = gv[1], gg = gv[2], og = gv[3];
        const float c = cell[sidx];
        const int tp = dir == 0 ? t - 1 : t + 1;   // previous step of the recurrence
        const float cprev = (tp >= 0 && tp < len) ? cell[(((long)dir * T + tp) * B + b) * H + unit] : 0.f;
        const float tc = tanhf(c);
        const float dcar = step > 0 ? dcbuf[cb] : 0.f;
        const float dc = dcar + dh * og * (1.f - tc * tc);
        dgates[gbase] = dc * gg * ig * (1.f - ig);
        dgates[gbase + H] = dc * cprev * fg * (1.f - fg);
        dgates[gbase + 2l * H] = dc * ig * (1.f - gg * gg);
        dgates[gbase + 3l * H] = dh * tc * og * (1.f - og);
        dcbuf[cb] = dc * fg;
    }
}

// ------------------------------------------------------------------------------------------------ fast paths
// Same algorithms, shaped for the hardware: every operand a lane needs is a run of consecutive floats, so
// all global reads are 16-byte loads issued up front (one latency exposure per step instead of one per k).

// H = 64*KQ4, B <= 16*RT.  grid.x = 2 * (H/4).
template <int KQ4, int RT>
__global__ __launch_bounds__(256) void lstm_fwd_step_fast(const float* __restrict__ xproj, const float* __restrict__ whh_f,
                                                          const float* __restrict__ whh_r, const int32_t* __restrict__ lens,
                                                          float* __restrict__ y, float* __restrict__ gates,
                                                          float* __restrict__ cell, int T, int B, int step, int dbg) {
    constexpr int H = 64 * KQ4;
    __shared__ float red[4][RT * 16][17];
    __builtin_amdgcn_s_setprio(3);      // latency-critical: win issue arbitration against co-resident throughput kernels
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int ublocks = H >> 2;
    const int dir = blockIdx.x / ublocks, unit0 = (blockIdx.x % ublocks) * 4;
    const int t = dir == 0 ? step : T - 1 - step;
    const int lr = lane & 15, q = lane >> 4;
    // k owned by (wave, load i, lane-quarter q, element e) = wave*H/4 + 16*i + 4*q + e: the four quarters of a
    // row read one contiguous 64 B, so a wave-instruction touches 16 cache lines instead of 64
    const int kbase = wave * (H >> 2) + q * 4;
    // h_{t-1} is read back from y (zeros outside a sequence) and c_{t-1} from cell: no state buffers, 3 stores/cell
    const int tprev = dir == 0 ? t - 1 : t + 1;
    const float* hp = y + (long)(step > 0 ? tprev : 0) * B * 2 * H + dir * H;       // row stride 2H
    const float* whh = dir ? whh_r : whh_f;

    // cell-update operands are fetched before the MFMA phase so their latency hides under it
    const bool cellthr = tid < B * 4;
    const int cb_ = tid >> 2, cu = tid & 3, unit = unit0 + cu;
    const long gbase = (((long)dir * T + t) * B + (cellthr ? cb_ : 0)) * 4 * H + unit;
    float xp[4] = {0.f, 0.f, 0.f, 0.f};
    float cprev = 0.f;
    int len_b = 0;
    if (cellthr) {
#pragma unroll
        for (int g = 0; g < 4; ++g) xp[g] = xproj[gbase + (long)g * H];
        len_b = lens[cb_];
        if (step > 0) cprev = cell[(((long)dir * T + tprev) * B + cb_) * H + unit];
    }

    f32x4 acc[RT];
#pragma unroll
    for (int i = 0; i < RT; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (step > 0 && !(dbg & 2)) {
        const f32x4* wp = (const f32x4*)(whh + ((long)(lr >> 2) * H + unit0 + (lr & 3)) * H + kbase);
        f32x4 wv[KQ4], hv[RT][KQ4];
#pragma unroll
        for (int i = 0; i < KQ4; ++i) wv[i] = wp[i * 4];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            const int b = rt * 16 + lr;
            const f32x4* hq = (const f32x4*)(hp + (long)(b < B ? b : 0) * 2 * H + kbase);
#pragma unroll
            for (int i = 0; i < KQ4; ++i) {
                f32x4 v = hq[i * 4];
                if (b >= B) v = (f32x4){0.f, 0.f, 0.f, 0.f};
                hv[rt][i] = v;
            }
        }
        __builtin_amdgcn_sched_barrier(0);     // keep every load above the MFMA phase: one latency exposure per step
        if (!(dbg & 1)) {
#pragma unroll
        for (int i = 0; i < KQ4; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int rt = 0; rt < RT; ++rt)
                    acc[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(hv[rt][i][e], wv[i][e], acc[rt], 0, 0, 0);
        } else {
#pragma unroll
            for (int i = 0; i < KQ4; ++i)
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) acc[rt] += hv[rt][i] * wv[i];
        }
    }
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[wave][rt * 16 + q * 4 + r][lr] = acc[rt][r];
    __syncthreads();

    if (cellthr && !(dbg & 4)) {
        const int b = cb_, u = cu;
        const bool active = t < len_b;
        const long sidx = (((long)dir * T + t) * B + b) * H + unit;
        float pre[4];
#pragma unroll
        for (int g = 0; g < 4; ++g)
            pre[g] = ((red[0][b][g * 4 + u] + red[1][b][g * 4 + u]) + (red[2][b][g * 4 + u] + red[3][b][g * 4 + u])) + xp[g];
        float* yo = y + ((long)t * B + b) * 2 * H + dir * H + unit;
        f32x4* go = (f32x4*)(gates + sidx * 4);           // gates[dir][t][b][unit][i,f,g,o]: one 16-B store
        if (active) {
            const float ig = sigmoidf_(pre[0]), fg = sigmoidf_(pre[1]), gg = tanhf(pre[2]), og = sigmoidf_(pre[3]);
            const float c = fg * cprev + ig * gg;
            const float h = og * tanhf(c);
            *go = (f32x4){ig, fg, gg, og};
            cell[sidx] = c;
            *yo = h;
        } else {
            *go = (f32x4){0.f, 0.f, 0.f, 0.f};
            cell[sidx] = 0.f;
            *yo = 0.f;
        }
    }
}

// ------------------------------------------------------------------------------------------------ persistent sweeps
// One launch for all T steps of a layer.  A workgroup keeps its slice of W_hh in registers for the whole sweep (the
// per-step kernels re-fetch 8 MB of weights per step because the XCD L2s are dropped at every kernel boundary: PMC
// FETCH_SIZE 7-10 MB per step launch) and the cell state c in a register.  The only cross-workgroup traffic is the
// hand-off of h_t (forward) or of partial sums (backward):
//   producer: payload stored write-through (sc1) -> every storing wave drains vmcnt -> workgroup barrier -> ONE lane
//             raises this workgroup's arrival flag (an sc1 store of step+1)
//   consumer: the last wave (it owns no cell, so its memory queue holds nothing but polls) re-reads the flags of its
//             chain (relaxed agent-scope loads + s_sleep) until every workgroup has published the previous step ->
//             workgroup barrier -> EVERY load of the payload is an sc1 load (bypasses this CU's L1).
// This is the flag form of the release/acquire-free hand-off of cdna_hip_programming.md Guideline 16 (R1 + sc1 loads,
// table row 1).  Results are independent of dispatch order and XCD placement.  Every spin is bounded; a timeout sets
// status[0] and the sweep's last step then writes NaN, so a failed hand-off can never pass silently.  All workgroups
// must be co-resident (host check).
//
// History (MI355X, T=294 B=32 H=512, scripts/lstm_bench.py; one launch per step: 7.3 us fwd / 10.3 us bwd per step).
// A first persistent forward over all 256 CUs (4 units per workgroup, one chain of 128 workgroups per direction on 8
// XCDs) only reached 6.9 us: the signalling form (one counter, per-workgroup flags, polling wave 0 or 3), two
// workgroups per CU on independent chains, 8-byte vs 16-byte sc1 loads, and a flag-free variant that pre-fills y with
// a NaN pattern and re-reads h until no element is the pattern all stayed within +-0.5 us of it; a workgroup's own
// loop without any waiting was already 5.0 us.  What changed the picture is below: small chains that live on one XCD.
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------------------------------------ chain sweeps
// The recurrence never couples batch rows, so (direction, tile of 16 batch rows) is an independent chain: <= 8 of them.
// A chain's H/16 workgroups (<= 32: one XCD's CUs) take blockIdx = member*8 + chain, i.e. equal blockIdx % 8, which the
// dispatcher has been observed to deal to ONE XCD.  That placement is never assumed: every workgroup publishes its
// HW_REG_XCC_ID at the start (write-through), every member reads all of them, and only if the whole chain sits on one
// XCD does the chain switch its payload and flag stores from write-through (sc1) to plain stores.  Plain stores stay
// in that XCD's L2, which is the coherence point of all its CUs, and the consumer's sc1 loads (L1 bypassed,
// L2 served) then hit them at L2 latency instead of making a fabric round trip per step.  With any other placement the
// chain runs the write-through protocol described above unchanged, so results never depend on placement.
__device__ __forceinline__ unsigned xcc_id() {
    unsigned x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    return x;
}

// A hand-off timed out: the poison decision of THIS sweep reads the per-call status word (workspace, zeroed by the host's
// memset in front of every launch); the caller's health word only collects the event for reporting.  (The sweeps used to
// test the caller's sticky word itself: one transient timeout then NaN-poisoned every later sweep of the process.)
constexpr unsigned kHandoffSentinel = 0xFFFFFFFFu;      // "not written yet" (see the self-validating hand-off below)

__device__ __forceinline__ void raise_timeout(unsigned* status, unsigned* health) {
    __hip_atomic_store(status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (health) __hip_atomic_store(health, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Returns true iff all `members` workgroups of this chain report the same XCC id.  ids: one word per member, set to
// `unwritten` by the host (0 for the flag sweeps; 0xFFFFFFFF for the self-validating ones, whose ids, status word and ring
// are ONE 0xFF fill - a 2-KB fill of its own in front of every sweep was a kernel launch on the step's critical path).
// Called by every thread of the workgroup; scratch is one LDS word.
__device__ __forceinline__ bool chain_is_xcd_local(unsigned* ids, int members, int member, unsigned* status, unsigned* health, unsigned* scratch,
                                                   unsigned unwritten = 0u) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const unsigned mine = xcc_id() + 1u;
    if (tid == 0) __hip_atomic_store(ids + member, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (wave == 3) {
        unsigned spins = 0;
        bool same = false;
        for (;;) {
            unsigned v = mine;
            if (lane < members) v = __hip_atomic_load(ids + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (__all(v != unwritten)) { same = __all(v == mine); break; }
            __builtin_amdgcn_s_sleep(2);
            if (++spins > (1u << 22)) {
                if (lane == 0) raise_timeout(status, health);
                break;
            }
        }
        if (lane == 0) *scratch = same ? 1u : 0u;
    }
    __syncthreads();
    const bool r = *scratch != 0u;
    __syncthreads();
    return r;
}

// Forward chain sweep.  H = 64*KQ4; grid.x = 8 * (H/16); workgroup = (chain, 16 units): four 16-row gate tiles x 16
// batch rows x K = H, W_hh slice (64 x H floats, 128 VGPRs per lane at H = 512) resident.  Same MFMA and reduction
// order per output element as lstm_fwd_step_fast, so results are bit-identical.
template <int KQ4>
__global__ __launch_bounds__(256) void lstm_fwd_chain(const float* __restrict__ xproj, const float* __restrict__ whh_f,
                                                      const float* __restrict__ whh_r, const int32_t* __restrict__ lens,
                                                      float* y, float* __restrict__ gates, float* __restrict__ cell,
                                                      unsigned* flags, unsigned* ids, unsigned* status, unsigned* health, int T, int B, int nbt, int force_wt,
                                                      int s0, int s1) {
    constexpr int H = 64 * KQ4;
    constexpr int members = H >> 4;
    __shared__ float lds[4 * 16 * 65 + 4];
    float (*red)[16][65] = (float (*)[16][65])lds;
    const int chain = blockIdx.x & 7, member = blockIdx.x >> 3;
    if (chain >= 2 * nbt) return;
    __builtin_amdgcn_s_setprio(3);      // latency-critical: win issue arbitration against co-resident GEMM waves
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int dir = chain / nbt, bt = chain % nbt;
    const int unit0 = member * 16, b0 = bt * 16;
    const int nrows = min(B - b0, 16);
    const int lr = lane & 15, q = lane >> 4;
    const int kbase = wave * (H >> 2) + q * 4;
    const float* whh = dir ? whh_r : whh_f;
    unsigned* cflags = flags + chain * 32;
    const bool local = chain_is_xcd_local(ids + chain * 32, members, member, status, health, (unsigned*)(lds + 4 * 16 * 65)) && !(force_wt & 1);

    // resident W_hh fragments: tile j holds units unit0+4j..+3; B-operand lane lr = gate (lr>>2), unit 4j + (lr&3)
    f32x4 wv[4][KQ4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const f32x4* wp = (const f32x4*)(whh + ((long)(lr >> 2) * H + unit0 + 4 * j + (lr & 3)) * H + kbase);
#pragma unroll
        for (int i = 0; i < KQ4; ++i) wv[j][i] = wp[i * 4];
    }
    const int bl = tid >> 4, cu = tid & 15, cb_ = b0 + bl, unit = unit0 + cu;
    const bool cellthr = bl < nrows;
    const int len_b = cellthr ? lens[cb_] : 0;
    const int ccol = (cu >> 2) * 16 + (cu & 3);
    // steps [s0, s1) of the sweep: a later range resumes from what the earlier launch left in y (h) and cell (c)
    float cstate = 0.f;
    if (s0 > 0 && cellthr) {
        const int tp0 = dir == 0 ? s0 - 1 : T - s0;
        cstate = cell[(((long)dir * T + tp0) * B + cb_) * H + unit];
    }
    bool timed_out = false;
    const __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)y, 0, T * B * 2 * H * 4, 0x00020000);
    const int arow = b0 + lr;
    const bool av = arow < B;

    // The x-projection of a step (plain loads: an earlier kernel wrote it) is fetched one step ahead, behind the h loads:
    // issued at the top of its own step it sat in front of the polling wave's poll loads (loads return in order), so
    // every poll waited out an HBM round trip (4.6 -> 4.25 us per step).  Unconditional loads (row clamped) keep the
    // wait counts exact.  (The same move in the backward sweep measured slower, 6.8 vs 6.4 us: not done there.)
    const int xb = cellthr ? cb_ : b0;
    auto x_loads = [&](int st, float (&xv)[4]) {
        const int tt = dir == 0 ? st : T - 1 - st;
        const float* xrow = xproj + (((long)dir * T + tt) * B + xb) * 4 * H + unit;
#pragma unroll
        for (int g = 0; g < 4; ++g) xv[g] = xrow[(long)g * H];
    };
    float xn[4];
    x_loads(s0, xn);

    for (int step = s0; step < s1; ++step) {
        const int t = dir == 0 ? step : T - 1 - step;
        const int tprev = dir == 0 ? t - 1 : t + 1;
        float xp[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) xp[g] = xn[g];
        if (step == 0) x_loads(T > 1 ? 1 : 0, xn);      // (step 0 has no h to load: the prefetch rides alone)
        f32x4 acc[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (step > 0) {
            if (wave == 3 && !timed_out && step > s0) {
                unsigned spins = 0;
                for (;;) {
                    unsigned f0 = (unsigned)step;
                    if (lane < members) f0 = __hip_atomic_load(cflags + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (__all(f0 >= (unsigned)step)) break;
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > (1u << 22)) {
                        if (lane == 0) raise_timeout(status, health);
                        timed_out = true;
                        break;
                    }
                }
            }
            __syncthreads();
            const int off = ((tprev * B + (av ? arow : b0)) * 2 * H + dir * H + kbase) * 4;
            // Rows past B are clamped to a valid row, never masked: row r of A only reaches row r of the product.
            // Chain on one XCD: plain loads.  The payload sits in this XCD's L2 (sc0 stores), and this CU's L1 cannot
            // hold a line of y[t-1]: the kernel never reads a row of y before the step after it was written, chains
            // never share a 128-B line, and the L1 starts the kernel invalid.  (sc1 loads of the same lines take a
            // fabric round trip: 1.1 us per step in the stamps, 4.25 -> 3.8 us per step overall.)  The backward sweep
            // keeps the write-through hand-off: with L2-resident dgates and plain loads it measured 6.7 vs 6.4 us.
            u32x4_t hv[KQ4];
            if (local) {
#pragma unroll
                for (int i = 0; i < KQ4; ++i) hv[i] = __builtin_amdgcn_raw_buffer_load_b128(yrsrc, off + i * 64, 0, 0);
            } else {
#pragma unroll
                for (int i = 0; i < KQ4; ++i) hv[i] = __builtin_amdgcn_raw_buffer_load_b128(yrsrc, off + i * 64, 0, 16);      // aux 16 = sc1: L1 bypassed
            }
            x_loads(step + 1 < T ? step + 1 : step, xn);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < KQ4; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(hv[i][e]), wv[j][i][e], acc[j], 0, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[wave][q * 4 + r][j * 16 + lr] = acc[j][r];
        __syncthreads();

        if (cellthr) {
            const bool active = t < len_b;
            const long sidx = (((long)dir * T + t) * B + cb_) * H + unit;
            float pre[4];
#pragma unroll
            for (int g = 0; g < 4; ++g)
                pre[g] = ((red[0][bl][g * 4 + ccol] + red[1][bl][g * 4 + ccol]) + (red[2][bl][g * 4 + ccol] + red[3][bl][g * 4 + ccol])) + xp[g];
            float* yo = y + ((long)t * B + cb_) * 2 * H + dir * H + unit;
            f32x4* go = (f32x4*)(gates + sidx * 4);
            float h = 0.f;
            if (active) {
                const float ig = sigmoidf_(pre[0]), fg = sigmoidf_(pre[1]), gg = tanhf(pre[2]), og = sigmoidf_(pre[3]);
                const float c = fg * cstate + ig * gg;
                h = og * tanhf(c);
                *go = (f32x4){ig, fg, gg, og};
                cell[sidx] = c;
                cstate = c;
            } else {
                *go = (f32x4){0.f, 0.f, 0.f, 0.f};
                cell[sidx] = 0.f;
                cstate = 0.f;
            }
            if (step == s1 - 1 && __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 1u)
                h = __uint_as_float(0x7FC00000u);                                        // a hand-off timed out: fail loudly
            if (local) __hip_atomic_store(yo, h, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);      // stays in this XCD's L2
            else __hip_atomic_store(yo, h, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);                // write-through (sc1)
        }
        __builtin_amdgcn_s_waitcnt(0x0F70);                  // vmcnt(0): every storing wave drains before the flag
        __syncthreads();
        if (tid == 0) {
            if (local) __hip_atomic_store(cflags + member, (unsigned)(step + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else __hip_atomic_store(cflags + member, (unsigned)(step + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// Row layout of the sequence-side tensors (xproj, y, gates, cell, dy, dgates) as the 4-row chain sweeps address them.
//   dense  (packed_rows == 0): the reference's padded time-major order, row(t, b) = t*B + b; every chain runs all T steps and a row
//          past its length computes zeros.
//   packed (packed_rows  > 0): what pack_padded_sequence buys the reference (src/models/cnnlstm.py:288-290: cuDNN never touches a
//          padded frame).  The batch is sorted by length, so chain c (batch rows 4c .. 4c+3) needs L_c = lens[4c] steps and no more.
//          Rows come in GROUPS of 4 (one time step of one chain), chain-major: [zero group][chain 0: L_0 groups][zero group][chain 1:
//          L_1 groups] ... [zero group]; group(c, t) = 1 + c + sum_{c' < c} L_c' + t, row = 4*group + (b & 3), packed_rows =
//          4 * (sum L_c + chains + 1).  Inside a chain consecutive steps are 4 rows apart, and the all-zero group on either side
//          stands for h_{-1} / h_{L} = 0, so the recurrent weight gradient stays ONE product of row-shifted views (dgates rows
//          [4, R) against y rows [0, R - 4) forward, the other way round reverse), exactly as in the dense layout with a shift of B.
//          Every GEMM of the LSTM stack runs over the packed rows only; the zero groups (and a last chain's rows >= B) are zeroed by the
//          caller (vocr_lstm_fwd_packed / vocr_lstm_bwd_packed document it).  The reverse direction of a chain starts at its own last
//          frame t = L_c - 1; a row shorter than its chain is masked until t < len exactly as in the dense layout.
struct SeqRows {
    int base;        // row of (t = 0, first row of the chain)
    int stride;      // rows between consecutive time steps of the chain
    int steps;       // time steps this chain runs
    long total;      // rows of one direction's plane (T*B or packed_rows)
};
__device__ __forceinline__ SeqRows seq_rows(const int32_t* __restrict__ lens, int T, int B, int bt, int packed_rows) {
    SeqRows r;
    if (packed_rows == 0) {
        r.base = 4 * bt;
        r.stride = B;
        r.steps = T;
        r.total = (long)T * B;
    } else {
        int g = 1 + bt;
        for (int c = 0; c < bt; ++c) g += lens[4 * c];
        r.base = 4 * g;
        r.stride = 4;
        r.steps = min(lens[4 * bt], T);
        r.total = packed_rows;
    }
    return r;
}

// packed rows only: the chain's last group and the zero group behind it lie inside the plane the caller allocated (its `rows` must be
// packed_row_count(lens): a mismatch would write out of bounds)
__device__ __forceinline__ bool seq_rows_fit(const SeqRows& r, int packed_rows) {
    return packed_rows == 0 || (long)r.base + 4l * r.steps + 4 <= r.total;
}

template <int N, int I = 0, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<N, I + 1>(f);
    }
}

// Self-validating hand-off: the host fills the hand-off buffer with this bit pattern before a sweep (a NaN no arithmetic here
// produces: h is stored canonicalised); a consumer wave re-issues ITS OWN loads until no dword holds the pattern, a producer only
// stores.  Against the flag protocol of lstm_fwd_chain (flag poll and h loads = two L2 round trips in series, plus a store drain
// + barrier + flag store on the producing side): no flags, no drain, two barriers per step instead of four, and a wave starts its
// MFMAs as soon as its own k-slice has arrived.
// cache policy of a polling load: sc1 (never served from the CU's vector L1, which may hold the line from an earlier look or
// from the slot's previous use).  Every polling loop opens with POLL_FENCE: to the compiler the load is loop-invariant, and
// without a memory clobber in the loop it hoists the load and a poll that fails once spins on the same registers until it
// times out (seen in a variant of these kernels; the intrinsic's volatile bit does not stop the hoist and turns the load system-scope).
constexpr int kPollAux = 16;
#define POLL_FENCE() asm volatile("" ::: "memory")

// Forward chain sweep on 4-row chains ("chain4v"): 2*ceil(B/4) <= 16 chains of H/16 workgroups with 4 waves; up to 8 chains
// one workgroup per CU, above that two (chains c and c + 8 share an XCD), each with its own barriers and its own copy of its
// W_hh slice (128 VGPRs per lane at H = 512, K = H split over the 4 waves).  One v_mfma_f32_4x4x1_16b_f32 = 4 rows x 64 gate
// columns x one k (all 16 blocks take the same A rows through the instruction's A broadcast), so no operand is duplicated and
// h never passes through LDS.  Measured at T = 294, H = 512 (scripts/lstm_ab.py, lstm_stamp4.hip): B = 32 2.25-2.3 us per step
// (8-row flag chains 2.53), B = 16 1.55 (2.17).  What a step is made of at B = 16 (s_memtime ticks of 0.46 ns, wave 0): hand-off
// 930 (own stores -> every member's slice readable; a poll round trip is ~450 when the line is valid in L2 but ~900-1200 on the
// first read of a freshly written line) | 128 MFMAs 1170 (two accumulators: a single dependent chain took 1740) | barriers 350 |
// reduction + activation 260 | cell update 620.  With two workgroups per CU the MFMA phases of one hide behind the waits of the
// other only partly (random relative phase on every CU, and a chain moves at the pace of its slowest member).
// Measured and dropped for 16 < B <= 32: ONE 8-wave workgroup per CU running its two 4-row halves as alternating sub-chains with a
// shared W_hh slice (64 VGPRs), so that one half's hand-off travels while the other half computes: 2.73 us per step against 2.36
// here - inside one workgroup the barriers serialise poll round trip, MFMAs, reduction and cell update of each half (the MFMAs are
// a quarter of a sub-step), which is exactly what two independent workgroups let the hardware overlap.
template <int KQ4>
__global__ __launch_bounds__(256) void lstm_fwd_chain4v(const float* __restrict__ xproj, const float* __restrict__ whh_f,
                                                        const float* __restrict__ whh_r, const int32_t* __restrict__ lens,
                                                        float* y, float* __restrict__ gates, float* __restrict__ cell,
                                                        float* hx, unsigned* ids, unsigned* status, unsigned* health, int T, int B, int NT4,
                                                        int force_wt, int s0, int s1, int packed_rows) {
    constexpr int H = 64 * KQ4;
    constexpr int members = H >> 4;
    constexpr int KW = H / 4;                         // k per wave
    constexpr int NL = KW / 4;                        // 16-byte pieces of a row's k-slice
    __shared__ float lds[4 * 4 * 65 + 4 + 4 * 64];
    float (*red)[4][65] = (float (*)[4][65])lds;
    float (*actb)[64] = (float (*)[64])(lds + 4 * 4 * 65 + 4);
    const int nch = 2 * NT4;
    const int chain = nch > 8 ? (int)(blockIdx.x & 7) + 8 * (int)((blockIdx.x >> 3) & 1) : (int)(blockIdx.x & 7);
    const int member = nch > 8 ? blockIdx.x >> 4 : blockIdx.x >> 3;
    if (chain >= nch) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int dir = chain / NT4, bt = chain % NT4;
    const int unit0 = member * 16, b0 = bt * 4;
    const int nrows = min(B - b0, 4);
    const int kbase = wave * KW;
    const float* whh = dir ? whh_r : whh_f;
    const bool local = chain_is_xcd_local(ids + chain * 32, members, member, status, health, (unsigned*)(lds + 4 * 4 * 65), kHandoffSentinel) && !(force_wt & 1);

    // resident B operand: lane = gate column (gate, local unit) = (lane >> 4, lane & 15), the wave's KW k
    f32x4 wv[NL];
    {
        const f32x4* wp = (const f32x4*)(whh + ((long)(lane >> 4) * H + unit0 + (lane & 15)) * H + kbase);
#pragma unroll
        for (int i = 0; i < NL; ++i) wv[i] = wp[i];
    }
    // epilogue: (A) thread (wave = gate, lane = row*16 + unit) reduces the four partial tiles and activates ONE pre-activation,
    // (B) wave 0 (lane = row*16 + unit) combines the four gates of its cell
    const int egate = __builtin_amdgcn_readfirstlane(wave);
    const int erow = lane >> 4, ecol = egate * 16 + (lane & 15);
    const bool erowok = erow < nrows;
    const int bl = lane >> 4, cu = lane & 15, cb_ = b0 + bl, unit = unit0 + cu;
    const bool cellthr = tid < 64 && bl < nrows;
    const int len_b = cellthr ? lens[cb_] : 0;
    const SeqRows sr = seq_rows(lens, T, B, bt, packed_rows);
    const int Tc = sr.steps;
    s1 = min(s1, Tc);
    if (!seq_rows_fit(sr, packed_rows)) {             // the caller's row count does not belong to these lengths: fail loudly, touch nothing
        if (tid == 0) raise_timeout(status, health);
        return;
    }
    // the cell thread's own element of step t: plane-local row = sr.base + sr.stride*t + bl
    const long crow0 = (long)dir * sr.total + sr.base + bl;
    float cstate = 0.f;
    if (s0 > 0 && cellthr) {
        const int tp0 = dir == 0 ? s0 - 1 : Tc - s0;
        cstate = cell[(crow0 + (long)sr.stride * tp0) * H + unit];
    }
    bool timed_out = false;
    // hand-off ring hx[step & 3][chain][member][4 rows][16 units]: a member's block of a step is 256 contiguous bytes = two whole
    // 128-byte lines written by ONE store instruction (in y a line is shared by two members: a half-written line is a partial
    // line in L2 and a read of it goes to memory to merge), a consumer wave's 8 members are 2 KB contiguous, and the ring stays
    // in L2 (a poll that comes too early costs an L2 round trip, ~450 ticks, not a fetch of a never-touched line, ~1000).  A
    // block goes back to "not written yet" two steps after it was written, by the lane that wrote it: by then this workgroup has
    // read every member's h of the step after (all four waves are past their polls), and a member stores that only after ITS
    // four waves have read the block.  Ordered before anybody's poll of the slot's next use: the resetting wave's next poll
    // waits for all its outstanding stores, and its next h store - which every consumer needs to get that far - comes after.
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)hx, 0, 4 * nch * 4 * H * 4, 0x00020000);
    // A operand without LDS: the instruction's A-broadcast (cbsz = 4: all 16 blocks take block `abid`'s A lanes) lets one
    // VGPR carry 16 different k: load j of lane 4b + r is the 16-byte piece k = kbase + 64j + 4b .. +3 of row r (an
    // instruction covers 256 contiguous bytes of each row), and the MFMA for k = 64j + 4b + e names register (j, e) with
    // abid = b (probe: scripts/mfma_cbsz_probe.hip)
    constexpr int KB = KW / 16;
    constexpr int NP = KB / 4;                        // 16-byte loads per lane
    int poff[NP];
    {
        const int prow = (lane & 3) < nrows ? (lane & 3) : 0;             // rows clamped, never masked: A row r only reaches output row r
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            const int kk = kbase + 64 * j + 4 * (lane >> 2);
            poff[j] = (((kk >> 4) * 4 + prow) * 16 + (kk & 15)) * 4;
        }
    }

    const float* xrowp = xproj + ((long)dir * sr.total + sr.base + (erowok ? erow : 0)) * 4 * H + (long)egate * H + unit0 + (lane & 15);
    const long xstep = (long)sr.stride * 4 * H;
    auto x_load = [&](int st) {
        const int tt = dir == 0 ? st : Tc - 1 - st;
        return xrowp[tt * xstep];
    };
    float xn = s0 < s1 ? x_load(s0) : 0.f;
    // the resident operand is complete before the loop (otherwise every iteration carries the first one's vmcnt waits, which
    // then also wait for whatever else is in flight)
#pragma unroll
    for (int i = 0; i < NL; ++i) asm volatile("" : "+v"(wv[i]));

    LSTM_STAMP_DECL;
    for (int step = s0; step < s1; ++step) {
        const int t = dir == 0 ? step : Tc - 1 - step;
        LSTM_STAMP(7);
        const float xp = xn;
        f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f}, acc1 = (f32x4){0.f, 0.f, 0.f, 0.f};      // even / odd 4-k pieces: two independent MFMA chains
        u32x4_t pv[NP];
        if (step > 0) {
            const int toff = (((step - 1) & 3) * nch + chain) * 4 * H * 4;
            unsigned spins = 0;
            for (;;) {
                POLL_FENCE();
                unsigned mx = 0u;
#pragma unroll
                for (int j = 0; j < NP; ++j) {
                    pv[j] = __builtin_amdgcn_raw_buffer_load_b128(xrsrc, toff + poff[j], 0, kPollAux);
                    mx = max(max(mx, pv[j][0]), max(max(pv[j][1], pv[j][2]), pv[j][3]));
                }
                if (__all(mx != kHandoffSentinel) || timed_out) break;
                if (++spins > (1u << 22)) {
                    if (lane == 0) raise_timeout(status, health);
                    timed_out = true;
                    break;
                }
            }
        }
        LSTM_STAMP(0);              // own slice of h_{t-1} arrived (polls)
        xn = x_load(step + 1 < Tc ? step + 1 : step);          // behind the polls: loads return in order
        if (step > 0) {
            {
#pragma unroll
                for (int j = 0; j < NP; ++j)
                    static_for<16>([&](auto bc) {
                        constexpr int b = decltype(bc)::value;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            if constexpr (b % 2 == 0) acc = __builtin_amdgcn_mfma_f32_4x4x1f32(__uint_as_float(pv[j][e]), wv[16 * j + b][e], acc, 4, b, 0);
                            else acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(__uint_as_float(pv[j][e]), wv[16 * j + b][e], acc1, 4, b, 0);
                        }
                    });
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[r] += acc1[r];
        }
        // acc[r] = partial of (row r, column lane)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[wave][r][lane] = acc[r];
        LSTM_STAMP(1);              // MFMA + partial tile to LDS
        __syncthreads();
        LSTM_STAMP(2);
        {
            const float pre = ((red[0][erow][ecol] + red[1][erow][ecol]) + (red[2][erow][ecol] + red[3][erow][ecol])) + xp;
            actb[erow][ecol] = egate == 2 ? tanhf(pre) : sigmoidf_(pre);           // wave-uniform choice
            // the next step's x-projection value has long arrived: take it HERE, so that no load is pending at the loop's
            // back edge (the compiler otherwise closes every iteration with vmcnt(0) = wave 0 waiting for its stores' acks)
            asm volatile("" : "+v"(xn));
        }
        LSTM_STAMP(3);              // reduce + activation
        __syncthreads();
        LSTM_STAMP(4);
        if (cellthr) {
            const bool active = t < len_b;
            const long sidx = (crow0 + (long)sr.stride * t) * H + unit;
            float* yo = y + (sr.base + bl + (long)sr.stride * t) * 2 * H + dir * H + unit;
            f32x4* go = (f32x4*)(gates + sidx * 4);
            float h = 0.f, c = 0.f;
            f32x4 gv = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (active) {
                const float ig = actb[bl][cu], fg = actb[bl][16 + cu], gg = actb[bl][32 + cu], og = actb[bl][48 + cu];
                c = fg * cstate + ig * gg;
                h = og * tanhf(c);
                gv = (f32x4){ig, fg, gg, og};
            }
            cstate = c;
            if (h != h) h = __uint_as_float(0x7FC00000u);                                // never the hand-off pattern
            if (step == s1 - 1 && __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 1u)
                h = __uint_as_float(0x7FC00000u);                                        // a hand-off timed out: fail loudly
            float* xo = hx + ((long)((step & 3) * nch + chain) * 4 * H + (member * 4 + bl) * 16 + cu);
            float* xr = hx + ((long)(((step - 2) & 3) * nch + chain) * 4 * H + (member * 4 + bl) * 16 + cu);
            if (local) {
                __hip_atomic_store(xo, h, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);      // stays in this XCD's L2
                if (step >= 2) __hip_atomic_store((unsigned*)xr, kHandoffSentinel, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            } else {
                __hip_atomic_store(xo, h, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);          // write-through (sc1)
                if (step >= 2) __hip_atomic_store((unsigned*)xr, kHandoffSentinel, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            *yo = h;
            *go = gv;
            cell[sidx] = c;
        }
        LSTM_STAMP(5);              // cell update + stores issued (wave 0)
    }
#ifdef VOCR_LSTM_STAMPS
    if (lane == 0 && (wave == 0 || wave == 3) && g_lstm_stamp_out) {
        unsigned long long* o = g_lstm_stamp_out + ((size_t)blockIdx.x * 2 + (wave == 3)) * 8;
        for (int k = 0; k < 8; ++k) o[k] = st_acc[k];
    }
#endif
}

// The next layer's x-projection, computed WHILE this layer's forward sweep runs (cuDNN's RNN overlaps the two:
// src/models/cnnlstm.py:148-149,288-290).  A sweep is latency-bound: its 4x4x1 MFMAs keep a CU's matrix pipe ~42 % busy.  Four MORE
// waves of the sweep's own workgroup (waves 8 - 11 of lstm_fwd_chain4w<8, true>: one per SIMD, 168 registers each like the chain's)
// multiply the rows the chain has already produced:
//   unit = 16 consecutive steps of the chain = 64 rows x K = H (the chain's OWN direction half of y: a direction's rows appear in its
//   own time order, so neither direction waits for the other) x this member's 256 of the next layer's 8H gate columns (both
//   directions); out[src dir][tgt dir][row][4H'] - the next sweep adds the two source planes (xproj2);
//   A = the 64 x 512 panel (x the inter-layer dropout mask, if any) staged once per unit into LDS as MFMA fragments
//   [16-byte piece kq][row ^ 8 (kq & 1)] (conflict-free 16-byte writes and reads); B = vocr_lstm_xproj_pack's fragment order, one
//   contiguous KB per (32 columns, 8 k) straight from L2 into registers, three chunks in flight; v_mfma_f32_32x32x2_f32, wave tile
//   64 x 64 in 64 AGPRs.
// Why inside the workgroup (all measured, scripts/follow_ab.py, HISTORY of round 6 in DESIGN.md): as a kernel of its own beside the
// sweep (a) a never-stalling MFMA stream takes the matrix pipe from the chain whatever its instruction length, stream priority or
// s_setprio - the sweep ran at HALF speed -, (b) its workgroups land on other XCDs than their chain (the dispatcher's round robin
// continues where the previous kernel stopped) and sc1 stores are not seen there before the producer's final L2 write-back, (c) placed
// first it can keep the sweep's workgroups off the CU.  Here the throttle is explicit: the chain's wave 0 raises `phase` (LDS) when its
// MFMAs of a step begin; at the next yield point (after every 16 MFMAs = 1024 cycles of the pipe, in every wait loop) a follower wave
// goes to the step's two barriers and sits there through the chain's MFMAs, reduction and activation - its window is the chain's
// cell update + hand-off round trip, ~half of a step - and it arrives at the first barrier long before the chain does.
// Gates: rows of chain steps <= joined - 3 are complete when this workgroup has passed `joined` first-barriers (every wave has then
// seen its slice of h of step joined - 2, and a member's cell waves drain their own y stores before they store the next h); the
// chain's last rows: the 32 `done` words (one per cell wave) of all 16 members, stored behind a release.  A chain spread over several
// XCDs (never seen; the hand-off then runs write-through) only trusts the `done` words: its projection runs as a tail.
// Sync among the four follower waves (panel staged / panel consumed): LDS counters, never s_barrier.
// Stores through a wave-uniform 64-bit base (SGPR pair) + a 32-bit per-thread byte offset: the compiler keeps a 64-bit VGPR address per
// stream instead (10 VGPRs in lstm_fwd_chain4w's cell section, which has none to spare beside the follower waves)
__device__ __forceinline__ void sstore_b32(const void* sbase, unsigned voff, float v) {
    asm volatile("global_store_dword %0, %1, %2" ::"v"(voff), "v"(v), "s"(sbase) : "memory");
}
__device__ __forceinline__ void sstore_b32_sc0(const void* sbase, unsigned voff, unsigned v) {
    asm volatile("global_store_dword %0, %1, %2 sc0" ::"v"(voff), "v"(v), "s"(sbase) : "memory");
}
__device__ __forceinline__ void sstore_b32_sc1(const void* sbase, unsigned voff, unsigned v) {
    asm volatile("global_store_dword %0, %1, %2 sc1" ::"v"(voff), "v"(v), "s"(sbase) : "memory");
}
__device__ __forceinline__ void sstore_b128(const void* sbase, unsigned voff, f32x4 v) {
    asm volatile("global_store_dwordx4 %0, %1, %2" ::"v"(voff), "v"(v), "s"(sbase) : "memory");
}

constexpr unsigned kFollowDone = 0xFFFFFFFCu;      // a member's `done` word (0xFF fill) once both its cell waves have cleared their bit
struct FollowArgs {
    const float* wpack;      // next layer's W_ih, vocr_lstm_xproj_pack order; nullptr: nobody follows
    const float* mask;       // [rows][2H] pre-scaled dropout mask or nullptr
    const float* bias;       // [2][4H'] or nullptr
    float* out;              // [2 src][2 tgt][rows][4H']
    unsigned* done;          // the sweep's id block: word [chain*32 + 16 + member] (the upper half of a chain's 32 id words is free)
    unsigned long long* dbg; // -DVOCR_FOLLOW_PROBE builds: stamps (the workspace's first words, unused by a forward sweep)
    int mode;                // -DVOCR_FOLLOW_PROBE builds: parts of the follower switched off (WRONG results)
};

template <bool MASK, typename Yield>
__device__ __forceinline__ void xproj_follow_waves(const FollowArgs& fa, const float* y, float* panel, int* fsync, const SeqRows& sr, int dir, int nrows,
                                                   int chain, int slice, bool local, const int& joined, Yield&& yield, unsigned* status, unsigned* health) {
    constexpr int H = 512, G = 4 * H, TG = 16, NJ = H / 8;
    const int tid = threadIdx.x - 512, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int Tc = sr.steps;
    const long R = sr.total;
    const __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)y, 0, (int)(R * 2 * H * 4), 0x00020000);
    // staging: lane = (row_lo = lane & 7, kq_lo = lane >> 3): 8 rows x 128 contiguous bytes per instruction
    const int srow_lo = lane & 7, skq_lo = lane >> 3;
    // B fragments of this wave's two 32-column tiles: scalar offset = (tile, chunk), vector offset = lane
    const int col0 = slice * 256 + wave * 64;
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)fa.wpack, 0, 2 * 8 * H * H * 4, 0x00020000);
    const int wbase = ((dir * (8 * H / 32) + col0 / 32) * NJ) * 1024;        // bytes; tile ct: + NJ KB, chunk j: + 1 KB
    auto b_load = [&](int ct, int j) {
        const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(wrsrc, lane * 16, wbase + (ct * NJ + j) * 1024, 0);
        return (f32x4){__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3])};
    };
    // the member's 256 columns lie in ONE of the four [src dir][tgt dir] planes
    const int tgt = col0 / G;
    const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(fa.out + (long)(dir * 2 + tgt) * R * G), 0, (int)(R * G * 4), 0x00020000);
    const int li = lane & 31, lk = lane >> 5;
    float bv[2];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) bv[ct] = (dir == 0 && fa.bias) ? fa.bias[col0 + 32 * ct + li] : 0.f;
    int* fstaged = fsync;            // + 1 per wave and unit when its quarter of the panel is in LDS
    int* fconsumed = fsync + 1;      // + 1 per wave and unit when it has read the panel for the last time
    bool failed = false;
    const int units = (Tc + TG - 1) / TG;
    for (int u = 0; u < units; ++u) {
        const int s_lo = u * TG, s_hi = min(s_lo + TG, Tc);
#ifdef VOCR_FOLLOW_PROBE
        if (tid == 0 && slice == 0 && chain < 2 && u < 24) fa.dbg[chain * 64 + u] = __builtin_amdgcn_s_memrealtime();
        if (tid == 0 && slice == 0 && chain < 2 && u < 24) fa.dbg[chain * 64 + 32 + u] = joined;
#endif
        // ---- gate: rows of steps < s_hi are complete, and every wave is done with the previous unit's panel
        {
            const bool by_done = !local || s_hi + 2 > Tc;                   // joined never exceeds Tc
            unsigned spins = 0;
            for (;;) {
                yield();
                bool ok;
                if (by_done) {
                    unsigned v = kFollowDone;
                    if (lane < 16) v = __hip_atomic_load(fa.done + chain * 32 + 16 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    ok = __all(v == kFollowDone);
                    if (ok) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                } else {
                    ok = joined >= s_hi + 2;
                }
                if (ok && __hip_atomic_load(fconsumed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >= 4 * u) break;
                if (__hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 1u || ++spins > (1u << 24)) { failed = true; break; }
                if (by_done) __builtin_amdgcn_s_sleep(8); else __builtin_amdgcn_s_sleep(2);
                if (by_done && !ok) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            }
        }
#ifdef VOCR_FOLLOW_PROBE
        const int pmode = fa.mode;
#else
        constexpr int pmode = 0;
#endif
        // ---- stage this wave's quarter of the 64 x 512 panel
        if (!failed && pmode != 2) {
#pragma unroll 1
            for (int i = 0; i < 32; i += 4) {
                u32x4_t v[4];
                f32x4 mk[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int c = (i + q) * 4 + wave, row = (c & 7) * 8 + srow_lo, kq = (c >> 3) * 8 + skq_lo;
                    const int s = s_lo + (row >> 2), r = row & 3;
                    const bool ok = s < s_hi && r < nrows;
                    const int t = dir == 0 ? s : Tc - 1 - s;
                    const long grow = sr.base + (long)sr.stride * t + r;
                    // rows past the chain's end / the batch: an out-of-range offset reads as zeros
                    v[q] = __builtin_amdgcn_raw_buffer_load_b128(yrsrc, ok ? (int)((grow * 2 * H + dir * H + 4 * kq) * 4) : -16, 0, kPollAux);
                    if (MASK) mk[q] = *(const f32x4*)(fa.mask + (ok ? grow * 2 * H + dir * H + 4 * kq : 0));
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int c = (i + q) * 4 + wave, row = (c & 7) * 8 + srow_lo, kq = (c >> 3) * 8 + skq_lo;
                    f32x4 a = (f32x4){__uint_as_float(v[q][0]), __uint_as_float(v[q][1]), __uint_as_float(v[q][2]), __uint_as_float(v[q][3])};
                    if (MASK) a = a * mk[q];
                    *(f32x4*)(panel + ((long)kq * 64 + (row ^ ((kq & 1) << 3))) * 4) = a;
                }
                yield();
            }
        }
        if (lane == 0) __hip_atomic_fetch_add(fstaged, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        {
            unsigned spins = 0;
            while (__hip_atomic_load(fstaged, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < 4 * (u + 1)) {
                yield();
                __builtin_amdgcn_s_sleep(1);
                if (++spins > (1u << 24)) { failed = true; break; }
            }
        }
        // ---- 64 x 64 wave tile over K = 512
        f32x16 acc[2][2];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
        if (!failed && pmode != 1 && pmode != 2) {
            constexpr int D = 4;                              // B chunks in flight
            f32x4 bq[D][2];
#pragma unroll
            for (int d = 0; d < D; ++d)
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) bq[d][ct] = b_load(ct, d);
            const float* ap = panel + ((long)lk * 64 + (li ^ (lk << 3))) * 4;       // piece kq = 2j + lk, row tile rt: + 32 rows
            f32x4 af[2][2];                                   // [parity of the chunk][row tile]: one chunk ahead
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) af[0][rt] = *(const f32x4*)(ap + 32 * rt * 4);
#pragma unroll 1
            for (int j0 = 0; j0 < NJ; j0 += D) {
#pragma unroll
                for (int d = 0; d < D; ++d) {
                    const int j = j0 + d, jn = min(j + 1, NJ - 1), jl = min(j + D, NJ - 1);      // clamped: the last loads repeat, nothing branches
#pragma unroll
                    for (int rt = 0; rt < 2; ++rt) af[(d + 1) & 1][rt] = *(const f32x4*)(ap + ((long)(2 * jn) * 64 + 32 * rt) * 4);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int e = 0; e < 4; ++e)
#pragma unroll
                        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                            for (int ct = 0; ct < 2; ++ct)
                                acc[rt][ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[d & 1][rt][e], bq[d][ct][e], acc[rt][ct], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    // the slot just consumed takes chunk j + D: D - 1 chunks (3 x 1024 MFMA cycles) of flight
                    if (pmode != 3) {
#pragma unroll
                        for (int ct = 0; ct < 2; ++ct) bq[d][ct] = b_load(ct, jl);
                    }
                    if (pmode == 4) __builtin_amdgcn_s_sleep(8);
                    if (pmode == 5) __builtin_amdgcn_s_sleep(16);
                    yield();
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        if (lane == 0) __hip_atomic_fetch_add(fconsumed, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        // ---- plane[row][4H]: C/D map of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = 32 * rt + (r & 3) + 8 * (r >> 2) + 4 * lk;
                const int s = s_lo + (row >> 2), rr = row & 3;
                const int t = dir == 0 ? s : Tc - 1 - s;
                const long grow = sr.base + (long)sr.stride * t + rr;
                const int off = (s < s_hi && rr < nrows) ? (int)((grow * G + col0 % G + li) * 4) : -4;      // out of range: dropped
#pragma unroll
                for (int ct = 0; ct < 2; ++ct)
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(failed ? __uint_as_float(0x7FC00000u) : acc[rt][ct][r] + bv[ct]), orsrc, off, 32 * ct * 4, 0);
            }
            yield();
        }
    }
#ifdef VOCR_FOLLOW_PROBE
    if (tid == 0 && slice == 0 && chain < 2) fa.dbg[chain * 64 + 30] = __builtin_amdgcn_s_memrealtime();
#endif
    if (failed && lane == 0) raise_timeout(status, health);
}

// Forward sweep for 16 < B <= 32 at H = 512 ("chain4w": 4-row chains with WIDE members): 16 chains of 16 members with 32
// units (128 gate columns) each = ONE 8-wave workgroup per CU.  Same FLOPs per CU and step as two lstm_fwd_chain4v workgroups,
// but nothing shares the CU: a chain no longer moves at the pace of its most-disturbed member (with two workgroups per CU the
// polls took 1470 instead of 690 ticks and the barrier behind them 930 instead of 235), the hand-off has 16 producers instead of
// 32, and every wave has two independent accumulator chains by construction.  K = H split over the 8 waves (64 k each), lane =
// gate column of a 64-column block, two blocks: 128 VGPRs of W_hh per lane, no duplicates; A straight from the hand-off load (one
// 16-byte load per lane and step) through the instruction's A broadcast.  Epilogue: wave w reduces and activates gate w & 3 of
// rows 2(w >> 2), +1 (a wave-uniform activation function), waves 0-1 update the 128 cells.  Ring and reset: lstm_fwd_chain4v.
template <int KQ4, bool FOLLOW>
__global__ __launch_bounds__(FOLLOW ? 768 : 512) void lstm_fwd_chain4w(const float* __restrict__ xproj, const float* __restrict__ whh_f,
                                                        const float* __restrict__ whh_r, const int32_t* __restrict__ lens,
                                                        float* y, float* __restrict__ gates, float* __restrict__ cell,
                                                        float* hx, unsigned* ids, unsigned* status, unsigned* health, int T, int B, int NT4,
                                                        int force_wt, int s0, int s1, int nap, int packed_rows,
                                                        const float* __restrict__ xproj2, FollowArgs fa) {
    constexpr int H = 64 * KQ4;
    constexpr int members = H >> 5;                   // 32 units each
    constexpr int KW = H / 8;                         // k per wave
    constexpr int NL = KW / 4;                        // 16-byte pieces of a row's k-slice = A blocks in use
    static_assert(NL == 16, "one 16-byte load per lane covers the wave's k-slice only at H = 512");
    constexpr int RP = 132;                           // pitch of a partial-tile row (128 columns)
    __shared__ float lds[8 * 4 * RP + 4 + 4 * 128];
    float (*red)[4][RP] = (float (*)[4][RP])lds;                            // [wave][row][column]
    float (*actb)[128] = (float (*)[128])(lds + 8 * 4 * RP + 4);            // [row][gate*32 + unit]
    __shared__ int sig[2];                                                  // step count of the two cell waves' last h store
    __shared__ int phase;                                                   // FOLLOW: steps whose MFMAs wave 0 has begun (the followers' cue)
    __shared__ int fsync[2];                                                // FOLLOW: panel staged / consumed counters of the four follower waves
    extern __shared__ __attribute__((aligned(16))) float panel[];           // FOLLOW: 128 KB, dynamic (a static array of that size makes the
                                                                            // compiler assume one wave per SIMD and pad the register allocation)
    const int nch = 2 * NT4;
    const int chain = (int)(blockIdx.x & 7) + 8 * (int)((blockIdx.x >> 3) & 1), member = blockIdx.x >> 4;
    if (chain >= nch) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int dir = chain / NT4, bt = chain % NT4;
    const int unit0 = member * 32, b0 = bt * 4;
    const int nrows = min(B - b0, 4);
    const int kbase = wave * KW;
    const float* whh = dir ? whh_r : whh_f;
    if (FOLLOW && tid == 0) { phase = 0; fsync[0] = 0; fsync[1] = 0; }          // in front of chain_is_xcd_local's barriers
    const bool local = chain_is_xcd_local(ids + chain * 32, members, member, status, health, (unsigned*)(lds + 8 * 4 * RP), kHandoffSentinel) && !(force_wt & 1);
    if constexpr (FOLLOW) {
        if (tid >= 512) {
            // waves 8 - 11: the next layer's x-projection (xproj_follow_waves above); two barriers per step of the chain, in step
            const SeqRows fsr = seq_rows(lens, T, B, bt, packed_rows);
            if (!seq_rows_fit(fsr, packed_rows)) return;
            const int fTc = fsr.steps;
            int joined = 0;
            auto yield = [&]() {
                if (joined < fTc && __hip_atomic_load(&phase, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) > joined) {
                    __builtin_amdgcn_s_barrier();
                    __builtin_amdgcn_s_barrier();
                    ++joined;
                }
            };
            if (fa.mask) xproj_follow_waves<true>(fa, y, panel, fsync, fsr, dir, nrows, chain, member, local, joined, yield, status, health);
            else xproj_follow_waves<false>(fa, y, panel, fsync, fsr, dir, nrows, chain, member, local, joined, yield, status, health);
            while (joined < fTc) {
                yield();
                __builtin_amdgcn_s_sleep(2);
            }
            return;
        }
        __builtin_amdgcn_s_setprio(3);                // the chain is the latency-bound half of the workgroup
    }

    // resident B operand: column 64*cb + lane of the workgroup's 128 = (gate, local unit) = (col >> 5, col & 31)
    f32x4 wv[2][NL];
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
        const int col = 64 * cb + lane;
        const f32x4* wp = (const f32x4*)(whh + ((long)(col >> 5) * H + unit0 + (col & 31)) * H + kbase);
#pragma unroll
        for (int i = 0; i < NL; ++i) wv[cb][i] = wp[i];
    }
    const int egate = __builtin_amdgcn_readfirstlane(wave) & 3;
    const int erow = 2 * (wave >> 2) + (lane >> 5), eu = lane & 31, ecol = egate * 32 + eu;
    const bool erowok = erow < nrows;
    const int crow = tid >> 5, cu = tid & 31, cb_ = b0 + crow, unit = unit0 + cu;       // cell threads: waves 0, 1
    const bool cellthr = tid < 128 && crow < nrows;
    const int len_b = cellthr ? lens[cb_] : 0;
    const SeqRows sr = seq_rows(lens, T, B, bt, packed_rows);
    const int Tc = sr.steps;
    s1 = min(s1, Tc);
    if (!seq_rows_fit(sr, packed_rows)) {             // the caller's row count does not belong to these lengths: fail loudly, touch nothing
        if (tid == 0) raise_timeout(status, health);
        return;
    }
    // the cell thread's own element of step t: plane-local row = sr.base + sr.stride*t + crow.  Addresses = a wave-uniform 64-bit base of
    // the step (scalar arithmetic) + a 32-bit per-thread byte offset that never changes (the planes stay below 2 GB: host check)
    const long crow0 = (long)dir * sr.total + sr.base + crow;
    const unsigned cvoff = (unsigned)((crow0 * H + unit) * 4);                          // cell; gates: x 4
    const unsigned yvoff = (unsigned)((((long)sr.base + crow) * 2 * H + dir * H + unit) * 4);
    const unsigned rvoff = (unsigned)(((member * 4 + crow) * 32 + cu) * 4);             // hand-off ring block of this member
    float cstate = 0.f;
    if (s0 > 0 && cellthr) {
        const int tp0 = dir == 0 ? s0 - 1 : Tc - s0;
        cstate = cell[(crow0 + (long)sr.stride * tp0) * H + unit];
    }
    bool timed_out = false;
    if (tid < 2) sig[tid] = s0;                       // read by the other waves from step s0 + 1 on: behind step s0's barriers
    // ring hx[step & 3][chain][member][4 rows][32 units]: a member's block of a step = 512 contiguous bytes
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)hx, 0, 4 * nch * 4 * H * 4, 0x00020000);
    int poff;
    {
        const int prow = (lane & 3) < nrows ? (lane & 3) : 0;             // rows clamped, never masked
        const int kk = kbase + 4 * (lane >> 2);
        poff = (((kk >> 5) * 4 + prow) * 32 + (kk & 31)) * 4;
    }
    const unsigned xvoff = (unsigned)((((long)dir * sr.total + sr.base + (erowok ? erow : 0)) * 4 * H + (long)egate * H + unit0 + eu) * 4);
    const long xstep = (long)sr.stride * 4 * H * 4;                // bytes per time step
    auto x_load = [&](int st) {
        const int tt = dir == 0 ? st : Tc - 1 - st;
        float v = *(const float*)((const char*)xproj + tt * xstep + xvoff);
        // second addend: the other source direction's plane of the layer below (xproj_follow_waves writes one plane per direction)
        if (xproj2) v += *(const float*)((const char*)xproj2 + tt * xstep + xvoff);
        return v;
    };
    float xn = s0 < s1 ? x_load(s0) : 0.f;
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int i = 0; i < NL; ++i) asm volatile("" : "+v"(wv[cb][i]));      // complete before the loop (see lstm_fwd_chain4v)

    LSTM_STAMP_DECL;
#ifdef VOCR_FOLLOW_PROBE
    if (FOLLOW && tid == 0 && member == 0 && chain < 2) fa.dbg[chain * 64 + 28] = __builtin_amdgcn_s_memrealtime();
#endif
    for (int step = s0; step < s1; ++step) {
        const int t = dir == 0 ? step : Tc - 1 - step;
        LSTM_STAMP(7);
        const float xp = xn;
        f32x4 acc[2];
        acc[0] = acc[1] = (f32x4){0.f, 0.f, 0.f, 0.f};
        u32x4_t pv;
        if (step > 0) {
            const int toff = (((step - 1) & 3) * nch + chain) * 4 * H * 4;
            unsigned spins = 0;
            // the waves without cells are here a cell update + a store's flight before anything can have arrived: a poll issued now
            // comes back empty and the one that counts queues behind it (2.24 us per step).  They wait (LDS, no memory traffic)
            // until this workgroup's own cell waves have stored (2.12; a further fixed nap has a narrow optimum and is left at 0)
            if (tid >= 128 && step > s0 && nap >= 0) {
                while (__hip_atomic_load(&sig[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < step ||
                       __hip_atomic_load(&sig[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < step)
                    __builtin_amdgcn_s_sleep(1);
                for (int i = 0; i < nap; ++i) __builtin_amdgcn_s_sleep(2);
            }
            for (;;) {
                POLL_FENCE();
                pv = __builtin_amdgcn_raw_buffer_load_b128(xrsrc, toff + poff, 0, kPollAux);
                const unsigned mx = max(max(pv[0], pv[1]), max(pv[2], pv[3]));
                if (__all(mx != kHandoffSentinel) || timed_out) break;
                if (++spins > (1u << 22)) {
                    if (lane == 0) raise_timeout(status, health);
                    timed_out = true;
                    break;
                }
            }
        }
        LSTM_STAMP(0);              // own slice of h_{t-1} arrived (polls)
        if (FOLLOW && tid == 0) __hip_atomic_store(&phase, step + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);      // followers: to the barriers
        xn = x_load(step + 1 < Tc ? step + 1 : step);          // behind the polls: loads return in order
        if (step > 0) {
            static_for<NL>([&](auto bc) {
                constexpr int b = decltype(bc)::value;
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int cb = 0; cb < 2; ++cb)
                        acc[cb] = __builtin_amdgcn_mfma_f32_4x4x1f32(__uint_as_float(pv[e]), wv[cb][b][e], acc[cb], 4, b, 0);
            });
        }
        // acc[cb][r] = partial of (row r, column 64cb + lane)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[wave][r][64 * cb + lane] = acc[cb][r];
        LSTM_STAMP(1);              // MFMA + partial tile to LDS
        __syncthreads();
        LSTM_STAMP(2);
        {
            const float pre = (((red[0][erow][ecol] + red[1][erow][ecol]) + (red[2][erow][ecol] + red[3][erow][ecol])) +
                               ((red[4][erow][ecol] + red[5][erow][ecol]) + (red[6][erow][ecol] + red[7][erow][ecol]))) + xp;
            actb[erow][ecol] = egate == 2 ? tanhf(pre) : sigmoidf_(pre);           // wave-uniform choice
            asm volatile("" : "+v"(xn));      // no load pending at the back edge (see lstm_fwd_chain4v)
        }
        LSTM_STAMP(3);              // reduce + activation
        __syncthreads();
        LSTM_STAMP(4);
        if (cellthr) {
            const bool active = t < len_b;
            const long srow = (long)sr.stride * t;                                      // wave-uniform
            float h = 0.f, c = 0.f;
            f32x4 gv = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (active) {
                const float ig = actb[crow][cu], fg = actb[crow][32 + cu], gg = actb[crow][64 + cu], og = actb[crow][96 + cu];
                c = fg * cstate + ig * gg;
                h = og * tanhf(c);
                gv = (f32x4){ig, fg, gg, og};
            }
            cstate = c;
            if (h != h) h = __uint_as_float(0x7FC00000u);                                // never the hand-off pattern
            if (step == s1 - 1 && __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 1u)
                h = __uint_as_float(0x7FC00000u);                                        // a hand-off timed out: fail loudly
            const char* xo = (const char*)hx + (long)((step & 3) * nch + chain) * (4 * H * 4);
            const char* xr = (const char*)hx + (long)(((step - 2) & 3) * nch + chain) * (4 * H * 4);
            if (local) {                                                                 // stays in this XCD's L2
                sstore_b32_sc0(xo, rvoff, __float_as_uint(h));
                if (step >= 2) sstore_b32_sc0(xr, rvoff, kHandoffSentinel);
            } else {                                                                     // write-through (sc1)
                sstore_b32_sc1(xo, rvoff, __float_as_uint(h));
                if (step >= 2) sstore_b32_sc1(xr, rvoff, kHandoffSentinel);
            }
            sstore_b32((const char*)y + srow * (2 * H * 4), yvoff, h);
            sstore_b128((const char*)gates + srow * (H * 16), cvoff << 2, gv);
            sstore_b32((const char*)cell + srow * (H * 4), cvoff, c);
        }
        if (tid < 128 && lane == 0) __hip_atomic_store(&sig[wave], step + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        LSTM_STAMP(5);              // cell update + stores issued (waves 0, 1)
    }
#ifdef VOCR_FOLLOW_PROBE
    if (FOLLOW && tid == 0 && member == 0 && chain < 2) fa.dbg[chain * 64 + 29] = __builtin_amdgcn_s_memrealtime();
#endif
    if (FOLLOW && tid < 128) {      // this member's last rows are out: the followers' last units wait for all 16 members of the chain
        __builtin_amdgcn_s_waitcnt(0x0F70);                  // vmcnt(0): this cell wave's stores have landed
        if (lane == 0) {
            if (!local) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            __hip_atomic_fetch_and(fa.done + chain * 32 + 16 + member, ~(1u << wave), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
#if defined(VOCR_LSTM_STAMPS) && defined(VOCR_FOLLOW_PROBE)
    if (lane == 0 && (wave == 0 || wave == 7) && fa.dbg && blockIdx.x == 0) {
        unsigned long long* o = fa.dbg + 256 + (wave == 7) * 8;
        for (int k = 0; k < 8; ++k) o[k] = st_acc[k];
    }
#endif
#ifdef VOCR_LSTM_STAMPS
    if (lane == 0 && (wave == 0 || wave == 7) && g_lstm_stamp_out) {
        unsigned long long* o = g_lstm_stamp_out + ((size_t)blockIdx.x * 2 + (wave == 7)) * 8;
        for (int k = 0; k < 8; ++k) o[k] = st_acc[k];
    }
#endif
}

// Gradient of one LSTM cell; shared by the per-step and the persistent backward kernels so both contract the same
// expressions (the two sweeps are compared bit for bit).  Returns the dc carried to the previous step.
__device__ __forceinline__ float lstm_cell_grad(float dh, float dcar, float ig, float fg, float gg, float og, float c, float cprev,
                                                float (&dg)[4]) {
    const float tc = tanhf_(c);
    const float dc = dcar + dh * og * (1.f - tc * tc);
    dg[0] = dc * gg * ig * (1.f - ig);
    dg[1] = dc * cprev * fg * (1.f - fg);
    dg[2] = dc * ig * (1.f - gg * gg);
    dg[3] = dh * tc * og * (1.f - og);
    return dc * fg;
}

// H = 128*NCH.  grid.x = 2 * (H/16) * RT : one workgroup per (direction, 16 units, 16 batch rows).
// whhT = transposed recurrent weights [H][4H] so the B operand (W_hh[n][unit], n running) is contiguous.
// UT = units owned by a workgroup (16, or 8 when 16 would leave half the chip idle: the MFMA N-tile is then half
// used, but each CU pulls half the recurrent weights — the per-CU fetch rate from the Infinity Cache is the limiter).
template <int NCH, int NW, int UT>
__global__ __launch_bounds__(64 * NW) void lstm_bwd_step_fast(const float* __restrict__ dy, const float* __restrict__ whht_f,
                                                          const float* __restrict__ whht_r, const int32_t* __restrict__ lens,
                                                          const float* __restrict__ gates, const float* __restrict__ cell,
                                                          float* __restrict__ dgates, float* __restrict__ dcbuf, int T, int B,
                                                          int step, int RT, int dbg) {
    constexpr int H = 128 * NCH;
    constexpr int WPG = NW / 4;                 // waves per gate block
    constexpr int CH = NCH / WPG;               // 32-wide chunks per lane-quarter
    static_assert(NCH % WPG == 0, "bad wave split");
    __shared__ float red[NW][16][17];
    __builtin_amdgcn_s_setprio(3);      // latency-critical: win issue arbitration against co-resident throughput kernels
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int ublocks = H / UT;
    int bid = blockIdx.x;
    const int bt = bid % RT;
    bid /= RT;
    const int dir = bid / ublocks, unit0 = (bid % ublocks) * UT;
    const int t = dir == 0 ? T - 1 - step : step;
    const int tv = dir == 0 ? t + 1 : t - 1;
    const int lr = lane & 15, q = lane >> 4;
    // n owned by (wave, chunk c, load i, quarter q, element e) = wave range + 128*c + 16*i + 4*q + e (64-B runs per row)
    const int nbase = (wave / WPG) * H + (wave % WPG) * (H / WPG) + q * 4;

    // epilogue operands first (latency hides under the MFMA phase): one cell per thread
    const int eb = bt * 16 + ((tid & 255) >> 4), ej = tid & 15, eunit = unit0 + (ej & (UT - 1));
    const bool ev = eb < B && tid < 256 && ej < UT;
    const int ebs = ev ? eb : 0;
    const int len = lens[ebs];
    const long gbase = (((long)dir * T + t) * B + ebs) * 4 * H + eunit;
    const long cb = ((long)dir * B + ebs) * H + eunit;
    const bool act = ev && t < len;
    float ig = 0.f, fg = 0.f, gg = 0.f, og = 0.f, c = 0.f, cprev = 0.f, dyv = 0.f, dcar = 0.f;
    if (act) {
        const long sidx = (((long)dir * T + t) * B + ebs) * H + eunit;
        const f32x4 gv = *(const f32x4*)(gates + sidx * 4);
        ig = gv[0];
        fg = gv[1];
        gg = gv[2];
        og = gv[3];
        c = cell[sidx];
        const int tp = dir == 0 ? t - 1 : t + 1;
        cprev = (tp >= 0 && tp < len) ? cell[(((long)dir * T + tp) * B + ebs) * H + eunit] : 0.f;
        dyv = dy[((long)t * B + ebs) * 2 * H + dir * H + eunit];
        dcar = step > 0 ? dcbuf[cb] : 0.f;
    }

    // two accumulator chains: a dependent v_mfma_f32_16x16x4_f32 has 40 cycles of latency but issues every 32
    f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f}, acc2 = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (step > 0 && !(dbg & 2)) {
        const int b = bt * 16 + lr;
        const bool bv = b < B;
        const f32x4* ap = (const f32x4*)(dgates + (((long)dir * T + tv) * B + (bv ? b : 0)) * 4 * H + nbase);
        const float* whht = dir ? whht_r : whht_f;
        const f32x4* bp = (const f32x4*)(whht + (long)(unit0 + (lr & (UT - 1))) * 4 * H + nbase);   // lanes past UT mirror
        f32x4 av[2][8], bw[2][8];
#pragma unroll
        for (int i = 0; i < 8; ++i) { av[0][i] = ap[i * 4]; bw[0][i] = bp[i * 4]; }
#pragma unroll
        for (int cidx = 0; cidx < CH; ++cidx) {
            const int cur = cidx & 1, nxt = cur ^ 1;
            if (cidx + 1 < CH) {
#pragma unroll
                for (int i = 0; i < 8; ++i) { av[nxt][i] = ap[(cidx + 1) * 32 + i * 4]; bw[nxt][i] = bp[(cidx + 1) * 32 + i * 4]; }
            }
            __builtin_amdgcn_sched_barrier(0);   // next chunk's loads stay above this chunk's MFMAs
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int e = 0; e < 4; e += 2) {
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(bv ? av[cur][i][e] : 0.f, bw[cur][i][e], acc, 0, 0, 0);
                    acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(bv ? av[cur][i][e + 1] : 0.f, bw[cur][i][e + 1], acc2, 0, 0, 0);
                }
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) red[wave][q * 4 + r][lr] = acc[r] + acc2[r];
    __syncthreads();

    if (ev) {
        if (!act) {
            dgates[gbase] = 0.f;
            dgates[gbase + H] = 0.f;
            dgates[gbase + 2l * H] = 0.f;
            dgates[gbase + 3l * H] = 0.f;
            dcbuf[cb] = 0.f;
        } else {
            const int bl = tid >> 4;
            float rs = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) rs += red[w][bl][ej];
            float dg[4];
            const float dcn = lstm_cell_grad(dyv + rs, dcar, ig, fg, gg, og, c, cprev, dg);
            dgates[gbase] = dg[0];
            dgates[gbase + H] = dg[1];
            dgates[gbase + 2l * H] = dg[2];
            dgates[gbase + 3l * H] = dg[3];
            dcbuf[cb] = dcn;
        }
    }
}

// Backward chain sweep, "K-owner" form.  The obvious (N-owner) form - every workgroup loads the chain's dgates_{t+-1},
// 128 KB per workgroup and 32 times the same bytes per XCD, and multiplies them with its 16 x 4H slice of W_hh^T - spent
// 4.7 of its 6.3 us per step on that load (measured, then removed from the tree).  Here a workgroup
// multiplies the dgates it has just produced itself (16 rows x 64 gate-units, straight from LDS) with its 64 x H slice of
// W_hh and hands out PARTIAL sums instead: out[row][n] for all H units, one 1-KB block per consumer.  The exchange is
// then 32 KB written and 32 KB read per workgroup per step, the reads are 32 fully coalesced dword loads per thread
// summed in a fixed order (deterministic), and no global load sits in front of the MFMAs.
//   MFMA roles: A = W_hh^T fragment (m = unit within a 16-unit tile), B = own dgates (n = batch row); k = 4*ks + q is
//   mapped to (gate q, local unit ks) so that both operands are 16 contiguous floats per lane.
//   D[m = 4q + r][n = lr] = partial of (unit n0 + 4q + r, row lr): four consecutive units per lane -> one 16-byte
//   write-through store per tile, no regrouping.
// Partial buffers are double-buffered by step parity (a producer can only be one step ahead of its slowest consumer).
template <int NCH>
__global__ __launch_bounds__(256) void lstm_bwd_kowner(const float* __restrict__ dy, const float* __restrict__ whht_f,
                                                       const float* __restrict__ whht_r, const int32_t* __restrict__ lens,
                                                       const float* __restrict__ gates, const float* __restrict__ cell,
                                                       float* __restrict__ dgates, float* partials, unsigned* flags, unsigned* ids,
                                                       unsigned* status, unsigned* health, int T, int B, int RT, int force_wt) {
    constexpr int H = 128 * NCH;
    constexpr int members = H / 16;
    constexpr int TPW = members / 4;                  // 16-unit output tiles per wave
    constexpr int DP = 68;                            // LDS row pitch (floats): 16-B reads of 16 rows hit 64 distinct banks
    __shared__ float dgl[16 * DP + 4];                // own dgates [row][gate][local unit] (+ one scratch word)
    const int chain = blockIdx.x & 7, member = blockIdx.x >> 3;
    if (chain >= 2 * RT) return;
    __builtin_amdgcn_s_setprio(3);      // latency-critical: win issue arbitration against co-resident GEMM waves
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int dir = chain / RT, bt = chain % RT, unit0 = member * 16;
    const int lr = lane & 15, q = lane >> 4;
    unsigned* cflags = flags + chain * 32;
    // chain on one XCD: partials and flags stay in its L2 (sc0 stores); the loads stay sc1 (L1 bypassed) because a
    // partial block is rewritten every other step, so an L1 copy of it would be stale
    const bool local = chain_is_xcd_local(ids + chain * 32, members, member, status, health, (unsigned*)(dgl + 16 * DP)) && !(force_wt & 1);

    // resident A fragments: aw[j][i][e] = W_hh[q*H + unit0 + 4i + e][n0 + lr] = whht[n0 + lr][q*H + unit0 + 4i + e]
    f32x4 aw[TPW][4];
    {
        const float* whht = dir ? whht_r : whht_f;
#pragma unroll
        for (int j = 0; j < TPW; ++j) {
            const f32x4* wp = (const f32x4*)(whht + (long)((wave * TPW + j) * 16 + lr) * 4 * H + q * H + unit0);
#pragma unroll
            for (int i = 0; i < 4; ++i) aw[j][i] = wp[i];
        }
    }
    const int b0 = bt * 16;
    const int eb = b0 + (tid >> 4), ej = tid & 15, eunit = unit0 + ej;
    const bool ev = eb < B;
    const int ebs = ev ? eb : b0;
    const int len = lens[ebs];
    float dcar = 0.f;
    bool timed_out = false;
    const __amdgpu_buffer_rsrc_t prsrc = __builtin_amdgcn_make_buffer_rsrc((void*)partials, 0, 2 * 8 * 32 * 32 * 256 * 4, 0x00020000);

    for (int step = 0; step < T; ++step) {
        const int t = dir == 0 ? T - 1 - step : step;
        const bool act = ev && t < len;
        // epilogue operands (written by earlier kernels: plain loads), issued ahead of the wait
        float ig = 0.f, fg = 0.f, gg = 0.f, og = 0.f, c = 0.f, cprev = 0.f, dyv = 0.f;
        if (act) {
            const long sidx = (((long)dir * T + t) * B + ebs) * H + eunit;
            const f32x4 gv = *(const f32x4*)(gates + sidx * 4);
            ig = gv[0];
            fg = gv[1];
            gg = gv[2];
            og = gv[3];
            c = cell[sidx];
            const int tp = dir == 0 ? t - 1 : t + 1;
            cprev = (tp >= 0 && tp < len) ? cell[(((long)dir * T + tp) * B + ebs) * H + eunit] : 0.f;
            dyv = dy[((long)t * B + ebs) * 2 * H + dir * H + eunit];
        }
        float rs = 0.f;
        if (step > 0) {
            if (wave == 3 && !timed_out) {
                unsigned spins = 0;
                for (;;) {
                    unsigned f0 = (unsigned)step;
                    if (lane < members) f0 = __hip_atomic_load(cflags + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (__all(f0 >= (unsigned)step)) break;
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > (1u << 22)) {
                        if (lane == 0) raise_timeout(status, health);
                        timed_out = true;
                        break;
                    }
                }
            }
            __syncthreads();
            // this workgroup's block of every member's partials of the previous step: [member m][row][unit] -> thread = (row, unit)
            const int pbase = (((((step - 1) & 1) * 8 + chain) * 32 + member) * 32 * 256 + tid) * 4;
            float pv[members];
#pragma unroll
            for (int m = 0; m < members; ++m) pv[m] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(prsrc, pbase + m * 1024, 0, 16));   // sc1
#pragma unroll
            for (int m = 0; m < members; ++m) rs += pv[m];
        }
        float dg[4] = {0.f, 0.f, 0.f, 0.f};
        if (act) dcar = lstm_cell_grad(dyv + rs, dcar, ig, fg, gg, og, c, cprev, dg);
        else dcar = 0.f;
        if (step == T - 1 && __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 1u)
            dg[0] = dg[1] = dg[2] = dg[3] = __uint_as_float(0x7FC00000u);        // a hand-off timed out: fail loudly
        if (ev) {
            const long gbase = (((long)dir * T + t) * B + eb) * 4 * H + eunit;
#pragma unroll
            for (int g = 0; g < 4; ++g) dgates[gbase + (long)g * H] = dg[g];     // for the weight/input-gradient GEMMs: plain stores
        }
        if (step + 1 < T) {
#pragma unroll
            for (int g = 0; g < 4; ++g) dgl[(tid >> 4) * DP + g * 16 + ej] = dg[g];
            __syncthreads();
            f32x4 bv[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) bv[i] = *(const f32x4*)&dgl[lr * DP + q * 16 + 4 * i];
            f32x4 acc[TPW];
#pragma unroll
            for (int j = 0; j < TPW; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int j = 0; j < TPW; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(aw[j][i][e], bv[i][e], acc[j], 0, 0, 0);
            // consumer c = wave*TPW + j gets [producer = member][row = lr][units 4q..4q+3]
#pragma unroll
            for (int j = 0; j < TPW; ++j) {
                u32x4_t raw;
#pragma unroll
                for (int e = 0; e < 4; ++e) raw[e] = __float_as_uint(acc[j][e]);
                const int soff = (((((step & 1) * 8 + chain) * 32 + (wave * TPW + j)) * 32 + member) * 256 + lr * 16 + q * 4) * 4;
                if (local) __builtin_amdgcn_raw_buffer_store_b128(raw, prsrc, soff, 0, 1);        // sc0: stays in this XCD's L2
                else __builtin_amdgcn_raw_buffer_store_b128(raw, prsrc, soff, 0, 16);            // write-through (sc1), 16 B
            }
            __builtin_amdgcn_s_waitcnt(0x0F70);              // vmcnt(0): every storing wave drains before the flag
        }
        __syncthreads();                                     // also: dgl is free for the next step
        if (tid == 0) {
            if (local) __hip_atomic_store(cflags + member, (unsigned)(step + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else __hip_atomic_store(cflags + member, (unsigned)(step + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// Backward sweep on 4-row chains with a self-validating hand-off ("bwd chain4v"): the K-owner form of lstm_bwd_kowner
// (a workgroup multiplies its OWN 64 gate gradients into partial dh for all H units and every consumer sums the 32 partial
// blocks of its 16 units in a fixed order) on the geometry of lstm_fwd_chain4v: 2*ceil(B/4) <= 16 chains, workgroups of 4
// waves, two per CU above 8 chains.
//   MFMA: D[unit][row] orientation - A = W_hh^T (lane = unit, resident: 4 gates x 16 k x UW/64 column blocks = 128 VGPRs at
//   H = 512), B = the 4 rows of own dgates.  The instruction's B lane-group broadcast (blgp = 4 + g: every 16-lane group
//   takes group g's lanes; probe scripts/mfma_blgp_probe.hip) lets one VGPR carry 4 different k, so the whole B operand of a
//   step is 4 16-byte LDS reads per lane; a lane ends up with 4 consecutive units of one row = one 16-byte store, and the 16
//   lanes of a consumer's block write its 256 bytes = two whole lines in one instruction.
//   Hand-off: partial blocks live in a ring of 4 step slots [slot][chain][consumer][producer][4 rows][16 units] that is all
//   0xFFFFFFFF before the sweep.  Wave 0 (one cell per lane) re-issues its 32 block loads (sc1: never from the CU's L1, the
//   ring reuses addresses) until no dword holds the pattern; behind the barrier that follows the cell gradients, waves 1-3
//   put the pattern back into the blocks just consumed.  Why that is ordered before the producers' next writes to the slot
//   (3 steps later): the resetting waves wait for their stores' acknowledgements before the NEXT step's barrier, this
//   workgroup's partials of the next step are stored behind that barrier, and a producer writes slot s+4 only after it has
//   consumed everybody's partials of step s+3 > s+1.  No flags, no store drain, one barrier per step.
template <int NCH>
__global__ __launch_bounds__(256) void lstm_bwd_chain4v(const float* __restrict__ dy, const float* __restrict__ whht_f,
                                                        const float* __restrict__ whht_r, const int32_t* __restrict__ lens,
                                                        const float* __restrict__ gates, const float* __restrict__ cell,
                                                        const float* __restrict__ dy_mask, float* __restrict__ dgates, float* ring, unsigned* ids, unsigned* status,
                                                        unsigned* health, float* bias_part, int T, int B, int NT4, int force_wt, int packed_rows) {
    constexpr int H = 128 * NCH;
    constexpr int members = H / 16;
    constexpr int UW = H / 4;                         // units per wave
    constexpr int NCB = UW / 64;                      // 64-unit column blocks per wave
    constexpr int DP = 80;                            // row pitch (floats), = 16 mod 64: the B-fragment read of lane (row j, group g) starts at bank
                                                      // 16 j + 4 g - 16 distinct 4-bank pieces, conflict-free (68: j + g collided, 4-way)
    __shared__ float dgl[2 * 4 * DP + 4];             // own dgates [parity][row][gate*16 + local unit] (+ one scratch word)
    const int nch = 2 * NT4;
    const int chain = nch > 8 ? (int)(blockIdx.x & 7) + 8 * (int)((blockIdx.x >> 3) & 1) : (int)(blockIdx.x & 7);
    const int member = nch > 8 ? blockIdx.x >> 4 : blockIdx.x >> 3;
    if (chain >= nch) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int dir = chain / NT4, bt = chain % NT4, unit0 = member * 16, b0 = bt * 4;
    const bool local = chain_is_xcd_local(ids + chain * 32, members, member, status, health, (unsigned*)(dgl + 2 * 4 * DP), kHandoffSentinel) && !(force_wt & 1);

    // resident A operand: lane = unit u of the wave's column block cb; aw[cb][gate][i][e] = W_hh[gate*H + unit0 + 4i + e][u]
    f32x4 aw[NCB][4][4];
    {
        const float* whht = dir ? whht_r : whht_f;
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) {
            const int u = wave * UW + 64 * cb + lane;
#pragma unroll
            for (int gate = 0; gate < 4; ++gate) {
                const f32x4* wp = (const f32x4*)(whht + (long)u * 4 * H + gate * H + unit0);
#pragma unroll
                for (int i = 0; i < 4; ++i) aw[cb][gate][i] = wp[i];
            }
        }
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
            for (int gate = 0; gate < 4; ++gate)
#pragma unroll
                for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(aw[cb][gate][i]));     // complete before the loop (see lstm_fwd_chain4v)
    }
    const bool cellw = tid < 64;                      // wave 0: one cell (row, unit) per lane
    const int erow = lane >> 4, ej = lane & 15, eb = b0 + erow, eunit = unit0 + ej;
    const bool ev = cellw && eb < B;
    const int ebs = eb < B ? eb : b0;
    const int len = lens[ebs];
    const SeqRows sr = seq_rows(lens, T, B, bt, packed_rows);
    const int Tc = sr.steps;
    if (!seq_rows_fit(sr, packed_rows)) {             // the caller's row count does not belong to these lengths: fail loudly, touch nothing
        if (tid == 0) raise_timeout(status, health);
        return;
    }
    const long prow0 = (long)dir * sr.total + sr.base + (ebs - b0);       // plane-local row of the lane's cell at t = 0
    const long yrow0 = sr.base + (ebs - b0);
    float dcar = 0.f;
    float bs[4] = {0.f, 0.f, 0.f, 0.f};
    bool timed_out = false;
    const int ring_bytes = 4 * nch * 32 * 32 * 256;
    const __amdgpu_buffer_rsrc_t prsrc = __builtin_amdgcn_make_buffer_rsrc((void*)ring, 0, ring_bytes, 0x00020000);

    // the records of a step (gates, c, c_prev, dy of the lane's cell) are fetched one step ahead, behind the polls: loads
    // return in order, a poll issued behind an HBM round trip would wait it out
    f32x4 gv_n = (f32x4){0.f, 0.f, 0.f, 0.f};
    float c_n = 0.f, cprev_n = 0.f, dy_n = 0.f;
    auto fetch = [&](int st) {
        const int tt = dir == 0 ? Tc - 1 - st : st;
        gv_n = (f32x4){0.f, 0.f, 0.f, 0.f};
        c_n = cprev_n = dy_n = 0.f;
        if (ev && tt < len) {
            const long sidx = (prow0 + (long)sr.stride * tt) * H + eunit;
            gv_n = *(const f32x4*)(gates + sidx * 4);
            c_n = cell[sidx];
            const int tp = dir == 0 ? tt - 1 : tt + 1;
            if (tp >= 0 && tp < len) cprev_n = cell[(prow0 + (long)sr.stride * tp) * H + eunit];
            const long di = (yrow0 + (long)sr.stride * tt) * 2 * H + dir * H + eunit;
            dy_n = dy[di];
            if (dy_mask) dy_n *= dy_mask[di];         // the inter-layer dropout's backward, fused into the read
        }
    };
    if (cellw && Tc > 0) fetch(0);

    for (int step = 0; step < Tc; ++step) {
        const int t = dir == 0 ? Tc - 1 - step : step;
        float* dgw = dgl + (step & 1) * 4 * DP;
        float dg[4] = {0.f, 0.f, 0.f, 0.f};
        if (cellw) {
            const bool act = ev && t < len;
            const f32x4 gv = gv_n;
            const float c = c_n, cprev = cprev_n, dyv = dy_n;
            float pv[members];
            // this workgroup's block of every member's partials of the previous step
            const int pbase = (((((step - 1) & 3) * nch + chain) * 32 + member) * 32 * 64 + lane) * 4;
            if (step > 0) {
#pragma unroll
                for (int m = 0; m < members; ++m) pv[m] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(prsrc, pbase + m * 256, 0, kPollAux));
            }
            // everything that does not depend on dh_t is computed while the first round of block loads is in flight (tanh(c) is most
            // of a cell's arithmetic; this section was 1740 of a step's 5760 ticks when it ran behind the polls):
            // dc = dcar + dh*ka, dgates = dc*k0, dc*k1, dc*k2, dh*k3
            const float tc = tanhf_(c);
            const float ka = gv[3] * (1.f - tc * tc);
            const float k0 = gv[2] * gv[0] * (1.f - gv[0]), k1 = cprev * gv[1] * (1.f - gv[1]), k2 = gv[0] * (1.f - gv[2] * gv[2]);
            const float k3 = tc * gv[3] * (1.f - gv[3]);
            float rs = 0.f;
            if (step > 0) {
                unsigned spins = 0;
                for (;;) {
                    unsigned mx = 0u;
#pragma unroll
                    for (int m = 0; m < members; ++m) mx = max(mx, __float_as_uint(pv[m]));
                    if (__all(mx != kHandoffSentinel) || timed_out) break;
                    if (++spins > (1u << 22)) {
                        if (lane == 0) raise_timeout(status, health);
                        timed_out = true;
                        break;
                    }
                    POLL_FENCE();
#pragma unroll
                    for (int m = 0; m < members; ++m) pv[m] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(prsrc, pbase + m * 256, 0, kPollAux));
                }
                // fixed order: a balanced tree over the producers
#pragma unroll
                for (int w = 1; w < members; w *= 2)
#pragma unroll
                    for (int m = 0; m + w < members; m += 2 * w) pv[m] += pv[m + w];
                rs = pv[0];
            }
            if (act) {
                const float dh = dyv + rs;
                const float dc = dcar + dh * ka;
                dg[0] = dc * k0;
                dg[1] = dc * k1;
                dg[2] = dc * k2;
                dg[3] = dh * k3;
                dcar = dc * gv[1];
            } else {
                dcar = 0.f;
            }
#pragma unroll
            for (int g = 0; g < 4; ++g)
                if (dg[g] != dg[g]) dg[g] = __uint_as_float(0x7FC00000u);              // never the hand-off pattern
            if (step == Tc - 1 && __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 1u)
                dg[0] = dg[1] = dg[2] = dg[3] = __uint_as_float(0x7FC00000u);        // a hand-off timed out: fail loudly
#pragma unroll
            for (int g = 0; g < 4; ++g) dgw[erow * DP + g * 16 + ej] = dg[g];
        } else {
            __builtin_amdgcn_s_waitcnt(0x0F70);              // vmcnt(0): last step's resets are acknowledged before this step's barrier
        }
        __syncthreads();
        if (cellw) {
            // off the critical path (the other waves are already multiplying): next step's records, this step's dgates for the GEMMs
            if (step + 1 < Tc) fetch(step + 1);
            if (ev) {
                const long gbase = (prow0 + (long)sr.stride * t) * 4 * H + eunit;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    dgates[gbase + (long)g * H] = dg[g];
                    bs[g] += dg[g];
                }
            }
        }
        if (step + 1 < Tc) {
            // B operand: lane = (16-lane group g, .., row j = lane & 3): bv[q][e] = dgates[row j][k = 16q + 4g + e]
            f32x4 bv[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) bv[q] = *(const f32x4*)&dgw[(lane & 3) * DP + 16 * q + 4 * (lane >> 4)];
            f32x4 acc[NCB];
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) acc[cb] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int q = 0; q < 4; ++q)
                static_for<4>([&](auto gc) {
                    constexpr int g = decltype(gc)::value;
#pragma unroll
                    for (int e = 0; e < 4; ++e)
#pragma unroll
                        for (int cb = 0; cb < NCB; ++cb)
                            acc[cb] = __builtin_amdgcn_mfma_f32_4x4x1f32(aw[cb][q][g][e], bv[q][e], acc[cb], 0, 0, 4 + g);
                });
            // acc[cb][r] = partial of (unit wave*UW + 64cb + 4b + r, row j) for lane 4b + j: 4 consecutive units of one row
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) {
                u32x4_t raw;
#pragma unroll
                for (int e = 0; e < 4; ++e) raw[e] = __float_as_uint(acc[cb][e]);
                const int ug = wave * UW + 64 * cb + 4 * (lane >> 2);
                const int soff = ((((((step & 3) * nch + chain) * 32 + (ug >> 4)) * 32 + member) * 4 + (lane & 3)) * 16 + (ug & 15)) * 4;
                if (local) __builtin_amdgcn_raw_buffer_store_b128(raw, prsrc, soff, 0, 1);        // sc0: stays in this XCD's L2
                else __builtin_amdgcn_raw_buffer_store_b128(raw, prsrc, soff, 0, 16);            // write-through (sc1)
            }
        }
        if (!cellw && step > 0) {
            // the blocks wave 0 consumed in front of the barrier go back to "not written yet" (members * 256 bytes, contiguous)
            const u32x4_t pat = {kHandoffSentinel, kHandoffSentinel, kHandoffSentinel, kHandoffSentinel};
            const int rbase = ((((step - 1) & 3) * nch + chain) * 32 + member) * 32 * 256;
#pragma unroll
            for (int n = 0; n < (members * 256 + 3071) / 3072; ++n) {
                const int off = (n * 192 + (tid - 64)) * 16;
                if (off < members * 256) {
                    if (local) __builtin_amdgcn_raw_buffer_store_b128(pat, prsrc, rbase + off, 0, 1);
                    else __builtin_amdgcn_raw_buffer_store_b128(pat, prsrc, rbase + off, 0, 16);
                }
            }
        }
    }
    // bias gradient of this chain's rows: sum the 4 rows in a fixed order -> bias_part[chain][gate*H + unit]
    if (bias_part) {
        __syncthreads();
        if (cellw) {
#pragma unroll
            for (int g = 0; g < 4; ++g) dgl[erow * DP + g * 16 + ej] = bs[g];
        }
        __syncthreads();
        if (tid < 64) {
            const int g = tid >> 4, u = tid & 15;
            float v = 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) v += dgl[r * DP + g * 16 + u];
            bias_part[(long)chain * 4 * H + g * H + unit0 + u] = v;
        }
    }
}

// Backward sweep for 16 < B <= 32 at H = 512 ("bwd chain4w"): lstm_bwd_chain4v's protocol on the geometry of lstm_fwd_chain4w -
// 16 chains of 16 members with 32 units each, ONE 8-wave workgroup per CU.  A member owns 128 gate gradients (k = gate*32 + unit):
// A = W_hh^T slice (lane = unit of the wave's 64, 4 gates x 32 k: 128 VGPRs, resident), B = own dgates through the B lane-group
// broadcast (8 LDS reads per lane and step), two accumulators (even / odd 16-k groups).  A consumer sums 16 partial blocks of
// [4 rows][32 units] (512 bytes, written whole by one store instruction); cells on waves 0-1, resets by waves 2-7.
template <int NCH>
__global__ __launch_bounds__(512) void lstm_bwd_chain4w(const float* __restrict__ dy, const float* __restrict__ whht_f,
                                                        const float* __restrict__ whht_r, const int32_t* __restrict__ lens,
                                                        const float* __restrict__ gates, const float* __restrict__ cell,
                                                        const float* __restrict__ dy_mask, float* __restrict__ dgates, float* ring, unsigned* ids, unsigned* status,
                                                        unsigned* health, float* bias_part, int T, int B, int NT4, int force_wt, int packed_rows) {
    constexpr int H = 128 * NCH;
    static_assert(H == 512, "8 waves x 64 units");
    constexpr int members = H / 32;
    constexpr int DP = 144;                           // = 16 mod 64 (see lstm_bwd_chain4v; at 132 the eight 16-byte fragment reads of a step cost 32
                                                      // conflict cycles per wave: SQ_LDS_BANK_CONFLICT 1.9e7 per sweep)
    __shared__ float dgl[2 * 4 * DP + 4];             // own dgates [parity][row][gate*32 + local unit] (+ one scratch word)
    const int nch = 2 * NT4;
    const int chain = (int)(blockIdx.x & 7) + 8 * (int)((blockIdx.x >> 3) & 1), member = blockIdx.x >> 4;
    if (chain >= nch) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int dir = chain / NT4, bt = chain % NT4, unit0 = member * 32, b0 = bt * 4;
    const bool local = chain_is_xcd_local(ids + chain * 32, members, member, status, health, (unsigned*)(dgl + 2 * 4 * DP), kHandoffSentinel) && !(force_wt & 1);

    // resident A operand: lane = unit u = 64*wave + lane; aw[gate][i][e] = W_hh[gate*H + unit0 + 4i + e][u], i < 8
    f32x4 aw[4][8];
    {
        const float* whht = dir ? whht_r : whht_f;
        const int u = 64 * wave + lane;
#pragma unroll
        for (int gate = 0; gate < 4; ++gate) {
            const f32x4* wp = (const f32x4*)(whht + (long)u * 4 * H + gate * H + unit0);
#pragma unroll
            for (int i = 0; i < 8; ++i) aw[gate][i] = wp[i];
        }
#pragma unroll
        for (int gate = 0; gate < 4; ++gate)
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("" : "+v"(aw[gate][i]));     // complete before the loop
    }
    const bool cellw = tid < 128;                     // waves 0-1: one cell (row, unit) per thread
    const int erow = tid >> 5, ej = tid & 31, eb = b0 + erow, eunit = unit0 + ej;
    const bool ev = cellw && eb < B;
    const int ebs = eb < B ? eb : b0;
    const int len = lens[cellw ? ebs : b0];
    const SeqRows sr = seq_rows(lens, T, B, bt, packed_rows);
    const int Tc = sr.steps;
    if (!seq_rows_fit(sr, packed_rows)) {             // the caller's row count does not belong to these lengths: fail loudly, touch nothing
        if (tid == 0) raise_timeout(status, health);
        return;
    }
    const long prow0 = (long)dir * sr.total + sr.base + (cellw ? ebs - b0 : 0);      // plane-local row of the thread's cell at t = 0
    const long yrow0 = sr.base + (cellw ? ebs - b0 : 0);
    float dcar = 0.f;
    float bs[4] = {0.f, 0.f, 0.f, 0.f};
    bool timed_out = false;
    // ring [slot][chain][consumer][producer][4 rows][32 units]
    const int ring_bytes = 4 * nch * 16 * 16 * 512;
    const __amdgpu_buffer_rsrc_t prsrc = __builtin_amdgcn_make_buffer_rsrc((void*)ring, 0, ring_bytes, 0x00020000);

    f32x4 gv_n = (f32x4){0.f, 0.f, 0.f, 0.f};
    float c_n = 0.f, cprev_n = 0.f, dy_n = 0.f;
    auto fetch = [&](int st) {
        const int tt = dir == 0 ? Tc - 1 - st : st;
        gv_n = (f32x4){0.f, 0.f, 0.f, 0.f};
        c_n = cprev_n = dy_n = 0.f;
        if (ev && tt < len) {
            const long sidx = (prow0 + (long)sr.stride * tt) * H + eunit;
            // The records of a step are read once and the gate gradients written once: streamed with the non-temporal policy, so that
            // they do not push the partial-block ring (1 MB per XCD, rewritten every four steps) out of L2 - evicted ring lines were
            // 460 MB of HBM writes per sweep beside the 154 MB of gate gradients.
            gv_n = __builtin_nontemporal_load((const f32x4*)(gates + sidx * 4));
            c_n = __builtin_nontemporal_load(cell + sidx);
            const int tp = dir == 0 ? tt - 1 : tt + 1;
            if (tp >= 0 && tp < len) cprev_n = __builtin_nontemporal_load(cell + (prow0 + (long)sr.stride * tp) * H + eunit);
            const long di = (yrow0 + (long)sr.stride * tt) * 2 * H + dir * H + eunit;
            dy_n = __builtin_nontemporal_load(dy + di);
            if (dy_mask) dy_n *= __builtin_nontemporal_load(dy_mask + di);         // the inter-layer dropout's backward, fused into the read
        }
    };
    if (cellw && Tc > 0) fetch(0);

    LSTM_STAMP_DECL;
    for (int step = 0; step < Tc; ++step) {
        const int t = dir == 0 ? Tc - 1 - step : step;
        float* dgw = dgl + (step & 1) * 4 * DP;
        LSTM_STAMP(7);
        float dg[4] = {0.f, 0.f, 0.f, 0.f};
        if (cellw) {
            const bool act = ev && t < len;
            const f32x4 gv = gv_n;
            const float c = c_n, cprev = cprev_n, dyv = dy_n;
            float pv[members];
            // this workgroup's block of every member's partials of the previous step
            const int pbase = (((((step - 1) & 3) * nch + chain) * 16 + member) * 16 * 128 + tid) * 4;
            if (step > 0) {
#pragma unroll
                for (int m = 0; m < members; ++m) pv[m] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(prsrc, pbase + m * 512, 0, kPollAux));
            }
            // everything that does not depend on dh_t is computed while the first round of block loads is in flight (tanh(c) is most
            // of a cell's arithmetic; this section was 1740 of a step's 5760 ticks when it ran behind the polls):
            // dc = dcar + dh*ka, dgates = dc*k0, dc*k1, dc*k2, dh*k3
            const float tc = tanhf_(c);
            const float ka = gv[3] * (1.f - tc * tc);
            const float k0 = gv[2] * gv[0] * (1.f - gv[0]), k1 = cprev * gv[1] * (1.f - gv[1]), k2 = gv[0] * (1.f - gv[2] * gv[2]);
            const float k3 = tc * gv[3] * (1.f - gv[3]);
            float rs = 0.f;
            if (step > 0) {
                unsigned spins = 0;
                for (;;) {
                    unsigned mx = 0u;
#pragma unroll
                    for (int m = 0; m < members; ++m) mx = max(mx, __float_as_uint(pv[m]));
                    if (__all(mx != kHandoffSentinel) || timed_out) break;
                    if (++spins > (1u << 22)) {
                        if (lane == 0) raise_timeout(status, health);
                        timed_out = true;
                        break;
                    }
                    POLL_FENCE();
#pragma unroll
                    for (int m = 0; m < members; ++m) pv[m] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(prsrc, pbase + m * 512, 0, kPollAux));
                }
                LSTM_STAMP(0);      // partial blocks of the previous step arrived (polls)
                // fixed order: a balanced tree over the producers
#pragma unroll
                for (int w = 1; w < members; w *= 2)
#pragma unroll
                    for (int m = 0; m + w < members; m += 2 * w) pv[m] += pv[m + w];
                rs = pv[0];
            }
            if (act) {
                const float dh = dyv + rs;
                const float dc = dcar + dh * ka;
                dg[0] = dc * k0;
                dg[1] = dc * k1;
                dg[2] = dc * k2;
                dg[3] = dh * k3;
                dcar = dc * gv[1];
            } else {
                dcar = 0.f;
            }
#pragma unroll
            for (int g = 0; g < 4; ++g)
                if (dg[g] != dg[g]) dg[g] = __uint_as_float(0x7FC00000u);              // never the hand-off pattern
            if (step == Tc - 1 && __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 1u)
                dg[0] = dg[1] = dg[2] = dg[3] = __uint_as_float(0x7FC00000u);        // a hand-off timed out: fail loudly
#pragma unroll
            for (int g = 0; g < 4; ++g) dgw[erow * DP + g * 32 + ej] = dg[g];
        } else {
            __builtin_amdgcn_s_waitcnt(0x0F70);              // vmcnt(0): last step's resets are acknowledged before this step's barrier
        }
        LSTM_STAMP(1);              // sum + cell gradient + dgates to LDS and memory (waves 0, 1)
        __syncthreads();
        if (cellw) {
            // off the critical path (the other waves are already multiplying): next step's records, this step's dgates for the GEMMs
            if (step + 1 < Tc) fetch(step + 1);
            if (ev) {
                const long gbase = (prow0 + (long)sr.stride * t) * 4 * H + eunit;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    __builtin_nontemporal_store(dg[g], dgates + gbase + (long)g * H);
                    bs[g] += dg[g];
                }
            }
        }
        LSTM_STAMP(2);
        if (step + 1 < Tc) {
            // B operand: lane = (16-lane group g, row j = lane & 3): bv[q][e] = dgates[row j][k = 16q + 4g + e], q < 8
            f32x4 bv[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) bv[q] = *(const f32x4*)&dgw[(lane & 3) * DP + 16 * q + 4 * (lane >> 4)];
            f32x4 acc[2];
            acc[0] = acc[1] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int q = 0; q < 8; ++q)
                static_for<4>([&](auto gc) {
                    constexpr int g = decltype(gc)::value;
#pragma unroll
                    for (int e = 0; e < 4; ++e)     // k = 16q + 4g + e = gate (q >> 1), 4*(4*(q & 1) + g) + e within the gate
                        acc[q & 1] = __builtin_amdgcn_mfma_f32_4x4x1f32(aw[q >> 1][4 * (q & 1) + g][e], bv[q][e], acc[q & 1], 0, 0, 4 + g);
                });
            // acc[r] = partial of (unit 64*wave + 4b + r, row j) for lane 4b + j
            u32x4_t raw;
#pragma unroll
            for (int e = 0; e < 4; ++e) raw[e] = __float_as_uint(acc[0][e] + acc[1][e]);
            const int ug = 64 * wave + 4 * (lane >> 2);
            const int soff = ((((((step & 3) * nch + chain) * 16 + (ug >> 5)) * 16 + member) * 4 + (lane & 3)) * 32 + (ug & 31)) * 4;
            if (local) __builtin_amdgcn_raw_buffer_store_b128(raw, prsrc, soff, 0, 1);        // sc0: stays in this XCD's L2
            else __builtin_amdgcn_raw_buffer_store_b128(raw, prsrc, soff, 0, 16);            // write-through (sc1)
        }
        LSTM_STAMP(3);              // LDS fragments + MFMA + partial stores issued
        if (!cellw && step > 0) {
            // the 16 blocks waves 0-1 consumed in front of the barrier go back to "not written yet" (8 KB, contiguous)
            const u32x4_t pat = {kHandoffSentinel, kHandoffSentinel, kHandoffSentinel, kHandoffSentinel};
            const int rbase = ((((step - 1) & 3) * nch + chain) * 16 + member) * 16 * 512;
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                const int off = (n * 384 + (tid - 128)) * 16;
                if (off < 16 * 512) {
                    if (local) __builtin_amdgcn_raw_buffer_store_b128(pat, prsrc, rbase + off, 0, 1);
                    else __builtin_amdgcn_raw_buffer_store_b128(pat, prsrc, rbase + off, 0, 16);
                }
            }
        }
    }
#ifdef VOCR_LSTM_STAMPS
    if (lane == 0 && (wave == 0 || wave == 7) && g_lstm_stamp_out) {
        unsigned long long* o = g_lstm_stamp_out + ((size_t)(256 + blockIdx.x) * 2 + (wave == 7)) * 8;
        for (int k = 0; k < 8; ++k) o[k] = st_acc[k];
    }
#endif
    if (bias_part) {
        __syncthreads();
        if (cellw) {
#pragma unroll
            for (int g = 0; g < 4; ++g) dgl[erow * DP + g * 32 + ej] = bs[g];
        }
        __syncthreads();
        if (tid < 128) {
            const int g = tid >> 5, u = tid & 31;
            float v = 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) v += dgl[r * DP + g * 32 + u];
            bias_part[(long)chain * 4 * H + g * H + unit0 + u] = v;
        }
    }
}

// dbias[dir][j] = sum over the direction's batch-tile chains, fixed order
__global__ void lstm_bias_combine_kernel(const float* __restrict__ part, float* __restrict__ dbias, int G, int nt) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x, dir = blockIdx.y;
    if (j >= G) return;
    float v = 0.f;
    for (int c = 0; c < nt; ++c) v += part[(long)(dir * nt + c) * G + j];
    dbias[(long)dir * G + j] = v;
}

template <int KQ4>
bool launch_fwd_fast(int rt, dim3 grid, hipStream_t s, const float* xproj, const float* wf, const float* wr, const int32_t* lens,
                     float* y, float* gates, float* cell, int T, int B, int step) {
#ifdef VOCR_LSTM_DIAG      // diagnostic builds only: parts of the per-step kernels switched off, WRONG results - never in libvocr.so
    static const int dbg = getenv("VOCR_LSTM_DEBUG") ? atoi(getenv("VOCR_LSTM_DEBUG")) : 0;
#else
    constexpr int dbg = 0;
#endif
    switch (rt) {
        case 1: lstm_fwd_step_fast<KQ4, 1><<<grid, 256, 0, s>>>(xproj, wf, wr, lens, y, gates, cell, T, B, step, dbg); return true;
        case 2: lstm_fwd_step_fast<KQ4, 2><<<grid, 256, 0, s>>>(xproj, wf, wr, lens, y, gates, cell, T, B, step, dbg); return true;
        case 3: lstm_fwd_step_fast<KQ4, 3><<<grid, 256, 0, s>>>(xproj, wf, wr, lens, y, gates, cell, T, B, step, dbg); return true;
        case 4: lstm_fwd_step_fast<KQ4, 4><<<grid, 256, 0, s>>>(xproj, wf, wr, lens, y, gates, cell, T, B, step, dbg); return true;
    }
    return false;
}


// ------------------------------------------------------------------------------------------------ x-projection behind the sweep
// (the follower itself: xproj_follow_waves in front of lstm_fwd_chain4w)
__global__ __launch_bounds__(256) void lstm_xproj_pack_kernel(const float* __restrict__ w_f, const float* __restrict__ w_r, float* __restrict__ wpack, int H) {
    // wpack[src dir d][column tile ct (8H/32)][chunk j (H/8)][lane][4]: lane (c = lane & 31, hh = lane >> 5) holds
    // W[32 ct + c][d*H + 8j + 4hh .. +3], W = [w_f; w_r] ([8H][2H])
    const long piece = (long)blockIdx.x * 256 + threadIdx.x;
    const int lane = piece & 63;
    const long rest = piece >> 6;
    const int nj = H / 8, nct = 8 * H / 32;
    const int j = rest % nj, ct = (rest / nj) % nct, d = rest / ((long)nj * nct);
    if (d >= 2) return;
    const int col = 32 * ct + (lane & 31);
    const float* w = col < 4 * H ? w_f + (long)col * 2 * H : w_r + (long)(col - 4 * H) * 2 * H;
    ((f32x4*)wpack)[piece] = *(const f32x4*)(w + d * H + 8 * j + 4 * (lane >> 5));
}

inline bool aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

// workgroups of 256 threads that are certainly co-resident: one per CU (a persistent sweep deadlocks if a
// workgroup it waits for is not running; the kernels need < 1/4 of a CU's registers and LDS, so one per CU is safe)
int resident_workgroup_capacity() {
    static int cus[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
    if (!cus[dev]) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return 0;
        cus[dev] = prop.multiProcessorCount;
    }
    return cus[dev];
}

}  // namespace

// ---- which sweep a shape runs on.  Newest first; a shape takes the first one it fits:
//   wide4    lstm_fwd_chain4w / lstm_bwd_chain4w   16 < B <= 32, H = 512: 4-row chains of 16 wide members, one 8-wave workgroup per CU,
//                                                  self-validating hand-off (the BASELINE configuration)
//   chain4   lstm_fwd_chain4v / lstm_bwd_chain4v   B <= 32, H in {256, 512}: 4-row chains, self-validating hand-off
//   chain16  lstm_fwd_chain / lstm_bwd_kowner      B <= 64, H in {64 (forward only), 128, 256, 512}: 16-row chains, arrival flags
//   step     lstm_*_step_fast / lstm_*_step_kernel one launch per time step (any shape; also when the grid cannot be co-resident)
// (Round 2's 8-row flag chains were no shape's default since round 3 and are gone.)
// Runtime knobs (include/vocr.h): VOCR_LSTM_SWEEP=<name> starts the search at that kind for both directions - `step` is the one an
// operator needs: several processes sharing one GPU, whose persistent sweeps would wait for each other's CUs - VOCR_LSTM_SWEEP_FWD /
// VOCR_LSTM_SWEEP_BWD for one direction; VOCR_LSTM_PERSISTENT=0 is accepted as the old spelling of VOCR_LSTM_SWEEP=step; an unknown
// value is an error of the call (VOCR_EINVAL), not a silent default.  VOCR_LSTM_WRITE_THROUGH=1 forces the hand-off mode of a chain
// spread over several XCDs.
enum SweepKind { SWEEP_WIDE4 = 0, SWEEP_CHAIN4 = 1, SWEEP_CHAIN16 = 2, SWEEP_STEP = 3, SWEEP_BAD = -1 };

static SweepKind sweep_floor(bool backward) {
    static int cache[2] = {-2, -2};
    if (cache[backward] == -2) {
        const char* v = getenv(backward ? "VOCR_LSTM_SWEEP_BWD" : "VOCR_LSTM_SWEEP_FWD");
        if (!v) v = getenv("VOCR_LSTM_SWEEP");
        int k = SWEEP_WIDE4;
        if (v) {
            if (!strcmp(v, "wide4") || !*v) k = SWEEP_WIDE4;
            else if (!strcmp(v, "chain4")) k = SWEEP_CHAIN4;
            else if (!strcmp(v, "chain16")) k = SWEEP_CHAIN16;
            else if (!strcmp(v, "step")) k = SWEEP_STEP;
            else k = SWEEP_BAD;
        } else {
            const char* p = getenv("VOCR_LSTM_PERSISTENT");
            if (p && !strcmp(p, "0")) k = SWEEP_STEP;
        }
        cache[backward] = k;
    }
    return (SweepKind)cache[backward];
}

#define VOCR_CHECK_SWEEP_ENV(backward) \
    VOCR_CHECK_ARG(sweep_floor(backward) != SWEEP_BAD, "VOCR_LSTM_SWEEP%s: unknown sweep kind (wide4, chain4, chain16, step)", "")

static bool sweep_write_through() {
    static const int v = getenv("VOCR_LSTM_WRITE_THROUGH") ? atoi(getenv("VOCR_LSTM_WRITE_THROUGH")) : 0;
    return v != 0;
}

// `persistent_ok`: the shape and its buffers allow a persistent sweep at all (fast path, alignment, 32-bit offsets, co-residency)
static SweepKind lstm_sweep_kind(bool backward, int b, int h, bool persistent_ok) {
    const SweepKind floor_ = sweep_floor(backward);
    if (!persistent_ok || floor_ == SWEEP_STEP || floor_ == SWEEP_BAD || b <= 0) return SWEEP_STEP;
    const int cap = resident_workgroup_capacity();
    const int nt4 = (b + 3) / 4;
    const bool h45 = h == 512 || h == 256;
    if (floor_ <= SWEEP_WIDE4 && h == 512 && nt4 > 4 && nt4 <= 8 && 256 <= cap) return SWEEP_WIDE4;
    if (floor_ <= SWEEP_CHAIN4 && 2 * nt4 <= 16 && h45 && (2 * nt4 > 8 ? 16 : 8) * (h / 16) <= 2 * cap) return SWEEP_CHAIN4;
    return SWEEP_CHAIN16;
}

// The self-validating forward sweeps keep a 4-KB block [XCC ids 512 words | status word | ...] directly in front of their hand-off
// ring, so that one 0xFF fill prepares all three.
static int lstm_fwd_selfval_prep(void* block, size_t ring_bytes, bool first_range, hipStream_t s) {
    // a later range of a sweep keeps the ring, but its workgroups may sit on other CUs: the ids (and the status word) start over
    if (hipMemsetAsync(block, 0xFF, first_range ? 4096 + ring_bytes : 4096, s) != hipSuccess) {
        vocr_set_error("vocr_lstm_fwd: memset failed");
        return VOCR_ELAUNCH;
    }
    return VOCR_OK;
}

static size_t lstm_ws_handoff_offset(int b, int h) {
    const size_t o = (size_t)6 * 2 * b * h * sizeof(float) + 4096 + ((size_t)16 << 20) + (size_t)16 * 4 * h * sizeof(float) +
                     (size_t)64 * 4 * h * sizeof(float);
    return (o + 255) / 256 * 256;
}

extern "C" size_t vocr_lstm_workspace_bytes(int t, int b, int h) {
    if (t <= 0 || b <= 0 || h <= 0) return 0;
    // dc carry [2][B][H] (+ spare); 4 KiB of arrival flags / status; 16 MiB of partial-sum blocks for the backward chain
    // sweep (2 parities x 8 chains x 32 consumers x 32 producers x 1 KiB)
    // + per-chain bias-gradient rows [8][4H]
    // + 64 partial rows [4H] for the fixed-order column sums of the bias gradient (paths without in-sweep accumulation)
    // + the forward chain sweep's hand-off ring [4 step slots][16 chains][4 rows][H] (must survive from one vocr_lstm_fwd_range
    //   call of a sweep to the next)
    return lstm_ws_handoff_offset(b, h) + 4096 + (size_t)4 * 16 * 4 * h * sizeof(float);
}

static bool lstm_packed_kind_ok(int b, int h);

static int lstm_fwd_impl(const float* xproj, const float* whh_fwd, const float* whh_rev, const int32_t* lens, float* y,
                         float* gates, float* cell, void* workspace, int t, int b, int h, int step_begin, int step_end,
                         int packed_rows, int32_t* health, void* stream, const float* xproj2 = nullptr, const FollowArgs* follow = nullptr) {
    VOCR_CHECK_ARG(xproj && whh_fwd && whh_rev && lens && y && gates && cell && workspace, "vocr_lstm_fwd: null pointer");
    VOCR_CHECK_ARG(t > 0 && b > 0 && b <= 64 && h > 0 && h % 16 == 0, "vocr_lstm_fwd: need 1<=B<=64 and H%%16==0 (B=%d H=%d)", b, h);
    VOCR_CHECK_ARG(0 <= step_begin && step_begin < step_end && step_end <= t, "vocr_lstm_fwd: bad step range [%d, %d) of %d", step_begin, step_end, t);
    VOCR_CHECK_SWEEP_ENV(false);
        hipStream_t s = (hipStream_t)stream;
    const int rt = (b + 15) / 16;
    const bool fast = (h == 64 || h == 128 || h == 256 || h == 512) && aligned16(whh_fwd) && aligned16(whh_rev) && aligned16(y) &&
                      aligned16(gates);
    VOCR_CHECK_ARG(aligned16(gates), "vocr_lstm_fwd: gates must be 16-byte aligned");
    const dim3 grid(2 * (h / 4));
    // the chain kernels address y through a buffer descriptor with 32-bit byte offsets
    const bool fits32 = (packed_rows ? (long)packed_rows : (long)t * b) * 2 * h * 4 < (1l << 31);
    const SweepKind kind = lstm_sweep_kind(false, b, h, fast && fits32 && 8 * (h / 16) <= resident_workgroup_capacity());
    if (packed_rows) {
        VOCR_CHECK_ARG(lstm_packed_kind_ok(b, h) && (kind == SWEEP_WIDE4 || kind == SWEEP_CHAIN4) && step_begin == 0 && step_end == t,
                       "vocr_lstm_fwd_packed: the packed row layout needs a 4-row chain sweep (ask vocr_lstm_packed_supported; B=%d H=%d)", b, h);
        VOCR_CHECK_ARG(packed_rows % 4 == 0 && packed_rows >= 4 * (1 + 2 * ((b + 3) / 4)), "vocr_lstm_fwd_packed: bad row count %d", packed_rows);
    }
    VOCR_CHECK_ARG((!xproj2 && !follow) || (kind == SWEEP_WIDE4 && step_begin == 0 && step_end == t),
                   "vocr_lstm_fwd_lead: two-plane x-projections / a follower need the wide 4-row chain sweep (ask vocr_lstm_follow_supported; B=%d H=%d)", b, h);
    if (kind != SWEEP_STEP) {
        // arrival flags: [chain <= 8][32 workgroups] at [0..255], XCC ids at [256..511]; status words at [512..]
        unsigned* flags = (unsigned*)workspace;
        unsigned* status = flags + 512;          // the sweep's own status word is per call; a timeout is also reported in the caller's health word
        unsigned* hword = (unsigned*)health;
        const dim3 cg(8 * (h / 16));
        const int fwt = sweep_write_through() ? 1 : 0;
        const int nt4 = (b + 3) / 4;
        if (kind == SWEEP_WIDE4 || kind == SWEEP_CHAIN4) {
            // self-validating hand-off: the ring starts as the "not written yet" pattern (first range of a sweep only)
            unsigned* blk = (unsigned*)((char*)workspace + lstm_ws_handoff_offset(b, h));
            float* hx = (float*)(blk + 1024);
            if (lstm_fwd_selfval_prep(blk, (size_t)4 * 2 * nt4 * 4 * h * sizeof(float), step_begin == 0, s) != VOCR_OK) return VOCR_ELAUNCH;
            if (kind == SWEEP_WIDE4) {
                static const int nap4w = VOCR_EXPERIMENT_INT("VOCR_LSTM_NAP", 0);      // -1: polls start at once (experiments)
                if (follow) {
                    // waves 8 - 11 of every workgroup: the next layer's x-projection (xproj_follow_waves); 128 KB of dynamic LDS for their panel
                    constexpr int kPanelBytes = 128 * 64 * 4 * 4;
                    static bool lds_ok = false;
                    if (!lds_ok) {
                        const hipError_t e = hipFuncSetAttribute((const void*)lstm_fwd_chain4w<8, true>, hipFuncAttributeMaxDynamicSharedMemorySize, kPanelBytes);
                        if (e != hipSuccess) {
                            vocr_set_error("vocr_lstm_fwd_lead: hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed: %s", hipGetErrorString(e));
                            return VOCR_ELAUNCH;
                        }
                        lds_ok = true;
                    }
                    FollowArgs fa = *follow;
                    fa.done = blk;
                    fa.dbg = (unsigned long long*)workspace;
#ifdef VOCR_FOLLOW_PROBE
                    fa.mode = getenv("VOCR_FOLLOW_MODE") ? atoi(getenv("VOCR_FOLLOW_MODE")) : 0;
#endif
                    lstm_fwd_chain4w<8, true><<<256, 768, kPanelBytes, s>>>(xproj, whh_fwd, whh_rev, lens, y, gates, cell, hx, blk, blk + 512, hword, t, b, nt4, fwt, step_begin,
                                                                           step_end, nap4w, packed_rows, xproj2, fa);
                } else {
                    lstm_fwd_chain4w<8, false><<<256, 512, 0, s>>>(xproj, whh_fwd, whh_rev, lens, y, gates, cell, hx, blk, blk + 512, hword, t, b, nt4, fwt, step_begin,
                                                                  step_end, nap4w, packed_rows, xproj2, FollowArgs{nullptr, nullptr, nullptr, nullptr, nullptr, (unsigned long long*)workspace, 0});
                }
            } else {
                const dim3 cg4((2 * nt4 > 8 ? 16 : 8) * (h / 16));
                if (h == 512) lstm_fwd_chain4v<8><<<cg4, 256, 0, s>>>(xproj, whh_fwd, whh_rev, lens, y, gates, cell, hx, blk, blk + 512, hword, t, b, nt4, fwt, step_begin, step_end, packed_rows);
                else lstm_fwd_chain4v<4><<<cg4, 256, 0, s>>>(xproj, whh_fwd, whh_rev, lens, y, gates, cell, hx, blk, blk + 512, hword, t, b, nt4, fwt, step_begin, step_end, packed_rows);
            }
            VOCR_CHECK_LAUNCH("vocr_lstm_fwd(4-row chains, self-validating)");
            return VOCR_OK;
        }
        if (hipMemsetAsync(flags, 0, 520 * sizeof(unsigned), s) != hipSuccess) {
            vocr_set_error("vocr_lstm_fwd: memset failed");
            return VOCR_ELAUNCH;
        }
#define VOCR_CHAIN(KQ4) lstm_fwd_chain<KQ4><<<cg, 256, 0, s>>>(xproj, whh_fwd, whh_rev, lens, y, gates, cell, flags, flags + 256, status, hword, t, b, rt, fwt, step_begin, step_end)
        if (h == 64) VOCR_CHAIN(1); else if (h == 128) VOCR_CHAIN(2); else if (h == 256) VOCR_CHAIN(4); else VOCR_CHAIN(8);
#undef VOCR_CHAIN
        VOCR_CHECK_LAUNCH("vocr_lstm_fwd(chain)");
        return VOCR_OK;
    }
    for (int step = step_begin; step < step_end; ++step) {
        if (fast) {
            switch (h) {
                case 64: launch_fwd_fast<1>(rt, grid, s, xproj, whh_fwd, whh_rev, lens, y, gates, cell, t, b, step); break;
                case 128: launch_fwd_fast<2>(rt, grid, s, xproj, whh_fwd, whh_rev, lens, y, gates, cell, t, b, step); break;
                case 256: launch_fwd_fast<4>(rt, grid, s, xproj, whh_fwd, whh_rev, lens, y, gates, cell, t, b, step); break;
                default: launch_fwd_fast<8>(rt, grid, s, xproj, whh_fwd, whh_rev, lens, y, gates, cell, t, b, step); break;
            }
        } else {
            lstm_fwd_step_kernel<<<grid, 256, 0, s>>>(xproj, whh_fwd, whh_rev, lens, y, gates, cell, t, b, h, step);
        }
    }
    VOCR_CHECK_LAUNCH("vocr_lstm_fwd");
    return VOCR_OK;
}

extern "C" int vocr_lstm_fwd_range(const float* xproj, const float* whh_fwd, const float* whh_rev, const int32_t* lens, float* y,
                                   float* gates, float* cell, void* workspace, int t, int b, int h, int step_begin, int step_end,
                                   int32_t* health, void* stream) {
    return lstm_fwd_impl(xproj, whh_fwd, whh_rev, lens, y, gates, cell, workspace, t, b, h, step_begin, step_end, 0, health, stream);
}

extern "C" int vocr_lstm_fwd(const float* xproj, const float* whh_fwd, const float* whh_rev, const int32_t* lens, float* y,
                             float* gates, float* cell, void* workspace, int t, int b, int h, int32_t* health, void* stream) {
    return lstm_fwd_impl(xproj, whh_fwd, whh_rev, lens, y, gates, cell, workspace, t, b, h, 0, t, 0, health, stream);
}

// ---- packed row layout (struct SeqRows above; include/vocr.h)
static bool lstm_packed_kind_ok(int b, int h) {
    // the predicate of lstm_fwd_impl / lstm_bwd_impl minus what depends on the buffers (alignment; 32-bit offsets: rows * 2H * 4 < 2 GB,
    // the caller's to check): a device or partition that cannot hold the persistent grid answers "not supported" here
    const bool fits = 8 * (h / 16) <= resident_workgroup_capacity();
    const SweepKind kf = lstm_sweep_kind(false, b, h, fits), kb = lstm_sweep_kind(true, b, h, fits);
    return (kf == SWEEP_WIDE4 || kf == SWEEP_CHAIN4) && (kb == SWEEP_WIDE4 || kb == SWEEP_CHAIN4);
}

extern "C" int vocr_lstm_packed_supported(int b, int h) {
    if (b <= 0 || b > 32 || (h != 256 && h != 512)) return 0;
    return lstm_packed_kind_ok(b, h) ? 1 : 0;
}

extern "C" int vocr_lstm_fwd_packed(const float* xproj, const float* whh_fwd, const float* whh_rev, const int32_t* lens, float* y,
                                    float* gates, float* cell, void* workspace, int t, int b, int h, int rows, int32_t* health,
                                    void* stream) {
    VOCR_CHECK_ARG(rows > 0, "vocr_lstm_fwd_packed: rows must be positive");
    return lstm_fwd_impl(xproj, whh_fwd, whh_rev, lens, y, gates, cell, workspace, t, b, h, 0, t, rows, health, stream);
}

// ---- the next layer's x-projection behind the sweep (lstm_xproj_follow above)
extern "C" int vocr_lstm_follow_supported(int b, int h) {
    if (b <= 16 || b > 32 || h != 512) return 0;
    return lstm_sweep_kind(false, b, h, 8 * (h / 16) <= resident_workgroup_capacity()) == SWEEP_WIDE4 ? 1 : 0;
}

extern "C" size_t vocr_lstm_xproj_pack_bytes(int h) { return h > 0 ? (size_t)2 * 8 * h * h * sizeof(float) : 0; }

extern "C" int vocr_lstm_xproj_pack(const float* w_ih_fwd, const float* w_ih_rev, float* wpack, int h, void* stream) {
    VOCR_CHECK_ARG(w_ih_fwd && w_ih_rev && wpack, "vocr_lstm_xproj_pack: null pointer");
    VOCR_CHECK_ARG(h == 512, "vocr_lstm_xproj_pack: H must be 512 (got %d)", h);
    VOCR_CHECK_ARG(aligned16(w_ih_fwd) && aligned16(w_ih_rev) && aligned16(wpack), "vocr_lstm_xproj_pack: pointers must be 16-byte aligned");
    const long pieces = 2l * (8 * h / 32) * (h / 8) * 64;
    lstm_xproj_pack_kernel<<<(unsigned)(pieces / 256), 256, 0, (hipStream_t)stream>>>(w_ih_fwd, w_ih_rev, wpack, h);
    VOCR_CHECK_LAUNCH("vocr_lstm_xproj_pack");
    return VOCR_OK;
}

extern "C" int vocr_lstm_fwd_lead(const float* xproj, const float* xproj2, const float* whh_fwd, const float* whh_rev, const int32_t* lens,
                                  float* y, float* gates, float* cell, void* workspace, int t, int b, int h, int rows,
                                  const float* next_wpack, const float* next_bias, const float* y_mask, float* next_planes,
                                  int32_t* health, void* stream) {
    VOCR_CHECK_ARG(rows >= 0, "vocr_lstm_fwd_lead: rows must be >= 0 (0: dense)");
    if (!next_wpack) {
        VOCR_CHECK_ARG(!next_planes, "vocr_lstm_fwd_lead: next_planes without next_wpack");
        return lstm_fwd_impl(xproj, whh_fwd, whh_rev, lens, y, gates, cell, workspace, t, b, h, 0, t, rows, health, stream, xproj2, nullptr);
    }
    VOCR_CHECK_ARG(next_planes, "vocr_lstm_fwd_lead: next_wpack without next_planes");
    VOCR_CHECK_ARG(vocr_lstm_follow_supported(b, h), "vocr_lstm_fwd_lead: shape not taken (ask vocr_lstm_follow_supported; B=%d H=%d)", b, h);
    VOCR_CHECK_ARG(aligned16(next_wpack) && aligned16(next_planes) && (!y_mask || aligned16(y_mask)), "vocr_lstm_fwd_lead: pointers must be 16-byte aligned");
    VOCR_CHECK_ARG((rows ? (long)rows : (long)t * b) < (1l << 18), "vocr_lstm_fwd_lead: a plane of the next x-projection must stay below 2 GB (rows < 262144)");
    const FollowArgs fa{next_wpack, y_mask, next_bias, next_planes, nullptr, nullptr, 0};
    return lstm_fwd_impl(xproj, whh_fwd, whh_rev, lens, y, gates, cell, workspace, t, b, h, 0, t, rows, health, stream, xproj2, &fa);
}

// to_packed[t*B + b] = packed row of frame (t, b), or -1 where the packed layout has none (t >= the length of b's chain);
// to_dense[r] = t*B + b of packed row r, or -1 for the all-zero groups and a last chain's rows >= B
__global__ void seq_rowmap_kernel(const int32_t* __restrict__ lens, int T, int B, int rows, int32_t* __restrict__ to_packed,
                                  int32_t* __restrict__ to_dense) {
    __shared__ int gbase[17];          // first group of chain c (its t = 0); [nch] = the trailing zero group
    const int nch = (B + 3) / 4;
    if (threadIdx.x == 0) {
        int g = 1;
        for (int c = 0; c < nch; ++c) {
            gbase[c] = g;
            g += min(lens[4 * c], T) + 1;
        }
        gbase[nch] = g - 1;
    }
    __syncthreads();
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (to_packed && i < (long)T * B) {
        const int t = (int)(i / B), b = (int)(i % B), c = b >> 2;
        to_packed[i] = t < min(lens[4 * c], T) ? 4 * (gbase[c] + t) + (b & 3) : -1;
    }
    if (to_dense && i < rows) {
        const int g = (int)(i >> 2), j = (int)(i & 3);
        int v = -1;
        for (int c = 0; c < nch; ++c) {
            const int L = min(lens[4 * c], T);
            if (g >= gbase[c] && g < gbase[c] + L && 4 * c + j < B) v = (g - gbase[c]) * B + 4 * c + j;
        }
        to_dense[i] = v;
    }
}

extern "C" int vocr_seq_rowmap(const int32_t* lens, int t, int b, int rows, int32_t* to_packed, int32_t* to_dense, void* stream) {
    VOCR_CHECK_ARG(lens && (to_packed || to_dense), "vocr_seq_rowmap: null pointer");
    VOCR_CHECK_ARG(t > 0 && b > 0 && b <= 64 && rows > 0 && rows % 4 == 0, "vocr_seq_rowmap: bad shape (T=%d B=%d rows=%d)", t, b, rows);
    const long n = (long)t * b > rows ? (long)t * b : rows;
    seq_rowmap_kernel<<<vocr_cdiv(n, 256), 256, 0, (hipStream_t)stream>>>(lens, t, b, rows, to_packed, to_dense);
    VOCR_CHECK_LAUNCH("vocr_seq_rowmap");
    return VOCR_OK;
}

// dst[r][:] = src[map[r]][:] (map[r] >= 0) or fill[:] / 0 (map[r] < 0): one wave per row, 16-byte pieces when n % 4 == 0
__global__ __launch_bounds__(256) void gather_rows_kernel(const float* __restrict__ src, float* __restrict__ dst, const int32_t* __restrict__ map,
                                                          long nrows, int n, const float* __restrict__ fill, int vec) {
    const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= nrows) return;
    const int lane = threadIdx.x & 63;
    const int m = map[r];
    float* d = dst + r * n;
    if (vec) {
        const f32x4* s4 = m >= 0 ? (const f32x4*)(src + (long)m * n) : nullptr;
        for (int i = lane; i < (n >> 2); i += 64) {
            f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (s4) v = __builtin_nontemporal_load(s4 + i);
            else if (fill) v = ((const f32x4*)fill)[i];
            ((f32x4*)d)[i] = v;
        }
    } else {
        for (int i = lane; i < n; i += 64) d[i] = m >= 0 ? src[(long)m * n + i] : (fill ? fill[i] : 0.f);
    }
}

extern "C" int vocr_gather_rows(const float* src, float* dst, const int32_t* map, long nrows, int n, const float* fill, void* stream) {
    VOCR_CHECK_ARG(src && dst && map, "vocr_gather_rows: null pointer");
    VOCR_CHECK_ARG(nrows > 0 && n > 0, "vocr_gather_rows: bad shape (%ld x %d)", nrows, n);
    // 16-byte pieces when the row length and every pointer allow it (a fill row that is a view into a flat parameter buffer need not be
    // aligned), element by element otherwise
    const int vec = (n & 3) == 0 && ((((uintptr_t)src | (uintptr_t)dst | (uintptr_t)fill) & 15) == 0);
    gather_rows_kernel<<<vocr_cdiv(nrows, 4), 256, 0, (hipStream_t)stream>>>(src, dst, map, nrows, n, fill, vec);
    VOCR_CHECK_LAUNCH("vocr_gather_rows");
    return VOCR_OK;
}

// Gradient of vocr_gather_rows' fill row: dfill[j] = sum over the rows r with map[r] < 0 of dout[r][j], in a fixed order (64 row
// ranges, then their partial sums one after the other)
__global__ __launch_bounds__(256) void gather_fill_grad_partial(const float* __restrict__ dout, const int32_t* __restrict__ map, long nrows, int n,
                                                                float* __restrict__ part) {
    const long chunk = (nrows + gridDim.x - 1) / gridDim.x, r0 = blockIdx.x * chunk, r1 = min(r0 + chunk, nrows);
    for (int j = threadIdx.x; j < n; j += 256) {
        float a = 0.f;
        for (long r = r0; r < r1; ++r)
            if (map[r] < 0) a += dout[r * n + j];
        part[(long)blockIdx.x * n + j] = a;
    }
}
__global__ void gather_fill_grad_final(const float* __restrict__ part, int nparts, int n, float* __restrict__ dfill) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    float a = 0.f;
    for (int p = 0; p < nparts; ++p) a += part[(long)p * n + j];
    dfill[j] = a;
}
extern "C" size_t vocr_gather_rows_fill_grad_workspace_bytes(int n) { return n > 0 ? (size_t)64 * n * sizeof(float) : 0; }
extern "C" int vocr_gather_rows_fill_grad(const float* dout, const int32_t* map, long nrows, int n, float* dfill, void* workspace, void* stream) {
    VOCR_CHECK_ARG(dout && map && dfill && workspace, "vocr_gather_rows_fill_grad: null pointer");
    VOCR_CHECK_ARG(nrows > 0 && n > 0, "vocr_gather_rows_fill_grad: bad shape (%ld x %d)", nrows, n);
    gather_fill_grad_partial<<<64, 256, 0, (hipStream_t)stream>>>(dout, map, nrows, n, (float*)workspace);
    gather_fill_grad_final<<<vocr_cdiv(n, 256), 256, 0, (hipStream_t)stream>>>((const float*)workspace, 64, n, dfill);
    VOCR_CHECK_LAUNCH("vocr_gather_rows_fill_grad");
    return VOCR_OK;
}

extern "C" int vocr_colsum(const float* x, float* out, int m, int n, void* workspace, void* stream);

extern "C" int vocr_lstm_bwd(const float* dy, const float* whht_fwd, const float* whht_rev, const int32_t* lens,
                             const float* gates, const float* cell, float* dgates, void* workspace, int t, int b, int h,
                             int32_t* health, void* stream) {
    return vocr_lstm_bwd_bias(dy, whht_fwd, whht_rev, lens, gates, cell, dgates, nullptr, workspace, t, b, h, health, stream);
}

// every path but the 8-row K-owner kernel: bias gradient = column sums of the finished dgates
static int lstm_bias_by_colsum(const float* dgates, float* dbias, void* workspace, int t, int b, int h, void* stream) {
    if (!dbias) return VOCR_OK;
    // partial rows live in the last 64 x 4H floats of the LSTM workspace (vocr_lstm_workspace_bytes)
    void* cws = (char*)workspace + (size_t)6 * 2 * b * h * sizeof(float) + 4096 + ((size_t)16 << 20) + (size_t)16 * 4 * h * sizeof(float);
    for (int dir = 0; dir < 2; ++dir) {
        const int rc = vocr_colsum(dgates + (size_t)dir * t * b * 4 * h, dbias + (size_t)dir * 4 * h, t * b, 4 * h, cws, stream);
        if (rc != VOCR_OK) return rc;
    }
    return VOCR_OK;
}

// which self-validating backward sweep (if any) a shape gets: 2 = wide members, 1 = 4-row chains, 0 = another path
static int lstm_bwd_selfval_kind(int b, int h) {
    const SweepKind k = lstm_sweep_kind(true, b, h, (h == 128 || h == 256 || h == 512) && 8 * (h / 16) <= resident_workgroup_capacity());
    return k == SWEEP_WIDE4 ? 2 : k == SWEEP_CHAIN4 ? 1 : 0;
}

static int lstm_bwd_impl(const float* dy, const float* dy_mask, const float* whht_fwd, const float* whht_rev, const int32_t* lens,
                         const float* gates, const float* cell, float* dgates, float* dbias, bool combine, void* workspace, int t,
                         int b, int h, int32_t* health, void* stream, int packed_rows = 0);

extern "C" int vocr_lstm_bwd_bias(const float* dy, const float* whht_fwd, const float* whht_rev, const int32_t* lens,
                                  const float* gates, const float* cell, float* dgates, float* dbias, void* workspace, int t,
                                  int b, int h, int32_t* health, void* stream) {
    return lstm_bwd_impl(dy, nullptr, whht_fwd, whht_rev, lens, gates, cell, dgates, dbias, true, workspace, t, b, h, health, stream);
}

extern "C" int vocr_lstm_bwd_parts_supported(int t, int b, int h) {
    return t > 0 && lstm_bwd_selfval_kind(b, h) != 0 ? 1 : 0;
}

extern "C" int vocr_lstm_bwd_parts(const float* dy, const float* dy_mask, const float* whht_fwd, const float* whht_rev, const int32_t* lens,
                                   const float* gates, const float* cell, float* dgates, void* workspace, int t, int b, int h,
                                   int32_t* health, void* stream) {
    VOCR_CHECK_ARG(vocr_lstm_bwd_parts_supported(t, b, h), "vocr_lstm_bwd_parts: no fused path for T=%d B=%d H=%d (ask vocr_lstm_bwd_parts_supported)", t, b, h);
    VOCR_CHECK_ARG(aligned16(whht_fwd) && aligned16(whht_rev) && aligned16(dgates) && aligned16(gates), "vocr_lstm_bwd_parts: 16-byte alignment");
    return lstm_bwd_impl(dy, dy_mask, whht_fwd, whht_rev, lens, gates, cell, dgates, (float*)workspace /* any non-null: parts wanted */, false, workspace, t, b, h,
                         health, stream);
}

extern "C" int vocr_lstm_bias_from_parts(float* dbias, const void* workspace, int t, int b, int h, void* stream) {
    VOCR_CHECK_ARG(dbias && workspace && vocr_lstm_bwd_parts_supported(t, b, h), "vocr_lstm_bias_from_parts: bad argument");
    const float* bpart = (const float*)((const char*)workspace + 4096 + ((size_t)16 << 20));      // [chain][4H], left by vocr_lstm_bwd_parts
    lstm_bias_combine_kernel<<<dim3(vocr_cdiv(4 * h, 256), 2), 256, 0, (hipStream_t)stream>>>(bpart, dbias, 4 * h, (b + 3) / 4);
    VOCR_CHECK_LAUNCH("vocr_lstm_bias_from_parts");
    return VOCR_OK;
}

extern "C" int vocr_lstm_bwd_packed(const float* dy, const float* dy_mask, const float* whht_fwd, const float* whht_rev, const int32_t* lens,
                                    const float* gates, const float* cell, float* dgates, float* dbias, void* workspace, int t, int b,
                                    int h, int rows, int32_t* health, void* stream) {
    VOCR_CHECK_ARG(rows > 0 && rows % 4 == 0 && vocr_lstm_packed_supported(b, h),
                   "vocr_lstm_bwd_packed: the packed row layout needs a 4-row chain sweep (ask vocr_lstm_packed_supported; B=%d H=%d rows=%d)", b, h, rows);
    VOCR_CHECK_ARG(aligned16(whht_fwd) && aligned16(whht_rev) && aligned16(dgates) && aligned16(gates), "vocr_lstm_bwd_packed: 16-byte alignment");
    return lstm_bwd_impl(dy, dy_mask, whht_fwd, whht_rev, lens, gates, cell, dgates, dbias, true, workspace, t, b, h, health, stream, rows);
}

static int lstm_bwd_impl(const float* dy, const float* dy_mask, const float* whht_fwd, const float* whht_rev, const int32_t* lens,
                         const float* gates, const float* cell, float* dgates, float* dbias, bool combine, void* workspace, int t,
                         int b, int h, int32_t* health, void* stream, int packed_rows) {
    const float* whh_fwd = whht_fwd;
    const float* whh_rev = whht_rev;
    VOCR_CHECK_ARG(dy && whh_fwd && whh_rev && lens && gates && cell && dgates && workspace, "vocr_lstm_bwd: null pointer");
    VOCR_CHECK_ARG(t > 0 && b > 0 && b <= 64 && h > 0 && h % 16 == 0, "vocr_lstm_bwd: need 1<=B<=64 and H%%16==0 (B=%d H=%d)", b, h);
    VOCR_CHECK_SWEEP_ENV(true);
    hipStream_t s = (hipStream_t)stream;
    float* dcb = (float*)workspace;
    const int rt = (b + 15) / 16;
    const bool fast = (h == 128 || h == 256 || h == 512) && aligned16(whh_fwd) && aligned16(whh_rev) && aligned16(dgates);
    const SweepKind kind = lstm_sweep_kind(true, b, h, fast && 8 * (h / 16) <= resident_workgroup_capacity() && aligned16(gates));
    VOCR_CHECK_ARG(!packed_rows || kind == SWEEP_WIDE4 || kind == SWEEP_CHAIN4, "vocr_lstm_bwd_packed: no 4-row chain sweep for B=%d H=%d", b, h);
    if (kind != SWEEP_STEP) {
        // arrival flags: [chain <= 8][32 workgroups] at [0..255]; status words at [512..]
        unsigned* flags = (unsigned*)workspace;
        unsigned* status = flags + 512;
        unsigned* hword = (unsigned*)health;
        const dim3 g(8 * (h / 16));
        // K-owner form: partial sums [parity][chain][consumer][producer][16 x 16] behind the flags/status words
        float* partials = (float*)((char*)workspace + 4096);
        const int nt4 = (b + 3) / 4;
        const int fwt = sweep_write_through() ? 1 : 0;
        float* bpart = dbias ? (float*)((char*)workspace + 4096 + ((size_t)16 << 20)) : nullptr;     // [chain][4H]
        if (kind == SWEEP_WIDE4 || kind == SWEEP_CHAIN4) {
            // self-validating hand-off.  One fill: [XCC ids | status word | ... 4 KB][ring of partial blocks = "not written yet"]
            const int nch = 2 * nt4;
            const size_t ring = kind == SWEEP_WIDE4 ? (size_t)4 * nch * 16 * 16 * 512 : (size_t)4 * nch * 32 * 32 * 256;
            if (hipMemsetAsync(flags, 0xFF, 4096 + ring, s) != hipSuccess) {
                vocr_set_error("vocr_lstm_bwd: memset failed");
                return VOCR_ELAUNCH;
            }
            if (kind == SWEEP_WIDE4) {
                lstm_bwd_chain4w<4><<<256, 512, 0, s>>>(dy, whh_fwd, whh_rev, lens, gates, cell, dy_mask, dgates, partials, flags, status, hword, bpart, t, b, nt4, fwt, packed_rows);
            } else {
                const dim3 g4((nch > 8 ? 16 : 8) * (h / 16));
                if (h == 512)
                    lstm_bwd_chain4v<4><<<g4, 256, 0, s>>>(dy, whh_fwd, whh_rev, lens, gates, cell, dy_mask, dgates, partials, flags, status, hword, bpart, t, b, nt4, fwt, packed_rows);
                else
                    lstm_bwd_chain4v<2><<<g4, 256, 0, s>>>(dy, whh_fwd, whh_rev, lens, gates, cell, dy_mask, dgates, partials, flags, status, hword, bpart, t, b, nt4, fwt, packed_rows);
            }
            VOCR_CHECK_LAUNCH("vocr_lstm_bwd(k-owner, 4-row chains, self-validating)");
            if (dbias && combine) {
                lstm_bias_combine_kernel<<<dim3(vocr_cdiv(4 * h, 256), 2), 256, 0, s>>>(bpart, dbias, 4 * h, nt4);
                VOCR_CHECK_LAUNCH("vocr_lstm_bwd(bias combine)");
            }
            return VOCR_OK;
        }
        if (hipMemsetAsync(flags, 0, 520 * sizeof(unsigned), s) != hipSuccess) {
            vocr_set_error("vocr_lstm_bwd: memset failed");
            return VOCR_ELAUNCH;
        }
        if (h == 128) lstm_bwd_kowner<1><<<g, 256, 0, s>>>(dy, whh_fwd, whh_rev, lens, gates, cell, dgates, partials, flags, flags + 256, status, hword, t, b, rt, fwt);
        else if (h == 256) lstm_bwd_kowner<2><<<g, 256, 0, s>>>(dy, whh_fwd, whh_rev, lens, gates, cell, dgates, partials, flags, flags + 256, status, hword, t, b, rt, fwt);
        else lstm_bwd_kowner<4><<<g, 256, 0, s>>>(dy, whh_fwd, whh_rev, lens, gates, cell, dgates, partials, flags, flags + 256, status, hword, t, b, rt, fwt);
        VOCR_CHECK_LAUNCH("vocr_lstm_bwd(k-owner)");
        return lstm_bias_by_colsum(dgates, dbias, workspace, t, b, h, stream);
    }
    for (int step = 0; step < t; ++step) {
        if (fast) {
#ifdef VOCR_LSTM_DIAG
            static const int dbg = getenv("VOCR_LSTM_DEBUG") ? atoi(getenv("VOCR_LSTM_DEBUG")) : 0;
#else
            constexpr int dbg = 0;
#endif
            const bool half = (dbg & 8) != 0;          // 8 units per workgroup (256 WGs at B=32): measured no faster than 16
            const dim3 g(2 * (h / (half ? 8 : 16)) * rt);
#define VOCR_BWD(NCH, UT) lstm_bwd_step_fast<NCH, 4, UT><<<g, 256, 0, s>>>(dy, whh_fwd, whh_rev, lens, gates, cell, dgates, dcb, t, b, step, rt, dbg)
            if (h == 128) { if (half) VOCR_BWD(1, 8); else VOCR_BWD(1, 16); }
            else if (h == 256) { if (half) VOCR_BWD(2, 8); else VOCR_BWD(2, 16); }
            else { if (half) VOCR_BWD(4, 8); else VOCR_BWD(4, 16); }
#undef VOCR_BWD
        } else {
            lstm_bwd_step_kernel<<<2 * (h / 16), 256, 0, s>>>(dy, whh_fwd, whh_rev, lens, gates, cell, dgates, dcb, t, b, h, step);
        }
    }
    VOCR_CHECK_LAUNCH("vocr_lstm_bwd");
    return lstm_bias_by_colsum(dgates, dbias, workspace, t, b, h, stream);
}
