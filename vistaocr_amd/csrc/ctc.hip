// CTC loss + gradient w.r.t. pre-softmax activations (stands in for warpctc_pytorch.CTCLoss, reference call
// sites src/train_cnn_lstm.py:358,138), greedy best-path decode (src/models/cnnlstm.py:479-541), and the
// fused gradient-clamp + Adam update (src/train_cnn_lstm.py:143-149,363).
//
// CTC: blank = 0, extended label l' of length S = 2L+1, log-space alpha/beta recursions in fp32 with the
// max-shifted log-sum-exp (the same formulation as the PyTorch-CPU criterion the parity target uses).
//   kernel 1  row-wise log-softmax            : one wave per (t,b) row
//   kernel 2  alpha and beta sweeps           : one wave per (sample, direction); lanes over s; the previous
//                                               row lives in LDS
//   kernel 3  gradient                        : one wave per (t,b); lanes over the alphabet
#include "vocr_common.h"

namespace {

constexpr float NEG_INF = -INFINITY;

__global__ __launch_bounds__(256) void log_softmax_rows_kernel(const float* __restrict__ x, float* __restrict__ lp,
                                                               int rows, int V) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float* xr = x + (long)row * V;
    float m = NEG_INF;
    for (int v = lane; v < V; v += 64) m = fmaxf(m, xr[v]);
    m = wave_max(m);
    float s = 0.f;
    for (int v = lane; v < V; v += 64) s += expf(xr[v] - m);
    s = wave_sum(s);
    const float lse = m + logf(s);
    float* lr = lp + (long)row * V;
    for (int v = lane; v < V; v += 64) lr[v] = xr[v] - lse;
}

__device__ __forceinline__ float lse3(float a, float b, float c) {
    float m = fmaxf(a, fmaxf(b, c));
    if (m == NEG_INF) return NEG_INF;
    return logf(expf(a - m) + expf(b - m) + expf(c - m)) + m;
}

// grid.x = 2*B (even: alpha of sample b, odd: beta); 64 threads.  ab[b][dir][t][s], row pitch SP.
__global__ __launch_bounds__(64) void ctc_alpha_beta_kernel(const float* __restrict__ lp, const int32_t* __restrict__ labels,
                                                            const int32_t* __restrict__ label_offsets,
                                                            const int32_t* __restrict__ label_lens,
                                                            const int32_t* __restrict__ act_lens, float* __restrict__ ab,
                                                            float* __restrict__ nll, int T, int B, int V, int SP) {
    extern __shared__ float sm[];            // [2][SP+2] rows (with 2 leading -inf pads) + int ext[SP]
    const int b = blockIdx.x >> 1, dirn = blockIdx.x & 1;
    const int lane = threadIdx.x;
    const int L = label_lens[b], S = 2 * L + 1, Tb = act_lens[b];
    const int32_t* lab = labels + label_offsets[b];
    float* row0 = sm;
    float* row1 = sm + (SP + 4);
    int* ext = (int*)(sm + 2 * (SP + 4));
    for (int s = lane; s < SP; s += 64) ext[s] = (s < S && (s & 1)) ? lab[s >> 1] : 0;
    for (int s = lane; s < SP + 4; s += 64) { row0[s] = NEG_INF; row1[s] = NEG_INF; }
    __syncthreads();
    float* out = ab + ((long)(b * 2 + dirn) * T) * SP;
    if (Tb <= 0) {
        if (dirn == 0 && lane == 0) nll[b] = (S == 1) ? 0.f : INFINITY;
        return;
    }
    float* prev = row0;
    float* cur = row1;
    if (dirn == 0) {
        // alpha_0
        const float* l0 = lp + ((long)0 * B + b) * V;
        for (int s = lane; s < SP; s += 64) {
            float v = NEG_INF;
            if (s == 0) v = l0[0];
            else if (s == 1 && S > 1) v = l0[ext[1]];
            prev[2 + s] = v;
            out[s] = v;
        }
        __syncthreads();
        for (int t = 1; t < Tb; ++t) {
            const float* lt = lp + ((long)t * B + b) * V;
            for (int s = lane; s < SP; s += 64) {
                float v = NEG_INF;
                if (s < S) {
                    const int e = ext[s];
                    const float a1 = prev[2 + s], a2 = prev[1 + s];
                    const float a3 = (s >= 2 && e != 0 && e != ext[s - 2]) ? prev[s] : NEG_INF;
                    const float l = lse3(a1, a2, a3);
                    v = (l == NEG_INF) ? NEG_INF : l + lt[e];
                }
                cur[2 + s] = v;
                out[(long)t * SP + s] = v;
            }
            __syncthreads();
            float* tmp = prev; prev = cur; cur = tmp;
        }
        if (lane == 0) {
            const float a = prev[2 + S - 1];
            const float c = S > 1 ? prev[2 + S - 2] : NEG_INF;
            const float m = fmaxf(a, c);
            nll[b] = (m == NEG_INF) ? INFINITY : -(logf(expf(a - m) + expf(c - m)) + m);
        }
    } else {
        // beta_{Tb-1}; stored rows use the same [t][s] indexing; prev[s] holds beta[s], pads sit at S, S+1
        const float* lT = lp + ((long)(Tb - 1) * B + b) * V;
        for (int s = lane; s < SP + 2; s += 64) {
            float v = NEG_INF;
            if (s == S - 1) v = lT[0];
            else if (s == S - 2 && S > 1) v = lT[ext[S - 2]];
            prev[s] = v;
            if (s < SP) out[(long)(Tb - 1) * SP + s] = v;
        }
        __syncthreads();
        for (int t = Tb - 2; t >= 0; --t) {
            const float* lt = lp + ((long)t * B + b) * V;
            for (int s = lane; s < SP + 2; s += 64) {
                float v = NEG_INF;
                if (s < S) {
                    const int e = ext[s];
                    const float b1 = prev[s], b2 = prev[s + 1];
                    const float b3 = (s + 2 < S && e != 0 && e != ext[s + 2]) ? prev[s + 2] : NEG_INF;
                    const float l = lse3(b1, b2, b3);
                    v = (l == NEG_INF) ? NEG_INF : l + lt[e];
                }
                cur[s] = v;
                if (s < SP) out[(long)t * SP + s] = v;
            }
            __syncthreads();
            float* tmp = prev; prev = cur; cur = tmp;
        }
    }
}

// The same sweeps for S = 2L+1 <= 64 (L <= 31: every BASELINE workload): one extended-label position per lane, the previous
// row stays in a register and neighbours are fetched with wave shuffles (no LDS, no barrier), and the one global operand of
// a step, lp[t][b][l'_s], does not depend on the recursion, so it is gathered PF steps ahead.  The generic kernel paid an
// L2/HBM round trip per time step for it (0.73 us x 294 steps = 215 us on the critical path of every training step).
// Same expressions in the same order per element as ctc_alpha_beta_kernel: results are bit-identical.
__global__ __launch_bounds__(64) void ctc_alpha_beta64_kernel(const float* __restrict__ lp, const int32_t* __restrict__ labels,
                                                              const int32_t* __restrict__ label_offsets,
                                                              const int32_t* __restrict__ label_lens,
                                                              const int32_t* __restrict__ act_lens, float* __restrict__ ab,
                                                              float* __restrict__ nll, int T, int B, int V) {
    constexpr int SP = 64, PF = 8;
    const int b = blockIdx.x >> 1, dirn = blockIdx.x & 1;
    const int lane = threadIdx.x;
    const int L = label_lens[b], S = 2 * L + 1, Tb = act_lens[b];
    const int32_t* lab = labels + label_offsets[b];
    float* out = ab + ((long)(b * 2 + dirn) * T) * SP;
    if (Tb <= 0) {
        if (dirn == 0 && lane == 0) nll[b] = (S == 1) ? 0.f : INFINITY;
        return;
    }
    const bool in = lane < S;
    const int e = (in && (lane & 1)) ? lab[lane >> 1] : 0;
    const int e_m2 = __shfl_up(e, 2, 64), e_p2 = __shfl_down(e, 2, 64);
    const float* col = lp + (long)b * V + e;              // lp[t][b][e] = col[t * B * V]
    const long tstride = (long)B * V;
    float buf[PF];
    if (dirn == 0) {
        const bool skip = lane >= 2 && e != 0 && e != e_m2;
        float v = NEG_INF;
        if (lane == 0) v = lp[(long)b * V];
        else if (lane == 1 && S > 1) v = col[0];
        out[lane] = v;
#pragma unroll
        for (int k = 0; k < PF; ++k) buf[k] = col[(long)min(1 + k, Tb - 1) * tstride];
        for (int t0 = 1; t0 < Tb; t0 += PF) {
#pragma unroll
            for (int k = 0; k < PF; ++k) {
                const int t = t0 + k;
                if (t < Tb) {                                   // wave-uniform
                    const float lpe = buf[k];
                    buf[k] = col[(long)min(t + PF, Tb - 1) * tstride];
                    float a2 = __shfl_up(v, 1, 64), a3 = __shfl_up(v, 2, 64);
                    if (lane < 1) a2 = NEG_INF;
                    if (!skip) a3 = NEG_INF;
                    const float l = lse3(v, a2, a3);
                    v = (in && l != NEG_INF) ? l + lpe : NEG_INF;
                    out[(long)t * SP + lane] = v;
                }
            }
        }
        const float a = __shfl(v, S - 1, 64);
        const float c = S > 1 ? __shfl(v, S - 2, 64) : NEG_INF;
        if (lane == 0) {
            const float m = fmaxf(a, c);
            nll[b] = (m == NEG_INF) ? INFINITY : -(logf(expf(a - m) + expf(c - m)) + m);
        }
    } else {
        const bool skip = lane + 2 < S && e != 0 && e != e_p2;
        float v = NEG_INF;
        if (lane == S - 1) v = lp[((long)(Tb - 1) * B + b) * V];
        else if (lane == S - 2 && S > 1) v = col[(long)(Tb - 1) * tstride];
        out[(long)(Tb - 1) * SP + lane] = v;
#pragma unroll
        for (int k = 0; k < PF; ++k) buf[k] = col[(long)max(Tb - 2 - k, 0) * tstride];
        for (int t0 = Tb - 2; t0 >= 0; t0 -= PF) {
#pragma unroll
            for (int k = 0; k < PF; ++k) {
                const int t = t0 - k;
                if (t >= 0) {
                    const float lpe = buf[k];
                    buf[k] = col[(long)max(t - PF, 0) * tstride];
                    float b2 = __shfl_down(v, 1, 64), b3 = __shfl_down(v, 2, 64);
                    if (lane >= 63) b2 = NEG_INF;
                    if (!skip) b3 = NEG_INF;
                    const float l = lse3(v, b2, b3);
                    v = (in && l != NEG_INF) ? l + lpe : NEG_INF;
                    out[(long)t * SP + lane] = v;
                }
            }
        }
    }
}

// The same for 64 < S <= 128 (32 <= L <= 63: the long lines of BASELINE configs[3], ~1200 px with W / 30 labels): TWO extended-label
// positions per lane (s = lane and s = lane + 64), the seam between the halves crossed with two broadcasts per step.  The generic
// kernel took 622 us for T = 576 (1.08 us per frame, chip otherwise idle: the backward waits for it).  Same expressions in the same
// order per element: bit-identical to ctc_alpha_beta_kernel.
__global__ __launch_bounds__(64) void ctc_alpha_beta128_kernel(const float* __restrict__ lp, const int32_t* __restrict__ labels,
                                                               const int32_t* __restrict__ label_offsets,
                                                               const int32_t* __restrict__ label_lens,
                                                               const int32_t* __restrict__ act_lens, float* __restrict__ ab,
                                                               float* __restrict__ nll, int T, int B, int V) {
    constexpr int SP = 128, PF = 8;
    const int b = blockIdx.x >> 1, dirn = blockIdx.x & 1;
    const int lane = threadIdx.x;
    const int L = label_lens[b], S = 2 * L + 1, Tb = act_lens[b];
    const int32_t* lab = labels + label_offsets[b];
    float* out = ab + ((long)(b * 2 + dirn) * T) * SP;
    if (Tb <= 0) {
        if (dirn == 0 && lane == 0) nll[b] = (S == 1) ? 0.f : INFINITY;
        return;
    }
    const int s0 = lane, s1 = lane + 64;
    const bool in0 = s0 < S, in1 = s1 < S;
    auto ext = [&](int s_) { return (s_ >= 0 && s_ < S && (s_ & 1)) ? lab[s_ >> 1] : 0; };
    const int e0 = ext(s0), e1 = ext(s1);
    const float* col0 = lp + (long)b * V + e0;             // lp[t][b][e] = col[t * B * V]
    const float* col1 = lp + (long)b * V + e1;
    const long tstride = (long)B * V;
    float buf0[PF], buf1[PF];
    if (dirn == 0) {
        const bool skip0 = s0 >= 2 && e0 != 0 && e0 != ext(s0 - 2);
        const bool skip1 = e1 != 0 && e1 != ext(s1 - 2);
        float v0 = NEG_INF, v1 = NEG_INF;
        if (lane == 0) v0 = lp[(long)b * V];
        else if (lane == 1 && S > 1) v0 = col0[0];
        out[s0] = v0;
        out[s1] = v1;
#pragma unroll
        for (int k = 0; k < PF; ++k) {
            buf0[k] = col0[(long)min(1 + k, Tb - 1) * tstride];
            buf1[k] = col1[(long)min(1 + k, Tb - 1) * tstride];
        }
        for (int t0 = 1; t0 < Tb; t0 += PF) {
#pragma unroll
            for (int k = 0; k < PF; ++k) {
                const int t = t0 + k;
                if (t < Tb) {                                   // wave-uniform
                    const float lpe0 = buf0[k], lpe1 = buf1[k];
                    buf0[k] = col0[(long)min(t + PF, Tb - 1) * tstride];
                    buf1[k] = col1[(long)min(t + PF, Tb - 1) * tstride];
                    const float w63 = __shfl(v0, 63, 64), w62 = __shfl(v0, 62, 64);
                    float a2 = __shfl_up(v0, 1, 64), a3 = __shfl_up(v0, 2, 64);
                    float c2 = __shfl_up(v1, 1, 64), c3 = __shfl_up(v1, 2, 64);
                    if (lane < 1) { a2 = NEG_INF; c2 = w63; }
                    if (lane == 0) c3 = w62;
                    if (lane == 1) c3 = w63;
                    if (!skip0) a3 = NEG_INF;
                    if (!skip1) c3 = NEG_INF;
                    const float l0 = lse3(v0, a2, a3), l1 = lse3(v1, c2, c3);
                    v0 = (in0 && l0 != NEG_INF) ? l0 + lpe0 : NEG_INF;
                    v1 = (in1 && l1 != NEG_INF) ? l1 + lpe1 : NEG_INF;
                    out[(long)t * SP + s0] = v0;
                    out[(long)t * SP + s1] = v1;
                }
            }
        }
        // alpha[S-1], alpha[S-2] (a short line of a batch with long ones keeps both in the first half)
        const float a = S - 1 >= 64 ? __shfl(v1, S - 1 - 64, 64) : __shfl(v0, S - 1, 64);
        const float c = S < 2 ? NEG_INF : S - 2 >= 64 ? __shfl(v1, S - 2 - 64, 64) : __shfl(v0, S - 2, 64);
        if (lane == 0) {
            const float m = fmaxf(a, c);
            nll[b] = (m == NEG_INF) ? INFINITY : -(logf(expf(a - m) + expf(c - m)) + m);
        }
    } else {
        const bool skip0 = s0 + 2 < S && e0 != 0 && e0 != ext(s0 + 2);
        const bool skip1 = s1 + 2 < S && e1 != 0 && e1 != ext(s1 + 2);
        float v0 = NEG_INF, v1 = NEG_INF;
        {
            const float last = lp[((long)(Tb - 1) * B + b) * V];
            if (s0 == S - 1) v0 = last;
            else if (s0 == S - 2 && S > 1) v0 = col0[(long)(Tb - 1) * tstride];
            if (s1 == S - 1) v1 = last;
            else if (s1 == S - 2 && S > 1) v1 = col1[(long)(Tb - 1) * tstride];
        }
        out[(long)(Tb - 1) * SP + s0] = v0;
        out[(long)(Tb - 1) * SP + s1] = v1;
#pragma unroll
        for (int k = 0; k < PF; ++k) {
            buf0[k] = col0[(long)max(Tb - 2 - k, 0) * tstride];
            buf1[k] = col1[(long)max(Tb - 2 - k, 0) * tstride];
        }
        for (int t0 = Tb - 2; t0 >= 0; t0 -= PF) {
#pragma unroll
            for (int k = 0; k < PF; ++k) {
                const int t = t0 - k;
                if (t >= 0) {
                    const float lpe0 = buf0[k], lpe1 = buf1[k];
                    buf0[k] = col0[(long)max(t - PF, 0) * tstride];
                    buf1[k] = col1[(long)max(t - PF, 0) * tstride];
                    const float u0 = __shfl(v1, 0, 64), u1 = __shfl(v1, 1, 64);
                    float b2 = __shfl_down(v0, 1, 64), b3 = __shfl_down(v0, 2, 64);
                    float d2 = __shfl_down(v1, 1, 64), d3 = __shfl_down(v1, 2, 64);
                    if (lane == 63) { b2 = u0; b3 = u1; d2 = NEG_INF; }
                    if (lane == 62) b3 = u0;
                    if (!skip0) b3 = NEG_INF;
                    if (!skip1) d3 = NEG_INF;
                    const float l0 = lse3(v0, b2, b3), l1 = lse3(v1, d2, d3);
                    v0 = (in0 && l0 != NEG_INF) ? l0 + lpe0 : NEG_INF;
                    v1 = (in1 && l1 != NEG_INF) ? l1 + lpe1 : NEG_INF;
                    out[(long)t * SP + s0] = v0;
                    out[(long)t * SP + s1] = v1;
                }
            }
        }
    }
}

// one wave per (t,b): grad[v] = exp(lp[v]) - exp(lse_{s: l'_s = v}(alpha+beta) + nll - lp[v]); zero for t >= act_len
__global__ __launch_bounds__(256) void ctc_grad_kernel(const float* __restrict__ lp, const int32_t* __restrict__ labels,
                                                       const int32_t* __restrict__ label_offsets,
                                                       const int32_t* __restrict__ label_lens,
                                                       const int32_t* __restrict__ act_lens, const float* __restrict__ ab,
                                                       const float* __restrict__ nll, float* __restrict__ grad, int T, int B,
                                                       int V, int SP) {
    extern __shared__ float sm[];           // per wave: V floats (log-sum accumulators)
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + wave;
    float* accv = sm + wave * V;
    const bool valid = row < T * B;
    const int t = valid ? row / B : 0, b = valid ? row % B : 0;
    const int Tb = act_lens[b];
    const bool act = valid && t < Tb;
    float* g = grad + (long)row * V;
    const int L = label_lens[b], S = 2 * L + 1;
    const int32_t* lab = labels + label_offsets[b];
    const float* al = ab + ((long)(b * 2 + 0) * T + t) * SP;
    const float* be = ab + ((long)(b * 2 + 1) * T + t) * SP;
    const float* l = lp + (long)(valid ? row : 0) * V;
    if (act) {
        for (int v = lane; v < V; v += 64) accv[v] = NEG_INF;
    }
    __syncthreads();
    if (act) {
        // blanks (even s) all map to v = 0: reduce them across lanes
        float mb = NEG_INF;
        for (int s = 2 * lane; s < S; s += 128) mb = fmaxf(mb, al[s] + be[s]);
        mb = wave_max(mb);
        float sb = 0.f;
        if (mb != NEG_INF)
            for (int s = 2 * lane; s < S; s += 128) sb += expf(al[s] + be[s] - mb);
        sb = wave_sum(sb);
        // labels (odd s): lane 0 walks them in order; L is small (tens), duplicates stay exact and ordered
        if (lane == 0) {
            accv[0] = (mb == NEG_INF) ? NEG_INF : mb + logf(sb);
            for (int i = 0; i < L; ++i) {
                const int s = 2 * i + 1, v = lab[i];
                const float x = al[s] + be[s];
                const float cur = accv[v];
                const float m = fmaxf(cur, x);
                accv[v] = (m == NEG_INF) ? NEG_INF : m + logf(expf(cur - m) + expf(x - m));
            }
        }
    }
    __syncthreads();
    if (act) {
        const float nl = nll[b];
        for (int v = lane; v < V; v += 64) {
            const float lpv = l[v];
            const float a = accv[v];
            const float occ = (a == NEG_INF) ? 0.f : expf(a + nl - lpv);
            g[v] = expf(lpv) - occ;
        }
    } else if (valid) {
        for (int v = lane; v < V; v += 64) g[v] = 0.f;
    }
}

__global__ __launch_bounds__(256) void argmax_rows_kernel(const float* __restrict__ x, int32_t* __restrict__ idx,
                                                          float* __restrict__ maxv, int rows, int V) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float* xr = x + (long)row * V;
    float m = NEG_INF;
    int mi = 0x7fffffff;
    for (int v = lane; v < V; v += 64) {
        const float f = xr[v];
        if (mi == 0x7fffffff || f > m) { m = f; mi = v; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float om = __shfl_xor(m, o, 64);
        const int oi = __shfl_xor(mi, o, 64);
        if (om > m || (om == m && oi < mi)) { m = om; mi = oi; }   // first maximum wins, like numpy.argmax
    }
    if (lane == 0) { idx[row] = mi; maxv[row] = m; }
}

// one thread per sample (T is a few hundred): blank / low-activation / repeat collapse of decode_without_lm
__global__ void greedy_collapse_kernel(const int32_t* __restrict__ idx, const float* __restrict__ maxv,
                                       const int32_t* __restrict__ lens, const int32_t* __restrict__ canon,
                                       int32_t* __restrict__ out_labels, int32_t* __restrict__ out_counts, int T, int B,
                                       float thresh) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const int len = min(lens[b], T);
    int prev = -1, n = 0;
    for (int t = 0; t < len; ++t) {
        const int k = idx[(long)t * B + b];
        if (k == 0) { prev = -1; continue; }
        if (maxv[(long)t * B + b] < thresh) { prev = -1; continue; }
        const int ck = canon[k];
        if (ck == prev) continue;
        out_labels[(long)b * T + n++] = k;
        prev = ck;
    }
    out_counts[b] = n;
}

// NaN policy: grad.clamp_(-5, 5) of the reference propagates NaN (torch.clamp), so a NaN gradient must reach the
// weights and the next loss instead of being turned into a finite +-clamp update by fminf/fmaxf; the first NaN seen is
// also recorded in the caller's health word (health[1]) so the host can fail the step without an extra sync.
__device__ __forceinline__ float clamp_keep_nan(float x, float c) { return (x != x) ? x : fminf(fmaxf(x, -c), c); }

__global__ __launch_bounds__(256) void clamp_adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                         float* __restrict__ m, float* __restrict__ v, size_t n, float lr,
                                                         float omb1, float beta2, float omb2, float eps, float wd, float clampv,
                                                         float gscale, float step_size, float bc2_sqrt, int32_t* health) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    bool bad = false;
    for (; i < n; i += stride) {
        float gr = g[i] * gscale;
        bad |= (gr != gr);
        gr = clamp_keep_nan(gr, clampv);
        const float pv = p[i];
        if (wd != 0.f) gr = gr + wd * pv;
        const float mo = m[i];
        const float mi = mo + omb1 * (gr - mo);                     // exp_avg.lerp_(grad, 1-beta1)
        const float vi = v[i] * beta2 + omb2 * (gr * gr);           // exp_avg_sq.mul_(beta2).addcmul_(g, g, 1-beta2)
        m[i] = mi;
        v[i] = vi;
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        p[i] = pv - step_size * (mi / denom);
    }
    if (health && bad) health[1] = 1;
}

// in-place elementwise clamp (the reference's `param.grad.data.clamp_(min=-5, max=5)` loop) for optimisers that are not
// FlatClampAdam; NaN stays NaN
__global__ void clamp_kernel(float* __restrict__ x, size_t n, float c, int vec, int32_t* health) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t n4 = vec ? n / 4 : 0;
    bool bad = false;
    for (size_t q = i; q < n4; q += stride) {
        f32x4 t = ((f32x4*)x)[q];
#pragma unroll
        for (int k = 0; k < 4; ++k) { bad |= (t[k] != t[k]); t[k] = clamp_keep_nan(t[k], c); }
        ((f32x4*)x)[q] = t;
    }
    for (size_t e = n4 * 4 + i; e < n; e += stride) { bad |= (x[e] != x[e]); x[e] = clamp_keep_nan(x[e], c); }
    if (health && bad) health[1] = 1;
}

}  // namespace

static inline int sp_for(int max_label_len) { return ((2 * max_label_len + 1 + 63) / 64) * 64; }

extern "C" size_t vocr_ctc_workspace_bytes(int t, int b, int v, int max_label_len) {
    if (t <= 0 || b <= 0 || v <= 0 || max_label_len < 0) return 0;
    const size_t sp = sp_for(max_label_len);
    return ((size_t)t * b * v + (size_t)b * 2 * t * sp) * sizeof(float);
}

extern "C" int vocr_ctc_loss_grad(const float* logits, const int32_t* labels, const int32_t* label_offsets,
                                  const int32_t* label_lens, const int32_t* act_lens, float* nll, float* dlogits,
                                  void* workspace, int t, int b, int v, int max_label_len, void* stream) {
    VOCR_CHECK_ARG(logits && labels && label_offsets && label_lens && act_lens && nll && workspace, "vocr_ctc_loss_grad: null pointer");
    VOCR_CHECK_ARG(t > 0 && b > 0 && v > 1 && max_label_len >= 0, "vocr_ctc_loss_grad: bad shape");
    const int sp = sp_for(max_label_len);
    VOCR_CHECK_ARG((size_t)(2 * (sp + 4) + sp) * 4 <= 64 * 1024 && (size_t)4 * v * 4 <= 64 * 1024,
                   "vocr_ctc_loss_grad: label length %d or alphabet %d too large", max_label_len, v);
    hipStream_t s = (hipStream_t)stream;
    float* lp = (float*)workspace;
    float* ab = lp + (size_t)t * b * v;
    const int rows = t * b;
    log_softmax_rows_kernel<<<vocr_cdiv(rows, 4), 256, 0, s>>>(logits, lp, rows, v);
    VOCR_CHECK_LAUNCH("vocr_ctc_loss_grad(log_softmax)");
    const size_t smem = (size_t)(2 * (sp + 4) + sp) * sizeof(float);
    static const int generic_only = VOCR_EXPERIMENT_INT("VOCR_CTC_GENERIC", 0);      // tests: compare the two kernels
    if (sp == 64 && !generic_only) ctc_alpha_beta64_kernel<<<2 * b, 64, 0, s>>>(lp, labels, label_offsets, label_lens, act_lens, ab, nll, t, b, v);
    else if (sp == 128 && !generic_only) ctc_alpha_beta128_kernel<<<2 * b, 64, 0, s>>>(lp, labels, label_offsets, label_lens, act_lens, ab, nll, t, b, v);
    else ctc_alpha_beta_kernel<<<2 * b, 64, smem, s>>>(lp, labels, label_offsets, label_lens, act_lens, ab, nll, t, b, v, sp);
    VOCR_CHECK_LAUNCH("vocr_ctc_loss_grad(alpha_beta)");
    if (dlogits) {
        ctc_grad_kernel<<<vocr_cdiv(rows, 4), 256, (size_t)4 * v * sizeof(float), s>>>(lp, labels, label_offsets, label_lens,
                                                                                      act_lens, ab, nll, dlogits, t, b, v, sp);
        VOCR_CHECK_LAUNCH("vocr_ctc_loss_grad(grad)");
    }
    return VOCR_OK;
}

extern "C" int vocr_argmax_rows(const float* x, int32_t* idx, float* maxv, int rows, int v, void* stream) {
    VOCR_CHECK_ARG(x && idx && maxv && rows > 0 && v > 0, "vocr_argmax_rows: bad argument");
    argmax_rows_kernel<<<vocr_cdiv(rows, 4), 256, 0, (hipStream_t)stream>>>(x, idx, maxv, rows, v);
    VOCR_CHECK_LAUNCH("vocr_argmax_rows");
    return VOCR_OK;
}

extern "C" int vocr_greedy_collapse(const int32_t* idx, const float* maxv, const int32_t* lens, const int32_t* canon,
                                    int32_t* out_labels, int32_t* out_counts, int t, int b, float thresh, void* stream) {
    VOCR_CHECK_ARG(idx && maxv && lens && canon && out_labels && out_counts && t > 0 && b > 0, "vocr_greedy_collapse: bad argument");
    greedy_collapse_kernel<<<vocr_cdiv(b, 64), 64, 0, (hipStream_t)stream>>>(idx, maxv, lens, canon, out_labels, out_counts, t, b, thresh);
    VOCR_CHECK_LAUNCH("vocr_greedy_collapse");
    return VOCR_OK;
}

extern "C" int vocr_clamp_adam(float* p, const float* g, float* m, float* v, size_t count, float lr, float beta1,
                               float beta2, float eps, float weight_decay, float clamp, float grad_scale, int step,
                               int32_t* health, void* stream) {
    VOCR_CHECK_ARG(p && g && m && v && step >= 1, "vocr_clamp_adam: bad argument");
    if (count == 0) return VOCR_OK;
    // bias corrections in double on the host, exactly the scalars torch.optim.Adam derives per step
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    const float step_size = (float)((double)lr / bc1);
    const float bc2_sqrt = (float)sqrt(bc2);
    const float omb1 = (float)(1.0 - (double)beta1), omb2 = (float)(1.0 - (double)beta2);
    size_t gsz = (count + 255) / 256;
    if (gsz > 4096) gsz = 4096;
    clamp_adam_kernel<<<(int)gsz, 256, 0, (hipStream_t)stream>>>(p, g, m, v, count, lr, omb1, beta2, omb2, eps, weight_decay, clamp,
                                                                grad_scale, step_size, bc2_sqrt, health);
    VOCR_CHECK_LAUNCH("vocr_clamp_adam");
    return VOCR_OK;
}

extern "C" int vocr_clamp(float* x, size_t count, float clamp, int32_t* health, void* stream) {
    VOCR_CHECK_ARG(x && clamp >= 0.f, "vocr_clamp: bad argument");
    if (count == 0) return VOCR_OK;
    size_t gsz = (count / 4 + 255) / 256;
    if (gsz > 2048) gsz = 2048;
    if (gsz < 1) gsz = 1;
    clamp_kernel<<<(int)gsz, 256, 0, (hipStream_t)stream>>>(x, count, clamp, ((((uintptr_t)x) & 15) == 0) ? 1 : 0, health);
    VOCR_CHECK_LAUNCH("vocr_clamp");
    return VOCR_OK;
}
