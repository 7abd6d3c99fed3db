// BatchNorm2d(+ReLU) training/eval statistics, apply and backward; FractionalMaxPool2d 2x2; ReLU+MaxPool2d(2,2).
// All HBM-bound streaming kernels over NCHW planes (reference src/models/cnnlstm.py:119-120,127,130,265-266).
// Reductions accumulate in double and combine in a fixed order (bitwise reproducible).
#include "vocr_common.h"

namespace {

constexpr int BN_CHUNK_ELEMS = 16384;   // elements of one channel handled by one workgroup of the stats pass

__host__ __device__ inline int bn_nchunk(long per_channel) {
    long n = (per_channel + BN_CHUNK_ELEMS - 1) / BN_CHUNK_ELEMS;
    return (int)(n < 1 ? 1 : (n > 512 ? 512 : n));
}

// element e of channel c in [0, N*HW): plane n = e / HW, offset e % HW
__device__ __forceinline__ long chan_addr(long e, int c, int C, long HW) {
    const long n = e / HW, r = e - n * HW;
    return (n * C + c) * HW + r;
}

// Walks the elements of one channel in steps of `step` without a 64-bit division per access (chan_addr's e / HW cost
// more than the loads it addressed: the partial-reduction passes ran at 2.0-2.5 TB/s): one division at the start, then
// the (plane, offset) pair is advanced incrementally.
struct ChanWalk {
    long n;
    long r;
    __device__ __forceinline__ ChanWalk(long e, long HW) : n(e / HW), r(e - (e / HW) * HW) {}
    __device__ __forceinline__ long addr(int c, int C, long HW) const { return (n * C + c) * HW + r; }
    __device__ __forceinline__ void advance(long step, long HW) {
        r += step;
        while (r >= HW) { r -= HW; ++n; }
    }
};

__device__ __forceinline__ void block_reduce2(double& a, double& b, double* red) {
    a = wave_sum_d(a);
    b = wave_sum_d(b);
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { red[wave * 2] = a; red[wave * 2 + 1] = b; }
    __syncthreads();
    a = (red[0] + red[2]) + (red[4] + red[6]);
    b = (red[1] + red[3]) + (red[5] + red[7]);
}

// V = elements per load (4, 2 or 1): chosen by the host so that HW % V == 0 and the chunk size is a multiple of V, hence a
// vector never straddles two (n, c) planes and every plane base is V*4-byte aligned.
template <int V> struct VecT { typedef float type __attribute__((ext_vector_type(V))); };
template <> struct VecT<1> { typedef float type; };
template <int V> __device__ __forceinline__ float vget(const typename VecT<V>::type& v, int i) { return v[i]; }
template <> __device__ __forceinline__ float vget<1>(const float& v, int) { return v; }
template <int V> __device__ __forceinline__ void vset(typename VecT<V>::type& v, int i, float x) { v[i] = x; }
template <> __device__ __forceinline__ void vset<1>(float& v, int, float x) { v = x; }

template <int V>
__global__ __launch_bounds__(256) void bn_stats_partial_kernel(const float* __restrict__ y, double* __restrict__ part,
                                                               int N, int C, long HW, int nchunk, long per) {
    typedef typename VecT<V>::type vec;
    __shared__ double red[8];
    const int c = blockIdx.x, j = blockIdx.y;
    const long total = (long)N * HW;
    const long beg = j * per, end = min(total, beg + per);
    double s = 0.0, q = 0.0;
    ChanWalk wk(beg + (long)threadIdx.x * V, HW);
    for (long e = beg + (long)threadIdx.x * V; e < end; e += 256 * V, wk.advance(256 * V, HW)) {
        const vec v = *(const vec*)(y + wk.addr(c, C, HW));
#pragma unroll
        for (int i = 0; i < V; ++i) {
            const double x = (double)vget<V>(v, i);
            s += x;
            q += x * x;
        }
    }
    block_reduce2(s, q, red);
    if (threadIdx.x == 0) {
        part[((long)c * nchunk + j) * 2] = s;
        part[((long)c * nchunk + j) * 2 + 1] = q;
    }
}

__global__ void bn_stats_final_kernel(const double* __restrict__ part, int C, int nchunk, long count, float eps,
                                      float momentum, float* __restrict__ mean, float* __restrict__ invstd,
                                      float* __restrict__ rmean, float* __restrict__ rvar, long long* __restrict__ nbt,
                                      float* __restrict__ xhat_sum) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c == 0 && nbt) *nbt += 1;            // BatchNorm2d.num_batches_tracked += 1 (was a separate torch kernel per layer)
    if (c >= C) return;
    double s = 0.0, q = 0.0;
    for (int j = 0; j < nchunk; ++j) {
        s += part[((long)c * nchunk + j) * 2];
        q += part[((long)c * nchunk + j) * 2 + 1];
    }
    const double m = s / (double)count;
    double var = q / (double)count - m * m;
    if (var < 0.0) var = 0.0;
    mean[c] = (float)m;
    invstd[c] = (float)(1.0 / sqrt(var + (double)eps));
    // sum over the batch of xhat = (y - mean)*invstd with the fp32 mean the later passes use: the rounding residue of the
    // mean, ~0.  The backward needs it for the gradient of the conv bias in front (see bn_bwd_final_kernel).
    if (xhat_sum) xhat_sum[c] = (float)((s - (double)count * (double)mean[c]) * (double)invstd[c]);
    if (rmean) rmean[c] = (1.f - momentum) * rmean[c] + momentum * (float)m;
    if (rvar) {
        const double unbiased = count > 1 ? var * (double)count / (double)(count - 1) : var;
        rvar[c] = (1.f - momentum) * rvar[c] + momentum * (float)unbiased;
    }
}

__global__ void bn_eval_stats_kernel(const float* __restrict__ rmean, const float* __restrict__ rvar, int C, float eps,
                                     float* __restrict__ mean, float* __restrict__ invstd) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    mean[c] = rmean[c];
    invstd[c] = 1.0f / sqrtf(rvar[c] + eps);
}

// ---- the "final" step of a two-pass reduction, done by every workgroup of the pass that consumes it (round 4).  The partial-sum pass
// leaves part[c][nchunk][2] (double); bn_stats_final_kernel / bn_bwd_final_kernel used to add a channel's chunks in chunk order in a
// launch of their own - 7 forward + 7 backward launches of ~6 us with a kernel boundary on either side, all on the step's critical
// path.  Here the consuming workgroup stages its channel's chunks in LDS (one load per thread) and thread 0 adds them in the SAME
// order, so every workgroup of a channel obtains bit-identical sums; the workgroup (first chunk, first image) also stores the
// per-channel results the later passes and the optimiser read.
constexpr int BN_MAX_CHUNKS = 512;
__device__ __forceinline__ void combine_partials(const double* __restrict__ part, int c, int nchunk, double* stage /* [2 * BN_MAX_CHUNKS + 2] */,
                                                 double& s, double& q) {
    for (int j = threadIdx.x; j < nchunk; j += blockDim.x) {
        stage[2 * j] = part[((long)c * nchunk + j) * 2];
        stage[2 * j + 1] = part[((long)c * nchunk + j) * 2 + 1];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double a = 0.0, b = 0.0;
        for (int j = 0; j < nchunk; ++j) {
            a += stage[2 * j];
            b += stage[2 * j + 1];
        }
        stage[2 * BN_MAX_CHUNKS] = a;
        stage[2 * BN_MAX_CHUNKS + 1] = b;
    }
    __syncthreads();
    s = stage[2 * BN_MAX_CHUNKS];
    q = stage[2 * BN_MAX_CHUNKS + 1];
}

// mean / invstd of a channel from its combined sums, exactly as bn_stats_final_kernel computes them; `writer`: also store them and
// update the running statistics
__device__ __forceinline__ void bn_train_stats_of(double s, double q, long count, float eps, float momentum, int c, bool writer,
                                                  float* __restrict__ mean, float* __restrict__ invstd, float* __restrict__ rmean,
                                                  float* __restrict__ rvar, long long* __restrict__ nbt, float* __restrict__ xhat_sum,
                                                  float& mu, float& is) {
    const double m = s / (double)count;
    double var = q / (double)count - m * m;
    if (var < 0.0) var = 0.0;
    mu = (float)m;
    is = (float)(1.0 / sqrt(var + (double)eps));
    if (writer && threadIdx.x == 0) {
        if (c == 0 && nbt) *nbt += 1;
        mean[c] = mu;
        invstd[c] = is;
        if (xhat_sum) xhat_sum[c] = (float)((s - (double)count * (double)mu) * (double)is);
        if (rmean) rmean[c] = (1.f - momentum) * rmean[c] + momentum * (float)m;
        if (rvar) {
            const double unbiased = count > 1 ? var * (double)count / (double)(count - 1) : var;
            rvar[c] = (1.f - momentum) * rvar[c] + momentum * (float)unbiased;
        }
    }
}

struct BnFusedStats {          // non-null part: take mean / invstd from the partial sums instead of the mean / invstd arrays
    const double* part;
    int nchunk;
    long count;
    float eps, momentum;
    float *mean, *invstd, *rmean, *rvar, *xhat_sum;
    long long* nbt;
};

// grid: (chunks over HW, N*C planes)
template <int V>
__global__ __launch_bounds__(256) void bn_relu_apply_kernel(const float* __restrict__ y, const float* __restrict__ mean,
                                                            const float* __restrict__ invstd,
                                                            const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, float* __restrict__ out,
                                                            int C, long HW, BnFusedStats fs) {
    typedef typename VecT<V>::type vec;
    __shared__ double stage[2 * BN_MAX_CHUNKS + 2];
    const long plane = blockIdx.y;
    const int c = (int)(plane % C);
    float mu, is;
    if (fs.part) {
        double s_, q_;
        combine_partials(fs.part, c, fs.nchunk, stage, s_, q_);
        bn_train_stats_of(s_, q_, fs.count, fs.eps, fs.momentum, c, blockIdx.x == 0 && plane < C, fs.mean, fs.invstd, fs.rmean, fs.rvar, fs.nbt,
                          fs.xhat_sum, mu, is);
    } else {
        mu = mean[c];
        is = invstd[c];
    }
    const float g = gamma[c], b = beta[c];
    const vec* yp = (const vec*)(y + plane * HW);
    vec* op = (vec*)(out + plane * HW);
    const long nv = HW / V;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nv; i += (long)gridDim.x * 256) {
        const vec in = yp[i];
        vec o;
#pragma unroll
        for (int k = 0; k < V; ++k) {
            const float v = (vget<V>(in, k) - mu) * is * g + b;
            vset<V>(o, k, v > 0.f ? v : 0.f);
        }
        op[i] = o;
    }
}

template <int V>
__global__ __launch_bounds__(256) void bn_bwd_partial_kernel(const float* __restrict__ da, const float* __restrict__ y,
                                                             const float* __restrict__ mean,
                                                             const float* __restrict__ invstd,
                                                             const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, double* __restrict__ part,
                                                             int N, int C, long HW, int nchunk, long per) {
    typedef typename VecT<V>::type vec;
    __shared__ double red[8];
    const int c = blockIdx.x, j = blockIdx.y;
    const float mu = mean[c], is = invstd[c], g = gamma[c], b = beta[c];
    const long total = (long)N * HW;
    const long beg = j * per, end = min(total, beg + per);
    double s1 = 0.0, s2 = 0.0;
    // This pass sits on the backward's critical path and usually shares the chip with a weight-gradient kernel on the side stream,
    // whose workgroups leave room for ONE of ours per CU (56 registers per lane): with one load per tensor in flight it then ran at a
    // fifth of its stand-alone rate.  Four iterations' loads are issued before the first is consumed (same summation order), and the
    // waves ask for issue priority over the co-resident MFMA waves.
    __builtin_amdgcn_s_setprio(3);
    constexpr int U = 4;
    ChanWalk wk(beg + (long)threadIdx.x * V, HW);
    for (long e = beg + (long)threadIdx.x * V; e < end; e += (long)U * 256 * V) {
        vec yv[U], dv[U];
        bool ok[U];
        const long a0 = wk.addr(c, C, HW);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            ok[u] = e + (long)u * 256 * V < end;
            const long a = ok[u] ? wk.addr(c, C, HW) : a0;
            yv[u] = *(const vec*)(y + a);
            dv[u] = *(const vec*)(da + a);
            wk.advance(256 * V, HW);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            // the V elements of one load are summed in fp32 first (<= 4 terms: exact to ~1 ulp), then carried in double:
            // a quarter of the fp64 conversions/adds of the per-element form
            float p1 = 0.f, p2 = 0.f;
#pragma unroll
            for (int k = 0; k < V; ++k) {
                const float xh = (vget<V>(yv[u], k) - mu) * is;
                const float o = xh * g + b;
                const float dz = o > 0.f ? vget<V>(dv[u], k) : 0.f;
                p1 += dz;
                p2 += dz * xh;
            }
            if (ok[u]) {
                s1 += (double)p1;
                s2 += (double)p2;
            }
        }
    }
    block_reduce2(s1, s2, red);
    if (threadIdx.x == 0) {
        part[((long)c * nchunk + j) * 2] = s1;
        part[((long)c * nchunk + j) * 2 + 1] = s2;
    }
}

// The same two sums for a layer whose activation went straight into FractionalMaxPool2d (bn_relu_fracpool_fwd_kernel):
// da is the scatter of the pooled gradient, so sum_p dz(p) f(p) = sum_o dout(o) [relu active at the winner] f(winner(o)):
// one pass over the POOLED tensors (35 % of the plane) with a gather of y, instead of materialising da and reading it back.
__global__ __launch_bounds__(256) void bn_pool_bwd_partial_kernel(const float* __restrict__ dout, const int32_t* __restrict__ idx,
                                                                  const float* __restrict__ y, const float* __restrict__ mean,
                                                                  const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                                  const float* __restrict__ beta, double* __restrict__ part, int N,
                                                                  int C, long HW, long OP, int nchunk, long per) {
    __shared__ double red[8];
    const int c = blockIdx.x, j = blockIdx.y;
    const float mu = mean[c], is = invstd[c], g = gamma[c], b = beta[c];
    const long total = (long)N * OP;
    const long beg = j * per, end = min(total, beg + per);
    double s1 = 0.0, s2 = 0.0;
    // four strided outputs per thread and iteration: their (dout, idx) loads and then the four dependent gathers of y are
    // in flight together (one output at a time was a chain of two memory latencies per element)
    long wn[4], wr[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const long e0 = beg + threadIdx.x + 256 * u;
        wn[u] = e0 / OP;
        wr[u] = e0 - wn[u] * OP;
    }
    for (long e = beg + threadIdx.x; e < end; e += 1024) {
        float d[4], yv[4];
        int ix[4];
        bool ok[4];
        long pl[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            ok[u] = e + 256 * u < end;
            pl[u] = (ok[u] ? wn[u] : 0) * C + c;
            const long r = ok[u] ? wr[u] : 0;
            d[u] = dout[pl[u] * OP + r];
            ix[u] = idx[pl[u] * OP + r];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) yv[u] = y[pl[u] * HW + ix[u]];
        float p1 = 0.f, p2 = 0.f;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float xh = (yv[u] - mu) * is;
            const float o = xh * g + b;
            const float dz = (ok[u] && o > 0.f) ? d[u] : 0.f;
            p1 += dz;
            p2 += dz * xh;
        }
        s1 += (double)p1;
        s2 += (double)p2;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            wr[u] += 1024;
            while (wr[u] >= OP) { wr[u] -= OP; ++wn[u]; }
        }
    }
    block_reduce2(s1, s2, red);
    if (threadIdx.x == 0) {
        part[((long)c * nchunk + j) * 2] = s1;
        part[((long)c * nchunk + j) * 2 + 1] = s2;
    }
}

// Also the gradient of the conv bias in front of the BatchNorm: sum over (n,h,w) of dy = g*is*(dz - mean(dz) - xh*mean(dz*xh))
// = -g*is*mean(dz*xh)*sum(xh); sum(xh) (xhat_sum, from the forward statistics pass) is the fp32 rounding residue of the batch
// mean, so this is rounding noise around zero exactly as in the reference, but reproducible (it used to be float atomics over
// the dy planes).
__global__ void bn_bwd_final_kernel(const double* __restrict__ part, int C, int nchunk, double inv_count,
                                    const float* __restrict__ gamma, const float* __restrict__ invstd,
                                    const float* __restrict__ xhat_sum, float* __restrict__ dgamma, float* __restrict__ dbeta,
                                    float* __restrict__ dconv_bias) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double s1 = 0.0, s2 = 0.0;
    for (int j = 0; j < nchunk; ++j) {
        s1 += part[((long)c * nchunk + j) * 2];
        s2 += part[((long)c * nchunk + j) * 2 + 1];
    }
    dbeta[c] = (float)s1;
    dgamma[c] = (float)s2;
    if (dconv_bias) dconv_bias[c] = xhat_sum ? (float)(-(double)gamma[c] * (double)invstd[c] * (s2 * inv_count) * (double)xhat_sum[c]) : 0.f;
}

template <int V>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ da, const float* __restrict__ y,
                                                           const float* __restrict__ mean,
                                                           const float* __restrict__ invstd,
                                                           const float* __restrict__ gamma,
                                                           const float* __restrict__ beta,
                                                           float* __restrict__ dgamma,
                                                           float* __restrict__ dbeta, float* __restrict__ dy,
                                                           int C, long HW, float inv_count, const double* __restrict__ part, int nchunk,
                                                           double inv_count_d, const float* __restrict__ xhat_sum,
                                                           float* __restrict__ dconv_bias) {
    typedef typename VecT<V>::type vec;
    __shared__ double stage[2 * BN_MAX_CHUNKS + 2];
    const long plane = blockIdx.y;
    const int c = (int)(plane % C);
    const float mu = mean[c], is = invstd[c], g = gamma[c], b = beta[c];
    // dbeta / dgamma of this channel from the partial sums (what bn_bwd_final_kernel did in a launch of its own, same order)
    double s1, s2;
    combine_partials(part, c, nchunk, stage, s1, s2);
    const float dbeta_c = (float)s1, dgamma_c = (float)s2;
    if (blockIdx.x == 0 && plane < C && threadIdx.x == 0) {
        dbeta[c] = dbeta_c;
        dgamma[c] = dgamma_c;
        if (dconv_bias) dconv_bias[c] = xhat_sum ? (float)(-(double)g * (double)is * (s2 * inv_count_d) * (double)xhat_sum[c]) : 0.f;
    }
    const float k1 = dbeta_c * inv_count, k2 = dgamma_c * inv_count, gs = g * is;
    const vec* yp = (const vec*)(y + plane * HW);
    const vec* dp = (const vec*)(da + plane * HW);
    vec* op = (vec*)(dy + plane * HW);
    const long nv = HW / V;
    __builtin_amdgcn_s_setprio(3);                   // see bn_bwd_partial_kernel
    constexpr int U = V == 4 ? 3 : 4;                // three 16-byte pairs in flight keep the kernel at <= 56 registers (four: 58)
    const long stride = (long)gridDim.x * 256;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nv; i += U * stride) {
        vec yv[U], dv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long iu = i + u * stride < nv ? i + u * stride : i;
            yv[u] = yp[iu];
            dv[u] = dp[iu];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            vec o;
#pragma unroll
            for (int k = 0; k < V; ++k) {
                const float xh = (vget<V>(yv[u], k) - mu) * is;
                const float ov = xh * g + b;
                const float dz = ov > 0.f ? vget<V>(dv[u], k) : 0.f;
                vset<V>(o, k, gs * (dz - k1 - xh * k2));
            }
            if (i + u * stride < nv) op[i + u * stride] = o;
        }
    }
}

// ------------------------------------------------------------------ fractional max pool
// ATen rule in float32 (aten/src/ATen/native/FractionalMaxPool2d.cpp, restated in oracle/vista_oracle.py
// fracpool_intervals): start(i) = int((i + u) * alpha) - int(u * alpha), last = in - 2.  No FMA contraction.
__device__ __forceinline__ int frac_start(int i, float u, float alpha, int in, int out) {
    if (i == out - 1) return in - 2;
    const float a = __fmul_rn(__fadd_rn((float)i, u), alpha);
    const float b = __fmul_rn(u, alpha);
    return (int)a - (int)b;
}

__global__ __launch_bounds__(256) void fracpool_fwd_kernel(const float* __restrict__ x, const float* __restrict__ samples,
                                                           float* __restrict__ out, int32_t* __restrict__ idx, int H,
                                                           int W, int OH, int OW, float alpha_h, float alpha_w) {
    const long plane = blockIdx.y;
    const float uw = samples[plane * 2], uh = samples[plane * 2 + 1];
    const float* xp = x + plane * (long)H * W;
    const int total = OH * OW;
    for (int o = blockIdx.x * 256 + threadIdx.x; o < total; o += gridDim.x * 256) {
        const int oh = o / OW, ow = o % OW;
        const int hs = frac_start(oh, uh, alpha_h, H, OH);
        const int ws = frac_start(ow, uw, alpha_w, W, OW);
        int best = hs * W + ws;
        float mv = -INFINITY;
#pragma unroll
        for (int dh = 0; dh < 2; ++dh)
#pragma unroll
            for (int dw = 0; dw < 2; ++dw) {
                const int p = (hs + dh) * W + ws + dw;
                const float v = xp[p];
                if (v > mv || v != v) { mv = v; best = p; }
            }
        out[plane * total + o] = mv;
        idx[plane * total + o] = best;
    }
}

__global__ __launch_bounds__(256) void pool_bwd_scatter_kernel(const float* __restrict__ dout,
                                                               const int32_t* __restrict__ idx, float* __restrict__ dx,
                                                               long in_plane, int out_plane) {
    const long plane = blockIdx.y;
    for (int o = blockIdx.x * 256 + threadIdx.x; o < out_plane; o += gridDim.x * 256) {
        const long oi = plane * out_plane + o;
        atomicAdd(dx + plane * in_plane + idx[oi], dout[oi]);   // <= 2 contributions per input: order-independent
    }
}

// BatchNorm-apply + ReLU + fractional max pool in one pass over the conv output: where a pooling layer follows, the
// activation a = relu(bn(y)) itself is needed by nobody (the backward recomputes it from y, the next layer reads the
// pooled tensor), so it is never written.  Same arithmetic per element as bn_relu_apply_kernel and the same window /
// first-maximum rule as fracpool_fwd_kernel.
__global__ __launch_bounds__(256) void bn_relu_fracpool_fwd_kernel(const float* __restrict__ y, const float* __restrict__ mean,
                                                                   const float* __restrict__ invstd,
                                                                   const float* __restrict__ gamma,
                                                                   const float* __restrict__ beta,
                                                                   const float* __restrict__ samples, float* __restrict__ out,
                                                                   int32_t* __restrict__ idx, int C, int H, int W, int OH, int OW,
                                                                   float alpha_h, float alpha_w, BnFusedStats fs) {
    __shared__ double stage[2 * BN_MAX_CHUNKS + 2];
    const long plane = blockIdx.y;
    const int c = (int)(plane % C);
    float mu, is;
    if (fs.part) {
        double s_, q_;
        combine_partials(fs.part, c, fs.nchunk, stage, s_, q_);
        bn_train_stats_of(s_, q_, fs.count, fs.eps, fs.momentum, c, blockIdx.x == 0 && plane < C, fs.mean, fs.invstd, fs.rmean, fs.rvar, fs.nbt,
                          fs.xhat_sum, mu, is);
    } else {
        mu = mean[c];
        is = invstd[c];
    }
    const float g = gamma[c], b = beta[c];
    const float uw = samples[plane * 2], uh = samples[plane * 2 + 1];
    const float* yp = y + plane * (long)H * W;
    const int total = OH * OW;
    for (int o = blockIdx.x * 256 + threadIdx.x; o < total; o += gridDim.x * 256) {
        const int oh = o / OW, ow = o % OW;
        const int hs = frac_start(oh, uh, alpha_h, H, OH);
        const int ws = frac_start(ow, uw, alpha_w, W, OW);
        int best = hs * W + ws;
        float mv = -INFINITY;
#pragma unroll
        for (int dh = 0; dh < 2; ++dh)
#pragma unroll
            for (int dw = 0; dw < 2; ++dw) {
                const int p = (hs + dh) * W + ws + dw;
                const float t = (yp[p] - mu) * is * g + b;
                const float v = t > 0.f ? t : 0.f;
                if (v > mv || v != v) { mv = v; best = p; }
            }
        out[plane * total + o] = mv;
        idx[plane * total + o] = best;
    }
}

// Same gradient with the input plane assembled in LDS: one workgroup per (n, c) plane zeroes an LDS image, adds its
// outputs' gradients there (ds_add_f32; <= 2 contributions per pixel, so the order cannot matter) and writes the plane out
// once, in full lines.  No zero-fill pass over dx and no global atomics: 250 MB instead of ~550 MB of traffic for the
// first pooling layer (30x600 planes, 72 KB of LDS).
template <int CAP>
__global__ __launch_bounds__(512) void pool_bwd_lds_kernel(const float* __restrict__ dout, const int32_t* __restrict__ idx,
                                                           float* __restrict__ dx, int in_plane, int out_plane) {
    __shared__ __attribute__((aligned(16))) float pl[CAP];
    const long plane = blockIdx.x;
    const int tid = threadIdx.x;
    const float* dp = dout + plane * out_plane;
    const int32_t* ip = idx + plane * out_plane;
    // the first batch of (gradient, winner index) pairs is fetched before the LDS image is cleared, so its latency hides
    constexpr int U = 4;
    float v[U];
    int ix[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int o = tid + u * 512;
        v[u] = o < out_plane ? dp[o] : 0.f;
        ix[u] = o < out_plane ? ip[o] : 0;
    }
    for (int i = tid * 4; i < CAP; i += 512 * 4) *(f32x4*)(pl + i) = (f32x4){0.f, 0.f, 0.f, 0.f};
    __syncthreads();
    for (int base = tid;; ) {
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (base + u * 512 < out_plane) atomicAdd(&pl[ix[u]], v[u]);
        base += U * 512;
        if (base >= out_plane) break;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int o = base + u * 512;
            v[u] = o < out_plane ? dp[o] : 0.f;
            ix[u] = o < out_plane ? ip[o] : 0;
        }
    }
    __syncthreads();
    float* xp = dx + plane * (long)in_plane;
    if ((in_plane & 3) == 0 && ((((uintptr_t)dx) & 15) == 0)) {
        for (int i = tid * 4; i < in_plane; i += 512 * 4) *(f32x4*)(xp + i) = *(const f32x4*)(pl + i);
    } else {
        for (int i = tid; i < in_plane; i += 512) xp[i] = pl[i];
    }
}

// dy of a pooled layer in ONE pass: gradient of FractionalMaxPool2d (gather form) + ReLU + batch-stat BatchNorm.
// One workgroup per (n, c) plane.  The pooling windows are rebuilt from the samples with the forward's rule.  The window
// starts of outputs 0..out-2 are strictly increasing (alpha >= 1.01), so a column is the first column of at most one window
// and the second column of at most one: two inverse maps per axis in LDS (wa: start == w, wb: start == w-1).  The LAST
// window, pinned at in-2, is kept out of the maps and handled on its own (float32 rounding can make it coincide with its
// neighbour).  A thread owns 4 consecutive pixels of a row: the windows that can have elected one of them are the <= 5 with
// start in [w0-1, w0+3], in <= 2 window rows; their (winner index, gradient) pairs are fetched unconditionally (20
// independent loads in flight per thread) and matched in registers.  No atomics, no LDS image of the plane, full occupancy;
// the BatchNorm backward is applied on the way out, so neither da nor a zero-filled plane ever goes to HBM.
__global__ __launch_bounds__(256) void bn_pool_bwd_apply_kernel(const float* __restrict__ dout, const int32_t* __restrict__ idx,
                                                                const float* __restrict__ samples, const float* __restrict__ y,
                                                                const float* __restrict__ mean, const float* __restrict__ invstd,
                                                                const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                                float* __restrict__ dy, int C, int H, int W, int OH, int OW,
                                                                float alpha_h, float alpha_w, float inv_count, const double* __restrict__ part,
                                                                int nchunk, double inv_count_d, const float* __restrict__ xhat_sum,
                                                                float* __restrict__ dconv_bias) {
    __shared__ double stage[2 * BN_MAX_CHUNKS + 2];
    extern __shared__ int inv[];                 // wa[W + 4], wb[W + 4], ha[H], hb[H]
    int* wa = inv;
    int* wb = inv + W + 4;
    int* ha = inv + 2 * W + 8;
    int* hb = inv + 2 * W + 8 + H;
    const long plane = blockIdx.x;
    const int c = (int)(plane % C);
    const int tid = threadIdx.x;
    const float uw = samples[plane * 2], uh = samples[plane * 2 + 1];
    for (int i = tid; i < 2 * W + 8 + 2 * H; i += 256) inv[i] = -1;
    __syncthreads();
    for (int o = tid; o < OW - 1; o += 256) {
        const int st = frac_start(o, uw, alpha_w, W, OW);
        wa[st] = o;
        wb[st + 1] = o;
    }
    for (int o = tid; o < OH - 1; o += 256) {
        const int st = frac_start(o, uh, alpha_h, H, OH);
        ha[st] = o;
        hb[st + 1] = o;
    }
    __syncthreads();
    const float mu = mean[c], is = invstd[c], g = gamma[c], b = beta[c];
    double s1, s2;                               // see bn_bwd_apply_kernel
    combine_partials(part, c, nchunk, stage, s1, s2);
    const float dbeta_c = (float)s1, dgamma_c = (float)s2;
    if (plane < C && tid == 0) {
        dbeta[c] = dbeta_c;
        dgamma[c] = dgamma_c;
        if (dconv_bias) dconv_bias[c] = xhat_sum ? (float)(-(double)g * (double)is * (s2 * inv_count_d) * (double)xhat_sum[c]) : 0.f;
    }
    const float k1 = dbeta_c * inv_count, k2 = dgamma_c * inv_count, gs = g * is;
    const float* dp = dout + plane * (long)OH * OW;
    const int32_t* ip = idx + plane * (long)OH * OW;
    const f32x4* yp = (const f32x4*)(y + plane * (long)H * W);
    f32x4* op = (f32x4*)(dy + plane * (long)H * W);
    const int GW = W >> 2, G = H * GW;
    for (int gi = tid; gi < G; gi += 256) {
        const int h = gi / GW, w0 = (gi - h * GW) << 2;
        const f32x4 yv = yp[gi];
        // candidate window rows (start == h, start == h-1) and columns (start == w0-1 .. w0+3); -1 = none
        const int orow[2] = {ha[h], hb[h]};
        const int ocol[5] = {wb[w0], wa[w0], wa[w0 + 1], wa[w0 + 2], wa[w0 + 3]};
        float da[4] = {0.f, 0.f, 0.f, 0.f};
        const int p0 = h * W + w0;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            // a wave's 64 pixel groups span at most two image rows, and with window strides >= 2 a row has ONE candidate
            // window row: skip the other wave-uniformly (its loads were half of this kernel's address traffic)
            if (!__any(orow[r] >= 0)) continue;
            int wi[5];
            float wd[5];
#pragma unroll
            for (int k = 0; k < 5; ++k) {
                const int o = max(orow[r], 0) * OW + max(ocol[k], 0);       // clamped: the load is unconditional
                wi[k] = ip[o];
                wd[k] = dp[o];
            }
#pragma unroll
            for (int k = 0; k < 5; ++k) {
                const bool ok = orow[r] >= 0 && ocol[k] >= 0;
                const int j = wi[k] - p0;                                    // winner relative to this thread's 4 pixels
#pragma unroll
                for (int q = 0; q < 4; ++q) da[q] += (ok && j == q) ? wd[k] : 0.f;
            }
        }
        // the pinned last window row / column (rare: last two rows, last pixel group of a row)
        if (h >= H - 2 || w0 + 3 >= W - 2) {
            const int lr = h >= H - 2 ? OH - 1 : -1, lc = w0 + 3 >= W - 2 ? OW - 1 : -1;
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const int orr = r < 2 ? orow[r] : lr;
#pragma unroll
                for (int k = 0; k < 6; ++k) {
                    const int occ = k < 5 ? ocol[k] : lc;
                    if ((r < 2 && k < 5) || orr < 0 || occ < 0) continue;   // regular x regular pairs were counted above
                    const int o = orr * OW + occ;
                    const int j = ip[o] - p0;
                    if (j >= 0 && j < 4) da[j] += dp[o];
                }
            }
        }
        f32x4 ov;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float xh = (yv[q] - mu) * is;
            const float act = xh * g + b;
            const float dz = act > 0.f ? da[q] : 0.f;
            ov[q] = gs * (dz - k1 - xh * k2);
        }
        op[gi] = ov;
    }
}

__global__ __launch_bounds__(256) void relu_maxpool2_fwd_kernel(const float* __restrict__ x, float* __restrict__ out,
                                                                int32_t* __restrict__ idx, int H, int W, int OH, int OW) {
    const long plane = blockIdx.y;
    const float* xp = x + plane * (long)H * W;
    const int total = OH * OW;
    for (int o = blockIdx.x * 256 + threadIdx.x; o < total; o += gridDim.x * 256) {
        const int oh = o / OW, ow = o % OW;
        int best = (2 * oh) * W + 2 * ow;
        float mv = -INFINITY;
#pragma unroll
        for (int dh = 0; dh < 2; ++dh)
#pragma unroll
            for (int dw = 0; dw < 2; ++dw) {
                const int p = (2 * oh + dh) * W + 2 * ow + dw;
                const float v = fmaxf(xp[p], 0.f);
                if (v > mv) { mv = v; best = p; }
            }
        out[plane * total + o] = mv;
        idx[plane * total + o] = best;
    }
}

// windows do not overlap: plain stores into a zero-filled dx; gradient passes only where relu was active
__global__ __launch_bounds__(256) void relu_maxpool2_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ out,
                                                                const int32_t* __restrict__ idx, float* __restrict__ dx,
                                                                long in_plane, int out_plane) {
    const long plane = blockIdx.y;
    for (int o = blockIdx.x * 256 + threadIdx.x; o < out_plane; o += gridDim.x * 256) {
        const long oi = plane * out_plane + o;
        dx[plane * in_plane + idx[oi]] = out[oi] > 0.f ? dout[oi] : 0.f;
    }
}

}  // namespace

// widest load (in floats) usable on planes of hw elements starting at these bases
static inline int vec_width(int hw, const void* a, const void* b) {
    const uintptr_t m = (uintptr_t)a | (uintptr_t)b;
    if (hw % 4 == 0 && (m & 15) == 0) return 4;
    if (hw % 2 == 0 && (m & 7) == 0) return 2;
    return 1;
}
// elements of one channel per reduction chunk, rounded up to a multiple of 4 so that vectors never straddle chunks
static inline long chunk_len(long per_channel, int nchunk) {
    const long per = (per_channel + nchunk - 1) / nchunk;
    return (per + 3) / 4 * 4;
}

extern "C" size_t vocr_bn_workspace_bytes(int n, int c, int hw) {
    if (n <= 0 || c <= 0 || hw <= 0) return 0;
    return (size_t)c * bn_nchunk((long)n * hw) * 2 * sizeof(double);
}

extern "C" int vocr_bn_train_stats(const float* y, int n, int c, int hw, float eps, float momentum, float* mean,
                                   float* invstd, float* running_mean, float* running_var, int64_t* num_batches_tracked,
                                   float* xhat_sum, void* workspace, void* stream) {
    VOCR_CHECK_ARG(y && mean && invstd && workspace && n > 0 && c > 0 && hw > 0, "vocr_bn_train_stats: bad argument");
    hipStream_t s = (hipStream_t)stream;
    const int nchunk = bn_nchunk((long)n * hw);
    const int V = vec_width(hw, y, y);
    const long per = chunk_len((long)n * hw, nchunk);
    if (V == 4) bn_stats_partial_kernel<4><<<dim3(c, nchunk), 256, 0, s>>>(y, (double*)workspace, n, c, hw, nchunk, per);
    else if (V == 2) bn_stats_partial_kernel<2><<<dim3(c, nchunk), 256, 0, s>>>(y, (double*)workspace, n, c, hw, nchunk, per);
    else bn_stats_partial_kernel<1><<<dim3(c, nchunk), 256, 0, s>>>(y, (double*)workspace, n, c, hw, nchunk, per);
    VOCR_CHECK_LAUNCH("vocr_bn_train_stats(partial)");
    bn_stats_final_kernel<<<vocr_cdiv(c, 64), 64, 0, s>>>((const double*)workspace, c, nchunk, (long)n * hw, eps, momentum,
                                                          mean, invstd, running_mean, running_var, (long long*)num_batches_tracked, xhat_sum);
    VOCR_CHECK_LAUNCH("vocr_bn_train_stats(final)");
    return VOCR_OK;
}

extern "C" int vocr_bn_eval_stats(const float* running_mean, const float* running_var, int c, float eps, float* mean,
                                  float* invstd, void* stream) {
    VOCR_CHECK_ARG(running_mean && running_var && mean && invstd && c > 0, "vocr_bn_eval_stats: bad argument");
    bn_eval_stats_kernel<<<vocr_cdiv(c, 64), 64, 0, (hipStream_t)stream>>>(running_mean, running_var, c, eps, mean, invstd);
    VOCR_CHECK_LAUNCH("vocr_bn_eval_stats");
    return VOCR_OK;
}

static inline dim3 plane_grid(long planes, long per_plane) {
    long gx = (per_plane + 1023) / 1024;
    if (gx < 1) gx = 1;
    if (gx > 64) gx = 64;
    return dim3((unsigned)gx, (unsigned)planes);
}

extern "C" int vocr_bn_relu_apply(const float* y, const float* mean, const float* invstd, const float* gamma,
                                  const float* beta, float* out, int n, int c, int hw, void* stream) {
    VOCR_CHECK_ARG(y && mean && invstd && gamma && beta && out && n > 0 && c > 0 && hw > 0, "vocr_bn_relu_apply: bad argument");
    VOCR_CHECK_ARG((long)n * c <= 65535, "vocr_bn_relu_apply: n*c > 65535 planes");
    const int V = vec_width(hw, y, out);
    const dim3 grid = plane_grid((long)n * c, hw / V);
    const BnFusedStats none = {nullptr, 0, 0, 0.f, 0.f, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    if (V == 4) bn_relu_apply_kernel<4><<<grid, 256, 0, (hipStream_t)stream>>>(y, mean, invstd, gamma, beta, out, c, hw, none);
    else if (V == 2) bn_relu_apply_kernel<2><<<grid, 256, 0, (hipStream_t)stream>>>(y, mean, invstd, gamma, beta, out, c, hw, none);
    else bn_relu_apply_kernel<1><<<grid, 256, 0, (hipStream_t)stream>>>(y, mean, invstd, gamma, beta, out, c, hw, none);
    VOCR_CHECK_LAUNCH("vocr_bn_relu_apply");
    return VOCR_OK;
}

namespace {
// the statistics pass of a training-mode layer: partial sums only; the pass that follows combines them (BnFusedStats)
int bn_launch_partial(const float* y, int n, int c, int hw, void* workspace, hipStream_t s, int* nchunk_out) {
    const int nchunk = bn_nchunk((long)n * hw);
    const int V = vec_width(hw, y, y);
    const long per = chunk_len((long)n * hw, nchunk);
    if (V == 4) bn_stats_partial_kernel<4><<<dim3(c, nchunk), 256, 0, s>>>(y, (double*)workspace, n, c, hw, nchunk, per);
    else if (V == 2) bn_stats_partial_kernel<2><<<dim3(c, nchunk), 256, 0, s>>>(y, (double*)workspace, n, c, hw, nchunk, per);
    else bn_stats_partial_kernel<1><<<dim3(c, nchunk), 256, 0, s>>>(y, (double*)workspace, n, c, hw, nchunk, per);
    *nchunk_out = nchunk;
    return VOCR_OK;
}
}  // namespace

extern "C" int vocr_bn_train_relu_apply(const float* y, const float* gamma, const float* beta, float* out, int n, int c, int hw, float eps,
                                        float momentum, float* mean, float* invstd, float* running_mean, float* running_var,
                                        int64_t* num_batches_tracked, float* xhat_sum, void* workspace, void* stream) {
    VOCR_CHECK_ARG(y && gamma && beta && out && mean && invstd && workspace && n > 0 && c > 0 && hw > 0, "vocr_bn_train_relu_apply: bad argument");
    VOCR_CHECK_ARG((long)n * c <= 65535, "vocr_bn_train_relu_apply: n*c > 65535 planes");
    hipStream_t s = (hipStream_t)stream;
    int nchunk = 0;
    bn_launch_partial(y, n, c, hw, workspace, s, &nchunk);
    VOCR_CHECK_LAUNCH("vocr_bn_train_relu_apply(partial)");
    const BnFusedStats fs = {(const double*)workspace, nchunk, (long)n * hw, eps, momentum, mean, invstd, running_mean, running_var, xhat_sum,
                             (long long*)num_batches_tracked};
    const int V = vec_width(hw, y, out);
    const dim3 grid = plane_grid((long)n * c, hw / V);
    if (V == 4) bn_relu_apply_kernel<4><<<grid, 256, 0, s>>>(y, mean, invstd, gamma, beta, out, c, hw, fs);
    else if (V == 2) bn_relu_apply_kernel<2><<<grid, 256, 0, s>>>(y, mean, invstd, gamma, beta, out, c, hw, fs);
    else bn_relu_apply_kernel<1><<<grid, 256, 0, s>>>(y, mean, invstd, gamma, beta, out, c, hw, fs);
    VOCR_CHECK_LAUNCH("vocr_bn_train_relu_apply(apply)");
    return VOCR_OK;
}

extern "C" int vocr_bn_relu_bwd(const float* da, const float* y, const float* mean, const float* invstd,
                                const float* gamma, const float* beta, const float* xhat_sum, float* dy, float* dgamma, float* dbeta,
                                float* dconv_bias, int n, int c, int hw, void* workspace, void* stream) {
    VOCR_CHECK_ARG(da && y && mean && invstd && gamma && beta && dy && dgamma && dbeta && workspace, "vocr_bn_relu_bwd: null pointer");
    VOCR_CHECK_ARG(n > 0 && c > 0 && hw > 0 && (long)n * c <= 65535, "vocr_bn_relu_bwd: bad shape");
    hipStream_t s = (hipStream_t)stream;
    const int nchunk = bn_nchunk((long)n * hw);
    const int V = (vec_width(hw, da, y) < vec_width(hw, dy, dy)) ? vec_width(hw, da, y) : vec_width(hw, dy, dy);
    const long per = chunk_len((long)n * hw, nchunk);
    if (V == 4) bn_bwd_partial_kernel<4><<<dim3(c, nchunk), 256, 0, s>>>(da, y, mean, invstd, gamma, beta, (double*)workspace, n, c, hw, nchunk, per);
    else if (V == 2) bn_bwd_partial_kernel<2><<<dim3(c, nchunk), 256, 0, s>>>(da, y, mean, invstd, gamma, beta, (double*)workspace, n, c, hw, nchunk, per);
    else bn_bwd_partial_kernel<1><<<dim3(c, nchunk), 256, 0, s>>>(da, y, mean, invstd, gamma, beta, (double*)workspace, n, c, hw, nchunk, per);
    VOCR_CHECK_LAUNCH("vocr_bn_relu_bwd(partial)");
    // (the chunks of a channel are added by every workgroup of the apply pass, in chunk order: no launch of its own for that)
    const dim3 grid = plane_grid((long)n * c, hw / V);
    const float inv_count = 1.0f / (float)((long)n * hw);
    const double inv_count_d = 1.0 / (double)((long)n * hw);
    const double* part = (const double*)workspace;
    if (V == 4) bn_bwd_apply_kernel<4><<<grid, 256, 0, s>>>(da, y, mean, invstd, gamma, beta, dgamma, dbeta, dy, c, hw, inv_count, part, nchunk, inv_count_d, xhat_sum, dconv_bias);
    else if (V == 2) bn_bwd_apply_kernel<2><<<grid, 256, 0, s>>>(da, y, mean, invstd, gamma, beta, dgamma, dbeta, dy, c, hw, inv_count, part, nchunk, inv_count_d, xhat_sum, dconv_bias);
    else bn_bwd_apply_kernel<1><<<grid, 256, 0, s>>>(da, y, mean, invstd, gamma, beta, dgamma, dbeta, dy, c, hw, inv_count, part, nchunk, inv_count_d, xhat_sum, dconv_bias);
    VOCR_CHECK_LAUNCH("vocr_bn_relu_bwd(apply)");
    return VOCR_OK;
}

extern "C" int vocr_bn_relu_fracpool2x2_bwd_supported(int h, int w, int oh, int ow) {
    if (h < 2 || w < 2 || oh < 2 || ow < 2 || oh > h - 1 || ow > w - 1) return 0;
    // strictly increasing window starts need alpha comfortably above 1 (float32 interval arithmetic)
    // ... and rows are processed as 16-byte pixel groups
    return ((float)(h - 2) / (float)(oh - 1) >= 1.01f && (float)(w - 2) / (float)(ow - 1) >= 1.01f && w % 4 == 0 &&
            (2 * w + 8 + 2 * h) * 4 <= 64 * 1024) ? 1 : 0;
}

extern "C" int vocr_bn_relu_fracpool2x2_bwd(const float* dout, const int32_t* idx, const float* samples, const float* y,
                                            const float* mean, const float* invstd, const float* gamma, const float* beta,
                                            const float* xhat_sum, float* dy, float* dgamma, float* dbeta, float* dconv_bias, int n,
                                            int c, int h, int w, int oh, int ow, void* workspace, void* stream) {
    VOCR_CHECK_ARG(dout && idx && samples && y && mean && invstd && gamma && beta && dy && dgamma && dbeta && workspace,
                   "vocr_bn_relu_fracpool2x2_bwd: null pointer");
    VOCR_CHECK_ARG(n > 0 && c > 0 && (long)n * c <= 65535 && vocr_bn_relu_fracpool2x2_bwd_supported(h, w, oh, ow),
                   "vocr_bn_relu_fracpool2x2_bwd: unsupported shape h=%d w=%d oh=%d ow=%d", h, w, oh, ow);
    hipStream_t s = (hipStream_t)stream;
    const long op = (long)oh * ow, hw = (long)h * w;
    const int nchunk = bn_nchunk((long)n * hw);                  // sized like the dense pass (same workspace)
    const long per = ((long)n * op + nchunk - 1) / nchunk;
    bn_pool_bwd_partial_kernel<<<dim3(c, nchunk), 256, 0, s>>>(dout, idx, y, mean, invstd, gamma, beta, (double*)workspace, n, c, hw, op,
                                                              nchunk, per);
    VOCR_CHECK_LAUNCH("vocr_bn_relu_fracpool2x2_bwd(partial)");
    const float alpha_h = (float)(h - 2) / (float)(oh - 1), alpha_w = (float)(w - 2) / (float)(ow - 1);
    bn_pool_bwd_apply_kernel<<<(unsigned)(n * c), 256, (size_t)(2 * w + 8 + 2 * h) * sizeof(int), s>>>(
        dout, idx, samples, y, mean, invstd, gamma, beta, dgamma, dbeta, dy, c, h, w, oh, ow, alpha_h, alpha_w,
        1.0f / (float)((long)n * hw), (const double*)workspace, nchunk, 1.0 / (double)((long)n * hw), xhat_sum, dconv_bias);
    VOCR_CHECK_LAUNCH("vocr_bn_relu_fracpool2x2_bwd(apply)");
    return VOCR_OK;
}

extern "C" int vocr_fracpool2x2_fwd(const float* x, const float* samples, float* out, int32_t* idx, int n, int c, int h,
                                    int w, int oh, int ow, void* stream) {
    VOCR_CHECK_ARG(x && samples && out && idx, "vocr_fracpool2x2_fwd: null pointer");
    VOCR_CHECK_ARG(n > 0 && c > 0 && h >= 2 && w >= 2 && oh >= 1 && ow >= 1 && oh <= h - 1 && ow <= w - 1 && (long)n * c <= 65535,
                   "vocr_fracpool2x2_fwd: bad shape h=%d w=%d oh=%d ow=%d", h, w, oh, ow);
    // alpha in float32 exactly as ATen: float(in - pool) / float(out - 1)
    const float alpha_h = oh > 1 ? (float)(h - 2) / (float)(oh - 1) : 0.f;
    const float alpha_w = ow > 1 ? (float)(w - 2) / (float)(ow - 1) : 0.f;
    fracpool_fwd_kernel<<<plane_grid((long)n * c, (long)oh * ow), 256, 0, (hipStream_t)stream>>>(x, samples, out, idx, h, w, oh,
                                                                                             ow, alpha_h, alpha_w);
    VOCR_CHECK_LAUNCH("vocr_fracpool2x2_fwd");
    return VOCR_OK;
}

extern "C" int vocr_bn_relu_fracpool2x2_fwd(const float* y, const float* mean, const float* invstd, const float* gamma,
                                            const float* beta, const float* samples, float* out, int32_t* idx, int n, int c,
                                            int h, int w, int oh, int ow, void* stream) {
    VOCR_CHECK_ARG(y && mean && invstd && gamma && beta && samples && out && idx, "vocr_bn_relu_fracpool2x2_fwd: null pointer");
    VOCR_CHECK_ARG(n > 0 && c > 0 && h >= 2 && w >= 2 && oh >= 1 && ow >= 1 && oh <= h - 1 && ow <= w - 1 && (long)n * c <= 65535,
                   "vocr_bn_relu_fracpool2x2_fwd: bad shape h=%d w=%d oh=%d ow=%d", h, w, oh, ow);
    const float alpha_h = oh > 1 ? (float)(h - 2) / (float)(oh - 1) : 0.f;
    const float alpha_w = ow > 1 ? (float)(w - 2) / (float)(ow - 1) : 0.f;
    const BnFusedStats none = {nullptr, 0, 0, 0.f, 0.f, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    bn_relu_fracpool_fwd_kernel<<<plane_grid((long)n * c, (long)oh * ow), 256, 0, (hipStream_t)stream>>>(
        y, mean, invstd, gamma, beta, samples, out, idx, c, h, w, oh, ow, alpha_h, alpha_w, none);
    VOCR_CHECK_LAUNCH("vocr_bn_relu_fracpool2x2_fwd");
    return VOCR_OK;
}

extern "C" int vocr_bn_train_relu_fracpool2x2_fwd(const float* y, const float* gamma, const float* beta, const float* samples, float* out,
                                                  int32_t* idx, int n, int c, int h, int w, int oh, int ow, float eps, float momentum,
                                                  float* mean, float* invstd, float* running_mean, float* running_var,
                                                  int64_t* num_batches_tracked, float* xhat_sum, void* workspace, void* stream) {
    VOCR_CHECK_ARG(y && gamma && beta && samples && out && idx && mean && invstd && workspace, "vocr_bn_train_relu_fracpool2x2_fwd: null pointer");
    VOCR_CHECK_ARG(n > 0 && c > 0 && h >= 2 && w >= 2 && oh >= 1 && ow >= 1 && oh <= h - 1 && ow <= w - 1 && (long)n * c <= 65535,
                   "vocr_bn_train_relu_fracpool2x2_fwd: bad shape h=%d w=%d oh=%d ow=%d", h, w, oh, ow);
    hipStream_t s = (hipStream_t)stream;
    int nchunk = 0;
    bn_launch_partial(y, n, c, h * w, workspace, s, &nchunk);
    VOCR_CHECK_LAUNCH("vocr_bn_train_relu_fracpool2x2_fwd(partial)");
    const BnFusedStats fs = {(const double*)workspace, nchunk, (long)n * h * w, eps, momentum, mean, invstd, running_mean, running_var, xhat_sum,
                             (long long*)num_batches_tracked};
    const float alpha_h = oh > 1 ? (float)(h - 2) / (float)(oh - 1) : 0.f;
    const float alpha_w = ow > 1 ? (float)(w - 2) / (float)(ow - 1) : 0.f;
    bn_relu_fracpool_fwd_kernel<<<plane_grid((long)n * c, (long)oh * ow), 256, 0, s>>>(y, mean, invstd, gamma, beta, samples, out, idx, c, h, w, oh, ow,
                                                                                       alpha_h, alpha_w, fs);
    VOCR_CHECK_LAUNCH("vocr_bn_train_relu_fracpool2x2_fwd(apply)");
    return VOCR_OK;
}

extern "C" int vocr_fracpool2x2_bwd(const float* dout, const int32_t* idx, float* dx, int n, int c, int h, int w, int oh,
                                    int ow, void* stream) {
    VOCR_CHECK_ARG(dout && idx && dx && n > 0 && c > 0 && (long)n * c <= 65535, "vocr_fracpool2x2_bwd: bad argument");
    hipStream_t s = (hipStream_t)stream;
    const long in_plane = (long)h * w;
    static const int use_lds = VOCR_EXPERIMENT_INT("VOCR_POOL_LDS", 1);      // experiments
    if (use_lds && in_plane <= 8192) {
        pool_bwd_lds_kernel<8192><<<(unsigned)(n * c), 512, 0, s>>>(dout, idx, dx, (int)in_plane, oh * ow);
    } else if (use_lds && in_plane <= 18432) {
        pool_bwd_lds_kernel<18432><<<(unsigned)(n * c), 512, 0, s>>>(dout, idx, dx, (int)in_plane, oh * ow);
    } else if (use_lds && in_plane <= 36864) {
        pool_bwd_lds_kernel<36864><<<(unsigned)(n * c), 512, 0, s>>>(dout, idx, dx, (int)in_plane, oh * ow);
    } else {
        if (hipMemsetAsync(dx, 0, (size_t)n * c * in_plane * sizeof(float), s) != hipSuccess) {
            vocr_set_error("vocr_fracpool2x2_bwd: memset failed");
            return VOCR_ELAUNCH;
        }
        pool_bwd_scatter_kernel<<<plane_grid((long)n * c, (long)oh * ow), 256, 0, s>>>(dout, idx, dx, in_plane, oh * ow);
    }
    VOCR_CHECK_LAUNCH("vocr_fracpool2x2_bwd");
    return VOCR_OK;
}

extern "C" int vocr_relu_maxpool2_fwd(const float* x, float* out, int32_t* idx, int n, int c, int h, int w, void* stream) {
    VOCR_CHECK_ARG(x && out && idx && n > 0 && c > 0 && h >= 2 && w >= 2 && (long)n * c <= 65535, "vocr_relu_maxpool2_fwd: bad argument");
    const int oh = h / 2, ow = w / 2;
    relu_maxpool2_fwd_kernel<<<plane_grid((long)n * c, (long)oh * ow), 256, 0, (hipStream_t)stream>>>(x, out, idx, h, w, oh, ow);
    VOCR_CHECK_LAUNCH("vocr_relu_maxpool2_fwd");
    return VOCR_OK;
}

extern "C" int vocr_relu_maxpool2_bwd(const float* dout, const float* out, const int32_t* idx, float* dx, int n, int c,
                                      int h, int w, void* stream) {
    VOCR_CHECK_ARG(dout && out && idx && dx && n > 0 && c > 0 && (long)n * c <= 65535, "vocr_relu_maxpool2_bwd: bad argument");
    const int oh = h / 2, ow = w / 2;
    relu_maxpool2_bwd_kernel<<<plane_grid((long)n * c, (long)oh * ow), 256, 0, (hipStream_t)stream>>>(dout, out, idx, dx,
                                                                                                  (long)h * w, oh * ow);
    VOCR_CHECK_LAUNCH("vocr_relu_maxpool2_bwd");
    return VOCR_OK;
}
