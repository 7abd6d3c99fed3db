/*
 * vocr.h — C-ABI of libvocr.so: the MI355X (gfx950) kernels behind VistaOCR's CnnOcrModel hot path.
 *
 * The reference (isi-vista/VistaOCR) is pure Python and has NO FFI of its own: every entry point below
 * replaces a PyTorch/cuDNN/warp-ctc call that the reference makes from Python.  Each declaration cites the
 * reference call site it stands in for (paths under /root/reference/).  INTEGRATION.md shows the ctypes
 * binding a maintainer would add on the reference side.
 *
 * Conventions
 *   - plain C: raw DEVICE pointers, ints, floats; `stream` is a hipStream_t passed as void*.
 *   - every function returns 0 on success, a negative VOCR_E* code otherwise; vocr_last_error() returns a
 *     thread-local message.  No C++ exception crosses the boundary.
 *   - the caller owns every buffer; the library allocates nothing and never synchronises the stream.
 *   - tensors are contiguous fp32 unless stated; images NCHW; sequences time-major [T,B,...].
 */
#ifndef VOCR_H
#define VOCR_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VOCR_OK          0
#define VOCR_EINVAL     -1   /* bad argument (shape, null pointer, unsupported size) */
#define VOCR_ELAUNCH    -2   /* HIP launch error */
#define VOCR_ENODEVICE  -3   /* no gfx950 device visible */
#define VOCR_ECOMM      -4   /* RCCL missing or a collective failed */

/* "health" words: an optional caller-owned int32[2] in device memory.  The library only ever SETS them (report-only: no
 * kernel's result depends on their value), the caller zeroes them — once, and again after it has handled a report.
 *   health[0] != 0: a hand-off inside a persistent LSTM sweep timed out (THAT sweep's output is NaN-poisoned: the sweep
 *                   decides on a per-call word in its workspace, so later sweeps are unaffected);
 *   health[1] != 0: a NaN gradient reached vocr_clamp_adam / vocr_clamp (the reference's clamp_ propagates NaN too).
 * The host reads it whenever it synchronises anyway (vistaocr_amd.train() copies it next to the loss and clears it
 * before raising; vistaocr_amd.ops.reset_health() clears it explicitly). */

const char* vocr_last_error(void);
/* The contract this header describes.  History: 2 = round 2 (caller-owned health word); 3 = round 3/4 (health words report-only and
 * cleared by the caller, LSTM workspace survives between the vocr_lstm_fwd_range calls of a sweep, larger vocr_gemm_workspace_bytes,
 * vocr_gemm_pair / vocr_lstm_bwd_parts / vocr_profile_range_*, x-fastest conv weight pack); 4 = round 5 (packed sequence rows:
 * vocr_lstm_*_packed, vocr_seq_rowmap, vocr_gather_rows; VOCR_LSTM_SWEEP validated; experiment switches compiled out); 5 = round 6
 * (the next layer's x-projection behind the forward sweep: vocr_lstm_fwd_lead / vocr_lstm_xproj_pack,
 * vocr_dropout_mask).  A binding compares vocr_abi_version()
 * with the VOCR_ABI_VERSION it was written against and refuses a library that answers anything else. */
#define VOCR_ABI_VERSION 5
int  vocr_abi_version(void);
/* Named ranges on the profiler's timeline (rocprofv3 --marker-trace), nested push / pop on the calling thread.  roctx is dlopen'ed on
 * first use; returns 0 when the range was recorded, 1 when roctx is not available (not an error), negative on a bad argument.  The
 * reference's only instrumentation is wall-clock prints around the step (src/train_cnn_lstm.py:382-393): the Python mirror marks the
 * same phases (forward / loss / backward / exchange / optimizer, and the model's stages) when VOCR_ROCTX=1. */
int  vocr_profile_range_push(const char* name);
int  vocr_profile_range_pop(void);
/* number of visible HIP devices (>=0) or VOCR_ENODEVICE */
int  vocr_device_count(void);

/* ---- convolution: nn.Conv2d(k=3, pad=1) — src/models/cnnlstm.py:118,264 ------------------------------ */
/* wpack_fwd[(ci*9+kh*3+kw)][co] = w[co][ci][kh][kw];  wpack_dgrad[(co*9+kh*3+kw)][ci] = w[co][ci][2-kh][2-kw] */
int vocr_conv3x3_pack_weights(const float* w, float* wpack_fwd, float* wpack_dgrad, int cout, int cin, void* stream);
/* y[n,co,h,w] = bias[co] + sum wpack[k][co] * x[n,ci,h+kh-1,w+kw-1]   (bias may be NULL).
 * dgrad is the same call with x=dy, cin=Cout, wpack=wpack_dgrad, cout=Cin, bias=NULL. */
int vocr_conv3x3_fwd(const float* x, const float* wpack, const float* bias, float* y,
                     int n, int cin, int h, int w, int cout, void* stream);
/* dw[co][ci][3][3] = sum_{n,h,w} dy[n,co,h,w] * x[n,ci,h+kh-1,w+kw-1];  workspace from *_workspace_bytes. */
size_t vocr_conv3x3_wgrad_workspace_bytes(int n, int cin, int h, int w, int cout);
int vocr_conv3x3_wgrad(const float* x, const float* dy, float* dw, void* workspace,
                       int n, int cin, int h, int w, int cout, void* stream);
/* Same op with a minimal-filtering transform ALONG THE ROW (conv_wino.hip), fp32 operands and accumulation: F(4,3) - half of the
 * direct form's multiplications - for a convolution with at least 64 output channels, F(2,3) - two thirds - below.  A weight pack is
 * OPAQUE: the transformed filter rows in the layout of the kernel that vocr_conv3x3_wino_fwd will pick for a convolution with THAT
 * many output channels, followed by the direct pack's 9 rows per channel (the last partial round of tiles is computed in the direct
 * form).  vocr_conv3x3_wino_pack_floats(cout, cin) = floats of the pack of a convolution with `cout` outputs and `cin` inputs
 * (cout*cin*27 or *21; the data-gradient pack of a layer is the pack of the convolution with cin outputs: ..._pack_floats(cin, cout)),
 * 16-byte aligned.  Needs cout % 4 == 0 (input channels are padded inside) (vocr_conv3x3_wino_supported).
 * dgrad = vocr_conv3x3_wino_fwd(dy, wpack_dgrad, NULL, dx, n, cout, h, w, cin).  Tensors below 2^29 elements. */
int vocr_conv3x3_wino_supported(int cin, int cout);
size_t vocr_conv3x3_wino_pack_floats(int cout, int cin);
int vocr_conv3x3_wino_pack_weights(const float* w, float* wpack_fwd, float* wpack_dgrad, int cout, int cin, void* stream);
int vocr_conv3x3_wino_fwd(const float* x, const float* wpack, const float* bias, float* y,
                          int n, int cin, int h, int w, int cout, void* stream);
/* Weight gradient with the transposed minimal-filtering transform F(3,2) along the row and - when cin * cout is a multiple of 4 -
 * across row pairs as well (16 multiplications per 2 x 2 block of gradient pixels and filter instead of 36; otherwise along the row
 * only: 4 per column pair instead of 6): same contract as vocr_conv3x3_wgrad, needs cin >= 4; the workspace (ask
 * vocr_conv3x3_wgrad_wino_workspace_bytes: it differs between the two forms) holds the split partial planes that a second kernel
 * adds in a fixed order - the result is bitwise reproducible. */
size_t vocr_conv3x3_wgrad_wino_workspace_bytes(int n, int cin, int h, int w, int cout);
int vocr_conv3x3_wgrad_wino(const float* x, const float* dy, float* dw, void* workspace,
                            int n, int cin, int h, int w, int cout, void* stream);
/* Round 5: one input channel (the first layer of a grey-line model, configs[4]'s rapid_ds stage) as plain vector arithmetic - the layer is
 * bound by writing its output once, not by 576 FLOP per pixel.  w = the layer's own weight [cout][1][3][3] (no pack), y fp32
 * [n][cout][h][wd] (+ bias); round_f16 != 0 rounds x and w to fp16 first (the fp16-operand configuration's arithmetic; fp32 accumulate
 * either way).  vocr_conv3x3_c1_wgrad: dw[cout][1][3][3] from x [n][1][h][wd] and dy [n][cout][h][wd], exact fp32, split partial sums
 * added in a fixed order (workspace: vocr_conv3x3_c1_wgrad_workspace_bytes); dbias[cout] (may be NULL) = the per-channel sum of dy, from
 * the same pass over dy. */
int vocr_conv3x3_c1_fwd(const float* x, const float* w, const float* bias, float* y, int n, int h, int wd, int cout, int round_f16,
                        void* stream);
size_t vocr_conv3x3_c1_wgrad_workspace_bytes(int n, int h, int cout);
int vocr_conv3x3_c1_wgrad(const float* x, const float* dy, float* dw, float* dbias, void* workspace, int n, int h, int wd, int cout,
                          void* stream);
/* fp16-operand variant (BASELINE config 5: "fp16 conv MFMA", fp32 accumulate): tensors stay fp32 in HBM, operands are
 * rounded to fp16 on the way into the matrix cores (v_mfma_f32_32x32x16_f16).  Weight packs are fp16:
 * fwd  [ceil(cin/16)][9][2][cout][8],  dgrad [ceil(cout/16)][9][2][cin][8] (taps flipped); sizes from *_pack_bytes.
 * dgrad = vocr_conv3x3_f16_fwd(dy, wpack_dgrad, NULL, dx, n, cout, h, w, cin).
 * vocr_conv3x3_wgrad_f16: same contract and workspace as vocr_conv3x3_wgrad, dy and x rounded to fp16 for the MFMA. */
size_t vocr_conv3x3_f16_pack_bytes(int cout, int cin, int dgrad);
int vocr_conv3x3_f16_pack_weights(const float* w, void* wpack_fwd, void* wpack_dgrad, int cout, int cin, void* stream);
int vocr_conv3x3_f16_fwd(const float* x, const void* wpack, const float* bias, float* y,
                         int n, int cin, int h, int w, int cout, void* stream);
int vocr_conv3x3_wgrad_f16(const float* x, const float* dy, float* dw, void* workspace,
                           int n, int cin, int h, int w, int cout, void* stream);
/* Round 5: the same fp16-operand convolution on a channel-blocked fp16 copy of the activation, x16 = [n][cin/16][h][w][16] (cin % 16 == 0;
 * ask vocr_conv3x3_h16_supported): the 16 channels of a block are 32 contiguous bytes per pixel and consecutive pixels follow each
 * other, so the kernel's 16-byte LDS-DMA pieces use whole cache lines.  vocr_f32_to_f16_layouts (below) writes the copy in one HBM-bound
 * pass per use.  vocr_conv3x3_h16_fwd: x16 in that layout, the weight packs of vocr_conv3x3_f16_pack_weights, y fp32 [n][cout][h][w]
 * (+ bias): every operand reaches LDS by DMA and the loop holds nothing but MFMAs and their fragment reads.  Results equal
 * vocr_conv3x3_f16_fwd's bit for bit (same operand rounding, same summation order per output).
 * dgrad = vocr_conv3x3_h16_fwd(dy16, wpack_dgrad, NULL, dx, n, cout, h, w, cin). */
int vocr_conv3x3_h16_supported(int cin, int cout);
/* which forward / data-gradient kernel vocr_conv3x3_h16_fwd takes for a launch (0: shape not taken; 1 = 64-channel tile of 16 rows x 32
 * pixels; 2 / 3 / 4 = 128 channels x 8 / 12 / 16 rows x 32 pixels; 5 = 128 channels x 8 rows x 64 pixels): the tests assert that every
 * variant a BASELINE configuration runs is seen by them */
int vocr_conv3x3_h16_plan(int n, int cin, int h, int w, int cout);
int vocr_conv3x3_h16_fwd(const void* x16, const void* wpack, const float* bias, float* y,
                         int n, int cin, int h, int w, int cout, void* stream);
/* Round 5, weight gradient with fp16 operands: both operands as channel-major fp16 copies [n][c][h][WP] whose rows are zero-padded to
 * WP = vocr_f16_padded_row(w) = ceil8(w) + 8 (8 consecutive pixels of a channel = 16 bytes = what a lane half of the MFMA needs when the
 * contraction runs over pixels; the zero tail is every row's right AND the next row's left halo).  vocr_f32_to_f16_layouts writes the
 * channel-blocked copy [n][c/16][h][w][16] (`nhwc`, may be NULL; needs c % 16 == 0) and / or the padded channel-major one (`nchwp`, may
 * be NULL) in ONE pass over the fp32 tensor (c % 8 == 0).
 * vocr_conv3x3_wgrad_h16: x16p [n][cin][h][WP], dy16p [n][cout][h][WP] -> dw[cout][cin][3][3] fp32; cin % 64 == 0 and cout 64 or a
 * multiple of 128 (ask vocr_conv3x3_wgrad_h16_supported); workspace = split slabs added in a fixed order (bitwise reproducible).
 * Same arithmetic as vocr_conv3x3_wgrad_f16 (operands rounded to fp16, fp32 accumulate), another summation order. */
int vocr_f16_padded_row(int w);
int vocr_f32_to_f16_layouts(const float* x, void* nhwc, void* nchwp, int n, int c, int h, int w, void* stream);
int vocr_conv3x3_wgrad_h16_supported(int cin, int cout);
size_t vocr_conv3x3_wgrad_h16_workspace_bytes(int n, int cin, int h, int w, int cout);
int vocr_conv3x3_wgrad_h16(const void* x16p, const void* dy16p, float* dw, void* workspace,
                           int n, int cin, int h, int w, int cout, void* stream);
/* per-channel sum over (n,h,w): conv bias gradient.  out[c].  workspace (vocr_channel_sum_workspace_bytes, may be NULL =
 * one workgroup per channel) holds per-chunk partial sums that are added in a fixed order: bitwise reproducible. */
size_t vocr_channel_sum_workspace_bytes(int n, int c, int hw);
int vocr_channel_sum(const float* x, float* out, int n, int c, int hw, void* workspace, void* stream);

/* ---- BatchNorm2d + ReLU — src/models/cnnlstm.py:265-266 -------------------------------------------------- */
/* training statistics: mean[c], invstd[c] (biased var, eps) and running-stat update (momentum, unbiased var);
 * *num_batches_tracked (device int64) += 1; xhat_sum[c] = sum over the batch of (y-mean)*invstd (the rounding residue of the
 * fp32 mean; the backward turns it into the conv-bias gradient).  running_mean/var, num_batches_tracked, xhat_sum may be NULL.
 * workspace: 2*c*nchunk doubles, see vocr_bn_workspace_bytes. */
size_t vocr_bn_workspace_bytes(int n, int c, int hw);
int vocr_bn_train_stats(const float* y, int n, int c, int hw, float eps, float momentum,
                        float* mean, float* invstd, float* running_mean, float* running_var,
                        int64_t* num_batches_tracked, float* xhat_sum, void* workspace, void* stream);
/* eval: mean/invstd from running stats */
int vocr_bn_eval_stats(const float* running_mean, const float* running_var, int c, float eps,
                       float* mean, float* invstd, void* stream);
/* out = relu((y-mean)*invstd*gamma + beta) */
int vocr_bn_relu_apply(const float* y, const float* mean, const float* invstd, const float* gamma,
                       const float* beta, float* out, int n, int c, int hw, void* stream);
/* Training-mode BatchNorm2d + ReLU of a layer in TWO launches instead of three (round 4): the statistics pass leaves per-chunk
 * partial sums in `workspace` (vocr_bn_workspace_bytes) and every workgroup of the apply pass adds its channel's chunks itself, in
 * chunk order - bit-identical to vocr_bn_train_stats followed by vocr_bn_relu_apply (src/models/cnnlstm.py:265-266), without the
 * "final" launch and its two kernel boundaries on the forward's critical path.  mean / invstd / xhat_sum (saved for the backward)
 * and the running statistics / num_batches_tracked are written as vocr_bn_train_stats writes them. */
int vocr_bn_train_relu_apply(const float* y, const float* gamma, const float* beta, float* out, int n, int c, int hw, float eps,
                             float momentum, float* mean, float* invstd, float* running_mean, float* running_var,
                             int64_t* num_batches_tracked, float* xhat_sum, void* workspace, void* stream);
/* The same for a layer followed by FractionalMaxPool2d (vocr_bn_relu_fracpool2x2_fwd's pass, src/models/cnnlstm.py:127,130). */
int vocr_bn_train_relu_fracpool2x2_fwd(const float* y, const float* gamma, const float* beta, const float* samples, float* out,
                                       int32_t* idx, int n, int c, int h, int w, int oh, int ow, float eps, float momentum,
                                       float* mean, float* invstd, float* running_mean, float* running_var,
                                       int64_t* num_batches_tracked, float* xhat_sum, void* workspace, void* stream);
/* given da = dL/d(out): dgamma, dbeta and dy = dL/d(y) (training-mode batch-stat backward through ReLU);
 * dconv_bias (may be NULL) receives sum_{n,h,w} dy per channel = the gradient of the conv bias in front (rounding
 * noise around zero, from the fixed-order reduction: no float atomics anywhere in this call). */
int vocr_bn_relu_bwd(const float* da, const float* y, const float* mean, const float* invstd,
                     const float* gamma, const float* beta, const float* xhat_sum, float* dy, float* dgamma, float* dbeta,
                     float* dconv_bias, int n, int c, int hw, void* workspace, void* stream);

/* ---- FractionalMaxPool2d(2, output_ratio=(0.5,0.7)) — src/models/cnnlstm.py:127,130 --------------------- */
/* samples[n][c][2] (u_w, u_h) as ATen's _random_samples; idx = flat h*W+w of the winner (int32) */
int vocr_fracpool2x2_fwd(const float* x, const float* samples, float* out, int32_t* idx,
                         int n, int c, int h, int w, int oh, int ow, void* stream);
/* BatchNorm-apply + ReLU + the same pooling in one pass over the conv output y (vocr_bn_relu_apply followed by
 * vocr_fracpool2x2_fwd, bit for bit), for layers whose activation is only ever consumed by the pool. */
int vocr_bn_relu_fracpool2x2_fwd(const float* y, const float* mean, const float* invstd, const float* gamma,
                                 const float* beta, const float* samples, float* out, int32_t* idx,
                                 int n, int c, int h, int w, int oh, int ow, void* stream);
/* Backward of vocr_bn_relu_fracpool2x2_fwd in one go: dout/idx are the pooled gradient and winner indices, samples the
 * forward's; returns dy = dL/d(conv output), dgamma, dbeta, dconv_bias like vocr_bn_relu_bwd, without materialising the
 * un-pooled gradient.  Needs window starts that strictly increase (vocr_bn_relu_fracpool2x2_bwd_supported: both
 * (in-2)/(out-1) >= 1.01, true for the reference's 0.5 x 0.7 ratios); otherwise use vocr_fracpool2x2_bwd + vocr_bn_relu_bwd. */
int vocr_bn_relu_fracpool2x2_bwd_supported(int h, int w, int oh, int ow);
int vocr_bn_relu_fracpool2x2_bwd(const float* dout, const int32_t* idx, const float* samples, const float* y,
                                 const float* mean, const float* invstd, const float* gamma, const float* beta,
                                 const float* xhat_sum, float* dy, float* dgamma, float* dbeta, float* dconv_bias,
                                 int n, int c, int h, int w, int oh, int ow, void* workspace, void* stream);
/* dx[n][c][h][w] = sum of dout over the windows whose winner is that pixel; every element of dx is written
 * (no zero-fill needed) */
int vocr_fracpool2x2_bwd(const float* dout, const int32_t* idx, float* dx,
                         int n, int c, int h, int w, int oh, int ow, void* stream);

/* ---- rapid_ds tail: ReLU + MaxPool2d(2,2) — src/models/cnnlstm.py:119-120 -------------------------------- */
int vocr_relu_maxpool2_fwd(const float* x, float* out, int32_t* idx, int n, int c, int h, int w, void* stream);
int vocr_relu_maxpool2_bwd(const float* dout, const float* out, const int32_t* idx, float* dx,
                           int n, int c, int h, int w, void* stream);

/* ---- dense layers: bridge Linear+ReLU, LSTM input projections, prob Linear — cnnlstm.py:143-154,278,294 -- */
/* C[M,N] (ldc) = op(A)[M,K] * op(B)[K,N] (+ bias[N]) (relu) — row-major.
 * transa=0: A is [M,K] lda;  transa=1: A is stored [K,M] lda.   transb=0: B is [K,N] ldb;  transb=1: B is [N,K] ldb.
 * accumulate!=0: C += result.  Long-K products with few output tiles (the weight gradients) are split along K: each
 * slice writes its own [M][N] slab into `workspace` (vocr_gemm_workspace_bytes; 16-byte aligned; may be NULL or smaller
 * = fewer or no slices) and the slabs are added in slice order — results are bitwise reproducible run to run. */
size_t vocr_gemm_workspace_bytes(int m, int n, int k, int has_bias_or_relu);
int vocr_gemm(int transa, int transb, int m, int n, int k,
              const float* a, int lda, const float* b, int ldb, float* c, int ldc,
              const float* bias, int relu, int accumulate, void* workspace, size_t workspace_bytes, void* stream);
/* Two products of ONE shape in one launch — the two directions of a bidirectional nn.LSTM layer (cnnlstm.py:148-149,288-290):
 *   mode 0: c0 = op(a0) op(b0) (+bias0)(relu),  c1 = op(a1) op(b1) (+bias1)(relu)      (x W_ih^T of both directions: a0 == a1;
 *           the weight gradients dgates_dir^T x and dgates_dir^T h of both directions)
 *   mode 1: c0 = op(a0) op(b0) + op(a1) op(b1) (+bias0)(relu), c1 = bias1 = NULL        (dx = dgates_fwd W_ih_fwd + dgates_rev W_ih_rev:
 *           one product with the K dimension running through both operands pairs instead of a product and an accumulating one)
 * Same operand conventions as vocr_gemm; all leading dimensions are shared by the two products.  Large 16-byte aligned shapes run
 * as ONE launch of the panel kernel whose workgroups cover both products (no partial last round); anything else is two vocr_gemm
 * calls.  workspace: vocr_gemm_pair_workspace_bytes (K slabs, fixed-order sum: reproducible). */
size_t vocr_gemm_pair_workspace_bytes(int m, int n, int k, int mode);
int vocr_gemm_pair(int mode, int transa, int transb, int m, int n, int k, const float* a0, const float* a1, int lda,
                   const float* b0, const float* b1, int ldb, float* c0, float* c1, int ldc, const float* bias0, const float* bias1,
                   int relu, void* workspace, size_t workspace_bytes, void* stream);
/* out[N] = sum over M rows of x[M,N] (bias gradients); partial rows in `workspace` (vocr_colsum_workspace_bytes, may be
 * NULL = one workgroup per 64 columns), added in a fixed order */
size_t vocr_colsum_workspace_bytes(int m, int n);
int vocr_colsum(const float* x, float* out, int m, int n, void* workspace, void* stream);
/* dz = (out > 0) ? dy : 0  (ReLU backward, elementwise) */
int vocr_relu_bwd(const float* dy, const float* out, float* dz, size_t count, void* stream);
/* bchw -> (w,b,c*h) and back — cnn_output.permute(3,0,1,2).contiguous(), cnnlstm.py:276 */
int vocr_bchw_to_wbch(const float* x, float* out, int b, int c, int h, int w, void* stream);
int vocr_wbch_to_bchw(const float* x, float* out, int b, int c, int h, int w, void* stream);
/* out = x * mask (inter-layer LSTM dropout with an explicit, pre-scaled mask) */
int vocr_mul(const float* x, const float* mask, float* out, size_t count, void* stream);
/* out = x + y ; out = x * (*scalar) with the scalar read from device memory (no host sync) */
int vocr_add(const float* x, const float* y, float* out, size_t count, void* stream);
int vocr_scale_dev(const float* x, const float* scalar, float* out, size_t count, void* stream);
/* mask[i] = (hash(seed,i) >= p) ? 1/(1-p) : 0 ; out = x*mask */
int vocr_dropout_fwd(const float* x, float* out, float* mask, size_t count, float p, uint64_t seed, void* stream);
/* the mask of vocr_dropout_fwd alone (it depends on the seed and the element index only): lets a consumer that applies the mask itself
 * (vocr_lstm_xproj_follow) have it before x exists */
int vocr_dropout_mask(float* mask, size_t count, float p, uint64_t seed, void* stream);

/* ---- nn.LSTM recurrence on a packed, length-sorted batch — src/models/cnnlstm.py:148-149,288-290 ---------- */
/* One direction pair of one layer.  xproj[dir][T][B][4H] = x W_ih^T + b_ih + b_hh (gate order i,f,g,o),
 * whh_fwd / whh_rev [4H][H] (the two directions' weight_hh), lens[B] int32 (device, descending).  Outputs y[T][B][2H] (zeros past lens),
 * gates[dir][T][B][H][4] (post-activation i,f,g,o interleaved per unit, 16-byte aligned) and cell[dir][T][B][H]
 * for backward.  h_{t-1}/c_{t-1} are read back from y/cell, so the sweep keeps no separate state.
 * workspace: vocr_lstm_workspace_bytes (256-byte aligned device memory; contents need not be preserved between calls, except
 * between the vocr_lstm_fwd_range calls of ONE sweep: the forward sweep's hand-off buffer lives there).
 * Supports B <= 64 per call, H % 16 == 0 (a larger batch is several calls on batch tiles: the recurrence never couples batch rows and
 * a tile of a length-sorted batch is itself a valid packed batch - vistaocr_amd.CnnOcrModel does that).  For H in {64,128,256,512} a whole sweep is ONE persistent launch whose workgroups
 * hand h_t (forward) / partial sums (backward) to each other through memory: it needs its grid (<= two workgroups per CU)
 * co-resident, so do not run two sweeps concurrently on one device; a hand-off that times out (seconds) poisons the
 * output with NaN instead of hanging.  Environment: VOCR_LSTM_SWEEP=step selects one launch per time step instead (several processes
 * sharing one GPU; VOCR_LSTM_PERSISTENT=0 is accepted as the old spelling), VOCR_LSTM_SWEEP=wide4|chain4|chain16 starts the kernel
 * search at that kind (A/B, tests), an unknown value makes the call fail with VOCR_EINVAL; VOCR_LSTM_WRITE_THROUGH=1 forces the
 * hand-off mode of a chain whose workgroups are spread over several XCDs.  These are the library's only runtime switches: the
 * kernel-variant experiments of scripts/ exist only in -DVOCR_EXPERIMENTS builds.
 * `health` (may be NULL): see the note on health words at the top; the timeout flag the sweep itself tests lives in the workspace
 * (an implementation detail: callers watch `health`). */
size_t vocr_lstm_workspace_bytes(int t, int b, int h);
int vocr_lstm_fwd(const float* xproj, const float* whh_fwd, const float* whh_rev, const int32_t* lens, float* y,
                  float* gates, float* cell, void* workspace, int t, int b, int h, int32_t* health, void* stream);
/* Steps [step_begin, step_end) of the same sweep (step s is time s for the forward direction, T-1-s for the reverse
 * one): a later range resumes from the h (in y) and c (in cell) an earlier call left.  Lets the host overlap the
 * x-projection GEMM of the later time steps with the first half of the sweep.  vocr_lstm_fwd == range [0, t). */
int vocr_lstm_fwd_range(const float* xproj, const float* whh_fwd, const float* whh_rev, const int32_t* lens, float* y,
                        float* gates, float* cell, void* workspace, int t, int b, int h, int step_begin, int step_end,
                        int32_t* health, void* stream);
/* dy[T][B][2H] -> dgates[dir][T][B][4H] (gradient w.r.t. pre-activation gates = w.r.t. xproj).
 * whht_* are the TRANSPOSED recurrent weights [H][4H] (vocr_bchw_to_wbch with b=1 transposes a matrix). */
int vocr_lstm_bwd(const float* dy, const float* whht_fwd, const float* whht_rev, const int32_t* lens, const float* gates,
                  const float* cell, float* dgates, void* workspace, int t, int b, int h, int32_t* health, void* stream);
/* Same, and dbias[dir][4H] = sum over (t, b) of dgates[dir] (the gradient of b_ih and of b_hh): accumulated inside the
 * sweep where the kernel supports it, else by column sums afterwards.  dbias may be NULL. */
int vocr_lstm_bwd_bias(const float* dy, const float* whht_fwd, const float* whht_rev, const int32_t* lens, const float* gates,
                       const float* cell, float* dgates, float* dbias, void* workspace, int t, int b, int h, int32_t* health,
                       void* stream);
/* The same sweep for a caller that overlaps: (1) `dy_mask` (NULL: none): dy is multiplied element-wise by it as it is read - the
 * backward of nn.LSTM's inter-layer dropout (src/train_cnn_lstm.py:331: the mask that scaled this layer's output) without a pass of
 * its own; (2) the bias gradient stays as per-chain partial rows in the workspace and vocr_lstm_bias_from_parts finishes it on ANY
 * stream ordered behind this call (the workspace must live until then) - so the data-gradient GEMM that follows the sweep on the
 * critical path does not queue behind a reduction nobody downstream reads.  Only for shapes the self-validating 4-row sweeps cover:
 * ask vocr_lstm_bwd_parts_supported (else use vocr_lstm_bwd_bias). */
int vocr_lstm_bwd_parts_supported(int t, int b, int h);
int vocr_lstm_bwd_parts(const float* dy, const float* dy_mask, const float* whht_fwd, const float* whht_rev, const int32_t* lens,
                        const float* gates, const float* cell, float* dgates, void* workspace, int t, int b, int h,
                        int32_t* health, void* stream);
int vocr_lstm_bias_from_parts(float* dbias, const void* workspace, int t, int b, int h, void* stream);

/* ---- fp32 GEMM on the bf16 matrix pipe by EXACT operand splitting ("bf16x6") — the LSTM projections and their gradients (cnnlstm.py:143-154) ---- */
/* Every fp32 operand element is written as a0 + a1 + a2 (three bf16 values whose sum is the fp32 value exactly) and a product is accumulated in fp32
 * from the six partial products of order <= 2^-16; the three dropped ones are <= 2^-23 |a b| (one unit roundoff of an fp32 product).
 * v_mfma_f32_32x32x16_bf16 runs at 16x the rate of the f32 MFMA: six per product = 2.67x the f32 matrix peak.
 * vocr_gemm_x6_split writes the three planes of an operand in MFMA-fragment order (vocr_gemm_x6_planes_bytes(rows, k) bytes; rows padded to 256, K to
 * 32): x is [rows][k] with leading dimension ld (k_contiguous = 1) or [k][rows] (k_contiguous = 0: the transposed operands of the weight
 * gradients).  vocr_gemm_x6: C[m][n] = A[m][k] . B[n][k]^T (+ bias)(relu) from the planes of A and B, both outputs with ldc.  What is left of the
 * last round of 256 x 128 tiles (one workgroup per CU) is cut along K into slabs in `workspace` (vocr_gemm_x6_workspace_bytes; NULL: no cut)
 * that are added in a fixed order. */
size_t vocr_gemm_x6_planes_bytes(int rows, int k);
/* x may come in two pieces (the two direction planes of the LSTM's gate tensors, the two directions' weights): seg_axis = 0: k < seg from x, the
 * rest from x2 at k - seg; seg_axis = 1: row < seg from x, the rest from x2 at row - seg (seg <= 0: one piece; seg %% 8 == 0; both pieces with ld).
 * mask (K-contiguous only, may be NULL): an element-wise factor with x's addressing (the inter-layer dropout mask rides on the operand's read). */
int vocr_gemm_x6_split(const float* x, const float* x2, int seg, int seg_axis, const float* mask, long ld, int rows, int k, int k_contiguous,
                       void* planes, void* stream);
size_t vocr_gemm_x6_workspace_bytes(int m, int n, int k);
/* A = rows a_row0 .. + m and k16 steps a_kk0 .. + k/16 of a plane set written for (a_rows, a_k) - a view: the row-shifted operands of the recurrent
 * weight gradient are k windows of ONE plane set -; B likewise (its rows are the output columns).  Two outputs: columns >= csplit go to c1 at
 * col - csplit, or rows >= rsplit to c1 at row - rsplit (<= 0: no cut). */
int vocr_gemm_x6(const void* a_planes, int a_rows, int a_k, int a_row0, int a_kk0, const void* b_planes, int b_rows, int b_k, int b_row0, int b_kk0,
                 int m, int n, int k, float* c0, float* c1, int csplit, int rsplit, int ldc, const float* bias0, const float* bias1, int relu,
                 void* workspace, void* stream);

/* Two products of one shape from ONE pair of plane sets in one launch: rows < rsplit (a multiple of 256) of the m x n result read the views
 * (a_row0, a_kk0 | b_row0, b_kk0) and go to c0; rows >= rsplit read (a_row0 + rsplit .., a_kk0_2 | b_row0_2, b_kk0_2) and go to c1 at row - rsplit.
 * The recurrent weight gradients of both directions - each a time-shifted k window of the gate gradients' and the layer outputs' transposed planes -
 * with half the K-cut slabs of two launches. */
int vocr_gemm_x6_two_views(const void* a_planes, int a_rows, int a_k, int a_row0, const void* b_planes, int b_rows, int b_k, int m, int n, int k,
                           int rsplit, int a_kk0, int b_row0, int b_kk0, int a_kk0_2, int b_row0_2, int b_kk0_2, float* c0, float* c1, int ldc,
                           void* workspace, void* stream);

/* ---- the same products from TWO fp16 planes per operand and three partial products ("fp16x3") - OPT-IN, not what CnnOcrModel runs by default ---- */
/* a s = h0 + h1 with h0 = fp16(a s), h1 = fp16(a s - h0), s = the power of two that brings the ROW's largest magnitude into [2^14, 2^15): 22 - 23 of
 * fp32's 24 significant bits per element (elements more than 2^17 below their row's maximum lose low bits to fp16's exponent range: absolute error
 * <= 2^-40 of the row's maximum); a b ~ h0 g1 + h1 g0 + h0 g0, the dropped term <= 2^-24 |a b|; fp32 accumulate; both rows' scales undone exactly in
 * the epilogue.  Half the matrix instructions and 2/3 of the operand bytes of bf16x6 for a NORM-WISE error bound (<= 2^-22 sum |a||b| + K 2^-39
 * max|a| max|b| per dot product) instead of bf16x6's per-product one: measured against fp64 it stays below the f32-MFMA kernels' own error on the
 * training step's shapes (tests/test_x6_gpu.py), but it is an approximation of the operands, which bf16x6 is not.  Same contracts as the three entry
 * points above; a plane set is vocr_gemm_h3_planes_bytes(rows, k) bytes (the rows' maxima ride behind the planes) and is not interchangeable with a
 * bf16x6 one.  The split takes the maxima in a pass of its own (order-free integer maximum: deterministic). */
size_t vocr_gemm_h3_planes_bytes(int rows, int k);
/* bound > 0: the caller's bound on every |element| (after the mask) stands in for the rows' maxima - no pass over the source for them (an LSTM
 * output lies inside (-1, 1)); an element above the bound overflows fp16 and poisons its row, as an Inf in the source would.  bound = 0: measured. */
int vocr_gemm_h3_split(const float* x, const float* x2, int seg, int seg_axis, const float* mask, long ld, int rows, int k, int k_contiguous,
                       float bound, void* planes, void* stream);
int vocr_gemm_h3(const void* a_planes, int a_rows, int a_k, int a_row0, int a_kk0, const void* b_planes, int b_rows, int b_k, int b_row0, int b_kk0,
                 int m, int n, int k, float* c0, float* c1, int csplit, int rsplit, int ldc, const float* bias0, const float* bias1, int relu,
                 void* workspace, void* stream);
int vocr_gemm_h3_two_views(const void* a_planes, int a_rows, int a_k, int a_row0, const void* b_planes, int b_rows, int b_k, int m, int n, int k,
                           int rsplit, int a_kk0, int b_row0, int b_kk0, int a_kk0_2, int b_row0_2, int b_kk0_2, float* c0, float* c1, int ldc,
                           void* workspace, void* stream);

/* ---- the same recurrence without the padding: pack_padded_sequence's economy — src/models/cnnlstm.py:285-290 ---------- */
/* The reference packs the length-sorted batch before nn.LSTM, so cuDNN never computes a padded frame.  Here the rows of every
 * sequence-side tensor (bridge output, xproj, y, gates, cell, dy, dgates) can be kept in a PACKED, chain-major order instead of
 * [T][B]: batch rows are taken in chains of 4 (4c .. 4c+3; sorted lengths make lens[4c] the chain's longest, L_c); a GROUP is one
 * time step of one chain = 4 consecutive rows; the groups are [zero][chain 0: L_0 groups, t ascending][zero][chain 1: L_1]...[zero]:
 *     rows = 4 * (sum_c L_c + chains + 1),      row(t, b) = 4 * (1 + c + sum_{c' < c} L_c' + t) + (b & 3)   for t < L_c, c = b >> 2.
 * All GEMMs of the LSTM stack then run over `rows` rows instead of T*B, the recurrent weight gradient is still ONE product of
 * row-shifted views (shift 4: the all-zero group in front of / behind a chain is its h_{-1} / h_{L_c}), and a chain's workgroups leave
 * a sweep after their own L_c steps.  Contract: the caller zero-fills y before vocr_lstm_fwd_packed and dgates before
 * vocr_lstm_bwd_packed (the zero groups, a last chain's rows >= B and nothing else stay untouched by the sweeps); rows of a chain past
 * their own length are written as zeros like in the dense layout.  Results per frame are those of the dense entry points (same
 * kernels, same summation orders).  Shapes: ask vocr_lstm_packed_supported (B <= 32, H in {256, 512} on a device that can hold the
 * persistent sweep's grid) and keep rows * 2H * 4 below 2 GB (the sweeps' 32-bit offsets); otherwise use the dense entry points above.
 * vocr_seq_rowmap builds both index maps on the device from lens: to_packed[T*B] (packed row of frame (t, b), -1: none) and
 * to_dense[rows] (t*B + b of a packed row, -1: zero group / beyond B); either may be NULL.  vocr_gather_rows moves rows through such a
 * map: dst[r][0..n) = src[map[r]][0..n), or fill[0..n) (NULL: zeros) where map[r] < 0 (16-byte pieces when n % 4 == 0 and src, dst, fill
 * are 16-byte aligned, element by element otherwise) — packing (map = to_dense, nrows = rows) and
 * unpacking (map = to_packed, nrows = T*B; fill = the output layer's bias reproduces the reference's logits of a padded frame). */
int vocr_lstm_packed_supported(int b, int h);
int vocr_seq_rowmap(const int32_t* lens, int t, int b, int rows, int32_t* to_packed, int32_t* to_dense, void* stream);
int vocr_gather_rows(const float* src, float* dst, const int32_t* map, long nrows, int n, const float* fill, void* stream);
/* gradient of vocr_gather_rows' fill row: dfill[0..n) = sum of dout[r][0..n) over the rows with map[r] < 0 (fixed summation order);
 * workspace: vocr_gather_rows_fill_grad_workspace_bytes(n) */
size_t vocr_gather_rows_fill_grad_workspace_bytes(int n);
int vocr_gather_rows_fill_grad(const float* dout, const int32_t* map, long nrows, int n, float* dfill, void* workspace, void* stream);
int vocr_lstm_fwd_packed(const float* xproj, const float* whh_fwd, const float* whh_rev, const int32_t* lens, float* y,
                         float* gates, float* cell, void* workspace, int t, int b, int h, int rows, int32_t* health, void* stream);
/* dy_mask (may be NULL) as in vocr_lstm_bwd_parts; dbias (may be NULL) as in vocr_lstm_bwd_bias */
int vocr_lstm_bwd_packed(const float* dy, const float* dy_mask, const float* whht_fwd, const float* whht_rev, const int32_t* lens,
                         const float* gates, const float* cell, float* dgates, float* dbias, void* workspace, int t, int b, int h,
                         int rows, int32_t* health, void* stream);

/* ---- the next layer's x-projection BEHIND the forward sweep — src/models/cnnlstm.py:148-149,288-290 (cuDNN's multi-layer RNN ---------
 * overlaps layer l+1's input GEMM with layer l's recurrence; nn.LSTM's inter-layer dropout: src/train_cnn_lstm.py:331) */
/* A forward sweep is latency-bound and leaves most of every CU's matrix pipe idle.  vocr_lstm_fwd_lead is vocr_lstm_fwd (rows = 0) /
 * vocr_lstm_fwd_packed (rows > 0) with two additions:
 *  - xproj2 (may be NULL): a second addend of this layer's x-projection, same layout as xproj ([2][rows][4H]); the sweep reads
 *    xproj + xproj2.
 *  - next_wpack != NULL: four more waves in every workgroup of the sweep compute the NEXT layer's x-projection from the rows the sweep
 *    has produced so far (16 steps of a chain at a time, throttled by the chain so that they use the matrix pipe while it waits for its
 *    hand-off): the chain's OWN direction half of y (x y_mask, the pre-scaled inter-layer dropout mask [rows][2H], or NULL), K = H, by
 *    the matching half of W_ih of both directions of the next layer, so neither direction waits for the other:
 *        next_planes[src][tgt][row][4H] = y[row][src*H .. src*H+H) . W_ih(next, tgt)[:, src*H .. src*H+H)^T  (+ next_bias[tgt][4H] in src = 0)
 *    and the next layer's vocr_lstm_fwd_lead takes xproj = next_planes (planes [0][.]), xproj2 = next_planes + 2*rows*4H.
 *    next_wpack: W_ih of the next layer, both directions, in MFMA-fragment order: vocr_lstm_xproj_pack_bytes(h) bytes written by
 *    vocr_lstm_xproj_pack whenever the weights change; next_bias [2][4H] may be NULL.
 * Shapes: vocr_lstm_follow_supported (16 < B <= 32, H = 512 = the next layer's H, the wide 4-row chain sweep); rows < 262144. */
int vocr_lstm_follow_supported(int b, int h);
size_t vocr_lstm_xproj_pack_bytes(int h);
int vocr_lstm_xproj_pack(const float* w_ih_fwd, const float* w_ih_rev, float* wpack, int h, void* stream);
int vocr_lstm_fwd_lead(const float* xproj, const float* xproj2, const float* whh_fwd, const float* whh_rev, const int32_t* lens,
                       float* y, float* gates, float* cell, void* workspace, int t, int b, int h, int rows,
                       const float* next_wpack, const float* next_bias, const float* y_mask, float* next_planes,
                       int32_t* health, void* stream);

/* ---- CTC: warpctc_pytorch.CTCLoss — src/train_cnn_lstm.py:358,138 ------------------------------------------ */
/* logits[T][B][V] pre-softmax, blank = 0.  labels flat int32 (device), label_offsets[B], label_lens[B],
 * act_lens[B] (device int32).  nll[B] = -log p(labels|x); dlogits[T][B][V] = d(sum_b nll)/dlogits
 * (zeros for t >= act_lens[b]).  max_label_len = max(label_lens) (host value, sizes the workspace). */
size_t vocr_ctc_workspace_bytes(int t, int b, int v, int max_label_len);
int vocr_ctc_loss_grad(const float* logits, const int32_t* labels, const int32_t* label_offsets,
                       const int32_t* label_lens, const int32_t* act_lens, float* nll, float* dlogits,
                       void* workspace, int t, int b, int v, int max_label_len, void* stream);

/* ---- greedy decode: decode_without_lm / ArgmaxDecoder — cnnlstm.py:479-541, decoder.py:116-185 ------------ */
/* per row of x[rows][v]: first index of the maximum and the maximum */
int vocr_argmax_rows(const float* x, int32_t* idx, float* maxv, int rows, int v, void* stream);
/* CTC best-path collapse on device.  canon[v] = smallest index with the same alphabet string (string
 * comparison of the reference), thresh = 3/len(alphabet) applied to the raw maximum.
 * out_labels[B][T], out_counts[B]. */
int vocr_greedy_collapse(const int32_t* idx, const float* maxv, const int32_t* lens, const int32_t* canon,
                         int32_t* out_labels, int32_t* out_counts, int t, int b, float thresh, void* stream);

/* ---- optimiser: grad clamp + torch.optim.Adam — src/train_cnn_lstm.py:143-149,363 -------------------------- */
/* g = clamp(g*grad_scale, -clamp, clamp) (+ wd*p); Adam(m, v); step is the 1-based step count.  A NaN gradient stays NaN
 * (torch's clamp_ propagates NaN) and sets health[1] (health may be NULL). */
int vocr_clamp_adam(float* p, const float* g, float* m, float* v, size_t count, float lr, float beta1, float beta2,
                    float eps, float weight_decay, float clamp, float grad_scale, int step, int32_t* health, void* stream);
/* in place x = clamp(x, -clamp, clamp), NaN preserved: `param.grad.data.clamp_(min=-5, max=5)` (train_cnn_lstm.py:143-145)
 * for callers that keep their own optimiser (torch.optim.Adam). */
int vocr_clamp(float* x, size_t count, float clamp, int32_t* health, void* stream);

/* ---- data-parallel exchange: nn.DataParallel's gradient reduction — cnnlstm.py:198-199, train_cnn_lstm.py:333 ---- */
/* One process per GPU; the only collective of the path is a SUM all-reduce of the flat fp32 gradient before the clamp.
 * Thin RCCL wrapper (xGMI inside a node) for hosts that do not use torch.distributed; RCCL is dlopen'ed on first use.
 * Rank 0 calls vocr_comm_unique_id and hands the 128 bytes to every rank (any transport); every rank then calls
 * vocr_comm_create(device = its GPU).  The handle is the only object the library ever allocates. */
typedef struct vocr_comm vocr_comm;
int vocr_comm_unique_id(void* id128);
int vocr_comm_create(vocr_comm** out, const void* id128, int nranks, int rank, int device);
int vocr_allreduce_sum_f32(vocr_comm* comm, float* buf, size_t count, void* stream);   /* in place, stream-ordered */
int vocr_comm_destroy(vocr_comm* comm);

#ifdef __cplusplus
}
#endif
#endif
