/*
 * C ABI of libaq_hip.so: the MI355X (gfx950) kernels behind the reference's quantized
 * layers.  The reference is pure Python on PyTorch and has no FFI of its own; each entry point
 * below names the reference code whose arithmetic it replaces.  Conventions:
 *   - every pointer is a DEVICE pointer to contiguous fp32 (unless typed otherwise) owned by the
 *     caller; the library never allocates, frees or synchronises; scratch memory is the
 *     caller-provided `workspace`;
 *   - kernels are enqueued on `stream` (a hipStream_t passed as void*; NULL = default stream);
 *   - return value 0 on success, a negative errno-style code on failure, with a message available
 *     from aq_last_error() (thread-local);
 *   - calls on different streams may run concurrently as long as their buffers are distinct.
 */
#ifndef AQ_HIP_H
#define AQ_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AQ_ABI_VERSION 13
#define AQ_MAX_LEVELS 128     /* largest supported number of quantisation levels L */
#define AQ_MAX_ABS_BETA 64.0f /* |beta| above this is rejected (exp() range of the kernels) */
/* flags of aq_synth_fwd / aq_synth_bwd */
#define AQ_FLAG_EXACT 1u      /* bit-reproducible arithmetic (polynomial exp): results equal the CPU
                                 restatement of the same operation sequence bit for bit.  Without it exp
                                 runs on the transcendental unit (faster; same results within ~1e-6).
                                 Launches that emit codes (idx / W_hard) always use the exact sequence. */

/* One quantized layer ("segment") of a multi-layer launch.  rows = number of weights of the layer
 * (Cout*Cin*k*k for MyConv2d.p_c, in*out for MyLinear.p_c), L = number of levels (innermost,
 * contiguous axis of p_c).  Unused outputs may be NULL. */
typedef struct aq_seg {
    const float *p_c;   /* [rows, L] primary coefficients                                    */
    const float *q;     /* [L]       quantisation levels                                     */
    int64_t rows;
    int32_t L;
    int32_t reserved;
    /* forward outputs */
    float *W;           /* [rows] soft weight  sum_l s_c[l]*q[l]      (fan.py:563-568)       */
    uint8_t *idx;       /* [rows] code = argmax_l s_c[l]              (fan_test.py:304)      */
    float *W_hard;      /* [rows] q[idx]                              (fan_test.py:305)      */
    float *sumsq;       /* [1]    sum_rows W^2, the layer's term of model_loss (fan.py:752)  */
    /* backward inputs / outputs */
    const float *dW;    /* [rows] d loss / d W (NULL = zero)                                 */
    const float *dsumsq;/* [1]    d loss / d sumsq on the device (NULL = zero); the kernel adds
                                  2*dsumsq[0] to wd, so no host read-back is needed            */
    float *dp_c;        /* [rows, L] d loss / d p_c                                          */
    float *dq;          /* [L]    d loss / d q_level (NULL for frozen levels, fix.py:652)    */
    /* state of torch.optim.Adam for p_c (aq_synth_bwd_adam only; same tensors as optimizer.state[p_c]) */
    float *adam_m;      /* [rows, L] exp_avg                                                 */
    float *adam_v;      /* [rows, L] exp_avg_sq                                              */
    float *adam_step;   /* [1]    step count as fp32 (capturable Adam keeps it on the device) */
} aq_seg;

int aq_abi_version(void);
/* Name, with its template arguments, of the main kernel that the most recent aq_synth_* / aq_tern_* call of this thread
 * launched ("" before the first): what a benchmark should print next to a timing instead of a literal. */
const char *aq_last_kernel(void);
const char *aq_last_error(void);

/* Weight synthesis of nseg layers in one launch (+ one small reduction launch when any sumsq is
 * requested).  Replaces MyConv2d.forward lines adp_qtz_fan.py:563-571 / MyLinear.forward
 * :634-643 and the per-layer re-synthesis inside model_loss :747-752.
 * workspace must hold aq_synth_workspace_bytes(segs, nseg, 0) bytes. */
int aq_synth_fwd(const aq_seg *segs, int nseg, float beta, unsigned flags, void *workspace,
                 size_t workspace_bytes, void *stream);

/* aq_synth_fwd for callers that cannot allocate device memory (code under HIP graph capture, tight serving
 * loops, harnesses that forbid hipMalloc).  It has no counterpart in the reference.  Same arguments, results
 * and launches as aq_synth_fwd, but the per-wavefront partial sums of sum W^2 go to a workspace inside the
 * library (a device array of AQ_STATIC_SCRATCH_BYTES that every GPU the library runs on holds until the
 * process ends) and the call allocates nothing, on the device or through any library.  Layers whose
 * aq_synth_workspace_bytes(segs, nseg, 0) exceeds AQ_STATIC_SCRATCH_BYTES are rejected with -E2BIG (one layer
 * of up to ~33 M weights fits, ~16 M weights of 32 levels in a call of more than 128 MiB, which may run twice as
 * many wavefronts; calls without any sumsq output need no workspace at all).  All calls on one GPU
 * share that workspace: calls that overlap in time on one GPU (different streams or threads) race on it and
 * give wrong sumsq; nothing in it needs clearing between calls (every partial read is written first). */
#define AQ_STATIC_SCRATCH_BYTES (2u << 20)
int aq_synth_fwd_static(const aq_seg *segs, int nseg, float beta, unsigned flags, void *stream);

/* Gradient of the synthesis: for every layer g = dW + wd*W (wd = 2*l2_reg*dLoss/dReg folds the
 * regulariser of adp_qtz_fan.py:737-754 in), then dp_c and dq as autograd derives them from
 * adp_qtz_fan.py:563-568.  workspace must hold aq_synth_workspace_bytes(segs, nseg, 1) bytes. */
int aq_synth_bwd(const aq_seg *segs, int nseg, float beta, float wd, unsigned flags, void *workspace,
                 size_t workspace_bytes, void *stream);

/* aq_synth_bwd with the Adam update of p_c fused in: replaces `optimizer_prim.step()` for the p_c group
 * (optimizer_prim = optim.Adam(p_c parameters, lr), adp_qtz_fan.py:208-209, 291; default betas / eps,
 * no weight decay, no amsgrad).  Per layer: *adam_step += 1 (once per distinct pointer), then p_c,
 * adam_m, adam_v are updated IN PLACE (p_c is written although declared const in aq_seg) with the
 * gradient that aq_synth_bwd would have stored; dp_c may be NULL (gradient not stored).  dq as in
 * aq_synth_bwd.  At most 24 layers per call.  workspace: aq_synth_workspace_bytes(segs, nseg, 2). */
int aq_synth_bwd_adam(const aq_seg *segs, int nseg, float beta, float wd, double lr, double beta1,
                      double beta2, double eps, unsigned flags, void *workspace, size_t workspace_bytes,
                      void *stream);

/* backward: 0 = aq_synth_fwd, 1 = aq_synth_bwd, 2 = aq_synth_bwd_adam */
size_t aq_synth_workspace_bytes(const aq_seg *segs, int nseg, int backward);

/* 3x3 / pad 1 / stride 1|2 convolution, NCHW fp32, no bias: MyConv2d.forward lines
 * adp_qtz_fan.py:574-601 (== cross-correlation).  W is [Cout, Cin, 3, 3]. */
int aq_conv_fwd(const float *x, const float *W, float *y, int N, int Cin, int H, int Wd, int Cout,
                int ksize, int stride, int pad, void *stream);
/* y = conv(x, W) + res: the convolution and the shortcut addition that follows it in every block
 * (`out = self.conv2(out); out += self.shortcut(x)`, adp_qtz_fan.py:676-677, 691-692).  res has the
 * shape of y and may alias it. */
int aq_conv_fwd_add(const float *x, const float *W, const float *res, float *y, int N, int Cin, int H,
                    int Wd, int Cout, int ksize, int stride, int pad, void *stream);
/* aq_conv_fwd / aq_conv_fwd_add (res may be NULL) that also leaves the channel statistics of y behind for
 * the batch normalisation that consumes y next (`self.bnK(...)` after every convolution,
 * adp_qtz_fan.py:674-675, 690, 726, 730): every workgroup writes (mean, M2) of the pixels it produced, per
 * output channel, into `stats` ([Cout][desc->P] pairs; aq_conv_stats_bytes bytes), and *desc says how the
 * tiles were shared out.  aq_bn_relu_fwd / aq_bn_relu_pool_fwd take the pair (stats, desc) in place of
 * their own statistics pass over y.  aq_conv_stats_bytes == 0: this shape has no such epilogue (-ENOTSUP
 * from aq_conv_fwd_stats); use aq_conv_fwd and the plain batch normalisation. */
typedef struct aq_conv_stats {
    int32_t P;       /* partials per channel (workgroups of the convolution) */
    int32_t ntiles;  /* workgroup w produced tiles w, w + P, w + 2P, .. < ntiles */
    int32_t tile_px; /* output pixels per tile */
} aq_conv_stats;
size_t aq_conv_stats_bytes(int N, int Cin, int H, int Wd, int Cout, int ksize, int stride, int pad);
int aq_conv_fwd_stats(const float *x, const float *W, const float *res, float *y, int N, int Cin, int H,
                      int Wd, int Cout, int ksize, int stride, int pad, void *stats, size_t stats_bytes,
                      aq_conv_stats *desc, void *stream);
/* same, weights given as codes + levels (hard mode, adp_qtz_fan_test.py:304-331): W = q[idx] is
 * decoded while the weight tile is staged into LDS. */
int aq_conv_fwd_codes(const float *x, const uint8_t *idx, const float *q, int L, float *y, int N,
                      int Cin, int H, int Wd, int Cout, int ksize, int stride, int pad, void *stream);
/* dx = d loss / d x  (autograd of adp_qtz_fan.py:574-601 w.r.t. the input) */
int aq_conv_dgrad(const float *dy, const float *W, float *dx, int N, int Cin, int H, int Wd,
                  int Cout, int ksize, int stride, int pad, void *stream);
/* dx = aq_conv_dgrad(..) + res: the input gradient of a convolution whose input also feeds an identity
 * shortcut (`out = self.conv2(self.conv1(x)); out += x`, adp_qtz_fan.py:689-692), the gradient arriving over
 * the shortcut added in the epilogue.  res has the shape of dx and may alias it. */
int aq_conv_dgrad_add(const float *dy, const float *W, const float *res, float *dx, int N, int Cin, int H,
                      int Wd, int Cout, int ksize, int stride, int pad, void *stream);
/* aq_conv_dgrad whose input x is z = relu(batch_norm(xb)) (`self.convK(F.relu(self.bnK(x)))`,
 * adp_qtz_fan.py:674-676, 690-691): dz = d loss / d z as aq_conv_dgrad, and the epilogue also reduces what
 * the batch-norm backward of that z needs, per workgroup and channel, into `sums` ([Cin][desc->P] pairs,
 * aq_conv_dgrad_bn_bytes bytes): (sum g, sum g * xhat), g = dz where z > 0 else 0, xhat = (xb - mean) * rstd.
 * xb, gamma, beta, save_mean, save_rstd: as aq_bn_relu_fwd left them.  aq_bn_relu_bwd takes (sums, desc) in
 * place of its own reduction pass.  aq_conv_dgrad_bn_bytes == 0: no such epilogue for this shape. */
size_t aq_conv_dgrad_bn_bytes(int N, int Cin, int H, int Wd, int Cout, int ksize, int stride, int pad);
int aq_conv_dgrad_bn(const float *dy, const float *W, float *dz, int N, int Cin, int H, int Wd, int Cout,
                     int ksize, int stride, int pad, const float *xb, const float *gamma, const float *beta,
                     const float *save_mean, const float *save_rstd, void *sums, size_t sums_bytes,
                     aq_conv_stats *desc, void *stream);
/* Both gradients of one convolution (autograd of adp_qtz_fan.py:574-601) from one call: dx as aq_conv_dgrad
 * (or, with xb != NULL, as aq_conv_dgrad_bn with its sums / desc; xb == NULL: gamma .. desc may be NULL) and
 * the weight-gradient slabs as aq_conv_wgrad_partial (x = the convolution's input; workspace / nslab as
 * there, to be finished by aq_conv_wgrad_reduce).  The two computations are independent and each alone
 * leaves most of the chip idle at these sizes, so where a kernel pair exists for the shape they run as ONE
 * launch with their workgroups side by side on the same CUs; otherwise as two launches.  Results are those
 * of the separate entry points, bit for bit.  With xb != NULL and desc->P == 0 on return the shape has no
 * reduction epilogue: call aq_bn_relu_bwd without dgrad_sums.  aq_conv_bwd_sums_bytes: size of `sums`. */
size_t aq_conv_bwd_sums_bytes(int N, int Cin, int H, int Wd, int Cout, int ksize, int stride, int pad);
int aq_conv_bwd(const float *dy, const float *W, const float *x, float *dx, int N, int Cin, int H, int Wd,
                int Cout, int ksize, int stride, int pad, const float *xb, const float *gamma,
                const float *beta, const float *save_mean, const float *save_rstd, void *sums,
                size_t sums_bytes, aq_conv_stats *desc, void *workspace, size_t workspace_bytes, int *nslab,
                void *stream);
/* dW = d loss / d W  (autograd of adp_qtz_fan.py:574-601 w.r.t. conv_w); deterministic two-stage
 * reduction through `workspace` (aq_conv_wgrad_workspace_bytes). */
int aq_conv_wgrad(const float *x, const float *dy, float *dW, int N, int Cin, int H, int Wd,
                  int Cout, int ksize, int stride, int pad, void *workspace, size_t workspace_bytes,
                  void *stream);
size_t aq_conv_wgrad_workspace_bytes(int N, int Cin, int H, int Wd, int Cout, int ksize, int stride,
                                     int pad);
/* The two stages of aq_conv_wgrad separately, so that the slab reductions of all layers of a backward
 * pass can run as ONE launch at its end: aq_conv_wgrad_partial writes *nslab partial results
 * ([nslab][Cout*Cin*9]) into workspace (same size as for aq_conv_wgrad; it must stay untouched until
 * reduced); aq_conv_wgrad_reduce sums the slabs of every job in ascending order into its dW. */
typedef struct aq_wgrad_job {
    const float *part; /* workspace filled by aq_conv_wgrad_partial */
    float *dW;         /* [n] */
    int32_t n;         /* Cout*Cin*9 */
    int32_t nslab;
} aq_wgrad_job;
int aq_conv_wgrad_partial(const float *x, const float *dy, int N, int Cin, int H, int Wd, int Cout,
                          int ksize, int stride, int pad, void *workspace, size_t workspace_bytes,
                          int *nslab, void *stream);
/* aq_conv_wgrad_partial for nlayers layers of ONE shape in one launch (groups of 8): the weight gradients of a backward
 * pass (autograd of adp_qtz_fan.py:574-601, one per MyConv2d) depend only on their own (x, dy), so a training step can
 * collect the layers of a residual stage and run them together at the end of the pass.  x[l], dy[l], workspace[l]:
 * host arrays of device pointers; every workspace has workspace_bytes (aq_conv_wgrad_workspace_bytes); *nslab: slabs
 * per layer (the same for all; fewer than a single-layer call makes when the layers together fill the chip, so the
 * sums agree with nlayers calls of aq_conv_wgrad_partial up to fp32 summation order, and are reproducible). */
int aq_conv_wgrad_partial_group(const float *const *x, const float *const *dy, void *const *workspace,
                                int nlayers, int N, int Cin, int H, int Wd, int Cout, int ksize, int stride,
                                int pad, size_t workspace_bytes, int *nslab, void *stream);
int aq_conv_wgrad_reduce(const aq_wgrad_job *jobs, int njobs, void *stream);

/* MyLinear (adp_qtz_fan.py:646-647): y[N,Out] = x[N,In] @ W[In,Out] + bias[Out]; and its gradients
 * dx[N,In], dW[In,Out], dbias[Out] (any of the three may be NULL). */
int aq_linear_fwd(const float *x, const float *W, const float *bias, float *y, int N, int In, int Out,
                  void *stream);
int aq_linear_bwd(const float *x, const float *W, const float *dy, float *dx, float *dW, float *dbias,
                  int N, int In, int Out, void *stream);

/* Batch normalisation over (N, H*W) of an NCHW tensor fused with the ReLU that follows it in the
 * reference's blocks: `F.relu(self.bn1(x))`, `F.relu(self.bn2(self.conv1(out)))`
 * (adp_qtz_fan.py:674-675, 690, 726, 730; the modules are nn.BatchNorm2d, :661-663, 686, 700, 707).
 * Batch-statistics mode (default): y = [relu](gamma*(x-mean)*rstd + beta) with the biased batch
 * variance, stores mean and rstd = 1/sqrt(var+eps) in save_mean / save_rstd [C] for the backward
 * pass and, when running_mean / running_var are given, updates them as nn.BatchNorm2d does
 * (r += momentum*(batch - r), unbiased variance; *num_batches_tracked += 1 when non-NULL).
 * AQ_BN_EVAL: normalise with the running statistics instead (save_*, workspace unused).
 * workspace: aq_bn_workspace_bytes(N, C, HW) bytes.
 * conv_stats != NULL (batch-statistics mode): x was written by aq_conv_fwd_stats and `workspace` is that
 * call's `stats` buffer, *conv_stats its descriptor; the statistics pass over x is skipped. */
#define AQ_BN_RELU 1u
#define AQ_BN_EVAL 2u
int aq_bn_relu_fwd(const float *x, const float *gamma, const float *beta, float *running_mean,
                   float *running_var, long long *num_batches_tracked, int N, int C, int HW, float eps,
                   float momentum, unsigned flags, float *y, float *save_mean, float *save_rstd,
                   void *workspace, size_t workspace_bytes, const aq_conv_stats *conv_stats, void *stream);
/* Gradient of the above in batch-statistics mode: dy is d loss / d y; dx = d loss / d x [+ res when
 * res != NULL: the gradient arriving over the block's shortcut, `out + x` adp_qtz_fan.py:677, 692],
 * dgamma[C], dbeta[C] (either may be NULL).  x, gamma, beta must be unchanged since the forward call
 * (the ReLU mask is recomputed from them).  dx may alias dy or res.
 * dgrad_sums != NULL: dy was written by aq_conv_dgrad_bn for this x and `workspace` is that call's `sums`
 * buffer, *dgrad_sums its descriptor; the reduction pass is skipped (AQ_BN_RELU form only). */
int aq_bn_relu_bwd(const float *dy, const float *x, const float *gamma, const float *beta,
                   const float *save_mean, const float *save_rstd, const float *res, int N, int C, int HW,
                   unsigned flags, float *dx, float *dgamma, float *dbeta, void *workspace,
                   size_t workspace_bytes, const aq_conv_stats *dgrad_sums, void *stream);
/* aq_bn_relu_fwd for the first BatchNorm of a block that changes dimensions: the same launch also writes the
 * block's shortcut of the raw input, shortcut[N, C + pad_front + pad_back, H/2, W/2] =
 * ZeroPad2d(AvgPool2d(2)(x)) (adp_qtz_fan.py:669-671, what aq_shortcut_fwd gives). */
int aq_bn_relu_fwd_shortcut(const float *x, const float *gamma, const float *beta, float *running_mean,
                            float *running_var, long long *num_batches_tracked, int N, int C, int H, int Wd,
                            float eps, float momentum, unsigned flags, float *y, float *save_mean,
                            float *save_rstd, void *workspace, size_t workspace_bytes,
                            const aq_conv_stats *conv_stats, int pad_front, int pad_back, float *shortcut,
                            void *stream);
/* aq_bn_relu_bwd for the first BatchNorm of a block that changes dimensions: its input x also feeds the
 * block's shortcut s = ZeroPad2d(AvgPool2d(2)(x)) (adp_qtz_fan.py:669-671, 673-677), and the gradient arriving
 * over it is given as dshort = d loss / d s [N, C + pad_front + pad_back, H/2, W/2] instead of a tensor of x's
 * shape: dx = BatchNorm+ReLU backward + a quarter of dshort[n, c + pad_front, y/2, x/2] (what aq_shortcut_bwd
 * followed by aq_bn_relu_bwd(res) gives, without the intermediate tensor). */
int aq_bn_relu_bwd_shortcut(const float *dy, const float *x, const float *gamma, const float *beta,
                            const float *save_mean, const float *save_rstd, const float *dshort, int N, int C,
                            int H, int Wd, int pad_front, int pad_back, unsigned flags, float *dx,
                            float *dgamma, float *dbeta, void *workspace, size_t workspace_bytes,
                            const aq_conv_stats *dgrad_sums, void *stream);
/* The network's tail `out = F.relu(self.bnlast(out)); out = self.ap(out)` (adp_qtz_fan.py:730-731, ap =
 * nn.AvgPool2d(8) on an 8x8 map, i.e. the mean over the plane) in one step: pooled[N, C] = mean over
 * H*W of [relu](batch_norm(x)); the activated tensor is never written.  Other arguments as
 * aq_bn_relu_fwd.  aq_bn_relu_pool_bwd takes d loss / d pooled [N, C] and returns dx, dgamma, dbeta. */
int aq_bn_relu_pool_fwd(const float *x, const float *gamma, const float *beta, float *running_mean,
                        float *running_var, long long *num_batches_tracked, int N, int C, int HW, float eps,
                        float momentum, unsigned flags, float *pooled, float *save_mean, float *save_rstd,
                        void *workspace, size_t workspace_bytes, const aq_conv_stats *conv_stats, void *stream);
int aq_bn_relu_pool_bwd(const float *dpooled, const float *x, const float *gamma, const float *beta,
                        const float *save_mean, const float *save_rstd, int N, int C, int HW, unsigned flags,
                        float *dx, float *dgamma, float *dbeta, void *workspace, size_t workspace_bytes,
                        void *stream);

/* y = relu(batch_norm(x) + add) with batch statistics, and its gradients: the tail of a post-activation residual block
 * (torchvision's BasicBlock, `out = relu(bn2(conv2(..)) + shortcut)`; the reference's own pre-activation blocks
 * adp_qtz_fan.py:673-693 add the shortcut after the convolution instead, see aq_conv_fwd_add), so no counterpart in the
 * reference.  Arguments as aq_bn_relu_fwd / _bwd; `add` has the shape of x.  The backward takes the forward's OUTPUT y
 * (mask y > 0), writes dx and d loss / d add (the masked gradient). */
int aq_bn_add_relu_fwd(const float *x, const float *gamma, const float *beta, float *running_mean, float *running_var,
                       long long *num_batches_tracked, int N, int C, int HW, float eps, float momentum, const float *add,
                       float *y, float *save_mean, float *save_rstd, void *workspace, size_t workspace_bytes,
                       const aq_conv_stats *conv_stats, void *stream);
int aq_bn_add_relu_bwd(const float *dy, const float *x, const float *y, const float *gamma, const float *beta,
                       const float *save_mean, const float *save_rstd, int N, int C, int HW, float *dx, float *dadd,
                       float *dgamma, float *dbeta, void *workspace, size_t workspace_bytes, void *stream);
size_t aq_bn_workspace_bytes(int N, int C, int HW);

/* `self.convK(F.relu(self.bnK(x)))` of a block (adp_qtz_fan.py:674-676, 690-691) without writing the
 * activated tensor: aq_bnconv_fwd computes the batch statistics of the raw x (one small launch), then the
 * convolution normalises + rectifies its input while staging it into LDS and adds res (may be NULL) in its
 * epilogue; save_mean / save_rstd / running statistics as aq_bn_relu_fwd.  aq_bnconv_wgrad_partial is
 * aq_conv_wgrad_partial with the same transformation of x (recomputed from the saved statistics).  The
 * input gradient is aq_conv_dgrad followed by aq_bn_relu_bwd on the raw x.  Only for shapes where
 * aq_bnconv_supported returns 1 (others: -ENOTSUP; use the separate entry points).
 * workspace of aq_bnconv_fwd: aq_bn_workspace_bytes(N, Cin, H*Wd). */
int aq_bnconv_supported(int N, int Cin, int H, int Wd, int Cout, int ksize, int stride, int pad);
int aq_bnconv_fwd(const float *x, const float *gamma, const float *beta, float *running_mean,
                  float *running_var, long long *num_batches_tracked, float eps, float momentum,
                  const float *W, const float *res, float *y, float *save_mean, float *save_rstd, int N,
                  int Cin, int H, int Wd, int Cout, int ksize, int stride, int pad, void *workspace,
                  size_t workspace_bytes, void *stream);
int aq_bnconv_wgrad_partial(const float *x, const float *gamma, const float *beta, const float *save_mean,
                            const float *save_rstd, const float *dy, int N, int Cin, int H, int Wd,
                            int Cout, int ksize, int stride, int pad, void *workspace,
                            size_t workspace_bytes, int *nslab, void *stream);

/* Shortcut of the dimension-increasing blocks, nn.Sequential(nn.AvgPool2d(2), nn.ZeroPad2d((0, 0, 0, 0,
 * C/2, C/2))) (adp_qtz_fan.py:669-671): y[N, pad_front+C+pad_back, H/2, W/2] = 2x2 average of x[N, C, H, W]
 * in channels [pad_front, pad_front+C), zero elsewhere; and its gradient dx[N, C, H, W]. */
int aq_shortcut_fwd(const float *x, float *y, int N, int C, int H, int Wd, int pad_front, int pad_back,
                    void *stream);
int aq_shortcut_bwd(const float *dy, float *dx, int N, int C, int H, int Wd, int pad_front, int pad_back,
                    void *stream);

/* nn.MaxPool2d(3, 2, 1) and its gradient for the stem of resnet18.py (the reference's models pool with AvgPool2d
 * only, adp_qtz_fan.py:669-671, 716: no counterpart there).  y[N, C, OH, OW], OH = (H - 1) / 2 + 1; idx[N, C, OH, OW]
 * = position ky * 3 + kx of the window's first maximum (NaN wins, as in torch); dx[N, C, H, W] gathers dy through idx. */
int aq_maxpool3s2_fwd(const float *x, float *y, uint8_t *idx, int N, int C, int H, int Wd, void *stream);
int aq_maxpool3s2_bwd(const float *dy, const uint8_t *idx, float *dx, int N, int C, int H, int Wd, void *stream);

/* Packed storage of hard-mode weights: the codes idx = argmax s_c (adp_qtz_fan_test.py:304; values < L)
 * as a little-endian bit stream, `bits` (1..8; 5 for L = 31 / 32) bits per code, code i in stream bits
 * [i*bits, (i+1)*bits); the layer's weights are q_level[idx] (:305).  out / packed hold
 * aq_packed_bytes(n, bits) = ceil(n*bits/8) bytes.  The reference keeps no packed form on disk; this is
 * the export format of this library. */
size_t aq_packed_bytes(long long n, int bits);
int aq_pack_codes(const uint8_t *idx, long long n, int bits, uint8_t *out, void *stream);
int aq_unpack_codes(const uint8_t *packed, long long n, int bits, uint8_t *idx, void *stream);

/* One training batch from the raw image array, replacing the reference's per-image CPU transforms
 * RandomCrop(32, padding=4) + RandomHorizontalFlip() + ToTensor() + Normalize(mean, std)
 * (adp_qtz_fan.py:164-169) and the DataLoader gather (:178-183):
 *   data  [n_images, H, W, C] uint8, channels last (the layout of adp_qtz_fan.py:452-530), device memory
 *   index [B] image of each batch slot (NULL: slot b = image b)
 *   off_y, off_x [B] crop offsets inside the zero-padded image, each 0..2*pad (NULL: pad = no shift)
 *   flip  [B] nonzero: mirror horizontally after the crop (NULL: never)
 *   mean, std: HOST arrays of C floats;  out [B, C, H, W] fp32 = (u8/255 - mean[c]) / std[c]
 * Validation batches (ToTensor + Normalize only, :171-174) pass NULL offsets / flip.  The caller draws
 * the random numbers. */
int aq_augment_batch(const uint8_t *data, long long n_images, const int *index, const int *off_y,
                     const int *off_x, const uint8_t *flip, int B, int H, int W, int C, int pad,
                     const float *mean, const float *std, float *out, void *stream);

/* max |v| of n tensors in one launch (per 32 tensors): the scales s the power-of-two variants derive
 * from a trained full-precision model, `torch.max(torch.abs(param))` per conv / linear weight
 * (adp_qtz_fix.py:208-213, adp_qtz_hint.py:204-209).  tensors, sizes: HOST arrays of n device
 * pointers / element counts; out: device [n]. */
int aq_absmax(const float *const *tensors, const long long *sizes, int n, float *out, void *stream);

/* torch.optim.Adam's update (defaults: no weight decay, no amsgrad) of many small tensors in one launch per
 * 48 items: the reference's optimizer_q (q_level vectors) and optimizer (conv0, BatchNorm, linear bias)
 * steps, adp_qtz_fan.py:223-242, 296-304.  Per item: *step += 1 (one-element fp32 tensor, one per item, as
 * Adam keeps it with capturable=True), m/v = exp_avg / exp_avg_sq, p updated in place.  reg != 0: g += reg*p
 * first, written back to g (the regulariser's gradient 2*l2_reg*w of the dense "weight" tensors, model_loss
 * adp_qtz_fan.py:737-754).  items: HOST array. */
typedef struct aq_adam_item {
    float *p, *g, *m, *v, *step;
    int64_t n;
    float reg;
    int32_t reserved;
} aq_adam_item;
int aq_adam_step(const aq_adam_item *items, int nitems, double lr, double beta1, double beta2, double eps,
                 void *stream);

/* Tail of the step, `loss = criterion(logits, y) + model_loss(config, beta, model)` (adp_qtz_fan.py:288-289,
 * criterion = nn.CrossEntropyLoss(), :202; model_loss :737-754):
 * aq_sumsq: out[i] = sum of squares of tensor i (HOST arrays of n device pointers / sizes), one launch
 *   per 32 tensors: the `torch.sum(param**2)` terms of the dense "weight" tensors.  Tensors of more than 16384
 *   elements are shared by several workgroups, whose partial sums (double) pass through `workspace`
 *   (aq_sumsq_workspace_bytes(sizes, n) bytes, 8-byte aligned; 0 = none needed, workspace may be NULL) and are
 *   finished by a second launch; everything beyond a thread's first 64 elements is summed in double.
 * aq_ce_loss: logits [N, C] fp32, labels [N] int64 ->
 *   out[0] = mean cross entropy, out[1] = l2_reg * (sum sumsq_a[0..na) + sum sumsq_b[0..nb)), out[2] = their sum;
 *   dlogits [N, C] (may be NULL) = grad_scale * d out[0] / d logits = grad_scale * (softmax - onehot) / N
 *   (grad_scale = 1 / world size lets a data-parallel job SUM its gradients into the mean without a scaling
 *   pass).  A label outside [0, C) makes out[0] NaN.  Heads of more than 16384 logits (ImageNet-shaped) run one
 *   workgroup per image and pass the per-image terms through `workspace` (aq_ce_loss_workspace_bytes(N, C) bytes;
 *   0 for small heads, workspace may then be NULL). */
size_t aq_sumsq_workspace_bytes(const long long *sizes, int n);
int aq_sumsq(const float *const *tensors, const long long *sizes, int n, float *out, void *workspace, size_t workspace_bytes,
             void *stream);
size_t aq_ce_loss_workspace_bytes(int N, int C);
int aq_ce_loss(const float *logits, const long long *labels, int N, int C, const float *sumsq_a, int na,
               const float *sumsq_b, int nb, float l2_reg, float grad_scale, float *dlogits, float *out,
               void *workspace, size_t workspace_bytes, void *stream);

/* ---- Ternary and power-of-two quantisers of the papers the reference cites (README.md:3-29) -----------------
 * The reference's scripts hold no implementation of them (only the citations and, in the file headers
 * adp_qtz_fix.py:4-6, their names), so there is no reference code to cite line by line: the entry points below
 * restate the published algorithms, and their CPU checker was written from the same papers (parity unpinned).
 *   TWN (Li, Zhang, Liu 2016):  delta = 0.7 * mean|w|;  w~ = alpha * sign(w) where |w| > delta, 0 elsewhere,
 *                               alpha = mean of |w| over |w| > delta; straight-through gradient dw = dw~.
 *   TTQ (Zhu, Han, Mao, Dally 2017): w~ = +Wp where w > delta, -Wn where w < -delta (Wp, Wn learned, delta = t * max|w|
 *                               in the paper; factor * mean|w| selectable); dw = Wp * dw~ / dw~ / Wn * dw~ by region,
 *                               dWp = sum of dw~ over the positives, dWn = -(sum over the negatives).
 *   INQ (Zhou et al. 2017):     weights join a frozen partition largest magnitudes first and are rounded there to
 *                               {0, +-2^n2 .. +-2^n1}; the gradient of frozen weights is masked.
 * One aq_tern_seg per layer; unused pointers may be NULL. */
#define AQ_TERN_TWN 0
#define AQ_TERN_TTQ 1
#define AQ_TERN_INQ 2
typedef struct aq_tern_seg {
    const float *W;       /* [n] latent full-precision weights                                             */
    int64_t n;
    /* forward outputs */
    uint8_t *codes;       /* [n] 0 = zero, 1 = positive, 2 = negative                                      */
    uint8_t *packed;      /* [ceil(n/4)] the same codes, 2 bits each, code i in bits [2i, 2i+2) of the stream */
    float *Wt;            /* [n] quantized weights q[code]                                                 */
    float *stats;         /* [4] written: delta, positive scale, negative scale, number of non-zero codes  */
    float *q;             /* [3] written: codebook {0, +pos, -neg} (for aq_conv_fwd_codes with L = 3)       */
    const float *Wp, *Wn; /* [1] TTQ: learned scales (device memory)                                       */
    /* backward */
    const uint8_t *mask;  /* [n] INQ: 1 = trainable, 0 = frozen                                            */
    const float *dWt;     /* [n] d loss / d w~                                                             */
    float *dW;            /* [n] d loss / d w (latent)                                                     */
    float *dWp, *dWn;     /* [1] TTQ: d loss / d Wp, d loss / d Wn                                          */
    /* aq_inq_effective */
    const float *frozen;  /* [n] INQ: decoded values of the frozen weights (read where mask == 0)          */
    /* aq_tern_bwd_adam: torch.optim.Adam state (exp_avg, exp_avg_sq, one-element step) of W and of the TTQ scales */
    float *adam_m, *adam_v, *adam_step;
    float *Wp_m, *Wp_v, *Wp_step;
    float *Wn_m, *Wn_v, *Wn_step;
} aq_tern_seg;
size_t aq_tern_workspace_bytes(const aq_tern_seg *segs, int nseg);
/* Quantise nseg layers (at most 32 per call): per layer a reduction of |w| for the threshold
 * delta = factor * mean|w| (threshold_from_max != 0: factor * max|w|), then codes / packed codes / scales / codebook
 * (and Wt).  mode: AQ_TERN_TWN or AQ_TERN_TTQ.  Reductions run in a fixed order (deterministic). */
int aq_tern_fwd(const aq_tern_seg *segs, int nseg, int mode, float factor, int threshold_from_max, void *workspace,
                size_t workspace_bytes, void *stream);
/* Gradient of the latent weights (and of the TTQ scales) from d loss / d w~, one pass; mode AQ_TERN_TWN, _TTQ (uses
 * stats[0] = delta as aq_tern_fwd left it) or _INQ (mask). */
int aq_tern_bwd(const aq_tern_seg *segs, int nseg, int mode, void *workspace, size_t workspace_bytes, void *stream);
/* aq_tern_bwd with torch.optim.Adam's update (no weight decay / amsgrad) applied in the same pass: per layer
 * g = rule(dWt) + reg * W (reg = 2 * l2_reg of the regulariser on "weight" tensors, adp_qtz_fan.py:737-743), then W,
 * adam_m, adam_v are updated in place and adam_step advanced by one; dW (may be NULL) receives g.  TTQ: Wp and Wn
 * (written through their const pointers) are updated with d loss / d Wp, d loss / d Wn from the same pass and their
 * own Adam state; dWp / dWn (may be NULL) receive the two gradients.  The optimizer-side analogue of
 * aq_synth_bwd_adam for the quantisers the reference only cites; any number of layers per call up to 32. */
int aq_tern_bwd_adam(const aq_tern_seg *segs, int nseg, int mode, double lr, double beta1, double beta2, double eps,
                     float reg, void *workspace, size_t workspace_bytes, void *stream);
/* Effective weights of INQ layers: Wt[i] = mask[i] ? W[i] : frozen[i] for every layer in one launch (the weights a
 * layer computes with between INQ stages; frozen = the decoded power-of-two levels). */
int aq_inq_effective(const aq_tern_seg *segs, int nseg, void *stream);
/* The same, and sumsq[i] = sum of W[i]^2 over the whole latent tensor from the same pass (the layer's term of the l2
 * regulariser on "weight" tensors, adp_qtz_fan.py:737-743: aq_sumsq need not read these tensors again); per-wavefront
 * partials in workspace (aq_tern_workspace_bytes(segs, nseg), 16-byte aligned), added in double in a fixed order. */
int aq_inq_effective_sumsq(const aq_tern_seg *segs, int nseg, float *sumsq, void *workspace, size_t workspace_bytes,
                           void *stream);
/* One INQ partition step, in place: every weight with mask 1 and |w| >= tau is rounded to its power-of-two level
 * (exponents n2..n1, or 0) and its mask cleared; codes (may be NULL): 255 = still trainable, 0 = zero,
 * 1 + 2 * (k - n2) + (w < 0) for +-2^k.  The caller derives tau from the magnitude quantile of the stage and
 * n1 = floor(log2(4 * max|w| / 3)), n2 = n1 + 1 - 2^(bits-1) / 2 once, before the first stage. */
int aq_inq_partition(float *W, uint8_t *mask, uint8_t *codes, long long n, float tau, int n1, int n2, void *stream);

/* ---- Probabilistic Knowledge Transfer loss (Passalis & Tefas 2018, cited at README.md:24-29) ----------------------
 * No counterpart in the reference's scripts (adp_qtz_hint.py:301-318 trains without a teacher pass): restated from the
 * published method, parity unpinned.  student [N, Ds], teacher [N, Dt] feature matrices of one batch:
 *   xh = x / (|x| + eps) per row;  K = (xh xh^T + 1) / 2  (N x N, on the fp32 matrix cores);  P_ij = K_ij / sum_j K_ij;
 *   loss[0] = mean_ij Pt_ij * log((Pt_ij + eps) / (Ps_ij + eps)).
 * dstudent != NULL: also d loss / d student [N, Ds] (times grad_scale[0] when grad_scale != NULL, a device scalar); the
 * teacher is a constant.  workspace: aq_pkt_workspace_bytes(N, Ds, Dt) bytes, 16-byte aligned. */
size_t aq_pkt_workspace_bytes(int N, int Ds, int Dt);
int aq_pkt_loss(const float *student, const float *teacher, int N, int Ds, int Dt, float eps, float *loss, const float *grad_scale,
                float *dstudent, void *workspace, size_t workspace_bytes, void *stream);
/* K = (xh xh^T + 1) / 2 alone.  workspace: ceil256(N * D * 4) + ceil256(N * 4) bytes. */
int aq_pkt_kernel_matrix(const float *feat, int N, int D, float eps, float *K, void *workspace, size_t workspace_bytes, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* AQ_HIP_H */
