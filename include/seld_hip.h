/*
 * seld_hip.h -- C ABI of the MI355X (gfx950) DualQ-SELD-TCN hot path.
 *
 * The reference (AuroraEchos/Sound-Event-Localization-and-Detection) has no FFI of its own:
 * its seam is the Python call into ATen.  Every entry point below
 * names the reference call it stands in for.  Conventions, all entry points:
 *   - plain pointers and sizes only; every pointer is a DEVICE pointer borrowed for the call;
 *   - tensors are contiguous fp32, NCHW / NCT, hypercomplex components component-major
 *     along the channel axis;
 *   - work is enqueued on `stream` (a hipStream_t passed as void*); nothing synchronises,
 *     nothing allocates (workspaces are caller-provided), nothing throws;
 *   - return value 0 = success, negative = SELD_E* code below.
 */
#ifndef SELD_HIP_H
#define SELD_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SELD_OK            0
#define SELD_EINVAL       -1   /* bad descriptor (rank, channel divisibility, groups != 1 ...)      */
#define SELD_EWORKSPACE   -2   /* workspace too small                                              */
#define SELD_ELAUNCH      -3   /* hipLaunch reported an error (see seld_last_hip_error)            */
#define SELD_EUNSUPPORTED -4   /* valid request this build has no kernel for                       */

/* version / capability */
int         seld_abi_version(void);          /* bumps on any signature change                      */
const char* seld_build_arch(void);           /* "gfx950"                                           */
int         seld_last_hip_error(void);       /* last hipError_t seen by a launch in this thread    */
/* The SELD_* environment switches (csrc/env.h: kernel-generation selection, all result-preserving) are read once at
 * first use; seld_env_reload re-reads them.  seld_tuning_build: 1 if the library was compiled with -DSELD_TUNING,
 * which adds timing-experiment switches that produce WRONG results (never set for the shipped library). */
int         seld_env_reload(void);
int         seld_tuning_build(void);

/* ------------------------------------------------------------------------------------------
 * Hypercomplex convolution.  Replaces quaternion_conv (quaternion_ops.py:125-147),
 * dual_quaternion_conv (dual_quaternion_ops.py:111-153) and the real F.conv1d/2d of the
 * domain='R' model (model.py:81-86,106-107,185,198,276).  The kernels read the 1/4/8
 * COMPONENT weight tensors (Cout/A, Cin/A, kh, kw) directly and apply the Hamilton signs
 * while staging into LDS; the expanded (Cout, Cin) matrix never exists in memory.
 * ------------------------------------------------------------------------------------------ */
typedef struct seld_conv_desc {
    int32_t algebra;     /* 1 = real, 4 = quaternion, 8 = dual quaternion                       */
    int32_t ndim;        /* 1 (NCT) or 2 (NCHW); for ndim == 1 use in[0] = k[0] = 1 etc.         */
    int32_t N, Cin, Cout;
    int32_t in[2];       /* input  H, W  (T in in[1] for 1-D)                                   */
    int32_t k[2];        /* kernel kh, kw                                                       */
    int32_t stride[2];
    int32_t pad[2];
    int32_t dil[2];      /* the reference spells the argument `dilatation`                       */
    int32_t groups;      /* only 1 is supported (the reference never passes anything else)      */
} seld_conv_desc;

/* epilogue flags for seld_hc_conv_fwd_ex */
#define SELD_EPI_NONE        0
#define SELD_EPI_ACCUMULATE  1   /* y += conv(x)   (skip-connection running sum, model.py:210-212) */
#define SELD_EPI_ADD         2   /* y = conv(x) + addend  (x + conv2_residual(y), model.py:132)    */
#define SELD_EPI_STATS       4   /* also accumulate the per-channel sum / sum of squares of the stored result
                                    (BatchNorm batch statistics) into a stats buffer, see below            */

/* Layout of every `stats` buffer: SELD_STATS_REPLICAS rows of 2*C floats [sum(C) | sum of squares(C)];
 * producers add to row (workgroup index % replicas) so that thousands of workgroups do not serialise on
 * 2*C addresses; seld_bn_finalize sums the rows.  The caller zero-fills the buffer before use. */
#define SELD_STATS_REPLICAS 64

int seld_hc_conv_out_shape(const seld_conv_desc* d, int32_t out[2]);

int seld_hc_conv_fwd(const seld_conv_desc* d, const float* x, const float* const w[8],
                     const float* bias /* nullable, (Cout) */, float* y, void* stream);

int seld_hc_conv_fwd_ex(const seld_conv_desc* d, const float* x, const float* const w[8],
                        const float* bias, float* y, int32_t epilogue,
                        const float* addend /* SELD_EPI_ADD */, float* stats /* SELD_EPI_STATS, pre-zeroed */,
                        void* stream);

/* dx = conv_transpose(dy, W)   (autograd of the F.convNd call at quaternion_ops.py:147) */
int seld_hc_conv_bwd_data(const seld_conv_desc* d, const float* dy, const float* const w[8],
                          float* dx, void* stream);
/* Same with a caller-provided workspace of seld_hc_conv_bwd_data_workspace(d) bytes: the component tensors
 * are transposed into it ([c][o][k], one tiny launch) so that the data gradient stages its weight rows
 * with 16-byte loads exactly like the forward (about 2x faster on the TCN layers).  Without (or with too
 * small) a workspace the gather path of seld_hc_conv_bwd_data is used. */
size_t seld_hc_conv_bwd_data_workspace(const seld_conv_desc* d);
int seld_hc_conv_bwd_data_ex(const seld_conv_desc* d, const float* dy, const float* const w[8],
                             float* dx, void* workspace, size_t workspace_bytes, void* stream);

/* seld_hc_conv_bwd_data_ex in two steps: the weight re-layout (Wt[comp][c][o][k] = W[comp][o][c][k], into a workspace of
 * seld_hc_conv_bwd_data_workspace(d) bytes) can be issued ahead of time -- the host mirror does it on a side stream during
 * the forward pass -- and the data gradient then starts from the workspace. */
int seld_hc_conv_transpose_weights(const seld_conv_desc* d, const float* const w[8], void* workspace,
                                   size_t workspace_bytes, void* stream);
int seld_hc_conv_bwd_data_wt(const seld_conv_desc* d, const float* dy, const void* wt_workspace, float* dx, void* stream);

/* dw[c] (component gradients, same shapes as w[c]) and dbias (nullable).  The Hamilton fold
 * (sum of the signed blocks that share a component) is done on device with float atomics, so no
 * workspace is needed any more: seld_hc_conv_bwd_weight_workspace returns 0 and `workspace` may be NULL
 * (both kept for ABI stability). */
size_t seld_hc_conv_bwd_weight_workspace(const seld_conv_desc* d);
int seld_hc_conv_bwd_weight(const seld_conv_desc* d, const float* x, const float* dy,
                            float* const dw[8], float* dbias /* nullable */,
                            void* workspace, size_t workspace_bytes, void* stream);

/* Same, but ACCUMULATES: dw[c] += ..., dbias += ... .  Lets the caller point dw at slices of one flat
 * gradient buffer (zeroed once per step) so that no per-tensor gradient add is ever launched. */
int seld_hc_conv_bwd_weight_acc(const seld_conv_desc* d, const float* x, const float* dy,
                                float* const dw[8], float* dbias /* nullable */, void* stream);

/* ---- two convolutions of one geometry in one launch ------------------------------------------------------
 * A residual block calls the hypercomplex convolution twice on the same tensor with the same geometry:
 * conv1_filter | conv1_gate (model.py:121-122) and conv2_skip | conv2_residual (model.py:130-132).  The pair entry
 * points run both in one launch where that pays (data gradient: ONE kernel sums both contributions, its reduction
 * runs over dyA then dyB; weight gradient: the grid carries both; forward: one launch with two reduction passes per
 * workgroup on the 1x1 layers, two launches of the single kernel otherwise).  The layers are 30-70 us each on an
 * MI355X, so a launch's fixed cost is up to a third of it.
 * seld_hc_conv_pair_supported(d, which) (which: 0 forward, 1 data gradient, 2 weight gradient) tells whether the
 * pair form runs for this shape; if not, the entry point returns SELD_EUNSUPPORTED and the caller issues the two
 * single calls. */
int seld_hc_conv_pair_supported(const seld_conv_desc* d, int32_t which);


int seld_hc_conv_pair_fwd(const seld_conv_desc* d, const float* x, const float* const wA[8], const float* const wB[8],
                          const float* biasA /* nullable */, const float* biasB /* nullable */, float* yA, float* yB,
                          int32_t epilogueA, int32_t epilogueB, const float* addendA /* nullable */,
                          const float* addendB /* nullable */, float* statsA /* nullable */,
                          float* statsB /* nullable */, void* stream);

/* The same from two workspaces filled ahead of time by seld_hc_conv_transpose_weights */
int seld_hc_conv_pair_bwd_data_wt(const seld_conv_desc* d, const float* dyA, const float* dyB, const void* wtA,
                                  const void* wtB, float* dx, void* stream);

/* dx = dgrad(dyA, wA) + dgrad(dyB, wB); workspace: 2 * seld_hc_conv_bwd_data_workspace(d) bytes, required */
int seld_hc_conv_pair_bwd_data(const seld_conv_desc* d, const float* dyA, const float* dyB, const float* const wA[8],
                               const float* const wB[8], float* dx, void* workspace, size_t workspace_bytes,
                               void* stream);

/* dwA[c] += ..., dwB[c] += ..., dbiasA += ..., dbiasB += ... (bias gradients nullable) */
int seld_hc_conv_pair_bwd_weight_acc(const seld_conv_desc* d, const float* x, const float* dyA, const float* dyB,
                                     float* const dwA[8], float* const dwB[8], float* dbiasA, float* dbiasB,
                                     void* stream);

/* Diagnostics: label of the kernel symbol a call would launch ("hc_conv_kernel<4, 4, 1, 3>"), so that
 * HIP-event timings taken by the caller can be matched with rocprofv3's per-kernel statistics.
 * which: 0 forward, 1 data gradient, 2 weight gradient.  buflen >= 48. */
int seld_hc_conv_kernel_label(const seld_conv_desc* d, int32_t which, char* buf, int32_t buflen);

/* ------------------------------------------------------------------------------------------
 * Hypercomplex / real linear  y[rows, out] = x[rows, in] @ M + b.
 *   SELD_LIN_REAL   : torch.nn.Linear, weight (out, in)                 (model.py:23,439,454,458)
 *   SELD_LIN_QUAT   : quaternion_linear, weights (in/4, out/4)          (quaternion_ops.py:299-327)
 *   SELD_LIN_DUALQ  : dual_quaternion_linear, weights (in/8, out/8), the TRANSPOSED block
 *                     arrangement of dual_quaternion_ops.py:170-188
 * ------------------------------------------------------------------------------------------ */
#define SELD_LIN_REAL   1
#define SELD_LIN_QUAT   4
#define SELD_LIN_DUALQ  8

int seld_hc_linear_fwd(int32_t kind, int32_t rows, int32_t in_features, int32_t out_features,
                       const float* x, const float* const w[8], const float* bias, float* y, void* stream);
/* dx = dy @ M^T (nullable), dw[c] = folded x^T @ dy (nullable), dbias = column sums of dy (nullable).
 * workspace: seld_hc_linear_bwd_workspace bytes (up to 8 row splits of x^T @ dy and of the column sums, written with
 * plain stores and added in a fixed order by the fold: nothing is zeroed, no atomics, run-to-run identical). */
size_t seld_hc_linear_bwd_workspace(int32_t kind, int32_t in_features, int32_t out_features);
int seld_hc_linear_bwd(int32_t kind, int32_t rows, int32_t in_features, int32_t out_features,
                       const float* x, const float* dy, const float* const w[8],
                       float* dx /* nullable */, float* const dw[8] /* nullable */, float* dbias /* nullable */,
                       void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * BatchNorm (torch.nn.BatchNorm1d/2d at model.py:88-92,279; eps 1e-5, momentum 0.1) fused
 * with the activation that follows it in the reference graph.
 * x is (N, C, S) with S = prod(spatial).
 * ------------------------------------------------------------------------------------------ */
#define SELD_ACT_NONE    0
#define SELD_ACT_RELU    1
#define SELD_ACT_TANH    2
#define SELD_ACT_SIGMOID 3

/* per-channel sum / sum of squares of x (N, C, S) into a stats buffer (layout above) */
int seld_channel_stats(const float* x, int32_t N, int32_t C, int32_t S, float* stats, void* stream);

/* from raw sums: mean/invstd (saved for backward) and the running-stat update (train mode) */
int seld_bn_finalize(const float* stats, int32_t C, int64_t count, float eps, float momentum,
                     float* mean, float* invstd, float* running_mean /* nullable */,
                     float* running_var /* nullable */, void* stream);

/* same, plus the rest of torch.nn.BatchNorm's train-mode bookkeeping in the one launch: *num_batches_tracked += 1
 * (nullable; `_BatchNorm.forward` does it as a separate op) and, if clear_stats != 0, the stats buffer is left
 * zero-filled so that the caller can reuse it without another fill. */
int seld_bn_finalize_ex(float* stats, int32_t C, int64_t count, float eps, float momentum,
                        float* mean, float* invstd, float* running_mean /* nullable */,
                        float* running_var /* nullable */, int64_t* num_batches_tracked /* nullable */,
                        int32_t clear_stats, void* stream);

/* two BatchNorm layers of the same channel count in one launch (batch_filter2 / batch_gate2 of a residual block,
 * model.py:123-126, whose statistics the one filter|gate pair convolution gathered) */
int seld_bn_finalize2_ex(float* statsA, float* statsB, int32_t C, int64_t count, float eps, float momentum,
                         float* meanA, float* invstdA, float* running_meanA, float* running_varA,
                         int64_t* num_batches_trackedA, float* meanB, float* invstdB, float* running_meanB,
                         float* running_varB, int64_t* num_batches_trackedB, int32_t clear_stats, void* stream);

/* eval mode: mean = running_mean, invstd = 1/sqrt(running_var + eps) */
int seld_bn_eval_stats(const float* running_mean, const float* running_var, int32_t C, float eps,
                       float* mean, float* invstd, void* stream);

/* BatchNorm2d -> ReLU -> MaxPool2d(ph, pw) (stride = window, floor mode) in one pass (model.py:278-281):
 * reads the conv output y (N, C, H, W), writes the pooled map and a uint8 arg-max per pooled element. */
int seld_bn_relu_pool_fwd(const float* y, int32_t N, int32_t C, int32_t H, int32_t W, int32_t ph, int32_t pw,
                          const float* mean, const float* invstd, const float* gamma, const float* beta,
                          float* pooled, uint8_t* idx, void* stream);
/* Backward of the above: red (2C, pre-zeroed) receives dgamma | dbeta (reduced from pooled-size tensors
 * only), dy (N, C, H, W) the gradient w.r.t. the conv output.  train = 0: running statistics were used. */
int seld_bn_relu_pool_bwd(const float* dpooled, const float* pooled, const uint8_t* idx, const float* y,
                          int32_t N, int32_t C, int32_t H, int32_t W, int32_t ph, int32_t pw, const float* mean,
                          const float* invstd, const float* gamma, const float* beta, int32_t train,
                          float* red, float* dy, void* stream);
/* The stage's Dropout (model.py:282) in the same passes, for the shapes seld_bn_relu_pool_drop_ok accepts (pw == 1,
 * W % 4 == 0, H % ph == 0): forward also writes dropped = pooled * mask (the mask seld_dropout_fwd draws for the same
 * seed / offset / state on the pooled tensor); backward takes dpooled = the gradient BEHIND the Dropout and replays the
 * mask while it loads it.  drop_p == 0: identical to the two entries above (dropped is not written). */
int seld_bn_relu_pool_drop_ok(int32_t H, int32_t W, int32_t ph, int32_t pw);
int seld_bn_relu_pool_fwd_drop(const float* y, int32_t N, int32_t C, int32_t H, int32_t W, int32_t ph, int32_t pw,
                               const float* mean, const float* invstd, const float* gamma, const float* beta,
                               float* pooled, uint8_t* idx, float drop_p, uint64_t seed, uint64_t offset,
                               const uint64_t* state, float* dropped, void* stream);
int seld_bn_relu_pool_bwd_drop(const float* dpooled, const float* pooled, const uint8_t* idx, const float* y,
                               int32_t N, int32_t C, int32_t H, int32_t W, int32_t ph, int32_t pw, const float* mean,
                               const float* invstd, const float* gamma, const float* beta, int32_t train,
                               float* red, float* dy, float drop_p, uint64_t seed, uint64_t offset,
                               const uint64_t* state, void* stream);

/* The same backward for a convolution whose INPUT needs no gradient (the first layer), without ever writing the
 * gradient w.r.t. the conv output (1.6 GB at batch 32):
 *   seld_bn_relu_pool_bwd_coef        : the reductions (red = dgamma | dbeta) and coef = [c1 | a | c0] (3C) such that
 *                                       dy = y*c1 + dz*a + c0; conv_dbias (nullable, C) += sum over positions of dy
 *   seld_hc_conv_bwd_weight_bnpool_acc: dw[c] += weight gradient, dy formed from y / pooled / dpooled / idx / coef while
 *                                       the operand is staged (3x3 taps, pooling along H only; else SELD_EUNSUPPORTED) */
int seld_bn_relu_pool_bwd_coef(const float* dpooled, const float* pooled, const uint8_t* idx, const float* y,
                               int32_t N, int32_t C, int32_t H, int32_t W, int32_t ph, int32_t pw, const float* mean,
                               const float* invstd, const float* gamma, const float* beta, int32_t train,
                               float* red, float* coef, float* conv_dbias /* nullable */, void* stream);
/* the same with dpooled = the gradient BEHIND the stage's Dropout(drop_p) (model.py:282): its mask -- the one
 * seld_dropout_fwd draws for (seed, offset, state) -- is replayed while dpooled is loaded, no dropout-backward pass */
int seld_bn_relu_pool_bwd_coef_drop(const float* dpooled, const float* pooled, const uint8_t* idx, const float* y,
                                    int32_t N, int32_t C, int32_t H, int32_t W, int32_t ph, int32_t pw,
                                    const float* mean, const float* invstd, const float* gamma, const float* beta,
                                    int32_t train, float* red, float* coef, float* conv_dbias, float drop_p,
                                    uint64_t seed, uint64_t offset, const uint64_t* state, void* stream);
int seld_hc_conv_bwd_weight_bnpool_acc(const seld_conv_desc* d, const float* x, const float* y, const float* pooled,
                                       const float* dpooled, const uint8_t* idx, int32_t ph, const float* coef,
                                       float* const dw[8], void* stream);
int seld_hc_conv_bwd_weight_bnpool_drop_acc(const seld_conv_desc* desc, const float* x, const float* y,
                                            const float* pooled, const float* dpooled, const uint8_t* idx, int32_t ph,
                                            const float* coef, float* const dw[8], float drop_p, uint64_t seed,
                                            uint64_t offset, const uint64_t* state, void* stream);

/* y = act(gamma * (x - mean) * invstd + beta) */
int seld_bn_act_fwd(const float* x, int32_t N, int32_t C, int32_t S, const float* mean, const float* invstd,
                    const float* gamma, const float* beta, int32_t act, float* y, void* stream);

/* backward of the above in two passes:
 *   reduce: dgamma[c] = sum dz * xhat, dbeta[c] = sum dz, with dz = dy * act'(y)
 *   apply : dx = gamma * invstd * (dz - dbeta/M - xhat * dgamma/M)   (train: batch statistics)
 *           dx = gamma * invstd * dz                                  (eval: running statistics) */
int seld_bn_act_bwd_reduce(const float* dy, const float* x, const float* y, int32_t N, int32_t C, int32_t S,
                           const float* mean, const float* invstd, const float* gamma, const float* beta,
                           int32_t act, float* dgamma_dbeta /* (2C) pre-zeroed */, void* stream);
int seld_bn_act_bwd_apply(const float* dy, const float* x, const float* y, int32_t N, int32_t C, int32_t S,
                          const float* mean, const float* invstd, const float* gamma, const float* beta,
                          int32_t act, const float* dgamma_dbeta, int32_t train, float* dx, void* stream);

/* One-pass training-mode backward of the same op: dgamma / dbeta are ADDED to dgamma_dbeta (it need not be zero),
 * dx = BatchNorm-backward((dy [+ dy2]) * act'(y))  (dx nullable: parameter gradients only; dy2 nullable: the gradient
 * from y's second consumer -- the residual sum x_hat + conv2_residual(..) of model.py:132 -- added on load instead of
 * by a separate kernel).  SELD_EUNSUPPORTED when S % 4 != 0 or N*S > 32768: use the reduce + apply pair. */
int seld_bn_act_bwd_fused(const float* dy, const float* x, const float* y, int32_t N, int32_t C, int32_t S,
                          const float* mean, const float* invstd, const float* gamma, int32_t act,
                          float* dgamma_dbeta, const float* dy2, float* dx, void* stream);

/* gated activation of the residual block (model.py:121-128):
 *   y = tanh(bn_f(yf)) * sigmoid(bn_g(yg)) * mask[n, c]      (mask nullable = Dropout1d channel mask,
 *                                                            already scaled by 1/(1-p))            */
int seld_gate_fwd(const float* yf, const float* yg, int32_t N, int32_t C, int32_t S,
                  const float* mean_f, const float* invstd_f, const float* gamma_f, const float* beta_f,
                  const float* mean_g, const float* invstd_g, const float* gamma_g, const float* beta_g,
                  const float* mask, float* y, void* stream);
/* backward: reduce pass accumulates (dgamma_f, dbeta_f, dgamma_g, dbeta_g) into red[4C] (pre-zeroed),
 * apply pass writes dyf, dyg */
int seld_gate_bwd_reduce(const float* dy, const float* yf, const float* yg, int32_t N, int32_t C, int32_t S,
                         const float* mean_f, const float* invstd_f, const float* gamma_f, const float* beta_f,
                         const float* mean_g, const float* invstd_g, const float* gamma_g, const float* beta_g,
                         const float* mask, float* red, void* stream);
int seld_gate_bwd_apply(const float* dy, const float* yf, const float* yg, int32_t N, int32_t C, int32_t S,
                        const float* mean_f, const float* invstd_f, const float* gamma_f, const float* beta_f,
                        const float* mean_g, const float* invstd_g, const float* gamma_g, const float* beta_g,
                        const float* mask, const float* red, int32_t train, float* dyf, float* dyg, void* stream);

/* First stage with the pooling decision inside the convolution (csrc/hcq_conv.hip hcq_first_pool_kernel): for
 * conv -> BatchNorm2d -> ReLU -> MaxPool2d(8, 1) on the network input (model.py:273-281) the sign of gamma says whether a
 * window's maximum after BatchNorm + ReLU sits at the largest or the smallest conv output, so the convolution writes y,
 * the batch statistics and, per window, that raw value and its row; seld_bn_pool_finish applies relu(a v + b) on the
 * pooled-size tensor once the statistics are final.  wpack = seld_hcq_pack(desc, mode 2, ...); SELD_EUNSUPPORTED unless
 * seld_hcq_pack_floats(desc, 2, 1) > 0 (3x3 'same', 1-2 input block channels, height % 8 == 0, width % 64 == 0). */
int seld_hcq_first_pool(const seld_conv_desc* desc, const float* x, const float* wpack, const float* bias,
                        const float* gamma, int32_t want_stats, float* y, float* stats, float* pool_raw, uint8_t* idx,
                        void* stream);
/* The same pooling convolution when BatchNorm's statistics are known BEFORE it runs (seld_first_stage_bn): no y, no
 * statistics, and BatchNorm + ReLU + the stage's Dropout (model.py:279-282) are applied to the window value in the
 * epilogue: out = Dropout(relu(a raw + b)) with the mask seld_dropout_fwd draws for (seed, offset, state) on `out`
 * (drop_p = 0: none).  idx as above; pool_raw (same size, must be allocated) is written ONLY for channels with
 * gamma == 0 -- the one case in which seld_first_stage_bwd cannot recover xhat from out. */
int seld_hcq_first_pool_bn(const seld_conv_desc* desc, const float* x, const float* wpack, const float* bias,
                           const float* gamma, const float* beta, const float* mean, const float* invstd,
                           float drop_p, uint64_t seed, uint64_t offset, const uint64_t* state, float* pool_raw,
                           uint8_t* idx, float* out, void* stream);
/* out (nullable) = Dropout(p)(pooled) in the same pass, the mask seld_dropout_fwd would draw for (seed, offset, state) */
/* ------------------------------------------------------------------------------------------
 * First stage without its convolution output (csrc/first_stage.hip): conv3x3 on the 8-channel network input ->
 * BatchNorm2d (batch statistics) -> ReLU -> MaxPool2d(8, 1) -> Dropout, model.py:273-283.  The convolution is linear in
 * the input, so BatchNorm's statistics follow from the input's second moments (gram: 80 x 80 doubles, [G; s], count at
 * [72][72]) and the weights, and the backward pass needs the pooled-size tensors and x only.
 *   forward : seld_first_stage_gram -> seld_first_stage_bn (mean, invstd, running statistics, W G) ->
 *             seld_hcq_first_pool_bn (the stage's output + window row; three launches + two small folds in all)
 *   backward: seld_first_stage_bwd (dgamma, dbeta, component weight gradients; reproducible: no atomics)
 * Shapes: Cin = 8, 3x3 'same' stride 1, H % 8 == 0, W % 64 == 0 (backward: Cout in {64, 128, 192}; 64 only for algebra 1 / 4); workspace queries
 * return 0 otherwise. */
size_t seld_first_stage_gram_workspace(const seld_conv_desc* desc);
int seld_first_stage_gram(const seld_conv_desc* desc, const float* x, void* workspace, size_t workspace_bytes, void* stream);
int seld_first_stage_bn(const seld_conv_desc* desc, const float* const w[8], const float* bias, const double* gram, float eps,
                        float momentum, float* mean, float* invstd, float* running_mean, float* running_var,
                        int64_t* num_batches_tracked, float* wg /* (Cout, 72) or NULL */, void* stream);
size_t seld_first_stage_bwd_workspace(const seld_conv_desc* desc);
int seld_first_stage_bwd(const seld_conv_desc* desc, const float* x, const float* dout, const float* out, const float* raw,
                         const uint8_t* idx, const float* mean, const float* invstd, const float* gamma, const float* beta,
                         const float* bias, const double* gram, const float* wg, float* dgamma, float* dbeta,
                         float* const dw[8], float drop_p, void* workspace, size_t workspace_bytes, void* stream);

int seld_bn_pool_finish(const float* raw, int32_t N, int32_t C, int32_t S /* pooled H * W */, const float* mean,
                        const float* invstd, const float* gamma, const float* beta, float* pooled, float p, uint64_t seed,
                        uint64_t offset, const uint64_t* state, float* out, void* stream);

/* One-pass training-mode backward of the gate (adds into red[4C]; SELD_EUNSUPPORTED when S % 4 != 0 or N*S > 16384). */
int seld_gate_bwd_fused(const float* dy, const float* yf, const float* yg, int32_t N, int32_t C, int32_t S,
                        const float* mean_f, const float* invstd_f, const float* gamma_f, const float* beta_f,
                        const float* mean_g, const float* invstd_g, const float* gamma_g, const float* beta_g,
                        const float* mask, float* red, float* dyf, float* dyg, void* stream);

/* ------------------------------------------------------------------------------------------
 * Quaternion / dual-quaternion convolution by the 8-multiplication Hamilton product (csrc/hcq_conv.hip).
 * Same mathematics as seld_hc_conv_fwd / _bwd_data (quaternion_ops.py:125-147, dual_quaternion_ops.py:111-153):
 * the bilinear map w (x) x has rank 8, so the convolution is evaluated as 8 real GEMMs of a quarter of the K extent
 * on sums of two components (3 x 8 = 24 sub-products for the dual quaternion instead of 48) and the results
 * recombined in the epilogue; fp32 results differ from the block-matrix evaluation by rounding only.
 * Takes 'same' stride-1 convolutions with 1x1, 1x3 (any dilation) or 3x3 taps, row length a multiple of 64,
 * algebra 4 or 8, Cout/A a multiple of 16 (dual quaternion also 24).
 *   seld_hcq_pack_floats  size of the packed weight-form buffer (0: shape not taken, use seld_hc_conv_*)
 *   seld_hcq_pack         weight forms in MFMA fragment order, once per optimiser step and direction
 *                         (mode 0 forward, 1 data gradient); npair = 2 packs two convolutions of the same input
 *                         (conv1_filter | conv1_gate, conv2_skip | conv2_residual, model.py:121-132) for one launch:
 *                         two outputs (forward) or the sum of the two data gradients (mode 1)
 *   seld_hcq_conv         y[s] = W_s (x) x [+ bias][+ addend][statistics]  (mode 0), dx = data gradient (mode 1, x = dy)
 * ------------------------------------------------------------------------------------------ */
size_t seld_hcq_pack_floats(const seld_conv_desc* desc, int32_t mode, int32_t npair);
int seld_hcq_pack(const seld_conv_desc* desc, int32_t mode, int32_t npair, const float* const wA[8],
                  const float* const wB[8], float* wpack, void* stream);
/* mode 0: y[s] = W_s (x) x for s < npair.  mode 1: y[0] = dgrad(x, W_A) [+ dgrad(x2, W_B) if npair == 2]. */
int seld_hcq_conv(const seld_conv_desc* desc, int32_t mode, int32_t npair, const float* x, const float* x2,
                  const float* wpack, float* const y[2], const float* const bias[2], const int32_t epilogue[2],
                  const float* const addend[2], float* const stats[2], void* stream);
/* Every layer's weight forms in ONE launch per optimiser step: the caller builds a table of entries
 * (seld_hcq_pack_entry fills one entry of seld_hcq_pack_entry_bytes() bytes in host memory), copies it to the device
 * once, and calls seld_hcq_pack_table(table, entries, floats of the largest entry) after every weight update. */
int seld_hcq_kernel_label(const seld_conv_desc* desc, int32_t mode, int32_t npair, char* buf, int32_t buflen);
size_t seld_hcq_pack_entry_bytes(void);
int seld_hcq_pack_entry(const seld_conv_desc* desc, int32_t mode, int32_t npair, const float* const wA[8],
                        const float* const wB[8], float* wpack, void* entry_host);
int seld_hcq_pack_table(const void* table_dev, int32_t nentries, int64_t max_floats, void* stream);
/* the same, balanced over the entries: starts_dev[e] = first 256-float block of entry e (prefix sums of
 * ceil(floats_e / 256)), starts_dev[nentries] = total_blocks */
int seld_hcq_pack_flat(const void* table_dev, const int32_t* starts_dev, int32_t nentries, int32_t total_blocks,
                       void* stream);

/* Weight gradient on the fast product (csrc/hcq_wgrad.hip): dW = sum over positions of dy (x) conj(x) is again a Hamilton
 * product per (output block channel, input block channel, tap): 8 (24 for the dual quaternion) real sub-products
 * instead of 16 (48).  dwA[c] += gradient of conv(x; W_A) given dyA; npair == 2: also dwB[c] += that of a second
 * convolution of the same input given dyB (conv1_filter | conv1_gate, conv2_skip | conv2_residual).  Accumulating (the
 * caller's buffers are FlatAdam's gradient slices, train.py:557); no bias gradient (use seld_hc_conv_bwd_weight_acc).
 * seld_hcq_wgrad_supported: 1 if the shape is taken ('same' stride-1 1x1 / 1x3 / 3x3, row length a multiple of 32). */
int seld_hcq_wgrad_supported(const seld_conv_desc* desc, int32_t npair);
int seld_hcq_wgrad_label(const seld_conv_desc* desc, int32_t npair, char* buf, int32_t buflen);
int seld_hcq_wgrad_acc(const seld_conv_desc* desc, int32_t npair, const float* x, const float* dyA, const float* dyB,
                       float* const dwA[8], float* const dwB[8], void* stream);

/* The dual-quaternion weight gradient with 24 instead of 48 block products on the row-chunk GEMM (csrc/hcq_wgrad_row.hip;
 * dual_quaternion_ops.py:122-153 differentiated).  seld_hcq_wgrad_row_workspace: bytes of fp32 scratch for (desc, npair),
 * 0 = shape not taken.  The scratch must be zero before the first call; every call hands it back zeroed (one buffer per
 * stream serves all layers).  npair == 2: two convolutions of the same input (dwB[c] += ...), one launch family. */
size_t seld_hcq_wgrad_row_workspace(const seld_conv_desc* desc, int32_t npair);
int seld_hcq_wgrad_row_label(const seld_conv_desc* desc, int32_t npair, char* buf, int32_t buflen);
int seld_hcq_wgrad_row_acc(const seld_conv_desc* desc, int32_t npair, const float* x, const float* dyA, const float* dyB,
                           float* const dwA[8], float* const dwB[8], void* workspace, size_t workspace_bytes, void* stream);

/* SELD_DETERMINISTIC=1 (read with the other switches, seld_env_reload): reductions that are normally split over workgroups and
 * folded with float atomics -- BatchNorm statistics (seld_channel_stats), the two-pass BatchNorm / gate / pooling backward
 * reductions, linear-layer weight and bias gradients, the loss sum, the position splits of seld_hc_conv_bwd_weight* -- run as
 * ONE ordered chain per output element.  The host mirror then also keeps statistics out of the convolution epilogues and
 * takes the weight gradient below for every layer the grouped kernels do not take.
 * seld_hc_conv_bwd_weight_det: reproducible dw[c] += weight gradient of any convolution (real / quaternion / dual quaternion:
 * quaternion_ops.py:131-147, dual_quaternion_ops.py:122-153 differentiated), dbias += (nullable); workspace: Cout * Cin *
 * kh * kw floats.  Needs SELD_DETERMINISTIC set (SELD_EUNSUPPORTED otherwise). */
size_t seld_hc_conv_bwd_weight_det_workspace(const seld_conv_desc* desc);
int seld_hc_conv_bwd_weight_det(const seld_conv_desc* desc, const float* x, const float* dy, float* const dw[8], float* dbias,
                                void* workspace, size_t workspace_bytes, void* stream);

/* Weight gradients of a LIST of dual-quaternion convolutions in one grouped, persistent launch per shape family
 * (csrc/hcq_wgrad_grp.hip): dw[c] += d loss / d W_c for every job, i.e. dual_quaternion_conv
 * (dual_quaternion/dual_quaternion_ops.py:111-153) differentiated w.r.t. its eight component weights, which the reference
 * leaves to autograd through F.convNd (dual_quaternion_ops.py:153).  24 block products per layer; no atomics and a fixed
 * summation order: the result is reproducible from run to run.  x / dy: input and output-gradient tensors of the layer
 * (contiguous NCHW / NCT, desc.N images); dw: the eight component gradient tensors (Cout/8, Cin/8, kh, kw), accumulated into.
 * Shapes taken: algebra 8, 'same' stride-1 layers with rows of >= 128 positions (a multiple of 16) and
 * (Cout/8, Cin/8, kernel) = (48, 24, 1x3), (24, 48, 1x1), (24, 24, 3x3), (48, 48, 1x3): families 0..3, one persistent launch
 * per family present in the list.  seld_hcq_wgrad_group_workspace returns the scratch bytes (0: a job is not taken -- use the per-layer entry
 * points); the scratch needs no initialisation. */
typedef struct seld_wgrad_job {
    seld_conv_desc desc;
    const float* x;
    const float* dy;
    float* dw[8];
} seld_wgrad_job;
int seld_hcq_wgrad_group_family(const seld_conv_desc* desc);      /* 0..3 as listed above, -1 = not taken */
size_t seld_hcq_wgrad_group_workspace(const seld_wgrad_job* jobs, int32_t njobs);
int seld_hcq_wgrad_group(const seld_wgrad_job* jobs, int32_t njobs, void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * Elementwise / pooling / dropout   (torch.nn.ReLU/Tanh/MaxPool/Dropout at model.py:175-202,
 * 280-282, 449-451)
 * ------------------------------------------------------------------------------------------ */
int seld_act_fwd(const float* x, int64_t n, int32_t act, float* y, void* stream);
int seld_act_bwd(const float* dy, const float* y, int64_t n, int32_t act, float* dx, void* stream);

/* max pool over windows (ph, pw), stride = window, floor mode; x (NC, H, W) -> y (NC, H/ph, W/pw).
 * idx (uint8, nullable) receives the argmax position inside the window (row-major), as torch
 * keeps indices for backward. */
int seld_maxpool_fwd(const float* x, int64_t NC, int32_t H, int32_t W, int32_t ph, int32_t pw,
                     float* y, uint8_t* idx, void* stream);
int seld_maxpool_bwd(const float* dy, const uint8_t* idx, int64_t NC, int32_t H, int32_t W,
                     int32_t ph, int32_t pw, float* dx, void* stream);

/* dropout with a Philox-4x32-10 counter RNG: element i is kept iff u(seed, offset + i/4)[i%4] >= p.
 * `per_channel` (Dropout1d): one decision per (n, c) row of length S.  y = x * keep / (1 - p). */
/* `state` (nullable): device-resident step state (seld_step_begin below); state[0] is added to `offset`, so a launch
 * recorded in a HIP graph draws fresh numbers at every replay. */
int seld_dropout_fwd(const float* x, int64_t n, float p, uint64_t seed, uint64_t offset, const uint64_t* state,
                     float* y, void* stream);
int seld_dropout_mask_rows(int64_t rows, float p, uint64_t seed, uint64_t offset, const uint64_t* state,
                           float* mask, void* stream);

/* y = a + b ; y += b */
int seld_add(const float* a, const float* b, int64_t n, float* y, void* stream);
int seld_accumulate(float* dst, const float* src, int64_t n, void* stream);   /* dst += src */

/* ------------------------------------------------------------------------------------------
 * Multi-head self attention core (model.py:39-48): out = softmax(q k^T / sqrt(hd)) v, flash style
 * (the T x T energy tensor is never materialised).
 * q, k, v, out: (N, H*hd, T) -- the layout the 1x1 Conv1d projections of model.py:20-22 produce --
 * with head h on channels [h*hd, (h+1)*hd) (model.py:35-37).  hd <= 64.
 * lse (N, H, T): log-sum-exp of the scaled scores, saved for backward.
 * ------------------------------------------------------------------------------------------ */
int seld_mha_fwd(const float* q, const float* k, const float* v, int32_t N, int32_t T, int32_t H, int32_t hd,
                 float* out, float* lse, void* stream);
size_t seld_mha_bwd_workspace(int32_t N, int32_t T, int32_t H);
int seld_mha_bwd(const float* q, const float* k, const float* v, const float* out, const float* dout,
                 const float* lse, int32_t N, int32_t T, int32_t H, int32_t hd,
                 float* dq, float* dk, float* dv, void* workspace, size_t workspace_bytes, void* stream);
/* Self-attention on ONE projected tensor qkv (N, 3E, T) = [values | keys | queries] along the channels -- the three 1x1
 * projections of model.py:31-33 applied to the same input run as one convolution with the three weights stacked; dqkv has
 * the same layout, so one data-gradient and one weight-gradient launch follow.  Same arithmetic as seld_mha_fwd /
 * seld_mha_bwd on the three slices.  Matrix-core kernels only: SELD_EUNSUPPORTED unless seld_mha_packed_ok(T, hd) != 0
 * (hd in {16, 32, 48, 64}, T % 16 == 0).  workspace as seld_mha_bwd_workspace. */
int seld_mha_packed_ok(int32_t T, int32_t hd);
int seld_mha_fwd_packed(const float* qkv, int32_t N, int32_t T, int32_t H, int32_t hd, float* out, float* lse, void* stream);
int seld_mha_bwd_packed(const float* qkv, const float* out, const float* dout, const float* lse, int32_t N, int32_t T,
                        int32_t H, int32_t hd, float* dqkv, void* workspace, size_t workspace_bytes, void* stream);

/* (N, C, T) <-> (N, T, C) transposes (the permutes of model.py:30-37, 220-222, 318) */
int seld_transpose_nct_ntc(const float* x, int32_t N, int32_t C, int32_t T, float* y, void* stream);
int seld_transpose_ntc_nct(const float* x, int32_t N, int32_t T, int32_t C, float* y, void* stream);

/* ------------------------------------------------------------------------------------------
 * Loss (train.py:186-204): loss = w_sed * BCELoss(sed, t_sed) + w_doa * MSELoss(doa, t_doa), mean
 * reductions (train.py:498-499); sed/doa are the model OUTPUTS (after Sigmoid / Tanh).
 * target is (rows, n_sed + n_doa) row-major as produced by the preprocessing (train.py:191-192).
 * WRITES the scalar to loss[0] (workgroup sums added in a fixed order by the last workgroup to finish: no pre-zeroing,
 * run-to-run identical; the scratch is a per-device static of the library, so evaluations on ONE device must not overlap
 * on different streams) and, if non-null, writes dloss/dsed and dloss/ddoa (torch.nn.BCELoss semantics: logs clamped at
 * -100, backward denominator >= 1e-12).
 * ------------------------------------------------------------------------------------------ */
int seld_loss_fwd_bwd(const float* sed, const float* doa, const float* target,
                      int64_t rows, int32_t n_sed, int32_t n_doa, float w_sed, float w_doa,
                      float* loss, float* dsed, float* ddoa, void* stream);

/* ------------------------------------------------------------------------------------------
 * Adam (torch.optim.Adam defaults, train.py:502) over ONE flat fp32 buffer that holds every
 * parameter; grads/exp_avg/exp_avg_sq are parallel flat buffers.  step is 1-based.
 * ------------------------------------------------------------------------------------------ */
int seld_adam_flat(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                   float lr, float beta1, float beta2, float eps, float weight_decay, int32_t step,
                   float grad_scale, void* stream);

/* Device-resident step state, so that one training step (train.py:552-560) can be recorded once as a HIP graph and
 * replayed: 4 x uint64 = { Philox base added to every dropout offset, optimiser step (1-based), learning rate (float
 * bits in the low word), Philox draws per step }.
 * seld_step_begin replaces `optimizer.zero_grad()` (train.py:552): zeroes the flat gradient buffer (n floats, 16-byte
 * aligned) and, if state is non-null, advances state[1] += 1.
 * seld_adam_flat_state is seld_adam_flat with step = state[1] and lr = state[2]; as the last launch of a step it also
 * moves the Philox base past the step's draws, state[0] += state[3]. */
int seld_step_begin(float* flat_grad, int64_t n, uint64_t* state, void* stream);
int seld_adam_flat_state(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                         float beta1, float beta2, float eps, float weight_decay, float grad_scale,
                         uint64_t* state, void* stream);

/* ------------------------------------------------------------------------------------------
 * STFT magnitude / phase (utility_functions.py:129-155 = scipy.signal.stft(window='hamming',
 * boundary='zeros', padded=True) -> abs/angle -> drop DC bin -> drop last frame).
 * x (C, L) fp32; out (C or 2C, nperseg/2, frames-1) fp32, phase channels after magnitude channels.
 * nperseg must be a power of two <= 4096 (the reference uses 512).
 * ------------------------------------------------------------------------------------------ */
int seld_stft_frames(int32_t L, int32_t nperseg, int32_t noverlap);   /* frames AFTER the cut */
int seld_stft_magphase(const float* x, int32_t C, int32_t L, int32_t nperseg, int32_t noverlap,
                       int32_t output_phase, float* out, void* stream);
/* The same with every argument of spectrum_fast (utility_functions.py:129-130): cut_dc = 0 keeps all nperseg/2 + 1
 * bins, cut_last_timeframe = 0 keeps the last frame; `window` (nullable, device, nperseg floats) = window values
 * already divided by their sum (scipy's scaling='spectrum'), null = periodic Hamming (window='hamming').
 * out (C or 2C, nperseg/2 + 1 - cut_dc, seld_stft_frames_ex(...)). */
int seld_stft_frames_ex(int32_t L, int32_t nperseg, int32_t noverlap, int32_t cut_last_timeframe);
int seld_stft_magphase_ex(const float* x, int32_t C, int32_t L, int32_t nperseg, int32_t noverlap,
                          int32_t output_phase, int32_t cut_dc, int32_t cut_last_timeframe,
                          const float* window, float* out, void* stream);

/* ------------------------------------------------------------------------------------------
 * Dataset normalisation, in place on a resident predictor array x (items, channels, hw) fp32
 * (SURVEY 8(f) N1).
 * seld_dq_unit_norm replaces train.py:257-275 (and its copies for the validation / test arrays,
 *   277-308): channels 0..7 of every position are one dual quaternion (q, p); p <- p - (q.p/|q|^2) q,
 *   q <- q/|q|.  channels >= 8; further channels are left untouched.  Same operation order as the
 *   reference's torch expressions, every operation correctly rounded (a zero q yields NaN there and here).
 * seld_group_standardize replaces train.py:341-349 (and 350-405 for the other arrays / groups):
 *   g = x[:, c0:c1]; g <- (g - mean(g)) / std(g) with one scalar mean and one population std over the
 *   whole group.  work: 3 doubles of device scratch (cleared by the call); mean_std: 2 device floats
 *   receiving the float32 mean and std that were applied, or NULL.  An empty group is SELD_EINVAL.
 * ------------------------------------------------------------------------------------------ */
int seld_dq_unit_norm(float* x, int64_t items, int32_t channels, int64_t hw, void* stream);
int seld_group_standardize(float* x, int64_t items, int32_t channels, int32_t c0, int32_t c1, int64_t hw,
                           double* work, float* mean_std, void* stream);

/* ------------------------------------------------------------------------------------------
 * Post-processing + test metrics of evaluate_test (train.py:84-130) for a batch of
 * recordings whose outputs are resident: sed (clips, frames, classes*overlaps), doa (clips, frames,
 * 3*classes*overlaps), target (clips, frames, 4*classes*overlaps) = [activity | location] as the
 * reference's joint target.  One call replaces, per recording, gen_submission_list_task2
 * (utility_functions.py:184-210) on prediction and target, location_sensitive_detection
 * (metrics.py:123-182), segment_labels (Dcase21_metrics.py:239-278) and
 * SELDMetrics.update_seld_scores (Dcase21_metrics.py:51-154), and ADDS to
 *   counters[13] = { TP, FP, FN (L3DAS21, summed as train.py:124-126) ;
 *                    _TP, _FP, _FN, _S, _D, _I, _Nref, _DE_TP, _DE_FP, _DE_FN of SELDMetrics }
 *   total_de[1]  = SELDMetrics._total_DE
 * (device memory, zeroed by the caller before the first recording).  The scores of train.py:131-150 /
 * compute_seld_scores are a dozen scalar operations on these and stay on the host.
 * frames > num_frames is SELD_EINVAL; overlaps > 3, classes*overlaps > 64 or frames_per_block > 64 are
 * SELD_EUNSUPPORTED (the reference uses 14 x 3 and 10).
 * ------------------------------------------------------------------------------------------ */
#define SELD_METRIC_COUNTERS 13
int seld_metrics_accumulate(const float* sed, const float* doa, const float* target, int32_t clips, int32_t frames,
                            int32_t num_frames, int32_t classes, int32_t overlaps, float max_loc_value,
                            double spatial_threshold, double doa_threshold, int32_t frames_per_block,
                            int64_t* counters, double* total_de, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SELD_HIP_H */
