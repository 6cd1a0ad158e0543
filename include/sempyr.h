/*
 * sempyr.h — C ABI of libsempyr_hip.so: the MI355X (gfx950) operator backend for the
 * Semantic-Pyramid GAN training / synthesis hot path.
 *
 * The reference (tomer1727/Semantic_Pyramid_for_image_generation) has no FFI of its own:
 * its operator backend is whatever `torch.nn` dispatches to (SURVEY.md §8b "lower face").
 * Every entry point below therefore cites the reference call site whose torch operator it
 * replaces (file:line relative to the reference root).
 *
 * Conventions (all entry points):
 *   - return 0 = OK; <0 = invalid descriptor / unsupported shape (never a silent fallback);
 *     >0 = hipError_t of a failed launch. sempyr_last_error_string() describes the last failure.
 *   - plain pointers and sizes only; all device buffers (workspace included) are owned by the
 *     caller; nothing is allocated or freed here; no host synchronisation; kernels are enqueued
 *     on the hipStream_t passed as `void* stream` (0 = the null stream).
 *   - tensors are fp32, NCHW, contiguous. Element counts must be < 2^31; the operands of the convolution family must in
 *     addition be smaller than 2 GiB each (2^29 elements: their gathers are buffer loads with 32-bit byte offsets; checked).
 *   - re-entrant (the autograd worker thread calls the backward entry points); the only state is the per-device address
 *     of sempyr_gp_fwd_static's scratch array and the optional launch profiler (sempyr_profile_*).
 */
#ifndef SEMPYR_H
#define SEMPYR_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SEMPYR_ERR_INVALID (-1)
#define SEMPYR_ABI_VERSION 1

int sempyr_version(void);
const char* sempyr_last_error_string(void);

/* ------------------------------------------------------------------------------------------
 * Convolution family (implicit-GEMM on v_mfma_f32_32x32x2_f32).
 *
 * One descriptor describes a torch.nn.Conv2d: input B x Cin x Hin x Win (optionally seen through
 * a fused nearest x2 upsample, ups=1), weight Cout x Cin x kh x kw, stride, padding.
 * A torch.nn.ConvTranspose2d(Cin_t, Cout_t) is the *dgrad* of the Conv2d(Cout_t -> Cin_t) with the
 * same weight tensor, so it uses the same three primitives with the roles swapped.
 *
 * Replaces: nn.Conv2d / nn.ConvTranspose2d forward and (double-)backward at
 *   generator.py:23-25,115,119 ; discriminator.py:37,98,101,115 ; classifier.py:33,37-40,43
 *   (torchvision resnet18 convs and fc) ; featuresCreatorGen.py:13,67 ; and
 *   nn.UpsamplingNearest2d(2) generator.py:44,78 when ups=1.
 * ---------------------------------------------------------------------------------------- */
typedef struct sempyr_conv_desc {
    int32_t B, Cin, Hin, Win;   /* physical input (pre-upsample) */
    int32_t Cout, kh, kw;
    int32_t stride, pad;
    int32_t ups;                /* 1: the conv reads nearest-x2-upsampled input (2Hin x 2Win) */
    int32_t math;               /* SEMPYR_MATH_F32 (0, default): exact fp32 products on v_mfma_f32_32x32x2_f32.
                                 * SEMPYR_MATH_BF16X3 (opt-in): where the launch's kernel has the variant (3x3 / stride 1 / pad 1
                                 * FWD and DGRAD on 64-row tiles, channels % 16 == 0) every fp32 operand is split into two bf16
                                 * values and a product is hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_bf16 with fp32
                                 * accumulation (relative error ~2^-17 per product); every other launch computes in fp32.
                                 * Tensors, packed weights and workspace are fp32 either way. */
} sempyr_conv_desc;

enum { SEMPYR_MATH_F32 = 0, SEMPYR_MATH_BF16X3 = 1 };

enum { SEMPYR_CONV_FWD = 0, SEMPYR_CONV_DGRAD = 1, SEMPYR_CONV_WGRAD = 2 };

#define SEMPYR_CONV_MAX_PHASES 4

typedef struct sempyr_conv_plan {
    int32_t mode;
    int32_t Ho, Wo;                 /* conv output spatial dims */
    int32_t nphases;                /* 1 for FWD/WGRAD; stride^2 for DGRAD */
    int32_t K[SEMPYR_CONV_MAX_PHASES];      /* GEMM reduction length per phase (0 = phase has no taps) */
    int32_t Kpad[SEMPYR_CONV_MAX_PHASES];   /* padded to the k-chunk */
    int32_t tab_off[SEMPYR_CONV_MAX_PHASES];/* offset (entries) of the phase in the k-table */
    int32_t tab_entries;            /* total k-table entries (each = 2 int32 gather + 1 int32 weight src) */
    int32_t Mpad;                   /* packed-weight row length */
    int64_t wpack_floats;           /* size of the packed weight buffer (FWD, DGRAD), 0 for WGRAD */
    int64_t workspace_bytes;        /* split-K slabs */
    int32_t splits[SEMPYR_CONV_MAX_PHASES];
    int32_t cfg;                    /* tile configuration chosen */
    int32_t stats_granule;          /* FWD: pixels per group of the BatchNorm-statistics epilogue (sempyr_conv2d_fwd_stats);
                                       0 = this shape has none (split-K, RGB heads, shapes the 3x3 patch kernel does not take) */
} sempyr_conv_plan;

/* Shape inference + launch plan (host only, no GPU needed). */
int sempyr_conv2d_plan(const sempyr_conv_desc* d, int mode, sempyr_conv_plan* plan);

/* Host-side fill of the gather table (2*tab_entries int32: channel offset, packed dy/dx) and of the
 * weight-source table (tab_entries int32) for FWD or DGRAD. WGRAD uses the FWD gather table. */
int sempyr_conv2d_tables(const sempyr_conv_desc* d, int mode, int32_t* ktab_host, int32_t* wsrc_host);

/* Re-layout torch weights (Cout,Cin,kh,kw) into the k-major GEMM operand used by FWD / DGRAD. */
int sempyr_conv2d_pack_weights(const sempyr_conv_desc* d, int mode, const float* w,
                               const int32_t* wsrc_dev, float* wpacked, void* stream);

/* sempyr_conv2d_pack_weights for many (descriptor, mode, weight) items in one launch: every packed layout of a module is
 * refreshed right behind its optimizer step (train_model.py:117,143) instead of one small launch in front of each
 * convolution of the next pass. `items` is a HOST array (SEMPYR_PACK_MAX_ITEMS per launch). */
#define SEMPYR_PACK_MAX_ITEMS 64
typedef struct sempyr_pack_item {
    sempyr_conv_desc desc;
    int32_t mode;
    const float* w;
    const int32_t* wsrc_dev;
    float* wpacked;
} sempyr_pack_item;
int sempyr_conv2d_pack_weights_many(const sempyr_pack_item* items, int n_items, void* stream);

/* y[B,Cout,Ho,Wo] = act(scale[Cout] * conv(x, w) + bias[Cout] + res[B,Cout,Ho,Wo]); scale, bias and res are each optional
 * (NULL = absent). scale + bias fold an eval-mode BatchNorm2d into the epilogue and res the identity branch of a ResNet
 * BasicBlock (the frozen classifier, classifier.py:34-40 under .eval(); the generator under eval_model.py). They are
 * rejected for Cout <= 4 and together with y_pooled. res_half != 0: res is [B,Cout,Ho/2,Wo/2] and is read through a nearest
 * x2 upsample (the 1x1 skip branch of an upsampling block computed before the upsample, generator.py:44,60-62). act is a SEMPYR_ACT_* code applied in the
 * kernel epilogue (`conv -> LeakyReLU` of the critic stem discriminator.py:106-107, `conv -> Tanh` of the generator head
 * generator.py:116-120), slope the LeakyReLU slope. y_pooled (optional, [B,Cout,Ho/2,Wo/2]) also receives
 * AvgPool2d(2)(y) (discriminator.py:109), computed in the same epilogue when the launch is a single pass.
 * The critic stem (conv -> LeakyReLU -> AvgPool2d, discriminator.py:106-109) never needs its full-resolution activation
 * again except for the SIGN of every element (LeakyReLU's derivative), so with the fused pooled epilogue (single pass,
 * Cout > 4, Ho*Wo a multiple of 32; otherwise rejected):
 *   act_bits (optional, out, B*Cout*Ho*Wo/32 words): bit = 1 where the activated output is > 0, in "quad" pixel order
 *     (bit 4c+2a+b of plane (b, m) is fine pixel (2h+a, 2w+b) of coarse pixel c = h*(Wo/2)+w); y may then be NULL -- the
 *     268 MB activation of a 16-image batch is not written at all, 8.4 MB of bits are;
 *   mask_bits (optional, in, same layout; bias NULL, act NONE): the convolution's result is multiplied by
 *     (bit ? 1 : slope) before the pool, y NULL: the double backward of that stem (gradient penalty,
 *     train_model.py:69-76) as one launch instead of conv + full-resolution round trip + pool. */
int sempyr_conv2d_fwd(const sempyr_conv_desc* d, const float* x, const float* wpacked_fwd,
                      const int32_t* ktab_fwd_dev, const float* bias, const float* scale, const float* res,
                      int res_half, int act, float slope, float* y,
                      float* y_pooled, uint32_t* act_bits, const uint32_t* mask_bits,
                      void* workspace, size_t workspace_bytes, void* stream);

/* y = conv2d(x, w) (no bias / activation: the convolution in front of a training-mode BatchNorm2d, generator.py:22-27) that
 * also emits the BatchNorm statistics of y from its epilogue: tile_stats[(b*Cout + c) * G + g] = (sum, sum of squared
 * deviations from the group mean) as float2 over pixels [g*granule, (g+1)*granule) of plane (b, c), G = Ho*Wo / granule,
 * granule = sempyr_conv2d_plan().stats_granule (> 0 required).  sempyr_bn_stats_from_tiles turns them into the per-row
 * coefficients; the separate statistics pass over y (one full read) is gone. */
int sempyr_conv2d_fwd_stats(const sempyr_conv_desc* d, const float* x, const float* wpacked_fwd, const int32_t* ktab_fwd_dev,
                            float* y, float* tile_stats, void* stream);

/* gx[B,Cin,Hin<<ups,Win<<ups] = conv_transpose(gy, w) (+ bias[Cin] if bias != NULL: the bias of a
 * ConvTranspose2d whose forward this is). With ups=1 the result is the gradient in the upsampled
 * domain; reduce it with sempyr_pool2_down (scale 1). */
int sempyr_conv2d_dgrad(const sempyr_conv_desc* d, const float* gy, const float* wpacked_dgrad,
                        const int32_t* ktab_dgrad_dev, const float* bias, float* gx,
                        void* workspace, size_t workspace_bytes, void* stream);

/* The same input gradient for the critic stem (discriminator.py:101-109: Conv2d(3, 64, 3, 1, 1) -> LeakyReLU -> AvgPool2d(2)) from
 * what its forward kept: gy = 0.25 * up2(g_pooled) * (bit ? 1 : slope) is expanded on the fly from the gradient of the POOLED
 * output [B,Cout,Ho/2,Wo/2] and the sign bits sempyr_conv2d_fwd wrote (act_bits), so the full-resolution gradient never
 * exists in memory (train_model.py:69-76, the inner pass of the gradient penalty; train_model.py:151-156 the generator's
 * adversarial loss).  Needs the small-Cin 3x3 path (Cin <= 4, stride 1, pad 1) and Wo == 256. */
int sempyr_conv2d_dgrad_unpooled(const sempyr_conv_desc* d, const float* g_pooled, const uint32_t* act_bits, float slope,
                                 const float* wpacked_dgrad, const int32_t* ktab_dgrad_dev, float* gx, void* stream);

/* Weight and bias gradient of that stem, also from (g_pooled, act_bits): gw[64,3,3,3] (+)= sum gy * x, gbias[64] (+)= sum gy
 * with gy as above (gbias may be NULL).  Deterministic: per-workgroup partial sums in `workspace`
 * (sempyr_stem_wgrad_workspace reports the bytes; it fails for other shapes than Cin 3, Cout 64, 3x3 s1 p1, Win % 256 == 0, Hin even),
 * added in a fixed order.  Replaces un-pool + sempyr_conv2d_wgrad + sempyr_channel_sum for discriminator.py:105 in
 * loss.backward() (train_model.py:139) and in the penalty's double backward. */
int sempyr_stem_wgrad_workspace(const sempyr_conv_desc* d, size_t* bytes);
int sempyr_stem_wgrad(const sempyr_conv_desc* d, const float* g_pooled, const uint32_t* act_bits, float slope,
                      const float* x, float* gw, int accumulate_w, float* gbias, int accumulate_b,
                      void* workspace, size_t workspace_bytes, void* stream);

/* gw[Cout,Cin,kh,kw] = sum_{b,oy,ox} gy * x (deterministic split-K, fixed summation order); accumulate != 0: gw += that
 * sum (autograd's gradient accumulation for a weight used by several passes of one step — D(real), D(fake), D(x_hat) at
 * train_model.py:128-137 — folded into the combine pass instead of a separate add). */
int sempyr_conv2d_wgrad(const sempyr_conv_desc* d, const float* gy, const float* x,
                        const int32_t* ktab_fwd_dev, float* gw, int accumulate,
                        void* workspace, size_t workspace_bytes, void* stream);


/* ------------------------------------------------------------------------------------------
 * Normalisation: BatchNorm2d (train) / InstanceNorm2d(affine) / GroupNorm(1,C)
 * Tensors are seen as rows = N*C of len = H*W. `rowbuf` receives four per-row arrays
 * [mean_row | rstd_row | scale_row | shift_row] (each N*C floats); apply them with
 * sempyr_affine_act(x, mean_row, scale_row, shift_row, ...): y = (x - mean)*rstd*gamma + beta. Replaces nn.BatchNorm2d generator.py:28, and get_norm_layer
 * discriminator.py:16-26 (+ their autograd backward and double backward used by the gradient
 * penalty train_model.py:69-76).
 * ---------------------------------------------------------------------------------------- */
enum { SEMPYR_NORM_BATCH = 0, SEMPYR_NORM_INSTANCE = 1, SEMPYR_NORM_LAYER = 2 };
enum { SEMPYR_ACT_NONE = 0, SEMPYR_ACT_RELU = 1, SEMPYR_ACT_LRELU = 2, SEMPYR_ACT_TANH = 3 };

/* scratch: 2*N*C floats. running_mean/var/num_batches_tracked (BATCH only) may be NULL. rep >= 1: x stands for a tensor in
 * which every element occurs rep times (rep = 4: the nearest-x2-upsampled map of generator.py:44,60-62 whose 1x1 skip
 * convolution was taken before the upsample) -- only the unbiased running_var depends on it. */
int sempyr_norm_stats(int kind, const float* x, int N, int C, int HW, float eps,
                      const float* gamma, const float* beta,
                      float* running_mean, float* running_var, int64_t* num_batches_tracked, float momentum,
                      int rep, float* rowbuf, float* scratch, void* stream);

/* The statistics step of sempyr_norm_stats for SEMPYR_NORM_BATCH from the partial sums of sempyr_conv2d_fwd_stats instead of
 * a pass over x: tile_stats[(n*C + c) * (HW/granule) + g] = float2 (sum, sum of squared deviations from the group mean).
 * Writes the same rowbuf (mean, rstd, scale, shift per row) and running statistics (generator.py:23 BatchNorm2d in train
 * mode; `rep` as in sempyr_norm_stats). */
int sempyr_bn_stats_from_tiles(const float* tile_stats, int N, int C, int HW, int granule, float eps,
                               const float* gamma, const float* beta,
                               float* running_mean, float* running_var, int64_t* num_batches_tracked, float momentum,
                               int rep, float* rowbuf, void* stream);

/* sempyr_norm_stats + the apply pass y = act((x - mean)*rstd*gamma + beta) as one call; InstanceNorm2d rows of up to 16384
 * elements (every critic layer, discriminator.py:16-26) run as ONE launch that reads x from HBM once. scratch: 2*N*C floats. */
int sempyr_norm_fwd(int kind, const float* x, int N, int C, int HW, float eps,
                    const float* gamma, const float* beta,
                    float* running_mean, float* running_var, int64_t* num_batches_tracked, float momentum,
                    int act, float slope, float* y, float* rowbuf, float* scratch, void* stream);

/* eval-mode BatchNorm2d (classifier.py:34 with .eval(), train_model.py:153): per-row scale/shift
 * from the running statistics. */
int sempyr_bn_eval_coeffs(const float* gamma, const float* beta, const float* running_mean,
                          const float* running_var, float eps, int N, int C,
                          float* scale_row, float* shift_row, void* stream);

/* (gx, ggamma, gbeta) of y = act(norm(x)); y is needed only when act != NONE. scratch: 8*N*C floats.
 * ggamma / gbeta may be NULL. */
int sempyr_norm_bwd(int kind, const float* gy, const float* y, const float* x, const float* gamma,
                    const float* mean_row, const float* rstd_row, int N, int C, int HW,
                    int act, float slope, float* gx, float* ggamma, float* gbeta,
                    float* scratch, void* stream);

/* Backward of sempyr_norm_bwd's gx output: given u = dL/d(gx) returns dL/d(gy), dL/dx, dL/dgamma
 * (g_gamma may be NULL). scratch: 19*N*C floats. */
int sempyr_norm_bwd_bwd(int kind, const float* u, const float* gy, const float* y, const float* x,
                        const float* gamma, const float* mean_row, const float* rstd_row,
                        int N, int C, int HW, int act, float slope,
                        float* g_gy, float* g_x, float* g_gamma, float* scratch, void* stream);

/* ------------------------------------------------------------------------------------------
 * Elementwise (HBM-bound) kernels.
 * ---------------------------------------------------------------------------------------- */
/* y = act((x - sub_row[r])*scale_row[r] + shift_row[r] + res); sub optional, scale/shift both or neither, res optional.
 * Replaces norm apply + nn.ReLU/LeakyReLU/Tanh (generator.py:29,48,57,120; discriminator.py:39,56,
 * 99,102) and the residual adds generator.py:56,87 ; discriminator.py:63. */
int sempyr_affine_act(const float* x, const float* sub_row, const float* scale_row, const float* shift_row,
                      const float* res, float* y, int rows, int len, int act, float slope, void* stream);
/* y = act((x-sub)*scale+shift + (res-rsub)*rscale+rshift): two normalised branches summed and activated in one pass
 * (generator residual block: BatchNorm(conv2) + BatchNorm(skip) -> ReLU, generator.py:54-57). res_half_w > 0 (= the width
 * of a row of x): res has half the resolution of x and is read through a nearest x2 upsample -- the 1x1 skip convolution
 * of an upsampling block (generator.py:44,60-62) commutes with the upsample, so it runs on the low-resolution map. */
int sempyr_affine2_act(const float* x, const float* sub_row, const float* scale_row, const float* shift_row,
                       const float* res, const float* rsub_row, const float* rscale_row, const float* rshift_row,
                       float* y, int rows, int len, int act, float slope, int res_half_w, void* stream);
/* g' = gy*act'(y); gx = g'*scale_row (NULL -> 1), gres = g' ; either output may be NULL. */
int sempyr_affine_act_bwd(const float* gy, const float* y, const float* scale_row, float* gx, float* gres,
                          int rows, int len, int act, float slope, void* stream);
/* y = a*b : features[i]*masks[i] generator.py:126-129, train_model.py:219-221 */
int sempyr_mul(const float* a, const float* b, float* y, int64_t n, void* stream);
/* y = alpha*a + beta*b (b may be NULL) : gradient accumulation / scaling glue */
int sempyr_axpby(const float* a, const float* b, float alpha, float beta, float* y, int64_t n, void* stream);
/* Multi-tensor Adam step: optim.Adam(lr=2e-4, betas=(.5,.999)) train_model.py:273-274, .step() at :117,143 (eps 1e-8, no
 * weight decay, no amsgrad).  `tensors` is a HOST array (it travels in the kernel arguments, SEMPYR_ADAM_MAX_TENSORS per
 * launch); `step` is the 1-based step count shared by all listed tensors; grad is read as grad*grad_scale (1/N of a summed
 * data-parallel gradient) and overwritten with zeros when zero_grads != 0.  Operation order follows torch's foreach Adam:
 * m = lerp(m, g, 1-b1); v = v*b2 + (1-b2)*g*g; p -= lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps). */
#define SEMPYR_ADAM_MAX_TENSORS 64
typedef struct sempyr_adam_tensor {
    float* param;
    float* grad;
    float* exp_avg;
    float* exp_avg_sq;
    int64_t numel;
} sempyr_adam_tensor;
int sempyr_adam_step(const sempyr_adam_tensor* tensors, int n_tensors, int64_t step, double lr, double beta1, double beta2,
                     double eps, double grad_scale, int zero_grads, void* stream);
/* The same step for a caller that replays the launch from a captured HIP graph (kernel arguments are frozen at capture):
 * the two step-dependent scalars {-lr/(1-beta1^t), sqrt(1-beta2^t)} are read from device memory at run time.
 * sempyr_adam_scalars computes them on the host exactly as sempyr_adam_step does (doubles, one rounding to fp32);
 * hand them over with sempyr_store_words before each replay. */
int sempyr_adam_scalars(int64_t step, double lr, double beta1, double beta2, float* out2_host);
int sempyr_adam_step_dev(const sempyr_adam_tensor* tensors, int n_tensors, const float* step_scalars_dev, double beta1,
                         double beta2, double eps, double grad_scale, int zero_grads, void* stream);
/* dst_dev[0..n) = words_host[0..n) (n <= SEMPYR_STORE_MAX_WORDS), stream-ordered; the words travel in the kernel
 * arguments, so the host array may be reused at once.  Hands the per-iteration scalars of the loop body
 * (train_model.py:301,327 layer / crop draws, the Adam step scalars) to kernels that were captured into a graph. */
#define SEMPYR_STORE_MAX_WORDS 48
int sempyr_store_words(const uint32_t* words_host, int n, uint32_t* dst_dev, void* stream);
/* The feature masks of setCroppedMask / setMasksPart1 / setMasksPart2 (train_model.py:172-209) in one launch: mask i
 * ([rows][H][W] floats) = base_i everywhere except rows [x0,x1) x cols [y0,y1) = win_i.  spec_dev holds six int32 words
 * per mask {bits(base), bits(win), x0, x1, y0, y1} in DEVICE memory (so a captured graph follows the per-iteration
 * draws); `targets` is a HOST array. */
#define SEMPYR_MASKS_MAX 6
typedef struct sempyr_mask_target {
    float* mask;
    int32_t rows, H, W;
} sempyr_mask_target;
int sempyr_build_masks(const sempyr_mask_target* targets, int n, const int32_t* spec_dev, void* stream);
/* dst = [src_0 | src_1 | ...] (n <= SEMPYR_CONCAT_MAX contiguous float arrays; HOST tables): the real and the fake batch
 * as one critic batch (train_model.py:132-133 run as one pass when no layer mixes samples). */
#define SEMPYR_CONCAT_MAX 8
int sempyr_concat(const float* const* srcs_host, const int64_t* numels_host, int n, float* dst, void* stream);
/* dst[0..n) = value : grad_outputs=torch.ones_like(pred) train_model.py:73 */
int sempyr_fill_const(float* dst, float value, int64_t n, void* stream);

/* y = a + act(beta[c]), plus the running-statistics update (decay by 1 - momentum, num_batches_tracked += 1; buffers may
 * be NULL) of a training-mode BatchNorm2d applied to an all-zero input: conv(0) -> BatchNorm2d -> ReLU -> add of a generator
 * level whose mask is all zero (generator.py:126-129, quirk Q2) as one launch.  Bit for bit what the statistics pass over
 * the zero tensor, the apply pass and the add give.  Forward only. */
int sempyr_bn_zero_input_add(const float* a, const float* beta, float* y, int N, int C, int HW, int act, float slope,
                             float* running_mean, float* running_var, int64_t* num_batches_tracked, float momentum,
                             void* stream);
/* *out = ((w_0*x_0 + w_1*x_1 + ...) * s1) * s2 on device scalars, products and sums rounded one by one, left to right:
 * the loss arithmetic train_model.py:101-103,112,138 and :222 (`loss * (1/L) * 200`) in one launch. HOST tables. */
#define SEMPYR_LINCOMB_MAX 6
int sempyr_lincomb(const float* const* xs_host, const float* ws_host, int n, float s1, float s2, float* out, void* stream);
/* dst_i = sum_k src_i[k] (assign != 0) or dst_i += sum_k src_i[k] (sources added in table order; unused slots NULL) for a
 * HOST table of small tensors in one launch: autograd's gradient accumulation (`.backward()` train_model.py:116,142) for the
 * norm-layer affine parameters of a whole pass.  A destination may appear once per SEMPYR_ACCUM_MAX items. */
#define SEMPYR_ACCUM_MAX 32
#define SEMPYR_ACCUM_SRCS 4
typedef struct sempyr_accum_item {
    float* dst;
    const float* src[SEMPYR_ACCUM_SRCS];
    int64_t numel;
    int32_t assign;
} sempyr_accum_item;
int sempyr_accumulate_many(const sempyr_accum_item* items_host, int n, void* stream);
/* out = real + alpha_row[r]*(fake-real) : sample_line train_model.py:51-55 */
int sempyr_lerp_rows(const float* real, const float* fake, const float* alpha_row, float* out,
                     int rows, int len, void* stream);
/* transforms.Normalize train_model.py:240-245 : y=(x-mean_c)/std_c ; backward!=0: y = x/std_c */
int sempyr_normalize(const float* x, const float* mean_c, const float* std_c, float* y,
                     int N, int C, int HW, int backward, void* stream);
/* y[r][h][w] = scale*sum 2x2 : nn.AvgPool2d(2) (scale .25) discriminator.py:57,103 ; upsample backward (scale 1).
 * yref (optional, fine resolution, output of a ReLU/LeakyReLU): every x is first multiplied by act'(yref).
 * act_bits (optional, instead of yref): the sign bits sempyr_conv2d_fwd wrote for that output. */
int sempyr_pool2_down(const float* x, const float* yref, const uint32_t* act_bits, float* y, int rows, int H, int W, float scale,
                      int act, float slope, void* stream);
/* y[r][2h+a][2w+b] = scale*x[r][h][w] : nn.UpsamplingNearest2d(2) generator.py:44 ; avg-pool backward (scale .25).
 * yref (optional, fine resolution): the result is multiplied by act'(yref) -- the backward of
 * `LeakyReLU -> AvgPool2d` (discriminator.py:107-109) in one pass. row_partials (optional, [rows][h*w/512] floats, needs
 * h*w % 512 == 0): partial sums of y per row, so the bias gradient of the conv in front (a channel sum of y) needs no second
 * pass over the full-resolution tensor. act_bits (optional, instead of yref): sign bits as for sempyr_pool2_down. */
int sempyr_pool2_up(const float* x, const float* yref, const uint32_t* act_bits, float* y, int rows, int h, int w, float scale,
                    int act, float slope, float* row_partials, void* stream);
/* nn.MaxPool2d(3, 2, 1) of the ResNet18 stem classifier.py:36 ; idx = argmax offsets for the backward */
int sempyr_maxpool3s2_fwd(const float* x, float* y, int32_t* idx, int rows, int H, int W, void* stream);
int sempyr_maxpool3s2_bwd(const float* gy, const int32_t* idx, float* gx, int rows, int H, int W, void* stream);

/* ------------------------------------------------------------------------------------------
 * Loss-path reductions (deterministic). `scratch` = sempyr_reduce_scratch_floats() floats unless
 * stated. Scalar outputs/inputs live on the device (no host sync).
 * ---------------------------------------------------------------------------------------- */
int sempyr_reduce_scratch_floats(void);
/* *out = scale * mean(x) : d_loss_fn / g_loss_fn train_model.py:39-46 */
int sempyr_mean(const float* x, int64_t n, float scale, float* out, float* scratch, void* stream);
/* y[i] = scale * (*g) : backward of a mean */
int sempyr_fill_scalar(const float* g, float scale, float* y, int64_t n, void* stream);
/* *out = mean |a*m - b*m| (mask NULL -> |a-b|) : nn.L1Loss train_model.py:219-221,109 */
int sempyr_l1_mean(const float* a, const float* b, const float* mask, int64_t n, float* out, float* scratch, void* stream);
int sempyr_l1_mean_bwd(const float* a, const float* b, const float* mask, const float* gout, int64_t n,
                       float* ga, float* gb, void* stream);
/* norms[r] = ||grad[r,:]||_2 ; *gp = mean (norms-1)^2 : norm + one_mean_gp train_model.py:58-66. scratch (optional,
 * rows*SEMPYR_GP_SPLIT floats): long rows of a small batch are reduced by SEMPYR_GP_SPLIT workgroups each. */
#define SEMPYR_GP_SPLIT 48
int sempyr_gp_fwd(const float* grad, int rows, int len, float* norms, float* gp, float* scratch, void* stream);
/* sempyr_gp_fwd for callers that cannot allocate device memory (code under HIP graph capture, tight serving loops,
 * harnesses that forbid hipMalloc).  It has NO counterpart in the reference.  Same results as sempyr_gp_fwd called with a
 * scratch buffer, bit for bit, but the split partials live in a device array built into the library
 * (SEMPYR_GP_STATIC_MAX_ROWS * SEMPYR_GP_SPLIT floats = 48 KB per GPU, held until the process ends): the caller passes
 * inputs and outputs only, and the call performs no hipMalloc / hipFree (nor any allocating library routine).
 * rows > SEMPYR_GP_STATIC_MAX_ROWS is rejected with SEMPYR_ERR_INVALID.  All calls on one GPU share that GPU's array:
 * calls that overlap on one GPU (two streams, two threads) race -- serialise them.  Nothing in the array has to start at
 * zero (each call overwrites every partial it later reads).  The first call on a device looks up the array's device
 * address (hipGetSymbolAddress); make that first call outside stream capture.  The array used is the CURRENT device's: a
 * non-null `stream` must belong to the current device (checked: SEMPYR_ERR_INVALID otherwise). */
#define SEMPYR_GP_STATIC_MAX_ROWS 256
int sempyr_gp_fwd_static(const float* grad, int rows, int len, float* norms, float* gp, void* stream);
/* out = grad * (*gout) * (2/rows) * (norm-1)/norm : its backward (feeds the double backward) */
int sempyr_gp_bwd(const float* grad, const float* norms, const float* gout, int rows, int len, float* out, void* stream);
/* out[c] = scale * sum_{n,hw} x[n][c][hw] : conv bias gradient ; scratch = N*C floats ; accumulate != 0: out[c] += that */
int sempyr_channel_sum(const float* x, int N, int C, int HW, float scale, float* out, int accumulate, float* scratch,
                       void* stream);


/* ------------------------------------------------------------------------------------------
 * Input pipeline on the device: transforms.Resize(256) + CenterCrop(256) + ToTensor of the
 * reference's ImageFolder (train_model.py:233-238, eval_model.py:45-50) and the two transforms.Normalize maps that follow
 * (train_model.py:240-247,308-309) for ONE decoded image in one launch.
 *   src: uint8 [H][W][3] device memory.  htab / vtab: int32 device tables [crop_w][2 + hk] / [crop_h][2 + vk] =
 *   {first source index, taps, coefficients with 22 fractional bits} of PIL's bilinear ImagingResample for the columns / rows
 *   the crop keeps (host: imageio.resample_table restates Pillow's precompute_coeffs).  The kernel performs PIL's integer
 *   arithmetic (horizontal pass, clip to 8 bits, vertical pass, clip): pixels are bit-identical to PIL's.
 *   out01 (optional): the image in [0,1] (x / 255, correctly rounded), [3][crop_h][crop_w].
 *   out_a / out_b (optional): (x / 255 - mean[c]) / std[c] with mean_* / std_* = 3 floats each in HOST memory.
 * ---------------------------------------------------------------------------------------- */
int sempyr_resize_crop_normalize(const uint8_t* src, int H, int W, int crop_h, int crop_w,
                                 const int32_t* htab, int hk, const int32_t* vtab, int vk,
                                 const float* mean_a, const float* std_a, float* out_a,
                                 const float* mean_b, const float* std_b, float* out_b,
                                 float* out01, void* stream);

/* ------------------------------------------------------------------------------------------
 * Replay of a captured loop body (no counterpart in the reference: its loop body, train_model.py:300-349, is issued op by
 * op from Python).  sempyr_replay_create walks a captured hipGraph_t once (kernel / memset / 1-D device memcpy / empty
 * nodes and their dependency edges) and lays the DAG out on up to max_streams HIP streams; sempyr_replay_launch re-issues
 * every node with plain stream launches from native code -- stream 0 of the layout is the `stream` passed at launch, so the
 * replay is ordered after earlier work on it and later work on it is ordered after the whole replay.  The graph must
 * outlive the plan (its nodes own the kernel-argument copies).  One replay of a plan at a time.
 * stats (optional, 8 int32): kernels, memsets, memcpys, empty nodes, streams used, cross-stream waits, events, edges.
 * ---------------------------------------------------------------------------------------- */
typedef struct sempyr_replay sempyr_replay;
int sempyr_replay_create(void* hip_graph, int max_streams, sempyr_replay** out, int32_t* stats);
/* The same, with the first n_side_streams extra streams of the layout supplied by the caller (hipStream_t handles that must
 * outlive the plan); further ones come from a set the library keeps per device and shares between plans.  HIP maps streams
 * onto a few hardware queues (GPU_MAX_HW_QUEUES, 4 by default): a caller that replays on the very streams its eager path
 * uses keeps concurrently busy streams on distinct queues. */
int sempyr_replay_create_on(void* hip_graph, int max_streams, void* const* side_streams, int n_side_streams,
                            sempyr_replay** out, int32_t* stats);
int sempyr_replay_launch(sempyr_replay* plan, void* stream);
int sempyr_replay_destroy(sempyr_replay* plan);

/* ------------------------------------------------------------------------------------------
 * Measurement hooks (no counterpart in the reference; used by bench.py only).  Between begin and end
 * every implicit-GEMM launch is bracketed by HIP events on its own stream; end synchronises and returns
 * the summed kernel time, the launch count and the algorithmic FLOPs (2*M*N*K) of those launches.
 * ---------------------------------------------------------------------------------------- */
int sempyr_profile_begin(int max_launches);
/* launches the do-nothing kernel `sempyr_marker_kernel` on `stream`: brackets a region in a rocprofv3 kernel trace */
int sempyr_profile_marker(int tag, void* stream);
int sempyr_profile_end(double* kernel_ms, int64_t* launches, double* flops);
/* After sempyr_profile_end: per HBM-bound kernel that ran inside the bracket, one CSV line "kernel,launches,algorithmic_bytes,ms"
 * (algorithmic bytes per SURVEY.md 8d: 4 B x elements read once + written once; ms = summed HIP-event durations on the launch
 * streams).  No counterpart in the reference: measurement only (bench.py's `hbm` block, profiles/<round>_hbm_kernels.json). */
int sempyr_profile_hbm_report(char* out, int64_t cap);

#ifdef __cplusplus
}
#endif
#endif /* SEMPYR_H */
