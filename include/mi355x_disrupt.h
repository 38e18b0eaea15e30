/*
 * mi355x_disrupt.h -- C ABI of the MI355X (gfx950) hot path of the KSTAR disruption predictor.
 *
 * The reference (ZINZINBIN/Disruption-Prediciton-based-on-Multimodal-Deep-Learning) is pure
 * Python/PyTorch and defines no FFI; its boundary for this path is the nn.Module protocol
 * (SURVEY.md section 8b).  This header is the C-ABI layer underneath the Python mirror of that protocol:
 * every entry point names the reference construct it replaces (file:line, relative to the
 * reference root).  Conventions:
 *   - plain pointers and sizes only, no torch / HIP C++ types: `stream` is a hipStream_t passed as void*;
 *   - all data pointers are caller-owned DEVICE memory, fp32 unless stated;
 *   - returns 0 on success or a negative MD_ERR_* code; never throws, never allocates device memory
 *     (plans excepted: md_plan_create allocates host metadata only), never synchronises the device;
 *   - re-entrant across host threads and streams (backward is called from the autograd thread).
 *
 * Activation layout inside the library: channels-last [N][T][H][W][Cp], Cp = channels rounded up
 * to a multiple of 4 (16-byte pixels), pad channels always zero.  md_nchw_to_cl / md_cl_to_nchw
 * convert at the (B,C,T,H,W) boundary of the reference (src/dataset.py:229-230).
 */
#ifndef MI355X_DISRUPT_H
#define MI355X_DISRUPT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MD_OK 0
#define MD_ERR_BAD_SHAPE (-1)
#define MD_ERR_UNSUPPORTED (-2)
#define MD_ERR_WORKSPACE (-3)
#define MD_ERR_LAUNCH (-4)
#define MD_ERR_NULL (-5)

/* Library identity: returns the ABI version; *arch_out (optional) receives "gfx950". */
int md_version(const char** arch_out);

/* Arithmetic mode of the convolutions (process-wide, default 0):
 *   0  "bf16x3": unit-stride convolutions run on the LDS-patch kernels with every fp32 operand split into two
 *      bf16 halves (hi+lo) and three bf16 MFMAs per product, fp32 accumulation (~5e-6 relative error per conv);
 *   1  exact fp32: every convolution runs on v_mfma_f32_16x16x4_f32 (bitwise an fp32 fmaf chain, 1/16 the rate).
 * Packed weights depend on the mode: change it only between steps.  Returns the previous mode. */
int md_set_exact_fp32(int on);
int md_get_exact_fp32(void);
/* Test knob: n > 0 forces the persistent convolution kernels (weights resident in LDS, one workgroup walking many
 * boxes) to a grid of n workgroups, also on shapes too small to qualify by themselves; 0 restores the default
 * (one workgroup per CU, only where there are at least as many boxes).  Returns the previous value. */
int md_set_pers_grid(int n);
/* Test hook: 1 = weight gradients always on the first kernel form (the one that also reads the pre-split operand formats);
 * 0 = automatic (default).  Returns the previous setting. */
int md_set_wgrad_form(int first_form_only);

/* ------------------------------------------------------------------------------------------------
 * Convolution as implicit GEMM on the matrix cores (replaces nn.Conv3d fwd/bwd as used by
 * Conv3dBlock, src/models/R2Plus1D.py:44-51; covers both factors of SpatioTemporalConv :139-140,
 * :156-157 -- (1,k,k) spatial and (k,1,1) temporal are the same kernel with different taps).
 * ---------------------------------------------------------------------------------------------- */
typedef struct MdConvDesc {
  int32_t N, Ti, Hi, Wi, Cin;   /* input  (B, Cin, Ti, Hi, Wi) of the reference conv; N = batch         */
  int32_t To, Ho, Wo, Cout;     /* output (B, Cout, To, Ho, Wo)                                          */
  int32_t kt, kh, kw;           /* kernel_size                                                            */
  int32_t st, sh, sw;           /* stride                                                                 */
  int32_t pt, ph, pw;           /* padding                                                                */
} MdConvDesc;

/* "BN-on-read" view of an activation: value = leaky(x*scale[c] + shift[c], slope) applied while the
 * tensor is staged into LDS (fuses BatchNorm3d + LeakyReLU of Conv3dBlock.forward, R2Plus1D.py:56-58,
 * into the consumer).  scale == NULL means the tensor is used as stored. */
typedef struct MdActView {
  const float* data;   /* channels-last activation                                                       */
  const float* scale;  /* [Cp] or NULL                                                                   */
  const float* shift;  /* [Cp] or NULL                                                                   */
  float slope;         /* LeakyReLU negative slope                                                       */
} MdActView;

static inline int32_t md_cpad(int32_t c) { return (c + 3) & ~3; }

/* Packed-weight sizes (floats): forward GEMM operand [Cout16][Kp] and dgrad operand [Cin16][Kp']. */
size_t md_conv_wpack_fwd_floats(const MdConvDesc* d);
size_t md_conv_wpack_dgrad_floats(const MdConvDesc* d);
/* Pack the reference-layout weight (Cout,Cin,kt,kh,kw) into either operand (tiny kernel). */
int md_conv_pack_weights(const MdConvDesc* d, const float* w, float* wpack_fwd, float* wpack_dgrad, void* stream);
/* The same for n units in as few launches as possible (descs: array of n descriptors; wpack_fwd[i] / wpack_dgrad[i] may be NULL). */
int md_conv_pack_weights_batch(int32_t n, const MdConvDesc* descs, const float* const* w, float* const* wpack_fwd,
                               float* const* wpack_dgrad, void* stream);

/* Number of row-blocks the forward kernel uses = rows of the BatchNorm partial-statistics buffer. */
int32_t md_conv_fwd_stat_blocks(const MdConvDesc* d);

/* y_raw = conv(x_view); if stat_partial != NULL the epilogue also writes per-row-block partial
 * (sum, sum of squares) of y_raw per output channel: [blocks][2][md_cpad(Cout)]  (first half of
 * BatchNorm3d's batch statistics, R2Plus1D.py:53). */
int md_conv_fwd(const MdConvDesc* d, const MdActView* x, const float* wpack_fwd, float* y_raw,
                float* stat_partial, void* stream);

/* dx (+)= conv_transpose(dy_raw): gradient w.r.t. the (activated) conv input.  accumulate != 0 adds
 * into dx (used where a residual skip and a conv share their input, R2Plus1D.py:181-187). */
int md_conv_dgrad(const MdConvDesc* d, const float* dy_raw, const float* wpack_dgrad, float* dx,
                  int accumulate, void* stream);

/* dw (reference layout (Cout,Cin,kt,kh,kw), overwritten) = x_view^T * dy_raw.  `workspace` must hold
 * md_conv_wgrad_workspace_floats(d) floats (per-slice partial sums, reduced in a fixed order); it may be NULL
 * when that size is 0 (the strided / exact-fp32 path accumulates with float atomics instead). */
size_t md_conv_wgrad_workspace_floats(const MdConvDesc* d);
int md_conv_wgrad(const MdConvDesc* d, const MdActView* x, const float* dy_raw, float* dw, float* workspace,
                  void* stream);

/* ------------------------------------------------------------------------------------------------
 * BatchNorm3d (train mode) + LeakyReLU + residual: everything between two convolutions
 * (R2Plus1D.py:53-58, :179-187).
 * ---------------------------------------------------------------------------------------------- */
/* Second half of the batch statistics: reduce the conv epilogue's partials in fixed order (fp64),
 * produce mean / invstd / scale=gamma*invstd / shift=beta-mean*scale (all [Cp]) and update
 * running_mean / running_var (momentum 0.1, unbiased variance) like nn.BatchNorm3d. running_* may be NULL. */
int md_bn_finalize(const float* stat_partial, int32_t blocks, int32_t C, int64_t count,
                   const float* gamma, const float* beta, float eps, float momentum,
                   float* running_mean, float* running_var,
                   float* mean, float* invstd, float* scale, float* shift, void* stream);

/* Eval-mode counterpart (model.eval(), src/train.py:104): scale/shift from the running statistics. */
int md_bn_eval_params(int32_t C, const float* gamma, const float* beta, const float* running_mean,
                      const float* running_var, float eps, float* mean, float* invstd, float* scale,
                      float* shift, void* stream);

/* out = view(x)  (materialise BatchNorm+LeakyReLU); rows = N*T*H*W pixels, C real channels. */
int md_bn_act(const MdActView* x, int64_t rows, int32_t C, float* out, void* stream);

/* z = leaky(view(skip) + view(main), alpha): SpatioTemporalResBlock.forward's `self.relu(x + res)`,
 * R2Plus1D.py:187. */
int md_residual_fwd(const MdActView* skip, const MdActView* main, float alpha, int64_t rows, int32_t C,
                    float* z, void* stream);

/* BatchNorm backward, split like the forward:
 *  (1) md_bn_bwd_reduce : partial sums of g and g*xhat per channel, g = dA * leaky'(pre-activation).
 *      With `skip` != NULL the unit closes a residual block: dA := dZ * leaky'_alpha(view(skip)+view(main)).
 *  (2) md_bn_bwd_finalize: dgamma, dbeta and the per-channel coefficients of (3).
 *  (3) md_bn_bwd_apply  : d_raw = scale * (g - mean(g) - xhat * mean(g*xhat)); may run in place
 *      (d_raw == dA).  In the residual form it also writes dS = dZ*leaky'_alpha(...) (may alias dZ).
 */
int32_t md_bn_bwd_blocks(int64_t rows, int32_t C);
int md_bn_bwd_reduce(const float* dA, const MdActView* main, const MdActView* skip, float alpha,
                     const float* mean, const float* invstd, int64_t rows, int32_t C,
                     float* partial /* [blocks][2][Cp] */, void* stream);
int md_bn_bwd_finalize(const float* partial, int32_t blocks, int32_t C, int64_t count,
                       float* dgamma, float* dbeta, float* coef /* [2][Cp]: mean(g), mean(g*xhat) */,
                       void* stream);
int md_bn_bwd_apply(const float* dA, const MdActView* main, const MdActView* skip, float alpha,
                    const float* mean, const float* invstd, const float* coef, int64_t rows, int32_t C,
                    float* d_raw, float* dS, void* stream);

/* Fused form of (1) for the units that feed a convolution directly: the data gradient of the CONSUMER convolution
 * writes g = dA * leaky'(bn(y)) (y = `y_view`, the raw output + BatchNorm constants of the unit being differentiated)
 * instead of dA and leaves the partial sums of (1) in `partial`
 * ([md_conv_dgrad_bnred_blocks(d)][2][md_cpad(Cin)]), so no separate reduction pass reads the tensor again.
 * md_conv_dgrad_bnred_blocks returns 0 when this geometry has no fused form (use md_conv_dgrad + md_bn_bwd_reduce).
 * (3) then runs as md_bn_bwd_apply_g, which takes g instead of dA.  Backward of Conv3dBlock, R2Plus1D.py:53-58. */
int32_t md_conv_dgrad_bnred_blocks(const MdConvDesc* d);
int md_conv_dgrad_bnred(const MdConvDesc* d, const float* dy_raw, const float* wpack_dgrad, float* g_out, int accumulate,
                        const MdActView* y_view, const float* mean, const float* invstd, float* partial, void* stream);
int md_bn_bwd_apply_g(const float* g, const MdActView* main, const float* mean, const float* invstd, const float* coef,
                      int64_t rows, int32_t C, float* d_raw, void* stream);

/* Pre-split gradient format.  The data-gradient and weight-gradient kernels multiply fp32 operands as two bf16 halves
 * (hi = bf16(x), lo = bf16(x - hi)); splitting d_raw while it is staged into LDS costs them vector-ALU issue slots they
 * are short of, and every d_raw is staged twice (by the unit's data gradient and by its weight gradient).  With
 * split_out != 0 the apply pass -- an HBM-bound streaming kernel with idle ALUs -- writes d_raw already split:
 * [pixel][ceil(C/8) chunks]{hi: 8 x bf16 | lo: 8 x bf16}, 32 bytes per 8 channels (the same bytes as fp32 when
 * md_cpad(C) % 8 == 0, which is required; the pass may then run in place).  The consumers stage it with plain copies and
 * produce bit-identical results.  md_conv_split_dy_ok(d, need_dgrad) says whether the kernels that will read the d_raw of
 * convolution d (its weight gradient, and its data gradient when need_dgrad != 0) all accept the format.
 * md_bn_bwd_apply_fmt generalises md_bn_bwd_apply (g_in = 0) and md_bn_bwd_apply_g (g_in = 1, skip must be NULL);
 * md_conv_dgrad_fmt generalises md_conv_dgrad (y_view = NULL) and md_conv_dgrad_bnred; md_conv_wgrad_fmt md_conv_wgrad.
 * Backward of Conv3dBlock, R2Plus1D.py:44-58. */
int md_conv_split_dy_ok(const MdConvDesc* d, int need_dgrad);
int md_bn_bwd_apply_fmt(const float* dA, int g_in, const MdActView* main, const MdActView* skip, float alpha,
                        const float* mean, const float* invstd, const float* coef, int64_t rows, int32_t C,
                        void* d_raw, int split_out, float* dS, void* stream);
/* Finalize + apply in one launch (fp32 d_raw): the apply pass itself sums the `blocks` partial rows written by
 * md_bn_bwd_reduce -- or, with g_in != 0, by a data gradient's fused reduction (dA then holds g) -- in fp64 and a fixed order,
 * writes dgamma / dbeta (either may be NULL) and d_raw (and dS with a skip view); replaces md_bn_bwd_finalize +
 * md_bn_bwd_apply[_g] and their coefficient buffer.  count = rows.  Backward of R2Plus1D.py:53-58, :179-187. */
int md_bn_bwd_apply_fused(const float* dA, int g_in, const MdActView* main, const MdActView* skip, float alpha,
                          const float* mean, const float* invstd, const float* partial, int32_t blocks, int64_t count,
                          float* dgamma, float* dbeta, int64_t rows, int32_t C, float* d_raw, float* dS, void* stream);
int md_conv_dgrad_fmt(const MdConvDesc* d, const void* dy, int dy_split, const float* wpack_dgrad, float* dx,
                      int accumulate, const MdActView* y_view, const float* mean, const float* invstd,
                      float* partial, void* stream);
int md_conv_wgrad_fmt(const MdConvDesc* d, const MdActView* x, const void* dy, int dy_split, float* dw,
                      float* workspace, void* stream);
/* The X operand of a weight gradient as a pre-activated, pre-split copy: md_bn_act_split writes leaky(scale * y + shift) of a
 * unit (or a materialised tensor as it is, x->scale == NULL) as [row][ceil(Cp/8) chunks]{hi 8 x bf16 | lo 8 x bf16}
 * (md_bn_act_split_floats floats); md_conv_wgrad_fmt2 with x_split != 0 takes that buffer as x->data (scale / shift ignored) and
 * stages it by plain copy -- bit-identical to BatchNorm-on-read inside the kernel, a third less time per box.  Available where
 * md_conv_wgrad_xsplit_ok says so (LDS-patch kernel, not the pixel-pair stem).  The executor writes the copies on its side stream
 * during the forward pass. */
size_t md_bn_act_split_floats(int64_t rows, int32_t C);
int md_bn_act_split(const MdActView* x, int64_t rows, int32_t C, void* out, void* stream);
int md_conv_wgrad_xsplit_ok(const MdConvDesc* d);
int md_conv_wgrad_fmt2(const MdConvDesc* d, const MdActView* x, int x_split, const void* dy, int dy_split, float* dw,
                       float* workspace, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Boundary layout conversion, pooling, classifier head, losses.
 * ---------------------------------------------------------------------------------------------- */
/* (B,C,T,H,W) fp32 planar (src/dataset.py:229-230) <-> channels-last with pitch md_cpad(C). */
int md_nchw_to_cl(const float* x, int32_t B, int32_t C, int64_t thw, float* out, void* stream);
int md_cl_to_nchw(const float* x, int32_t B, int32_t C, int64_t thw, float* out, void* stream);

/* torch.cat([a, b], dim=1) of two channels-last tensors with `rows` pixels each (the lateral connection of SlowFast,
 * src/models/slowfast.py:26-40): out has pitch md_cpad(Ca + Cb), padding channels zero.  md_split_cl is its backward: the
 * gradient g (pitch md_cpad(Ca + Cb)) split into da (pitch md_cpad(Ca)) and db (pitch md_cpad(Cb)), padding channels zero. */
int md_cat_cl(const float* a, int32_t Ca, const float* b, int32_t Cb, int64_t rows, float* out, void* stream);
int md_split_cl(const float* g, int32_t Ca, int32_t Cb, int64_t rows, float* da, float* db, void* stream);

/* AdaptiveAvgPool3d(1) + view (R2Plus1D.py:215,224-225): feat[B][C] = mean over thw of x[B][thw][Cp]. */
int md_avgpool_fwd(const float* x, int32_t B, int32_t C, int64_t thw, float* feat, void* stream);
int md_avgpool_bwd(const float* dfeat, int32_t B, int32_t C, int64_t thw, float* dx, void* stream);

/* Classifier head Linear(D->Hd) + BatchNorm1d(Hd) + activation + Linear(Hd->K), R2Plus1D.py:243-248 (the same shape of
 * head closes SlowFast, CnnLSTM and MLSTM_FCN).  Activation: alpha >= 0 -> ELU(alpha) (alpha = 0 is ReLU);
 * alpha < 0 -> LeakyReLU(-alpha) (MLSTM_FCN.py:117).  save: workspace of md_head_save_floats() floats kept for backward. */
size_t md_head_save_floats(int32_t B, int32_t D, int32_t Hd);
int md_head_fwd(const float* feat, int32_t B, int32_t D, int32_t Hd, int32_t K,
                const float* w0, const float* b0, const float* gamma, const float* beta,
                const float* w1, const float* b1, float alpha, float eps, float momentum, int training,
                float* running_mean, float* running_var, float* logits, float* save, void* stream);
int md_head_bwd(const float* dlogits, const float* feat, int32_t B, int32_t D, int32_t Hd, int32_t K,
                const float* w0, const float* gamma, const float* w1, float alpha, const float* save,
                float* dfeat, float* dw0, float* db0, float* dgamma, float* dbeta, float* dw1, float* db1,
                void* stream);

/* Fused softmax + loss (src/loss.py): kind 0 = FocalLoss (:14-34, sum), 1 = LDAMLoss (:37-69,
 * weighted mean), 2 = CELoss (:71-81, sum).  Emits the scalar loss, d loss / d logits (for an upstream
 * gradient of 1; the caller scales) and pred = argmax softmax (src/train.py:70) in one launch.  class_weight / margins may be NULL (ones / zeros).  target is int64. */
int md_softmax_loss(int32_t kind, const float* logits, const int64_t* target, int32_t B, int32_t K,
                    const float* class_weight, const float* margins, float gamma_or_s,
                    float* loss, float* dlogits, int64_t* pred, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Elementwise pieces of the other encoders (next rows of the scope table).
 * ---------------------------------------------------------------------------------------------- */
/* SwishEfficient (src/models/resnet.py:70-81): y = x * sigmoid(x); dx = dy * s * (1 + x * (1 - s)), s = sigmoid(x). */
int md_swish_fwd(const float* x, int64_t n, float* y, void* stream);
int md_swish_bwd(const float* x, const float* dy, int64_t n, float* dx, void* stream);
/* NoiseLayer, training branch (src/models/NoiseLayer.py:12-14): out = x + (mean + noise * std); `noise` holds the
 * standard-normal draws (the reference draws them with the CPU generator; the caller keeps that choice). */
int md_add_noise(const float* x, const float* noise, float mean, float std, int64_t n, float* out, void* stream);
/* Squeeze-and-excitation gate + Swish of Bottleneck3D (src/models/resnet.py:182-190) on (N,C,T,H,W) tensors:
 * pool = mean_thw a; hidden = relu(W1 pool + b1) [N][Wd]; gate = sigmoid(W2 hidden + b2) [N][C]; out = swish(a * gate).
 * W1 = fc1.weight as [Wd][C], W2 = fc2.weight as [C][Wd].  pool / hidden / gate are kept by the caller for the backward,
 * which returns da, dW1, db1, dW2, db2 (scratch: 3*N*C + N*Wd floats).  Fixed-order reductions. */
int md_se_swish_fwd(const float* a, int32_t N, int32_t C, int64_t thw, int32_t Wd, const float* w1, const float* b1,
                    const float* w2, const float* b2, float* pool, float* hidden, float* gate, float* out, void* stream);
int md_se_swish_bwd(const float* a, const float* dout, int32_t N, int32_t C, int64_t thw, int32_t Wd, const float* w1,
                    const float* w2, const float* pool, const float* hidden, const float* gate, float* da, float* dw1,
                    float* db1, float* dw2, float* db2, float* scratch, void* stream);
/* Squeeze-excitation without the Swish (MLSTM_FCN's SqueezeExciteBlock on (N,C,T), MLSTM_FCN.py:17-33): out = a * gate, same
 * gate network and buffers as md_se_swish_* (its Linears have no bias: pass zero vectors). */
int md_se_scale_fwd(const float* a, int32_t N, int32_t C, int64_t thw, int32_t Wd, const float* w1, const float* b1,
                    const float* w2, const float* b2, float* pool, float* hidden, float* gate, float* out, void* stream);
int md_se_scale_bwd(const float* a, const float* dout, int32_t N, int32_t C, int64_t thw, int32_t Wd, const float* w1,
                    const float* w2, const float* pool, const float* hidden, const float* gate, float* da, float* dw1,
                    float* db1, float* dw2, float* db2, float* scratch, void* stream);
/* out = x * mask * scale: the inverted-dropout mask nn.LSTM applies between its layers in training mode. */
int md_mask_scale(const float* x, const float* mask, float scale, int64_t n, float* out, void* stream);
/* Residual close of Bottleneck3D (resnet.py:196-198): out = relu(a + b); dx = dout * (out > 0) for both inputs. */
int md_add_relu_fwd(const float* a, const float* b, int64_t n, float* out, void* stream);
/* leaky_relu(a + b, alpha) and its backward from the OUTPUT (alpha >= 0): residual close of a stand-alone
 * SpatioTemporalResBlock, R2Plus1D.py:183-187 (inside the trunk executor md_residual_fwd does this on the raw tensors). */
int md_add_leaky_fwd(const float* a, const float* b, float alpha, int64_t n, float* out, void* stream);
int md_add_leaky_bwd(const float* out, const float* dout, float alpha, int64_t n, float* dx, void* stream);
int md_add_relu_bwd(const float* out, const float* dout, int64_t n, float* dx, void* stream);
/* MaxPool3d(kernel (1,3,3), stride (1,2,2), padding (0,1,1)) of ResNet3D.layer0 (resnet.py:225) on `planes` = N*C*T
 * planes of H x W; idx keeps the flat h*W+w of the maximum (first one wins, NaN propagates, as ATen); the backward sums
 * dout over the windows that selected each input (gather form, deterministic). */
int md_maxpool_1x3x3_fwd(const float* x, int64_t planes, int32_t H, int32_t W, float* out, int32_t* idx, void* stream);
int md_maxpool_1x3x3_bwd(const float* dout, const int32_t* idx, int64_t planes, int32_t H, int32_t W, float* dx, void* stream);
/* F.adaptive_avg_pool3d(x, 1) on (N,C,T,H,W) (slowfast.py:33,86): mean[row] over the thw values of each (n,c) row. */
int md_rowmean_fwd(const float* x, int64_t rows, int64_t thw, float* mean, void* stream);
int md_rowmean_bwd(const float* dmean, int64_t rows, int64_t thw, float* dx, void* stream);

/* ViViT's tubelet / patch embedding (reference src/models/ViViT.py:141-148 'b t c (h p1) (w p2) -> b t (h w) (p1 p2 c)' + Linear,
 * :175-184 space token + positional table) as one gather-GEMM: out [B*T][n+1][dim], row 0 = token + pos[t][0], row 1+j =
 * patch_j . W'^T + bias + pos[t][1+j].  The clip is read in place through strides (floats) sb / st / sc of (b, t, c); rows of
 * W pixels contiguous.  w_perm [dim][C*p*p] is the Linear weight with its columns in (c, p1, p2) order (the reference's are
 * (p1, p2, c)); pos [T][n+1][dim].  patch: power of two >= 8.  MD_ERR_UNSUPPORTED in exact-fp32 mode (callers compose the
 * embedding from md_conv_fwd + md_channel_bias_*).  The weight gradient dw_perm [dim][C*p*p] (same column order) gathers the
 * patches again instead of keeping a rearranged copy; gout is the gradient of `out` (all n+1 rows per frame). */
int md_patch_embed_fwd(const float* x, int32_t B, int32_t T, int32_t C, int32_t H, int32_t W, int64_t sb, int64_t st, int64_t sc,
                       int32_t patch, const float* w_perm, const float* bias, const float* pos, const float* token, int32_t dim,
                       float* out, void* stream);
size_t md_patch_embed_wgrad_workspace_floats(int32_t B, int32_t T, int32_t C, int32_t H, int32_t W, int32_t patch, int32_t dim);
int md_patch_embed_wgrad(const float* x, int32_t B, int32_t T, int32_t C, int32_t H, int32_t W, int64_t sb, int64_t st, int64_t sc,
                         int32_t patch, const float* gout, int32_t dim, float* dw_perm, float* workspace, void* stream);

/* Per-channel bias on an (N,C,L) tensor (a Conv1d bias that is not absorbed by a following normalisation,
 * src/models/CnnLSTM.py:42) and its gradient db[c] = sum_{n,l} dout. */
int md_channel_bias_fwd(const float* x, const float* bias, int32_t N, int32_t C, int32_t L, float* out, void* stream);
size_t md_channel_bias_bwd_scratch_floats(int32_t N, int32_t C, int32_t L);   /* 0: no scratch needed */
int md_channel_bias_bwd(const float* dout, int32_t N, int32_t C, int32_t L, float* dbias, float* scratch /* may be NULL: the
                        one-workgroup-per-channel form is used */, void* stream);
/* out[b][d] = scale * sum_s x[b][s][d] and its adjoint: what CnnLSTM's attention pooling (:76-97) evaluates to -- the
 * softmax is taken over the same axis the result is averaged over, so every step gets weight 1/H (see src/models/CnnLSTM.py). */
int md_seq_sum_fwd(const float* x, int32_t B, int32_t S, int32_t D, float scale, float* out, void* stream);
int md_seq_sum_bwd(const float* dout, int32_t B, int32_t S, int32_t D, float scale, float* dx, void* stream);
/* Pieces of the Transformer 0D encoder (src/models/transformer.py:40-109 over nn.TransformerEncoderLayer, post-norm):
 *  - out = LayerNorm(a + b) * gamma + beta per row of D features (b may be NULL); xhat [rows][D] and rstd [rows] are kept for
 *    the backward, whose dx is the gradient of BOTH summands.
 *  - attention core of nn.MultiheadAttention: qkv [S][B][3D] (q | k | v), H heads, additive mask [S][S] (may be NULL, -inf
 *    allowed), drop [B*H][S][S] = keep-mask / keep-probability of the attention dropout (may be NULL); probs [B*H][S][S] are
 *    kept for the backward; out [S][B][D].
 *  - GELU: kind 0 = exact erf form (nn.GELU, transformer.py:85), kind 1 = the reference's tanh form (:35-37); with dy != NULL
 *    the call returns dy * gelu'(x). */
int md_add_layernorm_fwd(const float* a, const float* b, const float* gamma, const float* beta, int64_t rows, int32_t D, float eps,
                         float* out, float* xhat, float* rstd, float* sum_out /* may be NULL: a + b, the pre-norm residual stream
                         of ViViT's Transformer (src/models/ViViT.py:108-111) */, void* stream);
size_t md_add_layernorm_bwd_scratch_floats(int64_t rows, int32_t D);
/* dres (may be NULL): gradient that reached a + b through the residual stream, added to dx.  scratch: as sized above (may be
 * NULL when that is 0). */
int md_add_layernorm_bwd(const float* dout, const float* gamma, const float* xhat, const float* rstd, const float* dres,
                         int64_t rows, int32_t D, float* dx, float* dgamma, float* dbeta, float* scratch, void* stream);
/* One step of ViViT's pre-norm residual stream with the branch's tail folded in (src/models/ViViT.py:31-46,85-91,108-111: the branch
 * ends in nn.Linear -> nn.Dropout, then x = branch(x) + x and the next PreNorm's LayerNorm): y = the Linear's raw product [rows][D],
 * bias [D] (may be NULL), key / tag / keep = the dropout decisions of md_dropout_ctr (key NULL: no dropout),
 *   sum_out = (y + bias) * decision / keep + stream_in,   out = LayerNorm(sum_out) * gamma + beta   (xhat, rstd kept for the backward).
 * Backward: dstream = gradient of sum_out (LayerNorm's plus dres, what arrived through the stream); dbranch = dstream * decision /
 * keep = gradient of y; dgamma, dbeta and dbias (the column sum of dbranch) come out of the same pass + one small fixed-order
 * reduction; scratch: md_branch_layernorm_bwd_scratch_floats.
 * D % 4 == 0, D <= 256, 16-byte aligned tensors (md_branch_layernorm_supported). */
int md_branch_layernorm_supported(int64_t rows, int32_t D);
int md_branch_layernorm_fwd(const float* y, const float* bias, const int64_t* key, int32_t tag, float keep, const float* stream_in,
                            const float* gamma, const float* beta, int64_t rows, int32_t D, float eps, float* out, float* xhat,
                            float* rstd, float* sum_out, void* stream);
size_t md_branch_layernorm_bwd_scratch_floats(int64_t rows, int32_t D);
int md_branch_layernorm_bwd(const float* dout, const float* gamma, const float* xhat, const float* rstd, const float* dres,
                            const int64_t* key, int32_t tag, float keep, int64_t rows, int32_t D, float* dstream, float* dbranch,
                            float* dgamma, float* dbeta, float* dbias /* may be NULL */, float* scratch, void* stream);
/* batch_first = 0: qkv [S][B][3D], out [S][B][D] (nn.MultiheadAttention); 1: qkv [B][S][3D], out [B][S][D] (ViViT's Attention,
 * src/models/ViViT.py:69-88: 'b n (h d)' heads, scale d_head^-0.5, no mask).  S*16*4 bytes of LDS: S <= 937. */
int md_attention_fwd(const float* qkv, const float* mask, const float* drop, int32_t S, int32_t B, int32_t D, int32_t H,
                     int32_t batch_first, float* probs, float* out, void* stream);
int md_attention_bwd(const float* qkv, const float* probs, const float* drop, const float* dout, int32_t S, int32_t B, int32_t D,
                     int32_t H, int32_t batch_first, float* dqkv, float* ds_scratch /* B*H*S*S floats */, void* stream);

/* The same attention without the S x S matrices (ViViT.py:69-88, unmasked and dropout-free): the forward pass keeps one value per
 * query row -- lse[B*H][S] = max + log(sum) of its scaled scores -- and the backward pass recomputes the probabilities from q, k and
 * lse.  md_attention_lse_supported: 1 when the split-precision matrix-core kernels take this shape (S <= 256, d_head 32 or 64, not
 * in exact-fp32 mode); otherwise use md_attention_fwd / _bwd.  delta: [B*H][S] floats of scratch. */
int32_t md_attention_lse_supported(int32_t S, int32_t D, int32_t H);
int md_attention_lse_fwd(const float* qkv, int32_t S, int32_t B, int32_t D, int32_t H, int32_t batch_first, float* lse, float* out,
                         void* stream);
int md_attention_lse_bwd(const float* qkv, const float* lse, const float* dout, int32_t S, int32_t B, int32_t D, int32_t H,
                         int32_t batch_first, float* dqkv, float* delta, void* stream);
/* ELU (src/models/ViViT.py:166): with dy != NULL the call returns dy * elu'(x).  alpha = 0 is ReLU (MultiModal.py:23,29). */
int md_elu(const float* x, const float* dy, float alpha, int64_t n, float* out, void* stream);
int md_gelu(const float* x, const float* dy, int32_t kind, int64_t n, float* out, void* stream);

/* FeedForward hidden activation in one pass (ViViT.py:31-46: Linear bias -> nn.GELU -> nn.Dropout): out = gelu(x + bias[c]) * mask *
 * scale over x[rows][C] (C % 4 == 0, 16-byte aligned pointers; mask NULL: no dropout; kind as md_gelu).  With dout: the backward,
 * out = dout * mask * scale * gelu'(x + bias); the bias gradient is its column sum (md_channel_bias_bwd).  Bit-identical to
 * md_channel_bias_fwd -> md_gelu -> md_mask_scale. */
int md_bias_gelu_drop(const float* x, const float* bias, const float* mask, const float* dout, float scale, int32_t kind, int64_t rows,
                      int32_t C, float* out, void* stream);
/* Mask-free inverted dropout (nn.Dropout in training mode: ViViT.py:31-46,85-91; the reference draws a mask from the framework's
 * generator, whose stream is device- and version-specific anyway): element i is kept when a Philox4x32-10 word of
 * (state = a 128-bit key, a device int64[2] the caller draws once per forward --, tag = call site, i) is below keep; out = x * scale where kept, 0 elsewhere.  The
 * same (state, tag) regenerates the same decisions, so the backward pass is the same call on the gradient and no mask tensor is
 * written or read (at cfg3 the FeedForward's mask alone was a 68 MB write and two 68 MB reads per layer and step).
 * md_bias_gelu_drop_ctr: md_bias_gelu_drop with the decisions generated instead of read (dout != NULL: the backward). */
int md_dropout_ctr(const float* x, const int64_t* state, int32_t tag, float keep, float scale, int64_t n, float* out, void* stream);
int md_bias_gelu_drop_ctr(const float* x, const float* bias, const int64_t* state, int32_t tag, float keep, const float* dout,
                          float scale, int32_t kind, int64_t rows, int32_t C, float* out, void* stream);
/* Device-side tail of DatasetForVideo.get_video_data (src/dataset.py:124-144, without the cv2 augmentations): centre crop of
 * S x S (rows Hr/2 - S/2 .., columns Wr/2 - S/2 .., :241-246; S even), subtraction of the BGR means (:203-207, host array of 3)
 * and the (T,H,W,C) -> (C,T,H,W) transpose (:229-230) from uint8 frames [B][T][Hr][Wr][3].  layout 0: out [B][3][T][S][S];
 * layout 1: out [B][T][S][S][4] with channel 3 = 0 (the kernels' channels-last layout).  Exact: uint8 -> fp32 minus the mean. */
int md_clip_preprocess(const uint8_t* frames, int32_t B, int32_t T, int32_t Hr, int32_t Wr, int32_t S, const float* mean_bgr,
                       int32_t layout, float* out, void* stream);
/* The same with the six augmentations of :129-135, 152-227 between crop and mean subtraction.  params [B][10] int32 (device):
 * brightness mode (0 none / 1 add + clip to [10,255] / 2 add + horizontal flip), brightness amount, contrast on, contrast alpha,
 * blur on, blur kernel size, row_lo, row_hi, col_lo, col_hi (pixels outside [lo,hi) become 0 before the mean subtraction: the
 * reference's vertical_shift / horizontal_shift mask edges, they do not shift); gauss: the kernel-size Gaussian weights (device).
 * The host draws the decisions (src/utils/clip_preprocess.py::draw_augmentation: the reference's random-number calls in order). */
int md_clip_augment_preprocess(const uint8_t* frames, int32_t B, int32_t T, int32_t Hr, int32_t Wr, int32_t S, const float* mean_bgr,
                               int32_t layout, const int32_t* params, const float* gauss, float* out, void* stream);

/* Class-balanced re-sampling as a device-side, rank-sharded index stream (ImbalancedDatasetSampler, src/utils/sampler.py:5-35:
 * torch.multinomial with replacement over 1 / class-count weights).  cum_dist: the normalised cumulative distribution in
 * float64 (cum_dist[ncat-1] == 1), uniforms: nsamples float64 draws of the SAME CPU generator stream torch.multinomial would
 * consume; rank r of `world` receives draws r, r + world, ...: out[k] = index_map[leftmost c with cum_dist[c] >= u[r + k*world]]
 * (index_map may be NULL).  Index-exact with torch.multinomial for the same generator state.  md_multinomial_shard_count
 * gives the number of indices rank r receives. */
int64_t md_multinomial_shard_count(int64_t nsamples, int32_t rank, int32_t world);
int md_multinomial_shard(const double* cum_dist, int64_t ncat, const double* uniforms, int64_t nsamples, int32_t rank,
                         int32_t world, const int64_t* index_map, int64_t* out, void* stream);
/* Tensor fusion of TFN / TFN_GB (src/models/MultiModal.py:214-220, 301-307): out [B][(Da+1)*(Dc+1)] = [1 | a[b]] (x) [1 | c[b]]
 * (what the reference builds with torch.cat of ones + torch.bmm); the backward returns da [B][Da], dc [B][Dc]. */
int md_outer_fwd(const float* a, const float* c, int32_t B, int32_t Da, int32_t Dc, float* out, void* stream);
int md_outer_bwd(const float* a, const float* c, const float* dout, int32_t B, int32_t Da, int32_t Dc, float* da, float* dc,
                 void* stream);
/* One direction of one nn.LSTM layer (batch_first = False, zero initial state; gate order i, f, g, o), as used by CnnLSTM
 * (src/models/CnnLSTM.py:51,93-96) and MLSTM_FCN.  x [S][B][I]; h_all, c_all [S][B][H]; gates [S][B][4H] (activated gates,
 * kept for the backward); reverse != 0 processes t = S-1 .. 0 (the "_reverse" direction).  The backward takes the gradient
 * with respect to every output h_t and returns dx, dW_ih [4H][I], dW_hh [4H][H] and db [4H] (the gradient of b_ih and of
 * b_hh); dpre_scratch: S*B*4H floats. */
/* Register-resident form for H = 64 / 128 (md_lstm_rec_supported): the input projection xproj [S][B][4H] = x W_ih^T is computed by the
 * caller as one GEMM (which then also owns dx and dW_ih); the recurrence keeps W_hh in registers for the whole sequence.  The
 * backward returns dpre [S][B][4H] (the gradient of xproj), dW_hh and db (for b_ih and b_hh alike); with dw_hh = db = NULL only
 * dpre (the caller then forms dW_hh = dpre^T h_prev and db = column sums of dpre with the GEMM / reduction kernels). */
int md_lstm_rec_supported(int32_t H);
int md_lstm_rec_fwd(const float* xproj, const float* w_hh, const float* b_ih, const float* b_hh, int32_t S, int32_t B, int32_t H,
                    int32_t reverse, float* h_all, float* c_all, float* gates, void* stream);
int md_lstm_rec_bwd(const float* dh_all, const float* w_hh, const float* h_all, const float* c_all, const float* gates, int32_t S,
                    int32_t B, int32_t H, int32_t reverse, float* dpre, float* dw_hh, float* db, void* stream);
/* Both directions of one bidirectional layer in one launch each way (arrays of two device pointers: [0] forward in time, [1] reverse):
 * the two recurrences are independent chains of S dependent steps, run side by side instead of one after the other. */
int md_lstm_rec_fwd2(const float* const* xproj, const float* const* w_hh, const float* const* b_ih, const float* const* b_hh,
                     int32_t S, int32_t B, int32_t H, float* const* h_all, float* const* c_all, float* const* gates, void* stream);
int md_lstm_rec_bwd2(const float* const* dh_all, const float* const* w_hh, const float* const* h_all, const float* const* c_all,
                     const float* const* gates, int32_t S, int32_t B, int32_t H, float* const* dpre, float* const* dw_hh,
                     float* const* db, void* stream);
int md_lstm_fwd(const float* x, const float* w_ih, const float* w_hh, const float* b_ih, const float* b_hh, int32_t S,
                int32_t B, int32_t I, int32_t H, int32_t reverse, float* h_all, float* c_all, float* gates, void* stream);
int md_lstm_bwd(const float* dh_all, const float* x, const float* w_ih, const float* w_hh, const float* h_all,
                const float* c_all, const float* gates, int32_t S, int32_t B, int32_t I, int32_t H, int32_t reverse, float* dx,
                float* dw_ih, float* dw_hh, float* db, float* dpre_scratch, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Optimizer step: torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm) + torch.optim.AdamW.step()
 * (src/train.py:64-66, train_vision_network.py:277-278) over all parameter tensors in two launches.
 * `tensors` is a device array of MdOptTensor, `chunks` a device array of MdOptChunk covering every tensor in pieces
 * of md_opt_chunk_elems() elements (the last piece of a tensor may be short).  All pointers fp32, 4-byte aligned.
 * ---------------------------------------------------------------------------------------------- */
typedef struct MdOptTensor { float* param; float* grad; float* exp_avg; float* exp_avg_sq; int64_t numel; } MdOptTensor;
typedef struct MdOptChunk { int32_t tensor; int32_t offset; } MdOptChunk;   /* offset in chunks from the tensor's start */
int md_opt_chunk_elems(void);
/* norm_coef[0] = 2-norm over all gradients, norm_coef[1] = min(1, max_norm / (norm + 1e-6)) (1 if max_norm <= 0);
 * partial: nchunks floats of scratch.  Deterministic (fixed summation order). */
int md_opt_grad_norm(const void* tensors, const void* chunks, int32_t nchunks, float max_norm, float* partial,
                     float* norm_coef, void* stream);
/* One AdamW step (decoupled weight decay, bias correction for `step` >= 1, no amsgrad).  If norm_coef != NULL the
 * gradients are first scaled by norm_coef[1] in place, as clip_grad_norm_ does. */
int md_opt_adamw_step(const void* tensors, const void* chunks, int32_t nchunks, const float* norm_coef, float lr,
                      float beta1, float beta2, float eps, float weight_decay, int64_t step, void* stream);
/* Same update, applied only if the device scalar *ok_flag is exactly 1 (NULL: always).  The data-parallel loop passes the
 * all-reduced (AVG) "loss is finite on this rank" flag: the reference's per-batch skip (src/train.py:56-58) becomes a collective,
 * device-side decision with no host synchronisation in the step. */
int md_opt_adamw_step_if(const void* tensors, const void* chunks, int32_t nchunks, const float* norm_coef, float lr,
                         float beta1, float beta2, float eps, float weight_decay, int64_t step, const float* ok_flag,
                         void* stream);

/* ------------------------------------------------------------------------------------------------
 * Whole-trunk executor: R2Plus1DNet.forward / backward (R2Plus1D.py:207-226) as one plan.
 * ---------------------------------------------------------------------------------------------- */
typedef struct MdPlan MdPlan;
/* layer_sizes[4] as in R2Plus1DNet(layer_sizes) :208; alpha = LeakyReLU slope of the stem and of the
 * block-closing activations (:210-214). */
int md_plan_create(int32_t B, int32_t T, int32_t H, int32_t W, const int32_t* layer_sizes, float alpha,
                   MdPlan** out);
void md_plan_destroy(MdPlan* p);
/* Number of conv units (Conv3dBlock instances) and their order = the reference's module order
 * (stem spatial, stem temporal, then per block conv1.s, conv1.t, conv2.s, conv2.t[, downsample.s, .t]). */
int32_t md_plan_num_units(const MdPlan* p);
int md_plan_unit_desc(const MdPlan* p, int32_t i, MdConvDesc* out);
/* Device memory the plan needs (bytes): activations kept for backward + scratch. */
size_t md_plan_workspace_bytes(const MdPlan* p);
/* Diagnostics: where md_plan_forward leaves intermediate tensors in the workspace (offsets in floats).  Unit i: raw conv
 * output [rows][Cp] and the statistics block [mean | invstd | scale | shift][Cp]; materialised tensor zi (0 = the clip in
 * channels-last, 1 = the stem output, then one per residual block): [rows][md_cpad(C)].  Used by tools/kink_diag.py to
 * compare LeakyReLU pre-activations with the oracle's, element by element. */
int md_plan_unit_layout(const MdPlan* plan, int32_t unit, size_t* raw_off, size_t* stat_off, int64_t* rows, int32_t* Cp);
int32_t md_plan_num_z(const MdPlan* plan);
int md_plan_z_layout(const MdPlan* plan, int32_t zi, size_t* off, int64_t* rows, int32_t* C);
/* Parameter / gradient / buffer tables: arrays of device pointers indexed by unit:
 *   w[i] (Cout,Cin,kt,kh,kw), gamma[i], beta[i], running_mean[i], running_var[i] (may be NULL). */
int md_plan_forward(MdPlan* p, const float* x_ncthw, const float* const* w, const float* const* gamma,
                    const float* const* beta, float* const* running_mean, float* const* running_var,
                    int training, float* feat /* [B][C_out] */, void* workspace, void* stream);
int md_plan_backward(MdPlan* p, const float* dfeat, const float* const* w, const float* const* gamma,
                     float* const* dw, float* const* dgamma, float* const* dbeta,
                     void* workspace, void* stream);
/* Segmented backward for overlap with gradient all-reduce (src/distributed.py DP loop): runs the
 * backward of stages [stage_hi .. stage_lo] (4 = conv5 ... 0 = stem). */
int md_plan_backward_range(MdPlan* p, const float* dfeat, const float* const* w, const float* const* gamma,
                           float* const* dw, float* const* dgamma, float* const* dbeta,
                           void* workspace, int32_t stage_hi, int32_t stage_lo, void* stream);
int32_t md_plan_feat_dim(const MdPlan* p);
/* Backward schedule: with enable != 0 (default) the weight gradient of a unit is queued on an internal
 * stream as soon as the unit's output gradient is final, concurrently with the BatchNorm-backward / data-gradient chain
 * on the caller's stream; every md_plan_backward_range call joins the two before returning to the caller's stream
 * order.  enable == 0 queues everything on the caller's stream.  Same kernels, bit-identical results.  Since round 3 the
 * default is ONE stream (the weight-gradient kernels fill the chip by themselves: profiles/r03_schedule_sweep.txt); the side
 * stream is created on request (enable != 0 here, or MD_WGRAD_STREAM=1). */
int md_plan_use_side_stream(MdPlan* p, int32_t enable);
/* Data-parallel use: md_plan_defer_join(p, 1) makes md_plan_backward_range return WITHOUT ordering the caller's stream
 * after the side stream.  The caller then queues the consumer of a stage's weight gradients (the all-reduce) behind
 * md_plan_side_stream(p) (NULL when there is no side stream: nothing to wait for), so that the collective waits for the
 * weight gradients while the backward chain keeps running, and calls md_plan_join(p, stream) before it reads any
 * gradient on its own stream.  BatchNorm gradients (dgamma, dbeta) are always produced on the caller's stream. */
int md_plan_defer_join(MdPlan* p, int32_t defer);
void* md_plan_side_stream(MdPlan* p);
int md_plan_join(MdPlan* p, void* stream);
/* Measurement aid (bench.py roofline leg): bracket every convolution launch of the plan with HIP events on
 * the launch stream.  md_plan_profile_read sums, per kernel class (0 conv forward, 1 conv data-gradient,
 * 2 conv weight-gradient), the measured milliseconds, the launch count and the algorithmic FLOPs
 * (2 * output pixels * Cout * Cin * taps per launch) since the previous read; it waits on the events.
 * enable: 0 = off (records are kept), 1 = on and forget earlier records, 2 = on, keeping earlier records (to sample
 * some steps of a run: the events cost ~8 % of a step when every launch of every step is bracketed). */
int md_plan_profile_enable(MdPlan* p, int enable);
/* Create the events for `records` bracketed launches now (outside a timed region) instead of on first use. */
int md_plan_profile_reserve(MdPlan* p, int32_t records);
int md_plan_profile_read(MdPlan* p, double* ms, int64_t* launches, double* flops);

/* ------------------------------------------------------------------------------------------------
 * One training step of R2Plus1DClassifier in ONE call: replaces, for this model, the per-step sequence of the reference's
 * loop (src/train.py:40-66: optimizer.zero_grad -> model(data) -> loss_fn -> isfinite check -> loss.backward ->
 * clip_grad_norm_ -> optimizer.step) and the module forward it drives (src/models/R2Plus1D.py:262-265: trunk, then the
 * Linear -> BatchNorm1d -> ELU -> Linear head).  Everything is queued on `stream`; nothing is read back.
 * The reference's "skip the batch when the loss is not finite" (train.py:56-58) is a device flag here: ok_flag[0] = 1.0 when
 * the loss is finite, else 0.0, and the parameter update is applied only when it is 1 (the host looks at the flag later --
 * per epoch -- for the warning and the bookkeeping).  Gradients are WRITTEN (not accumulated) into dw / dgamma / dbeta / the
 * head gradient buffers, which the caller exposes as the parameters' .grad.
 * Tables w, gamma, beta, rmean, rvar, dw, dgamma, dbeta: host arrays of md_plan_num_units() device pointers, as for
 * md_plan_forward / md_plan_backward.  counters: DEVICE array of ncounters device pointers to int64 scalars
 * (BatchNorm num_batches_tracked), each incremented by one; may be NULL.  Optimizer: the tensor / chunk tables of
 * md_opt_grad_norm / md_opt_adamw_step_if over every parameter that is updated (opt_nchunks = 0: no update at all),
 * opt_partial = opt_nchunks + 2 floats ([0] receives the gradient norm, [1] the clip coefficient), max_norm <= 0: no
 * clipping; opt_step = the 1-based step count of the bias correction.
 * loss_kind / class_weight / margins / gamma_or_s as md_softmax_loss.  feat, dfeat: B x md_plan_feat_dim floats; logits,
 * dlogits: B x K; head_save: md_head_save_floats(B, D, Hd); loss: 1 float; pred: B int64 (argmax softmax, train.py:70). */
typedef struct MdTrainStepArgs {
  int32_t B, Hd, K, loss_kind;
  const float* x;              /* (B,3,T,H,W) fp32 */
  const int64_t* target;       /* B */
  const float* const* w; const float* const* gamma; const float* const* beta;
  float* const* rmean; float* const* rvar;
  float* const* dw; float* const* dgamma; float* const* dbeta;
  const float* w0; const float* b0; const float* hgamma; const float* hbeta; const float* w1; const float* b1;
  float* hrmean; float* hrvar;
  float* dw0; float* db0; float* dhgamma; float* dhbeta; float* dw1; float* db1;
  float head_alpha, head_eps, head_momentum, gamma_or_s;
  const float* class_weight; const float* margins;
  float* feat; float* dfeat; float* logits; float* dlogits; float* head_save; float* loss; int64_t* pred;
  void* workspace;             /* md_plan_workspace_bytes */
  int64_t* const* counters; int32_t ncounters;
  int32_t opt_nchunks;
  const void* opt_tensors; const void* opt_chunks; float* opt_partial;
  float max_norm, lr, beta1, beta2, eps, weight_decay;
  int64_t opt_step;
  float* ok_flag;
} MdTrainStepArgs;
int md_plan_train_step(MdPlan* plan, const MdTrainStepArgs* args, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MI355X_DISRUPT_H */
