/*
 * hfagp.h — C ABI of libhfagp_hip.so: the MI355X (gfx950) implementation of the
 * EG3D tri-plane generator hot path that HFA-GP calls as
 *   generator.synthesis(ws, c=label, noise_mode='const')['image']
 * (/root/reference/code/networks/headnerf.py:112,118,133,207,218,267,277).
 *
 * The reference has no FFI of its own (it is pure Python).  The
 * interfaces these entry points replace are the EG3D operator API one level
 * below that call (NVlabs/eg3d, not shipped in the reference tree):
 *   hfagp_raymarch_fwd      <- ImportanceRenderer.forward + RaySampler + OSGDecoder + MipRayMarcher2
 *   hfagp_modconv_fwd       <- modulated_conv2d(...) + bias_act(...) (SynthesisLayer / ToRGBLayer)
 *   hfagp_style_fwd         <- FullyConnectedLayer affine + demodulation coefficients
 *   hfagp_upfir_epilogue_fwd<- upfirdn2d(pad [1,1,1,1], gain 4) after the transposed conv + bias_act
 *   hfagp_upfirdn2d_fwd     <- upfirdn2d.upfirdn2d(x, f, up, down, padding, gain)
 *                              (in-repo twin: upfirdn2d_native, code/networks/encoder3d.py:23-45)
 *   hfagp_bias_act_fwd      <- bias_act.bias_act(x, b, act, alpha, gain, clamp)
 *                              (in-repo twin: fused_leaky_relu, code/networks/encoder3d.py:7-8)
 *   hfagp_torgb_fwd         <- ToRGBLayer for img_channels=3 + upsample2d(img) skip add
 *   hfagp_skip_upsample_add <- img = upsample2d(img, resample_filter) + y   (SynthesisBlock 'skip')
 *   hfagp_torgb_skip_fwd    <- ToRGBLayer (96 tri-plane channels) + the skip add, one streaming pass
 *   hfagp_torgb_finish_fwd  <- bias + clamp + skip add of a 3-channel ToRGBLayer whose sums rode in the conv's epilogue
 *   hfagp_upconv_fir_fwd    <- conv2d_resample(up=2) + bias_act of a SynthesisLayer in one pass (+ _scratch_bytes)
 *   hfagp_style_batch_fwd   <- the affine + demodulation of EVERY layer of one synthesis() call, one launch
 *   hfagp_fc_fwd            <- FullyConnectedLayer (MappingNetwork; EqualLinear twin: code/networks/encoder3d.py:112-139)
 *   hfagp_weight_prep[_split|_prec] <- (no reference counterpart: MFMA operand images of a conv weight, per weight version)
 *   hfagp_qr_gram_fwd / hfagp_qr_refine_fwd <- torch.qr(bases.T) of get_latent (code/networks/headnerf.py:91,187,246)
 *   hfagp_depth_clamp       <- MipRayMarcher2's torch.clamp(depth, min sample depth, max sample depth) over the batch, one launch (ABI 10)
 *   hfagp_planes_to_nhwc    <- planes.view(N, 3, 32, H, W) of TriPlaneGenerator.synthesis (layout change for the gather)
 *   hfagp_nchw_to_nhwc / hfagp_nhwc_to_nchw <- tensor layout at the module boundary (reference tensors are NCHW)
 *   hfagp_pool_mse_fwd/_bwd <- face_pool (AdaptiveAvgPool2d) + MSELoss of gen_update (code/trainer_rgb.py:63,84-85) and its backward
 *   hfagp_blur_down_fwd/_bwd<- Blur + stride-2 sampling in front of the ResBlock skip (code/networks/encoder3d.py:59-73,142-199)
 *   hfagp_allreduce_f32     <- the gradient all-reduce DistributedDataParallel does for the reference
 *                              (code/trainer_rgb.py:56, code/trainer_3dmm.py:29, code/trainer_audio.py:30-34; NCCL group: code/train_rgb.py:57)
 * backward (the reference gets these from torch.autograd through EG3D's custom ops; g_loss.backward(), code/trainer_rgb.py:93):
 *   hfagp_raymarch_bwd      <- autograd of the renderer: grid_sample / decoder / compositing adjoints
 *   hfagp_modconv_fwd modes HFAGP_CONV3X3_BWD / HFAGP_CONVS2_BWD <- conv2d_gradfix data gradients (the same GEMM kernel)
 *   hfagp_conv_wgrad (+ _workspace_bytes) <- conv2d_gradfix weight gradients
 *   hfagp_pointwise_bwd     <- bias_act backward + noise-strength / bias gradients + demodulation adjoint of a SynthesisLayer
 *   hfagp_upfir_bwd, hfagp_upsample2d_bwd, hfagp_upfirdn2d_bwd, hfagp_bias_act_bwd <- upfirdn2d / bias_act backward
 *   hfagp_style_bwd / hfagp_style_batch_bwd, hfagp_affine_grad, hfagp_channel_sum <- affine / demodulation / bias gradients
 *
 * Contract:
 *   - plain pointers and sizes only; all pointers are DEVICE pointers owned by the caller;
 *     the library never allocates, frees or retains them past the call;
 *   - all work is enqueued asynchronously on the hipStream_t passed in (void* here so that
 *     callers need no HIP headers); no internal synchronisation;
 *   - return 0 on success, <0 on error: -1 bad arguments, -2 unsupported shape/dtype,
 *     -3 HIP launch failure; hfagp_last_error() returns a thread-local message;
 *   - never aborts, never throws across the boundary.
 *
 * Activation layout inside the path is channels-last fp32:  x[b][y][x][c].
 * The tri-plane volume is plane-major channels-last:        planes[b][plane][y][x][32].
 */
#ifndef HFAGP_H_
#define HFAGP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HFAGP_ABI_VERSION 12

enum { HFAGP_OK = 0, HFAGP_EBADARG = -1, HFAGP_EUNSUPPORTED = -2, HFAGP_ELAUNCH = -3 };

int hfagp_abi_version(void);
const char* hfagp_last_error(void);

/* ------------------------------------------------------------------ ray march
 * One launch: ray generation -> stratified depths -> tri-plane bilinear gather
 * -> decoder MLP -> coarse compositing -> importance re-sampling -> second
 * gather+MLP -> depth merge -> final compositing.                              */
#define HFAGP_RAYMARCH_STATE_FLOATS_PER_SAMPLE 35   /* HfagpRaymarchArgs::state: 32 colours, depth, density, sort index */

typedef struct {
    const float* planes;      /* [B][3][H][W][32] fp32                                        */
    const float* cam2world;   /* [B][16] row-major 4x4 (label[:, :16])                        */
    const float* intrinsics;  /* [B][9]  row-major 3x3 (label[:, 16:25])                      */
    const float* u_strat;     /* [B][R][Sc]  uniforms of the stratified jitter                */
    const float* u_imp;       /* [B][R][Sf]  uniforms of the inverse-CDF draw                 */
    const float* dec_w0;      /* decoder.net.0.weight [64][32]  (raw parameter)               */
    const float* dec_b0;      /* decoder.net.0.bias   [64]                                    */
    const float* dec_w1;      /* decoder.net.2.weight [33][64]                                */
    const float* dec_b1;      /* decoder.net.2.bias   [33]                                    */
    float*       feat;        /* out [B][R][32]  composited features, scaled to (-1,1)        */
    float*       depth;       /* out [B][R]      expected depth, NaN->inf, NOT yet clamped    */
    float*       wsum;        /* out [B][R]      sum of weights                               */
    float*       tminmax;     /* out [B][R][2]   per-ray min / max sample depth               */
    int32_t B, H, W;          /* plane height/width                                           */
    int32_t res;              /* rays per side; R = res*res                                   */
    int32_t Sc, Sf;           /* coarse / importance samples per ray: 16,32 or 48 each        */
    int32_t plane_axes;       /* 0: (x,y),(x,z),(z,x) [eg3d original]; 1: third = (z,y)       */
    int32_t white_back;
    double ray_start, ray_end;/* python floats of rendering_kwargs (kept double: linspace/delta rounding) */
    float box_warp, decoder_lr_mul;
    /* optional: HFAGP_ABSMAX_FLOATS floats (64 slots, HFAGP_ABSMAX_STRIDE apart) whose maximum bounds |planes| (published by the kernel that wrote the planes,
     * hfagp_skip_upsample_add, or any upper bound).  With it the decoder MLP runs on the 16-bit matrix pipe with split
     * operands (fp16 hi + lo parts, 3 MFMAs per product as HFAGP_PREC_F16X3: ~2^-22, fp32-class; gradients: bf16 parts),
     * every operand scaled into fp16's range by an exact power of two derived from this bound and the weights.
     * NULL: the decoder runs on the exact fp32 matrix instructions (5x the matrix-pipe time).                      */
    const float* planes_absmax;
    /* optional [B][R][(Sc + Sf) * HFAGP_RAYMARCH_STATE_FLOATS_PER_SAMPLE] floats: in hfagp_raymarch_fwd the per-sample colours,
     * densities, depths and sort order of every ray are LEFT here (13.4 KB per ray at 48+48 samples); hfagp_raymarch_bwd
     * given the same buffer (fwd.state) reads them instead of gathering and decoding every sample again for the
     * compositing adjoint.  NULL: nothing is saved / everything is recomputed.                                       */
    float*       state;
} HfagpRaymarchArgs;

int hfagp_raymarch_fwd(const HfagpRaymarchArgs* a, void* stream);

/* ------------------------------------------------------------------ styles
 * styles[b][i] = (w[b] . A[i]) / sqrt(w_dim) * 1 + bias[i]   (then * style_gain)
 * dcoef[b][o]  = rsqrt( sum_i styles[b][i]^2 * wsq[o][i] + eps )   if wsq != NULL */
typedef struct {
    const float* w;           /* [B][w_stride] one latent row per sample                      */
    const float* affine_w;    /* [Cin][w_dim]                                                 */
    const float* affine_b;    /* [Cin]                                                        */
    const float* wsq;         /* [Cout][Cin] = sum_k weight[o][i][k]^2, or NULL (no demod)    */
    float*       styles;      /* out [B][Cin]                                                 */
    float*       dcoef;       /* out [B][Cout] or NULL                                        */
    int32_t B, w_dim, w_stride, Cin, Cout;
    float style_gain, eps;
} HfagpStyleArgs;

int hfagp_style_fwd(const HfagpStyleArgs* a, void* stream);
/* the same for n <= 32 layers in one launch per kernel (styles, then demodulation): the styles of a whole
 * synthesis pass depend only on ws, so the host computes them up front                                    */
int hfagp_style_batch_fwd(const HfagpStyleArgs* items, int32_t n, void* stream);

/* Householder QR of the latent basis A = (bases + 1e-8)^T [m x n], n <= 64 (headnerf.py:85-91), split as
 *   gram = A^T A [n][n] and top = A[0:n, 0:n] [n][n] (row-major, formed by the caller with one GEMM / one copy)
 *   -> R [n][n] upper triangular with LAPACK geqrf's sign convention, Rinv = R^-1;  the caller forms Q = A . Rinv.
 * One small kernel instead of the ~250 launches of a library geqrf + orgqr on a 7168 x 50 panel.            */
int hfagp_qr_gram_fwd(const float* gram, const float* top, float* R, float* Rinv, int32_t n, void* stream);
/* Re-orthogonalisation pass (CholeskyQR2's second step) that removes the cond(A)^2 eps orthogonality defect of the
 * Gram-matrix factorisation: gram = Q1^T Q1 [n][n] of the first pass' Q1 -> upper Cholesky factor R [n][n] (positive
 * diagonal, so the signs of pass 1 = LAPACK's are kept) and Rinv; the caller forms Q = Q1 . Rinv.
 * status (optional, device float[2]): [0] = max |gram - I| = the defect of pass 1 (the host's conditioning monitor:
 * HeadNeRF_* falls back to a library Householder QR when it grows), [1] = 1 when a pivot was not positive — the
 * outputs are then NaN, never silently wrong.                                                                   */
int hfagp_qr_refine_fwd(const float* gram, float* R, float* Rinv, float* status, int32_t n, void* stream);
/* ABI 12: out[n][n] = scale * X^T Y for tall-skinny X, Y [m x n], n <= 64, element (k, i) at k * rs + i * cs (row- or
 * column-major operands): the Gram matrices of the two passes above (A^T A, Q1^T Q1) and -Q^T dQ of the QR's backward, where a
 * GEMM library runs one macro-tile.  workspace: hfagp_tall_gram_workspace_bytes(m, n); fixed summation order.                */
size_t hfagp_tall_gram_workspace_bytes(int32_t m, int32_t n);
int hfagp_tall_gram(const float* X, int64_t x_rs, int64_t x_cs, const float* Y, int64_t y_rs, int64_t y_cs, float* workspace,
                    float* out, int32_t m, int32_t n, float scale, void* stream);

/* FullyConnectedLayer (mapping network): y = act((x . W^T) * lr_mul/sqrt(In) + bias*lr_mul) * gain   [B][Out] */
int hfagp_fc_fwd(const float* x, const float* weight, const float* bias, float* y, int32_t B, int32_t In, int32_t Out,
                 float lr_mul, int32_t act, float alpha, float gain, void* stream);

/* weight preparation (cached by the host while the generator is frozen)
 *   wt  [taps][Cin/4][Cout][4]  <- weight [Cout][Cin][kh][kw]      (MFMA B-operand image)
 *   wsq [Cout][Cin]             <- sum over taps of weight^2                           */
int hfagp_weight_prep(const float* weight, float* wt, float* wsq,
                      int32_t Cout, int32_t Cin, int32_t taps, void* stream);

/* split-bf16 weight image for the HFAGP_PREC_BF16X3 / _BF16X6 conv path:
 *   wb [nparts][taps][Cin/8][Cout][8] bf16  <- weight [Cout][Cin][kh][kw],  w = part0 + part1 (+ part2),
 *   each part the round-to-nearest bf16 of the residual left by the parts before it (nparts = 2 or 3).
 *   nparts = 1: the HFAGP_PREC_F16 image, same layout, elements rounded to IEEE fp16.
 *   When Cout % 128 != 0 (the 96-channel toRGB) the buffer must extend 512 bytes past the image: the 128-wide
 *   tile reads (and discards) one partial row beyond it.                                                   */
int hfagp_weight_prep_split(const float* weight, void* wb, int32_t Cout, int32_t Cin, int32_t taps,
                            int32_t nparts, void* stream);
/* the same image selected by HFAGP_PREC_* (F16: 1 fp16 part, BF16X3 / BF16X6: 2 / 3 bf16 parts,
 * F16X3: 2 fp16 parts = round-to-nearest fp16 of the weight and of its residual)                           */
int hfagp_weight_prep_prec(const float* weight, void* wb, int32_t Cout, int32_t Cin, int32_t taps,
                           int32_t precision, void* stream);

/* ABI 11: everything a step needs of a set of conv weights in ONE launch — for each item the forward image (layout of
 * hfagp_weight_prep_prec, `precision`), the image of the Cin/Cout TRANSPOSE (`precision_t`; the bwd-data GEMM's operand) and
 * wsq[Cout][Cin] = sum over taps of w^2; any of the three outputs may be NULL.  Cout, Cin multiples of 32; at most 48 items.
 * For a generator that is being tuned: its weights change every step (trainer_rgb.py:69-71), its images with them. */
typedef struct {
    const float* weight;      /* [Cout][Cin][taps] */
    void*        image;       /* [parts][taps][Cin/8][Cout][8] or NULL */
    void*        image_t;     /* [parts][taps][Cout/8][Cin][8] or NULL */
    float*       wsq;         /* [Cout][Cin] or NULL */
    int32_t Cout, Cin, taps;
    int32_t precision, precision_t;
} HfagpWeightPrepItem;
int hfagp_weight_prep_batch(const HfagpWeightPrepItem* items, int32_t n, void* stream);

/* ------------------------------------------------------------------ modulated conv
 * Implicit-GEMM on v_mfma_f32_32x32x2_f32 (exact fp32).  Input is scaled by
 * styles on the way into LDS; demodulation, noise, bias, leaky-ReLU, gain and
 * clamp are applied in the epilogue (mode 0/2) or by hfagp_upfir_epilogue_fwd
 * (mode 1, which writes the raw transposed-conv result).                          */
enum {
    HFAGP_CONV3X3 = 0,      /* y = corr3x3(x*s, W), padding 1, fused epilogue                               */
    HFAGP_CONVT3X3_UP2 = 1, /* y_t = conv_transpose2d(x*s, W, stride 2): RAW [B][2H+1][2W+1][Cout]          */
    HFAGP_CONV1X1 = 2,      /* y = x*s . W (toRGB with many output channels), fused epilogue               */
    HFAGP_CONV3X3_BWD = 3,  /* adjoint of mode 0 w.r.t. (x*s): x = grad [B][H][W][Cout_fwd], wt = prep of   */
                            /* weight^T (Cin/Cout swapped), Cin = Cout_fwd, Cout = Cin_fwd                  */
    HFAGP_CONVS2_BWD = 4    /* adjoint of mode 1 w.r.t. (x*s): x = 4 parity images of the y_t gradient      */
                            /* [2][2][B][H+1][W+1][Cin] (hfagp_upfir_bwd), H, W = resolution of dx          */
};
enum { HFAGP_ACT_LINEAR = 0, HFAGP_ACT_LRELU = 1 };
/* arithmetic of the GEMM (accumulation is always fp32):
 *   F32     v_mfma_f32_32x32x2_f32, exact fp32 products
 *   BF16X3  operands split into 2 bf16 parts (hi+lo), 3 v_mfma_f32_32x32x16_bf16 per product
 *           (hi.hi + lo.hi + hi.lo): relative product error ~2^-16
 *   BF16X6  3 parts, 6 MFMAs per product: relative product error ~2^-23 (fp32 class)
 *   F16X3   operands split into 2 fp16 parts (11 + 11 mantissa bits), 3 v_mfma_f32_32x32x16_f16 per product
 *           (hi.hi + lo.hi + hi.lo): relative product error ~2^-22 — fp32 class at the cost of BF16X3
 *   F16     operands rounded to fp16, ONE v_mfma_f32_32x32x16_f16 per product: relative product error ~2^-11.
 *           The arithmetic EG3D's CUDA path uses in its fp16 blocks (super-resolution, sr_num_fp16_res = 4:
 *           SURVEY.md U4) except that tensors stay fp32 in HBM and accumulation is fp32.  The caller keeps
 *           The fp16 kinds keep |x * style| <= |x| inside the kernel (styles scaled by a power of two per sample,
 *           undone on the accumulators: EG3D's fp16 pre-normalisation, exact); |x| itself must stay below 65504 unless
 *           the tensor's maximum is passed in x_absmax (then any fp32 magnitude is taken, see HfagpModconvArgs).
 * The 16-bit paths need Cin % 16 == 0 and Cout % 128 == 0 — or, except for HFAGP_CONVT3X3_UP2, Cout % 128 >= 96
 * (the 96-channel toRGB: computed on a 128-wide tile whose last columns are discarded) — HFAGP_EUNSUPPORTED otherwise.
 * HFAGP_PREC_F16X2 (ABI 9): the F16X3 weight image (two fp16 parts, 22 bits) against activations rounded to ONE fp16 part —
 *           two MFMAs per product.  The class of TF32 (11-bit x 11-bit), which the reference's cuDNN convolutions use on
 *           Ampere-class GPUs; NOT the default.  hfagp_torgb_skip_fwd / hfagp_upconv_fir_fwd run it as F16X3.               */
enum { HFAGP_PREC_F32 = 0, HFAGP_PREC_BF16X3 = 1, HFAGP_PREC_BF16X6 = 2, HFAGP_PREC_F16 = 3, HFAGP_PREC_F16X3 = 4,
       HFAGP_PREC_F16X2 = 5 };

typedef struct {
    const float* x;           /* [B][H][W][Cin]; x_batch_stride (elements) may be 0 (const)   */
    const void*  wt;          /* hfagp_weight_prep (F32) or hfagp_weight_prep_prec (16-bit precisions) */
    const float* styles;      /* [B][Cin] or NULL                                             */
    const float* dcoef;       /* [B][Cout] or NULL                                            */
    const float* noise;       /* [Ho][Wo] or NULL                                             */
    const float* bias;        /* [Cout] or NULL                                               */
    float*       y;           /* mode 0/2: [B][H][W][Cout]; mode 1: [B][2H+1][2W+1][Cout] raw */
    float*       workspace;   /* split-K partials, >= hfagp_modconv_workspace_bytes()          */
    int64_t x_batch_stride;
    int32_t B, H, W, Cin, Cout;
    int32_t mode, act;
    int32_t ksplit;           /* 0 = let the library choose                                   */
    float noise_strength, alpha, gain, clamp;   /* clamp < 0: none                            */
    int32_t precision;        /* HFAGP_PREC_*                                                 */
    /* fp16 range tracking (optional, both may be NULL).  A tensor without a clamp (EG3D's fp32 backbone,
     * conv_clamp = None) has no bound, and fp16 parts saturate at 65504 and lose bits below 2^-14: the producer of
     * such a tensor publishes max |y| into y_absmax (HFAGP_ABSMAX_FLOATS floats = 64 slots, zeroed by the caller before the
     * launch; blocks write different slots, the maximum over the slots is the tensor's), and the fp16 kinds (F16X3,
     * F16) that consume it read x_absmax and scale the operand by the power of two that brings max |x| to 2^15 —
     * undone on the accumulators, so the result is that of un-scaled arithmetic and nothing saturates.            */
    const float* x_absmax;    /* [HFAGP_ABSMAX_FLOATS] max |x| of the input tensor, or NULL (then |x| <= 65504 is the caller's promise) */
    float*       y_absmax;    /* [HFAGP_ABSMAX_FLOATS] receives max |y| of the fused-epilogue output, or NULL */
    /* fused toRGB of a block's last conv (optional, both or none; 16-bit precisions, modes 0 / 2 without split-K —
     * hfagp_modconv_workspace_bytes() == 0): while the output tile is still in registers the block also forms
     *   rgb[r] = sum_co y[co] * rgb_w[b][r][co]     (r < 3; rgb_w = toRGB weight * its styles, Cout entries per r)
     * over ITS output channels and writes the partial sums to rgb_part; hfagp_torgb_finish_fwd adds the parts, the
     * bias, the clamp and the up-sampled previous image.  Saves the separate toRGB pass over the activation.    */
    /* (with rgb_part, y may be NULL: the activation itself is not stored — the last super-resolution layer of a
     *  forward-only call, whose output feeds nothing but this toRGB)                                                */
    const float* rgb_w;       /* [B][3][Cout] or NULL */
    float*       rgb_part;    /* [hfagp_modconv_rgb_parts()][B][H][W][4] floats (3 used), written, or NULL */
    /* fp16 STORAGE of the activations (EG3D's fp16 blocks keep them in fp16: super-resolution, sr_num_fp16_res = 4):
     * x_f16 = 1: x holds IEEE fp16 values (same shape; pass the pointer as x), y_f16 = 1: y is written as fp16.
     * Only with precision HFAGP_PREC_F16, modes 0 and 1, no split-K (hfagp_modconv_workspace_bytes() == 0); the styles
     * are applied with packed fp16 multiplies and staging is a copy instead of a conversion.                        */
    int32_t x_f16, y_f16;
} HfagpModconvArgs;
/* number of partial-sum images a call with rgb_part writes: (Cout / 128) x 2 */
int32_t hfagp_modconv_rgb_parts(const HfagpModconvArgs* a);
#define HFAGP_ABSMAX_SLOTS 64
/* an absmax buffer is HFAGP_ABSMAX_FLOATS floats: slot i lives at float index i * HFAGP_ABSMAX_STRIDE (one 128-byte
 * line per slot, so the publishing waves' atomics spread over the L2 channels instead of queueing on two lines) */
#define HFAGP_ABSMAX_STRIDE 32
#define HFAGP_ABSMAX_FLOATS (HFAGP_ABSMAX_SLOTS * HFAGP_ABSMAX_STRIDE)

size_t hfagp_modconv_workspace_bytes(const HfagpModconvArgs* a);
int hfagp_modconv_fwd(const HfagpModconvArgs* a, void* stream);

/* FIR (4x4 [1,3,3,1]^2/64, pad [1,1,1,1], gain 4) over the raw transposed-conv
 * output + demod + noise + bias + act.  yt [B][2H+1][2W+1][C] -> y [B][2H][2W][C] */
typedef struct {
    const void*  yt;          /* fp32, or fp16 when io_f16 */
    const float* dcoef;       /* [B][C] or NULL */
    const float* noise;       /* [2H][2W] or NULL */
    const float* bias;        /* [C] or NULL */
    void*        y;           /* fp32, or fp16 when io_f16 */
    int32_t B, H, W, C;       /* H, W = INPUT resolution of the up-conv */
    int32_t act;
    float noise_strength, alpha, gain, clamp;
    float*       y_absmax;    /* optional [HFAGP_ABSMAX_FLOATS]: max |y| (see HfagpModconvArgs) */
    int32_t      io_f16;      /* 1: yt and y are IEEE fp16 tensors (same shapes; HfagpModconvArgs::x_f16 / y_f16), math stays fp32 */
} HfagpUpfirEpilogueArgs;

int hfagp_upfir_epilogue_fwd(const HfagpUpfirEpilogueArgs* a, void* stream);

/* The whole up-sampling layer in ONE pass over its output (EG3D conv2d_resample(up = 2) + bias_act; replaces
 * hfagp_modconv_fwd(mode HFAGP_CONVT3X3_UP2) + hfagp_upfir_epilogue_fwd):
 *   y [B][2H][2W][Cout] = act(FIR(conv_transpose2d(x * styles, W, stride 2)) * dcoef + noise + bias) * gain, clamp
 * The raw transposed-conv result never goes to HBM: a block owns a strip of 32 y_t columns, walks down a segment of
 * tiles and filters each y_t tile in LDS; the three output columns at every strip boundary and the three output rows at
 * every segment boundary are finished by a second, small kernel from raw strips in `scratch`.
 * `a` as for hfagp_modconv_fwd with mode = HFAGP_CONVT3X3_UP2, except that y is the FINAL tensor, dcoef / noise / bias / act /
 * alpha / gain / clamp / y_absmax apply (as in HfagpUpfirEpilogueArgs), workspace and ksplit are ignored and there is no
 * fused toRGB.  Precisions BF16X3, F16X3, F16 (x_f16 / y_f16 storage allowed with F16); Cin % 16 == 0, Cin <= 512,
 * Cout % 128 == 0; the launch must fill the chip (B * ceil((W+1)/16) * Cout/128 >= 256 strips).
 * Cin == 32 (the first super-resolution layer; Cout % 64 == 0, fp32 storage, precisions F16 / BF16X3 / F16X3 / F16X2) takes a
 * STREAMING kernel instead (round 6, csrc/upfir_lean.hip: FIR in registers, no scratch use, no strip kernel) at any batch.
 * hfagp_upconv_fir_scratch_bytes(): bytes of `scratch` the call needs (a non-zero token size where the streaming kernel runs:
 * the pointer must still be non-NULL), or 0 when the shape is not supported — the caller then uses the two-call form.        */
size_t hfagp_upconv_fir_scratch_bytes(const HfagpModconvArgs* a);
int hfagp_upconv_fir_fwd(const HfagpModconvArgs* a, void* scratch, void* stream);

/* skip connection: img_out = upsample2d(img_in) + y  (both channels-last, C channels;
 * img_in may be NULL -> img_out = y).  Optional plane-major output for the last block:
 * planes_out [B][3][2H][2W][C/3] when plane_major != 0.                              */
typedef struct {
    const float* img_in;      /* [B][H][W][C] or NULL */
    const float* y;           /* [B][2H][2W][C] (or [B][H][W][C] when img_in is NULL) */
    float*       img_out;
    int32_t B, H, W, C;       /* H, W = resolution of img_in */
    int32_t plane_major;
    float*       out_absmax;  /* optional [HFAGP_ABSMAX_FLOATS]: receives max |img_out| (HfagpRaymarchArgs::planes_absmax) */
} HfagpSkipArgs;

int hfagp_skip_upsample_add(const HfagpSkipArgs* a, void* stream);

/* toRGB of a backbone block with its skip connection in ONE streaming pass (EG3D SynthesisBlock.forward, architecture
 * 'skip': `img = upsample2d(img); y = self.torgb(x, ws); img = img.add_(y)`), for the 96-channel tri-plane image:
 *   img_out[b][p][co] = sum_i x[b][p][i] styles[b][i] w[co][i] + bias[co] + upsample2d(img_in)[b][p][co]
 * on the 16-bit matrix pipe with split operands (precision = HFAGP_PREC_F16X3 / BF16X3 / BF16X6 / F16, as hfagp_modconv_fwd),
 * every wave streaming 32 positions x Cin channels straight from HBM into MFMA operands; the toRGB output never
 * goes through memory.  Cout a multiple of 32 (<= 128), Cin a multiple of 16 (<= 512), W a multiple of 32, H*W of 128;
 * other shapes: hfagp_modconv_fwd (HFAGP_CONV1X1) + hfagp_skip_upsample_add.                                         */
typedef struct {
    const float* x;           /* [B][H][W][Cin] channels-last */
    const void*  wt;          /* hfagp_weight_prep_prec image of the [Cout][Cin][1][1] weight (taps = 1) */
    const float* styles;      /* [B][Cin] (already * 1/sqrt(Cin)) */
    const float* bias;        /* [Cout] */
    const float* img_in;      /* [B][H/2][W/2][Cout] channels-last, or NULL (first block: img_out = toRGB) */
    float*       img_out;     /* [B][H][W][Cout], or [B][3][H][W][Cout/3] when plane_major (needs Cout = 96) */
    const float* x_absmax;    /* optional [HFAGP_ABSMAX_FLOATS]: max |x| as published by x's producer (fp16 kinds) */
    float*       out_absmax;  /* optional [HFAGP_ABSMAX_FLOATS]: receives max |img_out| */
    int32_t B, H, W, Cin, Cout;
    int32_t precision;        /* HFAGP_PREC_* of the weight image (not F32) */
    int32_t plane_major;
} HfagpTorgbSkipArgs;

int hfagp_torgb_skip_fwd(const HfagpTorgbSkipArgs* a, void* stream);

/* toRGB with few output channels (super-resolution, img_channels = 3):
 * rgb_out[b][c][y][x] (NCHW) = sum_i x[b][y][x][i]*styles[b][i]*w[c][i] + bias[c]
 *                              (clamped) + upsample2d(rgb_in)[b][c][y][x]           */
typedef struct {
    const float* x;           /* [B][H][W][Cin] channels-last */
    const float* weight;      /* [Cout][Cin] (1x1) */
    const float* styles;      /* [B][Cin] (already * 1/sqrt(Cin)) */
    const float* bias;        /* [Cout] */
    const float* rgb_in;      /* NCHW [B][Cout][H/2][W/2] or NULL */
    float*       rgb_out;     /* NCHW [B][Cout][H][W] */
    float*       y_pre;       /* optional NCHW [B][Cout][H][W]: toRGB output before the skip add (for backward) */
    int32_t B, H, W, Cin, Cout;
    float clamp;
} HfagpTorgbArgs;

int hfagp_torgb_fwd(const HfagpTorgbArgs* a, void* stream);

/* second half of the fused toRGB (HfagpModconvArgs::rgb_part): rgb_out[b][c][y][x] (NCHW) =
 * clamp(sum_p part[p][b][y][x][c] + bias[c]) + upsample2d(rgb_in)[b][c][y][x]                         */
typedef struct {
    const float* part;        /* [nparts][B][H][W][4] */
    const float* bias;        /* [Cout] */
    const float* rgb_in;      /* NCHW [B][Cout][H/2][W/2] or NULL */
    float*       rgb_out;     /* NCHW [B][Cout][H][W] */
    float*       y_pre;       /* optional NCHW [B][Cout][H][W]: value before the clamp (backward mask) */
    int32_t nparts, B, H, W, Cout;   /* Cout <= 3 */
    float clamp;
} HfagpTorgbFinishArgs;
int hfagp_torgb_finish_fwd(const HfagpTorgbFinishArgs* a, void* stream);

/* ------------------------------------------------------------------ backward pass (generator frozen: d/d ws)
 * GEMM-shaped parts reuse hfagp_modconv_fwd (modes 3, 4 and 2 with transposed weights).               */

/* One fused streaming pass per activation tensor X [B][H][W][C] (output of layer P, input of its
 * consumers): sums the consumers' input gradients, reduces their style gradients, and pushes the result
 * through P's clamp / gain / leaky-ReLU / demodulation.
 *   gX      = dxs_conv*s_conv + dxs_rgb*s_rgb + s_small * (w_rgb_small^T g_rgb_small) + g_direct + [c < 3] (g_nchw3_a + g_nchw3_b)
 *             (g_rgb_small is masked with [|y_rgb_small| < clamp_rgb_small] when y_rgb_small is given: the clamp of the small toRGB)
 *   g_out   = has_producer ? gX * gain * lrelu'(X) * [|X|<clamp] * dcoef_p : gX
 *   sums[b][0] = sum_pix dxs_conv*X   sums[b][1] = sum_pix dxs_rgb*X   sums[b][2] = sum_pix (w^T g)*X
 *   sums[b][3] = sum_pix g_pre * conv_p      (gradient of P's demodulation coefficients)
 * and, when param_grads != 0 (generator being tuned):
 *   sums[b][4] = sum_pix g_pre (bias of P)   sums[b][5] = sum_pix g_pre * noise_p (per channel: d noise_strength)
 *   sums[b][6+c] = sum_pix g_rgb_small[c] * X   (small-toRGB weight gradient before the style factor)
 * sums is [B][10][C].                                                                                    */
typedef struct {
    const float* dxs_conv;    /* [B][H][W][C] or NULL */
    const float* s_conv;      /* [B][C] */
    const float* dxs_rgb;     /* [B][H][W][C] or NULL */
    const float* s_rgb;       /* [B][C] */
    const float* g_rgb_small; /* NCHW [B][Co][H][W] (already clamp-masked) or NULL; ABI 12: exclusive with dxs_rgb (a layer has one toRGB) */
    const float* w_rgb_small; /* [Co][C] */
    const float* s_small;     /* [B][C] */
    const float* g_direct;    /* [B][H][W][C] extra gradient added as is, or NULL */
    const float* x;           /* [B][H][W][C] saved activation */
    const float* dcoef_p;     /* [B][C] or NULL */
    const float* bias_p;      /* [C] or NULL */
    const float* noise_p;     /* [H][W] or NULL */
    float*       g_out;       /* [B][H][W][C] */
    float*       partial;     /* workspace [B][nchunks][10][C] */
    float*       sums;        /* out [B][10][C] */
    int32_t B, H, W, C, Co, nchunks, has_producer, act_p, param_grads;
    float noise_strength_p, alpha, gain, clamp;
    /* ABI 10 (all optional): */
    const float* y_rgb_small; /* NCHW [B][Co][H][W]: the small toRGB's pre-clamp output; with clamp_rgb_small >= 0 the kernel masks
                                 g_rgb_small itself (the caller then passes the UN-masked gradient) */
    const float* g_nchw3_a;   /* NCHW [B][3][H][W] gradients added to channels 0..2 of gX (image_raw = first three channels of  */
    const float* g_nchw3_b;   /* the feature image: the super-resolution skip path and the caller's d image_raw), or NULL          */
    float clamp_rgb_small;
    /* ABI 11 (optional): the producer's noise strength read from DEVICE memory (one float) instead of `noise_strength_p` — a generator
     * that is being tuned changes it every step, and a host copy would cost a device-to-host synchronisation per step */
    const float* noise_strength_dev;
} HfagpPointwiseBwdArgs;

int hfagp_pointwise_bwd(const HfagpPointwiseBwdArgs* a, void* stream);

/* ABI 12: hfagp_pointwise_bwd with sums = NULL leaves its per-chunk partial sums un-reduced; this reduces up to 32 such buffers in
 * one launch: sums[b][k] = sum over chunks of partial[b][chunk][k], k < n = 10 * C, in a fixed order (not the single-pass reducer's: the last bits may
 * differ) (a backward pass with the generator frozen consumes the sums only at its end: 19 launches -> 1).                               */
typedef struct {
    const float* partial;     /* [B][nchunks][n] */
    float*       sums;        /* [B][n] */
    int32_t B, nchunks, n;
} HfagpReducePartialsItem;
int hfagp_reduce_partials_batch(const HfagpReducePartialsItem* items, int32_t n, void* stream);

/* MipRayMarcher2's depth clamp (EG3D: `torch.clamp(depth, min(sample depths), max(sample depths))` over the WHOLE batch) in one
 * launch: depth [n] is clamped in place to [min_i tminmax[i][0], max_i tminmax[i][1]] (tminmax as hfagp_raymarch_fwd writes it).
 * Up to 64 workgroups, each reducing all of tminmax itself (L2-resident) and clamping its slice; any batch since ABI 11. */
int hfagp_depth_clamp(float* depth, const float* tminmax, int64_t n, void* stream);

/* adjoint of hfagp_upfir_epilogue_fwd's FIR: g_y [B][2H][2W][C] -> four parity images of the y_t gradient,
 * gph [2][2][B][H+1][W+1][C] (input of hfagp_modconv_fwd mode HFAGP_CONVS2_BWD)                          */
int hfagp_upfir_bwd(const float* gy, float* gph, int32_t B, int32_t H, int32_t W, int32_t C, void* stream);

/* adjoint of upsample2d: g [outer][2H][2W][inner] -> gin [outer][H][W][inner] (inner = C for channels-last,
 * 1 for NCHW with outer = B*C)                                                                           */
int hfagp_upsample2d_bwd(const float* g, float* gin, int64_t outer, int32_t H, int32_t W, int32_t inner, void* stream);

/* plane-major [B][3][H][W][Cp] -> channels-last [B][H][W][3*Cp] */
int hfagp_planes_to_nhwc(const float* pm, float* y, int32_t B, int32_t H, int32_t W, int32_t Cp, void* stream);

/* styles -> latent: dstot = (ds - styles * sum_o dd_o d_o^3 wsq[o][i]) * style_gain;
 * dw[b][:] (+)= dstot[b] . affine_w / sqrt(w_dim)                                                         */
typedef struct {
    const float* ds;          /* [B][Cin] gradient w.r.t. the styles (data path) */
    const float* dd;          /* [B][Cout] gradient w.r.t. the demod coefficients, or NULL */
    const float* styles;      /* [B][Cin] */
    const float* dcoef;       /* [B][Cout] */
    const float* wsq;         /* [Cout][Cin] */
    const float* affine_w;    /* [Cin][w_dim] */
    float*       dstot;       /* workspace [B][Cin] */
    float*       dw;          /* [B][dw_stride] row of d ws */
    int32_t B, Cin, Cout, w_dim, dw_stride, accumulate;
    float style_gain;
} HfagpStyleBwdArgs;

int hfagp_style_bwd(const HfagpStyleBwdArgs* a, void* stream);

/* the same for every affine layer of a backward pass in two launches (at most 32 items, sorted by `dw`: layers that
 * share a row of d ws are accumulated in item order by one block, so the result is deterministic).  ds / dd are read
 * with a row stride (elements), so the reductions of hfagp_pointwise_bwd can be passed without a copy; every dw row
 * is accumulated into (+=).                                                                                  */
typedef struct {
    const float* ds;          /* [B][Cin], row stride ds_stride */
    const float* dd;          /* [B][Cout], row stride dd_stride, or NULL (no demodulation) */
    const float* styles;      /* [B][Cin] */
    const float* dcoef;       /* [B][Cout] or NULL */
    const float* wsq;         /* [Cout][Cin] or NULL */
    const float* affine_w;    /* [Cin][w_dim] */
    float*       dstot;       /* out [B][Cin] */
    float*       dw;          /* row of d ws: dw[b * dw_stride + k] += ... */
    int32_t B, Cin, Cout, w_dim, dw_stride, ds_stride, dd_stride;
    float style_gain;
} HfagpStyleBwdItem;
int hfagp_style_batch_bwd(const HfagpStyleBwdItem* items, int32_t n, void* stream);

/* backward of hfagp_raymarch_fwd w.r.t. the tri-plane volume: recomputes the forward per ray, then
 * scatters d feat -> d planes with fp32 atomics (d_planes must be zero-initialised by the caller).
 * With plane_axes = 0 on square planes, planes 1 (x,z) and 2 (z,x) receive mirrored gradients: only plane 1
 * is scattered and plane 2 is WRITTEN as its transpose (a third fewer atomics).                            */
typedef struct {
    HfagpRaymarchArgs fwd;    /* same inputs as the forward call (feat/depth/wsum/tminmax unused) */
    const float* g_feat;      /* [B][R][32] gradient of the composited features */
    float*       d_planes;    /* [B][3][H][W][32], accumulated into */
    float*       rec;         /* workspace [B][R][Sc+Sf][4] floats (per-sample depth, omega, d sigma) */
    /* optional (all four or none): gradients of the decoder parameters, zero-initialised, accumulated into */
    float*       d_dec_w0;    /* [64][32] */
    float*       d_dec_b0;    /* [64] */
    float*       d_dec_w1;    /* [33][64] */
    float*       d_dec_b1;    /* [33] */
    /* optional (ABI 10): scratch [B][R][Sc+Sf][32] floats.  Given (and the column variant applies: plane_axes = 0, square planes
     * up to 256^2), pass 2 runs as two kernels — dL/dF of every sample into the scratch buffer, then the scatter alone — instead
     * of one whose waves alternate between the two in lock step; same arithmetic, the atomic order differs as it does run to run. */
    float*       df_scratch;
    /* optional (ABI 12): scratch of hfagp_raymarch_bwd_rows_bytes(&fwd) bytes.  Given, pass 2 runs as SORT + GATHER instead of
     * a per-sample scatter: the samples are counting-sorted by (frame, plane, 32-texel column strip, texel row), dL/dF of every
     * sample goes to its sorted slot, and every output row tile is a dense product  W[32 texels x 16 samples] . dL/dF[16 x 32]
     * on the 16-bit matrix pipe (split bf16 operands), W being grid_sample's bilinear weights.  d_planes receives two adds per
     * element and bin chunk instead of ~12 atomics per sample; takes precedence over df_scratch.  Decoder gradients (d_dec_*)
     * come from the same pass as dL/dF.  The buffer's contents are undefined before and after the call.                       */
    void*        rows_scratch;
    uint64_t     rows_scratch_bytes;
} HfagpRaymarchBwdArgs;

int hfagp_raymarch_bwd(const HfagpRaymarchBwdArgs* a, void* stream);
/* bytes of HfagpRaymarchBwdArgs::rows_scratch for this forward configuration (B, H, W, res, Sc, Sf, plane_axes are read);
 * 0 = the sort + gather form does not apply (more than 8192 bins per frame: planes beyond ~256 x 341 texels, or more than
 * 2^31 slots): leave rows_scratch NULL.  At 2 frames x 128^2 rays x 96 samples on mirrored 256^2 planes: 1.8 GB.            */
size_t hfagp_raymarch_bwd_rows_bytes(const HfagpRaymarchArgs* fwd);

/* ------------------------------------------------------------------ gradients w.r.t. the generator weights
 * (needed once HFA-GP calls tune_generator(), trainer_rgb.py:69-71)                                       */

/* dweight[Cout][Cin][k][k] = conv-weight-gradient( x * styles , g )  -  weight * sum_b dd d^3 styles^2
 * (second term: through the demodulation coefficients; dd may be NULL).
 *   mode HFAGP_CONV3X3      : g = gradient w.r.t. the raw conv output            [B][H][W][Cout]
 *   mode HFAGP_CONVT3X3_UP2 : g = parity images from hfagp_upfir_bwd             [2][2][B][H+1][W+1][Cout]
 *   mode HFAGP_CONV1X1      : g = gradient w.r.t. the toRGB output               [B][H][W][Cout]            */
typedef struct {
    const float* x;           /* [B][H][W][Cin] layer input */
    const float* styles;      /* [B][Cin] or NULL */
    const float* g;
    const float* weight;      /* [Cout][Cin][k][k] */
    const float* dd;          /* [B][Cout] or NULL */
    const float* dcoef;       /* [B][Cout] */
    float*       dweight;     /* out [Cout][Cin][k][k] (overwritten) */
    float*       workspace;   /* >= hfagp_wgrad_workspace_bytes() */
    int32_t B, H, W, Cin, Cout, mode;
    int32_t ksplit;           /* number of split-K slabs over (b, position tiles); >= 1 */
    int32_t precision;        /* HFAGP_PREC_F32 (exact fp32 MFMA) or HFAGP_PREC_BF16X3: modes 0 and 1 with Cin, Cout    */
                              /* multiples of 64 (mode 1 also Cin = 32) then run on the split-bf16 MFMA kernels, the rest fp32 */
    int32_t accumulate;       /* ABI 11: 1 = dweight += (the parameter's .grad slice: no separate add pass), 0 = overwrite */
    int32_t dd_stride;        /* ABI 12: row stride of dd in elements (0 = Cout): row 3 of hfagp_pointwise_bwd's sums without a copy */
} HfagpWgradArgs;

/* ABI 11: the library's split-K choice for this layer (everything but `ksplit` / `workspace` filled in): one block per CU for the
 * kernel that will run (64 x 64 (ci, co) tiles), never more slabs than position tiles */
int32_t hfagp_wgrad_ksplit(const HfagpWgradArgs* a);
size_t hfagp_wgrad_workspace_bytes(const HfagpWgradArgs* a);
int hfagp_conv_wgrad(const HfagpWgradArgs* a, void* stream);

/* affine layer: dA[Cin][w_dim] += dstot^T . w / sqrt(w_dim);  db[Cin] += sum_b dstot   (dstot from hfagp_style_bwd) */
int hfagp_affine_grad(const float* dstot, const float* w, float* dA, float* db, int32_t B, int32_t Cin, int32_t w_dim,
                      int32_t w_stride, void* stream);

/* ABI 12: the same for up to 32 affine layers in one launch (every dA / db accumulated into: the .grad slices themselves) */
typedef struct {
    const float* dstot;       /* [B][Cin] */
    const float* w;           /* row view [B][w_dim] of ws, row stride w_stride */
    float*       dA;          /* [Cin][w_dim] += */
    float*       db;          /* [Cin] += */
    int32_t B, Cin, w_dim, w_stride;
} HfagpAffineGradItem;
int hfagp_affine_grad_batch(const HfagpAffineGradItem* items, int32_t n, void* stream);

/* ABI 12: bias and noise-strength gradients of up to 32 synthesis layers in one launch, from the reductions hfagp_pointwise_bwd
 * (param_grads = 1) left in sums [B][10][C]:  dbias[c] += sum_b sums[b][4][c],  dnoise[0] += sum_{b,c} sums[b][5][c]
 * (either pointer may be NULL); fixed summation order.  Replaces two framework reductions + adds per layer.                  */
typedef struct {
    const float* sums;
    float*       dbias;
    float*       dnoise;
    int32_t B, C;
} HfagpBiasNoiseGradItem;
int hfagp_bias_noise_grads(const HfagpBiasNoiseGradItem* items, int32_t n, void* stream);

/* out[c] (+)= sum over npix rows of a [npix][C] channels-last tensor (C <= 256); partial: [nblocks][C] workspace;
 * ABI 11: nblocks * 256 must be a multiple of C (the tensor is walked as a flat array and a thread keeps its channel) */
int hfagp_channel_sum(const float* g, float* partial, float* out, int64_t npix, int32_t C, int32_t nblocks,
                      int32_t accumulate, void* stream);

/* ------------------------------------------------------------------ loss side of the fitting step
 * trainer_rgb.py:84-86 / trainer_3dmm.py:51-53 / trainer_audio.py:97-99:
 *   pooled = AdaptiveAvgPool2d((h, w))(img)  with img [BC][f*h][f*w] (NCHW, BC = B*C, integer factor f),
 *   loss   = mean((real - pooled)^2)          (MSELoss(reduction='mean'); deterministic reduction order)
 * and its adjoint  d_img = g_loss * 2 (pooled - real) / (BC*h*w * f^2)  (every element of d_img is written).
 * workspace: hfagp_pool_mse_workspace_bytes() bytes; loss, g_loss: one float on the device.              */
size_t hfagp_pool_mse_workspace_bytes(void);

/* ABI 11: torch.optim.Adam's update (trainer_rgb.py:58; no weight decay, no amsgrad) of MANY tensors in one call.
 *   tensor_table: device memory, ntensors x 6 64-bit words {param*, grad*, exp_avg*, exp_avg_sq*, step* (ONE float, advanced by
 *                 this call), numel};   chunk_table: device memory, nchunks x {int32 tensor index, int32 first element}, one entry
 *                 per hfagp_adam_chunk() elements of every tensor.  All tensors fp32, contiguous. */
int hfagp_adam_step(const void* tensor_table, const void* chunk_table, int32_t ntensors, int32_t nchunks, double lr, double beta1,
                    double beta2, double eps, void* stream);      /* (doubles: torch forms 1 - beta and the bias corrections in double) */
int32_t hfagp_adam_chunk(void);
int hfagp_pool_mse_fwd(const float* img, const float* real, float* pooled, float* loss, float* workspace,
                       int32_t BC, int32_t h, int32_t w, int32_t f, void* stream);
int hfagp_pool_mse_bwd(const float* pooled, const float* real, const float* g_loss, float* d_img,
                       int32_t BC, int32_t h, int32_t w, int32_t f, void* stream);

/* ------------------------------------------------------------------ standalone ops (NCHW, test surface) */
int hfagp_upfirdn2d_fwd(const float* x, const float* f, float* y,
                        int32_t N, int32_t C, int32_t H, int32_t W, int32_t fh, int32_t fw,
                        int32_t up, int32_t down, int32_t px0, int32_t px1, int32_t py0, int32_t py1,
                        float gain, void* stream);

int hfagp_bias_act_fwd(const float* x, const float* b, float* y, int64_t n, int32_t C, int64_t inner,
                       int32_t act, float alpha, float gain, float clamp, void* stream);
/* adjoints of the two operators above (EG3D's upfirdn2d / bias_act are differentiable):
 *   hfagp_upfirdn2d_bwd: dx [N][C][H][W] from dy [N][C][Ho][Wo]; H, W, up, down, padding: those of the FORWARD call
 *   hfagp_bias_act_bwd:  dx = dy * gain * (y < 0 ? alpha : 1) where |y| < clamp, else 0, from the forward OUTPUT y
 *                        (d bias = sum of dx over everything but the channel: the caller's reduction)               */
int hfagp_upfirdn2d_bwd(const float* dy, const float* f, float* dx, int32_t N, int32_t C, int32_t H, int32_t W,
                        int32_t fh, int32_t fw, int32_t up, int32_t down, int32_t px0, int32_t px1, int32_t py0,
                        int32_t py1, float gain, void* stream);
int hfagp_bias_act_bwd(const float* dy, const float* y, float* dx, int64_t n, int32_t act, float alpha, float gain,
                       float clamp, void* stream);

/* Blur(pad (1,1)) of the FIR [1,3,3,1] x [1,3,3,1] / 64 followed by stride-2 sampling, channels-last — the front of the 1x1
 * skip convolution of the RGB driver's ResBlock (/root/reference/code/networks/encoder3d.py:215, ConvLayer(..., 1,
 * downsample=True)): y[b][i][j][c] = sum_ab f[a] f[b] x[b][2i + a - 1][2j + b - 1][c], zero outside; H, W even, C % 4 == 0.
 * hfagp_blur_down_bwd is its adjoint (gx [B][H][W][C] from gy [B][H/2][W/2][C]).                                     */
int hfagp_blur_down_fwd(const float* x, float* y, int32_t B, int32_t H, int32_t W, int32_t C, void* stream);
int hfagp_blur_down_bwd(const float* gy, float* gx, int32_t B, int32_t H, int32_t W, int32_t C, void* stream);

/* layout helpers */
int hfagp_nchw_to_nhwc(const float* x, float* y, int32_t B, int32_t C, int32_t H, int32_t W, void* stream);
int hfagp_nhwc_to_nchw(const float* x, float* y, int32_t B, int32_t C, int32_t H, int32_t W, void* stream);

/* ------------------------------------------------------------------ the path's one exchange step, for non-PyTorch hosts
 * In-place all-reduce (sum, or mean when average != 0) of n floats at `buf` over the ranks of `comm`, an RCCL communicator
 * (ncclComm_t) the HOST created, enqueued on `stream`: the per-step reduction of the shared gradient buffer
 * [bases | delta | driver net | generator when tuned] that HFA-GP gets from DistributedDataParallel
 * (/root/reference/code/train_rgb.py:53-57,196-202; trainer_3dmm.py:29).  PyTorch hosts use torch.distributed instead
 * (hfa_gp_amd.trainer.FlatGrads / BucketedAllReduce).  The library does not link RCCL: ncclAllReduce is resolved from the
 * running process (or librccl.so on the loader path) at the first call; HFAGP_EUNSUPPORTED when it cannot be found.     */
int hfagp_allreduce_f32(void* buf, size_t n, void* comm, int32_t average, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* HFAGP_H_ */
