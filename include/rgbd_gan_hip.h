/*
 * rgbd_gan_hip.h -- C ABI of librgbdgan_hip.so (MI355X / gfx950 kernels for the RGBD-GAN hot path).
 *
 * The reference (nogu-atsu/RGBD-GAN) has no FFI layer: every FLOP of its training step runs inside
 * Chainer F.* / L.* calls (cuDNN / cuBLAS / CuPy).  Each entry point below replaces one group of
 * those call sites; the reference file:line is given per function.  The Python host in
 * rgbd_gan_amd/ binds these with ctypes (rgbd_gan_amd/_lib.py); INTEGRATION.md shows the stub a
 * maintainer of the reference would add.
 *
 * Conventions
 *   - plain pointers and sizes only; all pointers are DEVICE pointers owned by the caller unless a
 *     parameter is documented as host; nothing is allocated or freed inside
 *   - `stream` is a hipStream_t passed as void*; every launch goes to that stream, no call syncs
 *   - return value: 0 on success, negative on error (-1 bad argument, -2 HIP launch error);
 *     rgbd_last_error() returns a thread-local description
 *   - activations inside the conv stack are NHWC bf16; images at the API edge are NCHW fp32
 *     (the reference's layout); master weights are OIHW fp32 (the reference's layout)
 *   - thread-safe per stream; no global state
 */
#ifndef RGBD_GAN_HIP_H
#define RGBD_GAN_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RGBD_ABI_VERSION 20

int rgbd_abi_version(void);
const char* rgbd_last_error(void);

/* ------------------------------------------------------------------ 3D-consistency (warp) loss
 * Replaces common/loss_functions.py:63-146 (LossFuncRotate.__call__), :171-182 (warp / inv_warp)
 * and :185-228 (bilinear) -- about 100 CuPy kernels per call in the reference.
 *
 * img, img_rot : (b, 4, S, S) fp32 NCHW, channel 3 = depth.
 * coef         : (b, 24) fp32 = A(9, row-major) c(3) A'(9) c'(3) computed on the HOST in NumPy exactly as
 *                loss_functions.py:174,181 associate them (zp' = A (z p) - c ; zp'_rot = A' (z_rot p) + c').
 * flags        : bit0 occlusion_aware, bit1 use max_depth, bit2 use min_depth.
 * partials     : workspace, >= 6 * ceil(b*S*S/256) floats.
 * loss         : 1 float.
 * dbg_*        : optional (NULL to skip). dbg_zp (2,b,S*S,3) fp32; dbg_warped (2,b*S*S,4) fp32 (before the
 *                occlusion mask, like the reference's debug=True return); dbg_idx (2,b*S*S,4) int32 =
 *                masked u0, v0, v1 and the out-of-frame mask.  Index 0 = img->img_rot direction.
 * hinge_lambda, hinge_min: the depth-range hinge of updater.py:357-359 evaluated in the same pass:
 *                loss += hinge_lambda * mean over ALL 2b images' pixels of relu(hinge_min - depth)^2  (0 = off).
 * Index math is evaluated unfused (no FMA contraction) left to right so the integer outputs are
 * bit-exact against oracle/warp_loss.py:forward_np.
 */
#define RGBD_WARP_OCCLUSION 1
#define RGBD_WARP_MAX_DEPTH 2
#define RGBD_WARP_MIN_DEPTH 4
int rgbd_warp_loss_fwd(const float* img, const float* img_rot, const float* coef, int b, int S,
                       int flags, float lambda_geometric, float max_depth, float min_depth,
                       float hinge_lambda, float hinge_min,
                       float* partials, float* loss,
                       float* dbg_zp, float* dbg_warped, int32_t* dbg_idx, void* stream);

/* Backward of the above.  grad_loss: 1 float on the device (d objective / d loss), multiplied by grad_scale (host);
 * may be NULL (= 1).
 * grad_img, grad_img_rot: (b,4,S,S) fp32, 16-byte aligned; accumulate == 0: zeroed inside; accumulate != 0: the
 * gradients are ADDED to what the buffers hold (the caller has put the other terms of the image gradient there,
 * rgbd_image_grad_init).
 * workspace: rgbd_warp_loss_bwd_workspace(b, S) bytes of device scratch, 16-byte aligned, no initialisation needed.
 * The bilinear taps' scatter-add (the backward of the advanced-index gathers, loss_functions.py:221-226) is accumulated
 * there as 64-bit fixed point with INTEGER atomics and converted once: the result is bit-reproducible from run to run
 * (an fp32 atomic scatter is not).  S*S must be a multiple of 4. */
int64_t rgbd_warp_loss_bwd_workspace(int b, int S);
int rgbd_warp_loss_bwd(const float* img, const float* img_rot, const float* coef, int b, int S,
                       int flags, float lambda_geometric, float max_depth, float min_depth,
                       float hinge_lambda, float hinge_min,
                       const float* grad_loss, float grad_scale, float* grad_img, float* grad_img_rot, int accumulate,
                       void* workspace, void* stream);

/* The same loss for ANY channel count C >= 2 (the last channel is the depth) and either criterion: norm_l2 == 0
 * F.mean_absolute_error, != 0 F.mean_squared_error (loss_functions.py:137-145; `LossFuncRotate(norm="l2")` on the
 * 257-channel feature maps of updater.py:345-354).  img, img_rot, grad_*: (b,C,S,S) fp32; partials as above; no depth hinge,
 * no debug outputs.  The backward's scatter is order-independent too: per tap it accumulates sign(diff) * w (L1) or
 * diff * w (L2) as 64-bit integers in units of 2^-40 / 2^-32 in `workspace` (rgbd_warp_loss_nc_bwd_workspace(b, C, S)
 * bytes, 8-byte aligned, no initialisation needed) and multiplies by the channel's constant once; accumulate as above. */
int rgbd_warp_loss_nc_fwd(const float* img, const float* img_rot, const float* coef, int b, int C, int S, int flags,
                          int norm_l2, float lambda_geometric, float max_depth, float min_depth,
                          float* partials, float* loss, void* stream);
int rgbd_warp_loss_nc_bwd(const float* img, const float* img_rot, const float* coef, int b, int C, int S, int flags,
                          int norm_l2, float lambda_geometric, float max_depth, float min_depth,
                          const float* grad_loss, float grad_scale, float* grad_img, float* grad_img_rot, int accumulate,
                          void* workspace, void* stream);
int64_t rgbd_warp_loss_nc_bwd_workspace(int b, int C, int S);

/* ------------------------------------------------------------------ equalized-LR convolution engine
 * Replaces pggan.py:13-24 (EqualizedConv2d -> L.Convolution2D = cuDNN fprop/dgrad/wgrad) for the
 * 3x3 convolutions of net.py:124-125 (SynthesisBlock), :363,389-392 (Discriminator blocks), :615-616 (DCGAN).
 *
 * rgbd_pack_weights: master W (Cout,Cin,KH,KW) fp32 -> bf16 images with inv_c folded in:
 *   w_fprop [KH*KW][Cout][Cin]            (K-contiguous rows for the forward implicit GEMM)
 *   w_dgrad [KH*KW][Cin][Cout], taps flipped (so dgrad is the same kernel run on dY)
 * Either output may be NULL.
 */
int rgbd_pack_weights(const float* w, int cout, int cin, int kh, int kw, float scale,
                      void* w_fprop, void* w_dgrad, void* stream);

/* The same for every convolution of a network in ONE launch (after an optimizer update all layers need new bf16
 * images).  descs_device: n descriptors in DEVICE memory (they never change: master weights and packed images have
 * fixed addresses); descriptor i owns blocks [block_begin[i], block_begin[i+1]) of the total_blocks-block grid.
 */
typedef struct rgbd_pack_desc {
    const float* w;       /* (cout,cin,kh,kw) fp32 master; with `fold`: the reference-shaped master it is folded from */
    void* w_fprop;        /* [taps][cout][cin] bf16 or NULL */
    void* w_dgrad;        /* [taps][cin][cout] bf16, taps flipped, or NULL */
    int cout, cin, taps;
    float scale;
    int block_begin;
    int fold;             /* 0: none.  Else (ABI 20) w is a master parameter and (cout,cin,taps) its rgbd_fold_weight_f32
                           * rearrangement, made in the packing read: bits 0-1 = mode + 1 (mode 0: (Co,Ci,3,3,3) -> cin = 3 Cip,
                           * taps 9; 1: (Co,Ci,4,4) -> cin = 16 Cip, taps 1; 2: channel padding), bits 2-16 = Co, 17-31 = Ci
                           * (cout = Cop, Cip = cin / 3 | cin / 16 | cin); padding elements are zeros */
} rgbd_pack_desc;
int rgbd_pack_weights_multi(const rgbd_pack_desc* descs_device, int n, int total_blocks, void* stream);

/* Implicit-GEMM forward convolution on MFMA (bf16 in, fp32 accumulate), stride 1.
 *   x   : (B, Hin, Win, Cin) bf16 NHWC.  If `upsample` != 0 the convolution runs on the nearest-2x
 *         upsampled image (rescale.py:4-5 fused into the gather), so Hout = 2*Hin + 2*pad - KH + 1.
 *   wp  : packed weights [KH*KW][Cout][Cin] bf16 (rgbd_pack_weights).
 *   bias: (Cout) fp32 or NULL.  residual: (B,Hout,Wout,Cout) bf16 added after the bias, or NULL.
 *         lrelu_channels (multiple of 16): output channels [0, lrelu_channels) then get leaky-ReLU(slope) (0 = none;
 *         0 <= slope <= 1 required, the reference uses 0.2 everywhere), i.e.
 *         y = lrelu(conv + bias + residual) as in net.py:413-416.
 *   y   : (B, Hout, Wout, Cout) bf16 NHWC.
 *   y_pooled: NULL, or (B, Hout/2, Wout/2, Cout) bf16 that receives the 2x2 average of y (of the bf16 values stored to
 *         y) from the same epilogue -- the residual block's  downscale2x(leaky_relu(c1(h) + c_sc(x)))  (net.py:413-418,
 *         rescale.py:12-13) without re-reading y; needs KH = KW = 3, pad = 1, Hout and Wout multiples of 16.
 *   workspace: rgbd_conv2d_fprop_workspace(...) bytes of device scratch, or NULL.  Layers with few output tiles and
 *         a long reduction (the 4x4 .. 16x16 images) are split along K over several workgroups per tile; the fp32
 *         partial sums go through this scratch and a second kernel applies the epilogue.  NULL = never split.
 *   cus : compute-unit budget of THIS launch, 0 = all of the device.  The pipelined 3x3 kernel is persistent and holds a
 *         compute unit completely (130 KB of LDS, 512 registers per SIMD): while one launch covers the chip nothing of
 *         another stream starts.  A host that runs two streams side by side sizes the launches of the stream that is
 *         NOT its critical path for fewer compute units (grid = min(cus, the device's) workgroups, cut down further to
 *         what the number of rounds needs), and the other stream's kernels find free ones at once (RGBDUpdater: 224 of
 *         256 for the discriminator phases beside the generator's).  A per-launch argument: the library keeps no budget
 *         of its own (ABI <= 18 had a process-wide rgbd_set_cu_budget); a grid size is fixed when a launch is
 *         captured into a HIP graph.  Every conv entry point below takes the same argument in front of `stream`;
 *         launches that do not use the persistent kernel ignore it.
 * Requires Cin % 64 == 0 and Cout % 64 == 0.  dgrad = this function on dY with w_dgrad, pad' = KH-1-pad.
 */
int64_t rgbd_conv2d_fprop_workspace(int B, int Hin, int Win, int Cin, int Cout, int KH, int KW, int pad, int upsample);
int rgbd_conv2d_fprop_bf16(const void* x, const void* wp, const float* bias, const void* residual, void* y,
                           void* y_pooled, int B, int Hin, int Win, int Cin, int Cout, int KH, int KW, int pad,
                           int upsample, int lrelu_channels, float slope, void* workspace, int cus, void* stream);

/* Weight gradient: dw[co][ci][kh][kw] (+)= scale * sum_{b,h,w} dy[b,h,w,co] * x[b,h+kh-pad,w+kw-pad,ci]  (fp32).
 *   x  : (B,H,W,Cin) bf16, dy : (B,H,W,Cout) bf16 (same H,W: stride 1, pad = (K-1)/2), K in {1,3}.
 *   dw : (Cout,Cin,K,K) fp32 = the master-weight gradient (scale = inv_c of the equalized-LR conv).
 *   workspace: rgbd_conv2d_wgrad_workspace(...) bytes; holds one partial (K*K,Cout,Cin) slab per workgroup.
 *   upsample != 0: x is (B,H/2,W/2,Cin) and stands for its nearest-neighbour 2x upsampling (rescale.py:4-5, the
 *   generator's  c0(upscale2x(h))  at net.py:148-150) -- the halo gather reads source pixel (y/2, x/2), so the
 *   4x larger operand is never materialised.
 * Requires Cin % 64 == 0, Cout % 64 == 0, H and W powers of two >= 4.
 */
int64_t rgbd_conv2d_wgrad_workspace(int B, int H, int W, int Cin, int Cout, int K);
int rgbd_conv2d_wgrad_bf16(const void* x, const void* dy, void* workspace, float* dw,
                           int B, int H, int W, int Cin, int Cout, int K, float scale, int accumulate, int upsample,
                           void* stream);
/* The two halves of rgbd_conv2d_wgrad_bf16 for callers that collect the weight gradients of a whole backward pass
 * (Chainer runs them inside Convolution2DFunction.backward, pggan.py:13-24; order among them is free):
 *   rgbd_conv2d_wgrad_partial_bf16: the MFMA kernel only -- per-workgroup partial slabs into `workspace`
 *     (rgbd_conv2d_wgrad_workspace bytes = nsplit slabs of K*K*Cout*Cin floats);
 *   rgbd_wgrad_reduce_multi: slab sums + layout change + scale (+ accumulate) for n such workspaces in one launch per
 *     32 descriptors.  `descs` is HOST memory (copied into the kernel arguments). */
typedef struct rgbd_wgrad_reduce_desc {
    const float* workspace;   /* nsplit slabs of [taps][cout][cin] floats */
    float* dw;                /* (cout, cin, K, K) master-weight gradient */
    int32_t nsplit, taps, cout, cin;
    float scale;
    int32_t accumulate;
} rgbd_wgrad_reduce_desc;
int rgbd_conv2d_wgrad_partial_bf16(const void* x, const void* dy, void* workspace, int B, int H, int W, int Cin, int Cout,
                                   int K, int upsample, void* stream);
int rgbd_wgrad_reduce_multi(const rgbd_wgrad_reduce_desc* descs, int n, void* stream);
/* The partial (slab) kernel for up to 24 weight gradients in ONE launch: a launch per layer costs one slab per CU (38 MB
 * of write + read at 64x64x9 fp32) whatever the layer's size; one launch for all weight gradients of a backward pass
 * deals the CUs out over the layers in proportion to their work and costs 38 MB in total.  3x3 pad-1 convs on
 * power-of-two images; one call takes either problems of at least 8x16 pixels or smaller ones (4x4 .. 8x8: each planned
 * on its own, one launch instead of a dozen latency-bound ones), not a mix.  rgbd_conv2d_wgrad_multi_plan fills `nsplit` of every problem (HOST
 * arrays; total_workgroups <= 0: one per CU; a host running a second stream beside the launch passes fewer); the caller then provides workspace = nsplit * 9 * Cout * Cin floats per
 * problem and finishes with rgbd_wgrad_reduce_multi. */
typedef struct rgbd_wgrad_problem {
    const void* x;        /* (B,H,W,Cin) bf16, or (B,H/2,W/2,Cin) when upsample != 0 */
    const void* dy;       /* (B,H,W,Cout) bf16 */
    void* workspace;      /* nsplit * 9 * Cout * Cin fp32 */
    int32_t B, H, W, Cin, Cout, K, upsample, nsplit;
} rgbd_wgrad_problem;
int rgbd_conv2d_wgrad_multi_plan(rgbd_wgrad_problem* probs, int n, int total_workgroups);
int rgbd_conv2d_wgrad_partial_multi_bf16(const rgbd_wgrad_problem* probs, int n, void* stream);

/* ------------------------------------------------------------------ AdaIN (instance norm + style affine)
 * Replaces normalization/adain.py:54-73 (reshape + F.batch_normalization + broadcast mul/add) and its backward.
 *   x (B,HW,C) bf16 NHWC; scale, shift: B rows of C fp32 values, rows `ld` floats apart (ld = C for plain (B,C)
 *   arrays; ld = 2C with shift = scale + C when one fused linear produced [scale | shift], net.py:96-101);
 *   eps 1e-5; biased variance.
 *   sums: workspace of rgbd_adain_workspace(B,HW,C) floats = (ceil(HW/1024), B, C, 2) fp32, no initialisation
 *   needed: every 1024-pixel strip stores its partial sums with plain stores and the second launch adds the strips in
 *   index order, so the statistics are bit-reproducible (no atomics).  mean, rstd: (B,C) fp32 outputs (saved for
 *   backward).  Two launches: strip reduction, normalise + affine.
 */
int64_t rgbd_adain_workspace(int B, int HW, int C);   /* floats */
/* y_q / y_s (here and below; NULL = none): an MXFP8 copy of the bf16 tensor the call stores -- exactly rgbd_quantize_mxfp8 of
 * it, (.., C) bytes + (.., C/32) scale bytes -- for the convolution that reads it next (conv_dtype mxfp8). */
/* c_live (a multiple of 8 in (0, C]; C for an ordinary tensor): channels [c_live, C) of x are zero padding (a 32-channel
 * block of the DeepVoxels generator on the engine's 64-channel granularity, deepvoxels_generator.py:112-168): they have no
 * scale / shift entries -- a fused [scale | shift] window is 2 c_live floats wide, shift = scale + c_live -- and stay zero. */
int rgbd_adain_fwd(const void* x, const float* scale, const float* shift, void* y,
                   float* sums, float* mean, float* rstd, int B, int HW, int C, int c_live, int ld, float eps, void* y_q,
                   void* y_s, void* stream);
/* dy (B,HW,C) bf16 -> dx bf16, dscale/dshift fp32 rows `ld` apart like scale (overwritten).
 * sums: workspace of rgbd_adain_workspace(B,HW,C) floats, as above.
 * lrelu_slope > 0: x is the output of the leaky ReLU feeding this AdaIN (net.py:150-153: conv -> bias -> lrelu -> style)
 *   and dx additionally carries that activation's gradient, dx *= (x > 0 ? 1 : lrelu_slope); bias_grad (C) fp32 or NULL
 *   then accumulates sum_{b,p} dx (the gradient of the L.Bias in front of the activation). */
int rgbd_adain_bwd(const void* x, const void* dy, const float* scale, const float* mean, const float* rstd,
                   void* dx, float* dscale, float* dshift, float* sums, int B, int HW, int C, int c_live, int ld,
                   float lrelu_slope, float* bias_grad, void* dx_q, void* dx_s, void* stream);

/* ------------------------------------------------------------------ small fused elementwise / 1x1 kernels (HBM-bound)
 * rgbd_lrelu_bwd: dz = dy * (y > 0 ? 1 : slope) on channels [0, act_channels) of (M,C) bf16 tensors, pass-through on
 *   the rest (backward of F.leaky_relu, net.py:152,159,410,416, evaluated from the activation's OUTPUT).
 * rgbd_colsum_bf16: out[c] = sum_m w(m) x[m][c] (fp32): bias gradients of the convs; w(m) = row_scale[m / rows_per_sample]
 *   (per-sample weights) or 1 when row_scale is NULL.
 */
int rgbd_lrelu_bwd(const void* dy, const void* y, void* dz, int64_t M, int C, int act_channels, float slope,
                   float* bias_grad, const float* row_scale, int64_t rows_per_sample, void* stream);
int rgbd_colsum_bf16(const void* x, float* out, int64_t M, int C, int accumulate, const float* row_scale,
                     int64_t rows_per_sample, void* stream);
/* out = a + s[sample] * x on (B, elems_per_sample) bf16 tensors: folds the adversarial term of the discriminator loss
 * into the operand of the R1 double-backward weight gradients (see rgbd_gan_amd/functional.py:adversarial_injection). */
int rgbd_axpy_rows_bf16(const void* a, const void* x, const float* s, void* out, int64_t B, int64_t elems_per_sample,
                        void* stream);

/* 2x2 average pooling of the discriminator blocks (rescale.py:12-13) fused with the leaky-ReLU that precedes it:
 *   rgbd_unpool2_lrelu_bwd: dz[b,h,w,c] = 0.25 * dp[b,h/2,w/2,c] * lrelu'(y[b,h,w,c])   (y NULL: no mask); bias_grad as above;
 *                           bias_grad2 (or NULL) receives the same column sums: in a residual block (net.py:413-416) the
 *                           shortcut conv's bias has the gradient of the main conv's bias, dz being the gradient of both
 *   rgbd_pool2_masked     : out[b,hp,wp,c] = 0.25 * sum_{2x2} x * lrelu'(y)              (y NULL: plain average pool)
 * The two are adjoint (same mask), which closes the pair under differentiation (R1 double backward).
 */
int rgbd_unpool2_lrelu_bwd(const void* dp, const void* y, void* dz, int B, int H, int W, int C, float slope,
                           float* bias_grad, float* bias_grad2, const float* row_scale, void* dz_q, void* dz_s, void* stream);
int rgbd_pool2_masked(const void* x, const void* y, void* out, int B, int H, int W, int C, float slope, void* stream);
/* 2x2 SUMS of a (B,H,W,C) bf16 tensor -> (B,H/2,W/2,C): the adjoint of the nearest-2x upsample (rescale.py:4-5) where the
 * input-gradient kernel does not produce them in its epilogue (8x8 -> 4x4 layers). */
int rgbd_pool2_sum_bf16(const void* x, void* out, int B, int H, int W, int C, void* stream);

/* 1x1 convolutions between NCHW fp32 image planes (KP = 3 or 4 channels) and NHWC bf16 features (C channels):
 *   rgbd_from_planes: y[b,p,co] = act(wscale * sum_k w[co][k] x[b,k,p] + bias[co])   -- Discriminator.ins, net.py:449-455
 *   rgbd_to_planes  : out[b,k,p] = wscale * sum_c w[k][c] h[b,p,c] + bias[k]           -- StyleGenerator.outs, net.py:186-191
 *                     (also the input gradient of from_planes with w transposed)
 *   rgbd_planes_outer: o[k][c] = sum_{b,p} planes[b,k,p] * t[b,p,c]; tsum[c] = sum t   -- their weight / bias gradients;
 *                     psum[k] += sum_{b,p} planes[b,k,p] (ADDED: it is to_rgb's bias gradient buffer; o / tsum are overwritten)
 * w is fp32, row-major as written; bias / tsum / psum may be NULL.
 */
int rgbd_from_planes(const float* x, const float* w, const float* bias, void* y, int B, int HW, int KP, int C,
                     float wscale, int act, float slope, void* stream);
int rgbd_to_planes(const void* h, const float* w, const float* bias, float* out, int B, int HW, int KP, int C,
                   float wscale, void* stream);
int rgbd_planes_outer(const void* t, const float* planes, float* o, float* tsum, float* psum, int B, int HW, int KP,
                      int C, void* stream);

/* ------------------------------------------------------------------ small equalized-LR linear layers (M <= 64 rows, fp32)
 * Replace pggan.py:39-50 (EqualizedLinear = scale + L.Linear) + F.leaky_relu for the mapping MLP (net.py:58-62), the
 * pose-conditioned style (net.py:220-224) and the StyleBlock affines (net.py:96-101): launch-latency bound.
 *   fwd: y (M,N) = act(c * x (M,K) W(N,K)^T + bias);  act = leaky ReLU(slope) or identity.
 *   bwd: dz = dy * lrelu'(y) (when act);  dx (M,K) (+)= c * dz W;  dw (N,K) += c * dz^T x;  db (N) += sum_m dz.
 *        dx / dw / db may be NULL to skip; dw and db ACCUMULATE (they are the optimizer's flat gradient views).
 */
/* workspace: rgbd_linear_fwd_workspace(M,K,N) floats (0 = none) or NULL: for K >= 1024 (the discriminator's dense tail,
 * K = 4096) the forward splits K over blocks and a finishing launch applies scale, bias and activation. */
int64_t rgbd_linear_fwd_workspace(int M, int K, int N);
int rgbd_linear_fwd(const float* x, const float* w, const float* bias, float* y, int M, int K, int N, float c, int act,
                    float slope, float* workspace, void* stream);
int rgbd_linear_bwd(const float* dy, const float* y, const float* x, const float* w, float* dx, float* dw, float* db,
                    int M, int K, int N, float c, int act, float slope, int accumulate_dx, void* stream);
/* A CHAIN of L <= 8 such layers with K = N = C (256 or 512), leaky ReLU behind every one -- the mapping network, net.py:22-62
 * (`for l in self.l: h = l(h)`) -- as ONE launch per pass: a workgroup per 16 rows keeps the activations in LDS and streams the
 * weights of all layers.  w_host / b_host / dw_host / db_host: HOST arrays of L device pointers (copied into the launch).
 *   fwd: acts (L,M,C) receives every layer's output; acts[L-1] is the chain's result.
 *   bwd: dy (M,C) = gradient of acts[L-1]; writes dz (L,M,C) (scratch: the gradients of the pre-activations) and dx (M,C);
 *        when dw_host != NULL a second launch ADDS c * dz[l]^T in_l to dw_host[l] (C,C) and sum_m dz[l] to db_host[l] (C)
 *        for all layers (in_0 = x, in_l = acts[l-1]); a NULL entry skips that layer. */
int rgbd_mlp_fwd(const float* x, const float* const* w_host, const float* const* b_host, int L, int M, int C, float c,
                 float slope, float* acts, void* stream);
int rgbd_mlp_bwd(const float* dy, const float* x, const float* acts, const float* const* w_host, float* const* dw_host,
                 float* const* db_host, int L, int M, int C, float c, float slope, float* dz, float* dx, void* stream);
/* y = (c * x W^T) * lrelu'(mask_y): the derivative of rgbd_linear_bwd's dx w.r.t. its dy -- what the R1 double backward
 * (updater.py:414-422, chainer.grad(..., enable_double_backprop=True)) sends back through the discriminator's dense tail
 * (net.py:372-377).  mask_y (M,N): the activation OUTPUT whose slope mask applies. */
int rgbd_linear_fwd_masked(const float* x, const float* w, const float* mask_y, float* y, int M, int K, int N, float c,
                           float slope, float* workspace, void* stream);

/* ------------------------------------------------------------------ small fused ops of the training step (step_ops.hip)
 * Each replaces a run of elementwise / reduction launches of the reference's Chainer graph with one launch.
 * rgbd_real_batch_u8: SerialIterator + TransformDataset(x/127.5 - 1) (train_rgbd.py:308-310) + downsize_real
 *   (common/utils/pggan.py:6-50) from the uint8 data set resident in HBM: data (N,C,H,W) u8, idx (B) int64 ->
 *   out (B,C,S,S) fp32 = s x s block means (s = H/S) of data[idx]/127.5 - 1; fade != 0 (odd progressive stage):
 *   (1-alpha) * [2s x 2s block mean, nearest-upsampled] + alpha * [s x s block mean], alpha read from alpha_device
 *   (device float, so a captured graph follows the schedule) or, when that is NULL, the host value.
 * rgbd_zero_multi_f32: clear up to 8 fp32 buffers (16-byte aligned) in one launch (gradient buffers at step start).
 * rgbd_hidden_normalize: make_hidden's normalisation (net.py:333-343): out rows = z / sqrt(sum_c z^2 / ch + 1e-8) for
 *   z (M,C); written `copies` times, copy k at rows [k*M, (k+1)*M) (updater.py:300 repeats the latents per view pair).
 * rgbd_r1_penalty_fwd: updater.py:416-418 + loss_functions.py:7-8 on g (B,n) fp32:
 *   loss = coef * mean_b (sqrt(sum g_b^2))^2; workspace >= 16*B floats.  Its gradient w.r.t. g is
 *   rgbd_scale_by_scalar_f32(g, grad_loss_device, 2*coef/B): out = (scalar_device[0] * k) * x (scalar NULL = 1).
 * rgbd_image_grad_init: out (B,KP_out,HW) fp32: channels k < KP_in = ratio[b] * gx[b,k,:] (ratio NULL = 1), others 0 --
 *   the generator's output gradient before rgbd_warp_loss_bwd(accumulate=1) adds the 3-D consistency terms.
 * rgbd_const_input_{fwd,bwd}: SynthesisBlock 0 (net.py:130-153): out (B,HW,C) bf16 = lrelu(w[c,p] + bias[c]) for every
 *   sample; bwd ADDS sum_b dh * lrelu' to dw (C,HW) and its sum over p to db (C) (either may be NULL).
 * rgbd_nhwc_to_rows_f32 / rgbd_rows_to_nhwc_bf16: (B,HW,C) bf16 <-> (B, C*HW) fp32 rows in (c,p) order, the layout the
 *   4x4-valid conv of the discriminator's base block (net.py:363-365) consumes as a linear layer; mutually adjoint.
 */
int rgbd_real_batch_u8(const uint8_t* data, const int64_t* idx, float* out, int B, int C, int H, int W, int S, int fade,
                       const float* alpha_device, float alpha, void* stream);
int rgbd_zero_multi_f32(float* const* ptrs /* host array */, const int64_t* counts /* host array */, int n, void* stream);
int rgbd_hidden_normalize(const float* z, float* out, int M, int C, float ch, int copies, void* stream);
/* The same with the draw inside (net.py:333-343 `xp.random.normal` + the normalisation): out (copies*M, C) fp32 from
 * Philox4x32-10 + Box-Muller, keyed by state[0..1] (seed), counter (state[2] = launch number, row, column quad).  state: four
 * uint32 in DEVICE memory, {seed lo, seed hi, launch number, 0}; the kernel's last block bumps the launch number, so a launch
 * replayed from a captured graph draws new values every time.  C a multiple of 4, at most 1024. */
int rgbd_hidden_draw(uint32_t* state, float* out, int M, int C, float ch, int copies, void* stream);
int rgbd_r1_penalty_fwd(const float* g, int B, int64_t n, float coef, float* workspace, float* loss, void* stream);
int rgbd_scale_by_scalar_f32(const float* x, const float* scalar_device, float k, float* out, int64_t n, void* stream);
/* out[r, :] = a[r, :] + s[r] * x[r, :] over (rows, row_len) fp32, row_len % 4 == 0, 16-byte aligned; s NULL = 1; out may
 * alias a.  The sum of two gradient buffers (rows = 1) and the per-sample operand update of the adversarial injection
 * (updater.py:405-422 folded into the R1 double backward) at the image planes. */
int rgbd_axpy_rows_f32(const float* a, const float* x, const float* s, float* out, int64_t rows, int64_t row_len, void* stream);
int rgbd_image_grad_init(const float* gx, const float* ratio, float* out, int B, int KP_in, int KP_out, int HW,
                         void* stream);
int rgbd_const_input_fwd(const float* w, const float* bias, void* out, int B, int HW, int C, float slope, void* stream);
int rgbd_const_input_bwd(const void* dh, const float* w, const float* bias, float* dw, float* db, int B, int HW, int C,
                         float slope, void* stream);
/* rgbd_l2norm_{fwd,bwd}: DCGANBlock's F.normalize over channels (net.py:621-648) on (npix, C) bf16 rows, C in
 *   {128, 256, 512}: y = x / (||x||_2 + eps) with the norm in fp32; dx = dy / d - x (x . dy) / (n d^2), d = n + eps.
 * rgbd_blur3x3_bf16: rescale.py:20-25 (depthwise [1 2 1]x[1 2 1]/16, zero padding) on NHWC bf16, H x W = the image the
 *   blur acts on.  mode 0: y = blur(x); mode 1: y = blur(upscale2x(x)), x (B,H/2,W/2,C) (net.py:140-141);
 *   mode 2: y (B,H/2,W/2,C) = 2x2 sums of blur(x), the adjoint of mode 1.  Mode 0 is its own adjoint. */
/* Progressive fade-in (odd stages, net.py:283-290,490-497): alpha is read from alpha_device (a device float: a captured
 *   graph follows the schedule) or, when that is NULL, the host value.  H x W = the HIGH resolution.
 * rgbd_fade_planes_fwd: out = (1-a) * upscale2x(lo) + a * hi on (planes,H,W) fp32 NCHW planes, lo (planes,H/2,W/2).
 * rgbd_fade_planes_bwd: dhi = a * dout, dlo = (1-a) * 2x2 sums of dout (either may be NULL).
 * rgbd_lerp_bf16: mode 0: out = (1-a) p + a q; mode 1: out = (1-a) p, out2 = a p (its adjoint split); n bf16 elements.
 * rgbd_pool2_planes: adjoint 0: out (planes,H/2,W/2) = 2x2 averages of x (planes,H,W) (downscale2x of the image);
 *   adjoint != 0: out (planes,H,W) = 0.25 * x[y/2,x/2] with x (planes,H/2,W/2). */
int rgbd_fade_planes_fwd(const float* lo, const float* hi, float* out, int64_t planes, int H, int W,
                         const float* alpha_device, float alpha, void* stream);
int rgbd_fade_planes_bwd(const float* dout, float* dlo, float* dhi, int64_t planes, int H, int W,
                         const float* alpha_device, float alpha, void* stream);
int rgbd_lerp_bf16(const void* p, const void* q, void* out, void* out2, int64_t n, int mode, const float* alpha_device,
                   float alpha, void* stream);
int rgbd_pool2_planes(const float* x, float* out, int64_t planes, int H, int W, int adjoint, void* stream);
int rgbd_l2norm_fwd(const void* x, void* y, int64_t npix, int C, float eps, void* stream);
int rgbd_l2norm_bwd(const void* x, const void* dy, void* dx, int64_t npix, int C, float eps, void* stream);
int rgbd_blur3x3_bf16(const void* x, void* y, int B, int H, int W, int C, int mode, void* stream);
int rgbd_nhwc_to_rows_f32(const void* h, float* rows, int B, int HW, int C, void* stream);
int rgbd_rows_to_nhwc_bf16(const float* rows, void* h, int B, int HW, int C, void* stream);

/* ------------------------------------------------------------------ DeepVoxels frustum path (config 4)
 * rgbd_proj_idcs: deepvoxel/projection.py:48-105 (compute_proj_idcs) for a whole batch of cameras at once.
 *   cam2world (B,16) fp32 row-major 4x4.  Frustum of W x H x D elements, grid of G^3 voxels.
 *   Out: idx (B,N) int32 and coords (B,3,N) fp32, N = W*H*D, compacted IN ORDER (first counts[b] entries valid),
 *   counts (B) int32.  workspace: B * ceil(N/256) int32.  Index math unfused fp32, bit-exact vs oracle/deepvoxels.py.
 * rgbd_trilinear_{fwd,bwd}: deepvoxel.py:388-428 (interpolate_trilinear); grid (B,F,G,G,G) fp32 indexed [x][y][z]
 *   with x = coords[2], y = coords[1], z = coords[0]; out (B,F,N) (zero-filled inside; reshape to (B,F,D,H,W)).
 * rgbd_occlusion_accum_{fwd,bwd}: deepvoxel.py:574-587 (AccumulativeOcclusionNet, occnet_nf = 4) + the compositing of
 *   DeepVoxels.forward :886-889 + depth rescale :903-904.  vol (B,F,D,HW) fp32; W1 (4,F+1) [column 0 = depth
 *   coordinate], b1 (4), W2 (4), b2 (1) fp32 master weights (equalized-LR scales applied inside).
 *   fwd out: s, w (B,D,HW) (saved for bwd), feat (B,F,HW), depth (B,HW).
 *   bwd out: dvol (B,F,D,HW); dparams = [dW1 | db1 | dW2 | db2] (zeroed inside); dw_ws, ds_ws: (B,D,HW) workspaces.
 */
int rgbd_proj_idcs(const float* cam2world, int B, int W, int H, int D, int G, float voxel_size, float near_plane,
                   float fx, float fy, float cx, float cy, int32_t* idx, float* coords, int32_t* counts,
                   int32_t* workspace, void* stream);
/* Layout folds that put the DeepVoxels networks on the 2-D conv engine (deepvoxels_generator.py:112-222: 3x3x3 convs as
 * 3x3 convs over depth slices with the three depth taps folded into channels, 4x4 stride-2 convs as 1x1 convs over 16 folded
 * taps, channel counts padded to the engine's multiples).  One launch each, adjoint != 0 = the backward gather:
 *   rgbd_fold_depth_taps_bf16: x (B,D0,H,W,C) -> y (B*D,H,W,3C), D = 2 D0 when upsample_depth (nearest depth repeat folded in);
 *     adjoint: x = dy (B*D,H,W,3C) -> y = dx (B,D0,H,W,C).
 *   rgbd_fold_4x4s2_bf16: x (B,H,W,C) -> y (B,H/2,W/2,16C), channel (ky*4+kx)*C+c = x_pad1[2i+ky, 2j+kx, c]; adjoint likewise.
 *   rgbd_pad_last: rows x C0 -> rows x C1 elements of 2 or 4 bytes; C1 > C0 zero-fills the tail, C1 < C0 slices. */
int rgbd_fold_depth_taps_bf16(const void* x, void* y, int B, int D0, int H, int W, int C, int upsample_depth, int adjoint,
                              void* stream);
int rgbd_fold_4x4s2_bf16(const void* x, void* y, int B, int H, int W, int C, int adjoint, void* stream);
int rgbd_pad_last(const void* x, void* y, int64_t rows, int C0, int C1, int elem_bytes, void* stream);
/* Master parameter (reference shape, fp32) -> the (Cop,Cfp,KH,KW) weight the conv engine packs; adjoint != 0: gradient of the
 * folded weight -> gradient of the master (adjoint == 2: added to dst).  mode 0: (Co,Ci,3,3,3) -> (Cop,3*Cip,3,3) [deepvoxels_generator.py:112-168,
 * pggan.py:27-38]; mode 1: (Co,Ci,4,4) -> (Cop,16*Cip,1,1) [:191-205]; mode 2: (Co,Ci,K,K) -> (Cop,Cip,K,K) (zero padding). */
int rgbd_fold_weight_f32(const float* src, float* dst, int mode, int Co, int Ci, int K, int Cop, int Cip, int adjoint,
                         void* stream);
/* The same for n layers in ONE launch (a network's folds on a rebuild of its weight images; the adjoints behind a backward
 * pass): descs is a HOST array, copied by value into the kernel arguments. */
typedef struct rgbd_fold_desc {
    const float* src;
    float* dst;
    int32_t mode, Co, Ci, K, Cop, Cip, adjoint, reserved;
} rgbd_fold_desc;
int rgbd_fold_weight_multi_f32(const rgbd_fold_desc* descs, int n, void* stream);
int rgbd_trilinear_fwd(const float* grid, const int32_t* idx, const float* coords, const int32_t* counts, float* out,
                       int B, int F, int G, int N, void* stream);
int rgbd_trilinear_bwd(const float* dout, const int32_t* idx, const float* coords, const int32_t* counts, float* dgrid,
                       float* workspace /* B*G^3*F floats: feature-major scatter target */, int B, int F, int G, int N,
                       void* stream);
/* The feature-minor backward without the compacted list: the voxel coordinates of every frustum element are recomputed from
 * the camera (the arithmetic of rgbd_proj_idcs, bit for bit), a workgroup takes a 16 x 8 x 2 brick of the frustum, sorts the
 * brick's (voxel, sample, corner) contributions by voxel and issues ONE line atomic per distinct voxel of the brick.
 * dout (B,F,W*H*D) fp32, cam2world (B,4,4) device fp32, dgrid_fm (B,G,G,G,F) zero-filled inside.
 * rgbd_trilinear_bwd_frustum_supported: W % 16 == 0, H % 8 == 0, F <= 32, G^3 <= 2^20. */
int rgbd_trilinear_bwd_frustum_supported(int W, int H, int D, int G, int F);
/* ... and the forward the same way: out (B,F,W*H*D) written completely (zeros outside the grid: no fill in front of it), values bit-identical
 * to rgbd_trilinear_fwd_fm's on the elements the list holds.  F <= 32. */
int rgbd_trilinear_fwd_frustum(const float* grid_fm, const float* cam2world, int B, int F, int W, int H, int D, int G,
                               float voxel_size, float near_plane, float fx, float fy, float cx, float cy, float* out, void* stream);
int rgbd_trilinear_bwd_frustum(const float* dout, const float* cam2world, int B, int F, int W, int H, int D, int G,
                               float voxel_size, float near_plane, float fx, float fy, float cx, float cy, float* dgrid_fm,
                               void* stream);
/* The same resampling on a FEATURE-MINOR grid (B,G,G,G,F) -- what the voxel generator's NHWC conv stack produces, no
 * transposition on either side: 8 line reads per sample forward, and the backward's scatter target IS the gradient. */
int rgbd_trilinear_fwd_fm(const float* grid_fm, const int32_t* idx, const float* coords, const int32_t* counts, float* out,
                          int B, int F, int G, int N, void* stream);
int rgbd_trilinear_bwd_fm(const float* dout, const int32_t* idx, const float* coords, const int32_t* counts, float* dgrid_fm,
                          int B, int F, int G, int N, void* stream);
int rgbd_occlusion_accum_fwd(const float* vol, const float* W1, const float* b1, const float* W2, const float* b2,
                             float threshold, float voxel_size, float near_plane, float* s, float* w, float* feat,
                             float* depth, int B, int F, int D, int HW, void* stream);
int rgbd_occlusion_accum_bwd(const float* vol, const float* W1, const float* b1, const float* W2, const float* s,
                             const float* w, const float* dfeat, const float* ddepth, float voxel_size, float* dw_ws,
                             float* ds_ws, float* dvol, float* dparams, int B, int F, int D, int HW, void* stream);

/* ------------------------------------------------------------------ small fused pointwise ops
 * rgbd_conv2d_dgrad_bf16: input gradient of a stride-1 convolution, dx (B,H+2*(K-1-pad)-K+1,...,Cin) from
 *   dy (B,H,W,Cout) and the dgrad image of rgbd_pack_weights ([K*K][Cin][Cout], taps flipped): the same
 *   implicit-GEMM kernels as fprop (chainer's Convolution2DFunction backward, pggan.py:13-24); workspace =
 *   rgbd_conv2d_fprop_workspace(B, H, W, Cout, Cin, K, K, K-1-pad, 0) bytes or NULL.
 *   residual: NULL or a tensor of dx's shape added in the epilogue -- the input gradient of a residual block's entry,
 *   dgrad_c0(dz0) + dgrad_c_sc(dz1) (net.py:408-416: both convs read the block input), without an add pass.
 *   sum_pool2 != 0: dx is (B,H/2,W/2,Cin), the 2x2 sums of the input gradient -- the adjoint of the nearest-2x
 *   upsampling in front of the generator's c0 (net.py:148-150, rescale.py:4-5) taken in the conv epilogue; needs
 *   K = 3, pad = 1 and H, W multiples of 16.
 * rgbd_conv3x3_actgrad_bf16: y = (conv3x3_pad1(x, wp) + residual) * lrelu'(act_y), wp a packed image in the kernel's
 *   [9][Cout][Cin] order (the fprop image, or the dgrad image with Cin / Cout exchanged), act_y a GIVEN leaky-ReLU output
 *   of y's shape: the input gradient of a convolution and the activation gradient of the layer in front of it in ONE
 *   epilogue (net.py:408-416, h = lrelu(c0 x) -> c1: dz0 = dgrad_c1(dz1) * lrelu'(h)), instead of a conv launch and a
 *   3-tensor elementwise pass.  colsum (NULL or Cout floats, ACCUMULATED with fp32 atomics): the weighted column sums
 *   sum_b row_scale[b] sum_pixels y -- the bias gradient of that layer (row_scale NULL = 1; per-sample seeds of
 *   updater.py:405-422 otherwise).  Runs on the pipelined 3x3 kernel: H, W multiples of 16, Cin, Cout multiples of 64
 *   (rgbd_conv3x3_actgrad_supported tells, as a pure function of the shape).  y2 / row_scale2 (both or neither): a
 *   second output y2 = y + row_scale2[b] * act_y, i.e. rgbd_axpy_rows_bf16(y, act_y, row_scale2) -- in the R1 double
 *   backward the weight-gradient operand dd h0 + s_b h0 of the convolution behind (updater.py:405-422 folded in) -- from
 *   the tile the epilogue holds, instead of a 3-tensor pass.
 * rgbd_conv2d_fprop_stats_bf16: the generator's conv -> bias -> leaky ReLU (net.py:148-153,157-160; upsample != 0: nearest
 *   2x in front, rescale.py:4-5) as rgbd_conv2d_fprop_bf16 computes it, 3x3 pad 1 on output images that are multiples of
 *   16x16, PLUS the instance-norm statistics of the AdaIN that follows (adain.py:62-63): stats (B,Cout,2) int64, ZEROED
 *   by the caller, receives (sum y, sum y^2) over each image's pixels of the bf16 values stored, in units of 2^-32, by
 *   64-bit integer atomics -- order-independent, so the statistics are bit-reproducible.  rgbd_adain_apply_fixed is
 *   rgbd_adain_fwd without its reduction pass, reading those (same outputs: y, mean, rstd).
 * rgbd_pixelnorm_{fwd,bwd}: pggan.py:7-10 (feature_vector_normalization) on (M,C) fp32 rows:
 *   y = x * rsqrt(mean_c x^2 + eps);  dx = r * (dy - y * mean_c(dy * y)).
 * rgbd_depth_head_{fwd,bwd}: net.py:296 on (B,4,HW) fp32 planes: channels 0-2 pass through,
 *   y3 = 1 / (softplus(x3) + 1e-4);  dx3 = -dy3 * y3^2 * sigmoid(x3).
 * rgbd_ema_update: copy_param.py:17-40 (soft_copy_param) over a flat parameter buffer: dst = (1-tau) dst + tau src.
 */
int rgbd_conv2d_dgrad_bf16(const void* dy, const void* wp_dgrad, const void* residual, void* dx, int B, int H, int W,
                           int Cin, int Cout, int K, int pad, int sum_pool2, void* workspace, int cus, void* stream);
int rgbd_conv3x3_actgrad_supported(int B, int H, int W, int Cin, int Cout);
int rgbd_conv3x3_actgrad_bf16(const void* x, const void* wp, const void* residual, const void* act_y, float slope,
                              float* colsum, const float* row_scale, void* y, void* y2, const float* row_scale2, int B,
                              int H, int W, int Cin, int Cout, int cus, void* stream);
int rgbd_conv2d_fprop_stats_bf16(const void* x, const void* wp, const float* bias, void* y, int64_t* stats, int B, int Hin,
                                 int Win, int Cin, int Cout, int upsample, int lrelu_channels, float slope, int cus, void* stream);
int rgbd_adain_apply_fixed(const void* x, const float* scale, const float* shift, void* y, const int64_t* stats, float* mean,
                           float* rstd, int B, int HW, int C, int ld, float eps, void* y_q, void* y_s, void* stream);

/* ------------------------------------------------------------------ MXFP8 convolutions (BASELINE configuration 5)
 * The 3x3 convolutions of the 256x256 networks -- the blocks the reference keeps commented out at net.py:181-183,192-194,
 * again pggan.py:13-24 / cuDNN in the reference -- on the block-scaled fp8 matrix instruction of gfx950
 * (v_mfma_scale_f32_16x16x128_f8f6f4, fp32 accumulate): fprop and dgrad; weight gradients stay on the bf16 kernels.
 *
 * Operand format (OCP microscaling MXFP8, element type E4M3): along the REDUCTION index -- the channels of an NHWC
 * activation tensor, Cin of the fprop weight image, Cout of the dgrad image -- every 32 consecutive elements share one E8M0
 * scale byte s: value = e4m3(q) * 2^(s - 127), s = max(E - 8 + (m > 1.75), 0) with E / m the biased fp32 exponent / significand
 * of the block's largest magnitude (which therefore lands in (224, 448] and never saturates), q = e4m3_rne(clamp(x * 2^(127 - s),
 * -448, 448)).  Stateless (no amax history), bit-reproducible, restated in
 * oracle/mxfp8.py.
 *
 * rgbd_quantize_mxfp8: x (rows, C) bf16 -> q (rows, C) bytes + scales (rows, C/32) bytes; C % 128 == 0.
 * rgbd_pack_weights_mxfp8_multi: for every descriptor, master W (cout,cin,3,3) fp32 times `scale` (inv_c) ->
 *   wf_q [9][cout][cin] + wf_s [9][cout][cin/32]                (fprop image, blocks along cin), and
 *   wd_q [9][cin][cout] + wd_s [9][cin][cout/32], taps flipped  (dgrad image, blocks along cout);
 *   either pair may be NULL; cout, cin multiples of 32.  Descriptors live in DEVICE memory, descriptor i owns blocks
 *   [block_begin[i], block_begin[i+1]) of the grid (as rgbd_pack_weights_multi).
 * rgbd_conv2d_fprop_mxfp8 / _dgrad_mxfp8 / rgbd_conv3x3_actgrad_mxfp8 / rgbd_conv2d_fprop_stats_mxfp8: the launches of
 *   rgbd_conv2d_fprop_bf16 (3x3, pad 1) / rgbd_conv2d_dgrad_bf16 / rgbd_conv3x3_actgrad_bf16 / rgbd_conv2d_fprop_stats_bf16
 *   with (xq, xs) and (wq, ws) in place of the bf16 operands; every other argument, the epilogues and the bf16 NHWC outputs
 *   are theirs.  Shapes: rgbd_conv3x3_mxfp8_supported(B, Hout, Wout, K-dim, N-dim) -- output images multiples of 16x16, the
 *   reduction channels a multiple of 128, the output channels of 64.
 */
typedef struct rgbd_pack_mx8_desc {
    const float* w;       /* (cout,cin,3,3) fp32 master */
    void* wf_q;           /* [9][cout][cin] e4m3 or NULL */
    void* wf_s;           /* [9][cout][cin/32] e8m0 */
    void* wd_q;           /* [9][cin][cout] e4m3, taps flipped, or NULL */
    void* wd_s;           /* [9][cin][cout/32] e8m0 */
    int cout, cin;
    float scale;
    int block_begin;
} rgbd_pack_mx8_desc;
/* rgbd_conv3x3_ex: the pipelined 3x3 pad-1 launch with every option behind one descriptor -- what the entry points above
 *   are special cases of, plus MXFP8 copies of its outputs: y_q / y_s (and yp_q / yp_s for y_pooled) receive exactly
 *   rgbd_quantize_mxfp8 of the stored bf16 tensor (blocks of 32 along Cout), so that the NEXT convolution of the chain
 *   (net.py:408-418: c0 -> c1 -> pooled -> next block) needs no quantiser pass.  x / w are bf16 when x_scales / w_scales are
 *   NULL, else e4m3 + E8M0.  NULL / 0 for what is not wanted; output images must be multiples of 16x16. */
typedef struct rgbd_conv3x3_desc {
    const void* x; const void* x_scales;          /* (B,Hin,Win,Cin) bf16, or e4m3 bytes + (B,Hin,Win,Cin/32) scales */
    const void* w; const void* w_scales;          /* [9][Cout][Cin] image (fprop, or the dgrad image with Cin / Cout exchanged) */
    const float* bias; const void* residual;      /* (Cout) fp32; (B,Hout,Wout,Cout) bf16 */
    const void* act_y;                            /* masked form (rgbd_conv3x3_actgrad_bf16) */
    float* colsum; const float* row_scale;
    void* y; void* y_pooled; void* y2; const float* row_scale2;
    int64_t* stats;                               /* statistics form (rgbd_conv2d_fprop_stats_bf16), zeroed by the caller */
    void* y_q; void* y_s; void* yp_q; void* yp_s; /* MXFP8 copies of y / y_pooled */
    int B, Hin, Win, Cin, Cout, upsample, pool_sum, lrelu_channels;
    float slope;
    int cus;                                      /* compute-unit budget of this launch (see `cus` above); 0 = all */
} rgbd_conv3x3_desc;
int rgbd_conv3x3_ex(const rgbd_conv3x3_desc* desc, void* stream);
int rgbd_quantize_mxfp8(const void* x, void* q, void* scales, int64_t rows, int C, void* stream);
int rgbd_pack_weights_mxfp8_multi(const rgbd_pack_mx8_desc* descs_device, int n, int total_blocks, void* stream);
int rgbd_conv3x3_mxfp8_supported(int B, int Hout, int Wout, int Cin, int Cout);
int rgbd_conv2d_fprop_mxfp8(const void* xq, const void* xs, const void* wq, const void* ws, const float* bias,
                            const void* residual, void* y, void* y_pooled, int B, int Hin, int Win, int Cin, int Cout,
                            int upsample, int lrelu_channels, float slope, int cus, void* stream);
int rgbd_conv2d_dgrad_mxfp8(const void* dyq, const void* dys, const void* wdq, const void* wds, const void* residual,
                            void* dx, int B, int H, int W, int Cin, int Cout, int sum_pool2, int cus, void* stream);
int rgbd_conv3x3_actgrad_mxfp8(const void* xq, const void* xs, const void* wq, const void* ws, const void* residual,
                               const void* act_y, float slope, float* colsum, const float* row_scale, void* y, void* y2,
                               const float* row_scale2, int B, int H, int W, int Cin, int Cout, int cus, void* stream);
int rgbd_conv2d_fprop_stats_mxfp8(const void* xq, const void* xs, const void* wq, const void* ws, const float* bias, void* y,
                                  int64_t* stats, int B, int Hin, int Win, int Cin, int Cout, int upsample,
                                  int lrelu_channels, float slope, int cus, void* stream);
int rgbd_pixelnorm_fwd(const float* x, float* y, int M, int C, float eps, void* stream);
int rgbd_pixelnorm_bwd(const float* x, const float* dy, float* dx, int M, int C, float eps, void* stream);
int rgbd_depth_head_fwd(const float* x, float* y, int B, int HW, void* stream);
int rgbd_depth_head_bwd(const float* x, const float* y, const float* dy, float* dx, int B, int HW, void* stream);
int rgbd_ema_update(float* dst, const float* src, int64_t n, float tau, void* stream);
/* Logit heads of the non-saturating GAN loss (loss_functions.py:15-28, updater.py:331-336,404-408) on n logits, one launch:
 *   losses[0] = mean softplus(-y), losses[1] = mean softplus(y); seed_neg / seed_pos (n) their derivatives w.r.t. y
 *   (-sigmoid(-y)/n, sigmoid(y)/n, taken at max(y, -60)); ratio (n) = seed_neg / seed_pos = -exp(-max(y, -60)). */
int rgbd_gan_logit_heads(const float* y, int n, float* losses, float* seed_neg, float* seed_pos, float* ratio,
                         void* stream);
/* One term of the DCGAN-style losses (loss_functions.py:15-31; updater_deepvoxels.py:170,229-232) on n logits, one launch:
 *   loss[0] = mean_i softplus(sign y_i) * sigmoid(sign y_i)^gamma   (sign = +-1; gamma = 0: the plain term, > 0: focal),
 *   dy[i]   = d loss / d y_i. */
int rgbd_softplus_mean(const float* y, int n, float sign, float gamma, float* loss, float* dy, void* stream);
/* Clear n floats with a kernel launch (a plain kernel node inside captured HIP graphs, unlike hipMemsetAsync). */
int rgbd_zero_f32(float* p, int64_t n, void* stream);
/* updater.py:336,360,439 (`assert not xp.isnan(loss.data)`) without a host synchronisation: scalars_host is a HOST array of n <= 8
 * DEVICE pointers to fp32 scalars; bit i of *mask (device int32, OR-ed: sticky) is set when scalar i is NaN or +-Inf. */
int rgbd_nonfinite_mask_f32(const float* const* scalars_host, int n, int32_t* mask, void* stream);

/* ------------------------------------------------------------------ optimizer
 * Replaces chainer.optimizers.Adam + GradientClipping(5) (train_rgbd.py:151-161), one launch group per
 * optimizer instead of one elementwise kernel per parameter tensor.
 *   p, g, m, v : flat fp32 buffers of n elements (all parameters of one optimizer, contiguous).
 *   grad_scale : multiplied into g first (1/world_size after an all-reduce(sum)).
 *   Global L2 norm of grad_scale*g is computed on the device; rate = min(1, clip/norm).
 *   Segments (HOST arrays): seg_begin[i] .. seg_begin[i+1] (nseg+1 entries) use base step size seg_alpha[i].
 *   step: device int32 holding chainer's update counter t; the call increments it and applies the bias correction
 *         alpha_t = alpha * sqrt(1-beta2^t) / (1-beta1^t) on the device, so the launch sequence and its arguments are
 *         identical every iteration (the whole training step can be captured in a HIP graph).
 *   m += (1-b1)(g-m); v += (1-b2)(g*g-v); p -= alpha_t * m / (sqrt(v) + eps)   (eps outside the correction)
 *   workspace: >= 1024 + 8 floats.  norm_out: optional device float receiving the pre-clip norm.
 */
int rgbd_adam_clip_multi(float* p, float* g, float* m, float* v, int64_t n,
                         int nseg, const int64_t* seg_begin, const float* seg_alpha,
                         float beta1, float beta2, float eps, float clip, float grad_scale,
                         int32_t* step, float* workspace, float* norm_out, void* stream);

#ifdef __cplusplus
}
#endif
#endif
