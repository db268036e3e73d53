/*
 * rgbd_debug.h -- test / tuning hooks of librgbdgan_hip.so.  NOT part of the drop-in C ABI (include/rgbd_gan_hip.h): nothing
 * on the training path calls these; tests/ and scripts/ do (A/B timing of kernel variants, cross-checks of the conv planner's
 * choices, profiling labels).
 *
 * rgbd_last_conv_kernel is exported by every build (a thread-local label, no switch).  The two SWITCHES below are process-wide
 * state, which the ABI proper never has: they -- and the kernels they select (round 1's register-staged 3x3 kernel, the
 * tap-split weight-gradient body, the timing knock-outs of the pipelined kernel) -- exist only in librgbdgan_hip_debug.so
 * (`python -m rgbd_gan_amd.build --debug`, -DRGBD_DEBUG_BUILD; rgbd_gan_amd/_lib.py:debug_library).  The shipped
 * librgbdgan_hip.so contains neither.
 */
#pragma once
#ifdef __cplusplus
extern "C" {
#endif

/* Test hook: when on != 0, rgbd_conv2d_fprop_bf16 uses the generic gather kernel for every shape (by default 3x3
 * pad-1 convolutions on images of 16x16 and larger run the halo-patch kernel). */
int rgbd_debug_force_gather_kernel(int on);
/* Name of the kernel the last rgbd_conv2d_fprop_bf16 / rgbd_conv2d_dgrad_bf16 call of this process launched (the planner
 * picks between the pipelined 3x3 kernel and the gather kernel by shape); for profiling labels. */
const char* rgbd_last_conv_kernel(void);
/* Test / tuning hook: 0 = default kernels, 1 = the register-staged 3x3 halo-patch kernel instead of the pipelined LDS-DMA
 * one, 2 = the pipelined kernel with 64-channel output tiles everywhere, 11-16 = timing knock-outs (wrong results). */
int rgbd_debug_conv_variant(int v);
/* Test hook: when on != 0, rgbd_occlusion_accum_fwd runs its three-kernel form (score, scan, compose) for the 32-feature
 * grids too, which by default take the single-pass kernel: the bit-identity A/B of tests/test_deepvoxels.py. */
void rgbd_debug_occ_unfused(int on);

#ifdef __cplusplus
}
#endif
