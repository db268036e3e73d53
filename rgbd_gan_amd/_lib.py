"""ctypes binding of librgbdgan_hip.so -- the only door between the Python host and the HIP kernels.

The product path has no fallback: if the shared library is missing or an entry point fails, an exception
is raised (RuntimeError carrying rgbd_last_error()).
"""
import ctypes
import os
from ctypes import c_char_p, c_float, c_int, c_int32, c_int64, c_void_p, POINTER

_HERE = os.path.dirname(os.path.abspath(__file__))
# RGBD_LIB_PATH: A/B timing of another build of the same ABI; the default is the in-tree library
LIB_PATH = os.environ.get("RGBD_LIB_PATH") or os.path.join(_HERE, "librgbdgan_hip.so")
ABI_VERSION = 20

_P = c_void_p

# name -> argtypes, exactly the prototypes of include/rgbd_gan_hip.h
PROTOTYPES = {
    "rgbd_abi_version": ([], c_int),
    "rgbd_last_error": ([], c_char_p),
    "rgbd_warp_loss_fwd": ([_P, _P, _P, c_int, c_int, c_int, c_float, c_float, c_float, c_float, c_float, _P, _P, _P, _P,
                            _P, _P], c_int),
    "rgbd_warp_loss_bwd_workspace": ([c_int, c_int], c_int64),
    "rgbd_warp_loss_bwd": ([_P, _P, _P, c_int, c_int, c_int, c_float, c_float, c_float, c_float, c_float, _P, c_float, _P,
                            _P, c_int, _P, _P], c_int),
    "rgbd_warp_loss_nc_fwd": ([_P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_float, c_float, c_float, _P, _P, _P], c_int),
    "rgbd_warp_loss_nc_bwd": ([_P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_float, c_float, c_float, _P, c_float, _P, _P,
                               c_int, _P, _P], c_int),
    "rgbd_warp_loss_nc_bwd_workspace": ([c_int, c_int, c_int], c_int64),
    "rgbd_pack_weights": ([_P, c_int, c_int, c_int, c_int, c_float, _P, _P, _P], c_int),
    "rgbd_pack_weights_multi": ([_P, c_int, c_int, _P], c_int),
    "rgbd_conv2d_fprop_workspace": ([c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int], c_int64),
    "rgbd_conv2d_fprop_bf16": ([_P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                c_int, c_int, c_float, _P, c_int, _P], c_int),
    "rgbd_fold_depth_taps_bf16": ([_P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P], c_int),
    "rgbd_fold_4x4s2_bf16": ([_P, _P, c_int, c_int, c_int, c_int, c_int, _P], c_int),
    "rgbd_pad_last": ([_P, _P, c_int64, c_int, c_int, c_int, _P], c_int),
    "rgbd_fold_weight_f32": ([_P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P], c_int),
    "rgbd_fold_weight_multi_f32": ([_P, c_int, _P], c_int),
    "rgbd_conv2d_wgrad_workspace": ([c_int, c_int, c_int, c_int, c_int, c_int], c_int64),
    "rgbd_conv2d_wgrad_bf16": ([_P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_float, c_int, c_int, _P], c_int),
    "rgbd_conv2d_wgrad_partial_bf16": ([_P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P], c_int),
    "rgbd_wgrad_reduce_multi": ([_P, c_int, _P], c_int),
    "rgbd_conv2d_wgrad_multi_plan": ([_P, c_int, c_int], c_int),
    "rgbd_conv2d_wgrad_partial_multi_bf16": ([_P, c_int, _P], c_int),
    "rgbd_adain_workspace": ([c_int, c_int, c_int], c_int64),
    "rgbd_adain_fwd": ([_P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_float, _P, _P, _P], c_int),
    "rgbd_adain_bwd": ([_P, _P, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_float, _P, _P, _P, _P], c_int),
    "rgbd_lrelu_bwd": ([_P, _P, _P, c_int64, c_int, c_int, c_float, _P, _P, c_int64, _P], c_int),
    "rgbd_colsum_bf16": ([_P, _P, c_int64, c_int, c_int, _P, c_int64, _P], c_int),
    "rgbd_axpy_rows_bf16": ([_P, _P, _P, _P, c_int64, c_int64, _P], c_int),
    "rgbd_unpool2_lrelu_bwd": ([_P, _P, _P, c_int, c_int, c_int, c_int, c_float, _P, _P, _P, _P, _P, _P], c_int),
    "rgbd_pool2_masked": ([_P, _P, _P, c_int, c_int, c_int, c_int, c_float, _P], c_int),
    "rgbd_pool2_sum_bf16": ([_P, _P, c_int, c_int, c_int, c_int, _P], c_int),
    "rgbd_from_planes": ([_P, _P, _P, _P, c_int, c_int, c_int, c_int, c_float, c_int, c_float, _P], c_int),
    "rgbd_to_planes": ([_P, _P, _P, _P, c_int, c_int, c_int, c_int, c_float, _P], c_int),
    "rgbd_planes_outer": ([_P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, _P], c_int),
    "rgbd_linear_fwd_workspace": ([c_int, c_int, c_int], c_int64),
    "rgbd_linear_fwd": ([_P, _P, _P, _P, c_int, c_int, c_int, c_float, c_int, c_float, _P, _P], c_int),
    "rgbd_mlp_fwd": ([_P, POINTER(c_void_p), POINTER(c_void_p), c_int, c_int, c_int, c_float, c_float, _P, _P], c_int),
    "rgbd_mlp_bwd": ([_P, _P, _P, POINTER(c_void_p), POINTER(c_void_p), POINTER(c_void_p), c_int, c_int, c_int, c_float,
                      c_float, _P, _P, _P], c_int),
    "rgbd_linear_fwd_masked": ([_P, _P, _P, _P, c_int, c_int, c_int, c_float, c_float, _P, _P], c_int),
    "rgbd_real_batch_u8": ([_P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P, c_float, _P], c_int),
    "rgbd_zero_multi_f32": ([POINTER(c_void_p), POINTER(c_int64), c_int, _P], c_int),
    "rgbd_hidden_normalize": ([_P, _P, c_int, c_int, c_float, c_int, _P], c_int),
    "rgbd_hidden_draw": ([_P, _P, c_int, c_int, c_float, c_int, _P], c_int),
    "rgbd_r1_penalty_fwd": ([_P, c_int, c_int64, c_float, _P, _P, _P], c_int),
    "rgbd_scale_by_scalar_f32": ([_P, _P, c_float, _P, c_int64, _P], c_int),
    "rgbd_axpy_rows_f32": ([_P, _P, _P, _P, c_int64, c_int64, _P], c_int),
    "rgbd_image_grad_init": ([_P, _P, _P, c_int, c_int, c_int, c_int, _P], c_int),
    "rgbd_const_input_fwd": ([_P, _P, _P, c_int, c_int, c_int, c_float, _P], c_int),
    "rgbd_const_input_bwd": ([_P, _P, _P, _P, _P, c_int, c_int, c_int, c_float, _P], c_int),
    "rgbd_fade_planes_fwd": ([_P, _P, _P, c_int64, c_int, c_int, _P, c_float, _P], c_int),
    "rgbd_fade_planes_bwd": ([_P, _P, _P, c_int64, c_int, c_int, _P, c_float, _P], c_int),
    "rgbd_lerp_bf16": ([_P, _P, _P, _P, c_int64, c_int, _P, c_float, _P], c_int),
    "rgbd_pool2_planes": ([_P, _P, c_int64, c_int, c_int, c_int, _P], c_int),
    "rgbd_l2norm_fwd": ([_P, _P, c_int64, c_int, c_float, _P], c_int),
    "rgbd_l2norm_bwd": ([_P, _P, _P, c_int64, c_int, c_float, _P], c_int),
    "rgbd_blur3x3_bf16": ([_P, _P, c_int, c_int, c_int, c_int, c_int, _P], c_int),
    "rgbd_nhwc_to_rows_f32": ([_P, _P, c_int, c_int, c_int, _P], c_int),
    "rgbd_rows_to_nhwc_bf16": ([_P, _P, c_int, c_int, c_int, _P], c_int),
    "rgbd_linear_bwd": ([_P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_float, c_int, c_float, c_int, _P], c_int),
    "rgbd_proj_idcs": ([_P, c_int, c_int, c_int, c_int, c_int, c_float, c_float, c_float, c_float, c_float, c_float,
                        _P, _P, _P, _P, _P], c_int),
    "rgbd_trilinear_fwd": ([_P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, _P], c_int),
    "rgbd_trilinear_fwd_fm": ([_P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, _P], c_int),
    "rgbd_trilinear_bwd_fm": ([_P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, _P], c_int),
    "rgbd_trilinear_fwd_frustum": ([_P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_float, c_float, c_float, c_float,
                                    c_float, c_float, _P, _P], c_int),
    "rgbd_trilinear_bwd_frustum_supported": ([c_int, c_int, c_int, c_int, c_int], c_int),
    "rgbd_trilinear_bwd_frustum": ([_P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_float, c_float, c_float, c_float,
                                    c_float, c_float, _P, _P], c_int),
    "rgbd_trilinear_bwd": ([_P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, _P], c_int),
    "rgbd_occlusion_accum_fwd": ([_P, _P, _P, _P, _P, c_float, c_float, c_float, _P, _P, _P, _P, c_int, c_int, c_int,
                                  c_int, _P], c_int),
    "rgbd_occlusion_accum_bwd": ([_P, _P, _P, _P, _P, _P, _P, _P, c_float, _P, _P, _P, _P, c_int, c_int, c_int, c_int,
                                  _P], c_int),
    "rgbd_conv2d_dgrad_bf16": ([_P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P, c_int, _P], c_int),
    "rgbd_conv3x3_actgrad_supported": ([c_int, c_int, c_int, c_int, c_int], c_int),
    "rgbd_conv3x3_actgrad_bf16": ([_P, _P, _P, _P, c_float, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P], c_int),
    "rgbd_conv2d_fprop_stats_bf16": ([_P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_float, c_int, _P], c_int),
    "rgbd_adain_apply_fixed": ([_P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_float, _P, _P, _P], c_int),
    "rgbd_conv3x3_ex": ([_P, _P], c_int),
    "rgbd_quantize_mxfp8": ([_P, _P, _P, c_int64, c_int, _P], c_int),
    "rgbd_pack_weights_mxfp8_multi": ([_P, c_int, c_int, _P], c_int),
    "rgbd_conv3x3_mxfp8_supported": ([c_int, c_int, c_int, c_int, c_int], c_int),
    "rgbd_conv2d_fprop_mxfp8": ([_P, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_float,
                                 c_int, _P], c_int),
    "rgbd_conv2d_dgrad_mxfp8": ([_P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P], c_int),
    "rgbd_conv3x3_actgrad_mxfp8": ([_P, _P, _P, _P, _P, _P, c_float, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int,
                                    c_int, _P], c_int),
    "rgbd_conv2d_fprop_stats_mxfp8": ([_P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_float,
                                       c_int, _P], c_int),
    "rgbd_pixelnorm_fwd": ([_P, _P, c_int, c_int, c_float, _P], c_int),
    "rgbd_pixelnorm_bwd": ([_P, _P, _P, c_int, c_int, c_float, _P], c_int),
    "rgbd_depth_head_fwd": ([_P, _P, c_int, c_int, _P], c_int),
    "rgbd_depth_head_bwd": ([_P, _P, _P, _P, c_int, c_int, _P], c_int),
    "rgbd_gan_logit_heads": ([_P, c_int, _P, _P, _P, _P, _P], c_int),
    "rgbd_softplus_mean": ([_P, c_int, c_float, c_float, _P, _P, _P], c_int),
    "rgbd_ema_update": ([_P, _P, c_int64, c_float, _P], c_int),
    "rgbd_zero_f32": ([_P, c_int64, _P], c_int),
    "rgbd_nonfinite_mask_f32": ([POINTER(c_void_p), c_int, _P, _P], c_int),
    "rgbd_adam_clip_multi": ([_P, _P, _P, _P, c_int64, c_int, POINTER(c_int64), POINTER(c_float), c_float, c_float,
                              c_float, c_float, c_float, _P, _P, _P, _P], c_int),
}

_lib = None


# rgbd_gan_amd/csrc/rgbd_debug.h: NOT part of the drop-in ABI.  The label query is exported by every build; the two planner
# switches (and the A/B reference kernels they select) exist in the debug library only (python -m rgbd_gan_amd.build --debug)
DEBUG_PROTOTYPES = {
    "rgbd_last_conv_kernel": ([], c_char_p),
}
DEBUG_ONLY_PROTOTYPES = {
    "rgbd_debug_force_gather_kernel": ([c_int], c_int),
    "rgbd_debug_conv_variant": ([c_int], c_int),
    "rgbd_debug_occ_unfused": ([c_int], None),
}
DEBUG_LIB_PATH = os.path.join(_HERE, "librgbdgan_hip_debug.so")


def _bind(path, extra=()):
    lib = ctypes.CDLL(path)
    for name, (argtypes, restype) in list(PROTOTYPES.items()) + list(DEBUG_PROTOTYPES.items()) + list(extra):
        fn = getattr(lib, name)  # AttributeError if a declared symbol is not exported
        fn.argtypes = argtypes
        fn.restype = restype
    got = lib.rgbd_abi_version()
    if got != ABI_VERSION:
        raise RuntimeError(f"{os.path.basename(path)} ABI {got} != expected {ABI_VERSION}; rebuild")
    return lib


def load():
    """Load the library once; raise (never fall back) when it is absent or has the wrong ABI."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: build it with `python -m rgbd_gan_amd.build` "
            "(hipcc --offload-arch=gfx950). There is no CPU or PyTorch fallback for the hot path.")
    _lib = _bind(LIB_PATH)
    if os.environ.get("RGBD_CONV_VARIANT") and hasattr(_lib, "rgbd_debug_conv_variant"):
        # tuning aid (scripts/soak.py): RGBD_LIB_PATH pointing at the debug library + a variant of rgbd_debug_conv_variant
        _lib.rgbd_debug_conv_variant.argtypes, _lib.rgbd_debug_conv_variant.restype = [c_int], c_int
        _lib.rgbd_debug_conv_variant(int(os.environ["RGBD_CONV_VARIANT"]))
    return _lib


_debug_lib = None


def load_debug():
    """The debug library (A/B reference kernels + planner switches): tests and tuning scripts only."""
    global _debug_lib
    if _debug_lib is None:
        if not os.path.exists(DEBUG_LIB_PATH):
            raise RuntimeError(f"{DEBUG_LIB_PATH} not found: build it with `python -m rgbd_gan_amd.build --debug`")
        _debug_lib = _bind(DEBUG_LIB_PATH, DEBUG_ONLY_PROTOTYPES.items())
    return _debug_lib


class debug_library:
    """with _lib.debug_library() as lib: ...  -- inside, every wrapper of rgbd_gan_amd.kernels calls the DEBUG library, whose
    switches `lib.rgbd_debug_conv_variant` / `lib.rgbd_debug_force_gather_kernel` select the reference kernels; both are put
    back to 0 on the way out.  Cross-checks and A/B timing only: the training path never enters this."""

    def __enter__(self):
        global _lib
        self.saved = _lib
        _lib = load_debug()
        return _lib

    def __exit__(self, *exc):
        global _lib
        _lib.rgbd_debug_conv_variant(0)
        _lib.rgbd_debug_force_gather_kernel(0)
        _lib.rgbd_debug_occ_unfused(0)
        _lib = self.saved


def check(rc, what):
    if rc != 0:
        msg = load().rgbd_last_error()
        raise RuntimeError(f"{what} failed (rc={rc}): {msg.decode() if msg else '?'}")
