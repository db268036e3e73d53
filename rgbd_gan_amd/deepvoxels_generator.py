"""DeepVoxels generator with the reference's surface (deepvoxels_generator.py of nogu-atsu/RGBD-GAN) on the HIP
kernels: config 4 (configs/deepvoxels_shapenet_car.yml, occlusion_type "accumulative").

    MappingNetwork3D(ch)        .make_hidden(n), __call__(z)                          deepvoxels_generator.py:28-68
    VoxelGenerator(ch, ch_out)  __call__(w) -> (B,32,32,32,32)                         :112-188
    StyleGenerator(w_ch, in_ch, hidden_ch)   __call__(h, w, stage) -> (B,3,64,64)      :191-222   (the 2-D renderer)
    Generator(ch, occlusion_type, background_generator, config)                        :225-323
        .mapping, .projection, .make_hidden(n), __call__(z, stage, camera_matrices, z2=None, z3=None, z4=None, theta=None)

How the layers map onto the conv engine (rgbd_gan_amd/csrc/conv.hip, NHWC bf16, channels in multiples of 64):
  * 3x3x3 convolution: the three depth taps are folded into input channels -- the input volume (B,D,H,W,C) is
    concatenated with its two depth-shifted copies to (B*D,H,W,3C) and ONE 3x3 implicit-GEMM conv with the weight
    rearranged to (Cout, 3C, 3, 3) produces all 27 taps with a single fp32 accumulation.  Nearest 2x upsampling is
    a depth repeat plus the conv kernel's own fused H/W upsample.
  * 4x4 stride-2 convolution: the 16 taps are folded into input channels (B,H/2,W/2,16C) and run as a 1x1 conv (a
    plain GEMM, M = B*H*W/4, K = 16C).
  * channel counts that are not multiples of 64 (32-channel voxel blocks, the 288 -> 3 output conv) are zero padded;
    padded activations are exactly zero everywhere (zero weights, biases and style shifts), so results are unchanged.
The master parameters keep the reference's shapes and Chainer names; the rearranged weights are differentiable
views of them (functional.DerivedConvLayer).  Frustum resampling and the accumulative occlusion compositing are the
fp32 kernels of csrc/deepvoxels.hip (deepvoxel/projection.py, deepvoxel/deepvoxel.py).

Not supported (unreachable with the shipped config, asserted): occlusion types "deepvoxels" / "rendernet",
background_generator, enable_blur.
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import functional as Fn
from .deepvoxel.deepvoxel import (accumulative_occlusion, frustum_kernels_apply, interpolate_trilinear_batch,
                                  interpolate_trilinear_frustum)
from .deepvoxel.projection import ProjectionHelper
from .net import BF16, SQRT2, _Link, _as_device_tensor, _inv_c
from .params import ParamStore

GRID_FEATS = 32
OCC_NF = 4


def _pad_to(t, n, dim=-1):
    """Zero-pad dimension `dim` of t up to n entries (differentiable).  The common case -- the last dimension of a bf16 /
    fp32 device tensor -- is one HIP launch forward and one backward (rgbd_pad_last)."""
    extra = n - t.shape[dim]
    if extra == 0:
        return t
    dim = dim % t.dim()
    if dim == t.dim() - 1 and t.is_cuda and t.dtype in (torch.bfloat16, torch.float32):
        return Fn.pad_last(t, n)
    pads = [0, 0] * (t.dim() - 1 - dim) + [0, extra]
    return F.pad(t, pads)


def _ceil64(n):
    return (n + 63) // 64 * 64


def fold_conv3d_weight(W):
    """(Cout,Cin,3,3,3) -> (Cout64, 3*Cin64, 3, 3): input channel kd*Cin64 + ci carries depth tap kd."""
    w = _pad_to(_pad_to(W, _ceil64(W.shape[1]), 1), _ceil64(W.shape[0]), 0)
    return w.permute(0, 2, 1, 3, 4).reshape(w.shape[0], 3 * w.shape[1], 3, 3)


def fold_depth_taps(x):
    """(B,D,H,W,C) -> (B*D,H,W,3C): every depth slice next to its two zero-padded neighbours."""
    B, D, H, W, C = x.shape
    xp = F.pad(x, (0, 0, 0, 0, 0, 0, 1, 1))
    return torch.cat([xp[:, 0:D], xp[:, 1:D + 1], xp[:, 2:D + 2]], dim=-1).reshape(B * D, H, W, 3 * C)


def _fold_args(W, mode):
    """(mode, Cop, Cip) of kernels.fold_weight for master parameter W."""
    return (mode, _ceil64(W.shape[0]) if mode != 1 else W.shape[0], _ceil64(W.shape[1]) if mode != 1 else W.shape[1])


def _fold_w(W, mode):
    """Master parameter -> the conv engine's weight: mode 0 3x3x3 depth-tap fold, 1 4x4 stride-2 tap fold, 2 channel padding
    only; on the device one launch each way (rgbd_fold_weight_f32), on the CPU the torch formulations below."""
    if W.is_cuda:
        return Fn.fold_weight(W, *_fold_args(W, mode))
    if mode == 0:
        return fold_conv3d_weight(W)
    if mode == 1:
        return fold_4x4s2_weight(W)
    return _pad_to(_pad_to(W, _ceil64(W.shape[1]), 1), _ceil64(W.shape[0]), 0)


def fold_4x4s2_weight(W):
    """(Cout,Cin,4,4) -> (Cout,16*Cin,1,1): input channel (ky*4 + kx)*Cin + ci."""
    return W.permute(0, 2, 3, 1).reshape(W.shape[0], 16 * W.shape[1], 1, 1)


def fold_4x4s2(x):
    """(B,H,W,C) -> (B,H/2,W/2,16C): channel (ky*4 + kx)*C + c holds x_pad[2i+ky, 2j+kx, c] (pad 1)."""
    B, H, W, C = x.shape
    xp = F.pad(x, (0, 0, 1, 1, 1, 1))
    return torch.cat([xp[:, ky:ky + H:2, kx:kx + W:2] for ky in range(4) for kx in range(4)], dim=-1)


class MappingNetwork3D(_Link):
    """deepvoxels_generator.py:28-68.  Owned by Generator but NOT one of its registered children (:271), so it has
    its own optimizer ('map', train_rgbd.py:337) and its own snapshot file."""

    def __init__(self, ch=512, device="cuda:0", seed=0):
        self.ch = ch
        self.device = torch.device(device)
        specs = []
        for i in range(0, 16, 2):
            specs += [(f"l/{i}/c/W", (ch, ch), "normal"), (f"l/{i}/c/b", (ch,), "zeros")]
        self.store = ParamStore(specs, device, seed)
        self.stores = (("", self.store),)
        self.inv_c = _inv_c(ch)

    def make_hidden(self, batch_size):
        """:53-62: plain N(0,1), shape (B,ch,1,1,1) (the normalisation happens in forward)."""
        return torch.randn(batch_size, self.ch, 1, 1, 1, device=self.device)

    def __call__(self, x):
        h = Fn.pixel_norm(_as_device_tensor(x, self.device).reshape(x.shape[0], -1))
        p = self.store.params
        # one launch per pass (rgbd_mlp_fwd / rgbd_mlp_bwd, as net.MappingNetwork) where the width is covered; the mapping sits
        # at the head of both generator passes of a step, on the critical path
        return Fn.mlp_chain(h, [p[f"l/{i}/c/W"] for i in range(0, 16, 2)], [p[f"l/{i}/c/b"] for i in range(0, 16, 2)],
                            self.inv_c)

    forward = __call__


class _StyleMixin:
    _styles = None          # {style block name: (Fn.StyleGroup, index)} of the pass in flight (device tensors only)

    def _style_group(self, w, names):
        """The [scale | shift] affines (:96-101) of ALL style blocks `names` of this network -- they share the latent `w` -- as
        ONE linear over their parameters, which sit back to back in the flat buffer (voxel_specs / renderer_specs), and one
        shared gradient buffer backward.  A block whose tensor is zero padded to the engine's 64-channel granularity (the
        32-channel voxel blocks) keeps its unpadded window of 2 co columns: the AdaIN kernels take the live channel count."""
        p = self.p
        names_w, names_b, offsets, lives, tot = [], [], [], [], 0
        for nm in names:
            full = p.prefix + nm
            co = self.store.shapes[full + "/s/c/W"][0]
            names_w += [full + "/s/c/W", full + "/b/c/W"]
            names_b += [full + "/s/c/b", full + "/b/c/b"]
            offsets.append(tot)
            lives.append(co)
            tot += 2 * co
        W = self.store.fused(tuple(names_w), (tot, self.w_ch))
        b = self.store.fused(tuple(names_b), (tot,))
        ss = Fn.linear_act(w, W, b, _inv_c(self.w_ch, 1.0), act=False)
        group = Fn.StyleGroup(ss, offsets, lives)
        return {nm: (group, j) for j, nm in enumerate(names)}

    def _style(self, name, w, h):
        """StyleBlock (:96-109): AdaIN(h, s(w), b(w)) over all spatial axes; h is (B, ..., Cpad) bf16."""
        p = self.p
        c = _inv_c(self.w_ch, 1.0)
        C = h.shape[-1]
        full = p.prefix + name
        if self._styles is not None and name in self._styles:
            shp = h.shape
            return Fn.adain_window(h.reshape(shp[0], -1, 1, C), *self._styles[name]).reshape(shp)
        if h.is_cuda and self.store.shapes[full + "/s/c/W"][0] == C:
            # scale and shift affines sit back to back in the flat buffer: one linear, [scale | shift] read in place
            W = self.store.fused((full + "/s/c/W", full + "/b/c/W"), (2 * C, self.w_ch))
            b = self.store.fused((full + "/s/c/b", full + "/b/c/b"), (2 * C,))
            shp = h.shape
            return Fn.adain_fused(h.reshape(shp[0], -1, 1, C), Fn.linear_act(w, W, b, c, act=False)).reshape(shp)
        scale = _pad_to(Fn.linear_act(w, p[name + "/s/c/W"], p[name + "/s/c/b"], c, act=False), C)
        shift = _pad_to(Fn.linear_act(w, p[name + "/b/c/W"], p[name + "/b/c/b"], c, act=False), C)
        shp = h.shape
        out = Fn.adain(h.reshape(shp[0], -1, 1, C), scale, shift)
        return out.reshape(shp)


def voxel_channels(ch):
    return [(ch // 4, ch // 4), (ch // 4, ch // 4), (ch // 8, ch // 4), (ch // 8, ch // 8)]      # (out, in)


def voxel_specs(prefix, ch, ch_out):
    specs = []
    for i, (co, ci) in enumerate(voxel_channels(ch)):
        pre = f"{prefix}net/{i}"
        if i == 0:
            specs.append((pre + "/W", (ci, 4, 4, 4), "ones"))
        specs += [(pre + "/b0/b", (co,), "zeros"), (pre + "/b1/b", (co,), "zeros"),
                  (pre + "/n0/b/W", (co,), "zeros"), (pre + "/n1/b/W", (co,), "zeros")]
        specs += [(pre + "/c0/c/W", (co, ci, 3, 3, 3), "normal"), (pre + "/c1/c/W", (co, co, 3, 3, 3), "normal")]
    # the style affines of ALL blocks back to back, [scale W | shift W] block by block in the order the forward pass runs
    # them, and their biases likewise: one matrix, one linear per pass (_StyleMixin._style_group)
    for i, (co, ci) in enumerate(voxel_channels(ch)):
        for s in ("s0", "s1"):
            specs += [(f"{prefix}net/{i}/{s}/s/c/W", (co, ch), "normal"), (f"{prefix}net/{i}/{s}/b/c/W", (co, ch), "normal")]
    for i, (co, ci) in enumerate(voxel_channels(ch)):
        for s in ("s0", "s1"):
            specs += [(f"{prefix}net/{i}/{s}/s/c/b", (co,), "ones"), (f"{prefix}net/{i}/{s}/b/c/b", (co,), "zeros")]
    specs += [(prefix + "out/c/W", (ch_out, ch // 8, 1, 1, 1), "normal"), (prefix + "out/c/b", (ch_out,), "zeros")]
    return specs


class VoxelGenerator(_Link, _StyleMixin):
    """:171-188 with SynthesisBlock3D :112-168 (add_noise False, enable_blur False)."""

    def __init__(self, ch, ch_out, device="cuda:0", seed=1, store=None, prefix=""):
        self.ch = self.w_ch = ch
        self.ch_out = ch_out
        self.prefix = prefix
        if store is None:
            store = ParamStore(voxel_specs(prefix, ch, ch_out), device, seed)
            self.stores = (("", store),)
        self.store = store
        self.p = _Prefixed(store.params, prefix)
        self.chans = voxel_channels(ch)
        self.c0, self.c1 = [None], []
        for i, (co, ci) in enumerate(self.chans):
            if i > 0:
                self.c0.append(self._conv3d_layer(f"net/{i}/c0/c/W", ci))
            self.c1.append(self._conv3d_layer(f"net/{i}/c1/c/W", co))
        W = self.p["out/c/W"]
        # (a 1x1x1 convolution: the 5-D master is the (Cout,Cin,1,1) matrix in memory -- padding-only fold, in the pack group)
        self.out = Fn.DerivedConvLayer(lambda: _fold_w(W.reshape(W.shape[0], W.shape[1], 1, 1), 2), _inv_c(W.shape[1]), 1, 0,
                                       master=W, fold=_fold_args(W, 2))

    def _conv3d_layer(self, name, cin):
        """(Cout,Cin,3,3,3) -> (Cout64, 3*Cin64, 3, 3) with input channel index kd*Cin64 + ci.
        pggan.py:31: inv_c = sqrt(2) * sqrt(1 / (in_ch * ksize**2)) -- ksize squared, also in 3-D."""
        W = self.p[name]
        return Fn.DerivedConvLayer(lambda: _fold_w(W, 0), _inv_c(cin * 9), 3, 1, master=W, fold=_fold_args(W, 0))

    @staticmethod
    def _conv3d(x, layer, bias, upsample):
        """x (B,D,H,W,C) -> lrelu(conv3d(up(x)) + bias) as one 2-D conv over (B*D) depth slices."""
        B, D = x.shape[0], (2 * x.shape[1] if upsample else x.shape[1])
        if x.is_cuda:                                  # depth repeat + tap fold: one launch (rgbd_fold_depth_taps_bf16)
            folded = Fn.fold_depth_taps(x, upsample)
        else:
            if upsample:
                x = x.unsqueeze(2).expand(B, D // 2, 2, *x.shape[2:]).reshape(B, D, *x.shape[2:])
            folded = fold_depth_taps(x)
        y = Fn.conv_bias_lrelu(folded, layer, bias, upsample=upsample)
        return y.reshape(B, D, y.shape[1], y.shape[2], y.shape[3])

    def _block(self, i, w, x):
        p = self.p
        pre = f"net/{i}"
        co, ci = self.chans[i]
        C = _ceil64(co)
        if i == 0 and ci == C and p[pre + "/W"].is_cuda:
            # lrelu(W + b0) for every sample as (B,4,4,4,C) bf16 in ONE launch, gradients straight into the bound buffers
            # (rgbd_const_input_{fwd,bwd}, as SynthesisBlock 0 of the 2-D networks) instead of ~10 torch launches per call
            h = Fn.const_input(p[pre + "/W"], p[pre + "/b0/b"], w.shape[0])
        elif i == 0:
            const = p[pre + "/W"].permute(1, 2, 3, 0).unsqueeze(0)                       # (1,4,4,4,ci)
            h = _pad_to(Fn.lrelu(const + p[pre + "/b0/b"]), C).to(BF16)
            h = h.expand(w.shape[0], 4, 4, 4, C).contiguous()
        else:
            h = self._conv3d(x, self.c0[i], _pad_to(p[pre + "/b0/b"], C), True)
        h = self._style(pre + "/s0", w, h)
        h = self._conv3d(h, self.c1[i], _pad_to(p[pre + "/b1/b"], C), False)
        return self._style(pre + "/s1", w, h)

    def __call__(self, w, feature_minor=False):
        """feature_minor: return (B,32,32,32,ch_out) fp32 -- the conv stack's own layout, which the frustum resampling reads
        directly (rgbd_trilinear_fwd_fm) -- instead of the reference's (B,ch_out,32,32,32)."""
        h = None
        self._styles = self._style_group(w, [f"net/{i}/{s}" for i in range(4) for s in ("s0", "s1")]) if w.is_cuda else None
        try:
            for i in range(4):
                h = self._block(i, w, h)
        finally:
            self._styles = None
        B, D, H, W, C = h.shape
        y = Fn.conv_bias(h.reshape(B * D, H, W, C), self.out, _pad_to(self.p["out/c/b"], _ceil64(self.ch_out)))
        y = y.reshape(B, D, H, W, -1)[..., :self.ch_out]
        if feature_minor:
            return y.float().contiguous()
        return y.permute(0, 4, 1, 2, 3).float().contiguous()                            # b x ch_out x 32 x 32 x 32

    forward = __call__


class _Prefixed:
    """View of a parameter dict under a name prefix."""

    def __init__(self, params, prefix):
        self.params, self.prefix = params, prefix

    def __getitem__(self, name):
        return self.params[self.prefix + name]


def renderer_specs(prefix, w_ch, in_ch, hidden):
    h = hidden
    specs = []
    convs = {"c0": (2 * h, in_ch, 4), "c1": (4 * h, 2 * h, 4), "c4": (4 * h, 4 * h, 3), "c5": (2 * h, 4 * h, 3),
             "c6": (h, 4 * h, 3), "c7": (3, h + in_ch, 3)}
    for name, (co, ci, k) in convs.items():
        specs += [(f"{prefix}{name}/c/W", (co, ci, k, k), "normal"), (f"{prefix}{name}/c/b", (co,), "zeros")]
    styles = {"s0": 2 * h, "s1": 4 * h, "s4": 4 * h, "s5": 2 * h, "s6": h}      # forward order; all W's, then all b's (one linear)
    for name, co in styles.items():
        specs += [(f"{prefix}{name}/s/c/W", (co, w_ch), "normal"), (f"{prefix}{name}/b/c/W", (co, w_ch), "normal")]
    for name, co in styles.items():
        specs += [(f"{prefix}{name}/s/c/b", (co,), "ones"), (f"{prefix}{name}/b/c/b", (co,), "zeros")]
    return specs


class StyleGenerator(_Link, _StyleMixin):
    """The 2-D rendering network (:191-222): 64x64x32 features -> 64x64 RGB, U-shaped with two skip concatenations."""

    def __init__(self, w_ch, in_ch, hidden_ch=256, device="cuda:0", seed=2, store=None, prefix=""):
        self.w_ch, self.in_ch, self.hidden = w_ch, in_ch, hidden_ch
        if store is None:
            store = ParamStore(renderer_specs(prefix, w_ch, in_ch, hidden_ch), device, seed)
            self.stores = (("", store),)
        self.store = store
        self.p = p = _Prefixed(store.params, prefix)
        self.layers = {}
        for name in ("c0", "c1"):                       # 4x4 stride 2 pad 1 -> 1x1 over 16 folded taps
            W = p[name + "/c/W"]
            self.layers[name] = Fn.DerivedConvLayer(lambda W=W: _fold_w(W, 1), _inv_c(W.shape[1] * 16), 1, 0, master=W,
                                                    fold=_fold_args(W, 1))
        for name in ("c4", "c5", "c6"):
            W = p[name + "/c/W"]
            self.layers[name] = Fn.ConvLayer(W, _inv_c(W.shape[1] * 9), 1)
        W7 = p["c7/c/W"]
        self.layers["c7"] = Fn.DerivedConvLayer(lambda: _fold_w(W7, 2), _inv_c(W7.shape[1] * 9, 0.5), 3, 1, master=W7,
                                                fold=_fold_args(W7, 2))

    def __call__(self, h, w, stage=None):
        p = self.p
        L = self.layers
        x = h.permute(0, 2, 3, 1).to(BF16).contiguous()                                     # (B,64,64,32)
        fold = Fn.fold_4x4s2 if x.is_cuda else fold_4x4s2
        self._styles = self._style_group(w, ["s0", "s1", "s4", "s5", "s6"]) if w.is_cuda else None
        try:
            h1 = self._style("s0", w, Fn.conv_bias_lrelu(fold(x), L["c0"], p["c0/c/b"]))
            h2 = self._style("s1", w, Fn.conv_bias_lrelu(fold(h1), L["c1"], p["c1/c/b"]))
            h3 = self._style("s4", w, Fn.conv_bias_lrelu(h2, L["c4"], p["c4/c/b"]))
            h3 = Fn.conv_bias_lrelu(h3, L["c5"], p["c5/c/b"], upsample=True)
            h3 = torch.cat([self._style("s5", w, h3), h1], dim=-1)
            h3 = Fn.conv_bias_lrelu(h3, L["c6"], p["c6/c/b"], upsample=True)
            h3 = torch.cat([self._style("s6", w, h3), x], dim=-1)
        finally:
            self._styles = None
        h3 = _pad_to(h3, _ceil64(h3.shape[-1]))
        out = Fn.conv_bias(h3, L["c7"], _pad_to(p["c7/c/b"], 64))
        return out[..., :3].permute(0, 3, 1, 2).float().contiguous()

    forward = __call__


class DeepVoxels:
    """deepvoxel/deepvoxel.py:797-909 restricted to occlusion_type == "accumulative": frustum resampling of the feature
    grid for every camera, visibility weights from the two 1x1x1 convs, compositing along each ray, depth rescale."""

    def __init__(self, params, prefix, frustrum_img_dims, grid_dims, voxel_size, near_plane, threshold=None):
        self.p = _Prefixed(params, prefix + "occlusion_net/occlusion/")
        self.frustrum_img_dims = frustrum_img_dims
        self.frustrum_depth = int(np.ceil(np.sqrt(3) * grid_dims[-1]))
        self.voxel_size, self.near_plane = voxel_size, near_plane
        self.threshold = threshold if threshold else 4           # deepvoxel.py:555

    def __call__(self, idx, coords, counts, deepvoxels, feature_minor=False, frustum=None):
        if frustum is not None:             # straight from the cameras: no index list (Generator.__call__ decides)
            vol = interpolate_trilinear_frustum(deepvoxels, frustum)
        else:
            vol = interpolate_trilinear_batch(deepvoxels, idx, coords, counts, self.frustrum_img_dims, self.frustrum_depth,
                                              feature_minor=feature_minor)
        p = self.p
        W1, W2 = p["0/net/1/c/W"], p["2/net/1/c/W"]
        feats, depth, _ = accumulative_occlusion(vol, W1.reshape(W1.shape[0], W1.shape[1]), p["0/net/1/c/b"],
                                                 W2.reshape(1, -1), p["2/net/1/c/b"], float(self.threshold),
                                                 float(self.voxel_size), float(self.near_plane))
        return feats, depth


class Generator(_Link):
    """deepvoxels_generator.py:225-323."""

    def __init__(self, ch, occlusion_type="deepvoxels", background_generator=False, config=None, device="cuda:0",
                 seed=0):
        assert occlusion_type == "accumulative", \
            "only occlusion_type 'accumulative' (configs/deepvoxels_shapenet_car.yml) is built on the HIP kernels"
        assert not background_generator, "background_generator is not supported"
        assert ch % 256 == 0
        self.ch = ch
        self.device = torch.device(device)
        self.use_background_generator = False
        scale = 0.5
        grid_dim = 32
        near_plane = np.sqrt(3) / 4
        lift_intrinsic = np.array([[64 * 2., 0.0, 32., 0.0], [0.0, 64 * 2., 32., 0.0], [0.0, 0.0, 1.0, 0.0],
                                   [0.0, 0.0, 0.0, 1.0]])
        voxel_size = (1. / grid_dim) * 1.1 * scale
        depth_max = grid_dim * voxel_size + near_plane
        grid_dims = 3 * [grid_dim]
        proj_image_dims = [64, 64]
        frustrum_depth = int(np.ceil(np.sqrt(3) * grid_dims[-1]))
        self.projection = ProjectionHelper(projection_intrinsic=lift_intrinsic, lifting_intrinsic=lift_intrinsic,
                                           depth_min=0., depth_max=depth_max, projection_image_dims=proj_image_dims,
                                           lifting_image_dims=proj_image_dims, grid_dims=grid_dims,
                                           voxel_size=voxel_size, device=device, frustrum_depth=frustrum_depth,
                                           near_plane=near_plane)
        occ = "deepvoxel/occlusion_net/occlusion"
        specs = voxel_specs("voxel_gen/", ch, GRID_FEATS)
        specs += [(occ + "/0/net/1/c/W", (OCC_NF, GRID_FEATS + 1, 1, 1, 1), "normal"),
                  (occ + "/0/net/1/c/b", (OCC_NF,), "zeros"),
                  (occ + "/2/net/1/c/W", (1, OCC_NF, 1, 1, 1), "normal"), (occ + "/2/net/1/c/b", (1,), "zeros")]
        specs += renderer_specs("style_generator/", ch, GRID_FEATS, 256)
        for i, (co, ci) in zip((0, 2, 4), ((64, 8), (64, 64), (9, 64))):     # CameraParamGenerator (net.py:795-804): unused
            specs += [(f"camera_param_generator/net/{i}/c/W", (co, ci), "normal"),
                      (f"camera_param_generator/net/{i}/c/b", (co,), "zeros")]
        self.store = ParamStore(specs, device, seed + 1)
        self.stores = (("", self.store),)
        self.voxel_gen = VoxelGenerator(ch, GRID_FEATS, device, store=self.store, prefix="voxel_gen/")
        threshold = getattr(config, "accumulative_threshold", None) if config is not None else None
        self.deepvoxel = DeepVoxels(self.store.params, "deepvoxel/", proj_image_dims, grid_dims, voxel_size, near_plane,
                                    threshold)
        self.style_generator = StyleGenerator(ch, GRID_FEATS, 256, device, store=self.store, prefix="style_generator/")
        self.mapping = MappingNetwork3D(ch, device, seed)        # not a registered child in the reference (:271)
        self.train = True
        if torch.device(device).type == "cuda":
            # all conv weights of the generator (folded 3-D / stride-2 / padded layers and the plain ones): persistent
            # folded + packed images, rebuilt together and only after THIS network's optimizer step
            vg = self.voxel_gen
            self.pack_group = Fn.DerivedPackGroup([l for l in vg.c0 + vg.c1 if l is not None] + [vg.out]
                                                  + list(self.style_generator.layers.values()))

    def mark_weight_images_current(self):
        """The packed / folded weight images on the device are those of the current master weights (DeepVoxelsUpdater rebuilds them
        right behind the generator's update, inside a captured phase: Python's epochs cannot know after a replay)."""
        self.pack_group.mark_current()

    def rebuild_weight_images(self):
        self.pack_group.layers[0].packed()

    def make_hidden(self, batch_size):
        """:273-283."""
        z = torch.randn(batch_size, self.ch, 1, 1, device=self.device)
        return z / torch.sqrt(torch.sum(z * z, dim=1, keepdim=True) / self.ch + 1e-8)

    def __call__(self, z, stage, camera_matrices, z2=None, z3=None, z4=None, theta=None, cut=None):
        """cut (a dict, filled here): run the 2-D renderer behind a CUT in the autograd graph -- cut["leaves"] = detached
        (features, depth, w2) the renderer and the output are built on, cut["below"] = the tensors they were detached from.  A
        backward pass from the output then stops at the leaves (their .grad), a second one from cut["below"] with those gradients
        does the rest: DeepVoxelsUpdater runs the renderer's weight gradients on another stream between the two."""
        z = _as_device_tensor(z, self.device)
        if z2 is None:
            z2 = self.make_hidden(z.shape[0])
        z2 = _as_device_tensor(z2, self.device)
        # :297,309 map z and z2 with the same network in two calls; one batch of 2B rows here (the mapping network has
        # no cross-row coupling: pixel norm and the linears act per row)
        n = z.shape[0]
        if 2 * n <= 64:
            ww = self.mapping(torch.cat([z.reshape(n, -1), z2.reshape(n, -1)], dim=0))
            w, w2 = ww[:n], ww[n:]
        else:
            w, w2 = self.mapping(z), self.mapping(z2)
        fm = w.is_cuda
        voxel = self.voxel_gen(w, feature_minor=fm)
        # the frustum's index list (deepvoxel/projection.py:48-105: three launches + 2 x 9 MB of lists per call) is only built when
        # the resampling needs it: the feature-minor kernels work straight from the cameras
        fr = self.projection.frustum(camera_matrices) if fm else None
        if fm and frustum_kernels_apply(fr, voxel.shape[-1], voxel.shape[0]):
            novel_feats, depth = self.deepvoxel(None, None, None, voxel, feature_minor=True, frustum=fr)
        else:
            idx, coords, counts = self.projection.compute_proj_idcs_batch(camera_matrices)
            novel_feats, depth = self.deepvoxel(idx, coords, counts, voxel, feature_minor=fm)
        if cut is not None and torch.is_grad_enabled():
            cut["below"] = (novel_feats, depth, w2)
            novel_feats, depth, w2 = (t.detach().requires_grad_(True) for t in cut["below"])
            cut["leaves"] = (novel_feats, depth, w2)
        novel_img = self.style_generator(novel_feats, w2, stage)
        return torch.cat([novel_img, depth], dim=1)

    forward = __call__
