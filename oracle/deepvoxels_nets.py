"""DeepVoxels generator of the reference (config 4), restated as torch-CPU fp32 functions.

Test infrastructure only (see oracle/__init__.py).  PARITY UNPINNED.

Follows deepvoxels_generator.py:28-68 (MappingNetwork3D), :112-168 (SynthesisBlock3D), :171-188 (VoxelGenerator),
:191-222 (renderer StyleGenerator), :225-323 (Generator) and deepvoxel/deepvoxel.py:872-909 (DeepVoxels.forward) with
occlusion_type "accumulative" and no background generator -- the only branch configs/deepvoxels_shapenet_car.yml
reaches.  Parameters are flat dicts keyed by the Chainer ``namedparams`` paths; tensors are NC(D)HW float32.
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import deepvoxels as dv
from .nets import SQRT2, eq_linear, inv_c, lrelu, pixel_norm

GRID_FEATS = 32
OCC_NF = 4


def _normal(gen, *shape):
    return torch.randn(*shape, generator=gen)


def init_mapping3d(ch=256, seed=0):
    gen = torch.Generator().manual_seed(seed)
    p = {}
    for i in range(0, 16, 2):
        p[f"l/{i}/c/W"] = _normal(gen, ch, ch)
        p[f"l/{i}/c/b"] = torch.zeros(ch)
    return p


def voxel_channels(ch):
    """(out, in) of the four SynthesisBlock3D (deepvoxels_generator.py:176-179)."""
    return [(ch // 4, ch // 4), (ch // 4, ch // 4), (ch // 8, ch // 4), (ch // 8, ch // 8)]


def init_deepvoxels_generator(ch=256, seed=1, hidden=256):
    """Parameters of Generator minus the (separately optimised) mapping network; N(0,1) weights, zero biases, ones for
    the style-scale biases and the constant 4x4x4 input."""
    gen = torch.Generator().manual_seed(seed)
    p = {}
    for i, (co, ci) in enumerate(voxel_channels(ch)):
        pre = f"voxel_gen/net/{i}"
        if i == 0:
            p[pre + "/W"] = torch.ones(ci, 4, 4, 4)
        for b in ("b0", "b1"):
            p[f"{pre}/{b}/b"] = torch.zeros(co)
        for n in ("n0", "n1"):
            p[f"{pre}/{n}/b/W"] = torch.zeros(co)
        for s in ("s0", "s1"):
            p[f"{pre}/{s}/s/c/W"] = _normal(gen, co, ch)
            p[f"{pre}/{s}/s/c/b"] = torch.ones(co)
            p[f"{pre}/{s}/b/c/W"] = _normal(gen, co, ch)
            p[f"{pre}/{s}/b/c/b"] = torch.zeros(co)
        p[pre + "/c0/c/W"] = _normal(gen, co, ci, 3, 3, 3)
        p[pre + "/c1/c/W"] = _normal(gen, co, co, 3, 3, 3)
    p["voxel_gen/out/c/W"] = _normal(gen, GRID_FEATS, ch // 8, 1, 1, 1)
    p["voxel_gen/out/c/b"] = torch.zeros(GRID_FEATS)
    occ = "deepvoxel/occlusion_net/occlusion"
    p[occ + "/0/net/1/c/W"] = _normal(gen, OCC_NF, GRID_FEATS + 1, 1, 1, 1)
    p[occ + "/0/net/1/c/b"] = torch.zeros(OCC_NF)
    p[occ + "/2/net/1/c/W"] = _normal(gen, 1, OCC_NF, 1, 1, 1)
    p[occ + "/2/net/1/c/b"] = torch.zeros(1)
    h = hidden
    convs = {"c0": (2 * h, GRID_FEATS, 4), "c1": (4 * h, 2 * h, 4), "c4": (4 * h, 4 * h, 3), "c5": (2 * h, 4 * h, 3),
             "c6": (h, 4 * h, 3), "c7": (3, h + GRID_FEATS, 3)}
    for name, (co, ci, k) in convs.items():
        p[f"style_generator/{name}/c/W"] = _normal(gen, co, ci, k, k)
        p[f"style_generator/{name}/c/b"] = torch.zeros(co)
    for name, co in {"s0": 2 * h, "s1": 4 * h, "s4": 4 * h, "s5": 2 * h, "s6": h}.items():
        pre = f"style_generator/{name}"
        p[pre + "/s/c/W"] = _normal(gen, co, ch)
        p[pre + "/s/c/b"] = torch.ones(co)
        p[pre + "/b/c/W"] = _normal(gen, co, ch)
        p[pre + "/b/c/b"] = torch.zeros(co)
    for i, (co, ci) in zip((0, 2, 4), ((64, 8), (64, 64), (9, 64))):       # CameraParamGenerator (net.py:795-804), unused
        p[f"camera_param_generator/net/{i}/c/W"] = _normal(gen, co, ci)
        p[f"camera_param_generator/net/{i}/c/b"] = torch.zeros(co)
    return p


def mapping3d(pm, z):
    """deepvoxels_generator.py:64-68."""
    h = pixel_norm(z.reshape(z.shape[0], -1))
    for i in range(0, 16, 2):
        h = lrelu(eq_linear(h, pm, f"l/{i}"))
    return h


def eq_conv3d(x, p, name, pad):
    """pggan.py:27-38: inv_c = gain * sqrt(1 / (in_ch * ksize**2)) -- ksize SQUARED also for the 3-D kernel."""
    W = p[name + "/c/W"]
    b = p.get(name + "/c/b")
    return F.conv3d(inv_c(W.shape[1] * W.shape[2] ** 2) * x, W, b, padding=pad)


def eq_conv2d(x, p, name, stride, pad, gain=SQRT2):
    W = p[name + "/c/W"]
    return F.conv2d(inv_c(W.shape[1] * W.shape[2] ** 2, gain) * x, W, p[name + "/c/b"], stride=stride, padding=pad)


def adain_nd(x, scale, shift, eps=1e-5):
    """normalization/adain.py:10-77 for any number of spatial axes."""
    B, C = x.shape[:2]
    flat = x.reshape(B * C, -1)
    mean = flat.mean(dim=1, keepdim=True)
    var = ((flat - mean) ** 2).mean(dim=1, keepdim=True)
    xhat = ((flat - mean) * (var + eps) ** -0.5).reshape(x.shape)
    bshape = (B, C) + (1,) * (x.dim() - 2)
    return xhat * scale.reshape(bshape) + shift.reshape(bshape)


def style_block(p, name, w, h):
    """deepvoxels_generator.py:96-109: both affine maps have gain 1."""
    return adain_nd(h, eq_linear(w, p, name + "/s", gain=1.0), eq_linear(w, p, name + "/b", gain=1.0))


def up2_3d(x):
    """rescale.py:8-9 (unpooling_3d k=2 s=2): nearest replication."""
    return x.repeat_interleave(2, dim=2).repeat_interleave(2, dim=3).repeat_interleave(2, dim=4)


def up2_2d(x):
    return x.repeat_interleave(2, dim=2).repeat_interleave(2, dim=3)


def synthesis_block3d(p, pre, w, x, upsample):
    """deepvoxels_generator.py:137-168 with add_noise False (the default of the call at :186)."""
    if upsample:
        h = eq_conv3d(up2_3d(x), p, pre + "/c0", 1)
    else:
        W = p[pre + "/W"]
        h = W.unsqueeze(0).expand(w.shape[0], *W.shape)
    h = lrelu(h + p[pre + "/b0/b"].reshape(1, -1, 1, 1, 1))
    h = style_block(p, pre + "/s0", w, h)
    h = eq_conv3d(h, p, pre + "/c1", 1)
    h = lrelu(h + p[pre + "/b1/b"].reshape(1, -1, 1, 1, 1))
    return style_block(p, pre + "/s1", w, h)


def voxel_generator(p, w):
    """deepvoxels_generator.py:183-188 -> (B,32,32,32,32)."""
    h = None
    for i in range(4):
        h = synthesis_block3d(p, f"voxel_gen/net/{i}", w, h, upsample=i > 0)
    return eq_conv3d(h, p, "voxel_gen/out", 0)


def render_features(p, voxel, cams, fr=None, threshold=4.0):
    """deepvoxel.py:872-909 (accumulative occlusion): per sample frustum resampling + compositing."""
    fr = fr or dv.Frustum()
    occ = "deepvoxel/occlusion_net/occlusion"
    W1 = p[occ + "/0/net/1/c/W"].reshape(OCC_NF, -1)
    W2 = p[occ + "/2/net/1/c/W"].reshape(1, OCC_NF)
    feats, depths = [], []
    for i in range(voxel.shape[0]):
        lin, vc = dv.proj_idcs_np(np.asarray(cams[i], dtype="float32"), fr)
        vol = dv.trilinear_torch(voxel[i:i + 1], lin, vc, fr)
        f, d, _ = dv.occlusion_torch(vol, W1, p[occ + "/0/net/1/c/b"], W2, p[occ + "/2/net/1/c/b"], fr, threshold)
        feats.append(f)
        depths.append(d)
    return torch.cat(feats), torch.cat(depths)


def renderer(p, h, w):
    """deepvoxels_generator.py:208-222."""
    sg = "style_generator"
    h1 = style_block(p, sg + "/s0", w, lrelu(eq_conv2d(h, p, sg + "/c0", 2, 1)))
    h2 = style_block(p, sg + "/s1", w, lrelu(eq_conv2d(h1, p, sg + "/c1", 2, 1)))
    h3 = style_block(p, sg + "/s4", w, lrelu(eq_conv2d(h2, p, sg + "/c4", 1, 1)))
    h3 = lrelu(eq_conv2d(up2_2d(h3), p, sg + "/c5", 1, 1))
    h3 = torch.cat([style_block(p, sg + "/s5", w, h3), h1], dim=1)
    h3 = lrelu(eq_conv2d(up2_2d(h3), p, sg + "/c6", 1, 1))
    h3 = torch.cat([style_block(p, sg + "/s6", w, h3), h], dim=1)
    return eq_conv2d(h3, p, sg + "/c7", 1, 1, gain=0.5)


def deepvoxels_generator(p, pm, z, z2, cams, return_parts=False):
    """deepvoxels_generator.py:285-323: z, z2 (B,ch); cams (B,4,4) -> (B,4,64,64) = [RGB, depth]."""
    w = mapping3d(pm, z)
    voxel = voxel_generator(p, w)
    feats, depth = render_features(p, voxel, cams)
    w2 = mapping3d(pm, z2)
    img = renderer(p, feats, w2)
    out = torch.cat([img, depth], dim=1)
    if return_parts:
        return out, voxel, feats
    return out
