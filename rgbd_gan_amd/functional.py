"""Autograd glue: differentiable ops whose forward AND backward are HIP kernels behind the C ABI.

The convolution triple (fprop, dgrad, wgrad) is closed under differentiation, which is what the R1 penalty
(updater.py:414-418 of the reference: chainer.grad(..., enable_double_backprop=True)) needs:

    d fprop(x,W)  -> dgrad(dy,W), wgrad(x,dy)
    d dgrad(dy,W) -> fprop(ddx,W), wgrad(ddx,dy)

so every backward below is itself built from these Functions and stays differentiable.

Layout: activations are NHWC bf16 tensors of shape (B,H,W,C); master weights are OIHW fp32 leaves
(the reference's layout) and the bf16 packed images the kernels read are cached per layer.
"""
import contextlib

import torch
import torch.nn.functional as F

from . import kernels

_WEIGHT_EPOCH = 0          # bumped whenever master weights change (optimizer step, checkpoint load)
_SKIP_WGRAD = False        # set while only input gradients are wanted (R1's inner grad)


def bump_weight_epoch():
    global _WEIGHT_EPOCH
    _WEIGHT_EPOCH += 1


@contextlib.contextmanager
def input_grads_only():
    """Inside, conv backward skips weight gradients (used for g = dD(x)/dx of the R1 penalty)."""
    global _SKIP_WGRAD
    old = _SKIP_WGRAD
    _SKIP_WGRAD = True
    try:
        yield
    finally:
        _SKIP_WGRAD = old


class ConvLayer:
    """One equalized-LR 3x3 (or 1x1) convolution: master weight + cached packed bf16 images.

    pggan.py:13-24: y = conv(inv_c * x, W) (+ b).  inv_c is folded into the packed weights, so
    dW_master = inv_c * wgrad(x, dy).
    """

    def __init__(self, weight, inv_c, pad):
        self.weight = weight              # (Cout,Cin,K,K) fp32 leaf
        self.inv_c = float(inv_c)
        self.K = weight.shape[2]
        self.pad = pad
        self._epoch = -1
        self._wf = self._wd = None

    def packed(self):
        if self._epoch != _WEIGHT_EPOCH:
            with torch.no_grad():
                self._wf, self._wd = kernels.pack_weights(self.weight.detach(), self.inv_c)
            self._epoch = _WEIGHT_EPOCH
        return self._wf, self._wd


def _sum_pool2(x):
    """(B,2H,2W,C) -> (B,H,W,C): adjoint of nearest-2x upsampling."""
    B, H, W, C = x.shape
    return x.view(B, H // 2, 2, W // 2, 2, C).sum(dim=(2, 4))


def upsample2(x):
    """(B,H,W,C) -> (B,2H,2W,C) nearest (rescale.py:4-5)."""
    B, H, W, C = x.shape
    return x.view(B, H, 1, W, 1, C).expand(B, H, 2, W, 2, C).reshape(B, 2 * H, 2 * W, C)


class _ConvFprop(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, layer, ups):
        ctx.layer, ctx.ups = layer, ups
        ctx.save_for_backward(x, w)
        wf, _ = layer.packed()
        return kernels.conv2d_fprop(x.contiguous(), wf, layer.K, layer.K, layer.pad, upsample=ups)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dy = dy.contiguous()
        dx = _ConvDgrad.apply(dy, w, ctx.layer, ctx.ups) if ctx.needs_input_grad[0] else None
        dw = None
        if ctx.needs_input_grad[1] and not _SKIP_WGRAD:
            dw = _ConvWgrad.apply(x, dy, ctx.layer, ctx.ups)
        return dx, dw, None, None


class _ConvDgrad(torch.autograd.Function):
    @staticmethod
    def forward(ctx, dy, w, layer, ups):
        ctx.layer, ctx.ups = layer, ups
        ctx.save_for_backward(dy, w)
        _, wd = layer.packed()
        dx = kernels.conv2d_fprop(dy.contiguous(), wd, layer.K, layer.K, layer.K - 1 - layer.pad)
        return _sum_pool2(dx) if ups else dx

    @staticmethod
    def backward(ctx, ddx):
        dy, w = ctx.saved_tensors
        ddx = ddx.contiguous()
        g_dy = _ConvFprop.apply(ddx, w, ctx.layer, ctx.ups) if ctx.needs_input_grad[0] else None
        g_w = None
        if ctx.needs_input_grad[1] and not _SKIP_WGRAD:
            g_w = _ConvWgrad.apply(ddx, dy, ctx.layer, ctx.ups)
        return g_dy, g_w, None, None


class _ConvWgrad(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, dy, layer, ups):
        xe = upsample2(x).contiguous() if ups else x.contiguous()
        return kernels.conv2d_wgrad(xe, dy.contiguous(), layer.K, layer.inv_c)

    @staticmethod
    def backward(ctx, ddw):
        raise NotImplementedError("third-order derivatives through the conv engine are not supported")


def conv(x, layer, upsample=False):
    """NHWC bf16 conv through the MFMA implicit-GEMM kernels (no bias / activation)."""
    return _ConvFprop.apply(x, layer.weight, layer, bool(upsample))


class _AdaIN(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, scale, shift):
        y, mean, rstd = kernels.adain_fwd(x.contiguous(), scale.contiguous(), shift.contiguous())
        ctx.save_for_backward(x, scale, mean, rstd)
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dy):
        x, scale, mean, rstd = ctx.saved_tensors
        dx, dscale, dshift = kernels.adain_bwd(x.contiguous(), dy.contiguous(), scale.contiguous(), mean, rstd)
        return dx, dscale, dshift


def adain(x, scale, shift):
    """normalization/adain.py:76-77 on NHWC bf16; scale/shift are (B,C) fp32."""
    return _AdaIN.apply(x, scale.float(), shift.float())


class _WarpLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, img, img_rot, coef, flags, lam, max_depth, min_depth):
        ctx.cfg = (flags, lam, max_depth, min_depth)
        ctx.save_for_backward(img, img_rot, coef)
        return kernels.warp_loss_fwd(img, img_rot, coef, flags, lam, max_depth, min_depth).reshape(())

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gl):
        img, img_rot, coef = ctx.saved_tensors
        flags, lam, max_depth, min_depth = ctx.cfg
        gi, gr = kernels.warp_loss_bwd(img, img_rot, coef, flags, lam, max_depth, min_depth,
                                       gl.reshape(1).float().contiguous())
        return gi, gr, None, None, None, None, None


def warp_loss(img, img_rot, coef, flags, lambda_geometric, max_depth=0.0, min_depth=0.0):
    return _WarpLoss.apply(img.contiguous(), img_rot.contiguous(), coef, int(flags), float(lambda_geometric),
                           float(max_depth), float(min_depth))


def avg_pool2_nhwc(x):
    """rescale.py:12-13 on NHWC (differentiable, twice)."""
    B, H, W, C = x.shape
    return x.view(B, H // 2, 2, W // 2, 2, C).mean(dim=(2, 4))


def lrelu(x):
    return F.leaky_relu(x, 0.2)


class _ConvBiasLrelu(torch.autograd.Function):
    """y = lrelu(conv(x, W) + b) with bias and activation fused into the MFMA kernel's epilogue
    (net.py:144-152,155-159: c0/c1 -> L.Bias -> F.leaky_relu).  First-order only (generator path)."""

    @staticmethod
    def forward(ctx, x, w, bias, layer, ups):
        wf, _ = layer.packed()
        x = x.contiguous()
        y = kernels.conv2d_fprop(x, wf, layer.K, layer.K, layer.pad, bias=bias.contiguous(), upsample=ups,
                                 lrelu_channels=w.shape[0])
        ctx.layer, ctx.ups = layer, ups
        ctx.save_for_backward(x, y)
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dy):
        x, y = ctx.saved_tensors
        layer, ups = ctx.layer, ctx.ups
        dz = torch.where(y > 0, dy, dy * 0.2).contiguous()
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            _, wd = layer.packed()
            dx = kernels.conv2d_fprop(dz, wd, layer.K, layer.K, layer.K - 1 - layer.pad)
            if ups:
                dx = _sum_pool2(dx)
        if ctx.needs_input_grad[1]:
            xe = upsample2(x).contiguous() if ups else x
            dw = kernels.conv2d_wgrad(xe, dz, layer.K, layer.inv_c)
        if ctx.needs_input_grad[2]:
            db = dz.float().sum(dim=(0, 1, 2))
        return dx, dw, db, None, None


def conv_bias_lrelu(x, layer, bias, upsample=False):
    return _ConvBiasLrelu.apply(x, layer.weight, bias, layer, bool(upsample))
