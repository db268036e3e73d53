"""interpolate_trilinear and the accumulative occlusion compositing (reference deepvoxel/deepvoxel.py:388-428,
544-587, 886-889, 903-904) as differentiable ops on the HIP kernels.  fp32, NCDHW like the reference."""
import numpy as np
import torch

from .. import _lib
from ..kernels import _ptr, _stream, _timed

OCC_NF = 4
TRILINEAR_BWD_BRICKS = True     # the feature-minor backward over sorted bricks of the frustum (False: the row-wise list kernel)


class _Trilinear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, grid, idx, coords, counts, N):
        grid = grid.contiguous()
        B, F, G = grid.shape[0], grid.shape[1], grid.shape[2]
        out = torch.empty(B, F, N, dtype=torch.float32, device=grid.device)
        # algorithmic bytes: the grid read once + the resampled frustum written once
        rc = _timed("trilinear_fwd_kernel", 0.0, 4.0 * B * F * (G ** 3 + N),
                    lambda: _lib.load().rgbd_trilinear_fwd(_ptr(grid), _ptr(idx), _ptr(coords), _ptr(counts), _ptr(out),
                                                           B, F, G, N, _stream()))
        _lib.check(rc, "rgbd_trilinear_fwd")
        ctx.save_for_backward(idx, coords, counts)
        ctx.dims = (B, F, G, N)
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dout):
        idx, coords, counts = ctx.saved_tensors
        B, F, G, N = ctx.dims
        dgrid = torch.empty(B, F, G, G, G, dtype=torch.float32, device=dout.device)
        ws = torch.empty(B * G * G * G * F, dtype=torch.float32, device=dout.device)
        dout = dout.contiguous()
        rc = _timed("trilinear_bwd_kernel", 0.0, 4.0 * B * F * (G ** 3 + N),
                    lambda: _lib.load().rgbd_trilinear_bwd(_ptr(dout), _ptr(idx), _ptr(coords), _ptr(counts), _ptr(dgrid),
                                                           _ptr(ws), B, F, G, N, _stream()))
        _lib.check(rc, "rgbd_trilinear_bwd")
        return dgrid, None, None, None, None


class _TrilinearFM(torch.autograd.Function):
    """The same resampling on a feature-minor grid (B,G,G,G,F): the layout of the voxel generator's NHWC conv stack, so no
    transposition before the gather and none after the backward's scatter (its target IS the gradient)."""

    @staticmethod
    def forward(ctx, grid, idx, coords, counts, N):
        grid = grid.contiguous()
        B, G, F = grid.shape[0], grid.shape[1], grid.shape[4]
        out = torch.empty(B, F, N, dtype=torch.float32, device=grid.device)
        rc = _timed("trilinear_fwd_kernel", 0.0, 4.0 * B * F * (G ** 3 + N),
                    lambda: _lib.load().rgbd_trilinear_fwd_fm(_ptr(grid), _ptr(idx), _ptr(coords), _ptr(counts), _ptr(out),
                                                              B, F, G, N, _stream()))
        _lib.check(rc, "rgbd_trilinear_fwd_fm")
        ctx.save_for_backward(idx, coords, counts)
        ctx.dims = (B, F, G, N)
        ctx.frustum = getattr(idx, "_frustum", None) if TRILINEAR_BWD_BRICKS else None
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dout):
        idx, coords, counts = ctx.saved_tensors
        B, F, G, N = ctx.dims
        dgrid = torch.empty(B, G, G, G, F, dtype=torch.float32, device=dout.device)
        dout = dout.contiguous()
        fr = ctx.frustum
        lib = _lib.load()
        if fr is not None and fr[0].shape[0] == B and fr[1] * fr[2] * fr[3] == N and fr[4] == G and \
                lib.rgbd_trilinear_bwd_frustum_supported(fr[1], fr[2], fr[3], G, F):
            # bricks of the frustum, contributions sorted by voxel: one line atomic per distinct voxel of a brick
            cams = fr[0]
            rc = _timed("trilinear_bwd_kernel", 0.0, 4.0 * B * F * (G ** 3 + N),
                        lambda: lib.rgbd_trilinear_bwd_frustum(_ptr(dout), _ptr(cams), B, F, fr[1], fr[2], fr[3], G, fr[5], fr[6],
                                                               fr[7], fr[8], fr[9], fr[10], _ptr(dgrid), _stream()))
            _lib.check(rc, "rgbd_trilinear_bwd_frustum")
            return dgrid, None, None, None, None
        rc = _timed("trilinear_bwd_kernel", 0.0, 4.0 * B * F * (G ** 3 + N),
                    lambda: _lib.load().rgbd_trilinear_bwd_fm(_ptr(dout), _ptr(idx), _ptr(coords), _ptr(counts), _ptr(dgrid),
                                                              B, F, G, N, _stream()))
        _lib.check(rc, "rgbd_trilinear_bwd_fm")
        return dgrid, None, None, None, None


class _TrilinearFrustum(torch.autograd.Function):
    """The feature-minor resampling straight from the cameras (ProjectionHelper.frustum): no compacted list, no fill of the output
    in front of the forward (every element is written, zeros outside the grid), the backward over sorted bricks."""

    @staticmethod
    def forward(ctx, grid, cams, fr):
        grid = grid.contiguous()
        B, G, F = grid.shape[0], grid.shape[1], grid.shape[4]
        W, H, D = fr[1], fr[2], fr[3]
        N = W * H * D
        out = torch.empty(B, F, N, dtype=torch.float32, device=grid.device)
        rc = _timed("trilinear_fwd_kernel", 0.0, 4.0 * B * F * (G ** 3 + N),
                    lambda: _lib.load().rgbd_trilinear_fwd_frustum(_ptr(grid), _ptr(cams), B, F, W, H, D, G, fr[5], fr[6], fr[7], fr[8],
                                                                   fr[9], fr[10], _ptr(out), _stream()))
        _lib.check(rc, "rgbd_trilinear_fwd_frustum")
        ctx.save_for_backward(cams)
        ctx.fr, ctx.dims = fr, (B, F, G, N)
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dout):
        cams, = ctx.saved_tensors
        fr = ctx.fr
        B, F, G, N = ctx.dims
        dgrid = torch.empty(B, G, G, G, F, dtype=torch.float32, device=dout.device)
        dout = dout.contiguous()
        rc = _timed("trilinear_bwd_kernel", 0.0, 4.0 * B * F * (G ** 3 + N),
                    lambda: _lib.load().rgbd_trilinear_bwd_frustum(_ptr(dout), _ptr(cams), B, F, fr[1], fr[2], fr[3], G, fr[5], fr[6],
                                                                   fr[7], fr[8], fr[9], fr[10], _ptr(dgrid), _stream()))
        _lib.check(rc, "rgbd_trilinear_bwd_frustum")
        return dgrid, None, None


def frustum_kernels_apply(fr, F, B):
    """Can the list-free kernels take this frustum?  (16 x 8 bricks, <= 32 features, one camera per sample)"""
    return (TRILINEAR_BWD_BRICKS and fr is not None and fr[0].is_cuda and fr[0].shape[0] == B and
            bool(_lib.load().rgbd_trilinear_bwd_frustum_supported(fr[1], fr[2], fr[3], fr[4], F)))


def interpolate_trilinear_frustum(grid_fm, fr):
    """grid_fm (B,G,G,G,F) fp32, fr = ProjectionHelper.frustum(cameras) -> (B,F,depth,H,W)."""
    out = _TrilinearFrustum.apply(grid_fm, fr[0], tuple(fr))
    return out.reshape(grid_fm.shape[0], grid_fm.shape[4], fr[3], fr[2], fr[1])


def interpolate_trilinear_batch(grid, idx, coords, counts, img_shape, frustrum_depth, feature_minor=False):
    """grid (B,F,G,G,G) [feature_minor: (B,G,G,G,F)]; idx/coords/counts from ProjectionHelper.compute_proj_idcs_batch ->
    (B,F,depth,H,W)."""
    N = img_shape[0] * img_shape[1] * frustrum_depth
    if feature_minor:
        out = _TrilinearFM.apply(grid, idx, coords, counts, N)
        return out.reshape(grid.shape[0], grid.shape[4], frustrum_depth, img_shape[0], img_shape[1])
    out = _Trilinear.apply(grid, idx, coords, counts, N)
    return out.reshape(grid.shape[0], grid.shape[1], frustrum_depth, img_shape[0], img_shape[1])


def interpolate_trilinear(grid, lin_ind_frustrum, voxel_coords, img_shape, frustrum_depth):
    """Reference signature (deepvoxel.py:388): grid (1,F,G,G,G), one sample's compacted indices / coordinates."""
    N = img_shape[0] * img_shape[1] * frustrum_depth
    m = lin_ind_frustrum.shape[0]
    idx = torch.zeros(1, N, dtype=torch.int32, device=grid.device)
    coords = torch.zeros(1, 3, N, dtype=torch.float32, device=grid.device)
    idx[0, :m] = lin_ind_frustrum.to(torch.int32)
    coords[0, :, :m] = voxel_coords
    counts = torch.tensor([m], dtype=torch.int32, device=grid.device)
    return interpolate_trilinear_batch(grid, idx, coords, counts, img_shape, frustrum_depth)


class _OcclusionAccum(torch.autograd.Function):
    @staticmethod
    def forward(ctx, vol, W1, b1, W2, b2, threshold, voxel_size, near_plane):
        vol = vol.contiguous()
        B, F, D, H, W = vol.shape
        dev = vol.device
        s = torch.empty(B, D, H * W, dtype=torch.float32, device=dev)
        w = torch.empty_like(s)
        feat = torch.empty(B, F, H, W, dtype=torch.float32, device=dev)
        depth = torch.empty(B, 1, H, W, dtype=torch.float32, device=dev)
        W1c, b1c, W2c, b2c = (t.contiguous() for t in (W1, b1, W2, b2))
        # algorithmic bytes: the volume read once, s / w (saved for backward) and the composited planes written once
        rc = _timed("occlusion_accum_fwd_kernel", 0.0, 4.0 * B * H * W * (F * D + 2 * D + F + 1),
                    lambda: _lib.load().rgbd_occlusion_accum_fwd(_ptr(vol), _ptr(W1c), _ptr(b1c), _ptr(W2c), _ptr(b2c),
                                                                 float(threshold), float(voxel_size), float(near_plane),
                                                                 _ptr(s), _ptr(w), _ptr(feat), _ptr(depth), B, F, D,
                                                                 H * W, _stream()))
        _lib.check(rc, "rgbd_occlusion_accum_fwd")
        ctx.save_for_backward(vol, W1c, b1c, W2c, s, w)
        ctx.voxel_size = float(voxel_size)
        ctx.shapes = (feat.shape, depth.shape)
        ctx.set_materialize_grads(False)      # the weights output is a by-product nobody differentiates: no 9 MB of zeros for it
        return feat, depth, w.reshape(B, 1, D, H, W)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dfeat, ddepth, dw_unused):
        vol, W1, b1, W2, s, w = ctx.saved_tensors
        B, F, D, H, W = vol.shape
        dev = vol.device
        dw_ws = torch.empty(B, D, H * W, dtype=torch.float32, device=dev)
        ds_ws = torch.empty_like(dw_ws)
        dvol = torch.empty_like(vol)
        nparams = OCC_NF * (F + 1) + 2 * OCC_NF + 1
        dparams = torch.empty((nparams + 3) // 4 * 4, dtype=torch.float32, device=dev)
        if dw_unused is not None:
            raise RuntimeError("accumulative_occlusion: the weights output is not differentiable here")
        if dfeat is None and ddepth is None:
            return (None,) * 8
        dfeat = dfeat.contiguous() if dfeat is not None else torch.zeros(ctx.shapes[0], dtype=torch.float32, device=dev)
        ddepth = ddepth.contiguous() if ddepth is not None else torch.zeros(ctx.shapes[1], dtype=torch.float32, device=dev)
        # algorithmic bytes: the volume read, its gradient written, s / w read, the plane gradients read
        rc = _timed("occlusion_accum_bwd_kernel", 0.0, 4.0 * B * H * W * (2 * F * D + 2 * D + F + 1),
                    lambda: _lib.load().rgbd_occlusion_accum_bwd(_ptr(vol), _ptr(W1), _ptr(b1), _ptr(W2), _ptr(s), _ptr(w),
                                                                 _ptr(dfeat), _ptr(ddepth), ctx.voxel_size, _ptr(dw_ws),
                                                                 _ptr(ds_ws), _ptr(dvol), _ptr(dparams), B, F, D, H * W,
                                                                 _stream()))
        _lib.check(rc, "rgbd_occlusion_accum_bwd")
        n1 = OCC_NF * (F + 1)
        return (dvol, dparams[:n1].reshape(OCC_NF, F + 1), dparams[n1:n1 + OCC_NF],
                dparams[n1 + OCC_NF:n1 + 2 * OCC_NF], dparams[n1 + 2 * OCC_NF:n1 + 2 * OCC_NF + 1],
                None, None, None)


def accumulative_occlusion(vol, W1, b1, W2, b2, threshold=4.0, voxel_size=0.0171875, near_plane=float(np.sqrt(3) / 4)):
    """vol (B,F,D,H,W) -> (features (B,F,H,W), depth (B,1,H,W) in camera units, weights (B,1,D,H,W)).
    W1 (4,F+1), b1 (4,), W2 (1,4), b2 (1,): the two 1x1x1 equalized convs of AccumulativeOcclusionNet (the depth
    coordinate is input channel 0)."""
    return _OcclusionAccum.apply(vol, W1, b1, W2.reshape(-1), b2, threshold, voxel_size, near_plane)
