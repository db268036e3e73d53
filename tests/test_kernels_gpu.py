"""Parity of every HIP kernel (called through the C ABI) against the CPU oracle.  Needs an MI355X."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import camera, nets, step, warp_loss

pytestmark = pytest.mark.gpu


def dev():
    return torch.device("cuda:0")


def bf16_round(t):
    return t.to(torch.bfloat16).to(torch.float32)


# ------------------------------------------------------------------------------------------------ warp loss
def _warp_case(b, S, seed, pose_scale=0.3):
    rng = np.random.RandomState(seed)
    img = rng.uniform(-1, 1, (b, 4, S, S)).astype("float32")
    img_rot = rng.uniform(-1, 1, (b, 4, S, S)).astype("float32")
    img[:, 3] = rng.uniform(0.7, 1.3, (b, S, S))
    img_rot[:, 3] = rng.uniform(0.7, 1.3, (b, S, S))
    th = rng.uniform(-pose_scale, pose_scale, (2 * b, 6)).astype("float32")
    th[:, 2] = 0
    th[:, 3:] *= 0.1
    cams = camera.camera_matrices(th)
    return img, img_rot, cams[:b], cams[b:]


def _coef(cam, cam_rot, S):
    K, inv_K, _ = warp_loss.intrinsics(S)
    R, t = warp_loss.relative_pose(cam, cam_rot)
    A, c, A2, c2 = warp_loss.warp_coefficients(K, inv_K, R, t)
    b = len(cam)
    return np.concatenate([A.reshape(b, 9), c, A2.reshape(b, 9), c2], axis=1).astype("float32")


@pytest.mark.parametrize("b,S,occ", [(3, 16, False), (3, 16, True), (2, 32, True), (16, 128, True)])
def test_warp_loss_forward_bit_exact_indices(b, S, occ):
    from rgbd_gan_amd import kernels
    img, img_rot, cam, cam_rot = _warp_case(b, S, seed=10 + S)
    ref = warp_loss.forward_np(img, cam, img_rot, cam_rot, occlusion_aware=occ, lambda_geometric=2.0)
    coef = torch.from_numpy(_coef(cam, cam_rot, S)).to(dev())
    loss, zp, warped, idx = kernels.warp_loss_fwd(torch.from_numpy(img).to(dev()), torch.from_numpy(img_rot).to(dev()),
                                                  coef, 1 if occ else 0, 2.0, debug=True)
    idx = idx.cpu().numpy()
    zp = zp.cpu().numpy()
    warped = warped.cpu().numpy()
    # integer outputs: exact
    np.testing.assert_array_equal(idx[0, :, 0], ref["u0"])
    np.testing.assert_array_equal(idx[0, :, 1], ref["v0"])
    np.testing.assert_array_equal(idx[0, :, 2], ref["v1"])
    np.testing.assert_array_equal(idx[0, :, 3].astype(bool), ref["mask"])
    np.testing.assert_array_equal(idx[1, :, 0], ref["u0_rot"])
    np.testing.assert_array_equal(idx[1, :, 1], ref["v0_rot"])
    np.testing.assert_array_equal(idx[1, :, 3].astype(bool), ref["mask_rot"])
    # fp32 intermediates: bit-exact (same unfused evaluation order)
    np.testing.assert_array_equal(zp[0].view(np.uint32), ref["zp"].view(np.uint32))
    np.testing.assert_array_equal(zp[1].view(np.uint32), ref["zp_rot"].view(np.uint32))
    np.testing.assert_array_equal(warped[0].view(np.uint32), ref["warped"].view(np.uint32))
    np.testing.assert_array_equal(warped[1].view(np.uint32), ref["warped_rot"].view(np.uint32))
    assert ref["mask"].any() and not ref["mask"].all()
    # north_star tolerance: warp-consistency loss within 1e-4 of the reference math
    assert abs(float(loss.item()) - ref["loss"]) < 1e-4 * max(1.0, abs(ref["loss"]))


@pytest.mark.parametrize("b,S,occ", [(3, 16, False), (2, 32, True), (16, 128, True)])     # the last: the benchmark's own size
def test_warp_loss_backward_matches_autograd(b, S, occ):
    from rgbd_gan_amd import kernels
    img, img_rot, cam, cam_rot = _warp_case(b, S, seed=20 + S)
    ti = torch.from_numpy(img).requires_grad_(True)
    tr = torch.from_numpy(img_rot).requires_grad_(True)
    loss, _ = warp_loss.loss_torch(ti, cam, tr, cam_rot, occlusion_aware=occ, lambda_geometric=3.0)
    (loss * 1.7).backward()
    coef = torch.from_numpy(_coef(cam, cam_rot, S)).to(dev())
    go = torch.tensor([1.7], dtype=torch.float32, device=dev())
    gi, gr = kernels.warp_loss_bwd(torch.from_numpy(img).to(dev()), torch.from_numpy(img_rot).to(dev()), coef,
                                   1 if occ else 0, 3.0, 0.0, 0.0, go)
    scale = float(ti.grad.abs().max())
    # tolerance: fp32 with atomics (order-dependent last bits); gradients are O(1/(b*S*S))
    torch.testing.assert_close(gi.cpu(), ti.grad, atol=2e-5 * scale, rtol=1e-4)
    torch.testing.assert_close(gr.cpu(), tr.grad, atol=2e-5 * scale, rtol=1e-4)


def test_warp_loss_zero_for_identical_views_full_size():
    from rgbd_gan_amd import kernels
    S, b = 128, 16
    rng = np.random.RandomState(0)
    img = rng.uniform(-1, 1, (b, 4, S, S)).astype("float32")
    img[:, 3] = 1.0
    cam = camera.camera_matrices(np.zeros((b, 6), "float32"))
    coef = torch.from_numpy(_coef(cam, cam, S)).to(dev())
    t = torch.from_numpy(img).to(dev())
    loss = kernels.warp_loss_fwd(t, t.clone(), coef, 0, 3.0)
    assert float(loss.item()) < 1e-6


# ------------------------------------------------------------------------------------------------ conv engine
CONV_CASES = [
    # B, H, W, Cin, Cout, K, pad, ups, bias, lrelu_ch, resid
    (3, 4, 4, 64, 64, 3, 1, False, False, 0, False),      # M = 48 < one tile
    (2, 8, 8, 64, 128, 3, 1, False, True, 128, False),
    (2, 8, 8, 128, 64, 3, 1, True, True, 64, False),      # fused nearest-2x upsample -> 16x16
    (2, 16, 16, 64, 256, 3, 1, False, True, 128, False),  # c0 || c_sc style: lrelu on the first half only
    (2, 16, 16, 128, 128, 3, 1, False, True, 128, True),  # residual add, then lrelu
    (1, 32, 32, 64, 64, 1, 0, False, True, 64, False),    # 1x1
    (4, 1, 1, 256, 256, 4, 3, False, False, 0, False),    # "full" conv: dgrad of the 4x4 valid conv
    # halo-patch kernel (3x3 pad 1, >= 16x16): several patches per image, several channel slices, both widths
    (3, 32, 32, 256, 128, 3, 1, False, True, 128, False),
    (2, 64, 64, 64, 64, 3, 1, False, True, 64, True),
    (1, 128, 128, 64, 128, 3, 1, False, False, 0, False),
    (2, 16, 16, 256, 128, 3, 1, True, True, 128, False),   # upsample 16 -> 32, Cin 256
    (3, 32, 32, 128, 64, 3, 1, True, True, 64, False),     # upsample 32 -> 64, narrow output
    (2, 8, 8, 64, 256, 3, 1, True, False, 0, False),       # upsample 8 -> 16 (single patch)
]


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_fprop_matches_oracle(case):
    from rgbd_gan_amd import kernels
    B, H, W, Cin, Cout, K, pad, ups, use_bias, lrelu_ch, use_res = case
    g = torch.Generator().manual_seed(hash(case) % 1000)
    x = bf16_round(torch.randn(B, Cin, H, W, generator=g))
    w = torch.randn(Cout, Cin, K, K, generator=g)
    scale = float(np.sqrt(2.0 / (Cin * K * K)))
    bias = torch.randn(Cout, generator=g) if use_bias else None
    wq = bf16_round(w * scale)
    xin = nets.up2(x) if ups else x
    ref = F.conv2d(xin, wq, bias, padding=pad)
    res = None
    if use_res:
        res = bf16_round(torch.randn(ref.shape, generator=g))
        ref = ref + res
    if lrelu_ch:
        ref = torch.cat([F.leaky_relu(ref[:, :lrelu_ch], 0.2), ref[:, lrelu_ch:]], 1)
    wf, _ = kernels.pack_weights(w.to(dev()), scale, True, False)
    xd = x.permute(0, 2, 3, 1).contiguous().to(dev()).to(torch.bfloat16)
    rd = res.permute(0, 2, 3, 1).contiguous().to(dev()).to(torch.bfloat16) if use_res else None
    y = kernels.conv2d_fprop(xd, wf, K, K, pad, bias=bias.to(dev()) if use_bias else None, residual=rd,
                             upsample=ups, lrelu_channels=lrelu_ch)
    got = y.float().cpu().permute(0, 3, 1, 2)
    # tolerance: output is rounded to bf16 (2^-9 relative) on top of fp32 accumulation-order noise
    torch.testing.assert_close(got, ref, atol=2e-2, rtol=1e-2)


@pytest.mark.parametrize("B,H,W,Cin,Cout,K,pad", [(2, 8, 8, 64, 128, 3, 1), (2, 16, 16, 128, 64, 3, 1),
                                                    (3, 4, 4, 64, 64, 1, 0)])
def test_conv_dgrad_matches_autograd(B, H, W, Cin, Cout, K, pad):
    from rgbd_gan_amd import kernels
    g = torch.Generator().manual_seed(7)
    scale = float(np.sqrt(2.0 / (Cin * K * K)))
    w = torch.randn(Cout, Cin, K, K, generator=g)
    wq = bf16_round(w * scale)
    x = torch.randn(B, Cin, H, W, generator=g, requires_grad=True)
    dy = bf16_round(torch.randn(B, Cout, H, W, generator=g))
    F.conv2d(x, wq, None, padding=pad).backward(dy)
    _, wd = kernels.pack_weights(w.to(dev()), scale, False, True)
    dyd = dy.permute(0, 2, 3, 1).contiguous().to(dev()).to(torch.bfloat16)
    dx = kernels.conv2d_fprop(dyd, wd, K, K, K - 1 - pad)
    torch.testing.assert_close(dx.float().cpu().permute(0, 3, 1, 2), x.grad, atol=2e-2, rtol=1e-2)


@pytest.mark.parametrize("B,H,Cin,Cout", [(2, 16, 64, 128), (3, 32, 128, 64), (2, 8, 64, 64)])
def test_conv_with_pooled_second_output(B, H, Cin, Cout):
    """downscale2x(leaky_relu(c1(h) + c_sc(x))) (net.py:413-418): the pooled tensor out of the conv epilogue equals the
    separate pooling pass over the stored activations, and the activations are unchanged."""
    from rgbd_gan_amd import kernels
    g = torch.Generator().manual_seed(23)
    x = torch.randn(B, H, H, Cin, generator=g).to(dev()).to(torch.bfloat16)
    res = torch.randn(B, H, H, Cout, generator=g).to(dev()).to(torch.bfloat16)
    bias = torch.randn(Cout, generator=g).to(dev())
    wf, _ = kernels.pack_weights(torch.randn(Cout, Cin, 3, 3, generator=g).to(dev()), float(np.sqrt(2.0 / (Cin * 9))))
    y_ref = kernels.conv2d_fprop(x, wf, 3, 3, 1, bias=bias, residual=res, lrelu_channels=Cout)
    y, yp = kernels.conv2d_fprop(x, wf, 3, 3, 1, bias=bias, residual=res, lrelu_channels=Cout, avg_pool2=True)
    # small images: y_ref comes from the split-K gather kernel (other summation order), so one bf16 ulp apart at most
    torch.testing.assert_close(y.float(), y_ref.float(), atol=2 ** -7 * float(y_ref.float().abs().max()), rtol=0)
    pooled = y.float().view(B, H // 2, 2, H // 2, 2, Cout).mean(dim=(2, 4))
    torch.testing.assert_close(yp.float(), pooled.to(torch.bfloat16).float(), atol=1e-2, rtol=1e-2)
    torch.testing.assert_close(yp.float(), kernels.pool2_masked(y).float(), atol=1e-2, rtol=1e-2)


@pytest.mark.parametrize("B,H,Cin,Cout", [(2, 16, 128, 64), (3, 32, 64, 128), (2, 64, 128, 128), (2, 8, 64, 64)])
def test_conv_dgrad_through_the_upsampling(B, H, Cin, Cout):
    """Input gradient of c0(upscale2x(h)) (net.py:148-150): the 2x2 sums are taken in the epilogue of the halo-patch
    kernel (8x8: second pass).  Oracle: autograd through F.interpolate(nearest) + conv2d."""
    from rgbd_gan_amd import kernels
    g = torch.Generator().manual_seed(17)
    scale = float(np.sqrt(2.0 / (Cin * 9)))
    w = torch.randn(Cout, Cin, 3, 3, generator=g)
    wq = bf16_round(w * scale)
    x = torch.randn(B, Cin, H // 2, H // 2, generator=g, requires_grad=True)
    dy = bf16_round(torch.randn(B, Cout, H, H, generator=g))
    F.conv2d(F.interpolate(x, scale_factor=2, mode="nearest"), wq, None, padding=1).backward(dy)
    _, wd = kernels.pack_weights(w.to(dev()), scale, False, True)
    dyd = dy.permute(0, 2, 3, 1).contiguous().to(dev()).to(torch.bfloat16)
    dx = kernels.conv2d_dgrad(dyd, wd, 3, 1, sum_pool2=True)
    assert tuple(dx.shape) == (B, H // 2, H // 2, Cin)
    torch.testing.assert_close(dx.float().cpu().permute(0, 3, 1, 2), x.grad, atol=4e-2, rtol=1e-2)
    # against the two-pass form: only the bf16 rounding of the full-resolution intermediate differs
    full = kernels.conv2d_dgrad(dyd, wd, 3, 1).float()
    two_pass = full.view(B, H // 2, 2, H // 2, 2, Cin).sum(dim=(2, 4))
    torch.testing.assert_close(dx.float(), two_pass, atol=4e-2, rtol=2e-2)


@pytest.mark.parametrize("B,H,W,Cin,Cout,K", [(2, 4, 4, 64, 64, 3), (3, 8, 8, 64, 128, 3), (2, 16, 16, 128, 64, 3),
                                               (2, 32, 32, 64, 64, 3), (1, 64, 64, 64, 64, 3), (2, 16, 16, 64, 64, 1)])
def test_conv_wgrad_matches_autograd(B, H, W, Cin, Cout, K):
    from rgbd_gan_amd import kernels
    g = torch.Generator().manual_seed(11)
    x = bf16_round(torch.randn(B, Cin, H, W, generator=g))
    dy = bf16_round(torch.randn(B, Cout, H, W, generator=g))
    w = torch.zeros(Cout, Cin, K, K, requires_grad=True)
    F.conv2d(x, w, None, padding=(K - 1) // 2).backward(dy)
    xd = x.permute(0, 2, 3, 1).contiguous().to(dev()).to(torch.bfloat16)
    dyd = dy.permute(0, 2, 3, 1).contiguous().to(dev()).to(torch.bfloat16)
    dw = kernels.conv2d_wgrad(xd, dyd, K, 0.5)
    ref = w.grad * 0.5
    tol = 1e-3 * float(ref.abs().max())   # fp32 accumulate of exact bf16 products; only summation order differs
    torch.testing.assert_close(dw.cpu(), ref, atol=tol, rtol=1e-4)


@pytest.mark.parametrize("H,Cin,Cout,ups", [(128, 64, 128, False), (128, 128, 128, False), (64, 256, 256, False),
                                             (128, 128, 64, True), (32, 256, 256, True)])
def test_conv_engine_adjoint_identities_at_benchmark_sizes(H, Cin, Cout, ups):
    """At the benchmark's layer shapes (B = 32, too large for the CPU oracle) the three kernels of a conv check each
    other through identities that hold for any correct implementation:
        <fprop(x; W), dy> = <x, dgrad(dy; W)> = <W, wgrad(x, dy)> / scale     (one bilinear form, three evaluations)
    with the nearest-2x upsample folded in for the generator's c0 (fprop reads through it, dgrad returns 2x2 sums,
    wgrad reads x through the index map).  Tolerance: bf16 rounding of the stored tensors, averaged over >1e7 terms."""
    from rgbd_gan_amd import kernels
    B = 32
    g = torch.Generator().manual_seed(41)
    hs = H // 2 if ups else H
    scale = float(np.sqrt(2.0 / (Cin * 9)))
    x = torch.randn(B, hs, hs, Cin, generator=g).to(dev()).to(torch.bfloat16)
    dy = torch.randn(B, H, H, Cout, generator=g).to(dev()).to(torch.bfloat16)
    w = torch.randn(Cout, Cin, 3, 3, generator=g).to(dev())
    wf, wd = kernels.pack_weights(w, scale)
    y = kernels.conv2d_fprop(x, wf, 3, 3, 1, upsample=ups)
    dx = kernels.conv2d_dgrad(dy, wd, 3, 1, sum_pool2=ups)
    dw = kernels.conv2d_wgrad(x, dy, 3, scale, upsample=ups)
    assert tuple(dx.shape) == tuple(x.shape)
    a = float((y.double() * dy.double()).sum())
    b = float((x.double() * dx.double()).sum())
    # wgrad returns scale * sum dy (x) x for the MASTER weight; the packed weights are bf16(scale * W)
    wq = wf.float().permute(1, 2, 0).reshape(Cout, Cin, 3, 3)            # [tap][co][ci] -> (co, ci, kh, kw), = bf16(scale W)
    c = float((wq.double() * dw.double()).sum()) / scale
    ref = max(abs(a), abs(b), abs(c))
    norm = float(y.float().norm() * dy.float().norm())                    # Cauchy-Schwarz scale of the form
    assert abs(a - b) < 2e-4 * norm and abs(a - c) < 2e-4 * norm, (a, b, c, norm)
    assert ref > 0
    # linearity in x (bf16 outputs: one ulp of the larger side)
    x2 = torch.randn(B, hs, hs, Cin, generator=g).to(dev()).to(torch.bfloat16)
    y2 = kernels.conv2d_fprop(x2, wf, 3, 3, 1, upsample=ups)
    ysum = kernels.conv2d_fprop((x.float() + x2.float()).to(torch.bfloat16), wf, 3, 3, 1, upsample=ups)
    err = (ysum.float() - (y.float() + y2.float())).abs().max()
    assert float(err) < 2 ** -5 * float(ysum.float().abs().max())       # x1 + x2 is itself rounded to bf16


def test_gan_logit_heads_match_the_loss_functions():
    """loss_functions.py:15-28 on the logits, values and derivatives, incl. logits far in both tails."""
    from rgbd_gan_amd import kernels
    g = torch.Generator().manual_seed(31)
    y = torch.cat([torch.randn(29, 1, generator=g) * 4, torch.tensor([[-90.0], [75.0], [0.0]])])
    yl = y.clone().requires_grad_(True)
    l_neg = torch.sum(F.softplus(-yl)) / yl.numel()
    l_pos = torch.sum(F.softplus(yl)) / yl.numel()
    g_neg, = torch.autograd.grad(l_neg, yl)
    g_pos, = torch.autograd.grad(l_pos, yl)
    losses, sn, sp, ratio = kernels.gan_logit_heads(y.to(dev()))
    torch.testing.assert_close(losses.cpu(), torch.stack([l_neg, l_pos]).detach(), atol=1e-6, rtol=1e-5)
    torch.testing.assert_close(sn.cpu(), g_neg, atol=1e-9, rtol=1e-5)
    keep = y.reshape(-1) > -60                                   # below the clamp seed_pos is < 1e-26/n either way
    torch.testing.assert_close(sp.cpu()[keep], g_pos[keep], atol=1e-12, rtol=1e-5)
    assert float(sp.cpu()[~keep].abs().max()) < 1e-26
    torch.testing.assert_close(ratio.cpu(), -torch.exp(-y.clamp(min=-60.0)), atol=0, rtol=1e-5)
    assert torch.isfinite(ratio).all()


@pytest.mark.parametrize("sign,gamma", [(-1.0, 0.0), (1.0, 0.0), (-1.0, 2.0), (-1.0, 0.5)])
def test_softplus_mean_matches_the_loss_functions(sign, gamma):
    """rgbd_softplus_mean against loss_func_dcgan_gen (focal and plain) / the two terms of loss_func_dcgan_dis
    (loss_functions.py:15-31) in torch fp32 on the CPU: value 1e-6, derivative 1e-5 relative, logits far in both tails;
    through the autograd wrapper the derivative is scaled by the incoming gradient."""
    from rgbd_gan_amd import functional as Fn, kernels
    from rgbd_gan_amd.common.loss_functions import loss_func_dcgan_dis, loss_func_dcgan_gen
    g = torch.Generator().manual_seed(37)
    y = torch.cat([torch.randn(17, 1, generator=g) * 3, torch.tensor([[-80.0], [60.0], [0.0]])])
    yl = y.clone().requires_grad_(True)
    if sign < 0:
        ref = loss_func_dcgan_gen(yl, gamma)
    else:
        ref = loss_func_dcgan_dis(yl, torch.zeros(1, 1)) - float(np.log(2.0))          # softplus(-0) = log 2
    gref, = torch.autograd.grad(ref, yl)
    loss, dy = kernels.softplus_mean(y.to(dev()), sign, gamma)
    torch.testing.assert_close(loss.cpu().reshape(()), ref.detach(), atol=1e-6, rtol=1e-5)
    torch.testing.assert_close(dy.cpu(), gref, atol=1e-9, rtol=2e-5)
    yd = y.to(dev()).requires_grad_(True)
    (3.0 * Fn.softplus_mean(yd, sign, gamma)).backward()
    torch.testing.assert_close(yd.grad.cpu(), 3.0 * gref, atol=1e-9, rtol=2e-5)


def test_conv_wgrad_batch_matches_single_calls():
    """Collected weight gradients (functional.deferred_wgrads): the 3x3 layers on images >= 8x16 share one partial-sum
    launch (workgroups dealt out over the layers by work), the 3x3 layers on 4x4 / 8x8 images share another, 1x1 convs get
    a launch each, one multi-descriptor slab
    reduction finishes all: the same sums as the one-call-per-layer path, split differently over workgroups (fp32
    summation order differs)."""
    from rgbd_gan_amd import kernels
    g = torch.Generator().manual_seed(29)
    shapes = [(2, 8, 64, 128, 3, False), (2, 16, 128, 64, 3, True), (3, 4, 64, 64, 1, False), (1, 32, 64, 64, 3, False),
              (2, 64, 64, 128, 3, True), (1, 128, 128, 64, 3, False), (4, 16, 256, 256, 3, False), (5, 4, 128, 64, 3, False),
              (2, 8, 64, 64, 3, True)] * 4
    items, refs = [], []
    for i, (B, H, Cin, Cout, K, ups) in enumerate(shapes):       # 36 items: more than one reduction / multi launch
        hs = H // 2 if ups else H
        x = torch.randn(B, hs, hs, Cin, generator=g).to(dev()).to(torch.bfloat16)
        dy = torch.randn(B, H, H, Cout, generator=g).to(dev()).to(torch.bfloat16)
        init = torch.randn(Cout, Cin, K, K, generator=g).to(dev())
        ref = init.clone()
        kernels.conv2d_wgrad(x, dy, K, 0.5 + 0.01 * i, out=ref, accumulate=True, upsample=ups)
        items.append((x, dy, init, K, 0.5 + 0.01 * i, ups))
        refs.append(ref)
    kernels.conv2d_wgrad_batch(items)
    for (_, _, got, _, _, _), ref in zip(items, refs):
        torch.testing.assert_close(got, ref, rtol=1e-5, atol=2e-5 * float(ref.abs().max()))


@pytest.mark.parametrize("budget", [8, 40, 100, 224])
def test_compute_unit_budget_changes_grids_not_results(budget):
    """The conv entry points' `cus` argument (kernels.cu_budget: per stream, host side) / the weight-gradient plan's workgroup
    count (RGBDUpdater gives its side stream's chip-filling launches fewer compute units): the 3x3 kernel walks the same tiles with the same arithmetic on fewer persistent
    workgroups -- every form bit-identical; the weight gradients are the same sums split over a different number of slabs
    (fp32 summation order)."""
    from rgbd_gan_amd import kernels
    g = torch.Generator().manual_seed(budget)
    B, H, Cin, Cout = 5, 64, 128, 256                        # 80 pixel tiles x 2 channel tiles: 2 rounds on 40 workgroups
    x = torch.randn(B, H, H, Cin, generator=g).to(dev()).to(torch.bfloat16)
    w = torch.randn(Cout, Cin, 3, 3, generator=g).to(dev())
    bias = torch.randn(Cout, generator=g).to(dev())
    act = torch.randn(B, H, H, Cout, generator=g).to(dev()).to(torch.bfloat16)
    wf, wd = kernels.pack_weights(w, 0.03)
    dy = torch.randn(B, H, H, Cout, generator=g).to(dev()).to(torch.bfloat16)

    def run():
        y = kernels.conv2d_fprop(x, wf, 3, 3, 1, bias=bias, lrelu_channels=Cout)
        ya = kernels.conv3x3_actgrad(x, wf, act)
        ys = kernels.conv2d_fprop_stats(x, wf, bias=bias, lrelu_channels=Cout)
        items = [(x, dy, torch.zeros(Cout, Cin, 3, 3, device=dev()), 3, 1.0, False),
                 (x[:, ::2, ::2].contiguous(), dy, torch.zeros(Cout, Cin, 3, 3, device=dev()), 3, 0.5, True)]
        kernels.conv2d_wgrad_batch(items)
        return [y, ya] + [t for t in ys if torch.is_tensor(t)], [it[2] for it in items]

    ref_y, ref_w = run()
    with kernels.cu_budget(budget), kernels.wgrad_workgroups(budget):
        got_y, got_w = run()
    assert kernels._cus() == 0 and not kernels._STREAM_CUS      # the context put the default back
    other = torch.cuda.Stream()
    with kernels.cu_budget(budget):                             # a budget belongs to the stream it was set on ...
        assert kernels._cus() == budget
        with torch.cuda.stream(other):
            assert kernels._cus() == 0                          # ... another stream's launches do not see it
    for a, b in zip(ref_y, got_y):
        if a.dtype == torch.bfloat16:
            assert torch.equal(a.view(torch.int16), b.view(torch.int16))
        else:               # the instance-norm statistics: one fixed-point add per TILE (integer adds are associative), so the
            assert torch.equal(a, b)                                # tile-to-workgroup assignment cannot change them
    for a, b in zip(ref_w, got_w):
        torch.testing.assert_close(a, b, rtol=1e-5, atol=2e-5 * float(a.abs().max()))


@pytest.mark.parametrize("B,H,W,Cin,Cout", [(2, 4, 4, 64, 64), (3, 8, 8, 128, 64), (2, 32, 32, 64, 128), (1, 64, 64, 64, 64)])
def test_conv_wgrad_through_the_upsampling(B, H, W, Cin, Cout):
    """c0(upscale2x(h)) of a synthesis block (net.py:148-150, rescale.py:4-5): the weight gradient reads the
    half-resolution operand through the index map.  Oracle: autograd through F.interpolate(nearest) + conv2d."""
    from rgbd_gan_amd import kernels
    g = torch.Generator().manual_seed(12)
    x = bf16_round(torch.randn(B, Cin, H // 2, W // 2, generator=g))
    dy = bf16_round(torch.randn(B, Cout, H, W, generator=g))
    w = torch.zeros(Cout, Cin, 3, 3, requires_grad=True)
    F.conv2d(F.interpolate(x, scale_factor=2, mode="nearest"), w, None, padding=1).backward(dy)
    xd = x.permute(0, 2, 3, 1).contiguous().to(dev()).to(torch.bfloat16)
    dyd = dy.permute(0, 2, 3, 1).contiguous().to(dev()).to(torch.bfloat16)
    dw = kernels.conv2d_wgrad(xd, dyd, 3, 0.25, upsample=True)
    ref = w.grad * 0.25
    tol = 1e-3 * float(ref.abs().max())
    torch.testing.assert_close(dw.cpu(), ref, atol=tol, rtol=1e-4)
    # identical (same summation order) to the gradient on the materialised upsampled operand
    xe = xd.repeat_interleave(2, dim=1).repeat_interleave(2, dim=2).contiguous()
    assert torch.equal(dw, kernels.conv2d_wgrad(xe, dyd, 3, 0.25))


# ------------------------------------------------------------------------------------------------ AdaIN
@pytest.mark.parametrize("B,H,C", [(2, 4, 64), (3, 16, 128), (2, 64, 64)])
def test_adain_forward_backward(B, H, C):
    from rgbd_gan_amd import kernels
    g = torch.Generator().manual_seed(3)
    x = bf16_round(torch.randn(B, C, H, H, generator=g) * 1.5 + 0.3).requires_grad_(True)
    s = (torch.randn(B, C, generator=g) + 1).requires_grad_(True)
    t = torch.randn(B, C, generator=g).requires_grad_(True)
    dy = bf16_round(torch.randn(B, C, H, H, generator=g))
    y = nets.adain(x, s, t)
    y.backward(dy)
    xd = x.detach().permute(0, 2, 3, 1).contiguous().to(dev()).to(torch.bfloat16)
    yd, mean, rstd = kernels.adain_fwd(xd, s.detach().to(dev()), t.detach().to(dev()))
    torch.testing.assert_close(yd.float().cpu().permute(0, 3, 1, 2), y.detach(), atol=3e-2, rtol=1e-2)
    dyd = dy.permute(0, 2, 3, 1).contiguous().to(dev()).to(torch.bfloat16)
    dx, ds, dt = kernels.adain_bwd(xd, dyd, s.detach().to(dev()), mean, rstd)
    torch.testing.assert_close(dx.float().cpu().permute(0, 3, 1, 2), x.grad, atol=3e-2, rtol=2e-2)
    torch.testing.assert_close(ds.cpu(), s.grad, atol=1e-3 * H * H, rtol=1e-3)
    torch.testing.assert_close(dt.cpu(), t.grad, atol=1e-3 * H * H, rtol=1e-3)


@pytest.mark.parametrize("B,H,C", [(2, 8, 64), (3, 32, 128), (2, 16, 256), (2, 64, 128), (1, 96, 64)])
def test_adain_backward_with_fused_activation_gradient(B, H, C):
    """conv -> bias -> lrelu -> AdaIN (net.py:150-153): the AdaIN input IS the activation output, so rgbd_adain_bwd can
    apply the slope mask and take the bias sums in the same pass.  Oracle: autograd through lrelu + oracle AdaIN."""
    from rgbd_gan_amd import kernels
    g = torch.Generator().manual_seed(13)
    z = (torch.randn(B, C, H, H, generator=g) * 1.5).requires_grad_(True)       # pre-activation
    s = (torch.randn(B, C, generator=g) + 1)
    t = torch.randn(B, C, generator=g)
    dy = bf16_round(torch.randn(B, C, H, H, generator=g))
    a = bf16_round(nets.lrelu(z).detach())                                       # what the conv epilogue stores
    a_leaf = a.clone().requires_grad_(True)
    nets.adain(a_leaf, s, t).backward(dy)
    dz_ref = a_leaf.grad * torch.where(a > 0, 1.0, 0.2)
    ad = a.permute(0, 2, 3, 1).contiguous().to(dev()).to(torch.bfloat16)
    dyd = dy.permute(0, 2, 3, 1).contiguous().to(dev()).to(torch.bfloat16)
    _, mean, rstd = kernels.adain_fwd(ad, s.to(dev()), t.to(dev()))
    bg = torch.full((C,), 2.0, device=dev())
    dz, ds, dt = kernels.adain_bwd(ad, dyd, s.to(dev()), mean, rstd, lrelu_slope=0.2, bias_grad=bg)
    torch.testing.assert_close(dz.float().cpu().permute(0, 3, 1, 2), dz_ref, atol=3e-2, rtol=2e-2)
    torch.testing.assert_close(bg.cpu() - 2.0, dz.float().cpu().reshape(-1, C).sum(0), atol=2e-3 * H, rtol=1e-3)
    # the unfused pair of passes gives the same tensors up to the extra bf16 rounding between them
    dx, ds2, dt2 = kernels.adain_bwd(ad, dyd, s.to(dev()), mean, rstd)
    dz2 = kernels.lrelu_bwd(dx, ad, C)
    torch.testing.assert_close(dz.float(), dz2.float(), atol=2e-2, rtol=2e-2)
    assert torch.equal(ds, ds2) and torch.equal(dt, dt2)


# ------------------------------------------------------------------------------------------------ Adam + clip
@pytest.mark.parametrize("gnorm_scale", [0.01, 30.0])
def test_adam_clip_matches_chainer_restatement(gnorm_scale):
    from rgbd_gan_amd import kernels
    g = torch.Generator().manual_seed(5)
    n1, n2 = 1000, 333
    p = torch.randn(n1 + n2, generator=g)
    params = {"a": p[:n1].clone().requires_grad_(True), "b": p[n1:].clone().requires_grad_(True)}
    opt = step.ChainerAdam(params, alpha=1e-3, beta1=0.0, beta2=0.999, alpha_override={"b": 1e-5})
    pd = p.clone().to(dev())
    md, vd = torch.zeros_like(pd), torch.zeros_like(pd)
    ws = torch.empty(1024 + 8, device=dev())
    norm = torch.empty(1, device=dev())
    step_dev = torch.zeros(1, dtype=torch.int32, device=dev())
    for t in range(1, 4):
        grads = torch.randn(n1 + n2, generator=g) * gnorm_scale
        params["a"].grad = grads[:n1].clone()
        params["b"].grad = grads[n1:].clone()
        ref_norm = opt.update()
        kernels.adam_clip_multi(pd, (grads * 2.0).to(dev()), md, vd, [0, n1, n1 + n2], [1e-3, 1e-5], 0.0, 0.999,
                                1e-8, 5.0, 0.5, step_dev, ws, norm)
        assert int(step_dev.item()) == t
        assert abs(float(norm.item()) - ref_norm) < 1e-4 * ref_norm
        ref = torch.cat([params["a"].detach(), params["b"].detach()])
        torch.testing.assert_close(pd.cpu(), ref, atol=1e-6, rtol=1e-5)


# ------------------------------------------------------------------------------------------------ fused elementwise / 1x1
def test_lrelu_bwd_and_colsum():
    from rgbd_gan_amd import kernels
    g = torch.Generator().manual_seed(2)
    y = bf16_round(torch.randn(3, 8, 8, 128, generator=g))
    dy = bf16_round(torch.randn(3, 8, 8, 128, generator=g))
    dz = kernels.lrelu_bwd(dy.to(dev()).to(torch.bfloat16), y.to(dev()).to(torch.bfloat16), 64)
    ref = dy.clone()
    ref[..., :64] = torch.where(y[..., :64] > 0, dy[..., :64], dy[..., :64] * 0.2)
    torch.testing.assert_close(dz.float().cpu(), bf16_round(ref), atol=1e-6, rtol=1e-6)
    cs = kernels.colsum(dy.to(dev()).to(torch.bfloat16))
    torch.testing.assert_close(cs.cpu(), dy.reshape(-1, 128).sum(0), atol=1e-3, rtol=1e-4)
    # fused variant: same dz, column sums of dz ADDED to the given buffer
    bg = torch.full((128,), 2.0, device=dev())
    dz2 = kernels.lrelu_bwd(dy.to(dev()).to(torch.bfloat16), y.to(dev()).to(torch.bfloat16), 64, bias_grad=bg)
    assert torch.equal(dz2, dz)
    torch.testing.assert_close(bg.cpu(), dz.float().cpu().reshape(-1, 128).sum(0) + 2.0, atol=1e-3, rtol=1e-4)


@pytest.mark.parametrize("shape,act", [((5, 168, 160, 64), 64), ((3, 200, 232, 128), 64), ((2, 100, 180, 64), 64)])
def test_column_sum_passes_on_large_ragged_tensors(shape, act):
    """The strip plans of the passes that carry a column sum (elementwise.hip:plan_colsum): 1024-thread blocks on 2048-row
    strips from 131072 rows, 256-row strips from 32768, with row counts that are multiples of neither -- values bit-exact,
    sums to fp32 accumulation accuracy, per-sample weights included."""
    from rgbd_gan_amd import kernels
    g = torch.Generator().manual_seed(12)
    B, H, W, C = shape
    y = bf16_round(torch.randn(*shape, generator=g))
    dy = bf16_round(torch.randn(*shape, generator=g))
    yd, dyd = y.to(dev()).to(torch.bfloat16), dy.to(dev()).to(torch.bfloat16)
    ref = dy.clone()
    ref[..., :act] = torch.where(y[..., :act] > 0, dy[..., :act], dy[..., :act] * 0.2)
    ref = bf16_round(ref)
    tol = dict(atol=2e-4 * np.sqrt(B * H * W), rtol=1e-4)
    bg = torch.full((C,), 1.0, device=dev())
    dz = kernels.lrelu_bwd(dyd, yd, act, bias_grad=bg)
    assert torch.equal(dz.float().cpu(), ref)
    torch.testing.assert_close(bg.cpu() - 1.0, ref.double().reshape(-1, C).sum(0).float(), **tol)
    rs = torch.randn(B, generator=g)
    bg = torch.zeros(C, device=dev())
    kernels.lrelu_bwd(dyd, yd, act, bias_grad=bg, row_scale=rs.to(dev()))
    want = (ref.double() * rs.double().reshape(B, 1, 1, 1)).reshape(-1, C).sum(0).float()
    torch.testing.assert_close(bg.cpu(), want, **tol)
    cs = kernels.colsum(dyd, row_scale=rs.to(dev()), rows_per_sample=H * W)
    torch.testing.assert_close(cs.cpu(), (dy.double() * rs.double().reshape(B, 1, 1, 1)).reshape(-1, C).sum(0).float(), **tol)
    # unpool form: dp at half resolution
    dp = bf16_round(torch.randn(B, H // 2, W // 2, C, generator=g))
    up = dp.repeat_interleave(2, dim=1).repeat_interleave(2, dim=2)
    mask = torch.where(y > 0, torch.ones_like(y), torch.full_like(y, 0.2))
    bg, bg2 = torch.zeros(C, device=dev()), torch.zeros(C, device=dev())
    dzu = kernels.unpool2_lrelu_bwd(dp.to(dev()).to(torch.bfloat16), yd, shape, bias_grad=bg, bias_grad2=bg2)
    refu = bf16_round(0.25 * up * mask)
    assert torch.equal(dzu.float().cpu(), refu)
    torch.testing.assert_close(bg.cpu(), refu.double().reshape(-1, C).sum(0).float(), **tol)
    torch.testing.assert_close(bg2.cpu(), bg.cpu(), **tol)         # same sums, atomics in their own arrival order


@pytest.mark.parametrize("B,H,C,KP", [(2, 16, 64, 3), (3, 8, 256, 3), (2, 32, 128, 4)])
def test_plane_kernels_match_oracle(B, H, C, KP):
    from rgbd_gan_amd import kernels
    g = torch.Generator().manual_seed(9)
    x = torch.randn(B, KP, H, H, generator=g)
    w = torch.randn(C, KP, generator=g)
    b = torch.randn(C, generator=g)
    s = 0.7
    ref = F.leaky_relu(F.conv2d(x * s, w.reshape(C, KP, 1, 1), b), 0.2)
    y = kernels.from_planes(x.to(dev()), w.to(dev()), b.to(dev()), s, True)
    torch.testing.assert_close(y.float().cpu().permute(0, 3, 1, 2), ref, atol=2e-2, rtol=1e-2)
    # to_planes: h NHWC bf16 -> planes
    h = bf16_round(torch.randn(B, C, H, H, generator=g))
    w2 = torch.randn(KP, C, generator=g)
    b2 = torch.randn(KP, generator=g)
    ref2 = F.conv2d(h * s, w2.reshape(KP, C, 1, 1), b2)
    hd = h.permute(0, 2, 3, 1).contiguous().to(dev()).to(torch.bfloat16)
    out = kernels.to_planes(hd, w2.to(dev()), b2.to(dev()), s)
    torch.testing.assert_close(out.cpu(), ref2, atol=1e-3 * C ** 0.5, rtol=1e-4)
    # planes_outer: o[k][c] = sum planes[b,k,p] * t[b,p,c]
    o, ts = kernels.planes_outer(hd, x.to(dev()), True)
    ref_o = torch.einsum("bkhw,bchw->kc", x, h)
    torch.testing.assert_close(o.cpu(), ref_o, atol=1e-3 * float(ref_o.abs().max()), rtol=1e-4)
    torch.testing.assert_close(ts.cpu(), h.sum(dim=(0, 2, 3)), atol=1e-3 * float(h.sum(dim=(0, 2, 3)).abs().max()) + 1e-3,
                               rtol=1e-4)


@pytest.mark.parametrize("B,H,Cin,Cout,ups", [(2, 128, 64, 128, False), (2, 64, 128, 64, True), (4, 32, 256, 256, False)])
def test_patch_kernel_agrees_with_gather_kernel(B, H, Cin, Cout, ups):
    """Two independent implementations of the same convolution (halo-patch vs generic gather) at layer sizes."""
    from rgbd_gan_amd import _lib, kernels
    g = torch.Generator().manual_seed(21)
    x = torch.randn(B, H, H, Cin, generator=g).to(dev()).to(torch.bfloat16)
    w = torch.randn(Cout, Cin, 3, 3, generator=g).to(dev())
    bias = torch.randn(Cout, generator=g).to(dev())
    wf, _ = kernels.pack_weights(w, float(np.sqrt(2.0 / (Cin * 9))), True, False)
    y_patch = kernels.conv2d_fprop(x, wf, 3, 3, 1, bias=bias, upsample=ups, lrelu_channels=Cout)
    with _lib.debug_library() as lib:                  # the planner switch lives in the debug library only
        lib.rgbd_debug_force_gather_kernel(1)
        y_gather = kernels.conv2d_fprop(x, wf, 3, 3, 1, bias=bias, upsample=ups, lrelu_channels=Cout)
    # same bf16 inputs, fp32 accumulation in a different order, one bf16 rounding at the end
    torch.testing.assert_close(y_patch.float(), y_gather.float(), atol=2e-2, rtol=8e-3)


@pytest.mark.parametrize("B,H,Cin,Cout,res", [(32, 8, 256, 256, False), (32, 4, 256, 256, True), (2, 8, 64, 128, True),
                                             (8, 4, 128, 64, False), (32, 8, 256, 512, False)])
def test_small_image_kernel_matches_gather_kernel_and_fp32_conv(B, H, Cin, Cout, res):
    """conv3x3_small_kernel (4x4 / 8x8 images: whole images + nine weight tiles staged at once, one 64-channel input slice
    per workgroup, fp32 partials per slice) against the generic gather kernel on the same operands, against an fp32
    convolution of the same bf16 operands, and against itself over repeated launches; dgrad goes through the same kernel."""
    from rgbd_gan_amd import _lib, kernels
    g = torch.Generator().manual_seed(B + H + Cin + Cout)
    x = torch.randn(B, H, H, Cin, generator=g).to(dev()).to(torch.bfloat16)
    w = torch.randn(Cout, Cin, 3, 3, generator=g).to(dev())
    bias = torch.randn(Cout, generator=g).to(dev())
    r = torch.randn(B, H, H, Cout, generator=g).to(dev()).to(torch.bfloat16) if res else None
    scale = float(np.sqrt(2.0 / (Cin * 9)))
    wf, wd = kernels.pack_weights(w, scale)
    lib = _lib.load()
    y = kernels.conv2d_fprop(x, wf, 3, 3, 1, bias=bias, residual=r, lrelu_channels=Cout)
    assert lib.rgbd_last_conv_kernel().decode() == f"conv3x3_small_kernel<{H}>"
    for _ in range(5):
        assert torch.equal(kernels.conv2d_fprop(x, wf, 3, 3, 1, bias=bias, residual=r, lrelu_channels=Cout), y)
    with _lib.debug_library() as dlib:
        dlib.rgbd_debug_force_gather_kernel(1)
        y_gather = kernels.conv2d_fprop(x, wf, 3, 3, 1, bias=bias, residual=r, lrelu_channels=Cout)
        assert dlib.rgbd_last_conv_kernel().decode().startswith("conv_fprop_kernel")
    torch.testing.assert_close(y.float(), y_gather.float(), atol=2e-2, rtol=8e-3)
    wb = (w * scale).to(torch.bfloat16).float()
    ref = F.conv2d(x.float().permute(0, 3, 1, 2), wb, bias, padding=1)
    if res:
        ref = ref + r.float().permute(0, 3, 1, 2)
    ref = F.leaky_relu(ref, 0.2).permute(0, 2, 3, 1)
    torch.testing.assert_close(y.float(), ref, atol=2e-2, rtol=8e-3)
    # dgrad = the same kernel on the flipped / transposed image
    dy = torch.randn(B, H, H, Cout, generator=g).to(dev()).to(torch.bfloat16)
    dx = kernels.conv2d_dgrad(dy, wd, 3, 1)
    assert lib.rgbd_last_conv_kernel().decode() == f"conv3x3_small_kernel<{H}>"
    ref_dx = F.conv_transpose2d(dy.float().permute(0, 3, 1, 2), wb, padding=1).permute(0, 2, 3, 1)
    torch.testing.assert_close(dx.float(), ref_dx, atol=2e-2 * float(ref_dx.abs().max()) / 4, rtol=1e-2)


@pytest.mark.parametrize("B,S,Cin,Cout,res,weighted", [(4, 32, 128, 128, False, False), (2, 64, 64, 128, True, True),
                                                      (8, 16, 256, 256, False, True), (3, 32, 128, 64, True, False),
                                                      (32, 128, 64, 64, False, True)])
def test_conv3x3_actgrad_matches_conv_then_activation_gradient(B, S, Cin, Cout, res, weighted):
    """rgbd_conv3x3_actgrad_bf16 (3x3 conv + residual, times lrelu'(.) of a given activation output, weighted column sums:
    one epilogue) against the two launches it replaces -- the conv, then rgbd_lrelu_bwd with its fused bias gradient --
    and against fp32 torch on the same bf16 operands; through the dgrad image too (how the discriminator's backward uses it)."""
    from rgbd_gan_amd import kernels
    assert kernels.conv3x3_actgrad_supported(B, S, S, Cin, Cout)
    assert not kernels.conv3x3_actgrad_supported(B, 8, 8, Cin, Cout)
    g = torch.Generator().manual_seed(B + S + Cin + Cout)
    x = torch.randn(B, S, S, Cin, generator=g).to(dev()).to(torch.bfloat16)
    w = torch.randn(Cout, Cin, 3, 3, generator=g).to(dev())
    act_y = torch.randn(B, S, S, Cout, generator=g).to(dev()).to(torch.bfloat16)
    r = torch.randn(B, S, S, Cout, generator=g).to(dev()).to(torch.bfloat16) if res else None
    rs = (torch.rand(B, generator=g) + 0.5).to(dev()) if weighted else None
    scale = float(np.sqrt(2.0 / (Cin * 9)))
    wf, wd = kernels.pack_weights(w, scale)
    two = kernels.conv2d_fprop(x, wf, 3, 3, 1, residual=r)
    bias_two = torch.zeros(Cout, device=dev())
    two = kernels.lrelu_bwd(two, act_y, Cout, bias_grad=bias_two, row_scale=rs)
    bias_one = torch.full((Cout,), 3.0, device=dev())            # accumulated, not overwritten
    one = kernels.conv3x3_actgrad(x, wf, act_y, residual=r, bias_grad=bias_one, row_scale=rs)
    # the fused launch rounds once (fp32 -> mask -> bf16), the pair twice: a bf16 ulp on the slope branch
    torch.testing.assert_close(one.float(), two.float(), atol=1e-2, rtol=8e-3)
    torch.testing.assert_close(bias_one - 3.0, bias_two, atol=2e-2 * float(bias_two.abs().max()), rtol=1e-2)
    wb = (w * scale).to(torch.bfloat16).float()
    ref = F.conv2d(x.float().permute(0, 3, 1, 2), wb, None, padding=1).permute(0, 2, 3, 1)
    if res:
        ref = ref + r.float()
    ref = torch.where(act_y.float() > 0, ref, 0.2 * ref)
    torch.testing.assert_close(one.float(), ref, atol=2e-2, rtol=8e-3)
    wts = rs if weighted else torch.ones(B, device=dev())
    ref_b = (one.float() * wts[:, None, None, None]).sum(dim=(0, 1, 2))
    torch.testing.assert_close(bias_one - 3.0, ref_b, atol=1e-3 * float(ref_b.abs().max()) + 1e-3, rtol=1e-3)
    assert torch.equal(kernels.conv3x3_actgrad(x, wf, act_y, residual=r), one)       # no column sums: same image
    # second output: the stored tensor plus a per-sample multiple of the activation tile == the axpy_rows pass on both
    sc = (torch.rand(B, generator=g) - 0.5).to(dev())
    one_b, op = kernels.conv3x3_actgrad(x, wf, act_y, residual=r, operand_scale=sc)
    assert torch.equal(one_b, one)
    torch.testing.assert_close(op.float(), kernels.axpy_rows(one, act_y, sc).float(), atol=1e-2, rtol=8e-3)
    assert float((op.float() - kernels.axpy_rows(one, act_y, sc).float()).abs().max()) <= float(op.float().abs().max()) * 2 ** -7
    # as an input gradient: dy (B,S,S,Cout) through the dgrad image, masked by an activation output of x's shape
    dy = torch.randn(B, S, S, Cout, generator=g).to(dev()).to(torch.bfloat16)
    h0 = torch.randn(B, S, S, Cin, generator=g).to(dev()).to(torch.bfloat16)
    dz = kernels.conv3x3_actgrad(dy, wd, h0)
    ref_dx = F.conv_transpose2d(dy.float().permute(0, 3, 1, 2), wb, padding=1).permute(0, 2, 3, 1)
    ref_dz = torch.where(h0.float() > 0, ref_dx, 0.2 * ref_dx)
    torch.testing.assert_close(dz.float(), ref_dz, atol=2e-2 * float(ref_dx.abs().max()) / 4, rtol=1e-2)


@pytest.mark.parametrize("B,S,Cin,Cout,ups", [(4, 32, 128, 128, False), (3, 64, 128, 64, True), (8, 16, 256, 256, False),
                                            (6, 32, 256, 192, True), (64, 128, 64, 64, False)])
def test_conv_epilogue_statistics_feed_adain(B, S, Cin, Cout, ups):
    """rgbd_conv2d_fprop_stats_bf16 + rgbd_adain_apply_fixed (instance-norm statistics out of the conv epilogue as 2^-32
    fixed-point integer sums) against rgbd_conv2d_fprop_bf16 + rgbd_adain_fwd (a reduction pass over the stored tensor): the
    same image, the same statistics to fp32 rounding, bit-identical from launch to launch."""
    from rgbd_gan_amd import kernels
    g = torch.Generator().manual_seed(B + S + Cin + Cout)
    Sin = S // 2 if ups else S
    x = torch.randn(B, Sin, Sin, Cin, generator=g).to(dev()).to(torch.bfloat16)
    w = torch.randn(Cout, Cin, 3, 3, generator=g).to(dev())
    bias = torch.randn(Cout, generator=g).to(dev())
    ss = torch.randn(B, 2 * Cout + 8, generator=g).to(dev())
    wf, _ = kernels.pack_weights(w, float(np.sqrt(2.0 / (Cin * 9))), True, False)
    y_ref = kernels.conv2d_fprop(x, wf, 3, 3, 1, bias=bias, upsample=ups, lrelu_channels=Cout)
    out_ref, mean_ref, rstd_ref = kernels.adain_fwd(y_ref, ss, col_off=4)
    y, stats = kernels.conv2d_fprop_stats(x, wf, bias, upsample=ups, lrelu_channels=Cout)
    # (small problems take the split-K gather kernel without statistics: another summation order, a bf16 ulp)
    torch.testing.assert_close(y.float(), y_ref.float(), atol=2e-2, rtol=8e-3)
    out_ref, mean_ref, rstd_ref = kernels.adain_fwd(y, ss, col_off=4)
    yf = y.float()
    s1 = yf.sum(dim=(1, 2)).double()
    s2 = (yf * yf).sum(dim=(1, 2)).double()
    got = stats.double() * 2.0 ** -32
    torch.testing.assert_close(got[..., 0], s1, atol=1e-2, rtol=1e-5)       # the fp32 reference sums carry the error here
    torch.testing.assert_close(got[..., 1], s2, atol=1e-2, rtol=1e-5)
    out, mean, rstd = kernels.adain_apply_fixed(y, stats, ss, col_off=4)
    torch.testing.assert_close(mean, mean_ref, atol=1e-5, rtol=1e-5)
    torch.testing.assert_close(rstd, rstd_ref, atol=1e-5, rtol=2e-5)
    torch.testing.assert_close(out.float(), out_ref.float(), atol=2e-2, rtol=8e-3)
    for _ in range(5):
        y2, stats2 = kernels.conv2d_fprop_stats(x, wf, bias, upsample=ups, lrelu_channels=Cout)
        assert torch.equal(stats2, stats) and torch.equal(y2, y)
    assert torch.equal(kernels.adain_apply_fixed(y, stats, ss, col_off=4)[0], out)


def test_conv_epilogue_statistics_propagate_non_finite_values():
    """A NaN (or an overflow of the 2^-32 fixed-point sums) in a conv output must not leave finite, wrong instance-norm
    statistics behind (round-3 advisor finding): the sample / channel pairs a NaN reaches come out of the AdaIN as NaN, the
    others are untouched."""
    from rgbd_gan_amd import kernels
    g = torch.Generator().manual_seed(77)
    B, S, C = 2, 32, 128
    x = torch.randn(B, S, S, C, generator=g).to(torch.bfloat16)
    x[1, 7, 9, 3] = float("nan")
    x = x.to(dev())
    w = torch.randn(C, C, 3, 3, generator=g).to(dev())
    bias = torch.zeros(C, device=dev())
    ss = torch.randn(B, 2 * C, generator=g).to(dev())
    wf, _ = kernels.pack_weights(w, float(np.sqrt(2.0 / (C * 9))), True, False)
    y, stats = kernels.conv2d_fprop_stats(x, wf, bias, lrelu_channels=C)
    out, mean, rstd = kernels.adain_apply_fixed(y, stats, ss)
    assert torch.isfinite(out[0].float()).all() and torch.isfinite(mean[0]).all()
    assert torch.isnan(mean[1]).all() and torch.isnan(out[1].float()).all()       # every output channel sees input channel 3
    # overflow of the second moment: activations of ~3e3 over 1024 pixels sum to ~1e10 > 2^31
    big = (torch.randn(B, S, S, C, generator=g) * 2e3).to(torch.bfloat16).to(dev())
    y2, stats2 = kernels.conv2d_fprop_stats(big, wf, bias, lrelu_channels=C)
    out2, mean2, _ = kernels.adain_apply_fixed(y2, stats2, ss)
    ref, mean_ref, _ = kernels.adain_fwd(y2, ss)
    ok = torch.isfinite(mean2)
    torch.testing.assert_close(mean2[ok], mean_ref[ok], atol=1e-2, rtol=1e-4)     # where it is finite it is right ...
    assert (~ok).any()                                                           # ... and the overflowing pairs say so


CONV_VARIANT_CASES = [  # (B, Hout, Cin, Cout, upsample, residual, pooled output)
    (32, 64, 128, 128, False, False, False),     # two pixel tiles per persistent workgroup, two channel slices each
    (32, 64, 256, 256, False, True, True),       # four tiles per workgroup, residual + fused 2x2 average
    (8, 128, 64, 64, False, False, False),       # 64-channel tiles, one channel slice per tile
    (32, 16, 256, 256, False, True, False),      # 16x16 images: narrow tiles so that half the chip gets work
    (16, 64, 256, 128, True, False, False),      # upsample folded into the halo gather
    (6, 32, 128, 192, False, False, False),      # ragged: Cout not a multiple of 128, fewer tiles than workgroups
]


@pytest.mark.parametrize("B,H,Cin,Cout,ups,res,pool", CONV_VARIANT_CASES)
def test_pipelined_conv_matches_register_staged_kernel_bit_for_bit(B, H, Cin, Cout, ups, res, pool):
    """The LDS-DMA software-pipelined 3x3 kernel and the register-staged halo-patch kernel multiply the same fragments in
    the same order: identical bytes.  Repeated launches of the pipelined kernel must also agree with each other -- its
    staging is ordered by hand-counted vmcnt waits and barriers, and a misplaced wait shows up as an occasional stale
    tile, not as a systematic error."""
    from rgbd_gan_amd import _lib, kernels
    g = torch.Generator().manual_seed(B + H + Cin)
    Hin = H // 2 if ups else H
    x = torch.randn(B, Hin, Hin, Cin, generator=g).to(dev()).to(torch.bfloat16)
    w = torch.randn(Cout, Cin, 3, 3, generator=g).to(dev())
    bias = torch.randn(Cout, generator=g).to(dev())
    r = torch.randn(B, H, H, Cout, generator=g).to(dev()).to(torch.bfloat16) if res else None
    wf, _ = kernels.pack_weights(w, float(np.sqrt(2.0 / (Cin * 9))), True, False)

    def run():
        out = kernels.conv2d_fprop(x, wf, 3, 3, 1, bias=bias, residual=r, upsample=ups, lrelu_channels=Cout, avg_pool2=pool)
        return out if pool else (out,)
    lib = _lib.load()
    with _lib.debug_library() as dlib:                 # round 1's register-staged kernel: debug library only
        dlib.rgbd_debug_conv_variant(1)
        ref = run()
        assert dlib.rgbd_last_conv_kernel().decode().startswith("conv3x3_patch_kernel")
    for rep in range(12):
        got = run()
        assert lib.rgbd_last_conv_kernel().decode().startswith("conv3x3_sp_kernel")
        for a, b in zip(got, ref):
            assert torch.equal(a, b), f"launch {rep}: {int((a != b).sum())} of {a.numel()} values differ"


def assert_bf16_near(a, b, max_frac=2e-3):
    """bf16 tensors that hold the same fp32 sums taken in a different order: at most `max_frac` of the values differ, each by
    at most one bf16 unit in the last place (of the larger magnitude; 2^-7 relative) or, next to zero, an absolute 1e-3 of
    the tensor's scale (cancellation: a sum of O(1) products that lands near zero keeps the products' absolute noise)."""
    af, bf = a.float(), b.float()
    d = (af - bf).abs()
    frac = float((d > 0).float().mean())
    assert frac <= max_frac, f"{frac:.5f} of the values differ"
    tol = torch.maximum(af.abs(), bf.abs()) * 2.0 ** -7 + 1e-3 * float(bf.abs().max())
    assert bool((d <= tol).all()), f"worst excess {float((d - tol).max()):.3e}"


DW_CASES = [  # (B, Hout, Cin, Cout, upsample, form): 64-channel tiles, two workgroups per CU
    (8, 128, 64, 64, False, "plain"), (8, 128, 128, 64, False, "res"), (9, 128, 128, 64, True, "plain"),
    (8, 128, 64, 64, False, "pool"), (8, 128, 64, 64, False, "sumpool"), (8, 128, 128, 64, False, "actgrad"),
    (8, 128, 64, 64, False, "actgrad_y2"), (8, 128, 64, 64, False, "stats"), (10, 128, 128, 64, True, "stats"),
    (32, 64, 64, 64, False, "plain"), (3, 256, 64, 64, False, "res"),
]


@pytest.mark.parametrize("B,H,Cin,Cout,ups,form", DW_CASES)
def test_dual_workgroup_conv_matches_the_8_wave_kernel(B, H, Cin, Cout, ups, form):
    """Round 5's A/B kernel, kept correct so that its timing evidence (profiles/r05/ab_conv_dw*.txt) means something:
    conv3x3_dw_kernel (two 4-wave workgroups per CU, 32-channel slices; debug library, variant 8) against the shipped
    conv3x3_sp_kernel on every epilogue form: plain, residual, pooled second output, 2x2 sums, masked (activation gradient +
    column sums + second output), statistics.  bf16 outputs: assert_bf16_near; fp32 column sums / integer statistics: relative
    1e-4 of their scale (sums over 1e4..1e5 stored bf16 values, of which ~1e-4 differ by one ulp); repeated launches
    bit-exact (the statistics are integer sums, the column sums end in one fp32 atomic per channel per workgroup)."""
    from rgbd_gan_amd import _lib, kernels
    g = torch.Generator().manual_seed(B + H + Cin + len(form))
    Hin = H // 2 if ups else H
    x = torch.randn(B, Hin, Hin, Cin, generator=g).to(dev()).to(torch.bfloat16)
    w = torch.randn(Cout, Cin, 3, 3, generator=g).to(dev())
    bias = torch.randn(Cout, generator=g).to(dev())
    r = torch.randn(B, H, H, Cout, generator=g).to(dev()).to(torch.bfloat16)
    act = torch.randn(B, H, H, Cout, generator=g).to(dev()).to(torch.bfloat16)
    rs = (torch.rand(B, generator=g) + 0.5).to(dev())
    wf, wd = kernels.pack_weights(w, float(np.sqrt(2.0 / (Cin * 9))))

    def run():
        if form == "plain":
            return (kernels.conv2d_fprop(x, wf, 3, 3, 1, bias=bias, upsample=ups, lrelu_channels=Cout),)
        if form == "res":
            return (kernels.conv2d_fprop(x, wf, 3, 3, 1, bias=bias, residual=r, lrelu_channels=Cout),)
        if form == "pool":
            return kernels.conv2d_fprop(x, wf, 3, 3, 1, bias=bias, residual=r, lrelu_channels=Cout, avg_pool2=True)
        if form == "sumpool":       # the input gradient of an upsampling conv: r plays dy (B,H,H,Cout), dx has Cin channels
            return (kernels.conv2d_dgrad(r, wd, 3, 1, sum_pool2=True),)
        if form in ("actgrad", "actgrad_y2"):
            colsum = torch.zeros(Cout, device=dev())
            out = kernels.conv3x3_actgrad(x, wf, act, residual=r, bias_grad=colsum, row_scale=rs, slope=0.2,
                                          operand_scale=rs if form == "actgrad_y2" else None)
            return tuple(out if isinstance(out, tuple) else (out,)) + (colsum,)
        y, st = kernels.conv2d_fprop_stats(x, wf, bias, lrelu_channels=Cout, upsample=ups)
        return (y, st)
    ref = run()
    assert _lib.load().rgbd_last_conv_kernel().decode().startswith("conv3x3_sp_kernel")
    with _lib.debug_library() as dlib:
        try:
            dlib.rgbd_debug_conv_variant(8)
            first = run()
            assert dlib.rgbd_last_conv_kernel().decode().startswith("conv3x3_dw_kernel<64"), dlib.rgbd_last_conv_kernel().decode()
            again = [run() for _ in range(6)]
        finally:
            dlib.rgbd_debug_conv_variant(0)
    for a, b in zip(first, ref):
        assert a.shape == b.shape and a.dtype == b.dtype
        if a.dtype == torch.bfloat16:
            assert_bf16_near(a, b)
        elif a.dtype == torch.int64:            # instance-norm statistics in units of 2^-32
            af, bf = a.double() / 2.0 ** 32, b.double() / 2.0 ** 32
            assert float((af - bf).abs().max()) <= 1e-4 * float(bf.abs().max())
        else:
            assert float((a - b).abs().max()) <= 1e-4 * float(b.abs().max()) + 1e-6
    for rep, ag in enumerate(again):
        for a, f in zip(ag, first):
            if a.dtype == torch.float32:        # one fp32 atomic per channel per workgroup: order-dependent in the last bits
                assert float((a - f).abs().max()) <= 5e-6 * float(f.abs().max()) + 1e-7
            else:
                assert torch.equal(a, f), f"launch {rep}: {int((a != f).sum())} of {a.numel()} values differ"


@pytest.mark.parametrize("B,H,Cin,Cout,ups", [(32, 64, 128, 256, False), (4, 128, 64, 64, False), (16, 64, 256, 128, True),
                                              (2, 16, 192, 64, False), (32, 8, 256, 256, False)])
def test_wgrad_bodies_agree_and_repeat(B, H, Cin, Cout, ups):
    """All-taps-per-wave LDS-DMA body vs the tap-split register-staged body (different summation trees: fp32 rounding
    apart), and the DMA body against itself over repeated launches (bit for bit: its partial sums are ordered)."""
    from rgbd_gan_amd import _lib, kernels
    g = torch.Generator().manual_seed(H + Cin + Cout)
    Hx = H // 2 if ups else H
    x = torch.randn(B, Hx, Hx, Cin, generator=g).to(dev()).to(torch.bfloat16)
    dy = torch.randn(B, H, H, Cout, generator=g).to(dev()).to(torch.bfloat16)
    with _lib.debug_library() as dlib:
        dlib.rgbd_debug_conv_variant(3)
        ref = kernels.conv2d_wgrad(x, dy, 3, 1.0, upsample=ups)
    first = kernels.conv2d_wgrad(x, dy, 3, 1.0, upsample=ups)
    torch.testing.assert_close(first, ref, rtol=2e-5, atol=2e-5 * float(ref.abs().max()))
    for rep in range(8):
        again = kernels.conv2d_wgrad(x, dy, 3, 1.0, upsample=ups)
        assert torch.equal(again, first), f"launch {rep}: {int((again != first).sum())} values differ"


def test_pack_weights_multi_matches_single_layer_packing():
    """One launch for all layers of a network (tiled through LDS for 3x3 / 1x1 layers, element-wise for the rest) gives the
    bytes of the per-layer kernel."""
    from rgbd_gan_amd import kernels
    g = torch.Generator().manual_seed(3)
    shapes = [(64, 64, 3), (128, 64, 3), (256, 256, 3), (64, 128, 1), (96, 64, 3), (256, 64, 4), (32, 192, 3)]
    entries, refs = [], []
    for i, (co, ci, k) in enumerate(shapes):
        w = torch.randn(co, ci, k, k, generator=g).to(dev())
        scale = 0.1 + 0.01 * i
        wf = torch.zeros(k * k, co, ci, dtype=torch.bfloat16, device=dev())
        wd = torch.zeros(k * k, ci, co, dtype=torch.bfloat16, device=dev())
        entries.append((w, scale, wf, wd))
        refs.append(kernels.pack_weights(w, scale))
    kernels.pack_weights_multi(kernels.build_pack_table(entries))
    for (w, scale, wf, wd), (rf, rd) in zip(entries, refs):
        assert torch.equal(wf, rf) and torch.equal(wd, rd), tuple(w.shape)


def test_elementwise_adjoint_identities_at_benchmark_sizes():
    """B = 32, 128x128 (the benchmark's tensors): pairs of kernels that are each other's adjoint, and the bilinear form
    of the 1x1 plane convs evaluated three ways, agree to the bf16 rounding of their stored outputs:
        <pool2_masked(x; y), dp> = <x, unpool2_lrelu_bwd(dp; y)>
        <from_planes(p; w), t> = <p, to_planes(t; w^T)> = <w, planes_outer(t, p)^T>."""
    from rgbd_gan_amd import kernels
    g = torch.Generator().manual_seed(43)
    B, H, C = 32, 128, 64
    x = torch.randn(B, H, H, C, generator=g).to(dev()).to(torch.bfloat16)
    y = torch.randn(B, H, H, C, generator=g).to(dev()).to(torch.bfloat16)
    dp = torch.randn(B, H // 2, H // 2, C, generator=g).to(dev()).to(torch.bfloat16)
    a = float((kernels.pool2_masked(x, y).double() * dp.double()).sum())
    b = float((x.double() * kernels.unpool2_lrelu_bwd(dp, y, (B, H, H, C)).double()).sum())
    norm = float(x.float().norm() * dp.float().norm())
    assert abs(a - b) < 2e-4 * norm, (a, b, norm)
    p = torch.randn(B, 3, H, H, generator=g).to(dev())
    t = torch.randn(B, H, H, C, generator=g).to(dev()).to(torch.bfloat16)
    w = torch.randn(C, 3, generator=g).to(dev())
    f1 = float((kernels.from_planes(p, w, None, 0.7, act=False).double() * t.double()).sum())
    f2 = float((p.double() * kernels.to_planes(t, w.t().contiguous(), None, 0.7).double()).sum())
    o, _ = kernels.planes_outer(t, p)                                   # (3, C) = sum_{b,px} p (x) t
    f3 = 0.7 * float((w.t().double() * o.double()).sum())
    norm2 = float(p.norm() * t.float().norm()) * float(w.norm()) * 0.7
    assert abs(f1 - f2) < 2e-4 * norm2 and abs(f2 - f3) < 2e-4 * norm2, (f1, f2, f3, norm2)


def test_pool_and_unpool_lrelu_kernels():
    from rgbd_gan_amd import kernels
    g = torch.Generator().manual_seed(4)
    B, H, C = 2, 16, 128
    y = bf16_round(torch.randn(B, H, H, C, generator=g))
    x = bf16_round(torch.randn(B, H, H, C, generator=g))
    dp = bf16_round(torch.randn(B, H // 2, H // 2, C, generator=g))
    mask = torch.where(y > 0, torch.ones_like(y), torch.full_like(y, 0.2))
    up = dp.repeat_interleave(2, dim=1).repeat_interleave(2, dim=2)
    yd, xd, dpd = (t.to(dev()).to(torch.bfloat16) for t in (y, x, dp))
    # unpool (+ fused bias gradient)
    bg = torch.zeros(C, device=dev())
    bg2 = torch.ones(C, device=dev())                   # the shortcut bias of a residual block gets the same sums
    dz = kernels.unpool2_lrelu_bwd(dpd, yd, (B, H, H, C), bias_grad=bg, bias_grad2=bg2)
    torch.testing.assert_close(dz.float().cpu(), bf16_round(0.25 * up * mask), atol=1e-6, rtol=1e-6)
    torch.testing.assert_close(bg.cpu(), dz.float().cpu().reshape(-1, C).sum(0), atol=1e-3, rtol=1e-4)
    # (four blocks add to every address in arrival order, on top of different start values: equal to fp32 rounding only)
    torch.testing.assert_close(bg2.cpu() - 1.0, bg.cpu(), atol=1e-4, rtol=1e-5)
    dz0 = kernels.unpool2_lrelu_bwd(dpd, None, (B, H, H, C))
    torch.testing.assert_close(dz0.float().cpu(), bf16_round(0.25 * up), atol=1e-6, rtol=1e-6)
    # pool, masked and plain; adjointness <pool(x), dp> == <x, unpool(dp)>
    pm = kernels.pool2_masked(xd, yd)
    ref = (x * mask).reshape(B, H // 2, 2, H // 2, 2, C).sum(dim=(2, 4)) * 0.25
    torch.testing.assert_close(pm.float().cpu(), bf16_round(ref), atol=1e-2, rtol=1e-2)
    pp = kernels.pool2_masked(xd)
    ref0 = x.reshape(B, H // 2, 2, H // 2, 2, C).mean(dim=(2, 4))
    torch.testing.assert_close(pp.float().cpu(), bf16_round(ref0), atol=1e-2, rtol=1e-2)
    lhs = float((ref.double() * dp.double()).sum())
    rhs = float((x.double() * (0.25 * up * mask).double()).sum())
    assert abs(lhs - rhs) < 1e-6 * max(1.0, abs(lhs))


@pytest.mark.parametrize("M,K,N,act", [(32, 256, 256, True), (4, 265, 256, True), (64, 256, 128, False), (7, 256, 64, False),
                                       (100, 256, 256, True), (32, 4096, 256, True), (64, 4100, 256, False),
                                       (16, 1024, 40, True)])
def test_small_linear_forward_backward(M, K, N, act):
    from rgbd_gan_amd import functional as Fn
    g = torch.Generator().manual_seed(8)
    x = torch.randn(M, K, generator=g, requires_grad=True)
    w = torch.randn(N, K, generator=g, requires_grad=True)
    b = torch.randn(N, generator=g, requires_grad=True)
    dy = torch.randn(M, N, generator=g)
    c = float(np.sqrt(2.0 / K))
    ref = F.linear(x * c, w, b)
    if act:
        ref = F.leaky_relu(ref, 0.2)
    ref.backward(dy)
    xd, wd, bd = (t.detach().to(dev()).requires_grad_(True) for t in (x, w, b))
    y = Fn.linear_act(xd, wd, bd, c, act)
    y.backward(dy.to(dev()))
    torch.testing.assert_close(y.detach().cpu(), ref.detach(), atol=1e-4, rtol=1e-4)
    torch.testing.assert_close(xd.grad.cpu(), x.grad, atol=1e-4, rtol=1e-4)
    torch.testing.assert_close(wd.grad.cpu(), w.grad, atol=1e-4, rtol=1e-4)
    torch.testing.assert_close(bd.grad.cpu(), b.grad, atol=1e-4, rtol=1e-4)


@pytest.mark.parametrize("M,C,L,direct", [(16, 256, 8, True), (64, 256, 8, False), (2, 256, 8, True), (40, 256, 3, True),
                                          (32, 512, 8, True), (130, 256, 8, False), (7, 512, 2, False)])
def test_fused_mlp_chain_forward_backward(M, C, L, direct):
    """rgbd_mlp_fwd / rgbd_mlp_bwd (the mapping network's eight layers as one launch per pass, net.py:58-62) against torch fp32
    on the CPU and against the per-layer launches it replaces: output, input gradient, every weight and bias gradient --
    through autograd tensors (direct=False) and accumulated straight into bound gradient buffers that already hold values
    (direct=True: what the training step does)."""
    from rgbd_gan_amd import functional as Fn
    g = torch.Generator().manual_seed(M + C + L)
    x = torch.randn(M, C, generator=g, requires_grad=True)
    ws = [torch.randn(C, C, generator=g, requires_grad=True) for _ in range(L)]
    bs = [(0.3 * torch.randn(C, generator=g)).requires_grad_(True) for _ in range(L)]
    dy = torch.randn(M, C, generator=g)
    c = float(np.sqrt(2.0 / C))
    h = x
    for w, b in zip(ws, bs):
        h = F.leaky_relu(F.linear(h * c, w, b), 0.2)
    h.backward(dy)
    xd = x.detach().to(dev()).requires_grad_(True)
    wd = [w.detach().to(dev()).requires_grad_(True) for w in ws]
    bd = [b.detach().to(dev()).requires_grad_(True) for b in bs]
    seed = [0.5 * torch.randn(C, C, generator=g) for _ in range(L)], [0.5 * torch.randn(C, generator=g) for _ in range(L)]
    if direct:
        for t, s0 in zip(wd + bd, seed[0] + seed[1]):
            t.grad = s0.to(dev()).clone()
    y = Fn.mlp_chain(xd, wd, bd, c)
    assert y.grad_fn is not None and "MlpChain" in type(y.grad_fn).__name__          # the fused path is the one that ran
    if direct:
        with torch.no_grad():                              # a plain backward pass (no graph): gradients go into .grad directly
            torch.autograd.backward([y], [dy.to(dev())])
    else:
        y.backward(dy.to(dev()))
    tol = dict(atol=2e-4, rtol=2e-4)
    torch.testing.assert_close(y.detach().cpu(), h.detach(), **tol)
    torch.testing.assert_close(xd.grad.cpu(), x.grad, **tol)
    for i in range(L):
        gw, gb = wd[i].grad.cpu(), bd[i].grad.cpu()
        if direct:
            gw, gb = gw - seed[0][i], gb - seed[1][i]
        torch.testing.assert_close(gw, ws[i].grad, atol=2e-4 * max(1.0, float(ws[i].grad.abs().max())), rtol=2e-4)
        torch.testing.assert_close(gb, bs[i].grad, atol=2e-4 * max(1.0, float(bs[i].grad.abs().max())), rtol=2e-4)
    # ... and the per-layer kernels give the same numbers to fp32 summation order
    x2 = x.detach().to(dev()).requires_grad_(True)
    h2 = x2
    for w, b in zip(wd, bd):
        h2 = Fn.linear_act(h2, w.detach(), b.detach(), c, act=True)
    torch.testing.assert_close(h2.detach(), y.detach(), atol=1e-4, rtol=1e-4)


# ------------------------------------------------------------------------------------------------ small pointwise ops
@pytest.mark.parametrize("M,C", [(32, 256), (5, 512), (1, 265)])
def test_pixelnorm_forward_backward(M, C):
    from rgbd_gan_amd import functional as Fn
    g = torch.Generator().manual_seed(M + C)
    x = torch.randn(M, C, generator=g)
    dy = torch.randn(M, C, generator=g)
    xr = x.clone().requires_grad_(True)
    ref = nets.pixel_norm(xr)
    ref.backward(dy)
    xd = x.to(dev()).requires_grad_(True)
    got = Fn.pixel_norm(xd)
    got.backward(dy.to(dev()))
    torch.testing.assert_close(got.detach().cpu(), ref.detach(), rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(xd.grad.cpu(), xr.grad, rtol=1e-4, atol=1e-5)


def test_depth_head_forward_backward():
    from rgbd_gan_amd import functional as Fn
    g = torch.Generator().manual_seed(0)
    x = torch.randn(3, 4, 16, 16, generator=g) * 4
    x[0, 3, 0, :4] = torch.tensor([-30.0, 30.0, 0.0, -90.0])            # softplus tails
    dy = torch.randn(3, 4, 16, 16, generator=g)
    xr = x.clone().requires_grad_(True)
    ref = nets.depth_head(xr)
    ref.backward(dy)
    xd = x.to(dev()).requires_grad_(True)
    got = Fn.depth_head(xd)
    got.backward(dy.to(dev()))
    torch.testing.assert_close(got.detach().cpu(), ref.detach(), rtol=2e-6, atol=1e-6)
    torch.testing.assert_close(xd.grad.cpu(), xr.grad, rtol=1e-4, atol=1e-4)
    assert torch.equal(got[:, :3].detach().cpu(), x[:, :3])


def test_ema_update_matches_two_statement_form():
    from rgbd_gan_amd import kernels
    g = torch.Generator().manual_seed(1)
    dst, src = torch.randn(100003, generator=g), torch.randn(100003, generator=g)
    tau = 1.0 - 0.999
    ref = dst.clone()
    ref *= (1 - tau)                      # copy_param.py:30-31
    ref += tau * src
    d = dst.to(dev())
    kernels.ema_update(d, src.to(dev()), tau)
    torch.testing.assert_close(d.cpu(), ref, rtol=1e-6, atol=1e-7)


def test_smoothed_generator_tracks_updates():
    """keep_smoothed_gen (updater.py:397-400): after each step smoothed = 0.999 smoothed + 0.001 gen."""
    from rgbd_gan_amd.training import DeviceImageIterator, build_training
    from rgbd_gan_amd.utils.yaml_utils import Config
    cfg = dict(generator_architecture="stylegan", ch=256, stage_interval="0,0,0,0,0,0,0,100000,150000,160000,180000,300000",
               max_stage=11, start_rotation=2000, start_occlusion_aware=2000, lambda_depth=10, depth_min=1.0,
               x_rotate=0.3054, y_rotate=1.0472, z_rotate=0, x_translate=0, y_translate=0, z_translate=0, bigan=False,
               adam_alpha_g=0.001, adam_alpha_d=0.003, adam_beta1=0.0, adam_beta2=0.999, lambda_gp=1.0, smoothing=0.999,
               res_dis=True, sn=False, enable_blur=False, keep_smoothed_gen=True)
    images = np.random.RandomState(0).randint(0, 256, (8, 3, 128, 128)).astype("uint8")
    it = DeviceImageIterator(images, 4, "cuda:0", seed=3)
    gen, dis, opt, upd = build_training(Config(cfg), "cuda:0", iterator=it, fixed_stage=6.0, use_graphs=False,
                                        nan_check_interval=0)
    sm = upd.smoothed_gen
    assert sm is not None and not torch.equal(sm.gen.store.flat, gen.gen.store.flat)
    s0 = sm.gen.store.flat.clone()
    upd.update()
    expect = s0 * (1 - 0.001) + 0.001 * gen.gen.store.flat
    torch.testing.assert_close(sm.gen.store.flat, expect, rtol=1e-5, atol=1e-6)
    m0 = sm.mapping.store.flat.clone()
    upd.update()
    torch.testing.assert_close(sm.mapping.store.flat, m0 * (1 - 0.001) + 0.001 * gen.mapping.store.flat, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("B,H,Cin,Cout,K,pad,ups", [(32, 8, 256, 256, 3, 1, False), (32, 4, 256, 256, 3, 1, False),
                                                     (2, 16, 256, 256, 3, 1, False), (4, 4, 256, 256, 3, 1, True),
                                                     (2, 8, 256, 128, 1, 0, False), (3, 5, 64, 128, 3, 1, False)])
def test_split_k_path_matches_unsplit(B, H, Cin, Cout, K, pad, ups):
    """Small layers are split along K over several workgroups (fp32 partials + finishing kernel): same result as the
    single-pass kernel up to the fp32 summation order, including bias / residual / leaky ReLU / ragged M."""
    from rgbd_gan_amd import _lib, kernels
    lib = _lib.load()
    g = torch.Generator().manual_seed(B * 100 + H)
    x = torch.randn(B, H, H, Cin, generator=g).to(torch.bfloat16).to(dev())
    w = torch.randn(Cout, Cin, K, K, generator=g).to(dev())
    bias = torch.randn(Cout, generator=g).to(dev())
    Ho = (2 * H if ups else H) + 2 * pad - K + 1
    res = torch.randn(B, Ho, Ho, Cout, generator=g).to(torch.bfloat16).to(dev())
    wf, _ = kernels.pack_weights(w, float(np.sqrt(2.0 / (Cin * K * K))))
    assert lib.rgbd_conv2d_fprop_workspace(B, H, H, Cin, Cout, K, K, pad, int(ups)) > 0      # these shapes are split
    y_split = kernels.conv2d_fprop(x, wf, K, K, pad, bias=bias, residual=res, upsample=ups, lrelu_channels=Cout)
    # unsplit: call the C ABI without scratch
    y_ref = torch.empty_like(y_split)
    rc = lib.rgbd_conv2d_fprop_bf16(kernels._ptr(x), kernels._ptr(wf), kernels._ptr(bias), kernels._ptr(res),
                                    kernels._ptr(y_ref), None, B, H, H, Cin, Cout, K, K, pad, int(ups), Cout, 0.2, None,
                                    0, kernels._stream())
    assert rc == 0
    d = (y_split.float() - y_ref.float()).abs().max().item()
    scale = y_ref.float().abs().max().item()
    assert d <= 2 ** -7 * scale, (d, scale)            # one bf16 ulp of the largest output
    assert (y_split.float() - y_ref.float()).abs().mean().item() < 1e-3 * scale


@pytest.mark.parametrize("max_depth,min_depth,occ", [(1.0, None, False), (None, 0.95, True), (1.2, 0.8, True)])
def test_warp_loss_depth_range_masks(max_depth, min_depth, occ):
    """max_depth / min_depth of LossFuncRotate.__call__ (loss_functions.py:104-118, the background-generator branch of
    the deepvoxels updater): masks on the SOURCE depth, forward value and both gradients."""
    from rgbd_gan_amd.common.loss_functions import LossFuncRotate
    b, S = 2, 32
    img, img_rot, cam, cam_rot = _warp_case(b, S, seed=77)
    ref = warp_loss.forward_np(img, cam, img_rot, cam_rot, occlusion_aware=occ, lambda_geometric=3.0,
                               max_depth=max_depth, min_depth=min_depth)
    ti = torch.from_numpy(img).requires_grad_(True)
    tr = torch.from_numpy(img_rot).requires_grad_(True)
    lt, _ = warp_loss.loss_torch(ti, cam, tr, cam_rot, occlusion_aware=occ, lambda_geometric=3.0, max_depth=max_depth,
                                 min_depth=min_depth)
    lt.backward()
    di = torch.from_numpy(img).to(dev()).requires_grad_(True)
    dr = torch.from_numpy(img_rot).to(dev()).requires_grad_(True)
    fn = LossFuncRotate(torch, lambda_geometric=3)
    loss, _ = fn(di, cam, dr, cam_rot, occlusion_aware=occ, max_depth=max_depth, min_depth=min_depth)
    loss.backward()
    assert abs(float(loss.detach()) - ref["loss"]) < 1e-4 * max(1.0, abs(ref["loss"]))
    assert abs(float(lt.detach()) - ref["loss"]) < 1e-5 * max(1.0, abs(ref["loss"]))
    scale = float(ti.grad.abs().max())
    torch.testing.assert_close(di.grad.cpu(), ti.grad, atol=2e-5 * scale, rtol=1e-4)
    torch.testing.assert_close(dr.grad.cpu(), tr.grad, atol=2e-5 * scale, rtol=1e-4)
    # the masks do remove pixels in this case (otherwise the test would not see the flags)
    base = warp_loss.forward_np(img, cam, img_rot, cam_rot, occlusion_aware=occ, lambda_geometric=3.0)["loss"]
    assert abs(base - ref["loss"]) > 1e-3


@pytest.mark.parametrize("C,norm,occ", [(4, "l2", True), (6, "l2", False), (9, "l1", True), (2, "l1", False)])
def test_loss_func_rotate_any_channel_count_and_l2(C, norm, occ):
    """LossFuncRotate(norm="l2") (loss_functions.py:137-140: F.mean_squared_error) and inputs with other channel counts
    than RGB-D (the last channel is the depth; updater.py:345-354 feeds 257-channel features): value, both gradients and the
    second return value against the oracle's differentiable restatement."""
    from rgbd_gan_amd.common.loss_functions import LossFuncRotate
    b, S = 2, 32
    img4, img_rot4, cam, cam_rot = _warp_case(b, S, seed=31)
    rng = np.random.RandomState(C)
    def widen(x):        # C - 1 feature channels + the depth channel of the 4-channel case
        feats = rng.uniform(-1, 1, (b, C - 1, S, S)).astype("float32")
        return np.concatenate([feats, x[:, 3:]], axis=1)
    img, img_rot = widen(img4), widen(img_rot4)
    ti = torch.from_numpy(img).requires_grad_(True)
    tr = torch.from_numpy(img_rot).requires_grad_(True)
    lt, zp_ref = warp_loss.loss_torch(ti, cam, tr, cam_rot, occlusion_aware=occ, lambda_geometric=3.0, norm=norm)
    lt.backward()
    di = torch.from_numpy(img).to(dev()).requires_grad_(True)
    dr = torch.from_numpy(img_rot).to(dev()).requires_grad_(True)
    loss, zp = LossFuncRotate(torch, norm=norm, lambda_geometric=3)(di, cam, dr, cam_rot, occlusion_aware=occ)
    loss.backward()
    assert abs(float(loss.detach()) - float(lt.detach())) < 1e-4 * max(1.0, abs(float(lt.detach())))
    scale = float(ti.grad.abs().max())
    torch.testing.assert_close(di.grad.cpu(), ti.grad, atol=2e-5 * scale, rtol=1e-4)
    torch.testing.assert_close(dr.grad.cpu(), tr.grad, atol=2e-5 * scale, rtol=1e-4)
    assert tuple(zp.shape) == (2 * b, S * S, 3)
    torch.testing.assert_close(zp.cpu(), zp_ref.detach(), atol=1e-4, rtol=1e-5)
    # the scatter-add of the taps is accumulated as integers (round 4): launch after launch the same bits, for both criteria
    from rgbd_gan_amd import kernels
    coef = torch.from_numpy(_coef(cam, cam_rot, S)).to(dev())
    go = torch.tensor([1.3], dtype=torch.float32, device=dev())
    first = kernels.warp_loss_nc_bwd(di.detach(), dr.detach(), coef, 1 if occ else 0, norm == "l2", 3.0, 0.0, 0.0, go)
    for _ in range(6):
        again = kernels.warp_loss_nc_bwd(di.detach(), dr.detach(), coef, 1 if occ else 0, norm == "l2", 3.0, 0.0, 0.0, go)
        assert torch.equal(again[0], first[0]) and torch.equal(again[1], first[1])
    torch.testing.assert_close(first[0].cpu(), 1.3 * ti.grad, atol=3e-5 * scale, rtol=1e-4)


def test_loss_func_rotate_debug_tuple():
    """debug=True returns (warped, mask, zp, warped_rot, mask_rot, zp_rot) like loss_functions.py:100-102."""
    from rgbd_gan_amd.common.loss_functions import LossFuncRotate
    b, S = 2, 16
    img, img_rot, cam, cam_rot = _warp_case(b, S, seed=5)
    ref = warp_loss.forward_np(img, cam, img_rot, cam_rot, occlusion_aware=False, lambda_geometric=3.0)
    out = LossFuncRotate(torch)(torch.from_numpy(img).to(dev()), cam, torch.from_numpy(img_rot).to(dev()), cam_rot,
                                debug=True)
    assert len(out) == 6
    warped, mask, zp, warped_rot, mask_rot, zp_rot = (t.cpu().numpy() for t in out)
    np.testing.assert_array_equal(warped.view(np.uint32), ref["warped"].view(np.uint32))
    np.testing.assert_array_equal(mask, ref["mask"])
    np.testing.assert_array_equal(zp.view(np.uint32), ref["zp"].view(np.uint32))
    np.testing.assert_array_equal(warped_rot.view(np.uint32), ref["warped_rot"].view(np.uint32))
    np.testing.assert_array_equal(mask_rot, ref["mask_rot"])
    np.testing.assert_array_equal(zp_rot.view(np.uint32), ref["zp_rot"].view(np.uint32))


def test_loss_func_rotate_second_value_is_the_projected_points():
    """loss_functions.py:146: the call returns (loss, F.concat([new_zp, new_zp_rot], axis=0)) -- (2b, hw, 3)."""
    from rgbd_gan_amd.common.loss_functions import LossFuncRotate
    b, S = 2, 16
    img, img_rot, cam, cam_rot = _warp_case(b, S, seed=6)
    ref = warp_loss.forward_np(img, cam, img_rot, cam_rot, occlusion_aware=True, lambda_geometric=3.0)
    loss, zp_cat = LossFuncRotate(torch)(torch.from_numpy(img).to(dev()), cam, torch.from_numpy(img_rot).to(dev()), cam_rot,
                                         occlusion_aware=True)
    assert tuple(zp_cat.shape) == (2 * b, S * S, 3) and not zp_cat.requires_grad
    want = np.concatenate([ref["zp"], ref["zp_rot"]], axis=0)
    np.testing.assert_array_equal(zp_cat.cpu().numpy().view(np.uint32), want.view(np.uint32))
    assert abs(float(loss) - ref["loss"]) < 1e-4 * max(1.0, abs(ref["loss"]))
