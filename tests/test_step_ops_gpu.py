"""The small fused ops of the training step (rgbd_gan_amd/csrc/step_ops.hip and the ABI-4 additions to the warp-loss,
AdaIN and linear kernels) against the oracle / the unfused compositions they replace.  Needs an MI355X."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import nets, step, warp_loss

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_real_batch_kernel_is_transform_plus_downsize_real():
    """train_rgbd.py:308-310 (x / 127.5 - 1) + common/utils/pggan.py:6-50 (downsize_real): even stages and fade-in
    stages, alpha from a device scalar, against the oracle's restatement on the same uint8 images."""
    from rgbd_gan_amd import kernels
    rng = np.random.RandomState(0)
    data = rng.randint(0, 256, (20, 3, 128, 128)).astype("uint8")
    idx = np.array([3, 17, 0, 9, 9, 12])
    dd, di = torch.from_numpy(data).to(DEV), torch.from_numpy(idx).to(DEV)
    x = torch.from_numpy(data[idx].astype("float32") / 127.5 - 1)
    for stage, size in ((10.0, 128), (8.0, 64), (6.0, 32), (2.0, 8)):
        got = kernels.real_batch(dd, di, size).cpu()
        ref = nets.downsize_real(x, stage)
        assert got.shape == ref.shape
        torch.testing.assert_close(got, ref, atol=2e-6, rtol=0)
    for stage, size in ((9.25, 128), (7.5, 64), (5.9, 32)):
        alpha = stage - int(stage)
        ref = nets.downsize_real(x, stage)
        got = kernels.real_batch(dd, di, size, alpha=torch.tensor(alpha, dtype=torch.float32, device=DEV)).cpu()
        torch.testing.assert_close(got, ref, atol=2e-6, rtol=0)
        got = kernels.real_batch(dd, di, size, alpha=alpha).cpu()
        torch.testing.assert_close(got, ref, atol=2e-6, rtol=0)
    # identity at full resolution: exactly the transform
    got = kernels.real_batch(dd, di, 128).cpu()
    assert float((got - x).abs().max()) <= 6e-8 * 2


def test_zero_multi_and_hidden_normalize():
    from rgbd_gan_amd import kernels
    bufs = [torch.randn(n, device=DEV) for n in (8, 4096, 526336, 12, 100004)]
    kernels.zero_multi(bufs)
    assert all(float(b.abs().max()) == 0.0 for b in bufs)
    bufs = [torch.randn(16 * (i + 1), device=DEV) for i in range(11)]         # more than one launch group
    kernels.zero_multi(bufs)
    assert all(float(b.abs().max()) == 0.0 for b in bufs)
    z = torch.randn(5, 512, device=DEV)
    out = kernels.hidden_normalize(z, 256.0, copies=2).cpu()
    zc = z.cpu().numpy()
    ref = zc / np.sqrt(np.sum(zc * zc, axis=1, keepdims=True) / 256 + 1e-8)      # net.py:333-343
    np.testing.assert_allclose(out[:5].numpy(), ref, rtol=2e-6, atol=1e-7)
    assert torch.equal(out[:5], out[5:])


def test_hidden_draw_is_make_hidden_with_the_draw_inside():
    """rgbd_hidden_draw (Philox4x32-10 + Box-Muller + net.py:333-343's normalisation in one launch): every row has
    sum z^2 / ch == 1, the raw draws are N(0,1) (moments and tail fractions over 2 M values, no correlation between the two
    Box-Muller outputs or neighbouring rows), the same torch seed reproduces the sequence, successive launches differ, and a
    launch replayed from a captured HIP graph draws fresh values on every replay (the launch number lives on the device)."""
    from rgbd_gan_amd import kernels
    torch.manual_seed(1234)
    st = kernels.new_hidden_rng_state(DEV)
    a1 = kernels.hidden_draw(st, 4096, 512, 256.0)
    a2 = kernels.hidden_draw(st, 4096, 512, 256.0)
    assert st.cpu().tolist()[2:] == [2, 0]                         # the launch number advanced on the device, the ticket is back at 0
    torch.manual_seed(1234)
    b1 = kernels.hidden_draw(kernels.new_hidden_rng_state(DEV), 4096, 512, 256.0)
    torch.manual_seed(1235)
    c1 = kernels.hidden_draw(kernels.new_hidden_rng_state(DEV), 4096, 512, 256.0)
    assert torch.equal(a1, b1) and not torch.equal(a1, a2) and not torch.equal(a1, c1)
    ms = (a1.double() ** 2).sum(dim=1) / 256.0
    assert float((ms - 1.0).abs().max()) < 1e-5
    # un-normalise: a row's scale is sqrt(sum z^2 / ch); over 512 draws it is sqrt(2) (1 +- 0.03), so the pooled moments of
    # z * sqrt(2) / ... are checked on the normalised values with their known scale: E[x^2] = ch / C = 1/2
    x = a1.double().flatten()
    n = x.numel()
    assert abs(float(x.mean())) < 4.0 * (0.5 / n) ** 0.5 * 1.5
    assert abs(float((x ** 2).mean()) - 0.5) < 1e-9 + 1e-12          # exact by the normalisation
    z = x / (0.5 ** 0.5)                                             # ~N(0,1) up to the 3 % row scale
    assert abs(float((z ** 3).mean())) < 0.02 and abs(float((z ** 4).mean()) - 3.0) < 0.06
    for t, p in ((1.0, 0.31731), (2.0, 0.04550), (3.0, 0.00270)):
        assert abs(float((z.abs() > t).double().mean()) - p) < 0.08 * p + 2e-4
    zz = z.reshape(4096, 128, 4)
    assert abs(float((zz[..., 0] * zz[..., 1]).mean())) < 0.01 and abs(float((zz[..., 0] * zz[..., 2]).mean())) < 0.01
    assert abs(float((zz[1:, :, 0] * zz[:-1, :, 0]).mean())) < 0.01
    # copies: rows m and m + M are the same latent
    c = kernels.hidden_draw(st, 5, 512, 256.0, copies=2)
    assert torch.equal(c[:5], c[5:]) and float((c[0] - c[1]).abs().max()) > 0.1
    # graph replay: new draws every time
    s = torch.cuda.Stream()
    out = torch.empty(8, 512, device=DEV)
    with torch.cuda.stream(s):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            out.copy_(kernels.hidden_draw(st, 8, 512, 256.0))
        seen = []
        for _ in range(3):
            g.replay()
            s.synchronize()
            seen.append(out.clone())
    assert not torch.equal(seen[0], seen[1]) and not torch.equal(seen[1], seen[2])


def test_preview_latents_are_seeded_and_leave_the_training_stream_alone():
    """sample_generate_light draws its latents once from a FIXED seed (train_rgbd.py:39-92).  StyleGAN's latents come from the
    library's Philox stream, so the preview sampler draws from a private stream seeded with its own seed: the same latents
    whatever torch's seed is (every run, after a resume), and the generator's training stream neither moves nor is shared
    by a second generator object (gen / smoothed_gen)."""
    from rgbd_gan_amd.common.utils.save_images import PreviewSampler
    from rgbd_gan_amd.net import StyleGANGenerator
    from rgbd_gan_amd.utils.yaml_utils import Config
    cfg = Config(generator_architecture="stylegan", rgb=False, test_y_rotate=0.5)

    def preview_latents(torch_seed, draws_before):
        torch.manual_seed(torch_seed)
        gen, smoothed = StyleGANGenerator(256, rgbd=True, device=DEV), StyleGANGenerator(256, rgbd=True, device=DEV, seed=1000)
        for _ in range(draws_before):
            gen.make_hidden(4)
        before = gen._latent_rng().cpu().tolist()
        ps = PreviewSampler(gen, "/tmp/unused", cfg, rows=2, cols=3, seed=0)
        ps.render(4.0)
        assert gen._latent_rng().cpu().tolist() == before               # the training stream did not move
        assert before[2] == draws_before
        train_next = gen.make_hidden(4)
        assert not torch.equal(train_next, smoothed.make_hidden(4))     # two generator objects, two streams
        return ps.z.clone(), train_next

    z_a, t_a = preview_latents(11, 0)
    z_b, t_b = preview_latents(12, 3)
    z_c, t_c = preview_latents(11, 0)
    assert z_a.shape == (6, 512, 1, 1) and torch.equal(z_a[0], z_a[1]) and not torch.equal(z_a[0], z_a[2])   # tiled over the rows
    assert torch.equal(z_a, z_b) and torch.equal(z_a, z_c)             # seed 0 -> the same preview latents, always
    assert torch.equal(t_a, t_c) and not torch.equal(t_a, t_b)         # the training stream follows torch's seed
    other = PreviewSampler(StyleGANGenerator(256, rgbd=True, device=DEV), "/tmp/unused", cfg, rows=2, cols=3, seed=1)
    other.render(4.0)
    assert not torch.equal(other.z, z_a)


def test_r1_penalty_kernel_and_gradient():
    from rgbd_gan_amd import functional as Fn
    g = torch.randn(6, 3, 64, 64, device=DEV) * 0.3
    gc = g.cpu().clone().requires_grad_(True)
    norm = torch.sqrt((gc ** 2).sum(dim=(1, 2, 3)))
    ref = 1.5 * ((norm - 0.0) ** 2).sum() / norm.numel()                          # step.r1_penalty's tail
    ref.backward()
    gd = g.clone().requires_grad_(True)
    got = Fn.r1_penalty(gd, 1.5)
    (got * 0.7).backward()
    assert abs(float(got) - float(ref)) < 1e-5 * float(ref)
    torch.testing.assert_close(gd.grad.cpu(), 0.7 * gc.grad, rtol=1e-5, atol=1e-9)


def test_axpy_rows_f32():
    """a + s[b] * x per sample (the adversarial injection's operand at the image planes) and, without s, the in-place sum
    of two flat gradient buffers (the join of D's two buffers)."""
    from rgbd_gan_amd import kernels
    g = torch.Generator().manual_seed(3)
    a, x = torch.randn(5, 3, 8, 8, generator=g).cuda(), torch.randn(5, 3, 8, 8, generator=g).cuda()
    sc = torch.randn(5, generator=g).cuda()
    got = kernels.axpy_rows_f32(a, x, sc)
    torch.testing.assert_close(got, a + sc.reshape(-1, 1, 1, 1) * x, rtol=0, atol=1e-6)
    flat, other = torch.randn(1000, generator=g).cuda(), torch.randn(1000, generator=g).cuda()
    want = flat + other
    kernels.axpy_rows_f32(flat, other, out=flat)
    assert torch.equal(flat, want)
    with pytest.raises(RuntimeError):
        kernels.axpy_rows_f32(a[:, :1, :1, :3].contiguous(), x[:, :1, :1, :3].contiguous(), sc)     # rows of 3 floats


def test_image_grad_init():
    from rgbd_gan_amd import kernels
    gx = torch.randn(4, 3, 16, 16, device=DEV)
    ratio = torch.tensor([-1.0, -0.5, -2.0, -1e3], device=DEV)
    out = kernels.image_grad_init(gx, ratio, 4)
    assert torch.equal(out[:, :3], gx * ratio.reshape(-1, 1, 1, 1)) and float(out[:, 3].abs().max()) == 0.0
    assert torch.equal(kernels.image_grad_init(gx, None, 3), gx)


def test_const_input_forward_backward():
    from rgbd_gan_amd import functional as Fn
    C, B = 256, 6
    w = torch.randn(C, 4, 4, device=DEV, requires_grad=True)
    b = torch.randn(C, device=DEV, requires_grad=True)
    h = Fn.const_input(w, b, B)
    assert h.shape == (B, 4, 4, C) and h.dtype == torch.bfloat16
    wr, br = w.detach().cpu().requires_grad_(True), b.detach().cpu().requires_grad_(True)
    ref = F.leaky_relu(wr + br.reshape(-1, 1, 1), 0.2).permute(1, 2, 0).unsqueeze(0).expand(B, 4, 4, C)
    torch.testing.assert_close(h.float().cpu(), ref.to(torch.bfloat16).float(), rtol=0, atol=0)
    dh = torch.randn(B, 4, 4, C, device=DEV).to(torch.bfloat16)
    h.backward(dh)
    ref.backward(dh.float().cpu())
    torch.testing.assert_close(w.grad.cpu(), wr.grad, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(b.grad.cpu(), br.grad, rtol=1e-5, atol=1e-4)


def test_nhwc_rows_roundtrip_and_adjoint():
    from rgbd_gan_amd import kernels
    h = torch.randn(5, 4, 4, 256, device=DEV).to(torch.bfloat16)
    rows = kernels.nhwc_to_rows(h)
    torch.testing.assert_close(rows.cpu(), h.float().permute(0, 3, 1, 2).reshape(5, -1).cpu(), rtol=0, atol=0)
    assert torch.equal(kernels.rows_to_nhwc(rows, 4, 4, 256), h)


@pytest.mark.parametrize("B", [4, 32, 70])
def test_dense_tail_twice_differentiable(B):
    """The discriminator's dense tail (net.py:372-377: 4x4-valid conv as a linear, leaky ReLU, output linear) on the HIP
    linear kernels: forward, input gradient, weight gradients, and the R1 pattern -- gradient of ||d y / d x||^2 w.r.t.
    the weights (double backward) -- against torch autograd on the same fp32 math."""
    from rgbd_gan_amd import functional as Fn
    C = 256
    c1, c2 = float(np.sqrt(2) / np.sqrt(C * 16)), float(1 / np.sqrt(C))
    g = torch.Generator().manual_seed(B)
    W1 = torch.randn(C, C, 4, 4, generator=g)
    b1 = torch.randn(C, generator=g) * 0.1
    W2 = torch.randn(1, C, generator=g)
    b2 = torch.randn(1, generator=g)
    x = torch.randn(B, C * 16, generator=g)

    def run(dev, dense):
        ps = [t.clone().to(dev).requires_grad_(True) for t in (W1, b1, W2, b2)]
        for p in ps:
            p.grad = torch.zeros_like(p)
        xx = x.clone().to(dev).requires_grad_(True)
        outs = []
        for r0 in range(0, B, 64):
            u = dense(xx[r0:r0 + 64], ps[0], ps[1], c1, True)
            outs.append(dense(u, ps[2], ps[3], c2, False))
        y = torch.cat(outs)
        gx, = torch.autograd.grad([y], [xx], [torch.ones_like(y)], create_graph=True)
        pen = (gx ** 2).sum() / B
        (pen + (y * torch.linspace(-1, 1, B, device=dev).reshape(-1, 1)).sum()).backward()
        return y.detach().cpu(), gx.detach().cpu(), [p.grad.cpu() for p in ps], xx.grad.cpu()

    def ref_dense(xx, w, b, c, act):
        y = F.linear(xx * c, w.reshape(w.shape[0], -1), b)
        return F.leaky_relu(y, 0.2) if act else y

    y_r, gx_r, gp_r, xg_r = run("cpu", ref_dense)
    y_d, gx_d, gp_d, xg_d = run(DEV, Fn.dense)
    torch.testing.assert_close(y_d, y_r, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(gx_d, gx_r, rtol=1e-4, atol=1e-6)
    for a, b in zip(gp_d, gp_r):
        torch.testing.assert_close(a, b, rtol=2e-4, atol=2e-5)
    torch.testing.assert_close(xg_d, xg_r, rtol=2e-4, atol=2e-6)


def _warp_inputs(b, S, seed):
    from oracle import camera
    rng = np.random.RandomState(seed)
    x = rng.uniform(-1, 1, (2 * b, 4, S, S)).astype("float32")
    x[:, 3] = rng.uniform(0.7, 1.3, (2 * b, S, S))
    th = rng.uniform(-0.3, 0.3, (2 * b, 6)).astype("float32")
    th[:, 2] = 0
    th[:, 3:] *= 0.1
    cams = camera.camera_matrices(th)
    K, inv_K, _ = warp_loss.intrinsics(S)
    R, t = warp_loss.relative_pose(cams[:b], cams[b:])
    A, c, A2, c2 = warp_loss.warp_coefficients(K, inv_K, R, t)
    coef = np.concatenate([A.reshape(b, 9), c, A2.reshape(b, 9), c2], axis=1).astype("float32")
    return x, cams, coef


@pytest.mark.parametrize("b,S", [(2, 16), (3, 32)])
def test_warp_loss_with_fused_hinge_and_accumulation(b, S):
    """ABI 4: the depth hinge of updater.py:357-359 evaluated inside the warp-loss kernels, and the backward ADDING
    lambda_rotate * d loss into a buffer that already holds the adversarial image gradient -- against the oracle's
    autograd on loss_rotate + hinge."""
    from rgbd_gan_amd import kernels
    x, cams, coef = _warp_inputs(b, S, seed=5)
    lam_geo, lam_depth, dmin, lam_rot = 2.0, 10.0, 1.0, 2.0
    xt = torch.from_numpy(x).requires_grad_(True)
    ref, _ = warp_loss.loss_torch(xt[:b], cams[:b], xt[b:], cams[b:], occlusion_aware=True, lambda_geometric=lam_geo)
    ref = ref + step.depth_hinge(xt, dmin, lam_depth)
    (ref * lam_rot).backward()
    xd = torch.from_numpy(x).to(DEV)
    cf = torch.from_numpy(coef).to(DEV)
    loss = kernels.warp_loss_fwd(xd[:b], xd[b:], cf, 1, lam_geo, hinge_lambda=lam_depth, hinge_min=dmin)
    assert abs(float(loss) - float(ref)) < 1e-5 * max(1.0, abs(float(ref)))
    base = torch.randn(2 * b, 4, S, S, device=DEV) * 1e-3
    gout = base.clone()
    kernels.warp_loss_bwd(xd[:b], xd[b:], cf, 1, lam_geo, 0.0, 0.0, None, hinge_lambda=lam_depth, hinge_min=dmin,
                          grad_scale=lam_rot, out=(gout[:b], gout[b:]))
    torch.testing.assert_close((gout - base).cpu(), xt.grad, rtol=1e-3, atol=2e-8)
    # hinge off + no accumulation = the ABI-3 behaviour
    l0 = kernels.warp_loss_fwd(xd[:b], xd[b:], cf, 1, lam_geo)
    assert float(l0) < float(loss)


def test_adain_statistics_are_bit_reproducible():
    """Strip partial sums are plain stores added in index order (no atomics): the same input gives the same bits."""
    from rgbd_gan_amd import kernels
    x = torch.randn(4, 128, 128, 64, device=DEV).to(torch.bfloat16)
    ss = torch.randn(4, 128, device=DEV)
    dy = torch.randn(4, 128, 128, 64, device=DEV).to(torch.bfloat16)
    runs = []
    for _ in range(3):
        y, mean, rstd = kernels.adain_fwd(x, ss)
        dx, dss, _ = kernels.adain_bwd(x, dy, ss, mean, rstd, fused=True)
        runs.append((y, mean, rstd, dx, dss))
    for r in runs[1:]:
        for a, b in zip(r, runs[0]):
            assert torch.equal(a, b)
    xf = x.float()
    m = xf.mean(dim=(1, 2))
    torch.testing.assert_close(runs[0][1], m, rtol=1e-4, atol=1e-5)


def test_generator_forward_is_bit_reproducible():
    from rgbd_gan_amd.net import StyleGANGenerator
    from oracle import camera
    gen = StyleGANGenerator(256, rgbd=True)
    rng = np.random.RandomState(0)
    zh = nets.make_hidden(2, 256, rng)
    z = np.concatenate([zh, zh])
    np.random.seed(1)
    t9 = camera.theta9(camera.PosePrior(0.3054, 1.0472, 0).sample(4))
    with torch.no_grad():
        a = gen(z, 10.0, t9)
        b = gen(z, 10.0, t9)
    assert torch.equal(a, b)


@pytest.mark.parametrize("C", [128, 256, 512])
def test_l2_normalize_forward_backward(C):
    """DCGANBlock's F.normalize over channels (net.py:621-648), chainer semantics x / (||x|| + 1e-5)."""
    from rgbd_gan_amd import functional as Fn
    g = torch.Generator().manual_seed(C)
    x = torch.randn(3, 8, 8, C, generator=g).to(torch.bfloat16)
    x[0, 0, 0] = 0                                            # a zero vector: y = 0, dx = dy / eps
    dy = torch.randn(3, 8, 8, C, generator=g).to(torch.bfloat16)
    xr = x.float().requires_grad_(True)
    yr = nets.l2_normalize(xr.permute(0, 3, 1, 2)).permute(0, 2, 3, 1)
    yr.backward(dy.float())
    xd = x.to(DEV).requires_grad_(True)
    yd = Fn.l2_normalize(xd)
    yd.backward(dy.to(DEV))
    torch.testing.assert_close(yd.float().cpu(), yr.detach().to(torch.bfloat16).float(), rtol=0, atol=1e-2)
    ref = xr.grad[1:]
    torch.testing.assert_close(xd.grad.float().cpu()[1:], ref, rtol=2e-2, atol=2e-2 * float(ref.abs().max()))


def test_blur_modes_and_adjoints():
    """rescale.py:20-25: plain blur, blur of the nearest-upsampled tensor, and its adjoint (2x2 sums of the blur)."""
    from rgbd_gan_amd import functional as Fn, kernels
    g = torch.Generator().manual_seed(2)
    x = torch.randn(2, 8, 8, 64, generator=g).to(torch.bfloat16)
    xn = x.float().permute(0, 3, 1, 2)
    got = kernels.blur3x3(x.to(DEV), 0).float().cpu().permute(0, 3, 1, 2)
    torch.testing.assert_close(got, nets.blur(xn).to(torch.bfloat16).float(), rtol=0, atol=2e-2)
    got = kernels.blur3x3(x.to(DEV), 1).float().cpu().permute(0, 3, 1, 2)
    torch.testing.assert_close(got, nets.blur(nets.up2(xn)).to(torch.bfloat16).float(), rtol=0, atol=2e-2)
    # adjoint pairs: <A x, y> = <x, A^T y>
    y = torch.randn(2, 16, 16, 64, generator=g).to(torch.bfloat16)
    lhs = float((kernels.blur3x3(x.to(DEV), 1).float() * y.to(DEV).float()).sum())
    rhs = float((x.to(DEV).float() * kernels.blur3x3(y.to(DEV), 2).float()).sum())
    assert abs(lhs - rhs) < 2e-2 * (abs(lhs) + 1.0), (lhs, rhs)
    z = torch.randn(2, 8, 8, 64, generator=g).to(torch.bfloat16)
    lhs = float((kernels.blur3x3(x.to(DEV), 0).float() * z.to(DEV).float()).sum())
    rhs = float((x.to(DEV).float() * kernels.blur3x3(z.to(DEV), 0).float()).sum())
    assert abs(lhs - rhs) < 2e-2 * (abs(lhs) + 1.0), (lhs, rhs)
    # the autograd Function differentiates twice (R1 goes through the discriminator's blur)
    xd = x.to(DEV).requires_grad_(True)
    seed = y[:, :8, :8].to(DEV).contiguous().requires_grad_(True)
    out = Fn.blur(xd)
    gx, = torch.autograd.grad([out], [xd], [seed], create_graph=True)       # = blur(seed): linear in the seed
    gx.float().pow(2).sum().backward()                                       # d/d seed goes through blur once more
    ref = 2 * kernels.blur3x3(kernels.blur3x3(seed.detach(), 0), 0).float()
    torch.testing.assert_close(seed.grad.float(), ref, rtol=5e-2, atol=5e-2 * float(ref.abs().max()))


def test_networks_with_enable_blur_match_oracle():
    """enable_blur (net.py:140-141,422-423): generator c0(blur(upscale2x(h))), discriminator blur(downscale2x(h))."""
    from oracle import camera
    from rgbd_gan_amd.net import Discriminator, StyleGANGenerator
    gp = nets.init_stylegan(256, seed=0)
    dp = nets.init_discriminator(256, seed=1)
    gen = StyleGANGenerator(256, rgbd=True, enable_blur=True)
    dis = Discriminator(256, res=True, enable_blur=True)
    gen.load_state_dict(gp)
    dis.load_state_dict(dp)
    rng = np.random.RandomState(0)
    zh = nets.make_hidden(1, 256, rng)
    z = np.concatenate([zh, zh])
    np.random.seed(1)
    t9 = camera.theta9(camera.PosePrior(0.3054, 1.0472, 0).sample(2))

    def rel(a, b):
        return float((a.double() - b.double()).norm() / b.double().norm())
    for stage in (6.0, 7.5):
        with torch.no_grad():
            ref = nets.stylegan_generator(gp, z, stage, t9, enable_blur=True)
            plain = nets.stylegan_generator(gp, z, stage, t9)
            got = gen(z, stage, t9).cpu()
        assert rel(got[:, :3], ref[:, :3]) < 4e-2 < rel(plain[:, :3], ref[:, :3])       # the blur is really applied
        x = ref[:, :3].contiguous()
        xr = x.clone().requires_grad_(True)
        yr = nets.discriminator(dp, xr, stage, enable_blur=True)
        yr.sum().backward()
        xd = x.to(DEV).requires_grad_(True)
        yd = dis(xd, stage)
        yd.sum().backward()
        assert float((yd.detach().cpu() - yr.detach()).abs().max()) < 4e-2 * max(1.0, float(yr.detach().abs().max()))
        assert rel(xd.grad.cpu(), xr.grad) < 0.15


@pytest.mark.parametrize("alpha_on_device", [False, True])
def test_fade_in_kernels(alpha_on_device):
    """Odd progressive stages (net.py:283-290,490-497): generator planes blend, discriminator feature blend, image
    down-scaling -- against torch autograd on the reference's formulas, alpha from the host or from a device scalar."""
    from rgbd_gan_amd import functional as Fn
    a = 0.3125
    alpha = torch.tensor(a, device=DEV) if alpha_on_device else a
    g = torch.Generator().manual_seed(4)
    lo, hi = torch.randn(3, 4, 8, 8, generator=g), torch.randn(3, 4, 16, 16, generator=g)
    lr, hr = lo.clone().requires_grad_(True), hi.clone().requires_grad_(True)
    ref = (1 - a) * nets.up2(lr) + a * hr
    gout = torch.randn(3, 4, 16, 16, generator=g)
    ref.backward(gout)
    ld, hd = lo.to(DEV).requires_grad_(True), hi.to(DEV).requires_grad_(True)
    out = Fn.fade_planes(ld, hd, alpha)
    out.backward(gout.to(DEV))
    torch.testing.assert_close(out.detach().cpu(), ref.detach(), rtol=1e-6, atol=1e-6)
    torch.testing.assert_close(ld.grad.cpu(), lr.grad, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(hd.grad.cpu(), hr.grad, rtol=1e-6, atol=1e-6)
    # feature blend, twice differentiable
    p, q = (torch.randn(2, 8, 8, 64, generator=g).to(torch.bfloat16) for _ in range(2))
    pd, qd = p.to(DEV).requires_grad_(True), q.to(DEV).requires_grad_(True)
    h = Fn.lerp(pd, qd, alpha)
    torch.testing.assert_close(h.float().cpu(), ((1 - a) * p.float() + a * q.float()).to(torch.bfloat16).float(), rtol=0, atol=1e-2)
    gy = torch.randn(2, 8, 8, 64, generator=g).to(torch.bfloat16).to(DEV).requires_grad_(True)
    gp, gq = torch.autograd.grad([h], [pd, qd], [gy], create_graph=True)
    torch.testing.assert_close(gp.detach().float(), ((1 - a) * gy.detach().float()).to(torch.bfloat16).float(), rtol=0, atol=1e-2)
    torch.testing.assert_close(gq.detach().float(), (a * gy.detach().float()).to(torch.bfloat16).float(), rtol=0, atol=1e-2)
    (gp.float().sum() + 2 * gq.float().sum()).backward()      # through _LerpSplit.backward: d/d gy = (1-a) + 2a
    torch.testing.assert_close(gy.grad.float(), torch.full_like(gy.grad.float(), (1 - a) + 2 * a), rtol=0, atol=1e-2)
    # image down-scaling and its adjoint
    x = torch.randn(2, 3, 16, 16, generator=g)
    xd = x.to(DEV).requires_grad_(True)
    y = Fn.avg_pool2_planes(xd)
    torch.testing.assert_close(y.detach().cpu(), F.avg_pool2d(x, 2, 2), rtol=1e-6, atol=1e-6)
    gy2 = torch.randn(2, 3, 8, 8, generator=g)
    seed = gy2.to(DEV).requires_grad_(True)
    gx, = torch.autograd.grad([y], [xd], [seed], create_graph=True)
    xr = x.clone().requires_grad_(True)
    F.avg_pool2d(xr, 2, 2).backward(gy2)
    torch.testing.assert_close(gx.detach().cpu(), xr.grad, rtol=1e-6, atol=1e-6)
    wgt = torch.randn(2, 3, 16, 16, generator=g).to(DEV)
    (gx * wgt).sum().backward()                                # d/d seed of <unpool(seed), wgt> = avg_pool(wgt)
    torch.testing.assert_close(seed.grad.cpu(), F.avg_pool2d(wgt.cpu(), 2, 2), rtol=1e-6, atol=1e-6)
