"""Known-answer tests that pin the CPU oracle.

The reference has no tests of its own, so these are the pins.
"""
import math

import numpy as np
import torch

from oracle import camera, nets, step, warp_loss

FFHQ = "0,0,0,0,0,0,0,100000, 150000, 160000, 180000, 300000"


def test_stage_schedule_known_answers():
    si = camera.parse_stage_interval(FFHQ)
    want = {0: 6.0, 50000: 6.5, 100000: 7.0, 150000: 8.0, 170000: 9.5, 180000: 10.0}
    for it, st in want.items():
        assert abs(camera.stage_of(it, si, 11) - st) < 1e-12, it
    assert camera.stage_of(300000, si, 11) == 11 - 1e-8
    assert camera.stage_of(999999, si, 11) == 11 - 1e-8


def test_camera_matrix_identity_pose():
    M = camera.camera_matrices(np.zeros((2, 6), "float32"))
    want = np.diag([1, 1, -1, 1]).astype("float32")
    want[2, 3] = 1
    assert M.dtype == np.float32
    np.testing.assert_array_equal(M[0], want)


def test_camera_matrix_is_rigid():
    np.random.seed(3)
    th = np.random.uniform(-1, 1, (5, 6)).astype("float32")
    M = camera.camera_matrices(th)
    R = M[:, :3, :3]
    np.testing.assert_allclose(np.matmul(R, R.transpose(0, 2, 1)), np.broadcast_to(np.eye(3), R.shape), atol=1e-6)
    # the initial offset (0,0,1) is rotated with the frame, then the translation is added
    np.testing.assert_allclose(np.linalg.norm(M[:, :3, 3] - th[:, 3:], axis=1), 1.0, atol=1e-6)


def test_prior_pairs_and_draw_order():
    prior = camera.PosePrior(0.3054, 1.0472, 0)
    np.random.seed(2)
    th = prior.sample(8)
    assert th.shape == (8, 6) and th.dtype == np.float32
    np.random.seed(2)
    t1 = np.random.uniform(-1, 1, size=(4, 6))
    np.testing.assert_allclose(th[:4], (t1 * prior.camera_param_range).astype("float32"))
    # second half moves towards zero by at most 0.5 * range on rotating axes
    d = np.abs(th[4:] - th[:4])
    assert (d[:, 0] <= 0.5 * 0.3054 + 1e-6).all() and (d[:, 2] == 0).all()


def test_intrinsics_128():
    K, inv_K, p = warp_loss.intrinsics(128)
    np.testing.assert_array_equal(K, np.array([[256, 0, 64], [0, 256, 64], [0, 0, 1]], "float32"))
    assert p.shape == (3, 128 * 128)
    n = 5 * 128 + 9
    assert p[0, n] == 9 and p[1, n] == 5 and p[2, n] == 1


def test_downsize_identity_and_fade():
    x = torch.randn(2, 3, 128, 128)
    assert torch.equal(nets.downsize_real(x, 10.0), x)
    lo = nets.downsize_real(x, 8.0)
    assert lo.shape == (2, 3, 64, 64)
    mid = nets.downsize_real(x, 9.25)
    want = 0.75 * nets.up2(torch.nn.functional.avg_pool2d(x, 2, 2)) + 0.25 * x
    torch.testing.assert_close(mid, want)


def test_equal_cameras_constant_depth_gives_zero_loss():
    rng = np.random.RandomState(0)
    S = 16
    img = rng.uniform(-1, 1, (2, 4, S, S)).astype("float32")
    img[:, 3] = 1.0
    # zero pose: R = M0^T M0 = I and K K^-1 = I exactly in fp32 (powers of two)
    cam = camera.camera_matrices(np.zeros((2, 6), "float32"))
    out = warp_loss.forward_np(img, cam, img.copy(), cam.copy())
    interior = out["mask"].reshape(2, S, S)[:, :S - 1, :S - 1]
    assert interior.all()
    assert out["loss"] < 1e-5
    lt, _ = warp_loss.loss_torch(torch.from_numpy(img), cam, torch.from_numpy(img), cam)
    assert float(lt) < 1e-5


def test_row_plus_one_tap_quirk():
    """loss_functions.py:219: the '+1 row' taps read row u0, so the loss does not depend on how
    fractional the row coordinate is -- a pure vertical sub-pixel shift in (u) only changes
    which integer row is read."""
    rng = np.random.RandomState(1)
    S = 8
    img = rng.uniform(-1, 1, (1, 4, S, S)).astype("float32")
    img[:, 3] = 2.0
    zp = np.zeros((1, S * S, 3), "float32")
    jj, ii = np.meshgrid(np.arange(S), np.arange(S))
    for frac in (0.0, 0.25, 0.75):
        zp[0, :, 0] = (jj.reshape(-1) * 0.5 + 0.3) * 2.0
        zp[0, :, 1] = (ii.reshape(-1) * 0.5 + frac) * 2.0
        zp[0, :, 2] = 2.0
        warped, mask, (u0, v0, v1) = warp_loss._bilinear_np(img, zp)
        u = zp[0, :, 1] / 2.0
        v = zp[0, :, 0] / 2.0
        want = ((v0 + 1 - v)[:, None] * img[0][:, u0, v0].T + (v - v0)[:, None] * img[0][:, u0, v1].T) * mask[:, None]
        np.testing.assert_allclose(warped, want, atol=1e-6)


def test_np_and_torch_warp_loss_agree():
    rng = np.random.RandomState(5)
    S, b = 16, 3
    img = rng.uniform(-1, 1, (b, 4, S, S)).astype("float32")
    img_rot = rng.uniform(-1, 1, (b, 4, S, S)).astype("float32")
    img[:, 3] = rng.uniform(0.8, 1.2, (b, S, S))
    img_rot[:, 3] = rng.uniform(0.8, 1.2, (b, S, S))
    th = rng.uniform(-0.2, 0.2, (2 * b, 6)).astype("float32")
    th[:, 3:] = 0
    cams = camera.camera_matrices(th)
    for occ in (False, True):
        ref = warp_loss.forward_np(img, cams[:b], img_rot, cams[b:], occlusion_aware=occ, lambda_geometric=2.0)
        lt, _ = warp_loss.loss_torch(torch.from_numpy(img), cams[:b], torch.from_numpy(img_rot), cams[b:],
                                     occlusion_aware=occ, lambda_geometric=2.0)
        assert abs(float(lt) - ref["loss"]) < 1e-5 * max(1.0, ref["loss"])
        assert ref["mask"].any() and not ref["mask"].all()


def test_adain_constant_input_is_shift_only():
    x = torch.full((2, 3, 4, 4), 1.5)
    s = torch.randn(2, 3)
    t = torch.randn(2, 3)
    out = nets.adain(x, s, t)
    torch.testing.assert_close(out, t.reshape(2, 3, 1, 1).expand_as(out))


def test_depth_head_initial_value():
    p = nets.init_stylegan(ch=16, seed=0)
    assert float(p["gen/outs/5/c/W"][-1].abs().max()) == 0.0
    assert abs(float(p["gen/outs/5/c/b"][-1]) - math.log(math.e - 1)) < 1e-6
    z = nets.make_hidden(2, 16, np.random.RandomState(1))
    assert z.shape == (2, 32, 1, 1)
    t9 = camera.theta9(np.zeros((2, 6), "float32"))
    x = nets.stylegan_generator(p, z, 10.0, t9)
    assert x.shape == (2, 4, 128, 128)
    # depth head starts at 1/(softplus(log(e-1)) + 1e-4) = 1/(1+1e-4)
    torch.testing.assert_close(x[:, 3], torch.full_like(x[:, 3], 1 / (1 + 1e-4)), atol=1e-5, rtol=0)


def test_generator_stage_shapes_and_fade():
    p = nets.init_stylegan(ch=8, seed=0)
    z = nets.make_hidden(2, 8, np.random.RandomState(1))
    t9 = camera.theta9(np.random.RandomState(2).uniform(-1, 1, (2, 6)).astype("float32"))
    sizes = {6.0: 32, 7.5: 64, 8.0: 64, 9.0: 128, 10.0: 128}
    for st, sz in sizes.items():
        assert nets.stylegan_generator(p, z, st, t9).shape == (2, 4, sz, sz)
    d = nets.init_discriminator(ch=8, seed=1)
    for st, sz in sizes.items():
        y, feat = nets.discriminator(d, torch.randn(2, 3, sz, sz), st, return_hidden=True)
        assert y.shape == (2, 1) and feat.shape[2] == 32


def test_dcgan_generator_shapes():
    p = nets.init_dcgan(in_ch=8, ch=16, seed=0)
    z = nets.make_hidden_dcgan(2, 8, np.random.RandomState(1))
    t9 = camera.theta9(np.zeros((2, 6), "float32"))
    assert nets.dcgan_generator(p, z, 8.0, t9).shape == (2, 4, 64, 64)
    assert nets.dcgan_generator(p, z, 9.5, t9).shape == (2, 4, 128, 128)


def test_chainer_adam_closed_form_beta1_zero():
    w = torch.tensor([1.0, -2.0, 3.0], requires_grad=True)
    opt = step.ChainerAdam({"w": w}, alpha=0.1, beta1=0.0, beta2=0.999, eps=1e-8, clip=5.0)
    w.grad = torch.tensor([0.5, -0.25, 0.0])
    opt.update()
    g = np.array([0.5, -0.25, 0.0])
    v = 0.001 * g * g
    want = np.array([1.0, -2.0, 3.0]) - 0.1 * math.sqrt(1 - 0.999) * g / (np.sqrt(v) + 1e-8)
    np.testing.assert_allclose(w.detach().numpy(), want, rtol=1e-6)
    # clipping: norm 50 -> grads scaled by 0.1
    w.grad = torch.tensor([30.0, 40.0, 0.0])
    n = opt.update()
    assert abs(n - 50.0) < 1e-9
    np.testing.assert_allclose(opt.m["w"].numpy(), [3.0, 4.0, 0.0], rtol=1e-6)


def test_r1_matches_finite_differences_on_tiny_net():
    torch.manual_seed(0)
    d = {k: v.double().requires_grad_(True) for k, v in nets.init_discriminator(ch=4, seed=1).items()}
    x = torch.randn(2, 3, 8, 8, dtype=torch.double, requires_grad=True)
    y = nets.discriminator(d, x, 2.0)
    gp = step.r1_penalty(y, x, 1.0)
    gp.backward()
    name = "blocks/1/c1/c/W"
    g_auto = d[name].grad[0, 0, 1, 1].item()
    eps = 1e-5

    def val(delta):
        with torch.no_grad():
            d[name][0, 0, 1, 1] += delta
        xx = x.detach().clone().requires_grad_(True)
        out = step.r1_penalty(nets.discriminator(d, xx, 2.0), xx, 1.0).item()
        with torch.no_grad():
            d[name][0, 0, 1, 1] -= delta
        return out

    g_fd = (val(eps) - val(-eps)) / (2 * eps)
    assert abs(g_auto - g_fd) < 1e-5 * max(1.0, abs(g_fd))
