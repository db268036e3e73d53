"""BASELINE configuration 5's networks: the 256 x 256 block the reference keeps commented out (net.py:181-183,192-194,
437-452: ch//8 channels), max_resolution=256, ch=512 -- forward / input-gradient parity with the oracle at stage 12 and
training steps through RGBDUpdater, on the bf16 convs and on the MXFP8 ones (`conv_dtype: mxfp8`: fprop / dgrad of the 3x3
convolutions on e4m3 operands with E8M0 block scales, csrc/mxfp8.hip; kernel-level parity with the fp8-emulating oracle is
tests/test_mxfp8_gpu.py -- here the networks as wholes against the bf16 engine and the fp32 oracle)."""
import numpy as np
import pytest
import torch

from oracle import camera, nets

pytestmark = pytest.mark.gpu
CH = 512


def rel_err(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a - b).norm() / (b.norm() + 1e-30))


def cosine(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a @ b) / (a.norm() * b.norm() + 1e-30))


def _models():
    from rgbd_gan_amd.net import Discriminator, StyleGANGenerator
    gp = nets.init_stylegan(CH, seed=0, max_resolution=256)
    dp = nets.init_discriminator(CH, seed=1, max_resolution=256)
    gen = StyleGANGenerator(CH, rgbd=True, max_resolution=256)
    dis = Discriminator(CH, res=True, max_resolution=256)
    assert gen.gen.max_stage == 19 and dis.max_stage == 19
    assert gen.gen.chans[6] == (64, 128) and dis.chans[6] == (64, 128)
    gen.load_state_dict(gp)
    dis.load_state_dict(dp)
    return gp, dp, gen, dis


@pytest.mark.parametrize("stage", [12.0, 11.5])
def test_256px_generator_and_discriminator_match_oracle(stage):
    gp, dp, gen, dis = _models()
    rng = np.random.RandomState(1)
    zh = nets.make_hidden(1, CH, rng)
    z = np.concatenate([zh, zh])
    np.random.seed(2)
    t9 = camera.theta9(camera.PosePrior(0.3054, 1.0472, 0).sample(2))
    with torch.no_grad():
        ref = nets.stylegan_generator(gp, z, stage, t9)
        got = gen(z, stage, t9).cpu()
    assert tuple(got.shape) == tuple(ref.shape) == (2, 4, 256, 256)
    assert rel_err(got[:, :3], ref[:, :3]) < 4e-2
    torch.testing.assert_close(got[:, 3], ref[:, 3], atol=1e-5, rtol=1e-5)
    g = torch.Generator().manual_seed(3)
    x = torch.rand(2, 3, 256, 256, generator=g) * 2 - 1
    xr = x.clone().requires_grad_(True)
    yr = nets.discriminator(dp, xr, stage)
    yr.sum().backward()
    xd = x.cuda().requires_grad_(True)
    yd = dis(xd, stage)
    yd.sum().backward()
    scale = float(yr.detach().abs().max())
    assert float((yd.detach().cpu() - yr.detach()).abs().max()) < 4e-2 * max(scale, 1.0)
    assert rel_err(xd.grad.cpu(), xr.grad) < 0.15, rel_err(xd.grad.cpu(), xr.grad)
    assert cosine(xd.grad.cpu(), xr.grad) > 0.99, cosine(xd.grad.cpu(), xr.grad)


def test_256px_training_steps_replay_from_graphs():
    """RGBDUpdater at stage 12 (256 x 256, rotation + occlusion loss, R1) on the stylegan config with ch=512 and
    max_resolution=256: finite losses, parameters move, the step is replayed from HIP graphs."""
    import os
    from rgbd_gan_amd.training import DeviceImageIterator, build_training
    from rgbd_gan_amd.utils import yaml_utils
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = yaml_utils.load(os.path.join(root, "configs", "stylegan_shapenet_car.yml"))
    cfg.ch, cfg.max_resolution, cfg.max_stage, cfg.batchsize = 512, 256, 13, 4
    images = np.random.RandomState(0).randint(0, 256, (16, 3, 256, 256)).astype("uint8")
    it = DeviceImageIterator(images, 4, "cuda:0", seed=0)
    gen, dis, opt, upd = build_training(cfg, "cuda:0", iterator=it, fixed_stage=12.0)
    upd.iteration = 200000              # past start_rotation / start_occlusion_aware
    w0 = gen.gen.store.params["blocks/6/c1/c/W"].clone()
    d0 = dis.store.params["blocks/6/c0/c/W"].clone()
    for _ in range(5):
        upd.update()
    torch.cuda.synchronize()
    obs = {k: float(v) for k, v in upd.observation.items() if torch.is_tensor(v) or isinstance(v, (int, float))}
    assert obs["image_size"] == 256 and upd.graphs_in_use
    for k in ("gen/loss_adv", "gen/loss_rotate", "dis/loss_adv", "dis/loss_gp"):
        assert np.isfinite(obs[k]), (k, obs)
    assert not torch.equal(gen.gen.store.params["blocks/6/c1/c/W"], w0)
    assert not torch.equal(dis.store.params["blocks/6/c0/c/W"], d0)


def test_256px_networks_on_mxfp8_convs_stay_close_to_the_bf16_engine_and_the_oracle():
    """Same weights, same inputs, conv_dtype bf16 vs mxfp8.  What the format costs on N(0,1)-initialised networks 13 / 14 conv
    layers deep -- both operands of every product carry three mantissa bits, i.e. ~3.6 % rms each, ~5 % per layer on a dot
    product, adding in quadrature over the depth -- as measured here: the generator's RGB output moves by 15 % relative L2
    against the bf16 engine (and against the fp32 oracle), the discriminator's logits by 6.5 % of their scale, its input
    gradient by 40 % relative L2 at cosine 0.92.  The bounds below are those figures with headroom; the kernel-level test
    (tests/test_mxfp8_gpu.py) is where the arithmetic is pinned to one bf16 rounding.  The depth channel is produced by a
    1x1 convolution and the fp32 depth head and keeps its agreement with the oracle."""
    from rgbd_gan_amd import functional as Fn
    gp, dp, gen, dis = _models()
    rng = np.random.RandomState(1)
    zh = nets.make_hidden(1, CH, rng)
    z = np.concatenate([zh, zh])
    np.random.seed(2)
    t9 = camera.theta9(camera.PosePrior(0.3054, 1.0472, 0).sample(2))
    g = torch.Generator().manual_seed(3)
    x = torch.rand(2, 3, 256, 256, generator=g) * 2 - 1
    out = {}
    for dt in ("bf16", "mxfp8"):
        Fn.set_conv_dtype(dt)
        with torch.no_grad():
            img = gen(z, 12.0, t9).cpu()
        xd = x.cuda().requires_grad_(True)
        yd = dis(xd, 12.0)
        yd.sum().backward()
        out[dt] = (img, yd.detach().cpu(), xd.grad.cpu())
    from rgbd_gan_amd import _lib
    with torch.no_grad():
        ref = nets.stylegan_generator(gp, z, 12.0, t9)
    (ib, yb, gb), (im, ym, gm) = out["bf16"], out["mxfp8"]
    e_img, e_ref = rel_err(im[:, :3], ib[:, :3]), rel_err(im[:, :3], ref[:, :3])
    e_y = float((ym - yb).abs().max()) / max(float(yb.abs().max()), 1.0)
    c_g, e_g = cosine(gm, gb), rel_err(gm, gb)
    print(f"mxfp8 vs bf16: generator rel L2 {e_img:.3e} (vs fp32 oracle {e_ref:.3e}), logits {e_y:.3e}, "
          f"dD/dx cosine {c_g:.4f} rel L2 {e_g:.3e}")
    assert not torch.equal(im, ib), "the mxfp8 switch changed nothing: the fp8 kernels did not run"
    assert e_img < 0.25 and e_ref < 0.25
    assert e_y < 0.15
    assert c_g > 0.85 and e_g < 0.6
    torch.testing.assert_close(im[:, 3], ref[:, 3], atol=1e-5, rtol=1e-5)


def test_256px_training_step_on_mxfp8_convs():
    """(B = 16: the benched per-GPU batch of configuration 5.)  `conv_dtype: mxfp8` through build_training: the step is replayed from graphs, the fp8 kernels are the ones that ran
    (launch profile of an eager step), losses are finite, parameters move, and the step's parameter gradients keep the
    direction of the bf16 step's from the same weights and inputs (flat-buffer cosine; measured: mapping 0.97, generator 0.84,
    discriminator 0.99)."""
    import os
    from rgbd_gan_amd import functional as Fn, kernels
    from rgbd_gan_amd.training import DeviceImageIterator, build_training
    from rgbd_gan_amd.utils import yaml_utils
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    grads = {}
    for dt in ("bf16", "mxfp8"):
        cfg = yaml_utils.load(os.path.join(root, "configs", "stylegan_shapenet_car.yml"))
        cfg.ch, cfg.max_resolution, cfg.max_stage, cfg.batchsize, cfg.conv_dtype = 512, 256, 13, 16, dt
        images = np.random.RandomState(0).randint(0, 256, (32, 3, 256, 256)).astype("uint8")
        it = DeviceImageIterator(images, 16, "cuda:0", seed=0)
        np.random.seed(5)
        torch.manual_seed(5)
        gen, dis, opt, upd = build_training(cfg, "cuda:0", iterator=it, fixed_stage=12.0, nan_check_interval=0)
        assert Fn.conv_dtype() == dt
        upd.iteration = 200000
        upd.use_graphs = False
        with kernels.launch_profile() as prof:
            upd.update()                              # ChainerMN-free single process: the first update steps
        names = set(prof.summary())
        assert any("mxfp8" in n for n in names) == (dt == "mxfp8"), names
        if dt == "mxfp8":
            assert "quantize_mx8_kernel" in names
        torch.cuda.synchronize()
        grads[dt] = {k: o.store.grad.clone() for k, o in opt.items()}
        if dt == "mxfp8":
            upd.use_graphs = True
            w0 = gen.gen.store.params["blocks/5/c1/c/W"].clone()
            for _ in range(4):
                upd.update()
            torch.cuda.synchronize()
            obs = {k: float(v) for k, v in upd.observation.items() if torch.is_tensor(v) or isinstance(v, (int, float))}
            assert upd.graphs_in_use
            for k in ("gen/loss_adv", "gen/loss_rotate", "dis/loss_adv", "dis/loss_gp"):
                assert np.isfinite(obs[k]), (k, obs)
            assert not torch.equal(gen.gen.store.params["blocks/5/c1/c/W"], w0)
    for k in grads["bf16"]:
        c = cosine(grads["mxfp8"][k], grads["bf16"][k])
        print(f"flat gradient buffer {k}: cosine(mxfp8, bf16) = {c:.4f}, norm ratio "
              f"{float(grads['mxfp8'][k].norm() / grads['bf16'][k].norm()):.3f}")
        # (B = 16: every layer from 32x32 up has its 64 tiles and runs on fp8 -- at B = 4, this test's size in round 4, the
        # 32x32 layers stayed on bf16 and the generator's buffer kept 0.94.  What pins the arithmetic is the MXFP8-emulating
        # oracle, tests/test_model_gpu.py::test_full_training_step_matches_mx8_emulating_oracle; this is a tripwire.)
        assert c > (0.75 if k == "gen" else 0.9), (k, c)


def test_mxfp8_training_tracks_bf16_over_400_steps():
    """Does `conv_dtype: mxfp8` TRAIN?  400 steps at 128x128, ch 256, B = 16 (every 3x3 launch from 16x16 up on the fp8 kernels:
    MX8_MIN_TILES = 0) on structured real images (utils/synthetic.py: a discriminator with something to model -- on uniform noise
    the game is degenerate and chaotic, which is what made round 5's soak look like a divergence), from the same seed with bf16
    and with mxfp8 convolutions, graphs and two streams.  The 3-D consistency loss -- the term the whole method is about -- must
    come down with fp8 as it does with bf16: mean of `gen/loss_rotate` over the last 100 steps within 2x of bf16's (measured over
    3 seeds x 2000 steps, profiles/r06/soak_fp8_ablation.txt: 1.02x, every fp8 seed inside the bf16 seed spread), and below half
    of its own first-100-step mean; nothing non-finite."""
    import os
    from rgbd_gan_amd import kernels
    from rgbd_gan_amd.training import DeviceImageIterator, build_training
    from rgbd_gan_amd.utils import yaml_utils
    from rgbd_gan_amd.utils.synthetic import procedural_images
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    images = procedural_images(256, 128, seed=0)
    saved = kernels.MX8_MIN_TILES
    kernels.MX8_MIN_TILES = 0
    try:
        curves = {}
        for dt in ("bf16", "mxfp8"):
            cfg = yaml_utils.load(os.path.join(root, "configs", "stylegan_shapenet_car.yml"))
            cfg.conv_dtype = dt
            np.random.seed(0)
            torch.manual_seed(0)
            it = DeviceImageIterator(images, 16, "cuda:0", seed=0)
            gen, dis, opt, upd = build_training(cfg, "cuda:0", iterator=it, fixed_stage=10.0, nan_check_interval=0)
            upd.iteration = 200000
            vals = []
            for _ in range(400):
                upd.update()
                vals.append(upd.observation["gen/loss_rotate"].detach().reshape(1).clone())   # read once, behind the loop
            vals = torch.cat(vals).cpu().numpy()
            assert np.isfinite(vals).all(), dt
            for link in (gen, dis):
                for _, store in link.stores:
                    assert bool(torch.isfinite(store.flat).all()), dt
            curves[dt] = (float(vals[:100].mean()), float(vals[-100:].mean()))
            del gen, dis, opt, upd, it
        print("gen/loss_rotate, mean of the first / last 100 of 400 steps:", curves)
        (b0, b1), (m0, m1) = curves["bf16"], curves["mxfp8"]
        assert b1 < 0.5 * b0, curves                        # the run is long enough for the loss to come down at all
        assert m1 < 0.5 * m0, curves
        assert m1 < 2.0 * b1, curves
    finally:
        kernels.MX8_MIN_TILES = saved
