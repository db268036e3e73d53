"""BASELINE configuration 5's networks: the 256 x 256 block the reference keeps commented out (net.py:181-183,192-194,
437-452: ch//8 channels), max_resolution=256, ch=512 -- forward / input-gradient parity with the oracle at stage 12 and
one training step through RGBDUpdater (bf16 MFMA convs; an fp8 conv kernel family is NOT built)."""
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
