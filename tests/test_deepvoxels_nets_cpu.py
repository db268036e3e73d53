"""CPU checks for the DeepVoxels generator path: known answers of the oracle
restatement (oracle/deepvoxels_nets.py) and the host-side layer rearrangements the HIP engine relies on
(rgbd_gan_amd/deepvoxels_generator.py: depth taps / stride-2 taps folded into channels), checked against torch's own
conv3d / strided conv2d in fp32."""
import numpy as np
import torch
import torch.nn.functional as F

from oracle import camera, deepvoxels as dv, deepvoxels_nets as dvn, nets


def test_equalized_conv3d_scale_uses_ksize_squared():
    """pggan.py:31: inv_c = sqrt(2) * sqrt(1 / (in_ch * ksize**2)) even for a 3x3x3 kernel."""
    p = {"c/c/W": torch.ones(1, 2, 3, 3, 3)}
    y = dvn.eq_conv3d(torch.ones(1, 2, 5, 5, 5), p, "c", 1)
    inv_c = np.sqrt(2) * np.sqrt(1 / (2 * 9))
    assert abs(float(y[0, 0, 2, 2, 2]) - 54 * inv_c) < 1e-4          # 27 taps x 2 channels
    assert abs(float(y[0, 0, 0, 0, 0]) - 16 * inv_c) < 1e-4          # corner: 8 taps x 2 channels


def test_parameter_inventory():
    p = dvn.init_deepvoxels_generator(256)
    assert p["voxel_gen/net/0/W"].shape == (64, 4, 4, 4) and float(p["voxel_gen/net/0/W"].min()) == 1.0
    assert p["voxel_gen/net/2/c0/c/W"].shape == (32, 64, 3, 3, 3)
    assert p["voxel_gen/out/c/W"].shape == (32, 32, 1, 1, 1)
    assert p["deepvoxel/occlusion_net/occlusion/0/net/1/c/W"].shape == (4, 33, 1, 1, 1)
    assert p["style_generator/c0/c/W"].shape == (512, 32, 4, 4)
    assert p["style_generator/c6/c/W"].shape == (256, 1024, 3, 3)
    assert p["style_generator/c7/c/W"].shape == (3, 288, 3, 3)
    assert float(p["style_generator/s4/s/c/b"].min()) == 1.0 and float(p["style_generator/s4/b/c/b"].abs().max()) == 0.0
    assert len(dvn.init_mapping3d(256)) == 16


def test_generator_shapes_and_depth_range():
    p, pm = dvn.init_deepvoxels_generator(256, seed=1), dvn.init_mapping3d(256, seed=0)
    g = torch.Generator().manual_seed(3)
    z, z2 = torch.randn(1, 256, generator=g), torch.randn(1, 256, generator=g)
    cams = camera.camera_matrices(np.array([[0.1, 0.5, 0, 0, 0, 0]], dtype="float32"))
    with torch.no_grad():
        out, voxel, feats = dvn.deepvoxels_generator(p, pm, z, z2, cams, return_parts=True)
    assert out.shape == (1, 4, 64, 64) and voxel.shape == (1, 32, 32, 32, 32) and feats.shape == (1, 32, 64, 64)
    fr = dv.Frustum()
    lo = (-0.5 + 0.5) * fr.depth * fr.voxel_size + fr.near_plane
    hi = (0.5 + 0.5) * fr.depth * fr.voxel_size + fr.near_plane
    assert float(out[:, 3].min()) >= lo - 1e-5 and float(out[:, 3].max()) <= hi + 1e-5
    assert torch.isfinite(out).all()


def test_mapping3d_is_scale_invariant():
    """feature_vector_normalization first (deepvoxels_generator.py:65): w(z) == w(3 z)."""
    pm = dvn.init_mapping3d(64, seed=2)
    z = torch.randn(3, 64, generator=torch.Generator().manual_seed(0))
    torch.testing.assert_close(dvn.mapping3d(pm, z), dvn.mapping3d(pm, 3 * z), rtol=1e-4, atol=1e-5)


def test_depth_tap_folding_equals_conv3d():
    from rgbd_gan_amd.deepvoxels_generator import fold_conv3d_weight, fold_depth_taps
    g = torch.Generator().manual_seed(1)
    x = torch.randn(2, 24, 5, 6, 7, generator=g)                      # NCDHW
    W = torch.randn(40, 24, 3, 3, 3, generator=g)
    ref = F.conv3d(x, W, padding=1)
    xn = F.pad(x.permute(0, 2, 3, 4, 1), (0, 64 - 24))                 # (B,D,H,W,64)
    y = F.conv2d(fold_depth_taps(xn).permute(0, 3, 1, 2), fold_conv3d_weight(W), padding=1)      # (B*D,64,H,W)
    y = y.reshape(2, 5, 64, 6, 7).permute(0, 2, 1, 3, 4)
    torch.testing.assert_close(y[:, :40], ref, rtol=1e-4, atol=1e-4)
    assert float(y[:, 40:].abs().max()) == 0.0                          # padded output channels stay exactly zero


def test_stride2_tap_folding_equals_strided_conv():
    from rgbd_gan_amd.deepvoxels_generator import fold_4x4s2, fold_4x4s2_weight
    g = torch.Generator().manual_seed(2)
    x = torch.randn(2, 8, 12, 12, generator=g)
    W = torch.randn(16, 8, 4, 4, generator=g)
    ref = F.conv2d(x, W, stride=2, padding=1)
    y = F.conv2d(fold_4x4s2(x.permute(0, 2, 3, 1)).permute(0, 3, 1, 2), fold_4x4s2_weight(W))
    torch.testing.assert_close(y, ref, rtol=1e-4, atol=1e-4)


def test_focal_generator_loss_known_answer():
    from oracle import step
    y = torch.zeros(4, 1)
    assert abs(float(step.loss_gen_adv_focal(y, 2.0)) - np.log(2) * 0.25) < 1e-6      # softplus(0) * sigmoid(0)^2
    assert abs(float(step.loss_gen_adv_focal(y, 0.0)) - float(step.loss_gen_adv(y))) < 1e-7


def test_deepvoxels_step_runs_and_updates_all_three_optimizers():
    """One oracle step on a single view pair: tiled latents for G, fresh latents for D through the UPDATED
    generator, three optimizers."""
    from oracle import step
    ch = 256
    gp, mp = dvn.init_deepvoxels_generator(ch, seed=1), dvn.init_mapping3d(ch, seed=0)
    dp = nets.init_discriminator(ch, seed=2)
    gpl = {k: v.clone().requires_grad_(True) for k, v in gp.items()}
    mpl = {k: v.clone().requires_grad_(True) for k, v in mp.items()}
    dpl = {k: v.clone().requires_grad_(True) for k, v in dp.items()}
    opt = {"map": step.ChainerAdam(mpl, 1e-5), "gen": step.ChainerAdam(gpl, 1e-3), "dis": step.ChainerAdam(dpl, 3e-3)}
    g = torch.Generator().manual_seed(5)
    zh, zh2 = torch.randn(1, ch, generator=g), torch.randn(1, ch, generator=g)
    zd, zd2 = torch.randn(2, ch, generator=g), torch.randn(2, ch, generator=g)
    np.random.seed(3)
    thetas = camera.PosePrior(0.3054, 3.1415, 0, uniform=True).sample(2)
    x_real = np.random.RandomState(1).rand(2, 3, 128, 128).astype("float32") * 2 - 1
    cfg = dict(lambda_gp=1.0, lambda_depth=10, depth_min=0.6, focal_loss_gamma=2.0, start_rotation=0)
    out = step.deepvoxels_step(gpl, mpl, dpl, opt, x_real, (torch.cat([zh, zh]), torch.cat([zh2, zh2]), zd, zd2), thetas,
                               cfg, iteration=5)
    assert all(np.isfinite(out[k]) for k in ("gen/loss_adv", "gen/loss_rotate", "dis/loss_adv", "dis/loss_gp"))
    assert opt["map"].t == opt["gen"].t == opt["dis"].t == 1
    moved = float((gpl["style_generator/c6/c/W"].detach() - gp["style_generator/c6/c/W"]).abs().max())
    assert 0 < moved <= 1e-3 * 1.001
    # unused parameters (noise scales) received zero gradient and did not move
    assert float((gpl["voxel_gen/net/1/n0/b/W"].detach() - gp["voxel_gen/net/1/n0/b/W"]).abs().max()) == 0.0
    # the discriminator-step fakes come from the updated generator with fresh latents
    assert out["x_fake_dis"].shape == (2, 4, 64, 64) and not torch.allclose(out["x_fake_dis"], out["x_fake"])
