"""One training step (RGBDUpdater.update_core) of the reference, restated on torch-CPU.

Test infrastructure only (see oracle/__init__.py).  PARITY UNPINNED.
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import camera, nets, warp_loss
from . import camera as camera_mod


def loss_gen_adv(y_fake):
    """loss_functions.py:11-14 with focal_loss_gamma = 0 (RGBDUpdater passes none)."""
    return F.softplus(-y_fake).sum() / y_fake.numel()


def loss_dis_adv(y_fake, y_real):
    """loss_functions.py:17-28."""
    return F.softplus(y_fake).sum() / y_fake.numel() + F.softplus(-y_real).sum() / y_real.numel()


def r1_penalty(y_real, x_real, lambda_gp):
    """updater.py:414-418: g = d sum(y_real)/d x_real (graph kept);
    lambda * mean_b( sqrt(sum g_b^2)^2 )."""
    g, = torch.autograd.grad(y_real.sum(), x_real, create_graph=True)
    norm = torch.sqrt((g ** 2).sum(dim=(1, 2, 3)))
    return lambda_gp * ((norm - 0.0) ** 2).sum() / norm.numel()


def depth_hinge(x_fake, depth_min, lambda_depth):
    """updater.py:357-359."""
    return torch.mean(F.relu(depth_min - x_fake[:, -1]) ** 2) * lambda_depth


class ChainerAdam:
    """chainer.optimizers.Adam (v7) + GradientClipping hook, as train_rgbd.py:151-161 sets it up.

    update(): zero-fill missing grads, clip by the global L2 norm of this optimizer's
    own parameters (rate = threshold / norm, applied when rate < 1), then per parameter
        m += (1-b1)(g-m);  v += (1-b2)(g*g-v);
        p -= alpha * sqrt(1-b2^t)/(1-b1^t) * m / (sqrt(v) + eps)      (eps outside the correction)
    `alpha_override` maps parameter names to their own alpha (train_rgbd.py:323-327).
    """

    def __init__(self, params, alpha, beta1=0.0, beta2=0.999, eps=1e-8, clip=5.0, alpha_override=None):
        self.params = params  # dict name -> tensor (leaf, requires_grad)
        self.alpha, self.beta1, self.beta2, self.eps, self.clip = alpha, beta1, beta2, eps, clip
        self.alpha_override = alpha_override or {}
        self.t = 0
        self.m = {k: torch.zeros_like(v) for k, v in params.items()}
        self.v = {k: torch.zeros_like(v) for k, v in params.items()}

    @torch.no_grad()
    def update(self):
        self.t += 1
        grads = {k: (p.grad if p.grad is not None else torch.zeros_like(p)) for k, p in self.params.items()}
        sq = sum(float((g.double() ** 2).sum()) for g in grads.values())
        norm = np.sqrt(sq)
        rate = self.clip / norm if norm > 0 else np.inf
        fix1 = 1.0 - self.beta1 ** self.t
        fix2 = 1.0 - self.beta2 ** self.t
        for k, p in self.params.items():
            g = grads[k] * rate if rate < 1 else grads[k]
            self.m[k] += (1 - self.beta1) * (g - self.m[k])
            self.v[k] += (1 - self.beta2) * (g * g - self.v[k])
            alpha = self.alpha_override.get(k, self.alpha)
            alpha_t = alpha * np.sqrt(fix2) / fix1
            p -= alpha_t * self.m[k] / (torch.sqrt(self.v[k]) + self.eps)
        return norm

    def zero_grad(self):
        for p in self.params.values():
            p.grad = None


def rgbd_step(gen_params, dis_params, opt, x_real_full, z, thetas, stage, cfg, iteration,
              architecture="stylegan", camera=True):
    """updater.py:274-448 on explicit inputs.

    gen_params / dis_params: dicts of leaf tensors (requires_grad).
    opt: {'map': ChainerAdam, 'gen': ChainerAdam, 'dis': ChainerAdam} ('map' absent for dcgan).
    x_real_full: (B,3,128,128) float32 in [-1,1]; z: the *repeated* latent batch (B, ...);
    thetas: (B,6) float32 from the prior.  cfg: dict with lambda_gp, lambda_depth, depth_min,
    lambda_geometric (or None), lambda_rotate (or None), start_rotation, start_occlusion_aware.
    Returns a dict of the reported scalars plus x_fake.
    """
    B = x_real_full.shape[0]
    if camera:
        use_rotate = iteration > cfg["start_rotation"]
        cams = camera_mod.camera_matrices(thetas)
        t9 = torch.from_numpy(camera_mod.theta9(thetas))
    else:       # RGBUpdater.update_core (updater.py:504-589): gen(z, stage), adversarial + R1 terms only
        use_rotate, cams, t9 = False, None, None
    for o in opt.values():
        o.zero_grad()
    x_real = nets.downsize_real(torch.as_tensor(x_real_full), stage).detach()
    image_size = x_real.shape[2]

    if architecture == "stylegan":
        x_fake = nets.stylegan_generator(gen_params, z, stage, t9, rgbd=camera)
    else:
        x_fake = nets.dcgan_generator(gen_params, z, stage, t9, rgbd=camera)
    rf = bool(cfg.get("rotate_feature")) and use_rotate          # updater.py:345,423
    if rf:
        y_fake, feat = nets.discriminator(dis_params, x_fake[:, :3], stage, return_hidden=True)
    else:
        y_fake = nets.discriminator(dis_params, x_fake[:, :3], stage)
    loss_adv_g = loss_gen_adv(y_fake)
    loss_gen = loss_adv_g
    out = {"gen/loss_adv": float(loss_adv_g.detach())}
    if use_rotate:
        lam_geo = cfg.get("lambda_geometric") or 3
        loss_rot, _ = warp_loss.loss_torch(x_fake[:B // 2], cams[:B // 2], x_fake[B // 2:], cams[B // 2:],
                                           occlusion_aware=iteration >= cfg["start_occlusion_aware"],
                                           lambda_geometric=lam_geo)
        if rf:
            loss_rot = loss_rot + feature_rotation_loss(feat, x_real, cams, iteration >= cfg["start_occlusion_aware"], lam_geo)[0]
        if cfg["lambda_depth"] > 0:
            loss_rot = loss_rot + depth_hinge(x_fake, cfg["depth_min"], cfg["lambda_depth"])
        out["gen/loss_rotate"] = float(loss_rot.detach())
        lam_rot = cfg.get("lambda_rotate") or 2
        lam_rot = lam_rot if image_size <= 128 else lam_rot * 2
        loss_gen = loss_gen + loss_rot * lam_rot
    loss_gen.backward()
    out["loss_gen_total"] = float(loss_gen.detach())
    if "map" in opt:
        out["norm_map"] = opt["map"].update()
    out["norm_gen"] = opt["gen"].update()
    opt["dis"].zero_grad()

    v_x_fake = x_fake.detach()[:, :3].clone().requires_grad_(rf)
    if rf:
        y_fake, feat = nets.discriminator(dis_params, v_x_fake, stage, return_hidden=True)
    else:
        y_fake = nets.discriminator(dis_params, v_x_fake, stage)
    x_real = x_real.clone().requires_grad_(True)
    y_real = nets.discriminator(dis_params, x_real, stage)
    loss_dis = loss_dis_adv(y_fake, y_real)
    out["dis/loss_adv_only"] = float(loss_dis.detach())
    sn = "blocks/0/c0/W_u" in dis_params                   # updater.py:414: `if not self.dis.sn and self.lambda_gp > 0`
    if cfg["lambda_gp"] > 0 and not sn:
        loss_gp = r1_penalty(y_real, x_real, cfg["lambda_gp"])
        out["dis/loss_gp"] = float(loss_gp.detach())
        loss_dis = loss_dis + loss_gp
    if rf:                                                 # updater.py:423-437
        lam_geo = cfg.get("lambda_geometric") or 3
        loss_rf, f257 = feature_rotation_loss(feat, x_real.detach(), cams, iteration >= cfg["start_occlusion_aware"], lam_geo)
        loss_dis = loss_dis - loss_rf
        if cfg["lambda_gp"] > 0 and not sn:
            g, = torch.autograd.grad([f257], [v_x_fake], [torch.ones_like(f257)], create_graph=True)   # chainer.grad seeds ones
            loss_dis = loss_dis + cfg["lambda_gp"] * ((torch.sqrt((g ** 2).sum(dim=(1, 2, 3))) - 0.0) ** 2).sum() / g.shape[0]
    out["dis/loss_adv"] = float(loss_dis.detach())
    loss_dis.backward()
    out["norm_dis"] = opt["dis"].update()
    out["x_fake"] = x_fake.detach()
    out["stage"], out["batch_size"], out["image_size"] = stage, B, image_size
    return out


def feature_rotation_loss(feat, x_real, cams, occlusion_aware, lambda_geometric):
    """updater.py:345-353 = 423-431: the warp loss with norm="l2" (updater.py:240) on the discriminator's hidden features of
    the two views, with the average-pooled LAST CHANNEL OF THE REAL BATCH appended as the "depth" channel (as written:
    x_real has three colour planes, so this is the pooled blue plane of unrelated real images)."""
    B = feat.shape[0]
    rate = x_real.shape[2] // feat.shape[2]
    depth = F.avg_pool2d(x_real[:, -1:], rate, rate)
    f = torch.cat([feat, depth], dim=1)
    loss, _ = warp_loss.loss_torch(f[:B // 2], cams[:B // 2], f[B // 2:], cams[B // 2:], occlusion_aware=occlusion_aware,
                                   lambda_geometric=lambda_geometric, norm="l2")
    return loss, f


def loss_gen_adv_focal(y_fake, gamma):
    """loss_functions.py:7-14: softplus(-y) * sigmoid(-y)^gamma, mean."""
    return (F.softplus(-y_fake) * torch.sigmoid(-y_fake) ** gamma).sum() / y_fake.numel()


def deepvoxels_step(gen_params, map_params, dis_params, opt, x_real_full, z_fake, thetas, cfg, iteration):
    """updater_deepvoxels.py:123-252 on explicit inputs.

    z_fake = (z, z2, z_dis, z2_dis): the tiled latents of the generator step and the fresh ones of the discriminator
    step, each (B,ch).  opt: {'map','gen','dis'} ChainerAdam.  cfg: lambda_gp, lambda_depth, depth_min,
    focal_loss_gamma, start_rotation, lambda_geometric (or None)."""
    from . import deepvoxels_nets as dvn
    B = x_real_full.shape[0]
    stage = 8.5
    use_rotate = iteration > cfg["start_rotation"]
    cams = camera.camera_matrices(thetas)
    for o in opt.values():
        o.zero_grad()
    x_real = torch.as_tensor(x_real_full)
    scale = x_real.shape[-1] // 64
    if scale > 1:
        x_real = F.avg_pool2d(x_real, scale, scale)
    z, z2, zd, zd2 = (torch.as_tensor(t) for t in z_fake)
    x_fake = dvn.deepvoxels_generator(gen_params, map_params, z, z2, cams)
    y_fake = nets.discriminator(dis_params, x_fake[:, :3], stage)
    loss_adv_g = loss_gen_adv_focal(y_fake, cfg["focal_loss_gamma"])
    loss_gen = loss_adv_g
    out = {"gen/loss_adv": float(loss_adv_g.detach())}
    if use_rotate:
        loss_rot, _ = warp_loss.loss_torch(x_fake[:B // 2], cams[:B // 2], x_fake[B // 2:], cams[B // 2:],
                                           occlusion_aware=False, lambda_geometric=cfg.get("lambda_geometric") or 3)
        loss_rot = loss_rot + depth_hinge(x_fake, cfg["depth_min"], cfg["lambda_depth"])
        out["gen/loss_rotate"] = float(loss_rot.detach())
        loss_gen = loss_gen + loss_rot * 0.3                                  # updater_deepvoxels.py:202 (see a27)
    loss_gen.backward()
    out["norm_map"] = opt["map"].update()
    out["norm_gen"] = opt["gen"].update()
    out["x_fake"] = x_fake.detach()
    opt["dis"].zero_grad()

    with torch.no_grad():
        x_fake_d = dvn.deepvoxels_generator(gen_params, map_params, zd, zd2, cams)
    y_fake = nets.discriminator(dis_params, x_fake_d[:, :3], stage)
    x_real = x_real.clone().requires_grad_(True)
    y_real = nets.discriminator(dis_params, x_real, stage)
    loss_adv = loss_dis_adv(y_fake, y_real)
    loss_dis = loss_adv
    if cfg["lambda_gp"] > 0:
        loss_gp = r1_penalty(y_real, x_real, cfg["lambda_gp"])
        out["dis/loss_gp"] = float(loss_gp.detach())
        loss_dis = loss_adv + loss_gp
    out["dis/loss_adv"] = float(loss_adv.detach())
    loss_dis.backward()
    out["norm_dis"] = opt["dis"].update()
    out["x_fake_dis"] = x_fake_d
    return out
