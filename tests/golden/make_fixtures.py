"""Generate the golden vectors in this directory FROM THE ORACLE (the reference has no fixtures and cannot run here:
parity with Chainer is unpinned; these pin the oracle and the HIP kernels to each other across revisions).

    python tests/golden/make_fixtures.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import camera, deepvoxels, warp_loss  # noqa: E402


def warp_case(seed, b, S, occ, lam):
    rng = np.random.RandomState(seed)
    img = rng.uniform(-1, 1, (b, 4, S, S)).astype("float32")
    img_rot = rng.uniform(-1, 1, (b, 4, S, S)).astype("float32")
    img[:, 3] = rng.uniform(0.7, 1.3, (b, S, S))
    img_rot[:, 3] = rng.uniform(0.7, 1.3, (b, S, S))
    th = rng.uniform(-0.3, 0.3, (2 * b, 6)).astype("float32")
    th[:, 2] = 0
    th[:, 3:] *= 0.1
    cams = camera.camera_matrices(th)
    ref = warp_loss.forward_np(img, cams[:b], img_rot, cams[b:], occlusion_aware=occ, lambda_geometric=lam)
    ti = torch.from_numpy(img).requires_grad_(True)
    tr = torch.from_numpy(img_rot).requires_grad_(True)
    loss, _ = warp_loss.loss_torch(ti, cams[:b], tr, cams[b:], occlusion_aware=occ, lambda_geometric=lam)
    loss.backward()
    coef = np.concatenate([ref["A"].reshape(b, 9), ref["c"], ref["A2"].reshape(b, 9), ref["c2"]], 1).astype("float32")
    return dict(img=img, img_rot=img_rot, thetas=th, cam=cams[:b], cam_rot=cams[b:], coef=coef,
                occlusion=np.array(occ), lambda_geometric=np.array(lam, "float32"),
                loss=np.array(ref["loss"]), zp=ref["zp"], zp_rot=ref["zp_rot"], warped=ref["warped"],
                warped_rot=ref["warped_rot"], mask=ref["mask"], mask_rot=ref["mask_rot"], u0=ref["u0"], v0=ref["v0"],
                v1=ref["v1"], u0_rot=ref["u0_rot"], v0_rot=ref["v0_rot"], v1_rot=ref["v1_rot"],
                grad_img=ti.grad.numpy(), grad_img_rot=tr.grad.numpy())


def deepvoxels_case(seed):
    """A small frustum (16x16 image, 8^3 grid, 14 depth samples): projection indices / coordinates, resampled
    volume, compositing outputs and first-order gradients."""
    fr = deepvoxels.Frustum(grid_dim=8, img=16)
    rng = np.random.RandomState(seed)
    th = np.zeros((2, 6), "float32")
    th[:, 0] = rng.uniform(-0.3, 0.3, 2)
    th[:, 1] = rng.uniform(-2.0, 2.0, 2)
    cams = camera.camera_matrices(th)
    g = torch.Generator().manual_seed(seed)
    F = 4
    grid = torch.randn(2, F, 8, 8, 8, generator=g)
    W1 = torch.randn(4, F + 1, generator=g)
    b1 = torch.randn(4, generator=g) * 0.1
    W2 = torch.randn(1, 4, generator=g) * 2
    b2 = torch.full((1,), 3.0)
    out = dict(thetas=th, cams=cams, grid=grid.numpy(), W1=W1.numpy(), b1=b1.numpy(), W2=W2.numpy(), b2=b2.numpy(),
               grid_dim=np.array(8), img=np.array(16), depth=np.array(fr.depth), voxel_size=np.array(fr.voxel_size),
               near_plane=np.array(fr.near_plane))
    for i in range(2):
        lin, v = deepvoxels.proj_idcs_np(cams[i], fr)
        gi = grid[i:i + 1].clone().requires_grad_(True)
        vol = deepvoxels.trilinear_torch(gi, lin, v, fr)
        feat, depth, w = deepvoxels.occlusion_torch(vol, W1, b1, W2, b2, fr)
        (feat.sum() + 2.0 * depth.sum()).backward()
        out.update({f"lin{i}": lin, f"coords{i}": v, f"vol{i}": vol.detach().numpy(), f"feat{i}": feat.detach().numpy(),
                    f"depth{i}": depth.detach().numpy(), f"weights{i}": w.detach().numpy(), f"dgrid{i}": gi.grad.numpy()})
    return out


STEP_TINY = dict(ch=16, B=4, cfg=dict(lambda_gp=1.0, lambda_depth=10, depth_min=1.0, lambda_geometric=None, lambda_rotate=None,
                                      start_rotation=2000, start_occlusion_aware=2000))


def step_tiny_case(stage):
    """One full update_core of the ORACLE (oracle/step.py:rgbd_step: generator step, discriminator step, R1, 3-D loss, clipped
    Chainer-flavoured Adam) on 16-channel networks: losses, pre-clip gradient norms, the norm and four probe entries of every
    parameter gradient, and the norm of every parameter after the update.  Runs in well under a second on the CPU box, so an
    edit of the oracle that changes a gradient or the optimizer is caught there (tests/test_golden.py), not only on the GPU
    box where the engine is compared with it."""
    from oracle import nets, step
    ch, B = STEP_TINY["ch"], STEP_TINY["B"]
    gp = {k: v.requires_grad_(True) for k, v in nets.init_stylegan(ch, seed=0).items()}
    dp = {k: v.requires_grad_(True) for k, v in nets.init_discriminator(ch, seed=1).items()}
    torch.manual_seed(0)
    for i in range(6):                                   # a depth head that sees geometry
        with torch.no_grad():
            gp[f"gen/outs/{i}/c/W"][-1] = torch.randn(gp[f"gen/outs/{i}/c/W"][-1].shape) * 0.1
    omap = {k: v for k, v in gp.items() if k.startswith("mapping/")}
    ogen = {k: v for k, v in gp.items() if k.startswith("gen/")}
    low = {k: 1e-5 for k in ("gen/l1/c/W", "gen/l1/c/b", "gen/l2/c/W", "gen/l2/c/b")}
    opt = {"map": step.ChainerAdam(omap, 1e-5), "gen": step.ChainerAdam(ogen, 1e-3, alpha_override=low),
           "dis": step.ChainerAdam(dp, 3e-3)}
    rng = np.random.RandomState(0)
    zh = nets.make_hidden(B // 2, ch, rng)
    z = np.concatenate([zh, zh])
    np.random.seed(2)
    thetas = camera.PosePrior(0.3054, 1.0472, 0).sample(B)
    x_real = rng.randint(0, 256, (B, 3, 128, 128)).astype("float32") / 127.5 - 1
    ref = step.rgbd_step(gp, dp, opt, x_real, z, thetas, stage, STEP_TINY["cfg"], 200000)
    out = {"stage": np.array(stage), "x_fake": ref["x_fake"].detach().numpy()}
    for k in ("gen/loss_adv", "gen/loss_rotate", "dis/loss_adv", "dis/loss_gp", "norm_map", "norm_gen", "norm_dis"):
        out["obs/" + k] = np.array(float(ref[k]), dtype=np.float64)
    names, gnorm, probe, pnorm = [], [], [], []
    for src in (gp, dp):
        for k in sorted(src):
            g = src[k].grad
            names.append(k)
            gnorm.append(0.0 if g is None else float(g.double().norm()))
            flat = torch.zeros(4) if g is None else g.flatten()[:: max(1, g.numel() // 4)][:4]
            probe.append(np.pad(flat.numpy(), (0, 4 - len(flat))))
            pnorm.append(float(src[k].detach().double().norm()))
    out.update(names=np.array(names), grad_norm=np.array(gnorm), grad_probe=np.stack(probe).astype("float32"),
               param_norm_after=np.array(pnorm))
    return out


def main():
    for stage in (4.0, 5.5):
        np.savez_compressed(os.path.join(HERE, f"step_tiny_stage{stage}.npz"), **step_tiny_case(stage))
    np.savez_compressed(os.path.join(HERE, "deepvoxels_small.npz"), **deepvoxels_case(103))
    np.savez_compressed(os.path.join(HERE, "warp_loss_b2_s16_occ.npz"), **warp_case(101, 2, 16, True, 3.0))
    np.savez_compressed(os.path.join(HERE, "warp_loss_b3_s8.npz"), **warp_case(102, 3, 8, False, 2.0))
    np.random.seed(7)
    prior = camera.PosePrior(0.3054, 1.0472, 0)
    th = prior.sample(8)
    np.random.seed(7)
    th_u = camera.PosePrior(0.3054, 3.1415, 0, uniform=True).sample(8)
    si = camera.parse_stage_interval("0,0,0,0,0,0,0,100000, 150000, 160000, 180000, 300000")
    its = np.array([0, 1, 49999, 50000, 100000, 125000, 150000, 155000, 160000, 170000, 180000, 240000, 300000, 999999])
    np.savez_compressed(os.path.join(HERE, "host_math.npz"), thetas=th, thetas_uniform=th_u,
                        cams=camera.camera_matrices(th), theta9=camera.theta9(th), iterations=its,
                        stages=np.array([camera.stage_of(int(i), si, 11) for i in its]))
    print("wrote", sorted(f for f in os.listdir(HERE) if f.endswith(".npz")))


if __name__ == "__main__":
    main()
