"""DeepVoxels generator / updater on the HIP kernels against the fp32 CPU oracle
(oracle/deepvoxels_nets.py, oracle/step.py:deepvoxels_step): same weights, same latents, same cameras.

Tolerances as in test_model_gpu.py: bf16 activations / conv operands with fp32 accumulation against an fp32 oracle;
tensors by relative L2 error, gradients by cosine similarity.  The frustum resampling and compositing stay fp32.
"""
import numpy as np
import pytest
import torch

from oracle import camera, deepvoxels_nets as dvn, nets, step

pytestmark = pytest.mark.gpu

CH = 256


def rel_err(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a - b).norm() / (b.norm() + 1e-30))


def cosine(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a @ b) / (a.norm() * b.norm() + 1e-30))


def _generator(seed=0):
    from rgbd_gan_amd.deepvoxels_generator import Generator
    from rgbd_gan_amd.utils.yaml_utils import Config
    gp = dvn.init_deepvoxels_generator(CH, seed=seed + 1)
    mp = dvn.init_mapping3d(CH, seed=seed)
    gen = Generator(CH, occlusion_type="accumulative", config=Config({}))
    gen.load_state_dict(gp)
    gen.mapping.load_state_dict(mp)
    return gp, mp, gen


def _inputs(B, seed=1):
    g = torch.Generator().manual_seed(seed)
    zh, zh2 = torch.randn(B // 2, CH, generator=g), torch.randn(B // 2, CH, generator=g)
    z, z2 = torch.cat([zh, zh]), torch.cat([zh2, zh2])
    np.random.seed(seed + 1)
    thetas = camera.PosePrior(0.3054, 3.1415, 0, uniform=True).sample(B)
    return z, z2, thetas


def test_parameter_names_and_shapes_match_oracle():
    gp, mp, gen = _generator()
    assert set(gen.state_dict().keys()) == set(gp.keys())
    assert set(gen.mapping.state_dict().keys()) == set(mp.keys())
    for k, v in gen.state_dict().items():
        assert tuple(v.shape) == tuple(gp[k].shape), k


def test_voxel_generator_matches_oracle():
    gp, mp, gen = _generator()
    z, _, _ = _inputs(2)
    with torch.no_grad():
        w_ref = dvn.mapping3d(mp, z)
        ref = dvn.voxel_generator(gp, w_ref)
        w = gen.mapping(z.cuda())
        got = gen.voxel_gen(w).cpu()
    assert rel_err(w.cpu(), w_ref) < 1e-4
    assert got.shape == ref.shape == (2, 32, 32, 32, 32)
    assert rel_err(got, ref) < 4e-2, rel_err(got, ref)


def test_renderer_matches_oracle():
    gp, mp, gen = _generator()
    g = torch.Generator().manual_seed(4)
    feats = torch.randn(2, 32, 64, 64, generator=g)
    w = torch.randn(2, CH, generator=g)
    with torch.no_grad():
        ref = dvn.renderer(gp, feats, w)
        got = gen.style_generator(feats.cuda(), w.cuda(), 8.5).cpu()
    assert got.shape == ref.shape == (2, 3, 64, 64)
    assert rel_err(got, ref) < 4e-2, rel_err(got, ref)


def test_generator_forward_and_gradients_match_oracle():
    gp, mp, gen = _generator()
    z, z2, thetas = _inputs(2)
    cams = camera.camera_matrices(thetas)
    gpl = {k: v.clone().requires_grad_(True) for k, v in gp.items()}
    mpl = {k: v.clone().requires_grad_(True) for k, v in mp.items()}
    ref = dvn.deepvoxels_generator(gpl, mpl, z, z2, cams)
    g = torch.Generator().manual_seed(9)
    probe = torch.randn(ref.shape, generator=g)
    (ref * probe).sum().backward()

    gen.cleargrads()
    gen.mapping.cleargrads()
    got = gen(z, 8.5, cams, z2=z2)
    (got * probe.cuda()).sum().backward()
    assert got.shape == ref.shape == (2, 4, 64, 64)
    assert rel_err(got[:, :3].detach().cpu(), ref[:, :3].detach()) < 5e-2
    assert rel_err(got[:, 3].detach().cpu(), ref[:, 3].detach()) < 2e-2
    for name in ("style_generator/c7/c/W", "style_generator/c6/c/W", "style_generator/c4/c/W", "style_generator/c1/c/W",
                 "style_generator/c0/c/W", "style_generator/s5/s/c/W", "style_generator/c0/c/b",
                 "deepvoxel/occlusion_net/occlusion/0/net/1/c/W", "voxel_gen/out/c/W", "voxel_gen/net/3/c1/c/W",
                 "voxel_gen/net/2/c0/c/W", "voxel_gen/net/1/c0/c/W", "voxel_gen/net/0/c1/c/W", "voxel_gen/net/0/W",
                 "voxel_gen/net/2/s0/b/c/W"):
        a, b = gen.store[name].grad.cpu(), gpl[name].grad
        # noise floor of bf16 activations through 8 3-D convs, the frustum resampling and 6 renderer convs: the deepest
        # (4^3) layer measures 0.89-0.91, the renderer's layers > 0.98
        assert cosine(a, b) > 0.85, (name, cosine(a, b))
        # (the clipped cumulative sum makes the occlusion-net gradients piecewise: bf16 voxel features move rays across
        # the clip, and fp32 atomics reorder sums from run to run)
        assert 0.65 < float(a.norm() / b.norm()) < 1.5, (name, float(a.norm() / b.norm()))
    # a bias in front of (leaky ReLU ->) AdaIN is nearly cancelled by the mean subtraction: its gradient is a small
    # difference of large sums and correspondingly noisy in bf16
    a, b = gen.store["voxel_gen/net/3/b1/b"].grad.cpu(), gpl["voxel_gen/net/3/b1/b"].grad
    assert cosine(a, b) > 0.6, cosine(a, b)
    for name in ("l/14/c/W", "l/0/c/W"):
        a, b = gen.mapping.store[name].grad.cpu(), mpl[name].grad
        assert cosine(a, b) > 0.85, (name, cosine(a, b))
    # parameters the forward never touches keep a zero gradient (noise scales, the unused first-block conv, the
    # camera-parameter MLP): chainer zero-fills them before the update
    for name in ("voxel_gen/net/1/n0/b/W", "voxel_gen/net/0/c0/c/W", "camera_param_generator/net/0/c/W"):
        assert float(gen.store[name].grad.abs().max()) == 0.0


CFG = dict(lambda_gp=1.0, lambda_depth=10, depth_min=0.6, focal_loss_gamma=2.0, start_rotation=0, lambda_geometric=None)


def test_deepvoxels_training_step_matches_oracle():
    from rgbd_gan_amd.net import Discriminator
    from rgbd_gan_amd.optimizer import FlatAdam
    from rgbd_gan_amd.updater import CameraParamPrior
    from rgbd_gan_amd.updater_deepvoxels import DeepVoxelsUpdater
    from rgbd_gan_amd.utils.yaml_utils import Config
    gp, mp, gen = _generator(seed=3)
    dp = nets.init_discriminator(CH, seed=8)
    dis = Discriminator(CH, res=True)
    dis.load_state_dict(dp)
    B = 4
    z, z2, thetas = _inputs(B, seed=5)
    g = torch.Generator().manual_seed(6)
    zd, zd2 = torch.randn(B, CH, generator=g), torch.randn(B, CH, generator=g)
    x_real = (np.random.RandomState(7).randint(0, 256, (B, 3, 128, 128)).astype("float32") / 127.5 - 1)
    iteration = 10

    gpl = {k: v.clone().requires_grad_(True) for k, v in gp.items()}
    mpl = {k: v.clone().requires_grad_(True) for k, v in mp.items()}
    dpl = {k: v.clone().requires_grad_(True) for k, v in dp.items()}
    oopt = {"map": step.ChainerAdam(mpl, 1e-5), "gen": step.ChainerAdam(gpl, 1e-3), "dis": step.ChainerAdam(dpl, 3e-3)}
    ref = step.deepvoxels_step(gpl, mpl, dpl, oopt, x_real, (z, z2, zd, zd2), thetas, CFG, iteration)

    cfg = Config(dict(generator_architecture="deepvoxels", stage_interval="0,0,0,0,0,0,0,0", max_stage=11,
                      start_rotation=0, start_occlusion_aware=0, lambda_depth=10, depth_min=0.6, focal_loss_gamma=2.0,
                      x_rotate=0.3054, y_rotate=3.1415, z_rotate=0, x_translate=0, y_translate=0, z_translate=0,
                      uniform_distribution=True, bigan=False))
    opt = {"map": FlatAdam(gen.mapping.store, 1e-5), "gen": FlatAdam(gen.store, 1e-3), "dis": FlatAdam(dis.store, 3e-3)}
    upd = DeepVoxelsUpdater(models=[gen, dis], config=cfg, optimizer=opt, iterator=None, lambda_gp=1.0, smoothing=0.999,
                            total_gpu=1, prior=CameraParamPrior(cfg))
    upd.iteration = iteration
    upd.update_core(batch=torch.from_numpy(x_real), z_fake=(z, z2, zd, zd2), thetas=thetas)
    obs = {k: float(v) for k, v in upd.observation.items()}
    assert obs["stage"] == 8.5 and obs["image_size"] == 64
    for key in ("gen/loss_adv", "gen/loss_rotate", "dis/loss_gp"):
        assert abs(obs[key] - ref[key]) < 6e-2 * max(1.0, abs(ref[key])), (key, obs[key], ref[key])
    # the discriminator's adversarial loss sees fakes from the UPDATED generator: with beta1 = 0 the first Adam step
    # moves every weight by +-alpha, the sign of near-zero gradients is rounding noise, so this one is loose
    assert abs(obs["dis/loss_adv"] - ref["dis/loss_adv"]) < 0.3 * ref["dis/loss_adv"], (obs["dis/loss_adv"], ref["dis/loss_adv"])
    for k, o, tol in (("norm_map", opt["map"], 0.15), ("norm_gen", opt["gen"], 0.15), ("norm_dis", opt["dis"], 0.3)):
        assert abs(float(o.grad_norm) - ref[k]) < tol * ref[k], (k, float(o.grad_norm), ref[k])
    # the generator moved: every live weight by about alpha (beta1 = 0, first update)
    w0, w1 = gp["style_generator/c6/c/W"], gen.store["style_generator/c6/c/W"].detach().cpu()
    wr = gpl["style_generator/c6/c/W"].detach()
    assert 0 < float((w1 - w0).abs().max()) <= 1e-3 * 1.001
    agree = float(((w1 - w0).sign() == (wr - w0).sign()).float().mean())
    assert agree > 0.85, agree
    assert opt["map"].t == opt["gen"].t == opt["dis"].t == 1


def test_deepvoxels_two_stream_step_equals_one_stream_step():
    """The four phases of a step (prep, D on the reals || generator step, D on the fresh fakes) on two streams, eager and
    replayed from graphs, against the same phases back to back on one stream: same losses and gradient norms.  (Not bit
    for bit: the frustum-resampling backward accumulates with fp32 atomics, so one arrangement does not repeat ITSELF to
    the last bit either.  Learning rates are 0: every call starts from the same weights, all four are comparable.)"""
    from rgbd_gan_amd.net import Discriminator
    from rgbd_gan_amd.optimizer import FlatAdam
    from rgbd_gan_amd.updater import CameraParamPrior
    from rgbd_gan_amd.updater_deepvoxels import DeepVoxelsUpdater
    from rgbd_gan_amd.utils.yaml_utils import Config
    B = 4
    z, z2, thetas = _inputs(B, seed=5)
    g = torch.Generator().manual_seed(6)
    zd, zd2 = torch.randn(B, CH, generator=g), torch.randn(B, CH, generator=g)
    x_real = torch.from_numpy(np.random.RandomState(7).randint(0, 256, (B, 3, 128, 128)).astype("float32") / 127.5 - 1)
    cfg = Config(dict(generator_architecture="deepvoxels", stage_interval="0,0,0,0,0,0,0,0", max_stage=11,
                      start_rotation=0, start_occlusion_aware=0, lambda_depth=10, depth_min=0.6, focal_loss_gamma=2.0,
                      x_rotate=0.3054, y_rotate=3.1415, z_rotate=0, x_translate=0, y_translate=0, z_translate=0,
                      uniform_distribution=True, bigan=False))
    runs = {}
    for concurrent in (False, True):
        _, _, gen = _generator(seed=3)
        dis = Discriminator(CH, res=True)
        dis.load_state_dict(nets.init_discriminator(CH, seed=8))
        opt = {"map": FlatAdam(gen.mapping.store, 0.0), "gen": FlatAdam(gen.store, 0.0), "dis": FlatAdam(dis.store, 0.0)}
        upd = DeepVoxelsUpdater(models=[gen, dis], config=cfg, optimizer=opt, iterator=None, lambda_gp=1.0,
                                smoothing=0.999, total_gpu=1, prior=CameraParamPrior(cfg), concurrent_phases=concurrent)
        upd.iteration = 10
        rows = []
        for it in range(4):                      # two eager steps, the capture, one replay
            upd.update_core(batch=x_real, z_fake=(z, z2, zd, zd2), thetas=thetas)
            torch.cuda.synchronize()
            row = {k: float(v) for k, v in upd.observation.items() if k.startswith(("gen/", "dis/"))}
            row.update({f"norm_{k}": float(o.grad_norm) for k, o in opt.items()})
            rows.append(row)
        assert len(upd._graphs) == 4 and opt["dis"].t == 4, (list(upd._graphs), opt["dis"].t)
        runs[concurrent] = rows
    for it, (a, b) in enumerate(zip(runs[False], runs[True])):
        assert set(a) == set(b) and {"dis/loss_adv", "dis/loss_gp", "gen/loss_adv", "gen/loss_rotate"} <= set(a)
        for k in a:
            # (losses: forward values; gradient norms sit behind the resampling backward's fp32 atomics, whose summation order
            # changes from launch to launch: the mapping network's norm spreads over 0.3 % between runs of ONE arrangement,
            # the losses repeat to the last digit printed)
            tol = 1e-2 if k.startswith("norm_") else 1e-5
            assert np.isfinite(a[k]) and abs(a[k] - b[k]) <= tol * max(1.0, abs(a[k])), (it, k, a[k], b[k])
            assert abs(a[k] - runs[False][0][k]) <= tol * max(1.0, abs(a[k])), (it, k)      # replay == capture == eager


def _dv_updater(lrs, **kwargs):
    from rgbd_gan_amd.net import Discriminator
    from rgbd_gan_amd.optimizer import FlatAdam
    from rgbd_gan_amd.updater import CameraParamPrior
    from rgbd_gan_amd.updater_deepvoxels import DeepVoxelsUpdater
    from rgbd_gan_amd.utils.yaml_utils import Config
    cfg = Config(dict(generator_architecture="deepvoxels", stage_interval="0,0,0,0,0,0,0,0", max_stage=11,
                      start_rotation=0, start_occlusion_aware=0, lambda_depth=10, depth_min=0.6, focal_loss_gamma=2.0,
                      x_rotate=0.3054, y_rotate=3.1415, z_rotate=0, x_translate=0, y_translate=0, z_translate=0,
                      uniform_distribution=True, bigan=False))
    _, _, gen = _generator(seed=3)
    dis = Discriminator(CH, res=True)
    dis.load_state_dict(nets.init_discriminator(CH, seed=8))
    opt = {"map": FlatAdam(gen.mapping.store, lrs[0]), "gen": FlatAdam(gen.store, lrs[1]), "dis": FlatAdam(dis.store, lrs[2])}
    upd = DeepVoxelsUpdater(models=[gen, dis], config=cfg, optimizer=opt, iterator=None, lambda_gp=1.0, smoothing=0.999,
                            total_gpu=1, prior=CameraParamPrior(cfg), **kwargs)
    upd.iteration = 10
    return gen, dis, opt, upd


def test_deepvoxels_step_with_the_next_forward_started_early_equals_the_plain_step():
    """prefetch_forward (updater_deepvoxels.py: the generator forward of step n+1 on the side stream under dis_fake of step n)
    against the plain four-phase step: the prior's poses drawn in the same order (one step early), the latents pinned, learning
    rates 0 -- so step k of one arrangement is step k of the other: same losses and gradient norms through the eager steps,
    the captures and the replays (tolerances of the two-stream test above: the resampling backward's fp32 atomics)."""
    B = 4
    x_real = torch.from_numpy(np.random.RandomState(7).randint(0, 256, (B, 3, 128, 128)).astype("float32") / 127.5 - 1)
    fixed = torch.randn(2 * B, CH, 1, 1, 1, generator=torch.Generator().manual_seed(12)).cuda()
    runs = {}
    for prefetch, split in ((False, False), (True, False), (True, True)):
        gen, dis, opt, upd = _dv_updater((0.0, 0.0, 0.0), prefetch_forward=prefetch, split_backward=split)
        assert upd.split_backward == split
        upd.get_z_fake_data = lambda n: fixed[:n]
        upd.call_log = []
        np.random.seed(21)
        rows = []
        for it in range(6):                      # two eager steps, the captures, replays
            upd.update_core(batch=x_real)
            upd.iteration += 1
            torch.cuda.synchronize()
            row = {k: float(v) for k, v in upd.observation.items() if k.startswith(("gen/", "dis/"))}
            row.update({f"norm_{k}": float(o.grad_norm) for k, o in opt.items()})
            rows.append(row)
        runs[prefetch, split] = rows
        names = [n for what, n, _ in upd.call_log if what == "phase"]
        rest = ["dv_gen_rest_a", "dv_gen_wgrad_a", "dv_gen_rest_b", "dv_gen_rest"] if split else ["dv_gen_rest"]
        if prefetch:
            # the first step runs its own forward; every step starts the next one's between G's update and dis_fake
            n = 5 + len(rest)
            assert names[:n] == ["dv_prep", "dv_dis_real", "dv_gen_fwd"] + rest + ["dv_gen_fwd", "dv_dis_fake"], names[:n]
            assert names[n:2 * n - 1] == ["dv_prep", "dv_dis_real"] + rest + ["dv_gen_fwd", "dv_dis_fake"], names[n:2 * n - 1]
            assert len(upd._graphs) == n - 1 and upd._pf is not None, list(upd._graphs)
            streams = {n: s for what, n, s in upd.call_log[-(n - 1):]}
            assert streams["dv_gen_fwd"] == streams["dv_dis_real"] != streams["dv_dis_fake"] == streams["dv_gen_rest"]
            assert not split or streams["dv_gen_wgrad_a"] == streams["dv_dis_real"] != streams["dv_gen_rest_b"]
        else:
            assert len(upd._graphs) == 4 and upd._pf is None and "dv_gen_fwd" not in names
    plain = runs[False, False]
    assert len({round(r["gen/loss_rotate"], 6) for r in plain}) > 1          # the poses do change from step to step
    for mode in ((True, False), (True, True)):
        for it, (a, b) in enumerate(zip(plain, runs[mode])):
            assert set(a) == set(b) and {"dis/loss_adv", "dis/loss_gp", "gen/loss_adv", "gen/loss_rotate"} <= set(a)
            for k in a:
                tol = 1e-2 if k.startswith("norm_") else 1e-5
                assert np.isfinite(a[k]) and abs(a[k] - b[k]) <= tol * max(1.0, abs(a[k])), (mode, it, k, a[k], b[k])


def test_deepvoxels_early_forward_reads_the_updated_generator():
    """... and with the optimizers running: the forward pass in flight after step n IS G_{n}'s (the weights behind step n's
    generator update, weight images rebuilt) on step n+1's poses -- recomputed here from the master weights after a
    synchronise -- in the eager steps, at the captures and in the replays; injected inputs drop the pass in flight and the pair
    of phases starts over."""
    B = 4
    x_real = torch.from_numpy(np.random.RandomState(7).randint(0, 256, (B, 3, 128, 128)).astype("float32") / 127.5 - 1)
    fixed = torch.randn(2 * B, CH, 1, 1, 1, generator=torch.Generator().manual_seed(12)).cuda()
    gen, dis, opt, upd = _dv_updater((1e-5, 1e-3, 3e-3))
    assert upd.prefetch_forward
    upd.get_z_fake_data = lambda n: fixed[:n]
    np.random.seed(22)
    zz, zz2 = fixed[:B // 2].repeat(2, 1, 1, 1, 1), fixed[B // 2:B].repeat(2, 1, 1, 1, 1)     # (one draw per pass, split)

    def check(tag):
        torch.cuda.synchronize()
        pf = upd._pf
        assert pf is not None and pf["st"]["fwd_x_fake"] is not None, tag
        with torch.no_grad():
            fresh = gen(zz, 8.5, pf["st"]["cams"], z2=zz2, theta=pf["st"]["theta9"])
        got = pf["st"]["fwd_x_fake"].detach()
        assert bool(torch.isfinite(got).all()) and rel_err(got, fresh) < 2e-3, (tag, rel_err(got, fresh))
        return got.clone()

    w0 = gen.store.flat.detach().clone()
    outs = []
    for it in range(5):
        upd.update_core(batch=x_real)
        upd.iteration += 1
        outs.append(check(it))
    assert float((gen.store.flat.detach() - w0).abs().max()) > 4e-3            # five Adam steps of 1e-3
    assert rel_err(outs[-1], outs[0]) > 2e-2                                  # (a stale generator would not pass `check`)
    assert opt["gen"].t == opt["dis"].t == 5
    # injected poses and latents: not the step the pass in flight was started for
    z, z2, thetas = _inputs(B, seed=5)
    upd.update_core(batch=x_real, z_fake=(z, z2, z, z2), thetas=thetas)
    upd.iteration += 1
    assert upd._pf is None and not any(k[-1].startswith("dv_gen_") for k in upd._graphs)
    # a change of the weights from outside: likewise
    for it in range(2):
        upd.update_core(batch=x_real)
        upd.iteration += 1
        check(("again", it))
    # another batch size (a data set's last, short batch): the pass in flight was drawn for four samples
    upd.update_core(batch=x_real[:2])
    upd.iteration += 1
    assert upd._pf is not None and upd._pf["B"] == 2 and upd.observation["batch_size"] == 2
    for it in range(2):
        upd.update_core(batch=x_real)
        upd.iteration += 1
        check(("after a short batch", it))
    from rgbd_gan_amd import functional as Fn
    with torch.no_grad():
        gen.store.flat.mul_(1.0)
    Fn.bump_weight_epoch()
    stale = upd._pf
    for it in range(4):
        upd.update_core(batch=x_real)
        upd.iteration += 1
        check(("after a load", it))
    assert upd._pf is not stale and opt["gen"].t == 15
    assert all(np.isfinite(float(v)) for k, v in upd.observation.items() if k.startswith(("gen/", "dis/")))
