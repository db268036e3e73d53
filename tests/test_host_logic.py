"""CPU tests of the host side of the product (no kernels run): schedule, cameras, prior, config, parameter store."""
import numpy as np
import pytest
import torch

from oracle import camera as ocam
from rgbd_gan_amd import updater as up
from rgbd_gan_amd.optimizer import FlatAdam
from rgbd_gan_amd.params import ParamStore
from rgbd_gan_amd.utils.yaml_utils import Config


def _cfg(**kw):
    base = dict(x_rotate=0.3054, y_rotate=1.0472, z_rotate=0, x_translate=0, y_translate=0, z_translate=0,
                stage_interval="0,0,0,0,0,0,0,100000, 150000, 160000, 180000, 300000", max_stage=11)
    base.update(kw)
    return Config(base)


def test_config_missing_keys_read_none():
    c = _cfg()
    assert c.rgb is None and c.lambda_rotate is None and c.uniform_distribution is None
    c.gpu = 3
    assert c.gpu == 3 and c["gpu"] == 3


def test_camera_matrices_match_oracle_bitwise():
    rng = np.random.RandomState(0)
    th = rng.uniform(-1, 1, (7, 6)).astype("float32")
    np.testing.assert_array_equal(up.get_camera_matries(th), ocam.camera_matrices(th))


def test_prior_matches_oracle_and_uniform_branch():
    for uniform, yrot in ((None, 1.0472), (None, 3.1415), (True, 3.1415)):
        cfg = _cfg(y_rotate=yrot, uniform_distribution=uniform)
        np.random.seed(4)
        a = up.CameraParamPrior(cfg).sample(10)
        np.random.seed(4)
        b = ocam.PosePrior(cfg.x_rotate, cfg.y_rotate, cfg.z_rotate, uniform=bool(uniform)).sample(10)
        np.testing.assert_array_equal(a, b)
        if uniform:   # only the uniform branch reflects the second view back into the range
            assert np.abs(a[:, 1]).max() <= yrot + 1e-6


def test_stage_schedule_matches_oracle():
    class U(up.RGBDUpdater):
        def __init__(self, cfg):
            self.config = cfg
            self.stage_interval = list(map(int, cfg.stage_interval.split(",")))
            self.fixed_stage = None
            self.iteration = 0
    u = U(_cfg())
    for it in (0, 1, 49999, 50000, 99999, 100000, 155000, 170000, 180000, 299999, 300000, 10 ** 6):
        u.iteration = it
        assert u.stage == ocam.stage_of(it, u.stage_interval, 11)


def test_downsize_real_matches_oracle():
    from oracle import nets
    x = torch.randn(2, 3, 128, 128)
    for st in (6.0, 7.3, 8.0, 9.5, 10.0, 10.99999999):
        torch.testing.assert_close(up.downsize_real(x, st), nets.downsize_real(x, st))


def test_param_store_grad_views_accumulate_in_place():
    store = ParamStore([("a/W", (3, 5), "normal"), ("a/b", (5,), "zeros"), ("c", (2, 2, 3), "ones")], "cpu", seed=1)
    assert store.numel % 4 == 0
    loss = (store["a/W"] ** 2).sum() + (store["c"] * 3).sum() + store["a/b"].sum()
    loss.backward()
    off = store.offsets["c"]
    assert torch.equal(store.grad[off:off + 12], torch.full((12,), 3.0))
    assert store["c"].grad.data_ptr() == store.grad.data_ptr() + 4 * off
    store.zero_grad()
    assert float(store.grad.abs().sum()) == 0.0 and float(store["a/W"].grad.abs().sum()) == 0.0
    sd = store.state_dict()
    sd["a/b"] = np.arange(5, dtype="float32")
    store.load(sd)
    assert float(store.flat[store.offsets["a/b"] + 4]) == 4.0


def test_adam_segments_follow_alpha_overrides():
    store = ParamStore([("l0/W", (8,), "normal"), ("l1/c/W", (8,), "normal"), ("l1/c/b", (4,), "zeros"),
                        ("l2/W", (8,), "normal")], "cpu")
    opt = FlatAdam(store, alpha=1e-3)
    opt.set_alpha("l1/c/W", 1e-5)
    opt.set_alpha("l1/c/b", 1e-5)
    begins, alphas = opt._segments()
    assert begins == [0, 8, 20, 28] and alphas == [1e-3, 1e-5, 1e-3]


def test_convert_batch_images_layout():
    """save_images.py:9-24: sample (r, c) at row-block 2r (RGB) and 2r+1 (depth as clip(128/d)), column-block c."""
    from rgbd_gan_amd.common.utils.save_images import convert_batch_images
    rows, cols, H = 2, 3, 4
    x = np.zeros((rows * cols, 4, H, H), dtype="float32")
    for i in range(rows * cols):
        x[i, :3] = i / 10.0 - 0.2
        x[i, 3] = 1.0 + i
    img = convert_batch_images(x, rows, cols)
    assert img.shape == (2 * rows * H, cols * H, 3) and img.dtype == np.uint8
    for r in range(rows):
        for c in range(cols):
            i = r * cols + c
            rgb = img[(2 * r) * H:(2 * r + 1) * H, c * H:(c + 1) * H]
            dep = img[(2 * r + 1) * H:(2 * r + 2) * H, c * H:(c + 1) * H]
            assert (rgb == np.uint8(np.clip((i / 10.0 - 0.2) * 127.5 + 127.5, 0, 255))).all()
            assert (dep == np.uint8(np.clip(128.0 / (1.0 + i), 0, 255))).all()
    rgb_only = convert_batch_images(x[:, :3], rows, cols)
    assert rgb_only.shape == (rows * H, cols * H, 3)


def test_downsized_size_matches_downsize_real():
    """updater.downsized_size is the arithmetic twin of downsize_real's output size (common/utils/pggan.py:6-50)."""
    import torch
    from rgbd_gan_amd.updater import downsize_real, downsized_size
    for st in (2.0, 3.25, 4.0, 5.5, 6.0, 6.999, 7.0, 8.0, 9.5, 10.0, 10.9999, 16.5):
        if downsized_size(st) <= 128:
            assert downsized_size(st) == downsize_real(torch.zeros(1, 1, 128, 128), st).shape[2], st


def test_upsample_planes_is_nearest_replication():
    import torch
    from rgbd_gan_amd.net import upsample_planes
    x = torch.arange(2 * 3 * 4 * 5, dtype=torch.float32).reshape(2, 3, 4, 5)
    ref = x.repeat_interleave(2, dim=2).repeat_interleave(2, dim=3)
    assert torch.equal(upsample_planes(x), ref)
    assert torch.equal(upsample_planes(x, 4), x.repeat_interleave(4, dim=2).repeat_interleave(4, dim=3))


def test_alpha_override_replaces_the_blend_factor_only():
    import torch
    from rgbd_gan_amd import net
    assert net._split_stage(9.25, 17) == (9, 0.25)
    a = torch.tensor(0.75)
    with net.alpha_override(a):
        fl, alpha = net._split_stage(9.25, 17)
        assert fl == 9 and alpha is a
        with net.alpha_override(None):
            assert net._split_stage(9.25, 17)[1] == 0.25
    assert net._split_stage(16.9999999999, 17)[0] == 16


def test_block_plan_of_the_progressive_networks():
    """Six blocks / stage ceiling 17 for the reference's 128 px networks (net.py:166,175-180,433); max_resolution=256 adds
    the block it keeps commented out (net.py:181,192: ch//8 channels), ceiling 19.  Engine and oracle agree on the plan."""
    from rgbd_gan_amd import net
    from oracle import nets as onets
    assert net._block_count(128) == 6 and net._block_count(256) == 7 and net._block_count(512) == 8
    assert net._synthesis_chans(256, 6) == [(256, 256)] * 4 + [(128, 256), (64, 128)]
    assert net._synthesis_chans(512, 7)[4:] == [(256, 512), (128, 256), (64, 128)]
    for res, ch in ((128, 256), (256, 512)):
        nb = net._block_count(res)
        assert onets.block_count(res) == nb and onets.synthesis_chans(ch, nb) == net._synthesis_chans(ch, nb)
    gp = onets.init_stylegan(256, max_resolution=256)
    dp = onets.init_discriminator(256, max_resolution=256)
    assert onets.max_stage_of(gp, "gen/outs") == 19 and onets.max_stage_of(dp, "ins") == 19
    assert onets.max_stage_of(onets.init_discriminator_sn(256), "ins") == 17
    assert tuple(gp["gen/blocks/6/c0/c/W"].shape) == (32, 64, 3, 3) and tuple(dp["blocks/6/c1/c/W"].shape) == (64, 64, 3, 3)
    for bad in (64, 192, 100):
        try:
            net._block_count(bad)
        except ValueError:
            continue
        raise AssertionError(f"max_resolution={bad} accepted")


def test_param_store_fused_views_share_storage_and_gradient():
    """ParamStore.fused: back-to-back parameters as one leaf (the style scale / shift affines served by one launch)."""
    import torch
    from rgbd_gan_amd.params import ParamStore
    specs = [("a/W", (4, 8), "normal"), ("b/W", (4, 8), "normal"), ("a/b", (4,), "ones"), ("b/b", (4,), "zeros"),
             ("c/W", (3, 5), "normal")]
    st = ParamStore(specs, "cpu", seed=0)
    W = st.fused(("a/W", "b/W"), (8, 8))
    b = st.fused(("a/b", "b/b"), (8,))
    assert torch.equal(W[:4], st["a/W"]) and torch.equal(W[4:], st["b/W"])
    assert torch.equal(b, torch.cat([st["a/b"], st["b/b"]]))
    (W.sum() * 2 + (b * torch.arange(8.0)).sum()).backward()
    assert torch.equal(st["a/W"].grad, torch.full((4, 8), 2.0)) and torch.equal(st["b/b"].grad, torch.arange(4.0, 8.0))
    st.zero_grad()
    assert float(W.grad.abs().sum()) == 0.0 and W.grad.data_ptr() == st["a/W"].grad.data_ptr()
    with pytest.raises(ValueError):
        st.fused(("a/W", "c/W"), (4 * 8 + 15,))                # not adjacent


def test_residual_tie_hand_overs_work_in_either_order():
    """functional.ResidualTie: a producer parks its term for the consumer when it runs first; when the consumer ran
    first it is told so and returns the term to autograd the ordinary way.  Both orders leave the tie clean."""
    from rgbd_gan_amd.functional import ResidualTie
    tie = ResidualTie(torch.zeros(4, requires_grad=True))
    term = torch.ones(2)
    # producer first: parked, consumer takes it
    assert tie.give("dx_sc", "entry_seen", term) is True
    assert tie.take("dx_sc", "entry_seen") is term
    assert tie.dx_sc is None and tie.entry_seen is False
    # consumer first: nothing parked, producer must keep its term
    assert tie.take("dx_sc", "entry_seen") is None
    assert tie.give("dx_sc", "entry_seen", term) is False
    assert tie.dx_sc is None and tie.entry_seen is False
    # the two junctions are independent
    assert tie.give("g_sc", "main_seen", term) is True and tie.dx_sc is None
    assert tie.take("g_sc", "main_seen") is term


def test_trainer_snapshot_uses_the_reference_key_layout_and_round_trips():
    """snapshot_iter_*.npz (train_rgbd.py:378-381 of the reference: chainer's trainer snapshot): the keys a Chainer trainer
    would write and read -- updater/iteration, updater/optimizer:<name>/<param>/{t,m,v}, updater/model:<name>/<param>,
    updater/iterator:main/..., extensions/LogReport/_log -- and a round trip through fresh optimizers; the flat layout this
    engine wrote in rounds 1-3 still loads; absent keys are skipped like load_npz(strict=False)."""
    from rgbd_gan_amd.common.utils import trainer_snapshot as ts
    from rgbd_gan_amd.optimizer import FlatAdam
    from rgbd_gan_amd.params import ParamStore
    from rgbd_gan_amd.training import DeviceImageIterator

    def make():
        stores = {"map": ParamStore([("l/0/c/W", (8, 8), "normal"), ("l/0/c/b", (8,), "zeros")], "cpu", seed=1),
                  "dis": ParamStore([("blocks/0/c0/c/W", (4, 3, 3, 3), "normal"), ("ins/5/b", (5,), "ones")], "cpu", seed=2)}
        return {k: FlatAdam(v, 1e-3) for k, v in stores.items()}
    opts = make()
    g = torch.Generator().manual_seed(0)
    for o in opts.values():
        o.m.copy_(torch.randn(o.m.shape, generator=g))
        o.v.copy_(torch.rand(o.v.shape, generator=g))
        o.step.fill_(7)
    images = np.zeros((20, 3, 4, 4), "uint8")
    it = DeviceImageIterator(images, 8, "cpu", seed=3)
    for _ in range(4):
        it.next_indices()
    log = [{"iteration": 100, "gen/loss_adv": 0.5}]
    with torch.no_grad():
        opts["dis"].store.flat.add_(0.25)                 # trained weights: the snapshot's updater/model:* keys carry them
    snap = ts.pack(1200, opts, it.state_dict(), log, 12.5, 100)
    for k in ("updater/iteration", "updater/optimizer:map/t", "updater/optimizer:map/l/0/c/W/m", "updater/optimizer:dis/ins/5/b/v",
              "updater/model:dis/blocks/0/c0/c/W", "updater/iterator:main/current_position", "updater/iterator:main/order",
              "extensions/LogReport/_log", "_snapshot_elapsed_time"):
        assert k in snap, k
    assert snap["updater/optimizer:map/l/0/c/W/m"].shape == (8, 8) and int(snap["updater/optimizer:dis/ins/5/b/t"]) == 7
    np.testing.assert_array_equal(snap["updater/model:map/l/0/c/W"], opts["map"].store["l/0/c/W"].detach().numpy())
    fresh = make()
    got = ts.unpack(snap, fresh)
    assert got["iteration"] == 1200 and got["log"] == log and got["elapsed_time"] == 12.5
    def same_moments(a, b):           # per parameter: the flat buffers also hold alignment padding, which no file carries
        for n in a.store.names:
            sl = slice(a.store.offsets[n], a.store.offsets[n] + int(np.prod(a.store.shapes[n])))
            if not (torch.equal(a.m[sl], b.m[sl]) and torch.equal(a.v[sl], b.v[sl])):
                return False
        return True
    for k in opts:
        assert same_moments(fresh[k], opts[k]) and fresh[k].t == 7
        for n in opts[k].store.names:                      # the optimizer's target link is restored from the snapshot as well
            assert torch.equal(fresh[k].store[n], opts[k].store[n]), (k, n)
    partial = {k: v for k, v in snap.items() if k != "rgbd_gan_amd/iterator:main/seed"}     # rng_state without its seed
    it4 = DeviceImageIterator(images, 8, "cpu", seed=77)
    it4.load_state_dict(ts.unpack(partial, make())["iterator"])
    assert it4.seed == 77
    it2 = DeviceImageIterator(images, 8, "cpu", seed=99)
    it2.load_state_dict(got["iterator"])
    assert [it2.next_indices().tolist() for _ in range(5)] == [it.next_indices().tolist() for _ in range(5)]
    # a file as the REFERENCE's trainer writes it: no generator state, '//'-joined parameter paths, only some keys
    chainer_like = {"updater/iteration": np.int64(500), "updater/optimizer:map/t": np.int64(3),
                    "updater/optimizer:map//l/0/c/W/m": np.full((8, 8), 2.0, "float32"),
                    "updater/iterator:main/current_position": np.int64(8), "updater/iterator:main/epoch": np.int64(1),
                    "updater/iterator:main/order": np.arange(20)[::-1].copy()}
    fresh = make()
    got = ts.unpack(chainer_like, fresh)
    assert got["iteration"] == 500 and got["log"] is None and fresh["map"].t == 3 and fresh["dis"].t == 0
    assert float(fresh["map"].m[:64].min()) == 2.0 and float(fresh["map"].v.abs().max()) == 0.0
    it3 = DeviceImageIterator(images, 8, "cpu", seed=5)
    it3.load_state_dict(got["iterator"])
    assert it3.next_indices().tolist() == [11, 10, 9, 8, 7, 6, 5, 4]
    # the flat layout of rounds 1-3
    old = {"iteration": np.int64(40), "map/t": 2, "map/m": opts["map"].m.numpy(), "map/v": opts["map"].v.numpy(),
           "log": np.asarray('[{"iteration": 40}]'), "elapsed_time": np.float64(1.0)}
    fresh = make()
    got = ts.unpack(old, fresh)
    assert got["iteration"] == 40 and got["log"] == [{"iteration": 40}] and same_moments(fresh["map"], opts["map"])


def test_side_stream_weight_gradient_budget_rule():
    """RGBDUpdater._side_wgrad_auto: workgroups of the side stream's batched weight-gradient launches when nothing has been
    measured on the device -- a line through the measured optima of five shapes (profiles/r06/cu_budget_sweep.txt): 112 + pixels /
    10240 of 256 compute units, multiples of 8, at least 64, at most all."""
    from rgbd_gan_amd.updater import RGBDUpdater

    class Shape:
        def __init__(self, *s):
            self.shape = s
    fake = type("U", (), {"device": "cpu", "_dfw_rule": staticmethod(RGBDUpdater._dfw_rule),
                          "_side_wgrad_pair": RGBDUpdater._side_wgrad_pair})()
    rule = lambda B, S: RGBDUpdater._side_wgrad_auto(fake, {"B": B, "x_real": Shape(B, 3, S, S)})
    pair = lambda B, S: RGBDUpdater._side_wgrad_pair(fake, {"B": B, "x_real": Shape(B, 3, S, S)})
    if not torch.cuda.is_available():
        assert rule(32, 128) == 160 and rule(8, 128) == 128 and rule(16, 256) == 216 and rule(32, 64) == 128
        assert rule(16, 128) == 136 and rule(64, 256) == 256 and rule(2, 16) == 112
        # the second launch (`dfw`, mostly behind the end of the generator's backward): half way to the whole chip
        assert pair(32, 128) == (160, 208) and pair(8, 128) == (128, 192) and pair(16, 256) == (216, 240)
    assert all(rule(B, S) % 8 == 0 and pair(B, S)[1] % 8 == 0 for B in (2, 8, 32) for S in (16, 64, 256))


def test_statistics_pool_counts_only_a_steps_own_takes():
    """kernels._StatsPool: while a stage is replayed from HIP graphs no begin_step runs in Python, but the
    preview sampler still calls the generator -- those takes must neither grow the pool nor move the step's slices."""
    from rgbd_gan_amd.kernels import _StatsPool
    cpu = torch.device("cpu")
    pool = _StatsPool()
    assert pool.take(1000, cpu) is None and pool.used == 0           # outside a step: caller falls back to torch.zeros
    bufs = []
    pool.begin_step(cpu, bufs)                                        # first step of a configuration: pool too small
    first = pool.buf.numel()
    assert len(bufs) == 1 and bufs[0].dtype == torch.float32 and bufs[0].numel() == 2 * first
    assert pool.take(3 * first, cpu) is None and pool.used == 3 * first
    pool.end_step()
    for _ in range(500):                                              # 500 previews between steps
        assert pool.take(3 * first, cpu) is None
    assert pool.used == 3 * first
    pool.begin_step(cpu, [])                                          # second step: exactly the first one's demand
    assert pool.buf.numel() == 3 * first and len(pool._retired) == 1
    a = pool.take(2 * first, cpu)
    b = pool.take(first, cpu)
    assert a.numel() == 2 * first and b.data_ptr() == a.data_ptr() + 8 * 2 * first
    pool.end_step()
    pool.begin_step(cpu, [])                                          # steady state: same buffer, same slices
    assert pool.buf.numel() == 3 * first and pool.take(2 * first, cpu).data_ptr() == a.data_ptr()


def test_data_parallel_budgets_leave_compute_units_to_the_collectives():
    """RGBDUpdater._dp_budgets: beside a pending all-reduce no weight-gradient plan is sized for the whole
    chip (the 3x3 grids are left alone: a power-of-two tile count makes any smaller grid a whole extra round), and the side
    stream's weight-gradient launches get more workgroups than the one-GPU rule gives them (so that D's gradients are on the
    wire before the generator's backward ends)."""
    from rgbd_gan_amd.updater import RGBDUpdater
    fake = type("U", (), {"device": "cpu", "dp_reserve_cus": 16, "dp_side_lead_workgroups": 32, "side_cu_budget": 224,
                          "side_wgrad_workgroups": None})()
    if torch.cuda.is_available():
        return
    st = {"side_wgrad_wgs": 64, "dfw_wgrad_wgs": 160}
    assert RGBDUpdater._dp_budgets(fake, st) == 240 and st == {"side_wgrad_wgs": 96, "dfw_wgrad_wgs": 192}
    st = {"side_wgrad_wgs": 208, "dfw_wgrad_wgs": 232}
    assert RGBDUpdater._dp_budgets(fake, st) == 240 and st == {"side_wgrad_wgs": 240, "dfw_wgrad_wgs": 240}
    fake.side_wgrad_workgroups = 96                                        # explicit counts are capped, not moved
    st = {"side_wgrad_wgs": 96, "dfw_wgrad_wgs": 0}
    assert RGBDUpdater._dp_budgets(fake, st) == 240 and st == {"side_wgrad_wgs": 96, "dfw_wgrad_wgs": 240}
    st = {}                                                                # one stream: no side counts, everything capped
    assert RGBDUpdater._dp_budgets(fake, st) == 240 and st == {"side_wgrad_wgs": 240, "dfw_wgrad_wgs": 240}


def test_updater_constructor_takes_every_documented_argument_and_rejects_unknown_ones():
    """RGBDUpdater.__init__ pops its keyword arguments one by one and raises on leftovers: an edit that drops one of the pops turns
    a documented argument into a TypeError (it happened to `tune_side_budget` in round 6, and only a GPU test noticed)."""
    import types
    from rgbd_gan_amd.updater import CameraParamPrior, RGBDUpdater
    from rgbd_gan_amd.utils.yaml_utils import Config
    cfg = Config(dict(stage_interval="0,0,0,0,0,0,0,100,200", max_stage=11, x_rotate=0.3, y_rotate=1.0, z_rotate=0, x_translate=0,
                      y_translate=0, z_translate=0, bigan=False, generator_architecture="stylegan"))
    gen, dis = types.SimpleNamespace(device=torch.device("cpu")), types.SimpleNamespace()
    base = dict(optimizer={}, iterator=None, lambda_gp=1.0, smoothing=0.999, total_gpu=1, prior=CameraParamPrior(cfg))
    extra = dict(nan_check_interval=10, nan_watch=False, fixed_stage=8.0, use_graphs=True, graph_warmup=2, graph_fallback=False,
                 concurrent_phases=True, dp_split_body=True, side_cu_budget=224, side_wgrad_workgroups=None, dfw_wgrad_workgroups=None,
                 tune_side_budget=True, dp_reserve_cus=16, dp_side_lead_workgroups=32)
    upd = RGBDUpdater(models=[gen, dis], config=cfg, **base, **extra)
    assert upd.tune_side_budget and upd.fixed_stage == 8.0 and upd.dp_reserve_cus == 16 and upd.stage == 8.0
    assert not upd.tuning_in_progress
    with pytest.raises(TypeError, match="unknown arguments"):
        RGBDUpdater(models=[gen, dis], config=cfg, **base, no_such_argument=1)


def test_deepvoxels_updater_arrangement_switches():
    """DeepVoxelsUpdater's arrangement arguments: the early forward needs the two-stream step and a step that is not data parallel
    (there the generator phase holds the collectives); the split backward needs the early forward."""
    import types
    from rgbd_gan_amd.updater import CameraParamPrior
    from rgbd_gan_amd.updater_deepvoxels import DeepVoxelsUpdater
    from rgbd_gan_amd.utils.yaml_utils import Config
    cfg = Config(dict(stage_interval="0,0,0,0,0,0,0,0", max_stage=11, x_rotate=0.3, y_rotate=1.0, z_rotate=0, x_translate=0,
                      y_translate=0, z_translate=0, bigan=False, generator_architecture="deepvoxels", lambda_geometric=None))
    K = np.array([[64.0, 0, 32], [0, 64.0, 32], [0, 0, 1]], dtype="float32")
    gen = types.SimpleNamespace(device=torch.device("cpu"), projection=types.SimpleNamespace(projection_intrinsic=K))

    def build(comm_active=False, **kw):
        opt = {"gen": types.SimpleNamespace(comm=types.SimpleNamespace(active=comm_active))}
        return DeepVoxelsUpdater(models=[gen, types.SimpleNamespace()], config=cfg, optimizer=opt, iterator=None, lambda_gp=1.0,
                                 smoothing=0.999, total_gpu=1, prior=CameraParamPrior(cfg), **kw)
    upd = build()
    assert upd.prefetch_forward and upd.split_backward and upd.forward_cu_budget == 96 and upd.renderer_wgrad_workgroups == 64
    assert upd._pf is None and upd.get_stage() == 8.5
    upd = build(prefetch_forward=False)
    assert not upd.prefetch_forward and not upd.split_backward
    upd = build(split_backward=False, forward_cu_budget=128, renderer_wgrad_workgroups=0)
    assert upd.prefetch_forward and not upd.split_backward and upd.forward_cu_budget == 128 and upd.renderer_wgrad_workgroups == 0
    upd = build(comm_active=True)
    assert not upd.use_graphs and not upd.prefetch_forward and not upd.split_backward


def test_pack_fold_code_is_the_headers_bit_layout():
    """rgbd_pack_desc.fold (include/rgbd_gan_hip.h): bits 0-1 = mode + 1, bits 2-16 = master Cout, bits 17-31 = master Cin, and it must fit
    the struct's int32 (kernels.PACK_DESC)."""
    from rgbd_gan_amd import kernels
    for mode, co, ci in ((0, 32, 64), (1, 1024, 512), (2, 3, 288), (2, 32767, 32767)):
        code = kernels.pack_fold_code(mode, co, ci)
        assert -2 ** 31 <= code < 2 ** 31 and np.zeros(1, dtype=kernels.PACK_DESC)["fold"].dtype == np.dtype("<i4")
        bits = code & 0xffffffff                       # what the kernel sees: (fold & 3) - 1, (fold >> 2) & 0x7fff, (fold >> 17) & 0x7fff
        assert (bits & 3) - 1 == mode and (bits >> 2) & 0x7fff == co and (bits >> 17) & 0x7fff == ci
        tab = np.zeros(1, dtype=kernels.PACK_DESC)
        tab["fold"] = code
        assert int(tab["fold"][0]) == code
    assert dict(kernels.PACK_DESC)["fold"] == "<i4" and np.dtype(kernels.PACK_DESC).itemsize == 48
    for bad in ((3, 8, 8), (0, 0, 8), (0, 8, 32768)):
        with pytest.raises(ValueError):
            kernels.pack_fold_code(*bad)
