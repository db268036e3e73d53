"""The training step: RGBDUpdater with the reference's constructor and properties (updater.py:214-448).

    RGBDUpdater(models=[gen, dis(, smoothed)], config=..., optimizer={'map','gen','dis'}, iterator=...,
                lambda_gp=..., smoothing=..., total_gpu=..., prior=...)
    .stage  .iteration  .update()  .update_core()  .observation

One update_core() computes what the reference's generator step + discriminator step compute (same losses, same
gradients, same Adam updates), arranged for the GPU:
    prep : clear gradient buffers, down-size the reals, repack the bf16 weight images
    gen  : x_fake = G(z, stage, theta9); ONE forward and ONE backward through D(x_fake) -- seeded with the
           discriminator loss it gives D's weight gradients for the fakes, rescaled per sample it gives the
           generator's adversarial image gradient (gen_a); 3D-consistency loss (HIP warp-loss kernel) + depth hinge;
           G backward (gen_b)
    dfw  : D's weight gradients for the fakes, collected during gen_a, issued on the second stream behind dis
    dis  : D(x_real), R1 first-order pass, double backward with the adversarial seeds on the reals folded in
           (runs on a second stream, concurrently with gen)
    join : merge D's two gradient buffers;  opt : clip + Adam for map / gen / dis (+ EMA generator)
Two streams (default): every phase is captured once per configuration as its own HIP graph and replayed on the stream it belongs
to (generator phases on the main stream, discriminator phases and D's optimizer on a second one), stream events between the launches;
one stream: the body as one graph (two under data parallelism) + the optimizer phase.
Data-parallel: each optimizer's flat gradient buffer is all-reduced once (RCCL), outside the graphs (train_rgbd.py:154-156).  Two
streams: D's all-reduce is enqueued from the side stream directly behind dfw and D's Adam step follows it there, under the generator's
backward; the generator's all-reduces and Adam step follow gen_b on the main stream.  On ONE stream
(RGBD_CONCURRENT_PHASES=0, the shared-device tests) the body is captured as two graphs, split where the generator's gradients are
final (after gen_b), and the map + gen all-reduces run under the discriminator half.
"""
import contextlib
import math
import os

import numpy as np
import torch
import torch.nn.functional as F

from . import functional as Fn
from . import kernels
from .common.utils.copy_param import soft_copy_param
from .common.loss_functions import LossFuncRotate, loss_func_dcgan_dis, loss_func_dcgan_gen, loss_l2


def update_camera_matrices(mat, axis1, axis2, theta):
    """updater.py:26-42: left-multiply every 4x4 by a rotation of `theta` in the (axis1, axis2) plane."""
    rot = np.zeros_like(mat)
    idx = np.arange(4)
    rot[:, idx, idx] = 1
    c, s = np.cos(theta), np.sin(theta)
    rot[:, axis1, axis1] = c
    rot[:, axis1, axis2] = -s
    rot[:, axis2, axis1] = s
    rot[:, axis2, axis2] = c
    return np.matmul(rot, mat)


def get_camera_matries(thetas, order=(0, 1, 2)):
    """updater.py:45-60 (name kept, typo included): (n,6) [x,y,z rotation, x,y,z translation] -> (n,4,4) float32."""
    mat = np.zeros((len(thetas), 4, 4), dtype="float32")
    idx = np.arange(4)
    mat[:, idx, idx] = [1, 1, -1, 1]
    mat[:, 2, 3] = 1
    for i in order:
        mat = update_camera_matrices(mat, (i + 1) % 3, (i + 2) % 3, thetas[:, i])
    mat[:, :3, 3] = mat[:, :3, 3] + thetas[:, 3:]
    return mat


class CameraParamPrior:
    """train_rgbd.py:192-217."""

    def __init__(self, config):
        self.rotation_range = np.array([config.x_rotate, config.y_rotate, config.z_rotate])
        self.camera_param_range = np.array([config.x_rotate, config.y_rotate, config.z_rotate,
                                            config.x_translate, config.y_translate, config.z_translate])
        self.uniform = config.uniform_distribution

    def sample(self, batch_size):
        half = batch_size // 2
        thetas = np.random.uniform(-1, 1, size=(half, 6))
        eps = np.random.uniform(0, 0.5, size=(half, 6))
        sign = np.random.choice(2, size=(half, 3)) * 2 - 1
        limit = np.clip(1 / (self.rotation_range + 1e-8), 0, 1)       # limit the angle difference
        if self.uniform:
            eps[:, :3] = eps[:, :3] * sign * limit
        else:
            wraps = self.rotation_range == 3.1415
            eps[:, :3] = eps[:, :3] * (sign * wraps + np.abs(sign) * (self.rotation_range != 3.1415)) * limit
        thetas2 = -eps * np.sign(thetas) + thetas
        if self.uniform:
            thetas2 = thetas2 * (-1 <= thetas2) * (thetas2 <= 1) + (-2 - thetas2) * (thetas2 < -1) + \
                      (2 - thetas2) * (thetas2 > 1)
        thetas = np.concatenate([thetas, thetas2], axis=0) * self.camera_param_range[None]
        return thetas.astype("float32")


def downsized_size(stage, max_stage=17):
    """Side length of downsize_real's output (pure arithmetic: running the pooling on a CPU dummy every step wakes the
    host's intra-op thread pool and stalled the launch thread for tens of milliseconds in the fade-in stages)."""
    fl = math.floor(min(stage, max_stage - 1e-8))
    k = (fl - 2) // 2 if fl % 2 == 0 else (fl - 1) // 2
    return 4 * (2 ** (k + 1))


def downsize_real(x_real, stage, max_stage=17):
    """common/utils/pggan.py:6-50 on a device tensor (NCHW fp32)."""
    import math
    size = x_real.shape[2]
    assert x_real.shape[2] == x_real.shape[3]
    from . import net
    stage, alpha = net._split_stage(stage, max_stage)       # alpha: float, or the device scalar of net.alpha_override
    if stage % 2 == 0:
        k = (stage - 2) // 2
        image_size = 4 * (2 ** (k + 1))
        assert image_size <= size
        scale = size // image_size
        return F.avg_pool2d(x_real, scale, scale) if scale > 1 else x_real
    k = (stage - 1) // 2
    lo, hi = 4 * (2 ** k), 4 * (2 ** (k + 1))
    assert hi <= size
    s_lo, s_hi = size // lo, size // hi
    r_lo, r_hi = x_real, x_real
    if s_lo > 1:
        r_lo = net.upsample_planes(F.avg_pool2d(x_real, s_lo, s_lo))
    if s_hi > 1:
        r_hi = F.avg_pool2d(x_real, s_hi, s_hi)
    return (1 - alpha) * r_lo + alpha * r_hi


def _alpha_ctx(st):
    from . import net
    return net.alpha_override(st["alpha"]) if st.get("alpha") is not None else contextlib.nullcontext()


_SHARED_STREAMS = {}


def shared_stream(device, role):
    """The process's ONE side stream / capture stream per device, shared by every updater built in it.  HIP deals a
    process's streams out over a handful of hardware queues in creation order: an updater that made its own pair -- the
    fourth one built in a process, as bench.py's other_configs did -- could get a side stream on the main stream's queue,
    and the two-stream step ran at the one-stream time (configuration 5 on bf16: 21.0 instead of 18.5 ms, exactly
    reproducible)."""
    key = (torch.device(device).index or 0, role)
    if key not in _SHARED_STREAMS:
        _SHARED_STREAMS[key] = torch.cuda.Stream(device=device)
    return _SHARED_STREAMS[key]


class _HostStager:
    """Small host -> device uploads without blocking the host on the stream: a ring of pinned staging buffers,
    each guarded by an event (the reference does a blocking cupy.asarray per step, updater.py:267,315-318)."""

    def __init__(self, shape, dtype, device, depth=4):
        self.dev = torch.empty(shape, dtype=dtype, device=device)
        self.pinned = [torch.empty(shape, dtype=dtype).pin_memory() for _ in range(depth)]
        self.events = [None] * depth
        self.i = 0

    def upload(self, array):
        k = self.i
        self.i = (self.i + 1) % len(self.pinned)
        if self.events[k] is not None:
            self.events[k].synchronize()
        self.pinned[k].copy_(torch.as_tensor(array).reshape(self.dev.shape))
        self.dev.copy_(self.pinned[k], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self.events[k] = ev
        return self.dev


class RGBDUpdater:
    camera_conditioned = True       # RGBUpdater: no pose input, no 3-D consistency loss

    def __init__(self, models, config, **kwargs):
        if len(models) == 2:
            models = list(models) + [None]
        if config.bigan:
            raise AssertionError("bigan is not supported")
        self.gen, self.dis, self.smoothed_gen = models
        self.config = config
        self.smoothing = kwargs.pop("smoothing")
        self.lambda_gp = kwargs.pop("lambda_gp")
        self.total_gpu = kwargs.pop("total_gpu")
        self.prior = kwargs.pop("prior")
        self._optimizers = kwargs.pop("optimizer")
        self._iterators = {"main": kwargs.pop("iterator")}
        lambda_geometric = config.lambda_geometric if config.lambda_geometric else 3
        self.loss_func_rotate = LossFuncRotate(torch, lambda_geometric=lambda_geometric)
        self.loss_func_rotate_feature = LossFuncRotate(torch, norm="l2", lambda_geometric=lambda_geometric)   # updater.py:240
        self.stage_interval = list(map(int, str(config.stage_interval).split(",")))
        self.camera_param_range = np.array([config.x_rotate, config.y_rotate, config.z_rotate,
                                            config.x_translate, config.y_translate, config.z_translate])
        self.iteration = 0
        self.observation = {}
        # the reference asserts not-NaN (a host sync) three times per step (updater.py:336,360,439); here the losses
        # stay on the device and are checked every `nan_check_interval` iterations (1 = the reference's behaviour)
        self.nan_check_interval = int(kwargs.pop("nan_check_interval", 100))
        # ... and EVERY step their finiteness is folded into a sticky device flag that the host reads one step later through a
        # pinned copy (no synchronisation): a diverging run stops within two steps, not within nan_check_interval
        self.nan_watch = bool(kwargs.pop("nan_watch", self.nan_check_interval > 0))
        self._nan_state = None
        self.fixed_stage = kwargs.pop("fixed_stage", None)   # bench / tests: pin the stage
        # HIP graphs: the step body and the optimizer phase are captured once per (batch, stage, loss flags) and
        # replayed; collectives stay outside the graphs.  A refused capture is an error unless graph_fallback is set.
        self.use_graphs = bool(kwargs.pop("use_graphs", True))
        self.graph_warmup = int(kwargs.pop("graph_warmup", 2))
        self.graph_fallback = bool(kwargs.pop("graph_fallback", False))
        # TWO arrangements of the same phases:
        #   two streams (default): the generator phase and the discriminator-on-reals phase are independent until the
        #     optimizer phase; on two streams the bubbles of one fill with the other's kernels.  One graph, fork and
        #     join inside it.
        #   one stream (concurrent_phases=False / RGBD_CONCURRENT_PHASES=0; the default of the shared-device test arrangement,
        #     unless RGBD_CONCURRENT_PHASES=1 asks for two streams there too):
        #     the phases back to back; under data parallelism the body is split where G's gradients are final so that
        #     the map + gen all-reduces travel under D's half (dp_split_body).
        env = os.environ.get("RGBD_CONCURRENT_PHASES")
        default = (env not in ("", "0")) if env is not None else not os.environ.get("RGBD_SHARE_DEVICE")
        self.concurrent_phases = bool(kwargs.pop("concurrent_phases", default))
        self.dp_split_body = bool(kwargs.pop("dp_split_body", True))
        # Two streams: the side stream (D on the reals, D's weight gradients) is NOT the step's critical path (dis + dfw
        # 6.8 ms beside gen_a + gen_b 7.9 ms), but its chip-filling launches -- the persistent 3x3 kernel, the 0.5 ms batched
        # weight-gradient launch -- hold every compute unit while they run, and the generator's dependent chain of small
        # kernels on the main stream waits.  Sized for FEWER compute units they leave the rest to the main stream
        # (profiles/r05/cu_budget_sweep.txt).  0 = all.
        # The weight-gradient launch is the one that pays: 160 of 256 workgroups at the benched shape (4040 -> 4290-4340
        # img/s), fewer where the step is short kernels (B = 8: 64-128, +4-5 %), nearly all where it is long ones (256x256,
        # B = 16: 224, +1 %); None = that dependence as a rule of thumb in the step's pixel count (_side_wgrad_auto).  The 3x3
        # kernels' budget (with grids already cut down to what their number of rounds needs, conv.hip): 224 of 256 is +3 % at
        # B = 8 and at 256x256 and neutral at the benched shape; 192 and below cost there.
        self.side_cu_budget = int(kwargs.pop("side_cu_budget", os.environ.get("RGBD_SIDE_CUS", "224")))
        env = os.environ.get("RGBD_SIDE_WGRAD_WGS")
        self.side_wgrad_workgroups = kwargs.pop("side_wgrad_workgroups", int(env) if env else None)
        # ... of which the SECOND launch (D's weight gradients for the fakes, `dfw`) mostly runs behind the end of the
        # generator's backward on most devices of the pool and is better off with (nearly) the whole chip, while on a fast one it
        # still overlaps it: its own count, by default half way between the first launch's and all (measured: autotune_side_budget)
        env = os.environ.get("RGBD_DFW_WGRAD_WGS")
        self.dfw_wgrad_workgroups = kwargs.pop("dfw_wgrad_workgroups", int(env) if env else None)
        # tune_side_budget: measure the side stream's two weight-gradient workgroup counts on THIS device for every (batch, image
        # size) the run meets, inside the ordinary training steps (SideBudgetTuner below), instead of trusting the rule of thumb --
        # round 5's rule was fitted on three shapes and missed a shape it was not fitted on by up to 20 % (stage 8, B = 32:
        # profiles/r06/cu_budget_sweep_rule_r05.txt).  train_rgbd.py and bench.py switch it on for one-GPU runs that replay graphs on
        # two streams (RGBD_TUNE_SIDE_BUDGET=0 keeps the rule); a bare RGBDUpdater(...) takes the rule; off under data parallelism
        # (the re-captures beside RCCL have never run on more than one device) and with explicit counts.
        env = os.environ.get("RGBD_TUNE_SIDE_BUDGET")
        self.tune_side_budget = bool(kwargs.pop("tune_side_budget", False)) and env not in ("", "0")
        self._tuner = None
        # Data parallel (N > 1) -- the same budgets priced for a step with collectives in it:
        #   dp_reserve_cus: RCCL's kernels need compute units of their own while the all-reduces travel beside the step, and a
        #     persistent launch that claims every CU while some are held by a collective runs its last workgroups as a SECOND
        #     round.  The batched WEIGHT-GRADIENT launches split their work over any number of workgroups, so leaving 16 of 256
        #     costs them 6 %: while comm.active they are never planned for more than cus - dp_reserve_cus.  The 3x3 kernel's grids
        #     are NOT reduced: every layer's tile count is a power of two, so a grid of 240 instead of 256 always adds a whole round
        #     (+12 % at 8 rounds, +50 % at 2, +100 % at 1: the 8-GPU job's B = 8 layers have 1-2) -- as much as the collision it
        #     would avoid, paid on every launch instead of the few that meet a collective.
        #   dp_side_lead_workgroups: D's 34 MB should be on the wire BEFORE the generator's backward ends (then only the
        #     generator's 29 MB are exposed); the side stream's weight-gradient launches get this many workgroups more than the
        #     one-GPU rule gives them, so that the side stream ends earlier than the main one instead of together with it.
        # Neither could be measured (one GPU per lease: a one-rank RCCL group moves nothing); both are arguments.
        self.dp_reserve_cus = int(kwargs.pop("dp_reserve_cus", os.environ.get("RGBD_DP_RESERVE_CUS", "16")))
        self.dp_side_lead_workgroups = int(kwargs.pop("dp_side_lead_workgroups", os.environ.get("RGBD_DP_SIDE_LEAD_WGS", "32")))
        if kwargs:
            raise TypeError(f"RGBDUpdater: unknown arguments {sorted(kwargs)}")
        self._side_stream = self._capture_stream = None
        self._graphs, self._eager_calls, self._stagers, self._ones = {}, {}, {}, {}
        self._warp_ws = {}
        self.device = self.gen.device

    # ---- chainer StandardUpdater surface
    def get_optimizer(self, name):
        return self._optimizers[name]

    def get_iterator(self, name):
        return self._iterators[name]

    @property
    def stage(self):
        return self.get_stage()

    def get_stage(self):
        """updater.py:252-256."""
        if self.fixed_stage is not None:
            return self.fixed_stage
        for i, interval in enumerate(self.stage_interval):
            if self.iteration + 1 <= interval:
                prev = self.stage_interval[i - 1]
                return i - 1 + (self.iteration - prev) / (interval - prev)
        return self.config.max_stage - 1e-8

    def get_x_real_data(self, batch, batch_size):
        """updater.py:259-268; device tensors pass straight through (no host round trip)."""
        if torch.is_tensor(batch):
            return batch.to(self.device, torch.float32)
        rows = []
        for i in range(batch_size):
            inst = batch[i]
            if isinstance(inst, tuple):
                inst = inst[0]
            rows.append(np.asarray(inst).astype("f"))
        return torch.from_numpy(np.stack(rows)).to(self.device)

    def get_z_fake_data(self, batch_size):
        return self.gen.make_hidden(batch_size)

    def update(self):
        tuner = self._tuner
        if tuner is not None:
            tuner.before_step()
        self.update_core()
        self.iteration += 1
        if self._tuner is not None:
            if self._tuner.after_step():
                self._tuner = None

    @property
    def tuning_in_progress(self):
        return self._tuner is not None

    def finish_tuning(self, max_steps=400):
        """Run ordinary training steps until the side-budget measurement of the current shape has finished (bench.py: before
        anything is timed).  -> number of steps taken."""
        n = 0
        if self.tune_side_budget and self._tuner is None and n < max_steps:
            self.update()                       # a first step makes the shape known (and starts the tuner if one is due)
            n += 1
        while self._tuner is not None and n < max_steps:
            self.update()
            n += 1
        return n

    NAN_KEYS = ("gen/loss_adv", "gen/loss_rotate", "dis/loss_adv")       # updater.py:336,360,439

    def _nan_watch_poll(self, block=False):
        """Raise if a step whose flag has reached the host reported a non-finite loss (block=True: wait for the newest)."""
        w = self._nan_state
        if w is None:
            return
        if block:
            w["event"].synchronize()
        if w["event"].query() and int(w["host"][0]) != 0:
            bad = [k for i, k in enumerate(w["keys"]) if int(w["host"][0]) >> i & 1]
            raise AssertionError(f"{', '.join(bad)} not finite at or before iteration {w['iteration']}")

    def _end_of_step_checks(self):
        """The non-finite checks every updater runs behind a step (DeepVoxelsUpdater too): the per-step device-side watch
        (no host synchronisation) and the reference's host check every nan_check_interval steps."""
        kernels.STATS_POOL.end_step()       # takes behind this point (previews, tests) are not the step's
        if self.nan_watch:
            self._nan_watch_poll()          # what the previous steps reported, if it has arrived (never waits)
            self._nan_watch_push()
        if self.nan_check_interval > 0 and (self.iteration + 1) % self.nan_check_interval == 0:
            self._check_finite()

    def assert_finite_so_far(self):
        """Wait for the newest step's flag (train_rgbd.py calls this before it writes a snapshot and after the last step: a
        NaN produced by the step in front of a snapshot must not be saved)."""
        if getattr(self, "nan_watch", False):
            self._nan_watch_poll(block=True)

    def _nan_watch_push(self):
        keys = [k for k in self.NAN_KEYS if torch.is_tensor(self.observation.get(k)) and self.observation[k].is_cuda
                and self.observation[k].dtype == torch.float32]
        if not keys:
            return
        if self._nan_state is None:
            self._nan_state = {"mask": torch.zeros(1, dtype=torch.int32, device=self.device),
                               "host": torch.zeros(1, dtype=torch.int32).pin_memory(), "event": torch.cuda.Event()}
        w = self._nan_state
        kernels.nonfinite_mask([self.observation[k].detach().reshape(1) for k in keys], w["mask"])
        w["host"].copy_(w["mask"], non_blocking=True)
        w["event"].record()
        w["keys"], w["iteration"] = keys, self.iteration

    def _check_finite(self):
        for key in ("gen/loss_adv", "gen/loss_rotate", "dis/loss_adv"):
            v = self.observation.get(key)
            if v is not None and not bool(torch.isfinite(v)):
                raise AssertionError(f"{key} is not finite at iteration {self.iteration}")

    def _stager(self, name, shape, dtype=torch.float32):
        key = (name, tuple(shape))
        if key not in self._stagers:
            self._stagers[key] = _HostStager(shape, dtype, self.device)
        return self._stagers[key]

    # ---- the phases of a step: prep, gen_a -> gen_b || dis -> dfw, join, opt (each one is capturable: device work only,
    #      fixed launch sequence)
    def _prep_phase(self, st):
        """Everything both concurrent phases depend on: cleared gradient buffers, the (down-sized) real batch, and the
        bf16 weight images of both networks (repacked here so neither phase does it behind the other's back)."""
        bufs = []
        for link in (self.gen, self.dis):
            for _, store in link.stores:
                store.zero_grad(defer=bufs)
        kernels.STATS_POOL.begin_step(self.device, bufs)           # + the step's instance-norm statistics scratch
        kernels.zero_multi(bufs)                                   # one launch for all flat gradient buffers
        if st.get("real_idx") is not None:
            # the uint8 data set lives in HBM: gather + x/127.5 - 1 + downsize_real (block means, fade-in blend) in ONE
            # kernel (train_rgbd.py:308-310, common/utils/pggan.py:6-50)
            top = self._net_max_stage
            fl = math.floor(min(st["stage"], top - 1e-8))
            alpha = None
            if fl % 2 == 1:
                alpha = st["alpha"] if st.get("alpha") is not None else float(min(st["stage"], top - 1e-8) - fl)
            st["x_real"] = kernels.real_batch(st["real_data"], st["real_idx"], downsized_size(st["stage"], top), alpha)
        else:
            with torch.no_grad():
                st["x_real"] = downsize_real(st["x_real_full"], st["stage"], self._net_max_stage).contiguous()
        for link in (getattr(self.gen, "gen", self.gen), self.dis):
            group = getattr(link, "pack_group", None)
            if group is not None:
                group.layers[0].packed()

    def _gen_a_phase(self, st):
        """G forward and the ONE pass through D(x_fake), up to the image gradient gx.  D's weight gradients for the fakes
        are only COLLECTED here (st['dfw']): they are issued as one batch by _dfw_phase -- behind the discriminator
        phase on the second stream, or after G's backward on one stream -- instead of lengthening G's dependent chain."""
        st["dfw"] = []
        with _alpha_ctx(st), Fn.deferred_wgrads(st["dfw"]):
            self._gen_forward_and_dfake(st)

    def _gen_b_phase(self, st):
        with _alpha_ctx(st):
            self._gen_backward(st)

    def _dfw_phase(self, st):
        """D's weight gradients for the fakes; on two streams this is the last writer of D's gradients (it follows `dis` on
        the side stream), so the two gradient buffers are merged here and D's all-reduce can start behind it."""
        with kernels.wgrad_workgroups(st.get("dfw_wgrad_wgs", st.get("side_wgrad_wgs", 0))):
            Fn.run_deferred_wgrads(st["dfw"])
        if st.get("concurrent"):
            for _, store in self.dis.stores:
                store.merge_alt()

    def _gen_seeds(self, st, x_d, y_fake, seed_g, seed_d, ratio):
        """D(x_fake) is evaluated and differentiated ONCE per step.  The reference runs the discriminator on the same
        fakes twice with identical weights (updater.py:331,404-405) and back-propagates twice: once from the
        generator's loss (to the image) and once from the discriminator's loss (to D's weights).  D treats samples
        independently (no batch statistics), so both backward passes are the same linear map applied to per-sample
        seeds dL/dy_b; one pass seeded with the discriminator's dL_D/dy_b yields D's weight gradients AND, rescaled
        per sample by (dL_G/dy_b) / (dL_D/dy_b), the generator's image gradient.  -> (gx, ratio)
        (tests/dp_worker.py overrides this with the generator's own seed to check the ratio chain at the logit clamp.)"""
        with contextlib.ExitStack() as stack:
            if st.get("concurrent"):          # D's real-batch gradients are being written on the other stream
                for _, store in self.dis.stores:
                    stack.enter_context(store.alt_grads())
            torch.autograd.backward([y_fake], [seed_d], inputs=[x_d] + list(self.dis.params()))
        return x_d.grad, ratio.reshape(-1).contiguous()

    def _gen_forward_and_dfake(self, st):
        stage, half = st["stage"], st["B"] // 2
        obs = self.observation
        if st["z"] is not None:
            z = st["z"]
        elif hasattr(self.gen, "make_hidden_pairs"):
            z = self.gen.make_hidden_pairs(half)                        # same latent for both views, one launch
        else:
            z_half = self.get_z_fake_data(half)
            z = torch.cat([z_half, z_half], dim=0)
        x_fake = self.gen(z, stage, st["theta9"])
        x_d = x_fake[:, :3].detach().contiguous().requires_grad_(True)
        y_fake = self.dis(x_d, stage=stage)      # (the hidden feature of updater.py:333 feeds only rotate_feature)
        # losses and seeds of both adversarial terms from the logits in one launch; seeds from logits clamped at -60:
        # below that the generator's seed -sigmoid(-y)/B equals -1/B to fp32 precision and the discriminator's
        # sigmoid(y)/B is < 1e-26/B either way, but their ratio stays finite (an unclamped logit of -90 would give 0 * inf)
        heads, seed_g, seed_d, ratio = kernels.gan_logit_heads(y_fake.detach())
        obs["gen/loss_adv"] = heads[0]                                      # loss_func_dcgan_gen(y_fake)
        st["loss_dfake"] = heads[1]                                         # fake term of loss_func_dcgan_dis
        st["gx"], st["ratio"] = self._gen_seeds(st, x_d, y_fake, seed_g, seed_d, ratio)
        st["x_fake"] = x_fake

    def _gen_backward(self, st):
        """3-D consistency loss, depth hinge and the generator's backward.  The output gradient of G is assembled by
        hand, not by autograd: rgbd_image_grad_init writes the adversarial part (per-sample ratio * dL_D/dx_fake on the
        RGB planes, zeros on the depth plane), the warp-loss backward ADDS lambda_rotate * d(loss_rotate)/dx_fake with
        the depth hinge of updater.py:357-359 evaluated in the same kernels, and x_fake.backward(that) runs G's
        backward: a handful of launches where slicing, hinge, scaling and the gradient sums were ~25."""
        cfg, obs = self.config, self.observation
        x_fake, half = st["x_fake"], st["B"] // 2
        gout = kernels.image_grad_init(st["gx"].contiguous(), st["ratio"], x_fake.shape[1])
        if st["use_rotate"]:
            lf = self.loss_func_rotate
            flags = 1 if st["occlusion"] else 0                      # loss_functions.WARP_OCCLUSION
            hinge = float(cfg.lambda_depth) if cfg.lambda_depth > 0 else 0.0
            lambda_rotate = cfg.lambda_rotate if cfg.lambda_rotate else 2
            lambda_rotate = lambda_rotate if st["x_real"].shape[2] <= 128 else lambda_rotate * 2
            xf = x_fake.detach()
            loss_rotate = kernels.warp_loss_fwd(xf[:half], xf[half:], st["coef"], flags, lf.lambda_geometric,
                                                hinge_lambda=hinge, hinge_min=float(cfg.depth_min or 0.0))
            obs["gen/loss_rotate"] = loss_rotate.reshape(())
            kernels.warp_loss_bwd(xf[:half], xf[half:], st["coef"], flags, lf.lambda_geometric, 0.0, 0.0, None,
                                  hinge_lambda=hinge, hinge_min=float(cfg.depth_min or 0.0),
                                  grad_scale=float(lambda_rotate), out=(gout[:half], gout[half:]))
            if cfg.use_occupancy_net_loss:
                raise AssertionError("occupancy-net loss is not supported")
        if cfg.optical_flow:
            raise AssertionError("optical flow loss is not supported")
        # the generator's weight gradients are leaves of this backward pass: collected while it runs, issued after it
        # as one batch (one slab-reduction launch for all of them instead of one per layer)
        wgrads = []
        with Fn.deferred_wgrads(wgrads):
            torch.autograd.backward([x_fake], [gout])
        with kernels.wgrad_workgroups(st.get("main_wgrad_wgs", 0)):      # data parallel: not every CU (dp_reserve_cus)
            Fn.run_deferred_wgrads(wgrads)
        st["x_fake_data"] = x_fake.detach()
        st["x_fake"] = st["gx"] = None                 # drop the autograd graph

    def _dis_phase(self, st):
        with _alpha_ctx(st):
            self._dis_phase_body(st)

    def _dis_phase_body(self, st):
        """D on the reals.  The fake half of loss_func_dcgan_dis is differentiated in the generator phase (its weight
        gradients go to D's gradient buffers, which are cleared at the start of the step); its value is added to the
        report at the join."""
        stage = st["stage"]
        obs = self.observation
        x_real_v = st["x_real"].detach().requires_grad_(True)
        y_real = self.dis(x_real_v, stage=stage)
        real_heads = kernels.gan_logit_heads(y_real.detach())
        reported = real_heads[0][0]                                     # mean softplus(-y_real)
        seed = real_heads[1]                                            # d mean softplus(-y_real) / dy
        if self.dis.sn or not self.lambda_gp > 0:
            # no R1 penalty (updater.py:414): the adversarial term on the reals is back-propagated the ordinary way
            st["dis_reported"] = reported
            torch.autograd.backward([y_real], [seed])
            return
        # loss_dis = softplus(-y_real).mean() + loss_gp.  The adversarial term on the reals is not back-propagated through
        # the recorded forward: its per-sample seeds are folded into the R1 passes (functional.adversarial_injection);
        # only the dense tail after the conv stack takes them the ordinary way.
        ones = self._ones.get(tuple(y_real.shape))
        if ones is None:
            ones = self._ones[tuple(y_real.shape)] = torch.ones_like(y_real)
        with Fn.input_grads_only(), Fn.adversarial_injection(seed):
            grad_x, = torch.autograd.grad([y_real], [x_real_v], [ones], create_graph=True)   # chainer.grad seeds ones
        # updater.py:416-418 + loss_functions.py:7-8: lambda * mean_b (sqrt(sum g_b^2))^2, one reduction
        loss_gp = Fn.r1_penalty(grad_x, self.lambda_gp)
        obs["dis/loss_gp"] = loss_gp.detach()
        st["dis_reported"] = reported + loss_gp.detach()
        torch.autograd.backward([y_real], [seed], inputs=self.dis.tail_params(), retain_graph=True)
        wgrads = []
        with Fn.adversarial_injection(seed), Fn.deferred_wgrads(wgrads):
            # d loss_gp / d grad_x = 2 lambda / B * grad_x, handed to the double backward directly
            ggx = kernels.scale_by_scalar(grad_x.detach().contiguous(), None, 2.0 * self.lambda_gp / grad_x.shape[0])
            torch.autograd.backward([grad_x], [ggx])
        # D's weight gradients are leaves of the double backward: one partial-sum launch for all of them
        # (rgbd_conv2d_wgrad_partial_multi_bf16: one slab per CU for the whole pass instead of per layer)
        with kernels.wgrad_workgroups(st.get("side_wgrad_wgs", 0)):
            Fn.run_deferred_wgrads(wgrads)

    def _join_phase(self, st):
        """After both phases: the reported discriminator loss gets its fake half."""
        self.observation["dis/loss_adv"] = st["dis_reported"] + st["loss_dfake"]

    def _prep_only_phase(self, st):
        with self._range("prep"), _alpha_ctx(st):     # fade-in: downsize_real blends with the device-resident alpha too
            self._prep_phase(st)

    def _body_phase(self, st):
        """One stream: prep, generator phase, D's weight gradients for the fakes, discriminator-on-reals phase, join."""
        self._prep_only_phase(st)
        self._body_g_tail(st)
        self._body_d_phase(st)

    def _body_g_tail(self, st):
        rng = self._range
        with rng("gen_a"):
            self._gen_a_phase(st)
        with rng("gen_b"):
            self._gen_b_phase(st)

    def _body_g_phase(self, st):
        """Data parallel, one stream: the body up to the point where map / gen gradients are final."""
        self._prep_only_phase(st)
        self._body_g_tail(st)

    def _body_d_phase(self, st):
        """... and the rest of it: everything that writes D's gradients."""
        rng = self._range
        with rng("dfw"):
            self._dfw_phase(st)
        with rng("dis"):
            self._dis_phase(st)
        with rng("join"):
            self._join_phase(st)

    @contextlib.contextmanager
    def _range(self, name):
        if not self.profile_ranges:
            yield
            return
        torch.cuda.nvtx.range_push(f"rgbd/{name}")
        try:
            yield
        finally:
            torch.cuda.nvtx.range_pop()

    def _opt_phase(self, st):
        self._opt_g_phase(st)
        self._opt_d_phase(st)

    def _opt_g_phase(self, st):
        for name in ("map", "gen"):
            if name in self._optimizers:
                self._optimizers[name].update()
        if self.smoothed_gen is not None:          # updater.py:397-400 (after the generator update; D never touches G)
            soft_copy_param(self.smoothed_gen, self.gen, 1.0 - self.smoothing)

    def _opt_d_phase(self, st):
        self._optimizers["dis"].update(bump=not st.get("d_step_on_side"))

    tune_side_budget = False    # class defaults shared with DeepVoxelsUpdater (its own __init__)
    _tuner = None
    timeline = None             # set to a dict: update_core leaves timing events of its last step there (tests, scripts)
    call_log = None             # set to a list: update_core appends (what, name, current stream handle) in HOST ORDER as it
                                # launches phases and starts all-reduces (tests pin the order; no timing involved)

    def _note(self, what, name):
        if self.call_log is not None:
            self.call_log.append((what, name, torch.cuda.current_stream().cuda_stream if torch.cuda.is_available() else 0))

    def _dp_budgets(self, st):
        """Data parallel -> the cap on every weight-gradient plan's workgroup count (cus - dp_reserve_cus); also moves the side
        stream's two counts in st by dp_side_lead_workgroups (rule-given counts only) and caps them (__init__)."""
        cus = torch.cuda.get_device_properties(self.device).multi_processor_count if torch.cuda.is_available() else 256
        cap = max(8, cus - max(0, self.dp_reserve_cus))
        for k in ("side_wgrad_wgs", "dfw_wgrad_wgs"):
            if self.side_wgrad_workgroups is None and st.get(k):
                st[k] = int(st[k]) + self.dp_side_lead_workgroups
            st[k] = min(cap, int(st.get(k) or cap))
        return cap

    def _mark(self, name, stream):
        if self.timeline is not None:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record(stream)
            self.timeline[name] = ev

    @property
    def _net_max_stage(self):
        """The networks' own stage ceiling: 17 for the reference's six blocks (net.py:166,433), 19 with a 256 px block."""
        return getattr(self.dis, "max_stage", 17)

    graph_fallback = False      # class defaults shared with DeepVoxelsUpdater (its own __init__)
    _capture_stream = None
    _replayed = False           # a captured phase ran since update_core started (it changed weights behind Python's back)
    profile_ranges = False      # train_rgbd.py sets it for `nvprof` / `enable_cuda_profiling` (train_rgbd.py:100,363-364,462)

    @property
    def graphs_in_use(self):
        """True once the step is being replayed from captured HIP graphs (bench.py reports it)."""
        return bool(self.use_graphs and self._graphs)

    def _side_wgrad_auto(self, st):
        """Workgroups of the side stream's batched weight-gradient launches when nothing has been MEASURED for this shape on this
        device (SideBudgetTuner / autotune_side_budget): a line through the measured optima of five shapes, 112 + pixels / 10240 of
        256 compute units in multiples of 8 (one per XCD) -- 128 / 136 / 160 / 216 at 32 x 64^2 / 16 x 128^2 / 32 x 128^2 / 16 x 256^2,
        where 128-144 / 144 / 144-176 / 208-224 were fastest (profiles/r06/cu_budget_sweep.txt; round 5's three-point curve gave
        64 at the first two and lost 20 % / 5 % there).  A prior, not a model: the measurement is the default (tune_side_budget)."""
        return self._side_wgrad_pair(st)[0]

    def _side_wgrad_pair(self, st):
        """-> (workgroups of the `dis` launch, of the `dfw` launch): measured (autotune_side_budget) or by rule."""
        shape = (int(st["B"]), int(st["x_real"].shape[2]), int(st["x_real"].shape[3]))
        tuned = getattr(self, "_side_wgrad_tuned", {}).get(shape)
        if tuned is not None:
            return tuned
        cus = torch.cuda.get_device_properties(self.device).multi_processor_count if torch.cuda.is_available() else 256
        px = float(shape[0]) * float(shape[1]) * float(shape[2])
        w = min(cus, max(64, int(round((0.4375 * cus + px / 10240.0 * cus / 256.0) / 8.0)) * 8))
        return w, self._dfw_rule(w, cus)

    @staticmethod
    def _dfw_rule(w, cus):
        return int(round((w + cus) / 16.0)) * 8                 # half way to the whole chip, a multiple of 8

    def autotune_side_budget(self, measure_steps=12, log=None):
        """Measure, on THIS device and at the CURRENT stage / batch, the workgroup count of the side stream's weight-gradient
        launches instead of taking the rule of thumb: the rule's value and five neighbours for the first launch (-32, +32, +64,
        then best -16 / +16; the second launch at half way to the whole chip), then two more for the second launch (the same as the
        first; the whole chip), each timed over `measure_steps` replayed steps behind a re-capture; the fastest pair is kept for
        this (batch, image size) (_side_wgrad_pair).  graph_warmup + 1 + 8 x (2 + measure_steps) ordinary training steps (115 by default) -- they update
        the networks like any others -- and nine re-captures of the step's graphs: one to four seconds.  The number of steps is
        fixed, whatever is measured, so the ranks of a data-parallel job stay in step (each keeps its own optimum).  No-op (returns None) without two streams + graphs, or with an explicit side_wgrad_workgroups."""
        import time
        if not (self.concurrent_phases and self.use_graphs and self.side_wgrad_workgroups is None and torch.cuda.is_available()):
            return None
        self.tune_side_budget, self._tuner = False, None         # this IS the measurement: the in-loop one stays out of its way
        for _ in range(self.graph_warmup + 1):                 # the shape's graphs exist, _last_shape is known
            self.update()
        shape = self._last_shape
        cus = torch.cuda.get_device_properties(self.device).multi_processor_count
        self._side_wgrad_tuned = getattr(self, "_side_wgrad_tuned", {})
        self._side_wgrad_tuned.pop(shape, None)
        w0 = self._side_wgrad_pair({"B": shape[0], "x_real": torch.empty(0, 0, shape[1], shape[2])})[0]
        clamp = lambda w: int(min(cus, max(32, w)))
        results = {}

        def timed(pair):
            self._side_wgrad_tuned[shape] = pair
            self._graphs.clear()                                # every phase is re-captured (the side phases bake the counts in)
            for _ in range(2):
                self.update()
            torch.cuda.synchronize(self.device)
            t0 = time.perf_counter()
            for _ in range(measure_steps):
                self.update()
            torch.cuda.synchronize(self.device)
            t = (time.perf_counter() - t0) / measure_steps
            results[pair] = min(t, results.get(pair, t))
            if log is not None:
                log(f"autotune_side_budget {shape}: dis {pair[0]} / dfw {pair[1]} workgroups {1e3 * t:.3f} ms per step")

        try:
            for d in (0, -32, 32, 64):                          # the first launch's count, the second at its rule
                w = clamp(w0 + d)
                timed((w, self._dfw_rule(w, cus)))
            best = min(results, key=results.get)
            for d in (-16, 16):
                w = clamp(best[0] + d)
                timed((w, self._dfw_rule(w, cus)))
            best = min(results, key=results.get)
            for dfw in (best[0], cus):                          # ... then the second launch's: the same as the first, the whole chip
                timed((best[0], dfw))
            best = min(results, key=results.get)
        except Exception:
            self._side_wgrad_tuned.pop(shape, None)             # back to the rule of thumb, nothing half-measured kept
            self._graphs.clear()
            raise
        self._side_wgrad_tuned[shape] = best
        self._graphs.clear()
        self.side_budget_tuning = {"shape": shape, "rule": (w0, self._dfw_rule(w0, cus)), "chosen": best,
                                   "ms_per_step": {f"{k[0]}/{k[1]}": round(1e3 * v, 4) for k, v in sorted(results.items())}}
        return best

    def _run_phase(self, name, fn, st, key, stream=None, cu_budget=0):
        """Eager for the first calls of a configuration, then capture once and replay -- on `stream` (default: the current
        one).  With profile_ranges every phase is bracketed by a roctx range (torch.cuda.nvtx maps to roctx on ROCm),
        visible to rocprofv3 --marker-trace.  cu_budget: compute units the phase's chip-filling launches size their grids
        for (kernels.cu_budget: the `cus` argument of every conv launch of the phase; fixed at capture)."""
        with self._range(name), (torch.cuda.stream(stream) if stream is not None else contextlib.nullcontext()):
            self._note("phase", name)
            self._run_phase_inner(name, self._budgeted(fn, cu_budget), st, key)

    @staticmethod
    def _budgeted(fn, cu_budget):
        """fn with its conv launches sized for `cu_budget` compute units.  The budget is entered where fn RUNS -- on the
        replay stream when eager, on the capture stream inside a capture -- because kernels.cu_budget belongs to the stream
        that is current at entry."""
        if not cu_budget:
            return fn

        def run(st):
            with kernels.cu_budget(cu_budget):
                fn(st)
        return run

    def _run_phase_inner(self, name, fn, st, key):
        if key is None:
            fn(st)
            return
        gkey = key + (name,)
        entry = self._graphs.get(gkey)
        if entry is None:
            n = self._eager_calls.get(gkey, 0)
            if n < self.graph_warmup:
                self._eager_calls[gkey] = n + 1
                fn(st)
                return
            graph = torch.cuda.CUDAGraph()
            saved = dict(self.observation)
            # data parallel: let the collectives that overlap this phase finish first, and keep the capture local to
            # this thread so the communicator's watchdog thread (event queries) cannot invalidate it
            for opt in self._optimizers.values():
                if getattr(opt, "_pending", None) is not None:
                    opt.comm.wait(opt._pending)
            try:
                # ONE capture stream for every phase, whichever stream replays it: autograd runs a leaf's AccumulateGrad on
                # the stream that leaf's accumulator was created on, so phases captured on different streams that share
                # parameters (D in gen_a and in dis) could come out with a fork / join around an accumulation; captured
                # on one stream every phase is a plain chain of kernel nodes (checked: hipGraphGetEdges = nodes - 1)
                if self._capture_stream is None:
                    self._capture_stream = shared_stream(self.device, "capture")
                with torch.cuda.graph(graph, stream=self._capture_stream, capture_error_mode="thread_local"):
                    fn(st)
            except Exception as exc:
                if not self.graph_fallback:
                    raise RuntimeError(f"HIP graph capture of phase '{name}' failed ({type(exc).__name__}: {exc}); pass "
                                       "use_graphs=False (eager step) or graph_fallback=True to continue") from exc
                import sys                # capture refused (driver / collective library state): stay correct, go eager
                print(f"[rgbd_gan_amd] HIP graph capture of phase '{name}' failed ({type(exc).__name__}: {exc}); "
                      "continuing without graphs", file=sys.stderr, flush=True)
                self.use_graphs = False
                self._graphs.clear()
                torch.cuda.synchronize()
                fn(st)
                return
            # keep every tensor the phase handed over alive: it lives in the graph's private pool
            entry = {"graph": graph, "st": dict(st), "obs": {k: v for k, v in self.observation.items()
                                                             if saved.get(k) is not v}}
            self._graphs[gkey] = entry
        else:
            st.update({k: v for k, v in entry["st"].items() if k in ("x_real", "x_fake_data", "loss_dfake", "dfw", "fwd_x_fake")})
            self.observation.update(entry["obs"])
        entry["graph"].replay()
        self._replayed = True

    def _feature_rotation_loss(self, feat, x_real, cams, occlusion):
        """updater.py:345-353 / 423-431 (`rotate_feature`): the 3-D consistency loss, L2 criterion, on the discriminator's
        hidden features of the two views (input of block 3: 256 x 32 x 32) with ONE more channel appended as their
        "depth" -- the reference takes it from the average-pooled last channel of the REAL batch (x_real[:, -1:], i.e. the
        blue plane of unrelated real images); restated as written."""
        half = feat.shape[0] // 2
        rate = x_real.shape[2] // feat.shape[2]
        depth = F.avg_pool2d(x_real[:, -1:], rate, rate) if rate > 1 else x_real[:, -1:]
        f = torch.cat([feat, depth], dim=1)
        loss, _ = self.loss_func_rotate_feature(f[:half], cams[:half], f[half:], cams[half:], occlusion)
        return loss, f

    def _literal_step(self, st, opt_g_m, opt_g_g, opt_d, cams):
        """The reference's LITERAL step (updater.py:274-448) over the engine's differentiable ops -- D on the fakes in the
        generator step, D on the same fakes again and on the reals in the discriminator step, R1 through autograd's
        double backward -- for the options the single-pass dataflow of the default step does not cover:
          * a spectral-norm discriminator (`sn: True`): every forward call of an SN layer runs a power iteration and moves
            its persistent vector, i.e. changes the function, so D(x_fake) evaluated once is not what the reference
            computes; no R1 penalty then (updater.py:414);
          * `rotate_feature` (updater.py:345-354,423-437): the feature-consistency term enters the generator's and (negated)
            the discriminator's loss at D's hidden layer, with a second gradient penalty on the features -- two
            differently seeded backward passes through D's first blocks.
        Eager, one stream; both options are off in every shipped config."""
        cfg, obs = self.config, self.observation
        stage, half = st["stage"], st["B"] // 2
        rf = bool(cfg.rotate_feature) and st["use_rotate"]
        r1 = not self.dis.sn and self.lambda_gp > 0
        self._prep_phase(st)
        x_real = st["x_real"]
        z = st["z"]
        if z is None:
            z_half = self.get_z_fake_data(half)
            z = torch.cat([z_half, z_half], dim=0)
        x_fake = self.gen(z, stage, st["theta9"])
        if rf:
            y_fake, feat = self.dis(x_fake[:, :3].contiguous(), stage=stage, return_hidden=True)
            if feat is None:
                raise AssertionError("rotate_feature needs the hidden features of block 3: stage >= 6")
        else:
            y_fake = self.dis(x_fake[:, :3].contiguous(), stage=stage)
        loss_gen = loss_func_dcgan_gen(y_fake)
        obs["gen/loss_adv"] = loss_gen.detach()
        if st["use_rotate"]:
            loss_rotate, _ = self.loss_func_rotate(x_fake[:half], cams[:half], x_fake[half:], cams[half:], st["occlusion"])
            if rf:
                loss_rotate = loss_rotate + self._feature_rotation_loss(feat, x_real, cams, st["occlusion"])[0]
            if cfg.lambda_depth > 0:
                loss_rotate = loss_rotate + torch.mean(F.relu(cfg.depth_min - x_fake[:, -1]) ** 2) * cfg.lambda_depth
            obs["gen/loss_rotate"] = loss_rotate.detach()
            lambda_rotate = cfg.lambda_rotate if cfg.lambda_rotate else 2
            lambda_rotate = lambda_rotate if x_real.shape[2] <= 128 else lambda_rotate * 2
            loss_gen = loss_gen + loss_rotate * lambda_rotate
        loss_gen.backward()
        for opt in (opt_g_m, opt_g_g):
            if opt is not None:
                opt.update()
        self.dis.cleargrads()                           # updater.py:395
        if self.smoothed_gen is not None:
            soft_copy_param(self.smoothed_gen, self.gen, 1.0 - self.smoothing)

        v_x_fake = x_fake.detach()[:, :3].contiguous().requires_grad_(rf and r1)
        if rf:
            y_fake, feat = self.dis(v_x_fake, stage=stage, return_hidden=True)
        else:
            y_fake = self.dis(v_x_fake, stage=stage)
        x_real_v = x_real.detach().requires_grad_(r1)
        y_real = self.dis(x_real_v, stage=stage)
        loss_dis = loss_func_dcgan_dis(y_fake, y_real)
        if r1:
            with Fn.input_grads_only():
                grad_x, = torch.autograd.grad([y_real], [x_real_v], [torch.ones_like(y_real)], create_graph=True)
            loss_gp = self.lambda_gp * torch.mean(torch.sum(grad_x ** 2, dim=(1, 2, 3)))   # loss_l2(sqrt(sum g^2), 0)
            obs["dis/loss_gp"] = loss_gp.detach()
            loss_dis = loss_dis + loss_gp
        if rf:
            loss_rf, f257 = self._feature_rotation_loss(feat, x_real, cams, st["occlusion"])
            loss_dis = loss_dis - loss_rf
            if r1:
                with Fn.input_grads_only():
                    grad_f, = torch.autograd.grad([f257], [v_x_fake], [torch.ones_like(f257)], create_graph=True)
                loss_dis = loss_dis + self.lambda_gp * torch.mean(torch.sum(grad_f ** 2, dim=(1, 2, 3)))
        obs["dis/loss_adv"] = loss_dis.detach()
        loss_dis.backward()
        opt_d.update()

    # ---- the step
    def update_core(self, batch=None, z_fake_data=None, thetas=None):
        """Optional arguments let tests and the benchmark inject fixed inputs; by default they are drawn exactly as
        the reference draws them (iterator, make_hidden, prior.sample)."""
        cfg = self.config
        stylegan = cfg.generator_architecture == "stylegan"
        opt_g_m = self.get_optimizer("map") if stylegan else None
        opt_g_g = self.get_optimizer("gen")
        opt_d = self.get_optimizer("dis")
        stage = self.stage
        real_idx = real_data = None
        it = self.get_iterator("main") if batch is None else None
        if it is not None and hasattr(it, "next_indices") and it.data.shape[1] == 3 and it.data.shape[2] == it.data.shape[3]:
            # data set resident in HBM: only the batch's indices move; the gather + normalise + down-size is one kernel
            # of the prep phase (reading the indices from a fixed buffer, so the phase can be replayed as a graph)
            idx = it.next_indices()
            batch_size = int(idx.numel())
            key_i = ("real_idx", batch_size)
            if key_i not in self._stagers:
                self._stagers[key_i] = torch.empty(batch_size, dtype=torch.int64, device=self.device)
            self._stagers[key_i].copy_(idx)
            real_idx, real_data = self._stagers[key_i], it.data
            x_real_data = None
            full_shape = (batch_size,) + tuple(it.data.shape[1:])
        else:
            if batch is None:
                batch = it.next()
            batch_size = len(batch)
            x_real_data = self.get_x_real_data(batch, batch_size)
            full_shape = tuple(x_real_data.shape)
        half = batch_size // 2

        # ---- host side, NumPy, exactly as the reference: pose prior, camera matrices, pose code, warp constants
        if self.camera_conditioned:
            if thetas is None:
                thetas = self.prior.sample(batch_size)
            thetas = np.asarray(thetas, dtype="float32")
            random_camera_matrices = get_camera_matries(thetas)             # (B,4,4) fp32
            theta9 = np.concatenate([np.cos(thetas[:, :3]), np.sin(thetas[:, :3]), thetas[:, 3:]],
                                    axis=1).astype("float32")
            use_rotate = self.iteration > cfg.start_rotation
            occlusion = self.iteration >= cfg.start_occlusion_aware
        else:                                     # RGBUpdater (updater.py:504-589): the prior is never sampled
            theta9, use_rotate, occlusion = None, False, False
        st = {"stage": stage, "B": batch_size, "use_rotate": use_rotate, "occlusion": occlusion,
              "x_real_full": x_real_data, "z": None, "real_idx": real_idx, "real_data": real_data}
        st["theta9"] = self._stager("theta9", (batch_size, 9)).upload(theta9) if theta9 is not None else None
        if use_rotate:
            image_size = downsized_size(stage, self._net_max_stage)
            coef = self.loss_func_rotate.coefficients_for_size(image_size, random_camera_matrices[:half],
                                                               random_camera_matrices[half:])
            st["coef"] = self._stager("coef", (half, 24)).upload(coef)
        if z_fake_data is not None:
            z_in = torch.as_tensor(z_fake_data).to(self.device, torch.float32)
            zkey = ("z",) + tuple(z_in.shape)
            if zkey not in self._stagers:
                self._stagers[zkey] = torch.empty_like(z_in)
            self._stagers[zkey].copy_(z_in)
            st["z"] = self._stagers[zkey]

        if getattr(self.dis, "sn", False) or (self.camera_conditioned and cfg.rotate_feature):
            self._literal_step(st, opt_g_m, opt_g_g, opt_d, random_camera_matrices if self.camera_conditioned else None)
            kernels.STATS_POOL.end_step()
            obs = self.observation
            obs["stage"], obs["batch_size"], obs["image_size"] = stage, batch_size, int(st["x_real"].shape[2])
            return

        fl = math.floor(min(stage, self._net_max_stage - 1e-8))
        key = None
        st["alpha"] = None
        if self.use_graphs:
            if fl % 2 == 1:
                # fade-in stage: the blend factor changes every iteration, so it lives in a device scalar that the
                # captured phases read (net.alpha_override); the launch sequence depends on floor(stage) only
                alpha = float(min(stage, self._net_max_stage - 1e-8) - fl)
                st["alpha"] = self._stager("alpha", (1,)).upload(np.array([alpha], dtype="float32"))[0]
            if x_real_data is not None:
                # graphs read their inputs from fixed addresses: park the batch in a persistent buffer
                skey = ("x_real_full",) + tuple(x_real_data.shape)
                if skey not in self._stagers:
                    self._stagers[skey] = torch.empty_like(x_real_data)
                self._stagers[skey].copy_(x_real_data)
                st["x_real_full"] = self._stagers[skey]
            # (the conv dtype is baked into a capture: a set_conv_dtype after the graphs exist gets graphs of its own)
            key = (batch_size, fl, use_rotate, occlusion, full_shape, z_fake_data is not None, real_idx is not None,
                   Fn.conv_dtype())

        st["concurrent"] = self.concurrent_phases
        dp = getattr(opt_d, "comm", None) is not None and opt_d.comm.active
        g_opts = [o for o in (opt_g_m, opt_g_g) if o is not None]
        if st["concurrent"]:
            # TWO STREAMS.  Every phase is its own captured graph, launched on the stream it belongs to, and the fork / join
            # dependencies are stream events between the launches:
            #     main:  prep -> gen_a ---------------------> gen_b -> [all-reduce map, gen] -> opt_g -> (wait side) -> join
            #     side:       \-> dis --------\-> dfw + merge -> [all-reduce dis] -> opt_d ---------/
            # ([..]: data parallel only.)  Same step time as ONE graph with the fork inside it (rounds 1-2); per-phase graphs let
            # every optimizer's all-reduce start the moment ITS gradients are final, on the stream that produced them.  (Overlap: events,
            # scripts/phase_timeline.py, or a kernel trace through scripts/trace_overlap.py -- a small kernel's duration
            # in a trace is its stretched length under the other queue's chip-filling kernel, not its cost.)
            if self._side_stream is None:
                self._side_stream = shared_stream(self.device, "side")
            main, side = torch.cuda.current_stream(), self._side_stream
            self._run_phase("prep", self._prep_only_phase, st, key)
            side.wait_stream(main)
            if self.side_wgrad_workgroups is None:
                st["side_wgrad_wgs"], st["dfw_wgrad_wgs"] = self._side_wgrad_pair(st)
            else:
                st["side_wgrad_wgs"] = st["dfw_wgrad_wgs"] = int(self.side_wgrad_workgroups)
            if self.dfw_wgrad_workgroups is not None:
                st["dfw_wgrad_wgs"] = int(self.dfw_wgrad_workgroups)
            side_cus, main_cus = self.side_cu_budget, 0
            if dp:
                st["main_wgrad_wgs"] = self._dp_budgets(st)
            self._last_shape = (int(st["B"]), int(st["x_real"].shape[2]), int(st["x_real"].shape[3]))
            if (self.tune_side_budget and self._tuner is None and key is not None and not dp and self.side_wgrad_workgroups is None
                    and self.dfw_wgrad_workgroups is None and self._last_shape not in getattr(self, "_side_wgrad_tuned", {})):
                self._tuner = SideBudgetTuner(self, self._last_shape)        # measures from the NEXT step on
            self._run_phase("dis", self._dis_phase, st, key, stream=side,         # D on the reals, R1, its weight gradients
                            cu_budget=side_cus)
            self._run_phase("gen_a", self._gen_a_phase, st, key, cu_budget=main_cus)   # G forward, the one pass through D(x_fake)
            side.wait_stream(main)
            self._run_phase("dfw", self._dfw_phase, st, key, stream=side,         # D's weight gradients for the fakes, merge
                            cu_budget=side_cus)
            self._mark("side_end", side)
            # D's gradients are final and nothing on the main stream reads D's weights any more (D(x_fake)'s backward finished
            # in gen_a): D's clip + Adam step runs HERE, on the side stream, under the generator's backward.  Data parallel:
            # its 34 MB all-reduce goes in front of it, started the moment the gradients are final, and the side stream -- not
            # the host, not the main stream -- waits for it.
            st["d_step_on_side"] = True
            with torch.cuda.stream(side):
                if dp:
                    self._note("allreduce", "dis")
                    opt_d.start_allreduce()
                    opt_d.finish_allreduce()
                self._mark("opt_d_start", side)
            self._run_phase("opt_d", self._opt_d_phase, st, key, stream=side)
            self._run_phase("gen_b", self._gen_b_phase, st, key, cu_budget=main_cus)   # 3-D loss, G backward, G's weight gradients
            self._mark("gen_b_end", main)
            # the generator's clip + Adam step (+ EMA) needs nothing of the side stream: it runs before the join, under
            # whatever D's weight gradients and D's own step still have to do.  Data parallel: behind the generator's two
            # all-reduces (2 + 27 MB), which are the step's exposed communication -- they cannot start before the last of
            # the generator's weight gradients, and the optimizer cannot start before they end.
            if dp:
                for opt in g_opts:
                    self._note("allreduce", "gen")
                    opt.start_allreduce()
                for opt in g_opts:
                    opt.finish_allreduce()
            self._mark("opt_g_start", main)
            self._run_phase("opt_g", self._opt_g_phase, st, key)
            st["opt_g_done"] = True
            main.wait_stream(side)
            self._run_phase("join", self._join_phase, st, key)
        elif dp:
            st["main_wgrad_wgs"] = self._dp_budgets(st)     # one stream: every weight-gradient plan leaves dp_reserve_cus
            if self.dp_split_body:
                self._run_phase("body_g", self._body_g_phase, st, key)
                for opt in g_opts:          # ~29 MB of generator gradients travel while D's half of the step runs
                    self._note("allreduce", "gen")
                    opt.start_allreduce()
                self._run_phase("body_d", self._body_d_phase, st, key)
            else:
                self._run_phase("body", self._body_phase, st, key)
        else:
            self._run_phase("body", self._body_phase, st, key)
        if st.get("d_step_on_side"):
            pass                        # two streams: both optimizer phases ran inside the arrangement above
        elif dp:
            # data parallel on ONE stream (train_rgbd.py:154-156: the multi-node optimizers all-reduce before they update):
            # one all-reduce per flat gradient buffer on the communicator's stream, the collectives outside the graphs;
            # the generator's Adam step runs while D's 34 MB all-reduce is still in flight
            for opt in g_opts + [opt_d]:
                if getattr(opt, "_pending", None) is None:
                    self._note("allreduce", "gen" if opt is not opt_d else "dis")
                    opt.start_allreduce()
            for opt in g_opts:
                opt.finish_allreduce()
            self._run_phase("opt_g", self._opt_g_phase, st, key)
            opt_d.finish_allreduce()
            self._run_phase("opt_d", self._opt_d_phase, st, key)
        else:
            self._run_phase("opt", self._opt_phase, st, key)
        if key is not None or st.get("d_step_on_side"):
            Fn.bump_weight_epoch(own_step=True)      # replays change the weights behind Python's back: invalidate packed caches

        obs = self.observation
        obs["stage"], obs["batch_size"], obs["image_size"] = stage, batch_size, int(st["x_real"].shape[2])
        self._end_of_step_checks()


class SideBudgetTuner:
    """The set-up measurement of RGBDUpdater.autotune_side_budget spread over ORDINARY training steps: the caller's loop keeps
    calling update() (its log / preview / snapshot triggers fire as always, `iteration` advances as always) and the tuner, from
    inside update(), walks a fixed plan of candidate (dis, dfw) workgroup pairs for ONE (batch, image size): per candidate the
    step's graphs are re-captured with the pair baked in (graph_warmup eager steps + the capturing one), then `measure` replayed
    steps are timed between two device synchronisations.  Plan: the rule's pair and -32 / +32 / +64 for the first launch (the second
    at its rule), best -16 / +16, then two more for the second launch (same as the first; the whole chip): 8 candidates x
    (graph_warmup + 1 + measure) steps = 120 by default, 1-4 s.  The fastest pair is kept for the shape
    (RGBDUpdater._side_wgrad_tuned) -- the rule's own pair if it is within 1 % of the fastest (what a 12-step window resolves) --
    and the run goes on with it.  A stage change in the middle (another image size) drops the
    half-finished measurement -- the new shape gets a tuner of its own, the old one its rule until it comes up again."""

    def __init__(self, upd, shape, measure=12):
        self.upd, self.shape, self.measure = upd, shape, int(measure)
        self.cus = torch.cuda.get_device_properties(upd.device).multi_processor_count
        upd._side_wgrad_tuned = getattr(upd, "_side_wgrad_tuned", {})
        self.rule = upd._side_wgrad_pair({"B": shape[0], "x_real": torch.empty(0, 0, shape[1], shape[2])})
        self.results, self.queue, self.round = {}, [], 0
        self.cand, self.c, self.t0 = None, 0, None
        self._next_round()

    def _clamp(self, w):
        return int(min(self.cus, max(32, w)))

    def _next_round(self):
        """-> False when the plan is exhausted."""
        up, cus = self.upd, self.cus
        best = min(self.results, key=self.results.get) if self.results else None
        if self.round == 0:
            ws = [self._clamp(self.rule[0] + d) for d in (0, -32, 32, 64)]
            self.queue = [(w, up._dfw_rule(w, cus)) for w in ws]
        elif self.round == 1:
            self.queue = [(w, up._dfw_rule(w, cus)) for w in (self._clamp(best[0] - 16), self._clamp(best[0] + 16))]
        elif self.round == 2:
            self.queue = [(best[0], best[0]), (best[0], cus)]
        else:
            return False
        self.round += 1
        self.queue = [p for i, p in enumerate(self.queue) if p not in self.results and p not in self.queue[:i]]
        return True if self.queue else self._next_round()

    def _install(self, pair):
        up = self.upd
        up._side_wgrad_tuned[self.shape] = pair
        up._graphs.clear()                  # every phase is re-captured (the side phases bake the counts in)
        up._eager_calls.clear()

    def before_step(self):
        up = self.upd
        if getattr(up, "_last_shape", self.shape) != self.shape:
            return
        if self.cand is None:
            self.cand, self.c = self.queue.pop(0), 0
            self._install(self.cand)
        if self.c == up.graph_warmup + 1:   # every phase has been captured: the timed replays start here
            torch.cuda.synchronize(up.device)
            import time
            self.t0 = time.perf_counter()

    def after_step(self):
        """-> True when this tuner is finished (measured, or abandoned because the shape changed)."""
        up = self.upd
        if self.cand is None:               # created during this very step: the plan starts with the next one
            return False
        if getattr(up, "_last_shape", self.shape) != self.shape or not up.graphs_in_use and self.c > up.graph_warmup:
            up._side_wgrad_tuned.pop(self.shape, None)      # back to the rule; nothing half-measured is kept
            up._graphs.clear()
            up._eager_calls.clear()
            return True
        self.c += 1
        if self.c < up.graph_warmup + 1 + self.measure:
            return False
        import time
        torch.cuda.synchronize(up.device)
        self.results[self.cand] = (time.perf_counter() - self.t0) / self.measure
        self.cand = None
        if self.queue or self._next_round():
            return False
        best = min(self.results, key=self.results.get)
        rule = tuple(self.rule)
        if rule in self.results and self.results[rule] <= 1.01 * self.results[best]:
            best = rule                     # a 12-step window resolves ~1 %: within that the rule's pair stays (flat landscapes)
        self._install(best)
        up.side_budget_tuning = {"shape": self.shape, "rule": tuple(self.rule), "chosen": best,
                                 "ms_per_step": {f"{k[0]}/{k[1]}": round(1e3 * v, 4) for k, v in sorted(self.results.items())}}
        return True


class RGBUpdater(RGBDUpdater):
    """updater.py:451-589 of the reference (selected by `config.rgb`, train_rgbd.py:357-358): the same generator step +
    discriminator step + R1 without camera poses -- the generator is called as gen(z, stage) and produces 3 channels,
    there is no 3-D consistency loss and no depth hinge, and the pose prior is never sampled.  Everything else (single
    pass through D(x_fake), injection of the adversarial seeds into the R1 double backward, graphs, two streams, data
    parallel all-reduce) is RGBDUpdater's."""
    camera_conditioned = False
