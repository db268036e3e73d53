"""DeepVoxelsUpdater with the reference's surface (updater_deepvoxels.py:76-252): one generator step and one
discriminator step of configs/deepvoxels_shapenet_car.yml on the HIP kernels.

Differences from RGBDUpdater that the reference has and this keeps:
  * the stage is the constant 8.5 (:105-106): the progressive discriminator of net.py runs its 64x64 path;
  * two latents per step, each drawn for half the batch and tiled so both views of a pair share them (:146-148);
  * the generator is RE-RUN with fresh latents for the discriminator step, after its own update (:221-228), so the
    single-pass dataflow of RGBDUpdater (one D(x_fake) evaluation per step) does not apply here;
  * the 3-D loss uses the projection intrinsics as K (:92-93), no occlusion flag (:193-196), weight 0.3 (the config
    lookup at :202 tests a misspelt key, so the YAML value can never be used), focal gamma from the YAML (:170).

The class reuses RGBDUpdater's plumbing (optimizer / iterator accessors, batch conversion, finite checks, pinned
upload rings, HIP-graph capture / replay of device-only phases) and replaces the step.  The reference's discriminator step
is a sum of two terms that share nothing but D's weights: softplus(-D(x_real)) + R1 on the reals, and softplus(D(G'(z)))
on fakes of the UPDATED generator.  Only the second has to wait for the generator step, so the step is four captured phases:

    main:  prep -> gen (G fwd, frozen D, 3-D loss, G bwd, G update) ---------> join -> dis_fake (G' fwd, D fwd/bwd, D update)
    side:       +-> dis_real (D(x_real), R1 double backward, weight gradients) -/

G' = G after its update is also the generator the NEXT step's generator phase starts from, and its forward pass needs nothing of
the discriminator: with `prefetch_forward` (default where the step draws its own poses and latents; RGBD_DV_PREFETCH=0 turns it
off) the generator phase is cut behind its forward pass, and the next step's forward runs on the side stream UNDER this step's
dis_fake -- same arithmetic per step, 1.2 ms off the critical path (profiles/r06/c4_prefetch.txt):

    main:  prep -> (wait fwd(n)) gen_rest(n): frozen D, 3-D loss, G bwd, G update, G's weight images -> (wait dis_real) dis_fake(n)
    side:       +-> dis_real(n) ---------------------------------------------------> (wait G update) fwd(n+1): G'(z) of step n+1

and with `split_backward` (default with prefetch_forward; RGBD_DV_SPLIT_BACKWARD=0) gen_rest is cut once more, at the 2-D renderer's input
(Generator.__call__'s `cut`): the renderer's weight gradients -- 45 % of G's -- leave the critical path for the side stream while the
backward pass goes on through the occlusion network, the resampling and the voxel generator (1260 -> 1300 img/s):

    main:  ... rest_a: frozen D, loss, backward down to the renderer's input -> rest_b: backward below it + voxel weight gradients -> (wait) rest_c: updates, images
    side:  ... dis_real ------------------------------------------------------> (wait rest_a) wgrad_a: the renderer's weight gradients

(`concurrent_phases=False` / RGBD_CONCURRENT_PHASES=0: the same four phases back to back on one stream -- identical
arithmetic, same accumulation order.)  ~600 mostly small launches per step (960 in round 3): the second stream fills the gaps between the
first one's kernels.  Everything data dependent (the number of frustum samples inside the grid) stays on the device.
"""
import os
import numpy as np
import torch
import torch.nn.functional as F

from . import functional as Fn
from . import kernels
from .common.loss_functions import LossFuncRotate
from .updater import RGBDUpdater, get_camera_matries, shared_stream

IMG_SIZE = 64
FIXED_STAGE = 8.5


def downsize_real(x, size):
    """Average-pool the real batch down to `size` pixels (updater_deepvoxels.py:23-26)."""
    factor = x.shape[-1] // size
    return x if factor <= 1 else F.avg_pool2d(x, factor, factor)


def pose_code(thetas):
    """(B,6) angles / translations -> (B,9) [cos, sin, t] (updater_deepvoxels.py:160-161)."""
    return np.concatenate([np.cos(thetas[:, :3]), np.sin(thetas[:, :3]), thetas[:, 3:]], axis=1).astype("float32")


class DeepVoxelsUpdater(RGBDUpdater):
    def __init__(self, models, config, **kwargs):
        models = list(models) + [None] * (4 - len(models))
        self.gen, self.dis, self.smoothed_gen, self.smoothed_map = models
        if self.smoothed_gen is not None:
            raise AssertionError("keep_smoothed_gen is not supported for the deepvoxels generator")
        self.config = config
        self.smoothing, self.lambda_gp = kwargs.pop("smoothing"), kwargs.pop("lambda_gp")
        self.total_gpu, self.prior = kwargs.pop("total_gpu"), kwargs.pop("prior")
        self._optimizers = kwargs.pop("optimizer")
        self._iterators = {"main": kwargs.pop("iterator")}
        self.nan_check_interval = int(kwargs.pop("nan_check_interval", 100))
        self.nan_watch = bool(kwargs.pop("nan_watch", self.nan_check_interval > 0))     # the per-step watch of RGBDUpdater
        self._nan_state = None
        self.loss_func_rotate = LossFuncRotate(torch, K=self.gen.projection.projection_intrinsic,
                                               lambda_geometric=config.lambda_geometric or 3)
        self.stage_interval = [int(v) for v in str(config.stage_interval).split(",")]
        self.camera_param_range = np.array([config.x_rotate, config.y_rotate, config.z_rotate,
                                            config.x_translate, config.y_translate, config.z_translate])
        self.iteration, self.observation = 0, {}
        self.fixed_stage = None
        self.device = self.gen.device
        comm = getattr(self._optimizers["gen"], "comm", None)
        self.use_graphs = bool(kwargs.pop("use_graphs", True)) and not (comm is not None and comm.active)
        self.graph_warmup = int(kwargs.pop("graph_warmup", 2))
        self.graph_fallback = bool(kwargs.pop("graph_fallback", False))
        env = os.environ.get("RGBD_CONCURRENT_PHASES")
        self.concurrent_phases = bool(kwargs.pop("concurrent_phases", env is None or env not in ("", "0")))
        self.side_cu_budget = int(kwargs.pop("side_cu_budget", os.environ.get("RGBD_SIDE_CUS", "192")))
        self.side_wgrad_workgroups = int(kwargs.pop("side_wgrad_workgroups", os.environ.get("RGBD_SIDE_WGRAD_WGS", "128")))
        env = os.environ.get("RGBD_DV_PREFETCH")
        self.prefetch_forward = bool(kwargs.pop("prefetch_forward", env is None or env not in ("", "0"))) \
            and not (comm is not None and comm.active)
        # ... and the early forward pass, which has 3 ms of dis_fake to hide 2 ms under, fewer still (16 / 32 / 48 / 64 / 96 / 128 /
        # 192 / all: 873 / 1130 / 1209 / 1239 / 1253 / 1233 / 1226 / 1224 img/s, profiles/r06/c4_prefetch.txt)
        self.forward_cu_budget = int(kwargs.pop("forward_cu_budget", os.environ.get("RGBD_DV_FWD_CUS", "96")))
        # split_backward (with prefetch_forward): the renderer's weight gradients on the side stream while the backward pass goes on
        # below the renderer (Generator.__call__'s cut)
        env = os.environ.get("RGBD_DV_SPLIT_BACKWARD")
        self.split_backward = bool(kwargs.pop("split_backward", env is None or env not in ("", "0"))) and self.prefetch_forward
        # (workgroups of that launch: 64 / 128 / 192 / all: 1307 / 1300 / 1296 / 1297 img/s -- it has until the end of the backward
        # pass below the cut, and every CU it takes is one the critical path does not get)
        self.renderer_wgrad_workgroups = int(kwargs.pop("renderer_wgrad_workgroups", os.environ.get("RGBD_DV_WGRAD_A_WGS", "64")))
        self._side_stream = None
        self._graphs, self._eager_calls, self._stagers, self._ones = {}, {}, {}, {}
        self._pf = None                 # the next step's generator forward in flight (see _start_forward)
        self._events = {}

    def get_stage(self):
        return FIXED_STAGE

    def get_z_fake_data(self, batch_size):
        return self.gen.mapping.make_hidden(batch_size)

    # ---- the phases of a step (device work only: capturable)
    def _dv_prep_phase(self, st):
        """What both streams depend on: cleared gradient buffers, the down-sized real batch, D's bf16 weight images (packed
        here so that neither stream does it behind the other's back; G's are rebuilt inside its own phases -- it is
        updated in the middle of the step)."""
        bufs = []
        for link in (self.gen, self.gen.mapping, self.dis):
            for _, store in link.stores:
                store.zero_grad(defer=bufs)
        kernels.zero_multi(bufs)                                   # one launch for all flat gradient buffers
        if st.get("real_idx") is not None:
            st["x_real"] = kernels.real_batch(st["real_data"], st["real_idx"], IMG_SIZE)
        else:
            with torch.no_grad():
                st["x_real"] = downsize_real(st["x_real_full"], IMG_SIZE).contiguous()
        group = getattr(self.dis, "pack_group", None)
        if group is not None:
            group.layers[0].packed()

    def _one(self, like):
        """The seed of a scalar loss's backward pass (a cached buffer instead of one ones_like launch per pass)."""
        key = ("one", tuple(like.shape))
        if key not in self._ones:
            self._ones[key] = torch.ones_like(like.detach())
        return self._ones[key]

    def _dv_dis_real_phase(self, st):
        """The half of the discriminator step that does not involve G (:229-246): adversarial term on the reals + R1."""
        obs = self.observation
        x_real = st["x_real"].detach().requires_grad_(True)
        y_real = self.dis(x_real, stage=FIXED_STAGE)
        adv = Fn.softplus_mean(y_real, -1.0)                   # loss_func_dcgan_dis' second term (value + derivative: 1 launch)
        st["adv_real"] = adv.detach()
        total = adv
        if not self.dis.sn and self.lambda_gp > 0:
            ones = self._ones.get(tuple(y_real.shape))                 # chainer.grad seeds ones (a cached buffer: no sum / expand)
            if ones is None:
                ones = self._ones[tuple(y_real.shape)] = torch.ones_like(y_real)
            with Fn.input_grads_only():
                g, = torch.autograd.grad([y_real], [x_real], [ones], create_graph=True)
            gp = Fn.r1_penalty(g, self.lambda_gp)              # lambda_gp * loss_l2(sqrt(sum g^2), 0) (:239-241), fused
            obs["dis/loss_gp"] = gp.detach()
            total = adv + gp
        wgrads = []
        with Fn.deferred_wgrads(wgrads):
            torch.autograd.backward([total], [self._one(total)])
        with kernels.wgrad_workgroups(st.get("side_wgrad_wgs", 0)):
            Fn.run_deferred_wgrads(wgrads)

    def _dv_gen_forward(self, st):
        half = st["B"] // 2
        if st["z"] is not None:
            z, z2 = st["z"][0], st["z"][1]
        else:                                                      # one latent per view PAIR (:146-148); both from one draw
            zz = self.get_z_fake_data(2 * half)
            z, z2 = zz[:half].repeat(2, 1, 1, 1, 1), zz[half:].repeat(2, 1, 1, 1, 1)
        return self.gen(z, FIXED_STAGE, st["cams"], z2=z2, theta=st["theta9"], cut=st.get("cut"))

    def _dv_gen_phase(self, st):
        self._dv_gen_rest(st, self._dv_gen_forward(st))

    # ---- ... and the generator phase cut behind its forward pass (prefetch_forward)
    def _dv_gen_fwd_phase(self, sf):
        sf["fwd_x_fake"] = self._dv_gen_forward(sf)

    def _dv_gen_rest_phase(self, st):
        # G's weight images on the device are current: the previous generator phase rebuilt them behind G's update (below), and
        # a forward pass of these weights has run since.  Python's epochs do not know it after a replayed step.
        self.gen.mark_weight_images_current()
        self._dv_gen_rest(st, st["fwd"]["fwd_x_fake"], fwd_stream=st["fwd"].get("stream"))
        # both readers of the updated generator -- this step's dis_fake and the next step's forward, on two streams -- find the
        # images built (persistent buffers: functional.DerivedPackGroup)
        self.gen.rebuild_weight_images()

    # ---- ... and, with split_backward, the rest in three: down to the renderer's input (its weight gradients then run on the
    #      side stream, _dv_gen_wgrad_a_phase), below it (+ the voxel generator's weight gradients), the update
    def _dv_gen_rest_a_phase(self, st):
        self.gen.mark_weight_images_current()
        st["wgrads_a"] = []
        self._dv_gen_rest(st, st["fwd"]["fwd_x_fake"], fwd_stream=st["fwd"].get("stream"), collect=st["wgrads_a"])

    def _dv_gen_wgrad_a_phase(self, st):
        with kernels.wgrad_workgroups(self.renderer_wgrad_workgroups):
            Fn.run_deferred_wgrads(st["wgrads_a"])      # (the operands stay referenced until the main stream has joined: rest_c)

    def _dv_gen_rest_b_phase(self, st):
        cut, fwd_stream = st["fwd"]["cut"], st["fwd"].get("stream")
        grads = [t.grad for t in cut["leaves"]]
        pairs = [(b, g) for b, g in zip(cut["below"], grads) if g is not None]
        for t in cut["leaves"]:
            t.grad = None
        wgrads = []
        cross = fwd_stream is not None and not torch.cuda.is_current_stream_capturing()
        if cross:
            fwd_stream.wait_stream(torch.cuda.current_stream())
        with Fn.deferred_wgrads(wgrads):
            torch.autograd.backward([b for b, _ in pairs], [g for _, g in pairs])
        if cross:
            torch.cuda.current_stream().wait_stream(fwd_stream)
        Fn.run_deferred_wgrads(wgrads)

    def _dv_gen_rest_c_phase(self, st):
        st["wgrads_a"] = None
        for name in ("map", "gen"):
            self.get_optimizer(name).update()
        self.gen.rebuild_weight_images()

    def _dv_gen_rest(self, st, x_fake, fwd_stream=None, collect=None):
        """collect (a list): the backward pass stops at the generator's cut; its weight-gradient launches are handed over in
        `collect` instead of being issued, and the optimizers are left to the caller."""
        cfg, obs = self.config, self.observation
        half = st["B"] // 2
        # The gradient w.r.t. G's output is assembled by hand, as in RGBDUpdater._gen_backward (slicing, hinge, scaling and
        # the loss arithmetic were ~60 torch launches of a few microseconds each): the focal adversarial term and its
        # derivative from one launch (loss_func_dcgan_gen, :170), back through the frozen D to the RGB planes,
        # rgbd_image_grad_init puts that on planes 0-2 and zeros on the depth plane, the warp-loss backward ADDS
        # weight * d(loss_rotate + lambda_depth * hinge)/dx_fake (:193-203) with the hinge evaluated in the same kernels.
        x3 = x_fake.detach()[:, :3].contiguous().requires_grad_(True)
        with self.dis.frozen():                                    # no D weight gradients in the generator step
            y_fake = self.dis(x3, stage=FIXED_STAGE)
            adv, gy = kernels.softplus_mean(y_fake.detach(), -1.0, cfg.focal_loss_gamma or 0.0)
            gx, = torch.autograd.grad([y_fake], [x3], [gy.reshape(y_fake.shape)])
        obs["gen/loss_adv"] = adv.reshape(())
        gout = kernels.image_grad_init(gx.contiguous(), None, x_fake.shape[1])
        if st["use_rotate"]:
            if cfg.background_generator:
                raise AssertionError("background_generator is not supported")
            lf = self.loss_func_rotate
            hinge, hmin = float(cfg.lambda_depth or 0.0), float(cfg.depth_min or 0.0)
            weight = cfg.lambda_loss_rotate if cfg.lambda_loss_rotatec else 0.3          # sic (:202)
            xf = x_fake.detach()
            rot = kernels.warp_loss_fwd(xf[:half], xf[half:], st["coef"], 0, lf.lambda_geometric, hinge_lambda=hinge,
                                        hinge_min=hmin)
            obs["gen/loss_rotate"] = rot.reshape(())
            kernels.warp_loss_bwd(xf[:half], xf[half:], st["coef"], 0, lf.lambda_geometric, 0.0, 0.0, None,
                                  hinge_lambda=hinge, hinge_min=hmin, grad_scale=float(weight),
                                  out=(gout[:half], gout[half:]))
        # weight gradients are leaves of the backward pass: collected while it runs, issued as one batch afterwards (the
        # folded 3-D / stride-2 layers go through temporaries and the folds' adjoints into their masters' gradients)
        wgrads = []
        cross = fwd_stream is not None and not torch.cuda.is_current_stream_capturing()
        if cross:       # eager steps only (a capture is one stream): autograd runs G's backward nodes on the forward pass's stream
            fwd_stream.wait_stream(torch.cuda.current_stream())
        with Fn.deferred_wgrads(wgrads):
            torch.autograd.backward([x_fake], [gout])
        if cross:
            torch.cuda.current_stream().wait_stream(fwd_stream)
        if collect is not None:
            collect.extend(wgrads)
            return
        Fn.run_deferred_wgrads(wgrads)
        for name in ("map", "gen"):
            self.get_optimizer(name).update()

    def _dv_dis_fake_phase(self, st):
        """... and the half that does: fresh fakes from the UPDATED generator (:221-228), their term of the loss, D's update
        (its gradient buffer already holds the reals' half)."""
        obs, B = self.observation, st["B"]
        if st["z"] is not None:
            z, z2 = st["z"][2], st["z"][3]
        else:
            zz = self.get_z_fake_data(2 * B)
            z, z2 = zz[:B], zz[B:]
        with torch.no_grad():
            x_fake = self.gen(z, FIXED_STAGE, st["cams"], z2=z2, theta=st["theta9"])
        y_fake = self.dis(x_fake[:, :3].contiguous(), stage=FIXED_STAGE)
        adv = Fn.softplus_mean(y_fake, 1.0)                            # loss_func_dcgan_dis' first term
        obs["dis/loss_adv"] = adv.detach() + st["adv_real"]
        wgrads = []
        with Fn.deferred_wgrads(wgrads):
            torch.autograd.backward([adv], [self._one(adv)])
        Fn.run_deferred_wgrads(wgrads)
        self.get_optimizer("dis").update()

    # ---- the next step's generator forward (prefetch_forward)
    def _event(self, name):
        if name not in self._events:
            self._events[name] = torch.cuda.Event()
        return self._events[name]

    def _forward_record(self, B, key, thetas, cams):
        """What a forward pass of the generator phase needs and leaves: its own copies of the pose uploads (the step's other
        phases read theirs while the next step's are already on their way), later x_fake."""
        sf = {"B": B, "z": None, "cams": self._stager("fwd_cams", (B, 4, 4)).upload(cams),
              "theta9": self._stager("fwd_theta9", (B, 9)).upload(pose_code(thetas)),
              "cut": {} if self.split_backward else None}
        return {"B": B, "key": key, "thetas": thetas, "st": sf, "external_epoch": Fn.external_epoch(), "done": None}

    def _start_forward(self, B, key, main, side):
        """Draw the NEXT step's poses (the prior's stream is consumed in step order, one step early), and run that step's
        generator forward on the side stream behind this step's generator update.  The pair (forward of step n+1, rest of
        step n+1) is eager / captured / replayed TOGETHER: both phases are called once per step under the same graph key, the
        forward one step ahead, so their call counts -- which decide (RGBDUpdater._run_phase_inner) -- stay equal."""
        if key is not None:
            use_rotate = self.iteration + 1 > self.config.start_rotation
            key = (key[0], use_rotate) + tuple(key[2:])
        thetas = np.asarray(self.prior.sample(B), dtype="float32")
        rec = self._forward_record(B, key, thetas, get_camera_matries(thetas))      # (uploads: on the main stream, before the event)
        updated = self._event("gen_updated")
        updated.record(main)
        side.wait_event(updated)
        rec["st"]["stream"] = side
        # (sized for forward_cu_budget compute units: the discriminator phase it runs under is the critical path)
        self._run_phase("dv_gen_fwd", self._dv_gen_fwd_phase, rec["st"], key, stream=side, cu_budget=self.forward_cu_budget)
        rec["done"] = torch.cuda.Event()
        rec["done"].record(side)
        return rec

    def _drop_forward(self, pf):
        """A forward pass in flight that is not the coming step's (another batch size or graph key, injected inputs, weights
        loaded from outside): forget it, and let its graph key start over -- the dropped call has put the forward phase's call
        count one ahead of the rest phase's."""
        if pf["key"] is not None:
            for name in ("dv_gen_fwd", "dv_gen_rest", "dv_gen_rest_a", "dv_gen_wgrad_a", "dv_gen_rest_b"):
                self._graphs.pop(pf["key"] + (name,), None)
                self._eager_calls.pop(pf["key"] + (name,), None)
        return None

    def update_core(self, batch=None, z_fake=None, thetas=None):
        """z_fake: optional (z, z2, z_dis, z2_dis) injected by tests; otherwise drawn as the reference draws them."""
        real_idx = real_data = None
        it = self.get_iterator("main") if batch is None else None
        if it is not None and hasattr(it, "next_indices") and it.data.shape[1] == 3 and it.data.shape[2] == it.data.shape[3] \
                and it.data.shape[2] % IMG_SIZE == 0:
            # data set resident in HBM (as in RGBDUpdater.update_core): only the batch's indices move; gather + x / 127.5 - 1 +
            # the block means of downsize_real are ONE kernel of the prep phase (rgbd_real_batch_u8) instead of six torch launches
            idx = it.next_indices()
            B = int(idx.numel())
            key_i = ("real_idx", B)
            if key_i not in self._stagers:
                self._stagers[key_i] = torch.empty(B, dtype=torch.int64, device=self.device)
            self._stagers[key_i].copy_(idx)
            real_idx, real_data = self._stagers[key_i], it.data
            x_real_full = None
        else:
            if batch is None:
                batch = it.next()
            B = len(batch)
            x_real_full = self.get_x_real_data(batch, B)
        half = B // 2
        # the generator forward a previous step started for this one (if any, and if it is still this step's: same batch size,
        # nothing injected, weights untouched from outside since)
        prefetch = self.prefetch_forward and self.concurrent_phases and z_fake is None and thetas is None
        pf, self._pf = self._pf, None
        if pf is not None and not (prefetch and pf["B"] == B and pf["external_epoch"] == Fn.external_epoch()):
            pf = self._drop_forward(pf)
        # host side, NumPy, as the reference: pose prior, camera matrices, pose code, warp constants
        if pf is not None:
            thetas = pf["thetas"]
        thetas = np.asarray(self.prior.sample(B) if thetas is None else thetas, dtype="float32")
        cams = get_camera_matries(thetas)
        use_rotate = self.iteration > self.config.start_rotation
        st = {"B": B, "use_rotate": use_rotate, "z": None, "real_idx": real_idx, "real_data": real_data}
        st["cams"] = self._stager("cams", (B, 4, 4)).upload(cams)
        st["theta9"] = self._stager("theta9", (B, 9)).upload(pose_code(thetas))
        if use_rotate:
            coef = self.loss_func_rotate.coefficients_for_size(IMG_SIZE, cams[:half], cams[half:])
            st["coef"] = self._stager("coef", (half, 24)).upload(coef)
        if z_fake is not None:
            zs = []
            for i, z in enumerate(z_fake):
                z = torch.as_tensor(z).to(self.device, torch.float32)
                zkey = ("z", i) + tuple(z.shape)
                if zkey not in self._stagers:
                    self._stagers[zkey] = torch.empty_like(z)
                self._stagers[zkey].copy_(z)
                zs.append(self._stagers[zkey])
            st["z"] = zs
        key = None
        if self.use_graphs:
            if x_real_full is not None:
                skey = ("x_real_full",) + tuple(x_real_full.shape)
                if skey not in self._stagers:
                    self._stagers[skey] = torch.empty_like(x_real_full)
                self._stagers[skey].copy_(x_real_full)
                x_real_full = self._stagers[skey]
            shape = tuple(x_real_full.shape) if x_real_full is not None else (B,) + tuple(real_data.shape[1:])
            key = (B, use_rotate, shape, z_fake is not None, Fn.conv_dtype(), real_idx is not None)
        st["x_real_full"] = x_real_full

        if pf is not None and pf["key"] != key:
            pf = self._drop_forward(pf)
        self._replayed = False
        self._run_phase("dv_prep", self._dv_prep_phase, st, key)
        if self.concurrent_phases:
            if self._side_stream is None:
                self._side_stream = shared_stream(self.device, "side")
            main, side = torch.cuda.current_stream(), self._side_stream
            side.wait_stream(main)
            # the side stream's chip-filling launches leave compute units to the generator phase (RGBDUpdater.side_cu_budget;
            # here 192 / 128 of 256: 977 -> 984-992 img/s, profiles/r05/cu_budget_sweep.txt)
            st["side_wgrad_wgs"] = self.side_wgrad_workgroups
            self._run_phase("dv_dis_real", self._dv_dis_real_phase, st, key, stream=side, cu_budget=self.side_cu_budget)
            if prefetch:
                dis_real_done = self._event("dis_real")
                dis_real_done.record(side)
                if pf is None:                  # nothing in flight (first step, or dropped): this step's forward, here
                    pf = self._forward_record(B, key, thetas, cams)
                    self._run_phase("dv_gen_fwd", self._dv_gen_fwd_phase, pf["st"], key)
                else:
                    main.wait_event(pf["done"])
                st["fwd"] = pf["st"]
                if self.split_backward:
                    self._run_phase("dv_gen_rest_a", self._dv_gen_rest_a_phase, st, key)
                    rest_a_done, wgrad_a_done = self._event("rest_a"), self._event("wgrad_a")
                    rest_a_done.record(main)
                    side.wait_event(rest_a_done)
                    self._run_phase("dv_gen_wgrad_a", self._dv_gen_wgrad_a_phase, st, key, stream=side)
                    wgrad_a_done.record(side)
                    self._run_phase("dv_gen_rest_b", self._dv_gen_rest_b_phase, st, key)
                    main.wait_event(wgrad_a_done)
                    self._run_phase("dv_gen_rest", self._dv_gen_rest_c_phase, st, key)
                else:
                    self._run_phase("dv_gen_rest", self._dv_gen_rest_phase, st, key)
                if key is None or key + ("dv_gen_rest",) not in self._graphs:
                    # an eager pair: let its autograd graph die before the next forward is built.  A live graph keeps its
                    # leaves' gradient accumulators, each tied to the stream it was made on (the side stream here); a forward
                    # CAPTURED while they live inherits them, and the capture of its backward forks onto that stream
                    # (hipStreamEndCapture then fails -- RGBDUpdater._run_phase_inner has the same rule for its phases)
                    pf["st"]["fwd_x_fake"] = pf["st"]["cut"] = st["fwd"] = None
                self._pf = self._start_forward(B, key, main, side)           # fwd(n+1) under dis_fake(n)
                main.wait_event(dis_real_done)
            else:
                self._run_phase("dv_gen", self._dv_gen_phase, st, key)
                main.wait_stream(side)
        else:
            self._run_phase("dv_dis_real", self._dv_dis_real_phase, st, key)
            self._run_phase("dv_gen", self._dv_gen_phase, st, key)
        self._run_phase("dv_dis_fake", self._dv_dis_fake_phase, st, key)
        if self._replayed:
            Fn.bump_weight_epoch(own_step=True)      # replays change the weights behind Python's back: invalidate packed caches

        obs = self.observation
        obs["stage"], obs["batch_size"], obs["image_size"] = FIXED_STAGE, B, IMG_SIZE
        self._end_of_step_checks()
