"""DeepVoxelsUpdater with the reference's surface (updater_deepvoxels.py:76-252): one generator step and one
discriminator step of configs/deepvoxels_shapenet_car.yml on the HIP kernels.

Differences from RGBDUpdater that the reference has and this keeps:
  * the stage is the constant 8.5 (:105-106): the progressive discriminator of net.py runs its 64x64 path;
  * two latents per step, each drawn for half the batch and tiled so both views of a pair share them (:146-148);
  * the generator is RE-RUN with fresh latents for the discriminator step, after its own update (:221-228), so the
    single-pass dataflow of RGBDUpdater (one D(x_fake) evaluation per step) does not apply here;
  * the 3-D loss uses the projection intrinsics as K (:92-93), no occlusion flag (:193-196), weight 0.3 (the config
    lookup at :202 tests a misspelt key, so the YAML value can never be used), focal gamma from the YAML (:170).

The class reuses RGBDUpdater's plumbing (optimizer / iterator accessors, batch conversion, finite checks) and
replaces the step.
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import functional as Fn
from .common.loss_functions import LossFuncRotate, loss_func_dcgan_dis, loss_func_dcgan_gen, loss_l2
from .updater import RGBDUpdater, get_camera_matries

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
        self.loss_func_rotate = LossFuncRotate(torch, K=self.gen.projection.projection_intrinsic,
                                               lambda_geometric=config.lambda_geometric or 3)
        self.stage_interval = [int(v) for v in str(config.stage_interval).split(",")]
        self.camera_param_range = np.array([config.x_rotate, config.y_rotate, config.z_rotate,
                                            config.x_translate, config.y_translate, config.z_translate])
        self.iteration, self.observation = 0, {}
        self.fixed_stage = None
        self.device = self.gen.device

    def get_stage(self):
        return FIXED_STAGE

    def get_z_fake_data(self, batch_size):
        return self.gen.mapping.make_hidden(batch_size)

    # ---- the two halves of a step
    def _generator_step(self, latents, cams, theta9, half):
        cfg, obs = self.config, self.observation
        z, z2 = latents
        x_fake = self.gen(z, FIXED_STAGE, cams, z2=z2, theta=theta9)
        with self.dis.frozen():                                  # no D weight gradients in the generator step
            y_fake = self.dis(x_fake[:, :3].contiguous(), stage=FIXED_STAGE)
        loss = loss_func_dcgan_gen(y_fake, cfg.focal_loss_gamma)
        obs["gen/loss_adv"] = loss.detach()
        if self.iteration > cfg.start_rotation:
            if cfg.background_generator:
                raise AssertionError("background_generator is not supported")
            rot, _ = self.loss_func_rotate(x_fake[:half], cams[:half], x_fake[half:], cams[half:])
            rot = rot + cfg.lambda_depth * torch.mean(F.relu(cfg.depth_min - x_fake[:, -1]) ** 2)
            obs["gen/loss_rotate"] = rot.detach()
            weight = cfg.lambda_loss_rotate if cfg.lambda_loss_rotatec else 0.3          # sic (:202)
            loss = loss + weight * rot
        loss.backward()
        for name in ("map", "gen"):
            self.get_optimizer(name).update()

    def _discriminator_step(self, latents, cams, theta9, x_real):
        obs = self.observation
        self.dis.cleargrads()
        z, z2 = latents
        with torch.no_grad():                                    # fresh fakes from the UPDATED generator (:221-228)
            x_fake = self.gen(z, FIXED_STAGE, cams, z2=z2, theta=theta9)
        y_fake = self.dis(x_fake[:, :3].contiguous(), stage=FIXED_STAGE)
        x_real = x_real.detach().requires_grad_(True)
        y_real = self.dis(x_real, stage=FIXED_STAGE)
        adv = loss_func_dcgan_dis(y_fake, y_real)
        obs["dis/loss_adv"] = adv.detach()
        total = adv
        if not self.dis.sn and self.lambda_gp > 0:
            with Fn.input_grads_only():
                g, = torch.autograd.grad([y_real.sum()], [x_real], create_graph=True)
            gp = self.lambda_gp * loss_l2(torch.sqrt(torch.sum(g ** 2, dim=(1, 2, 3))), 0.0)
            obs["dis/loss_gp"] = gp.detach()
            total = adv + gp
        total.backward()
        self.get_optimizer("dis").update()

    def update_core(self, batch=None, z_fake=None, thetas=None):
        """z_fake: optional (z, z2, z_dis, z2_dis) injected by tests; otherwise drawn as the reference draws them."""
        for link in (self.gen, self.gen.mapping, self.dis):
            link.cleargrads()
        if batch is None:
            batch = self.get_iterator("main").next()
        B = len(batch)
        half = B // 2
        x_real_full = self.get_x_real_data(batch, B)
        if z_fake is None:
            pair = lambda: self.get_z_fake_data(half).repeat(2, 1, 1, 1, 1)          # one latent per view PAIR
            g_lat, d_lat = (pair(), pair()), None
        else:
            dev = [torch.as_tensor(z).to(self.device, torch.float32) for z in z_fake]
            g_lat, d_lat = (dev[0], dev[1]), (dev[2], dev[3])
        thetas = np.asarray(self.prior.sample(B) if thetas is None else thetas, dtype="float32")
        cams, theta9 = get_camera_matries(thetas), pose_code(thetas)
        with torch.no_grad():
            x_real = downsize_real(x_real_full, IMG_SIZE).contiguous()

        self._generator_step(g_lat, cams, theta9, half)
        if d_lat is None:
            d_lat = (self.get_z_fake_data(B), self.get_z_fake_data(B))
        self._discriminator_step(d_lat, cams, theta9, x_real)

        obs = self.observation
        obs["stage"], obs["batch_size"], obs["image_size"] = FIXED_STAGE, B, int(x_real.shape[2])
        if self.nan_check_interval > 0 and (self.iteration + 1) % self.nan_check_interval == 0:
            self._check_finite()
