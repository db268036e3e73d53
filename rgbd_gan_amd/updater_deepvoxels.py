"""DeepVoxelsUpdater with the reference's surface (updater_deepvoxels.py:76-252): one generator step and one
discriminator step of configs/deepvoxels_shapenet_car.yml on the HIP kernels.

Differences from RGBDUpdater that the reference has and this keeps:
  * the stage is the constant 8.5 (:105-106): the progressive discriminator of net.py runs its 64x64 path;
  * two latents per step, each drawn for half the batch and tiled so both views of a pair share them (:146-148);
  * the generator is RE-RUN with fresh latents for the discriminator step, after its own update (:221-228);
  * the 3-D loss uses the projection intrinsics as K (:92-93), no occlusion flag (:193-196), weight 0.3 (the config
    lookup at :202 tests a misspelt key, so the YAML value can never be used), focal gamma from the YAML (:170).
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import functional as Fn
from .common.loss_functions import LossFuncRotate, loss_func_dcgan_dis, loss_func_dcgan_gen, loss_l2
from .updater import get_camera_matries

IMG_SIZE = 64


def downsize_real(x, size):
    """updater_deepvoxels.py:23-26."""
    scale = x.shape[-1] // size
    return F.avg_pool2d(x, scale, scale) if scale > 1 else x


class DeepVoxelsUpdater:
    def __init__(self, models, config, **kwargs):
        if len(models) == 2:
            models = list(models) + [None, None]
        self.gen, self.dis, self.smoothed_gen, self.smoothed_map = models
        if self.smoothed_gen is not None:
            raise AssertionError("keep_smoothed_gen is not supported for the deepvoxels generator")
        self.config = config
        self.smoothing = kwargs.pop("smoothing")
        self.lambda_gp = kwargs.pop("lambda_gp")
        self.total_gpu = kwargs.pop("total_gpu")
        self.prior = kwargs.pop("prior")
        self._optimizers = kwargs.pop("optimizer")
        self._iterators = {"main": kwargs.pop("iterator")}
        lambda_geometric = config.lambda_geometric if config.lambda_geometric else 3
        self.loss_func_rotate = LossFuncRotate(torch, K=self.gen.projection.projection_intrinsic,
                                               lambda_geometric=lambda_geometric)
        self.stage_interval = list(map(int, str(config.stage_interval).split(",")))
        self.camera_param_range = np.array([config.x_rotate, config.y_rotate, config.z_rotate,
                                            config.x_translate, config.y_translate, config.z_translate])
        self.nan_check_interval = int(kwargs.pop("nan_check_interval", 100))
        self.iteration = 0
        self.observation = {}
        self.device = self.gen.device

    def get_optimizer(self, name):
        return self._optimizers[name]

    def get_iterator(self, name):
        return self._iterators[name]

    @property
    def stage(self):
        return self.get_stage()

    def get_stage(self):
        return 8.5

    def get_x_real_data(self, batch, batch_size):
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
        return self.gen.mapping.make_hidden(batch_size)

    def update(self):
        self.update_core()
        self.iteration += 1

    def update_core(self, batch=None, z_fake=None, thetas=None):
        """z_fake: optional (z, z2, z_dis, z2_dis) injected by tests; otherwise drawn as the reference draws them."""
        cfg = self.config
        obs = self.observation
        use_rotate = self.iteration > cfg.start_rotation
        self.gen.cleargrads()
        self.gen.mapping.cleargrads()
        self.dis.cleargrads()
        opt_g_m, opt_g_g, opt_d = (self.get_optimizer(n) for n in ("map", "gen", "dis"))
        stage = self.stage
        if batch is None:
            batch = self.get_iterator("main").next()
        batch_size = len(batch)
        half = batch_size // 2
        x_real_data = self.get_x_real_data(batch, batch_size)
        if z_fake is None:
            z1 = self.get_z_fake_data(half).repeat(2, 1, 1, 1, 1)          # same latent for both views of a pair
            z2 = self.get_z_fake_data(half).repeat(2, 1, 1, 1, 1)
        else:
            z1, z2 = (torch.as_tensor(z).to(self.device, torch.float32) for z in z_fake[:2])
        if thetas is None:
            thetas = self.prior.sample(batch_size)
        thetas = np.asarray(thetas, dtype="float32")
        cams = get_camera_matries(thetas)
        theta9 = np.concatenate([np.cos(thetas[:, :3]), np.sin(thetas[:, :3]), thetas[:, 3:]], axis=1).astype("float32")
        with torch.no_grad():
            x_real = downsize_real(x_real_data, IMG_SIZE).contiguous()
        image_size = x_real.shape[2]

        # ---- generator step
        x_fake = self.gen(z1, stage, cams, z2=z2, theta=theta9)
        with self.dis.frozen():
            y_fake = self.dis(x_fake[:, :3].contiguous(), stage=stage)
        loss_gen = loss_func_dcgan_gen(y_fake, cfg.focal_loss_gamma)
        obs["gen/loss_adv"] = loss_gen.detach()
        if use_rotate:
            if cfg.background_generator:
                raise AssertionError("background_generator is not supported")
            loss_rotate, _ = self.loss_func_rotate(x_fake[:half], cams[:half], x_fake[half:], cams[half:])
            loss_rotate = loss_rotate + torch.mean(F.relu(cfg.depth_min - x_fake[:, -1]) ** 2) * cfg.lambda_depth
            obs["gen/loss_rotate"] = loss_rotate.detach()
            lambda_loss_rotate = cfg.lambda_loss_rotate if cfg.lambda_loss_rotatec else 0.3      # sic (:202)
            loss_gen = loss_gen + loss_rotate * lambda_loss_rotate
        loss_gen.backward()
        opt_g_m.update()
        opt_g_g.update()
        del loss_gen, y_fake, x_fake

        # ---- discriminator step: fresh latents through the UPDATED generator, no graph kept (:221-228)
        self.dis.cleargrads()
        if z_fake is None:
            z1, z2 = self.get_z_fake_data(batch_size), self.get_z_fake_data(batch_size)
        else:
            z1, z2 = (torch.as_tensor(z).to(self.device, torch.float32) for z in z_fake[2:])
        with torch.no_grad():
            x_fake = self.gen(z1, stage, cams, z2=z2, theta=theta9)
        y_fake = self.dis(x_fake[:, :3].contiguous(), stage=stage)
        x_real_v = x_real.detach().requires_grad_(True)
        y_real = self.dis(x_real_v, stage=stage)
        loss_adv = loss_func_dcgan_dis(y_fake, y_real)
        loss_dis = loss_adv
        if not self.dis.sn and self.lambda_gp > 0:
            with Fn.input_grads_only():
                grad_x, = torch.autograd.grad([y_real.sum()], [x_real_v], create_graph=True)
            grad_l2 = torch.sqrt(torch.sum(grad_x ** 2, dim=(1, 2, 3)))
            loss_gp = self.lambda_gp * loss_l2(grad_l2, 0.0)
            obs["dis/loss_gp"] = loss_gp.detach()
            loss_dis = loss_adv + loss_gp
        obs["dis/loss_adv"] = loss_adv.detach()
        loss_dis.backward()
        opt_d.update()
        obs["stage"], obs["batch_size"], obs["image_size"] = stage, batch_size, int(image_size)
        if self.nan_check_interval > 0 and (self.iteration + 1) % self.nan_check_interval == 0:
            for key in ("gen/loss_adv", "gen/loss_rotate", "dis/loss_adv"):
                v = obs.get(key)
                if v is not None and not bool(torch.isfinite(v)):
                    raise AssertionError(f"{key} is not finite at iteration {self.iteration}")
