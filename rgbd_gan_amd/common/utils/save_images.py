"""Preview tiles (reference common/utils/save_images.py:9-24 and train_rgbd.py:39-92).

`convert_batch_images` lays a (rows*cols, 3|4, H, W) batch out as one uint8 image: sample (r, c) sits at row-block r,
column-block c; for RGB-D batches every RGB row-block is followed by a row-block with the depth rendered as
clip(128 / d, 0, 255).  Host-side NumPy: a few hundred KB every `evaluation_sample_interval` iterations.
"""
import os

import numpy as np
import torch


def convert_batch_images(x, rows, cols):
    x = np.asarray(x)
    depth = None
    if x.shape[1] == 4:
        depth = np.tile(x[:, -1:], (1, 3, 1, 1))
        x = x[:, :-1]
    x = np.asarray(np.clip(x * 127.5 + 127.5, 0.0, 255.0), dtype=np.uint8)
    _, _, H, W = x.shape
    x = x.reshape((rows, cols, 3, H, W))
    if depth is not None:
        with np.errstate(divide="ignore"):
            depth = np.asarray(np.clip(1 / depth * 128, 0.0, 255.0), dtype=np.uint8)
        depth = depth.reshape((rows, cols, 3, H, W))
        x = np.concatenate([x, depth], axis=1).reshape(rows * 2, cols, 3, H, W)
    return x.transpose(0, 3, 1, 4, 2).reshape((-1, cols * W, 3))


class PreviewSampler:
    """sample_generate_light (train_rgbd.py:39-92): `cols` latents, each rendered under `rows` yaw angles swept over
    [-test_y_rotate, test_y_rotate], generator in eval mode, written to {dst}/{subdir}/image_latest.png and
    image{iteration // 10000 * 10000:08d}.png.  The latents are drawn once (first call) and reused, as in the
    reference's closure."""

    def __init__(self, gen, dst, config, rows=8, cols=8, z=None, seed=0, subdir="preview"):
        self.gen, self.dst, self.config = gen, dst, config
        self.rows, self.cols, self.z, self.seed, self.subdir = rows, cols, z, seed, subdir

    def _tile(self, z):
        z = z[:, None].expand(z.shape[0], self.rows, *z.shape[1:])
        return z.reshape(self.rows * self.cols, *z.shape[2:])

    @torch.no_grad()
    def render(self, stage):
        from ...updater import get_camera_matries
        cfg, gen, rows, cols = self.config, self.gen, self.rows, self.cols
        if self.z is None:
            n = rows * cols if cfg.rgb else cols
            if hasattr(gen, "_latent_rng") and torch.device(gen.device).type == "cuda":
                # StyleGAN latents come from the library's Philox stream: a PRIVATE stream seeded with the sampler's seed --
                # the same latents in every run and after a resume, and the training stream's launch counter is not touched
                from ... import kernels
                z = gen.make_hidden(n, rng_state=kernels.new_hidden_rng_state(gen.device, seed=self.seed))
            else:
                dev = torch.device(gen.device)
                state = torch.cuda.get_rng_state(dev) if dev.type == "cuda" else torch.get_rng_state()
                (torch.cuda.manual_seed if dev.type == "cuda" else torch.manual_seed)(self.seed)    # torch.randn generators
                try:
                    z = gen.make_hidden(n)
                finally:
                    torch.cuda.set_rng_state(state, dev) if dev.type == "cuda" else torch.set_rng_state(state)
            self.z = z if cfg.rgb else self._tile(z)
        z = self.z[:rows * cols]
        theta = cams = None
        if not cfg.rgb:
            th = np.zeros((rows * cols, 6))
            th[:, 1] = np.tile(np.linspace(-cfg.test_y_rotate, cfg.test_y_rotate, rows), cols)
            th = th.astype("float32")
            cams = get_camera_matries(th)
            theta = np.concatenate([np.cos(th[:, :3]), np.sin(th[:, :3]), th[:, 3:]], axis=1).astype("float32")
        was_training = getattr(gen, "train", True)
        gen.train = False
        try:
            if cfg.generator_architecture == "deepvoxels":
                z2 = self._tile(gen.make_hidden(cols))
                x = gen(z, stage, cams, z2=z2, theta=theta)
            else:
                x = gen(z, stage=stage, theta=theta)
        finally:
            gen.train = was_training
        return convert_batch_images(x.float().cpu().numpy(), rows, cols)

    def __call__(self, stage, iteration):
        from PIL import Image
        img = self.render(stage)
        d = f"{self.dst}/{self.subdir}"
        os.makedirs(d, exist_ok=True)
        Image.fromarray(img).save(d + "/image_latest.png")
        Image.fromarray(img).save(d + "/image{:0>8}.png".format(iteration // 10000 * 10000))
        return img
