"""Generator / discriminator with the reference's public surface (net.py of nogu-atsu/RGBD-GAN), built on
the HIP conv engine.

Drop-in surface kept:
    StyleGANGenerator(ch, enable_blur, rgbd, rotate_conv_input, use_encoder, use_occupancy_net, initial_depth)
        .mapping, .gen, .make_hidden(n), __call__(z, stage, theta=None, return_feature=False)    net.py:314-354
    DCGANGenerator(in_ch, ch, ...)   same call                                                    net.py:651-773
    Discriminator(ch, out_dim, enable_blur, sn, res)  __call__(x, stage, return_hidden=False), .sn  net.py:429-504
Inputs / outputs at this surface are NCHW fp32 device tensors like the reference's arrays; inside, activations
are NHWC bf16 and every 3x3 convolution runs in rgbd_gan_amd/csrc/conv.hip.  Parameters carry the reference's
Chainer names (``namedparams``) so its snapshots load unchanged.

enable_blur (rescale.py:20-25) is supported through rgbd_blur3x3_bf16.  Not supported (unreachable with the shipped
configs, asserted): sn, use_encoder (bigan), use_occupancy_net, rotate_conv_input.
"""
import contextlib
import math
import os

import numpy as np
import torch
import torch.nn.functional as F

from . import functional as Fn
from .params import ParamStore, depth_row_init

SQRT2 = float(np.sqrt(2))
BF16 = torch.bfloat16


def _inv_c(fan_in, gain=SQRT2):
    return float(gain * np.sqrt(1.0 / fan_in))


_ALPHA_OVERRIDE = None      # 0-dim device tensor holding the fade-in blend factor, or None


@contextlib.contextmanager
def alpha_override(alpha):
    """Inside, the fade-in blend factor of the progressive stages (net.py:283-290,490-497 of the reference) is read
    from this device scalar instead of being computed from the Python float `stage`: the launch sequence of a fade-in
    step then depends only on floor(stage), so it can be captured once and replayed while alpha moves every
    iteration."""
    global _ALPHA_OVERRIDE
    old = _ALPHA_OVERRIDE
    _ALPHA_OVERRIDE = alpha
    try:
        yield
    finally:
        _ALPHA_OVERRIDE = old


def _block_count(max_resolution):
    """Blocks of a progressive network that ends at max_resolution x max_resolution: 6 for the reference's 128 (net.py:175-
    180); its commented-out 256 / 512 / 1024 blocks (net.py:181-183,192-194: ch//8, ch//16, ch//32 channels) are blocks 6-8."""
    nb = int(max_resolution).bit_length() - 2
    if max_resolution < 128 or (1 << (nb + 1)) != max_resolution:
        raise ValueError(f"max_resolution must be a power of two >= 128 (got {max_resolution})")
    return nb


def _synthesis_chans(ch, nb):
    """(out, in) channels of generator blocks 0..nb-1: ch up to 32 px, then halving (net.py:175-183)."""
    return [(ch, ch)] * 4 + [(ch >> (i - 3), ch >> (i - 4)) for i in range(4, nb)]


def _split_stage(stage, max_stage):
    stage = min(stage, max_stage - 1e-8)
    fl = math.floor(stage)
    return fl, (stage - fl if _ALPHA_OVERRIDE is None else _ALPHA_OVERRIDE)


def upsample_planes(x, scale=2):
    """Nearest-neighbour upsampling of NCHW planes (F.unpooling_2d(k, k) of rescale.py:4-5) as an expand + reshape:
    one copy kernel, no index tensors (repeat_interleave builds its index on the host, which stalls graph replays)."""
    B, C, H, W = x.shape
    return x.reshape(B, C, H, 1, W, 1).expand(B, C, H, scale, W, scale).reshape(B, C, H * scale, W * scale)


def _as_device_tensor(a, device, dtype=torch.float32):
    if torch.is_tensor(a):
        return a.to(device=device, dtype=dtype)
    return torch.as_tensor(np.asarray(a), dtype=dtype).to(device)


class _Link:
    """Minimal stand-in for chainer.Link: named parameters over one or more ParamStores."""

    stores = ()

    def namedparams(self):
        for prefix, store in self.stores:
            for name in store.names:
                yield "/" + prefix + name, store.params[name]

    def params(self):
        for _, p in self.namedparams():
            yield p

    def cleargrads(self):
        for _, store in self.stores:
            store.zero_grad()

    def state_dict(self):
        out = {}
        for prefix, store in self.stores:
            for k, v in store.state_dict().items():
                out[prefix + k] = v
        return out

    def load_state_dict(self, arrays, strict=True):
        for prefix, store in self.stores:
            sub = {k[len(prefix):]: v for k, v in arrays.items() if k.startswith(prefix)}
            store.load(sub, strict=strict)

    def to_gpu(self, device=None):
        return self

    @contextlib.contextmanager
    def frozen(self):
        """Run a forward whose backward must not produce weight gradients (D inside the generator step)."""
        ps = list(self.params())
        for p in ps:
            p.requires_grad_(False)
        try:
            yield self
        finally:
            for p in ps:
                p.requires_grad_(True)


# ---------------------------------------------------------------------------------------------- StyleGAN
class MappingNetwork(_Link):
    """net.py:22-62: pixel-norm then 8 x (equalized linear, leaky ReLU).  Tiny fp32 GEMMs (M = 2B rows, N = K = ch) in a
    dependent chain: one fused launch per pass (rgbd_mlp_fwd / rgbd_mlp_bwd; per-layer rgbd_linear_* for other widths)."""

    def __init__(self, ch, device, seed=0):
        self.ch = ch
        specs = []
        for i in range(0, 16, 2):
            specs += [(f"l/{i}/c/W", (ch, ch), "normal"), (f"l/{i}/c/b", (ch,), "zeros")]
        self.store = ParamStore(specs, device, seed)
        self.stores = (("", self.store),)
        self.inv_c = _inv_c(ch)

    def __call__(self, z):
        h = Fn.pixel_norm(z.reshape(z.shape[0], -1))
        p = self.store.params
        # the eight layers as one launch per pass (rgbd_mlp_fwd / rgbd_mlp_bwd) instead of 8 forward + 16 backward ones
        return Fn.mlp_chain(h, [p[f"l/{i}/c/W"] for i in range(0, 16, 2)], [p[f"l/{i}/c/b"] for i in range(0, 16, 2)],
                            self.inv_c)

    forward = __call__


class StyleGenerator(_Link):
    """net.py:164-311.  Channel plan at ch: blocks (ch,ch,ch,ch,ch/2,ch/4) at 4..128 px; max_resolution 256 / 512 adds
    the blocks the reference keeps commented out (ch/8 at 256 px, ch/16 at 512 px: net.py:181-183,192-194)."""

    def __init__(self, ch, device, rgbd=True, initial_depth=1.0, seed=1, enable_blur=False, max_resolution=128):
        nb = _block_count(max_resolution)
        self.ch, self.rgbd, self.max_stage = ch, rgbd, 2 * nb + 5     # 17 for the reference's six blocks (net.py:166)
        self.enable_blur = bool(enable_blur)       # net.py:140-141: c0(blur(upscale2x(h))) instead of c0(upscale2x(h))
        self.chans = _synthesis_chans(ch, nb)      # (out, in)
        assert self.chans[-1][0] % 64 == 0, "the MFMA conv engine needs the last block's channels to be a multiple of 64"
        out_ch = 4 if rgbd else 3
        w_init, b_init = depth_row_init(initial_depth, out_ch, rgbd)
        specs = []
        # all style affines first, [scale | shift] rows of block 0 s0, block 0 s1, block 1 s0, ... back to back (and
        # their biases likewise): the affines of any run of consecutive blocks are then ONE matrix (ParamStore.fused)
        # and one launch (_style_group)
        for i, (co, ci) in enumerate(self.chans):
            for s in ("s0", "s1"):
                specs += [(f"blocks/{i}/{s}/s/c/W", (co, ch), "normal"), (f"blocks/{i}/{s}/b/c/W", (co, ch), "normal")]
        for i, (co, ci) in enumerate(self.chans):
            for s in ("s0", "s1"):
                specs += [(f"blocks/{i}/{s}/s/c/b", (co,), "ones"), (f"blocks/{i}/{s}/b/c/b", (co,), "zeros")]
        for i, (co, ci) in enumerate(self.chans):
            pre = f"blocks/{i}"
            if i == 0:
                specs.append((pre + "/W", (ci, 4, 4), "ones"))
            specs += [(pre + "/b0/b", (co,), "zeros"), (pre + "/b1/b", (co,), "zeros"),
                      (pre + "/n0/b/W", (co,), "zeros"), (pre + "/n1/b/W", (co,), "zeros")]
            specs += [(pre + "/c0/c/W", (co, ci, 3, 3), "normal"), (pre + "/c1/c/W", (co, co, 3, 3), "normal")]
        for i, (co, _) in enumerate(self.chans):
            specs += [(f"outs/{i}/c/W", (out_ch, co, 1, 1), w_init), (f"outs/{i}/c/b", (out_ch,), b_init)]
        if rgbd:
            specs += [("l1/c/W", (ch, ch + 9), "normal"), ("l1/c/b", (ch,), "zeros"),
                      ("l2/c/W", (ch, ch), "normal"), ("l2/c/b", (ch,), "zeros")]
        self.store = ParamStore(specs, device, seed)
        self.stores = (("", self.store),)
        p = self.store.params
        self.c0 = [None] + [Fn.ConvLayer(p[f"blocks/{i}/c0/c/W"], _inv_c(self.chans[i][1] * 9), 1)
                            for i in range(1, nb)]
        self.c1 = [Fn.ConvLayer(p[f"blocks/{i}/c1/c/W"], _inv_c(self.chans[i][0] * 9), 1) for i in range(nb)]
        self.pack_group = Fn.PackGroup(self.c0 + self.c1)

    # -- pieces
    def _style(self, name, w, h):
        """net.py:90-102 (StyleBlock): AdaIN(h, s(w), b(w)); both linears gain 1."""
        co = self.store.shapes[name + "/s/c/W"][0]
        W = self.store.fused((name + "/s/c/W", name + "/b/c/W"), (2 * co, self.ch))
        b = self.store.fused((name + "/s/c/b", name + "/b/c/b"), (2 * co,))
        ss = Fn.linear_act(w, W, b, _inv_c(self.ch, 1.0), act=False)           # [scale | shift] in one launch
        return Fn.adain_fused(h, ss)

    def _style_group(self, w, i0, i1):
        """StyleBlock affines (net.py:96-101) of blocks [i0, i1), all fed by the latent `w`, as one linear."""
        names_w, names_b, offsets, tot = [], [], {}, 0
        for i in range(i0, i1):
            co = self.chans[i][0]
            for s in ("s0", "s1"):
                names_w += [f"blocks/{i}/{s}/s/c/W", f"blocks/{i}/{s}/b/c/W"]
                names_b += [f"blocks/{i}/{s}/s/c/b", f"blocks/{i}/{s}/b/c/b"]
                offsets[(i, s)] = tot
                tot += 2 * co
        W = self.store.fused(tuple(names_w), (tot, self.ch))
        b = self.store.fused(tuple(names_b), (tot,))
        ss = Fn.linear_act(w, W, b, _inv_c(self.ch, 1.0), act=False)
        keys = list(offsets)
        group = Fn.StyleGroup(ss, [offsets[k] for k in keys])
        return {k: (group, j) for j, k in enumerate(keys)}

    def _block(self, i, w, x, styles=None):
        """net.py:130-161 (SynthesisBlock.forward), add_noise False (forced at net.py:243).  `styles`: the windows of
        a _style_group covering this block (else its two affines are computed here from `w`)."""
        p = self.store.params
        pre = f"blocks/{i}"
        if styles is not None:
            style = lambda name, w_, h_: Fn.adain_window(h_, *styles[(i, name[-2:])])
        else:
            style = self._style
        if i == 0:
            h = Fn.const_input(p[pre + "/W"], p[pre + "/b0/b"], w.shape[0])      # lrelu(W + b0), (B,4,4,ch) bf16
        elif styles is not None:      # conv -> bias -> lrelu -> style as one node (fused backward)
            if self.enable_blur:      # the blurred upsampled tensor is materialised (one HIP pass), the conv reads it plain
                h = Fn.conv_bias_lrelu_adain(Fn.upsample_blur(x), self.c0[i], p[pre + "/b0/b"], *styles[(i, "s0")])
            else:
                h = Fn.conv_bias_lrelu_adain(x, self.c0[i], p[pre + "/b0/b"], *styles[(i, "s0")], upsample=True)
        elif self.enable_blur:
            h = Fn.conv_bias_lrelu(Fn.upsample_blur(x), self.c0[i], p[pre + "/b0/b"])
        else:
            h = Fn.conv_bias_lrelu(x, self.c0[i], p[pre + "/b0/b"], upsample=True)
        if i == 0 or styles is None:
            h = style(pre + "/s0", w, h)
        if styles is not None:
            return Fn.conv_bias_lrelu_adain(h, self.c1[i], p[pre + "/b1/b"], *styles[(i, "s1")])
        h = Fn.conv_bias_lrelu(h, self.c1[i], p[pre + "/b1/b"])
        return style(pre + "/s1", w, h)

    def rotate_w(self, w, theta):
        """net.py:220-224."""
        p = self.store.params
        h = torch.cat([w, theta * 16], dim=1)
        h = Fn.linear_act(h, p["l1/c/W"], p["l1/c/b"], _inv_c(self.ch + 9), act=True)
        return Fn.linear_act(h, p["l2/c/W"], p["l2/c/b"], _inv_c(self.ch), act=True)

    def _to_rgbd(self, i, h):
        """outs[i]: 1x1 conv (gain 1) from NHWC bf16 to NCHW fp32, fp32 accumulate (bandwidth-bound, Cout = 4)."""
        p = self.store.params
        W = p[f"outs/{i}/c/W"]
        return Fn.to_planes(h, W.reshape(W.shape[0], W.shape[1]), p[f"outs/{i}/c/b"], _inv_c(W.shape[1], 1.0))

    def __call__(self, w, w2, stage, theta=None, add_noise=True, return_feature=False):
        st, alpha = _split_stage(stage, self.max_stage)
        if self.rgbd and theta is None:
            raise AssertionError("theta is None")
        feat = None
        h = None

        # blocks 0-1 take the pose-conditioned latent, block 2 `w`, blocks 3.. `w2` (net.py:258-263): the style
        # affines of each run are computed by one linear when its first block comes up
        groups = {}

        def run(i, w_cur, h):
            n_main = (st - 2) // 2 + 2 if st % 2 == 0 else (st - 1) // 2 + 1      # blocks of the main path
            seg = (0, min(2, n_main)) if i < 2 else ((2, 3) if i == 2 else (3, n_main))
            if seg not in groups:
                src = self.rotate_w(w_cur, theta) if (self.rgbd and i < 2) else w_cur
                groups[seg] = self._style_group(src, *seg)
            return self._block(i, w_cur, h, styles=groups[seg])

        if st % 2 == 0:
            k = (st - 2) // 2
            for i in range(0, k + 2):
                if i == 3:
                    w = w2
                h = run(i, w, h)
                if return_feature and i == 3:
                    feat = h
            out = self._to_rgbd(k + 1, h)
        else:
            k = (st - 1) // 2
            for i in range(0, k + 1):
                if i == 3:
                    w = w2
                h = run(i, w, h)
                if return_feature and i == 3:
                    feat = h
            lo = self._to_rgbd(k, h)
            hi = self._to_rgbd(k + 1, self._block(k + 1, w, h))      # net.py:290: un-rotated w
            out = Fn.fade_planes(lo, hi, alpha)                       # (1 - alpha) * upscale2x(lo) + alpha * hi
        if self.rgbd:
            out = Fn.depth_head(out)                                  # net.py:296
        out = out.contiguous()
        if return_feature:
            return out, (feat.permute(0, 3, 1, 2).float() if feat is not None else None)
        return out

    forward = __call__


class StyleGANGenerator(_Link):
    def __init__(self, ch, enable_blur=False, rgbd=False, rotate_conv_input=False, use_encoder=False,
                 use_occupancy_net=False, initial_depth=None, device="cuda:0", seed=0, max_resolution=128):
        """max_resolution (not a reference argument: its 256+ blocks are commented out in the source, net.py:181-183): 256
        needs ch = 512 (ch/8 = 64 channels in the last block)."""
        assert not rotate_conv_input and not use_encoder and not use_occupancy_net, "unsupported generator option"
        assert ch % 256 == 0, "the MFMA conv engine needs ch/4 to be a multiple of 64"
        self.ch = ch
        self.seed = int(seed)
        self.device = torch.device(device)
        self.mapping = MappingNetwork(ch, device, seed)
        self.gen = StyleGenerator(ch, device, rgbd, 1.0 if initial_depth is None else initial_depth, seed + 1,
                                  enable_blur=bool(enable_blur), max_resolution=max_resolution)
        self.stores = (("mapping/", self.mapping.store), ("gen/", self.gen.store))
        self.train = True

    def make_hidden(self, batch_size, rng_state=None):
        """net.py:333-343; drawn on the device (the reference draws with cupy when on GPU).  rng_state: a latent stream of the
        caller's own (kernels.new_hidden_rng_state(device, seed)) instead of this generator's training stream -- the preview
        sampler's seeded latents (train_rgbd.py:39-92) neither depend on nor advance the training draws."""
        from . import kernels
        if torch.device(self.device).type == "cuda":
            state = rng_state if rng_state is not None else self._latent_rng()
            return kernels.hidden_draw(state, batch_size, self.ch * 2, self.ch).reshape(batch_size, self.ch * 2, 1, 1)
        z = torch.randn(batch_size, self.ch * 2, device=self.device)
        return kernels.hidden_normalize(z, self.ch).reshape(batch_size, self.ch * 2, 1, 1)

    def make_hidden_pairs(self, half):
        """The step's latent batch (updater.py:300: the same `half` latents for both views of every pair) in one launch
        after the draw: rows [0, half) and [half, 2 half) are identical."""
        from . import kernels
        return kernels.hidden_draw(self._latent_rng(), half, self.ch * 2, self.ch, copies=2).reshape(2 * half, self.ch * 2, 1, 1)

    def _latent_rng(self):
        """This generator's latent stream (kernels.new_hidden_rng_state), seeded at the first draw from torch's seed and the
        generator's own construction seed (its role: 0 for the trained generator, 1000 for the smoothed one, training.py): a
        model built after torch.manual_seed(s) draws the same sequence every time, in every process, and gen / smoothed_gen
        never share a stream."""
        from . import kernels
        if getattr(self, "_rng_state", None) is None:
            self._rng_state = kernels.new_hidden_rng_state(
                self.device, seed=torch.initial_seed() + 0x9E3779B97F4A7C15 * int(getattr(self, "seed", 0)))
        return self._rng_state

    def __call__(self, z, stage, theta=None, return_feature=False):
        z = _as_device_tensor(z, self.device).reshape(-1, 2 * self.ch)
        theta = _as_device_tensor(theta, self.device) if theta is not None else None
        # net.py:348-350 maps the two latent halves separately with the same network: one batch of 2B rows here
        n = z.shape[0]
        ww = self.mapping(torch.cat([z[:, :self.ch], z[:, self.ch:]], dim=0))
        w, w2 = ww[:n], ww[n:]
        out = self.gen(w, w2=w2, stage=stage, theta=theta, return_feature=return_feature)
        if not self.train and not return_feature and out.shape[2] < 64:     # net.py:305-309 (eval-mode upsample)
            scale = 64 // out.shape[2]
            out = upsample_planes(out, scale)
        return out

    forward = __call__


# ---------------------------------------------------------------------------------------------- DCGAN (PGGAN)
class DCGANGenerator(_Link):
    """net.py:603-773."""

    def __init__(self, in_ch=128, ch=512, enable_blur=False, rgbd=False, use_encoder=False, use_occupancy_net=False,
                 initial_depth=None, device="cuda:0", seed=0):
        assert not use_encoder and not use_occupancy_net, "unsupported generator option"
        assert ch % 512 == 0, "the channel-wise normalise kernel takes 128, 256 or 512 channels (ch = 512)"
        self.in_ch, self.ch, self.rgbd, self.max_stage = in_ch, ch, rgbd, 17
        self.enable_blur = bool(enable_blur)
        self.device = torch.device(device)
        self.chans = [(ch, ch), (ch, ch), (ch, ch), (ch // 2, ch), (ch // 4, ch // 2)]
        out_ch = 4 if rgbd else 3
        w_init, b_init = depth_row_init(1.0 if initial_depth is None else initial_depth, out_ch, rgbd)
        specs = [("linear/c/W", (ch * 16, in_ch + (9 if rgbd else 0)), "normal"), ("linear/c/b", (ch * 16,), "zeros")]
        for i, (co, ci) in enumerate(self.chans):
            pre = f"blocks/{i}"
            specs += [(pre + "/b0/b", (co,), "zeros"), (pre + "/b1/b", (co,), "zeros"),
                      (pre + "/n0/b/W", (co,), "zeros"), (pre + "/n1/b/W", (co,), "zeros"),
                      (pre + "/c0/c/W", (co, ci, 3, 3), "normal"), (pre + "/c1/c/W", (co, co, 3, 3), "normal")]
        for i, (co, _) in enumerate(self.chans):
            specs += [(f"outs/{i}/c/W", (out_ch, co, 1, 1), w_init), (f"outs/{i}/c/b", (out_ch,), b_init)]
        self.store = ParamStore(specs, device, seed)
        self.stores = (("", self.store),)
        p = self.store.params
        self.c0 = [Fn.ConvLayer(p[f"blocks/{i}/c0/c/W"], _inv_c(self.chans[i][1] * 9), 1) for i in range(5)]
        self.c1 = [Fn.ConvLayer(p[f"blocks/{i}/c1/c/W"], _inv_c(self.chans[i][0] * 9), 1) for i in range(5)]
        self.pack_group = Fn.PackGroup(self.c0 + self.c1)
        self.train = True

    def make_hidden(self, batch_size):
        z = torch.randn(batch_size, self.in_ch, device=self.device)
        return z / torch.sqrt(torch.sum(z * z, dim=1, keepdim=True) / self.in_ch + 1e-8)

    @staticmethod
    def _normalize(h):
        """chainer F.normalize over channels: x / (||x||_2 + 1e-5); fp32 norm (rgbd_l2norm_*)."""
        return Fn.l2_normalize(h)

    def _block(self, i, x):
        p = self.store.params
        pre = f"blocks/{i}"
        if self.enable_blur:
            h = Fn.conv_bias_lrelu(Fn.upsample_blur(x), self.c0[i], p[pre + "/b0/b"])
        else:
            h = Fn.conv_bias_lrelu(x, self.c0[i], p[pre + "/b0/b"], upsample=True)
        return self._normalize(Fn.conv_bias_lrelu(self._normalize(h), self.c1[i], p[pre + "/b1/b"]))

    def _to_rgbd(self, i, h):
        p = self.store.params
        W = p[f"outs/{i}/c/W"]
        return Fn.to_planes(h, W.reshape(W.shape[0], W.shape[1]), p[f"outs/{i}/c/b"], _inv_c(W.shape[1], 1.0))

    def __call__(self, z, stage, theta=None, style_mixing_rate=None, add_noise=True, return_feature=False):
        z = _as_device_tensor(z, self.device).reshape(-1, self.in_ch)
        st, alpha = _split_stage(stage, self.max_stage)
        if self.rgbd:
            if theta is None:
                raise AssertionError("theta is None")
            h = torch.cat([z, _as_device_tensor(theta, self.device) * 10], dim=1)
        else:
            h = z
        p = self.store.params
        # input layer (net.py:671,713-719): equalized linear to (ch,4,4) on the small-batch HIP linear kernels, rows in
        # slices of 64; then NCHW fp32 rows -> NHWC bf16 (rgbd_rows_to_nhwc_bf16)
        k_in = h.shape[1]
        rows = [Fn.dense(h[r0:r0 + 64].contiguous(), p["linear/c/W"], p["linear/c/b"], _inv_c(k_in), act=False)
                for r0 in range(0, h.shape[0], 64)]
        h = Fn.rows_to_nhwc(rows[0] if len(rows) == 1 else torch.cat(rows), 4, 4, self.ch)
        if st % 2 == 0:
            k = (st - 2) // 2
            for i in range(0, k + 1):
                h = self._block(i, h)
            out = self._to_rgbd(k, h)
        else:
            k = (st - 1) // 2
            for i in range(0, k):
                h = self._block(i, h)
            lo = self._to_rgbd(k - 1, h)
            hi = self._to_rgbd(k, self._block(k, h))
            out = Fn.fade_planes(lo, hi, alpha)
        if self.rgbd:
            out = Fn.depth_head(out)
        return out.contiguous()

    forward = __call__


# ---------------------------------------------------------------------------------------------- discriminator
class Discriminator(_Link):
    """net.py:429-504 with res blocks (net.py:380-426) and the 4x4 base block (net.py:357-377)."""

    def __init__(self, ch=512, out_dim=1, enable_blur=False, sn=False, res=False, device="cuda:0", seed=100,
                 max_resolution=128):
        assert ch % 256 == 0
        nb = self.n_blocks = _block_count(max_resolution)      # 6: the reference's (net.py:437-452); 7: + its commented 256 block
        self.ch, self.sn, self.res, self.max_stage = ch, bool(sn), res, 2 * nb + 5
        gch = _synthesis_chans(ch, nb)
        assert gch[-1][0] % 64 == 0, "the MFMA conv engine needs the first block's channels to be a multiple of 64"
        self.chans = [None] + [(gch[i][0], gch[i][1]) for i in range(1, nb)]       # (in, out): the generator's, mirrored
        self.in_chans = [c[0] for c in gch]
        if self.sn:
            self._init_sn(ch, out_dim, enable_blur, res, device, seed)
            return
        self.enable_blur = bool(enable_blur)       # net.py:422-423: blur(downscale2x(h)) at the end of every block
        self.device = torch.device(device)
        specs = [("blocks/0/c0/c/W", (ch, ch, 3, 3), "normal"), ("blocks/0/c0/c/b", (ch,), "zeros"),
                 ("blocks/0/c1/c/W", (ch, ch, 4, 4), "normal"), ("blocks/0/c1/c/b", (ch,), "zeros"),
                 ("blocks/0/l2/c/W", (out_dim, ch), "normal"), ("blocks/0/l2/c/b", (out_dim,), "zeros")]
        for i in range(1, nb):
            ci, co = self.chans[i]
            for nm in (("c0", "c1", "c_sc") if res else ("c0", "c1")):
                cin = co if nm == "c1" else ci
                specs += [(f"blocks/{i}/{nm}/c/W", (co, cin, 3, 3), "normal"), (f"blocks/{i}/{nm}/c/b", (co,), "zeros")]
        for i, co in enumerate(self.in_chans):
            specs += [(f"ins/{i}/c/W", (co, 3, 1, 1), "normal"), (f"ins/{i}/c/b", (co,), "zeros")]
        self.store = ParamStore(specs, device, seed)
        self.stores = (("", self.store),)
        p = self.store.params
        self.conv = {}
        self.conv["blocks/0/c0"] = Fn.ConvLayer(p["blocks/0/c0/c/W"], _inv_c(ch * 9), 1)
        for i in range(1, nb):
            ci, co = self.chans[i]
            for nm in (("c0", "c1", "c_sc") if res else ("c0", "c1")):
                cin = co if nm == "c1" else ci
                self.conv[f"blocks/{i}/{nm}"] = Fn.ConvLayer(p[f"blocks/{i}/{nm}/c/W"], _inv_c(cin * 9), 1)
        self.pack_group = Fn.PackGroup(list(self.conv.values()))

    # ---------------------------------------------------------------- spectral-norm variant (net.py:366-370,391-396,455-463)
    # Every convolution / linear is a plain (NOT equalized-LR) layer with bias, W ~ U(-1, 1), wrapped by chainer's
    # SpectralNormalization link hook (chainer >= 7, n_power_iteration=1, eps=1e-6, no gamma): at every forward call of a
    # layer, in train mode,   v = l2n(u W_m), u <- l2n(W_m v)   (W_m = W reshaped (Cout, -1), u persistent, l2n(x) =
    # x / (|x| + eps), no gradient), sigma = u^T W_m v (differentiable in W), and the layer runs with W / sigma.
    # The normalised weights are derived tensors (functional.DerivedConvLayer): weight gradients flow back to the
    # masters through the division by sigma.  RGBDUpdater runs the reference's literal three-forward step for such a
    # discriminator (no R1 penalty: updater.py:414), because every forward call moves u and with it the function.
    SN_EPS = 1e-6

    def _init_sn(self, ch, out_dim, enable_blur, res, device, seed):
        self.enable_blur = bool(enable_blur)
        self.device = torch.device(device)
        uni = lambda shape, gen: torch.rand(shape, generator=gen) * 2 - 1           # chainer.initializers.Uniform(1)
        specs, self.sn_layers = [], []
        def add(name, shape):
            specs.extend([(name + "/W", shape, uni), (name + "/b", (shape[0],), "zeros")])
            self.sn_layers.append(name)
        add("blocks/0/c0", (ch, ch, 3, 3))
        add("blocks/0/c1", (ch, ch, 4, 4))
        add("blocks/0/l2", (out_dim, ch))
        for i in range(1, self.n_blocks):
            ci, co = self.chans[i]
            for nm in (("c0", "c1", "c_sc") if res else ("c0", "c1")):
                add(f"blocks/{i}/{nm}", (co, co if nm == "c1" else ci, 3, 3))
        for i, co in enumerate(self.in_chans):
            add(f"ins/{i}", (co, 3, 1, 1))
        self.store = ParamStore(specs, device, seed)
        self.stores = (("", self.store),)
        gen = torch.Generator().manual_seed(seed + 1)
        # the hook's persistent vector, one per layer (saved with the link as <layer>/W_u)
        self.sn_u = {n: torch.randn(self.store.shapes[n + "/W"][0], generator=gen).to(self.device) for n in self.sn_layers}
        self.pack_group = None
        self.train = True

    def _sn_normalize(self, names):
        """One power iteration + W / sigma for the layers a forward call is about to run (what the link hook's
        forward_preprocess does layer by layer) -> ({layer: W / sigma}, {layer: conv-engine layer object}).  The layer
        objects are per CALL: a call's backward must read the bf16 images of ITS normalised weights, not those of a
        later call (the discriminator step differentiates two calls after both have run)."""
        p = self.store.params
        w, conv = {}, {}
        for n in names:
            W = p[n + "/W"]
            Wm = W.reshape(W.shape[0], -1)
            with torch.no_grad():
                u = self.sn_u[n]
                v = u @ Wm
                v = v / (torch.linalg.vector_norm(v) + self.SN_EPS)
                u_new = Wm @ v
                u_new = u_new / (torch.linalg.vector_norm(u_new) + self.SN_EPS)
                if self.train:
                    u.copy_(u_new)
            sigma = torch.dot(u_new @ Wm, v)
            w[n] = W / sigma
            if W.dim() == 4 and W.shape[2] == 3:
                conv[n] = Fn.DerivedConvLayer(lambda t=w[n]: t, 1.0, 3, 1)
        return w, conv

    def _sn_call(self, x, stage, return_hidden):
        x = _as_device_tensor(x, self.device)
        st, alpha = _split_stage(stage, self.max_stage)
        p, blocks_of = self.store.params, lambda i: [f"blocks/{i}/{nm}" for nm in
                                                     (("c0", "c1", "l2") if i == 0 else
                                                      (("c0", "c1", "c_sc") if self.res else ("c0", "c1")))]
        if st % 2 == 0:
            k = (st - 2) // 2
            used = [f"ins/{k + 1}"] + [n for i in range(0, k + 2) for n in blocks_of(i)]
        else:
            k = (st - 1) // 2
            used = [f"ins/{k}", f"ins/{k + 1}"] + [n for i in range(0, k + 2) for n in blocks_of(i)]
        w, conv = self._sn_normalize(used)

        def from_rgb(i, img):
            W = w[f"ins/{i}"]
            return Fn.from_planes(img, W.reshape(W.shape[0], 3), p[f"ins/{i}/b"], 1.0, act=True)

        def block(i, h):
            pre = f"blocks/{i}"
            if i == 0:
                h = Fn.conv_bias_lrelu(h, conv[pre + "/c0"], p[pre + "/c0/b"])
                rows = Fn.nhwc_to_rows(h)
                outs = []
                for r0 in range(0, rows.shape[0], 64):
                    u = Fn.dense(rows[r0:r0 + 64], w[pre + "/c1"], p[pre + "/c1/b"], 1.0, act=True)
                    outs.append(Fn.dense(u, w[pre + "/l2"], p[pre + "/l2/b"], 1.0, act=False))
                return outs[0] if len(outs) == 1 else torch.cat(outs)
            hh = Fn.conv_bias_lrelu(h, conv[pre + "/c0"], p[pre + "/c0/b"])
            sc = Fn.conv_bias(h, conv[pre + "/c_sc"], p[pre + "/c_sc/b"]) if self.res else None
            hh = Fn.conv_bias_lrelu(hh, conv[pre + "/c1"], p[pre + "/c1/b"], residual=sc, pool=True)
            return Fn.blur(hh) if self.enable_blur else hh

        feat = None
        if st % 2 == 0:
            h = from_rgb(k + 1, x)
            for i in reversed(range(0, k + 2)):
                if i == 3:
                    feat = h
                h = block(i, h)
        else:
            h0 = from_rgb(k, Fn.avg_pool2_planes(x))
            h1 = block(k + 1, from_rgb(k + 1, x))
            h = Fn.lerp(h0, h1, alpha)
            for i in reversed(range(0, k + 1)):
                if i == 3:
                    feat = h
                h = block(i, h)
        if return_hidden:
            return h, (feat.permute(0, 3, 1, 2).float() if feat is not None else None)
        return h

    def state_dict(self):
        out = super().state_dict()
        if self.sn:
            out.update({n + "/W_u": u.detach().cpu().numpy().copy() for n, u in self.sn_u.items()})
        return out

    def load_state_dict(self, arrays, strict=True):
        super().load_state_dict(arrays, strict=strict)
        if self.sn:
            for n, u in self.sn_u.items():
                if n + "/W_u" in arrays:
                    u.copy_(torch.as_tensor(np.asarray(arrays[n + "/W_u"])).to(self.device, torch.float32))
                elif strict:
                    raise KeyError(n + "/W_u")


    def tail_params(self):
        """Parameters of the dense tail after the conv stack (4x4 valid conv as a linear + the output linear), the part
        of the discriminator that runs through torch ops."""
        p = self.store.params
        return [p["blocks/0/c1/c/W"], p["blocks/0/c1/c/b"], p["blocks/0/l2/c/W"], p["blocks/0/l2/c/b"]]

    def _from_rgb(self, i, x):
        """ins[i]: 1x1 conv 3 -> C on the NCHW fp32 image, + bias, leaky ReLU, to NHWC bf16 (one HBM-bound kernel)."""
        p = self.store.params
        W = p[f"ins/{i}/c/W"]
        return Fn.from_planes(x, W.reshape(W.shape[0], 3), p[f"ins/{i}/c/b"], _inv_c(3), act=True)

    def _block(self, i, x):
        p = self.store.params
        pre = f"blocks/{i}"
        if i == 0:
            h = Fn.conv_bias_lrelu(x, self.conv[pre + "/c0"], p[pre + "/c0/c/b"])
            # 4x4 valid conv == linear over (c,h,w) (net.py:363-365,372-377).  The ACTIVATION (0.5 MB) is brought into
            # the weight's own (ci,kh,kw) order as fp32 rows, not the 4 MB weight into NHWC order: no weight copy forward,
            # and the weight gradient comes out in the master layout.  Both layers run on the small-batch fp32 MFMA
            # linear kernels (rgbd_linear_*), twice differentiable for the R1 penalty; batches above 64 rows in slices.
            W = p[pre + "/c1/c/W"]
            rows = Fn.nhwc_to_rows(h)
            outs = []
            for r0 in range(0, rows.shape[0], 64):
                u = Fn.dense(rows[r0:r0 + 64], W, p[pre + "/c1/c/b"], _inv_c(W.shape[1] * 16), act=True)
                outs.append(Fn.dense(u, p[pre + "/l2/c/W"], p[pre + "/l2/c/b"], _inv_c(self.ch, 1.0), act=False))
            return outs[0] if len(outs) == 1 else torch.cat(outs)
        # net.py:408-426: h = lrelu(c0 x); h = lrelu(c1 h + c_sc x); avg-pool.  Bias, shortcut add and activation
        # all ride in the conv epilogues; the 2x2 average pool and its backward are fused with the activation gradient.
        tie = Fn.ResidualTie(p[pre + "/c_sc/c/b"]) if self.res else None
        h = Fn.conv_bias_lrelu(x, self.conv[pre + "/c0"], p[pre + "/c0/c/b"], entry_tie=tie)
        sc = Fn.conv_bias(x, self.conv[pre + "/c_sc"], p[pre + "/c_sc/c/b"], tie=tie) if self.res else None
        h = Fn.conv_bias_lrelu(h, self.conv[pre + "/c1"], p[pre + "/c1/c/b"], residual=sc, pool=True,
                               residual_tie=tie)
        return Fn.blur(h) if self.enable_blur else h

    def __call__(self, x, stage, return_hidden=False):
        if self.sn:
            return self._sn_call(x, stage, return_hidden)
        x = _as_device_tensor(x, self.device)
        st, alpha = _split_stage(stage, self.max_stage)
        feat = None
        if st % 2 == 0:
            k = (st - 2) // 2
            h = self._from_rgb(k + 1, x)
            for i in reversed(range(0, k + 2)):
                if i == 3:
                    feat = h
                h = self._block(i, h)
        else:
            k = (st - 1) // 2
            h0 = self._from_rgb(k, Fn.avg_pool2_planes(x))
            h1 = self._block(k + 1, self._from_rgb(k + 1, x))
            h = Fn.lerp(h0, h1, alpha)                                # (1 - alpha) * h0 + alpha * h1, one rounding
            for i in reversed(range(0, k + 1)):
                if i == 3:
                    feat = h
                h = self._block(i, h)
        if return_hidden:
            return h, (feat.permute(0, 3, 1, 2).float() if feat is not None else None)
        return h

    forward = __call__
