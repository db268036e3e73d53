"""Chainer-flavoured Adam with global-norm clipping over a flat parameter store, data-parallel aware.

Semantics restated from train_rgbd.py:151-161 + chainer v7 (see oracle/step.py:ChainerAdam):
  * GradientClipping(5): rate = 5 / ||g||_2 over THIS optimizer's parameters, applied when rate < 1
  * Adam: m += (1-b1)(g-m); v += (1-b2)(g^2-v); p -= alpha*sqrt(1-b2^t)/(1-b1^t) * m / (sqrt(v) + eps)
  * per-parameter alpha overrides (train_rgbd.py:323-327: gen.l1 / gen.l2 use alpha/100)
  * under data parallelism (ChainerMN multi-node optimizer, train_rgbd.py:154-156): gradients are averaged with ONE
    all-reduce on the flat buffer, then every rank clips and steps on identical values; the FIRST update() of a
    multi-node optimizer only broadcasts rank 0's parameters and does not step.
"""
import numpy as np
import torch

from . import functional, kernels


class FlatAdam:
    def __init__(self, store, alpha, beta1=0.0, beta2=0.999, eps=1e-8, clip=5.0, comm=None):
        self.store = store
        self.alpha, self.beta1, self.beta2, self.eps, self.clip = alpha, beta1, beta2, eps, clip
        self.comm = comm                      # rgbd_gan_amd.dist.Communicator or None
        self.step = torch.zeros(1, dtype=torch.int32, device=store.flat.device)   # chainer's t, kept on the device
        self.m = torch.zeros_like(store.flat)
        self.v = torch.zeros_like(store.flat)
        self.workspace = torch.empty(1024 + 8, dtype=torch.float32, device=store.flat.device)
        self.grad_norm = torch.zeros(1, dtype=torch.float32, device=store.flat.device)
        self._alpha_of = {}
        self._needs_broadcast = comm is not None and comm.active
        self._pending = None

    # chainer: param.update_rule.hyperparam.alpha = x
    def set_alpha(self, name, alpha):
        self._alpha_of[name] = alpha

    @property
    def target(self):
        return self.store

    def _segments(self):
        begins, alphas = [0], []
        for name in self.store.names:
            a = self._alpha_of.get(name, self.alpha)
            off = self.store.offsets[name]
            if alphas and alphas[-1] == a:
                continue
            if alphas:
                begins.append(off)
            alphas.append(a)
        begins.append(self.store.numel)
        return begins, alphas

    def start_allreduce(self):
        """Launch the gradient all-reduce (sum) asynchronously; update() waits for it."""
        if self.comm is not None and self.comm.active and not self._needs_broadcast:
            self._pending = self.comm.allreduce_async(self.store.grad)

    def finish_allreduce(self):
        """Make the current stream wait for this optimizer's gradient all-reduce (launching it now if nobody has).
        Called eagerly before the optimizer phase, so that phase holds no collective and can be replayed as a graph."""
        if self.comm is not None and self.comm.active and not self._needs_broadcast:
            if self._pending is None:
                self.start_allreduce()
            self.comm.wait(self._pending)
            self._pending = None
            self._reduced = True

    @torch.no_grad()
    def update(self, bump=True):
        """bump=False: the caller invalidates the packed-weight caches itself (RGBDUpdater steps D on the side stream while
        the generator's backward is still reading ITS packed weights on the main stream)."""
        if self._needs_broadcast:                      # ChainerMN: first update() = broadcast, no step
            self.comm.broadcast(self.store.flat)
            self._needs_broadcast = False
            functional.bump_weight_epoch(self.store.flat)
            return
        grad_scale = 1.0
        if self.comm is not None and self.comm.active:
            if not getattr(self, "_reduced", False):
                self.finish_allreduce()
            self._reduced = False
            grad_scale = 1.0 / self.comm.size
        begins, alphas = self._segments()
        kernels.adam_clip_multi(self.store.flat, self.store.grad, self.m, self.v, begins, alphas, self.beta1,
                                self.beta2, self.eps, self.clip, grad_scale, self.step, self.workspace, self.grad_norm)
        if bump:
            functional.bump_weight_epoch(self.store.flat)

    @property
    def t(self):
        return int(self.step.item())

    def state_dict(self):
        return {"t": self.t, "m": self.m.cpu().numpy(), "v": self.v.cpu().numpy()}

    def load_state_dict(self, sd):
        self.step.fill_(int(sd["t"]))
        self.m.copy_(torch.as_tensor(sd["m"]))
        self.v.copy_(torch.as_tensor(sd["v"]))
