"""Flat parameter storage: all parameters that one optimizer owns live in ONE contiguous fp32 buffer.

Why: (1) clipping + Adam become a handful of launches over flat arrays instead of one kernel per parameter
tensor (the reference's chainer.optimizers.Adam, train_rgbd.py:151-161, launches ~150); (2) the data-parallel
gradient exchange is ONE all-reduce per optimizer on the flat gradient buffer -- the same packing ChainerMN's
pure_nccl communicator does (train_rgbd.py:154-156) -- sized for xGMI (2.1 / 26.7 / 33.7 MB).

Names are the reference's Chainer ``namedparams`` paths so its .npz snapshots load directly.
"""
import contextlib
import math

import numpy as np
import torch


class ParamStore:
    def __init__(self, specs, device, seed=0):
        """specs: list of (name, shape, init) with init in {'normal','zeros','ones'} or a float constant
        or a callable(shape, generator) -> CPU tensor."""
        self.device = torch.device(device)
        self.names, self.shapes, self.offsets = [], {}, {}
        total = 0
        for name, shape, _ in specs:
            shape = tuple(int(s) for s in shape)
            self.names.append(name)
            self.shapes[name] = shape
            self.offsets[name] = total
            n = int(np.prod(shape)) if len(shape) else 1
            total += (n + 3) // 4 * 4          # keep every tensor 16-byte aligned for vector loads
        self.numel = total
        gen = torch.Generator().manual_seed(seed)
        host = torch.zeros(total, dtype=torch.float32)
        for name, shape, init in specs:
            n = int(np.prod(self.shapes[name]))
            off = self.offsets[name]
            if init == "normal":
                host[off:off + n] = torch.randn(n, generator=gen)
            elif init == "zeros":
                pass
            elif init == "ones":
                host[off:off + n] = 1.0
            elif callable(init):
                host[off:off + n] = init(self.shapes[name], gen).reshape(-1)
            else:
                host[off:off + n] = float(init)
        self.flat = host.to(self.device)
        self.grad = torch.zeros_like(self.flat)
        self.params = {}
        for name in self.names:
            n = int(np.prod(self.shapes[name]))
            off = self.offsets[name]
            p = self.flat[off:off + n].view(self.shapes[name])
            p.requires_grad_(True)
            self.params[name] = p
        self._fused = {}
        self.grad_alt = None          # second gradient buffer (alt_grads), allocated on first use
        self.bind_grads()

    def bind_grads(self, buf=None):
        """Point every .grad at its slice of the flat gradient buffer so autograd accumulates in place."""
        buf = self.grad if buf is None else buf
        self._bound = buf
        for name, p in self.params.items():
            n = p.numel()
            off = self.offsets[name]
            p.grad = buf[off:off + n].view(self.shapes[name])
        for (names, _), (p, off) in self._fused.items():
            p.grad = buf[off:off + p.numel()].view(p.shape)

    @contextlib.contextmanager
    def alt_grads(self):
        """Inside, gradients accumulate into a SECOND flat buffer: two passes that run concurrently on different
        streams (D on the fakes inside the generator phase, D on the reals in the discriminator phase) must not
        read-modify-write the same gradient memory.  merge_alt() adds it back."""
        if self.grad_alt is None:
            self.grad_alt = torch.zeros_like(self.flat)
        self.bind_grads(self.grad_alt)
        try:
            yield
        finally:
            self.bind_grads()

    def merge_alt(self):
        if self.grad_alt is not None:
            from . import kernels
            n4 = self.grad.numel() // 4 * 4          # flat buffers are padded to 16-byte granules per tensor
            kernels.axpy_rows_f32(self.grad[:n4], self.grad_alt[:n4], out=self.grad[:n4])
            if n4 != self.grad.numel():
                self.grad[n4:].add_(self.grad_alt[n4:])

    def fused(self, names, shape):
        """One leaf tensor over several parameters that sit back to back in the flat buffer (e.g. the scale and shift
        affines of a style block, stacked along the output dimension), with its gradient bound to the same span of the
        flat gradient buffer: a single kernel launch then serves all of them."""
        key = (tuple(names), tuple(shape))
        if key not in self._fused:
            off = self.offsets[names[0]]
            pos = off
            for nm in names:
                if self.offsets[nm] != pos:
                    raise ValueError(f"fused{tuple(names)}: parameters are not adjacent in the flat buffer")
                pos += int(np.prod(self.shapes[nm]))
                if int(np.prod(self.shapes[nm])) % 4:
                    raise ValueError(f"fused: {nm} is padded")
            n = pos - off
            if n != int(np.prod(shape)):
                raise ValueError("fused: shape does not cover the parameters")
            p = self.flat[off:off + n].view(shape)
            p.requires_grad_(True)
            p.grad = self.grad[off:off + n].view(shape)
            self._fused[key] = (p, off)
        return self._fused[key][0]

    def zero_grad(self, defer=None):
        """Clear the flat gradient buffer(s).  defer: a list -- the buffers are appended to it instead of cleared here,
        so the caller can clear the buffers of several stores with one launch (kernels.zero_multi)."""
        if defer is not None:
            defer.append(self.grad)
            if self.grad_alt is not None:
                defer.append(self.grad_alt)
        else:
            self.grad.zero_()
            if self.grad_alt is not None:
                self.grad_alt.zero_()
        for name, p in self.params.items():
            g = p.grad
            if g is None or g.data_ptr() != self.grad.data_ptr() + 4 * self.offsets[name]:
                self.bind_grads()
                break
        for p, off in self._fused.values():
            if p.grad is None or p.grad.data_ptr() != self.grad.data_ptr() + 4 * off:
                self.bind_grads()
                break

    def __getitem__(self, name):
        return self.params[name]

    def __contains__(self, name):
        return name in self.params

    def state_dict(self):
        return {k: v.detach().cpu().numpy().copy() for k, v in self.params.items()}

    @torch.no_grad()
    def load(self, arrays, strict=True):
        """arrays: name -> ndarray/tensor.  Missing keys are skipped when strict=False (chainer load_npz strict=False)."""
        from . import functional
        for name, p in self.params.items():
            if name not in arrays:
                if strict:
                    raise KeyError(name)
                continue
            src = torch.as_tensor(np.asarray(arrays[name]) if not torch.is_tensor(arrays[name]) else arrays[name])
            if tuple(src.shape) != tuple(p.shape):
                raise ValueError(f"{name}: shape {tuple(src.shape)} != {tuple(p.shape)}")
            p.copy_(src.to(self.device, torch.float32))
        functional.bump_weight_epoch()


def depth_row_init(initial_depth, out_ch, rgbd):
    """net.py:211-214: the depth row of every `outs` conv starts at W=0, b=log(e^initial_depth - 1)."""
    def w_init(shape, gen):
        w = torch.randn(*shape, generator=gen)
        if rgbd:
            w[-1] = 0
        return w

    def b_init(shape, gen):
        b = torch.zeros(*shape)
        if rgbd:
            b[-1] = math.log(math.e ** initial_depth - 1)
        return b
    return w_init, b_init
