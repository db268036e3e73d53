"""Autograd glue: differentiable ops whose forward AND backward are HIP kernels behind the C ABI.

The convolution triple (fprop, dgrad, wgrad) is closed under differentiation, which is what the R1 penalty
(updater.py:414-418 of the reference: chainer.grad(..., enable_double_backprop=True)) needs:

    d fprop(x,W)  -> dgrad(dy,W), wgrad(x,dy)
    d dgrad(dy,W) -> fprop(ddx,W), wgrad(ddx,dy)

so every backward below is itself built from these Functions and stays differentiable.

Layout: activations are NHWC bf16 tensors of shape (B,H,W,C); master weights are OIHW fp32 leaves
(the reference's layout) and the bf16 packed images the kernels read are cached per layer.
"""
import contextlib
import os

import torch
import torch.nn.functional as F

from . import kernels

_WEIGHT_EPOCH = 0          # bumped whenever master weights change (optimizer step, checkpoint load) ...
_EXTERNAL_EPOCH = 0        # ... and this one only when they change OUTSIDE a training step (checkpoint load, parameter copy, dtype
                           # switch): work a step has started for the next one (DeepVoxelsUpdater's generator forward) is stale then
_STORE_EPOCH = {}          # ... of which the optimizer steps name the flat buffer they changed (by storage address): a layer whose
                           # master lives in another buffer keeps its packed images (the DeepVoxels step updates G in the middle
                           # and D at the end: without this G's 18 folded + 16 packed weights were rebuilt twice per step)
_SKIP_WGRAD = False        # set while only input gradients are wanted (R1's inner grad)
_FROZEN_PTRS = frozenset()  # parameters (by storage address) whose gradients the current backward must not produce
_INJECT = None             # per-sample seeds (B,) fp32 of the adversarial loss while the R1 double backward runs
_MXFP8 = False             # 3x3 convolutions whose reduction channels are a multiple of 128 run their fprop / dgrad on MXFP8
                           # operands (BASELINE configuration 5; kernels.Mx8Image); weight gradients stay bf16


def set_conv_dtype(name):
    """"bf16" (default) or "mxfp8": the arithmetic of the 3x3 convolutions' fprop / dgrad launches from now on (YAML key
    `conv_dtype`).  Layers and launches the MXFP8 kernel does not cover (reduction channels not a multiple of 128, images
    below 16x16, 1x1 convs, weight gradients) stay on the bf16 kernels either way."""
    global _MXFP8
    if name not in ("bf16", "mxfp8", None):
        raise ValueError(f"conv_dtype must be 'bf16' or 'mxfp8', got {name!r}")
    new = name == "mxfp8"
    if new != _MXFP8:
        _MXFP8 = new
        bump_weight_epoch()         # the next packed() builds (or drops) the fp8 images


def conv_dtype():
    return "mxfp8" if _MXFP8 else "bf16"


MX8_COVERAGES = ("all", "fprop_only", "dis_only", "gen_only", "gen_fprop_only", "gen_skip_last2", "gen_fprop_only+skip_last2")


def apply_mx8_coverage(generator, discriminator, spec):
    """Which 3x3 launches `conv_dtype: mxfp8` covers (YAML key `mxfp8_coverage`; profiles/r06/soak_fp8_ablation.txt is why
    there is a choice):
        all               every eligible fprop and dgrad launch of both networks
        fprop_only        forward launches only; every input-gradient launch on bf16 operands (no E4M3 dy anywhere)
        dis_only / gen_only     one network on fp8, the other on bf16
        gen_fprop_only    D as `all`; the generator's forward on fp8, its backward (dy chain towards the styles and the
                          mapping network, the path the 3-D consistency loss trains) on bf16
        gen_skip_last2    D as `all`; the generator's last two synthesis blocks (in front of the RGB / depth heads) on bf16
        gen_fprop_only+skip_last2   both restrictions on the generator
    Sets per-layer flags; the fp8 weight images are still built for every eligible layer (one launch)."""
    if spec in (None, "", "all"):
        spec = "all"
    if spec not in MX8_COVERAGES:
        raise ValueError(f"mxfp8_coverage must be one of {MX8_COVERAGES}, got {spec!r}")
    gnet = getattr(generator, "gen", generator)
    g_layers = [l for l in getattr(getattr(gnet, "pack_group", None), "layers", [])]
    d_layers = [l for l in getattr(getattr(discriminator, "pack_group", None), "layers", [])]
    for l in g_layers + d_layers:
        l.mx_fprop = l.mx_dgrad = True
    if spec == "fprop_only":
        for l in g_layers + d_layers:
            l.mx_dgrad = False
    if spec == "dis_only":
        for l in g_layers:
            l.mx_fprop = l.mx_dgrad = False
    if spec == "gen_only":
        for l in d_layers:
            l.mx_fprop = l.mx_dgrad = False
    if spec in ("gen_fprop_only", "gen_fprop_only+skip_last2"):
        for l in g_layers:
            l.mx_dgrad = False
    if spec in ("gen_skip_last2", "gen_fprop_only+skip_last2"):
        c0, c1 = getattr(gnet, "c0", []), getattr(gnet, "c1", [])
        for l in [x for x in list(c0[-2:]) + list(c1[-2:]) if x is not None]:
            l.mx_fprop = l.mx_dgrad = False
    return spec


def bump_weight_epoch(flat=None, own_step=False):
    """Master weights changed: everywhere (flat=None), or in the one flat parameter buffer `flat`.  own_step: the caller is an
    updater invalidating Python's view behind its own replayed step (not an outside change)."""
    global _WEIGHT_EPOCH, _EXTERNAL_EPOCH
    if flat is None:
        _WEIGHT_EPOCH += 1
        if not own_step:
            _EXTERNAL_EPOCH += 1
    else:
        key = flat.untyped_storage().data_ptr()
        _STORE_EPOCH[key] = _STORE_EPOCH.get(key, 0) + 1


def external_epoch():
    return _EXTERNAL_EPOCH


def _epoch_of(w):
    """The (global, own buffer) epoch pair of a master weight tensor."""
    return _WEIGHT_EPOCH, _STORE_EPOCH.get(w.untyped_storage().data_ptr(), 0)


@contextlib.contextmanager
def input_grads_only():
    """Inside, conv backward skips weight gradients (used for g = dD(x)/dx of the R1 penalty)."""
    global _SKIP_WGRAD
    old = _SKIP_WGRAD
    _SKIP_WGRAD = True
    try:
        yield
    finally:
        _SKIP_WGRAD = old


@contextlib.contextmanager
def weight_grads_frozen(link):
    """Inside, backward passes skip the weight / bias gradients of `link`'s parameters although the forward recorded
    them as differentiable: the generator step differentiates D(x_fake) w.r.t. its input only, and the SAME recorded
    forward is differentiated w.r.t. the discriminator's weights later, in the discriminator step."""
    global _FROZEN_PTRS
    old = _FROZEN_PTRS
    _FROZEN_PTRS = old | frozenset(p.data_ptr() for p in link.params())
    try:
        yield
    finally:
        _FROZEN_PTRS = old


@contextlib.contextmanager
def adversarial_injection(seeds):
    """Fold the adversarial term of the discriminator loss into the R1 double backward (updater.py:405-422).

    With s_b = dL_adv/dy_b and dz1 the gradients of the R1 first-order pass (seed 1 per sample), the adversarial
    backward through D(x_real) is s_b * dz1 sample by sample (D has no batch coupling), so for every convolution
        dW = wgrad(ddx, dz1)  [R1 double backward]  +  wgrad(x, s_b * dz1)  [adversarial]  =  wgrad(ddx + s_b * x, dz1)
        db = sum_b s_b * colsum_b(dz1)
    i.e. the second input-gradient chain and the second weight-gradient pass of the reference's loss_dis.backward()
    are replaced by one elementwise operand update per layer.  Inside this context the double-backward nodes that
    know their forward input apply exactly that."""
    global _INJECT
    old = _INJECT
    _INJECT = seeds.reshape(-1).float().contiguous()
    try:
        yield
    finally:
        _INJECT = old


def _skip_grad_of(p):
    return _SKIP_WGRAD or (p is not None and p.data_ptr() in _FROZEN_PTRS)


class ConvLayer:
    """One equalized-LR 3x3 (or 1x1) convolution: master weight + cached packed bf16 images.

    pggan.py:13-24: y = conv(inv_c * x, W) (+ b).  inv_c is folded into the packed weights, so
    dW_master = inv_c * wgrad(x, dy).
    """

    def __init__(self, weight, inv_c, pad):
        self.weight = weight              # (Cout,Cin,K,K) fp32 leaf
        self.inv_c = float(inv_c)
        self.K = weight.shape[2]
        self.pad = pad
        self._epoch = -1
        self._wf = self._wd = None

    group = None              # PackGroup: all convolutions of a network repacked by one launch
    _mxf = _mxd = None        # kernels.Mx8Image twins of _wf / _wd while conv_dtype is "mxfp8" (None: not eligible)

    def _now(self):
        m = getattr(self, "master", None)
        w = m if m is not None else getattr(self, "_epoch_src", None)
        if w is None:
            w = self.__dict__.get("weight")          # (a derived layer without a master: any store's step invalidates it)
        return _epoch_of(w) if w is not None else (_WEIGHT_EPOCH, sum(_STORE_EPOCH.values()))

    def packed(self):
        if self._epoch != self._now():
            if self.group is not None:
                self.group.repack()
            else:
                with torch.no_grad():
                    w = self.weight.detach()
                    self._wf, self._wd = kernels.pack_weights(w, self.inv_c)
                    self._mxf = self._mxd = None
                    if _MXFP8 and self.K == 3 and self.pad == 1 and w.shape[0] % 32 == 0 and w.shape[1] % 32 == 0:
                        f, d = kernels.pack_weights_mx8(w, self.inv_c)
                        self._mxf = kernels.Mx8Image(self._wf, *f) if f is not None else None
                        self._mxd = kernels.Mx8Image(self._wd, *d) if d is not None else None
                self._epoch = self._now()
        if _MXFP8:
            return (self._mxf if self.mx_fprop else None) or self._wf, (self._mxd if self.mx_dgrad else None) or self._wd
        return self._wf, self._wd

    mx_fprop = mx_dgrad = True    # conv_dtype mxfp8: may THIS layer's fprop / dgrad launches take fp8 operands (apply_mx8_coverage)


class PackGroup:
    """The convolutions of one network share persistent bf16 weight images and a device-resident descriptor table:
    after an optimizer update the first layer that needs its image refreshes ALL of them with a single launch
    (rgbd_pack_weights_multi) instead of one launch per layer."""

    def __init__(self, layers):
        self.layers = [l for l in layers if l is not None]
        entries = []
        for l in self.layers:
            w = l.weight.detach()
            co, ci, kh, kw = w.shape
            l._wf = torch.empty(kh * kw, co, ci, dtype=torch.bfloat16, device=w.device)
            l._wd = torch.empty(kh * kw, ci, co, dtype=torch.bfloat16, device=w.device)
            l.group = self
            entries.append((w, l.inv_c, l._wf, l._wd))
        self.table = kernels.build_pack_table(entries)
        self.mx_table = None

    def _build_mx(self):
        """Persistent MXFP8 images of the eligible layers (3x3 pad 1; fprop image when Cin % 128 == 0, dgrad image when
        Cout % 128 == 0) and their descriptor table: one more launch per repack."""
        entries = []
        for l in self.layers:
            w = l.weight.detach()
            co, ci, kh, kw = w.shape
            l._mxf = l._mxd = None
            if not (kh == 3 and l.pad == 1 and co % 32 == 0 and ci % 32 == 0 and (ci % 128 == 0 or co % 128 == 0)):
                continue
            u8 = lambda *shape: torch.empty(*shape, dtype=torch.uint8, device=w.device)
            f = (u8(9, co, ci), u8(9, co, ci // 32)) if ci % 128 == 0 else (None, None)
            d = (u8(9, ci, co), u8(9, ci, co // 32)) if co % 128 == 0 else (None, None)
            l._mxf = kernels.Mx8Image(l._wf, *f) if f[0] is not None else None
            l._mxd = kernels.Mx8Image(l._wd, *d) if d[0] is not None else None
            entries.append((w, l.inv_c) + f + d)
        self.mx_table = kernels.build_pack_table_mx8(entries) if entries else ()

    def mark_current(self):
        """The persistent images ARE those of the current master weights (a captured phase rebuilt them on the device behind the
        last change, whatever Python's epochs say after a replay): the next packed() does not rebuild them."""
        for l in self.layers:
            if l._wf is not None:
                l._epoch = l._now()

    def repack(self):
        with torch.no_grad():
            kernels.pack_weights_multi(self.table)
            if _MXFP8:
                if self.mx_table is None:
                    self._build_mx()
                if self.mx_table:
                    kernels.pack_weights_mx8_multi(self.mx_table)
        for l in self.layers:
            l._epoch = l._now()


class DerivedPackGroup(PackGroup):
    """PackGroup for a network whose conv weights are (partly) DerivedConvLayers: every derived layer with a master gets a
    PERSISTENT folded fp32 buffer (rgbd_fold_weight_f32 writes into it), the packed bf16 images are persistent too, and one
    rgbd_pack_weights_multi launch packs all layers from (folded buffer | master).  Persistent because a captured phase that
    finds the images valid must be able to read, on every replay, what ANOTHER captured phase rebuilt (the DeepVoxels step:
    the discriminator's half rebuilds G's images after G's update, the next generator step reads them)."""

    def __init__(self, layers):
        self.layers = [l for l in layers if l is not None]
        entries = []
        for l in self.layers:
            fold = ()
            if isinstance(l, DerivedConvLayer):
                if l.master is None or l.fold is None:
                    raise ValueError("DerivedPackGroup: derived layers need (master, fold)")
                mode, cop, cip = l.fold
                m = l.master
                shape = (cop, 3 * cip, 3, 3) if mode == 0 else (cop, 16 * cip, 1, 1) if mode == 1 else (cop, cip, l.K, l.K)
                # the folded fp32 weight exists as a SHAPE only (the conv nodes' `w`: autograd routes its gradient through the
                # fold's adjoint); its values are never needed -- the packing kernel reads the master through the fold
                # (rgbd_pack_desc.fold, ABI 20: one launch and 8 bytes per folded element less on every rebuild)
                l._fold_buf = torch.zeros(shape, dtype=torch.float32, device=m.device)
                w = m.detach()
                if not w.is_contiguous():
                    raise ValueError("DerivedPackGroup: master parameters must be contiguous")
                fold = (shape, (mode, m.shape[0], m.shape[1]))
                co, ci, kh, kw = shape
            else:
                w = l.weight.detach()
                co, ci, kh, kw = w.shape
            l._wf = torch.empty(kh * kw, co, ci, dtype=torch.bfloat16, device=w.device)
            l._wd = torch.empty(kh * kw, ci, co, dtype=torch.bfloat16, device=w.device)
            l.group = self
            entries.append((w, l.inv_c, l._wf, l._wd) + fold)
        self.table = kernels.build_pack_table(entries)
        self.mx_table = ()          # (the networks of this kind run bf16 convs)

    def repack(self):
        with torch.no_grad():
            kernels.pack_weights_multi(self.table)                 # folds included: one launch for the whole network
        for l in self.layers:
            l._epoch = l._now()


class DerivedConvLayer(ConvLayer):
    """A convolution whose kernel-side weight is a differentiable rearrangement of a reference-shaped master
    parameter: zero padding of channels up to the 64-channel granularity of the MFMA kernels, the depth taps of a
    3x3x3 kernel folded into input channels, the 4x4 stride-2 taps folded into input channels (deepvoxels path).
    `derive()` returns the (Cout',Cin',K,K) fp32 tensor attached to the master, so weight gradients flow back through
    ordinary autograd instead of the direct flat-buffer accumulation."""

    def __init__(self, derive, inv_c, K, pad, master=None, fold=None):
        """master / fold = (mode, Cop, Cip) of kernels.fold_weight: when given, a backward pass run under deferred_wgrads
        routes this layer's weight gradient around autograd -- into a temporary through the batched weight-gradient launch,
        then through the fold's adjoint straight into master.grad."""
        self.derive = derive
        self.inv_c = float(inv_c)
        self.K, self.pad = K, pad
        self.master, self.fold = master, fold
        self._epoch = -1
        self._wf = self._wd = None

    @property
    def weight(self):
        # one derivation per weight epoch and autograd context: the conv node and packed() ask for it in the same forward
        # pass (a tensor derived with autograd history also serves a later no-grad caller of the same epoch)
        c = getattr(self, "_derived", None)
        now = self._now()
        if c is not None and c[0] == now and (c[1].requires_grad or not torch.is_grad_enabled()):
            return c[1]
        if self.group is not None:
            # grouped: the folded weight is the group's persistent buffer (refreshed by packed()); autograd sees it as a
            # function of the master through the fold's adjoint, without a launch of its own
            self.packed()
            w = _FoldView.apply(self.master, self) if torch.is_grad_enabled() else self._fold_buf
        else:
            w = self.derive()
        self._derived = (now, w)
        return w


def _direct_grad(p):
    """During a plain backward (no graph being built) weight / bias gradients are accumulated straight into the
    flat gradient buffer by the kernels (wgrad's accumulate mode, the fused bias-gradient atomics) instead of
    materialising a tensor for autograd's AccumulateGrad to add -- one tiny kernel per parameter saved."""
    return (not torch.is_grad_enabled()) and p.is_leaf and p.grad is not None and p.grad.is_contiguous()


_DEFERRED = None           # list collecting weight-gradient launches instead of issuing them (deferred_wgrads)


@contextlib.contextmanager
def deferred_wgrads(items):
    """Inside, the direct weight-gradient launches of a backward pass are not issued but appended to `items`
    (operands, bound gradient view, kernel parameters).  A weight gradient is a leaf of the backward dependency graph,
    so the caller can run the collected launches later, elsewhere (run_deferred_wgrads): the step moves D's
    weight gradients for the fakes off the generator phase's critical chain onto the stream that has gone idle."""
    global _DEFERRED
    old = _DEFERRED
    _DEFERRED = items
    try:
        yield items
    finally:
        _DEFERRED = old


def _master4(t, mode):
    """A 1x1x1 / KxK master of the padding-only fold seen as (Co,Ci,K,K) (the 3-D generator keeps its 1x1x1 convolutions'
    parameters 5-D, deepvoxels_generator.py:186: same memory)."""
    return t if mode == 0 or t.dim() == 4 else t.view(t.shape[0], t.shape[1], t.shape[-1], t.shape[-1])


def run_deferred_wgrads(items):
    kernels.conv2d_wgrad_batch([it[:7] for it in items])
    folds = []
    for it in items:
        if len(it) > 7:                       # derived layer: folded gradient -> master gradient (accumulating adjoint)
            master, (mode, cop, cip) = it[7]
            folds.append((it[2], _master4(master.grad, mode), mode, master.shape[0], master.shape[1], master.shape[-1], cop, cip,
                          True))
    kernels.fold_weight_multi(folds)          # ... of all derived layers of the pass: one launch


def _derived_deferrable(layer):
    """A DerivedConvLayer whose master parameter is bound to a flat gradient buffer, in a plain backward pass that is
    collecting its weight-gradient launches."""
    m = getattr(layer, "master", None)
    return (_DEFERRED is not None and m is not None and layer.fold is not None and not torch.is_grad_enabled()
            and m.is_leaf and m.grad is not None and m.grad.is_contiguous() and m.data_ptr() not in _FROZEN_PTRS)


def _wgrad_derived_deferred(x, dy, w, layer, ups):
    temp = torch.empty(tuple(w.shape), dtype=torch.float32, device=w.device)
    _DEFERRED.append((x.contiguous(), dy.contiguous(), temp, layer.K, layer.inv_c, bool(ups), False,
                      (layer.master, layer.fold)))


def _wgrad_into(x, dy, w, layer, ups):
    if _DEFERRED is not None:
        _DEFERRED.append((x.contiguous(), dy.contiguous(), w.grad, layer.K, layer.inv_c, bool(ups), True))
        return
    kernels.conv2d_wgrad(x.contiguous(), dy.contiguous(), layer.K, layer.inv_c, out=w.grad, accumulate=True,
                         upsample=bool(ups))


def _sum_pool2(x):
    """(B,2H,2W,C) -> (B,H,W,C): adjoint of nearest-2x upsampling."""
    B, H, W, C = x.shape
    return x.view(B, H // 2, 2, W // 2, 2, C).sum(dim=(2, 4))


def upsample2(x):
    """(B,H,W,C) -> (B,2H,2W,C) nearest (rescale.py:4-5)."""
    B, H, W, C = x.shape
    return x.view(B, H, 1, W, 1, C).expand(B, H, 2, W, 2, C).reshape(B, 2 * H, 2 * W, C)


class _ConvFprop(torch.autograd.Function):
    """y = conv(x, W) (+ residual).  mask_y (only inside the R1 double backward, see ResidualTie.fused_mask): the result
    times lrelu'(.) of that activation output, in the same epilogue."""

    @staticmethod
    def forward(ctx, x, w, layer, ups, residual=None, mask_y=None, operand_scale=None):
        """operand_scale (with mask_y): -> (y, y + operand_scale[b] * mask_y), the second tensor (not differentiable) being
        the injection operand of the convolution that reads mask_y in the forward pass (ResidualTie.c1_operand)."""
        ctx.layer, ctx.ups, ctx.masked = layer, ups, mask_y is not None
        ctx.save_for_backward(x, w)
        wf, _ = layer.packed()
        residual = residual.contiguous() if residual is not None else None
        if mask_y is not None:
            # (conv_dtype mxfp8: the masked result is the next convolution's input -- its fp8 copy leaves this epilogue)
            out = kernels.conv3x3_actgrad(x.contiguous(), wf, mask_y, residual=residual, operand_scale=operand_scale,
                                          emit_mx8=_MXFP8)
            if operand_scale is not None:
                ctx.mark_non_differentiable(out[1])
            return out
        return kernels.conv2d_fprop(x.contiguous(), wf, layer.K, layer.K, layer.pad, upsample=ups, residual=residual)

    @staticmethod
    def backward(ctx, dy, *unused):
        if ctx.masked:
            raise NotImplementedError("third-order derivatives through the conv engine are not supported")
        x, w = ctx.saved_tensors
        dy = dy.contiguous()
        dx = _ConvDgrad.apply(dy, w, ctx.layer, ctx.ups) if ctx.needs_input_grad[0] else None
        dw = None
        if ctx.needs_input_grad[1] and not _skip_grad_of(w):
            if _direct_grad(w):
                _wgrad_into(x, dy, w, ctx.layer, ctx.ups)
            elif _derived_deferrable(ctx.layer):
                _wgrad_derived_deferred(x, dy, w, ctx.layer, ctx.ups)
            else:
                dw = _ConvWgrad.apply(x, dy, ctx.layer, ctx.ups)
        return dx, dw, None, None, (dy if ctx.needs_input_grad[4:5] == (True,) else None), None, None


class _ConvDgrad(torch.autograd.Function):
    """dx = dgrad(dy, W) (+ resid).  `x_fwd` / `bias` (optional, not differentiated here) are the forward input and the
    bias of the convolution this node is the input gradient of: the adversarial injection needs them.  `tie` / `role`:
    the node belongs to a residual block (ResidualTie) as the input gradient of its shortcut conv (ROLE_SHORTCUT), main
    conv (ROLE_MAIN) or entry conv (ROLE_ENTRY); see ResidualTie for what the three share."""

    @staticmethod
    def forward(ctx, dy, w, layer, ups, x_fwd=None, bias=None, tie=None, role=0, resid=None, mask_y=None, mask_bias=None,
                mask_scale=None):
        """mask_y: the activation OUTPUT that fed the convolution (its forward input, when that is a leaky ReLU's result):
        dx comes out times lrelu'(mask_y), i.e. as the gradient of the pre-activation in front, and the (mask_scale-weighted)
        column sums of that go to mask_bias.grad -- the activation-gradient pass of the layer in front, in this epilogue."""
        ctx.layer, ctx.ups = layer, ups
        ctx.x_fwd, ctx.bias = x_fwd, bias
        ctx.tie, ctx.role = tie, role
        ctx.mask_y = mask_y
        ctx.save_for_backward(dy, w)
        _, wd = layer.packed()
        resid = resid.contiguous() if resid is not None else None
        if mask_y is not None:
            return kernels.conv3x3_actgrad(dy.contiguous(), wd, mask_y, residual=resid,
                                           bias_grad=mask_bias.grad if mask_bias is not None else None, row_scale=mask_scale,
                                           emit_mx8=_MXFP8)          # dz0 feeds the entry conv's input gradient
        return kernels.conv2d_dgrad(dy.contiguous(), wd, layer.K, layer.pad, sum_pool2=ups, residual=resid)

    @staticmethod
    def backward(ctx, ddx):
        dy, w = ctx.saved_tensors
        ddx = ddx.contiguous()
        tie, role = ctx.tie, ctx.role
        if ctx.mask_y is not None:
            # the node's output was mask * dgrad(dy, W): its adjoint starts with the same mask -- unless the node that
            # produced ddx (the entry conv's double-backward fprop) has applied it in its epilogue already
            if tie is not None and tie.dd_premasked:
                # the flag is the ADDRESS of the pre-masked tensor: a gradient that the engine has meanwhile summed with
                # another term (a first-order pass run without input_grads_only: c0's weight-gradient node then feeds this
                # one too) is a different tensor, partly masked, and cannot be repaired here
                expected, tie.dd_premasked = tie.dd_premasked, 0
                if expected != ddx.data_ptr():
                    raise RuntimeError("ResidualTie: the pre-masked double-backward term was combined with another gradient "
                                       "before it reached the fused node (run the first-order pass under input_grads_only())")
            else:
                ddx = _LreluGrad.apply(ddx, ctx.mask_y, ddx.shape[-1])
        g_dy = None
        if ctx.needs_input_grad[0]:
            if tie is not None and role == ROLE_MAIN:
                # d/d dz1 has two terms, c1(dd h0) here and c_sc(dd x) from the shortcut's node: one launch, the
                # shortcut's term riding in the epilogue, when that node ran first
                g_dy = _ConvFprop.apply(ddx, w, ctx.layer, ctx.ups, tie.take("g_sc", "main_seen"))
            elif tie is not None and role == ROLE_ENTRY and tie.fused_mask is not None and not ctx.ups and \
                    ctx.layer.K == 3 and ctx.layer.pad == 1 and tuple(tie.fused_mask.shape[:3]) == tuple(ddx.shape[:3]) and \
                    tie.fused_mask.shape[3] == w.shape[0] and kernels.conv3x3_actgrad_supported(*ddx.shape, w.shape[0]):
                # this node's dy is the OUTPUT of the main conv's fused (dgrad, activation-gradient) node, whose backward
                # masks what arrives with lrelu'(h0): applied here, in the epilogue of the conv that produces it
                if _INJECT is not None:
                    # ... and the same epilogue forms the main conv's injection operand dd h0 + s_b h0 (its axpy_rows pass)
                    g_dy, op2 = _ConvFprop.apply(ddx, w, ctx.layer, ctx.ups, None, tie.fused_mask, _INJECT)
                    tie.c1_operand = ((g_dy.data_ptr(), tie.fused_mask.data_ptr()), op2)
                else:
                    g_dy = _ConvFprop.apply(ddx, w, ctx.layer, ctx.ups, None, tie.fused_mask)
                tie.dd_premasked = g_dy.data_ptr()
            else:
                g_dy = _ConvFprop.apply(ddx, w, ctx.layer, ctx.ups)
                if tie is not None and role == ROLE_SHORTCUT and tie.give("g_sc", "main_seen", g_dy):
                    g_dy = None
        g_w = None
        if ctx.needs_input_grad[1] and not _skip_grad_of(w):
            operand = ddx
            if _INJECT is not None and ctx.x_fwd is not None:
                if ctx.ups or not _direct_grad(w):
                    raise RuntimeError("adversarial injection needs plain (non-upsampling) convs with bound gradients")
                x_fwd = ctx.x_fwd.detach().contiguous()
                key = (ddx.data_ptr(), x_fwd.data_ptr())
                if tie is not None and role in (ROLE_SHORTCUT, ROLE_ENTRY) and tie.operand is not None \
                        and tie.operand[0] == key:
                    operand, tie.operand = tie.operand[1], None      # the block's other entry conv already formed it
                elif tie is not None and role == ROLE_MAIN and tie.c1_operand is not None and tie.c1_operand[0] == key:
                    operand, tie.c1_operand = tie.c1_operand[1], None      # ... or the entry conv's double-backward epilogue
                else:
                    operand = kernels.axpy_rows(ddx, x_fwd, _INJECT)
                    if tie is not None and role in (ROLE_SHORTCUT, ROLE_ENTRY):
                        tie.operand = (key, operand)
                b = ctx.bias
                if b is not None and not _skip_grad_of(b):
                    if not _direct_grad(b):
                        raise RuntimeError("adversarial injection needs bias gradients bound to the flat buffer")
                    kernels.colsum(dy, out=b.grad, row_scale=_INJECT, rows_per_sample=dy.shape[1] * dy.shape[2])
            if _direct_grad(w):
                _wgrad_into(operand, dy, w, ctx.layer, ctx.ups)
            else:
                g_w = _ConvWgrad.apply(operand, dy, ctx.layer, ctx.ups)
        return (g_dy, g_w, None, None, None, None, None, None, (ddx if ctx.needs_input_grad[8:9] == (True,) else None),
                None, None, None)


class _ConvWgrad(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, dy, layer, ups):
        return kernels.conv2d_wgrad(x.contiguous(), dy.contiguous(), layer.K, layer.inv_c, upsample=bool(ups))

    @staticmethod
    def backward(ctx, ddw):
        raise NotImplementedError("third-order derivatives through the conv engine are not supported")


def conv(x, layer, upsample=False):
    """NHWC bf16 conv through the MFMA implicit-GEMM kernels (no bias / activation)."""
    return _ConvFprop.apply(x, layer.weight, layer, bool(upsample))


class _AdaIN(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, scale, shift):
        y, mean, rstd = kernels.adain_fwd(x.contiguous(), scale.contiguous(), shift.contiguous())
        ctx.save_for_backward(x, scale, mean, rstd)
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dy):
        x, scale, mean, rstd = ctx.saved_tensors
        dx, dscale, dshift = kernels.adain_bwd(x.contiguous(), dy.contiguous(), scale.contiguous(), mean, rstd)
        return dx, dscale, dshift


def adain(x, scale, shift):
    """normalization/adain.py:76-77 on NHWC bf16; scale/shift are (B,C) fp32."""
    return _AdaIN.apply(x, scale.float(), shift.float())


class _AdaINFused(torch.autograd.Function):
    """AdaIN whose scale and shift arrive as the two halves of one (B,2C) tensor (the output of the fused style
    affine): no slicing copies forward, one gradient tensor backward."""

    @staticmethod
    def forward(ctx, x, ss):
        ss = ss.contiguous()
        y, mean, rstd = kernels.adain_fwd(x.contiguous(), ss)
        ctx.save_for_backward(x, ss, mean, rstd)
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dy):
        x, ss, mean, rstd = ctx.saved_tensors
        dx, dss, _ = kernels.adain_bwd(x.contiguous(), dy.contiguous(), ss, mean, rstd, fused=True)
        return dx, dss


def adain_fused(x, scale_shift):
    return _AdaINFused.apply(x, scale_shift)


class StyleGroup:
    """[scale | shift] rows of several consecutive style blocks, produced by ONE linear from their common latent:
    `ss` (B, Wtot) with style block j in columns [offsets[j], offsets[j] + 2 C_j).  The AdaIN backward of every block
    writes its window of one shared gradient buffer; the block that ran FIRST in the forward (its backward runs last:
    every later block depends on its output) hands that buffer to autograd as the gradient of `ss`."""

    def __init__(self, ss, offsets, lives=None):
        # lives[j] (or None): the live channel count of block j when its tensor is zero padded beyond it (a 32-channel block
        # of the DeepVoxels generator on 64-channel tensors): its window is 2 lives[j] columns wide (kernels.adain_fwd c_live)
        self.ss, self.offsets, self.dss, self.lives = ss, offsets, None, lives


class _AdaINWindow(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, ss, group, j):
        y, mean, rstd = kernels.adain_fwd(x.contiguous(), ss, col_off=group.offsets[j],
                                          c_live=group.lives[j] if group.lives else None)
        ctx.group, ctx.j = group, j
        ctx.save_for_backward(x, ss, mean, rstd)
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dy):
        x, ss, mean, rstd = ctx.saved_tensors
        g = ctx.group
        if g.dss is None:
            g.dss = torch.empty_like(ss)
        dx, _, _ = kernels.adain_bwd(x.contiguous(), dy.contiguous(), ss, mean, rstd, fused=True,
                                     col_off=g.offsets[ctx.j], out=g.dss, c_live=g.lives[ctx.j] if g.lives else None)
        if ctx.j == 0:
            dss, g.dss = g.dss, None
            return dx, dss, None, None
        return dx, None, None, None


def adain_window(x, group, j):
    return _AdaINWindow.apply(x, group.ss, group, j)


class _ConvLreluAdaIN(torch.autograd.Function):
    """style(lrelu(conv(x, W) + b)) of a synthesis block (net.py:150-153 / 157-160) as one autograd node: bias and
    activation ride in the conv epilogue, and in the backward the AdaIN input gradient, the activation gradient and
    the bias gradient are ONE pass (rgbd_adain_bwd with lrelu_slope) -- the activation output is the AdaIN input, so
    the slope mask costs no extra read.  First order only (the generator is never differentiated twice)."""

    @staticmethod
    def forward(ctx, x, w, bias, ss, layer, ups, group, j):
        wf, _ = layer.packed()
        x = x.contiguous()
        Ho, Wo = (2 * x.shape[1], 2 * x.shape[2]) if ups else (x.shape[1], x.shape[2])
        if layer.K == 3 and layer.pad == 1 and \
                kernels.conv3x3_actgrad_supported(x.shape[0], Ho, Wo, x.shape[3], w.shape[0]):
            # images >= 16x16: the instance-norm statistics come out of the conv's epilogue (order-independent integer
            # sums), the AdaIN is its apply pass alone
            y, stats = kernels.conv2d_fprop_stats(x, wf, bias.contiguous(), upsample=ups, lrelu_channels=w.shape[0])
            out, mean, rstd = kernels.adain_apply_fixed(y, stats, ss, col_off=group.offsets[j], emit_mx8=_MXFP8)
        else:
            y = kernels.conv2d_fprop(x, wf, layer.K, layer.K, layer.pad, bias=bias.contiguous(), upsample=ups,
                                     lrelu_channels=w.shape[0])
            # (conv_dtype mxfp8: the AdaIN output is the next synthesis conv's input; its fp8 copy leaves the apply pass)
            out, mean, rstd = kernels.adain_fwd(y, ss, col_off=group.offsets[j], emit_mx8=_MXFP8 and y.shape[1] >= 8)
        ctx.layer, ctx.ups, ctx.group, ctx.j = layer, ups, group, j
        ctx.save_for_backward(x, w, y, bias, ss, mean, rstd)
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dout):
        x, w, y, bias, ss, mean, rstd = ctx.saved_tensors
        layer, ups, g = ctx.layer, ctx.ups, ctx.group
        if g.dss is None:
            g.dss = torch.empty_like(ss)
        want_b = ctx.needs_input_grad[2] and not _skip_grad_of(bias)
        fast_b = want_b and _direct_grad(bias)
        dz, _, _ = kernels.adain_bwd(y, dout.contiguous(), ss, mean, rstd, fused=True, col_off=g.offsets[ctx.j],
                                     out=g.dss, lrelu_slope=0.2, bias_grad=bias.grad if fast_b else None,
                                     emit_mx8=_MXFP8 and ctx.needs_input_grad[0] and y.shape[1] >= 16)   # dz feeds the dgrad
        dx = dw = db = dss = None
        if want_b and not fast_b:
            db = kernels.colsum(dz)
        if ctx.needs_input_grad[0]:
            _, wd = layer.packed()
            dx = kernels.conv2d_dgrad(dz, wd, layer.K, layer.pad, sum_pool2=ups)
        if ctx.needs_input_grad[1] and not _skip_grad_of(w):
            if _direct_grad(w):
                _wgrad_into(x, dz, w, layer, ups)
            else:
                dw = kernels.conv2d_wgrad(x, dz, layer.K, layer.inv_c, upsample=ups)
        if ctx.j == 0:
            dss, g.dss = g.dss, None
        return dx, dw, db, dss, None, None, None, None


def conv_bias_lrelu_adain(x, layer, bias, group, j, upsample=False):
    return _ConvLreluAdaIN.apply(x, layer.weight, bias, group.ss, layer, bool(upsample), group, j)


class _WarpLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, img, img_rot, coef, flags, lam, max_depth, min_depth, want_zp):
        ctx.cfg = (flags, lam, max_depth, min_depth)
        ctx.save_for_backward(img, img_rot, coef)
        if want_zp:
            loss, zp = kernels.warp_loss_fwd(img, img_rot, coef, flags, lam, max_depth, min_depth, want_zp=True)
        else:
            loss, zp = kernels.warp_loss_fwd(img, img_rot, coef, flags, lam, max_depth, min_depth), img.new_empty(0)
        ctx.mark_non_differentiable(zp)
        return loss.reshape(()), zp

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gl, _gzp):
        img, img_rot, coef = ctx.saved_tensors
        flags, lam, max_depth, min_depth = ctx.cfg
        gi, gr = kernels.warp_loss_bwd(img, img_rot, coef, flags, lam, max_depth, min_depth,
                                       gl.reshape(1).float().contiguous())
        return gi, gr, None, None, None, None, None, None


class _WarpLossNC(torch.autograd.Function):
    """Any channel count (last = depth), L1 or L2 criterion (kernels.warp_loss_nc_*)."""

    @staticmethod
    def forward(ctx, img, img_rot, coef, flags, l2, lam, max_depth, min_depth):
        ctx.cfg = (flags, l2, lam, max_depth, min_depth)
        ctx.save_for_backward(img, img_rot, coef)
        return kernels.warp_loss_nc_fwd(img, img_rot, coef, flags, l2, lam, max_depth, min_depth).reshape(())

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gl):
        img, img_rot, coef = ctx.saved_tensors
        flags, l2, lam, max_depth, min_depth = ctx.cfg
        gi, gr = kernels.warp_loss_nc_bwd(img, img_rot, coef, flags, l2, lam, max_depth, min_depth,
                                          gl.reshape(1).float().contiguous())
        return gi, gr, None, None, None, None, None, None


def warp_loss_nc(img, img_rot, coef, flags, norm_l2, lambda_geometric, max_depth=0.0, min_depth=0.0):
    return _WarpLossNC.apply(img.contiguous(), img_rot.contiguous(), coef, int(flags), bool(norm_l2), float(lambda_geometric),
                             float(max_depth), float(min_depth))


def warp_loss(img, img_rot, coef, flags, lambda_geometric, max_depth=0.0, min_depth=0.0, want_zp=False):
    """-> (loss, zp): the differentiable loss and, with want_zp, the projected points (2,b,S*S,3) of both directions (a
    by-product the reference returns as its second value, loss_functions.py:146; not differentiable here; else empty)."""
    return _WarpLoss.apply(img.contiguous(), img_rot.contiguous(), coef, int(flags), float(lambda_geometric),
                           float(max_depth), float(min_depth), bool(want_zp))


def avg_pool2_nhwc(x):
    """rescale.py:12-13 on NHWC bf16 (differentiable, twice)."""
    return _PoolMasked.apply(x, x.detach(), False)


class _MaskMul(torch.autograd.Function):
    """dy * mask with a constant mask: linear in dy, its own adjoint."""

    @staticmethod
    def forward(ctx, dy, mask):
        ctx.save_for_backward(mask)
        return dy * mask

    @staticmethod
    def backward(ctx, g):
        mask, = ctx.saved_tensors
        return _MaskMul.apply(g, mask), None


class _Lrelu(torch.autograd.Function):
    """Leaky ReLU (slope 0.2) on small fp32 tensors whose backward keeps no autograd edge to the input: the slope mask
    is a constant of the backward graph (torch's own double backward hands an all-zero gradient to the input, which
    makes a later backward walk the whole network behind it)."""

    @staticmethod
    def forward(ctx, x):
        y = F.leaky_relu(x, 0.2)
        ctx.save_for_backward(torch.where(x > 0, 1.0, 0.2).to(x.dtype))
        return y

    @staticmethod
    def backward(ctx, dy):
        mask, = ctx.saved_tensors
        return _MaskMul.apply(dy, mask)


def lrelu(x):
    return _Lrelu.apply(x)


class _LreluGrad(torch.autograd.Function):
    """dz = dy * lrelu'(.) evaluated from the activation OUTPUT y; linear in dy, so its own backward is itself.
    Callers pass y DETACHED: the mask is piecewise constant in y (zero derivative almost everywhere), and an autograd
    edge to y would make a later backward walk the whole recorded forward with all-zero gradients."""

    @staticmethod
    def forward(ctx, dy, y, act_channels, inject_bias=None):
        ctx.act_channels = act_channels
        ctx.save_for_backward(y)
        if inject_bias is not None:       # adversarial injection: d bias += sum_b s_b colsum_b(dz) in the same pass
            return kernels.lrelu_bwd(dy.contiguous(), y, act_channels, bias_grad=inject_bias.grad, row_scale=_INJECT)
        return kernels.lrelu_bwd(dy.contiguous(), y, act_channels)

    @staticmethod
    def backward(ctx, ddz):
        y, = ctx.saved_tensors
        return _LreluGrad.apply(ddz.contiguous(), y.detach(), ctx.act_channels), None, None, None


class _ColSum(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        ctx.shape = x.shape
        return kernels.colsum(x.contiguous())

    @staticmethod
    def backward(ctx, g):
        return g.to(torch.bfloat16).expand(ctx.shape)


class _UnpoolLreluGrad(torch.autograd.Function):
    """dz = 0.25 * upsample2(dp) * lrelu'(y): backward of (leaky ReLU -> 2x2 average pool) in one pass.  Linear in dp;
    its adjoint is _PoolMasked with the same mask."""

    @staticmethod
    def forward(ctx, dp, y, shape, use_mask, inject_bias=None, inject_bias2=None):
        ctx.use_mask = use_mask
        ctx.save_for_backward(y)
        if inject_bias is not None:
            return kernels.unpool2_lrelu_bwd(dp.contiguous(), y if use_mask else None, shape,
                                             bias_grad=inject_bias.grad, row_scale=_INJECT,
                                             bias_grad2=inject_bias2.grad if inject_bias2 is not None else None,
                                             emit_mx8=_MXFP8 and shape[1] >= 16)
        return kernels.unpool2_lrelu_bwd(dp.contiguous(), y if use_mask else None, shape, emit_mx8=_MXFP8 and shape[1] >= 16)

    @staticmethod
    def backward(ctx, ddz):
        y, = ctx.saved_tensors
        return _PoolMasked.apply(ddz.contiguous(), y.detach(), ctx.use_mask), None, None, None, None, None


class _PoolMasked(torch.autograd.Function):
    """out = 0.25 * sum_{2x2} x * lrelu'(y)  (use_mask False: plain 2x2 average pooling, rescale.py:12-13)."""

    @staticmethod
    def forward(ctx, x, y, use_mask):
        ctx.use_mask, ctx.shape = use_mask, tuple(x.shape)
        ctx.save_for_backward(y)
        return kernels.pool2_masked(x.contiguous(), y if use_mask else None)

    @staticmethod
    def backward(ctx, g):
        y, = ctx.saved_tensors
        return _UnpoolLreluGrad.apply(g.contiguous(), y.detach(), ctx.shape, ctx.use_mask), None, None


class _ConvBiasAct(torch.autograd.Function):
    """y = act(conv(x, W) + b + residual) [-> 2x2 average pool] with everything after the MFMA accumulation fused
    into the kernel's epilogue (net.py:144-159 c -> L.Bias -> leaky_relu; net.py:410-426 c1(h) + shortcut ->
    leaky_relu -> downscale2x).  The backward is assembled from differentiable pieces (lrelu-grad / unpool-lrelu-grad,
    dgrad, wgrad, column sum), so the R1 double backward goes through it."""

    @staticmethod
    def forward(ctx, x, w, bias, residual, layer, ups, act, pool, tie=None, role=0):
        wf, _ = layer.packed()
        x = x.contiguous()
        # conv_dtype mxfp8: an activation output feeds the next convolution (h0 -> c1; the pooled block output -> the next
        # block's c0 / c_sc), so its fp8 copy is written by this epilogue instead of by a quantiser pass
        y = kernels.conv2d_fprop(x, wf, layer.K, layer.K, layer.pad, bias=bias.contiguous(),
                                 residual=residual.contiguous() if residual is not None else None, upsample=ups,
                                 lrelu_channels=w.shape[0] if act else 0, avg_pool2=pool, emit_mx8=_MXFP8 and act)
        if pool:
            y, pooled = y
        ctx.layer, ctx.ups, ctx.act, ctx.pool = layer, ups, act, pool
        ctx.tie, ctx.role = tie, role
        if tie is not None and role == ROLE_ENTRY and act and not pool:
            tie.entry_bias, tie.entry_ptr = bias, y.data_ptr()       # see ResidualTie: premasked
        ctx.save_for_backward(x, w, y, bias)
        return pooled if pool else y

    @staticmethod
    def backward(ctx, dy):
        x, w, y, bias = ctx.saved_tensors
        layer, ups = ctx.layer, ctx.ups
        want_b = ctx.needs_input_grad[2] and not _skip_grad_of(bias)
        dx = dw = db = dres = None
        dy = dy.contiguous()
        fast_b = want_b and _direct_grad(bias)          # bias gradient rides along in the same pass
        # R1 first-order pass with the adversarial seeds known (adversarial_injection): the weighted bias sums are
        # taken here, fused with the activation gradient, instead of in a pass of their own during the double backward
        inj_b = bias if (_INJECT is not None and torch.is_grad_enabled() and ctx.needs_input_grad[2] and bias.is_leaf
                         and bias.grad is not None and bias.data_ptr() not in _FROZEN_PTRS) else None
        tie, tied_inject, role = ctx.tie, False, ctx.role
        if tie is not None and role == ROLE_SHORTCUT:
            # shortcut conv of a residual block: the main conv's activation-gradient pass (which ran just before, dz
            # being the gradient of both pre-activations) has already added the column sums to this bias
            if tie.done:
                tie.done = False
                tied_inject = inj_b is not None
                want_b, fast_b, inj_b = False, False, None
        if ctx.pool:
            tb = None
            if tie is not None and role == ROLE_MAIN and (fast_b or inj_b is not None) and tie.usable(not fast_b):
                tb = tie.bias
            if fast_b:
                dz = kernels.unpool2_lrelu_bwd(dy, y if ctx.act else None, tuple(y.shape), bias_grad=bias.grad,
                                               bias_grad2=tb.grad if tb is not None else None,
                                               emit_mx8=_MXFP8 and y.shape[1] >= 16)     # dz1 feeds both input gradients
                if tb is not None:
                    tie.done = True
            else:
                dz = _UnpoolLreluGrad.apply(dy, y.detach(), tuple(y.shape), ctx.act, inj_b,
                                            tb if inj_b is not None else None)
                if tb is not None and inj_b is not None:
                    tie.done = True
        elif ctx.act and tie is not None and role == ROLE_ENTRY and tie.premasked:
            # the main conv's backward produced dy in a fused (input gradient, activation gradient) launch: dy IS dz, and
            # this bias has its column sums (it decided with the same rules as above: fast_b or inj_b or no gradient).
            # Checked by address, like dd_premasked: h0 has one reader, so nothing can have been added on the way
            expected, tie.premasked = tie.premasked, 0
            if expected != dy.data_ptr():
                raise RuntimeError("ResidualTie: the fused activation gradient did not reach the entry conv unchanged")
            dz = dy
        elif ctx.act:
            dz = kernels.lrelu_bwd(dy, y, w.shape[0], bias_grad=bias.grad) if fast_b else \
                _LreluGrad.apply(dy, y.detach(), w.shape[0], inj_b)
        else:
            dz = dy
            if fast_b:
                kernels.colsum(dz, out=bias.grad)
            elif inj_b is not None:
                kernels.colsum(dz.detach(), out=bias.grad, row_scale=_INJECT, rows_per_sample=dz.shape[1] * dz.shape[2])
        if want_b and not fast_b:
            db = _ColSum.apply(dz)
        if ctx.needs_input_grad[0]:
            # bias=None when its injected gradient has been taken above
            inj_bias = None if (inj_b is not None or tied_inject) else bias
            if tie is not None and role == ROLE_ENTRY:
                # block input gradient = dgrad_c0(dz0) + dgrad_c_sc(dz1): the shortcut's term (its node ran first) is
                # added in this conv's epilogue instead of by the autograd engine
                dx = _ConvDgrad.apply(dz, w, layer, ups, x.detach(), inj_bias, tie, role, tie.take("dx_sc", "entry_seen"))
            elif tie is not None and role == ROLE_MAIN and _entry_fusable(tie, x, dz, layer, ups):
                # x is the entry conv's activation output h0 and nothing else reads it: the entry conv's activation
                # gradient (and bias gradient) ride in this input gradient's epilogue (ResidualTie: premasked)
                b0, scale0 = _entry_bias_mode(tie.entry_bias)
                h0 = x.detach()
                dx = _ConvDgrad.apply(dz, w, layer, ups, h0, inj_bias, tie, role, None, h0, b0, scale0)
                tie.premasked, tie.fused_mask = dx.data_ptr(), h0
            else:
                if tie is not None and role == ROLE_MAIN:      # unfused this time: nothing stale from an earlier pass
                    tie.premasked, tie.dd_premasked, tie.fused_mask = 0, 0, None
                dx = _ConvDgrad.apply(dz, w, layer, ups, x.detach(), inj_bias, tie, role)
                if tie is not None and role == ROLE_SHORTCUT and tie.give("dx_sc", "entry_seen", dx):
                    dx = None
        if ctx.needs_input_grad[1] and not _skip_grad_of(w):
            if _direct_grad(w):
                _wgrad_into(x, dz, w, layer, ups)
            elif _derived_deferrable(layer):
                _wgrad_derived_deferred(x, dz, w, layer, ups)
            else:
                dw = _ConvWgrad.apply(x, dz, layer, ups)
        if ctx.needs_input_grad[3]:
            dres = dz
        return dx, dw, db, dres, None, None, None, None, None, None


ROLE_SHORTCUT, ROLE_MAIN, ROLE_ENTRY = 1, 2, 3


def _entry_bias_mode(b0):
    """How the entry conv's backward would take its bias gradient (the rules of _ConvBiasAct.backward, evaluated on its
    bias): (bias whose .grad receives the column sums or None, per-sample weights or None); False when it would return the
    gradient through autograd (_ColSum), which a fused launch cannot."""
    want = b0.requires_grad and not _skip_grad_of(b0)
    fast = want and _direct_grad(b0)
    if want and not fast:
        return False
    inject = (_INJECT is not None and torch.is_grad_enabled() and b0.requires_grad and b0.is_leaf
              and b0.grad is not None and b0.data_ptr() not in _FROZEN_PTRS)
    if fast:
        return b0, None
    return (b0, _INJECT) if inject else (None, None)


def _entry_fusable(tie, x, dz, layer, ups):
    return (tie.entry_bias is not None and tie.entry_ptr == x.data_ptr() and not ups
            and layer.K == 3 and layer.pad == 1
            and _entry_bias_mode(tie.entry_bias) is not False
            and kernels.conv3x3_actgrad_supported(dz.shape[0], dz.shape[1], dz.shape[2], dz.shape[3], x.shape[3]))


class ResidualTie:
    """What the three convs of a residual block (net.py:408-416: h = lrelu(c0 x); y = lrelu(c1 h + c_sc x)) share in the
    backward passes, so that sums the autograd engine would form with add kernels ride in conv epilogues instead:

    * bias: the gradient w.r.t. the sum c1 h + c_sc x is the gradient of BOTH biases, so the main conv's fused
      activation-gradient pass adds its column sums to both buffers and the shortcut conv skips its own reduction;
    * dx_sc: the block input gradient dgrad_c0(dz0) + dgrad_c_sc(dz1) -- the shortcut's node (created later in the
      forward, so run earlier by the engine) parks its term here and the entry conv adds it as an epilogue residual;
    * g_sc: in the R1 double backward d/d dz1 = c1(dd h0) + c_sc(dd x), same arrangement one order up;
    * operand: the injection operand dd x + s_b x is the same tensor for c0 and c_sc, formed once;
    * premasked: h0 = lrelu(c0 x) is read by c1 alone, so dz0 = dgrad_c1(dz1) * lrelu'(h0) and c0's bias gradient are
      taken in the epilogue of c1's input gradient (rgbd_conv3x3_actgrad_bf16; entry_bias / entry_ptr are left by c0's
      forward) and c0's backward skips its activation-gradient pass; fused_mask / dd_premasked: the same one order up --
      in the R1 double backward c0's fprop of dd x applies lrelu'(h0) in its epilogue and the fused node's backward
      skips its own mask; c1_operand: that epilogue's second output dd h0 + s_b h0, c1's injection operand.

    Every hand-over has a fallback: if the consumer ran first it flags that, and the producer then returns its term to
    autograd the ordinary way (the engine adds)."""

    def __init__(self, bias):
        self.bias, self.done = bias, False
        self.dx_sc = self.g_sc = self.operand = None
        self.entry_seen = self.main_seen = False
        self.entry_bias = self.entry_ptr = self.fused_mask = self.c1_operand = None
        self.premasked = self.dd_premasked = 0          # addresses of the pre-masked tensors (0: none outstanding)

    def usable(self, inject):
        """Mirrors the conditions under which _ConvBiasAct.backward takes its own bias sums in the fused pass."""
        b = self.bias
        if inject:
            return b.requires_grad and b.is_leaf and b.grad is not None and b.data_ptr() not in _FROZEN_PTRS
        return b.requires_grad and _direct_grad(b) and not _skip_grad_of(b)

    def give(self, slot, seen_flag, value):
        """Producer side: park `value` for the consumer (True), unless the consumer already ran (False)."""
        if getattr(self, seen_flag):
            setattr(self, seen_flag, False)
            return False
        setattr(self, slot, value)
        return True

    def take(self, slot, seen_flag):
        """Consumer side: the parked value, or None (and remember that the consumer has run)."""
        value = getattr(self, slot)
        setattr(self, slot, None)
        if value is None:
            setattr(self, seen_flag, True)
        return value


BiasTie = ResidualTie


def conv_bias_lrelu(x, layer, bias, upsample=False, residual=None, pool=False, residual_tie=None, entry_tie=None):
    """residual_tie: this is the main conv (c1) of a residual block; entry_tie: its entry conv (c0)."""
    tie, role = (residual_tie, ROLE_MAIN) if residual_tie is not None else (entry_tie, ROLE_ENTRY if entry_tie else 0)
    return _ConvBiasAct.apply(x, layer.weight, bias, residual, layer, bool(upsample), True, bool(pool), tie, role)


def conv_bias(x, layer, bias, upsample=False, residual=None, tie=None):
    """tie: this is the shortcut conv (c_sc) of a residual block."""
    return _ConvBiasAct.apply(x, layer.weight, bias, residual, layer, bool(upsample), False, False, tie,
                              ROLE_SHORTCUT if tie is not None else 0)


# ---- 1x1 convolutions between NCHW fp32 image planes and NHWC bf16 features (fromRGB / toRGB)
class _FromPlanes(torch.autograd.Function):
    """y[b,p,c] = act(s * sum_k w[c][k] x[b,k,p] + bias[c])   (Discriminator.ins, net.py:449-455, 485,493-494)."""

    @staticmethod
    def forward(ctx, x, w, bias, scale, act):
        x = x.contiguous()
        y = kernels.from_planes(x, w.contiguous(), bias.contiguous() if bias is not None else None, scale, act)
        ctx.scale, ctx.act, ctx.has_bias = scale, act, bias is not None
        ctx.bias_ref = bias
        ctx.save_for_backward(x, w, y)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, y = ctx.saved_tensors
        dz = _LreluGrad.apply(dy.contiguous(), y.detach(), y.shape[-1]) if ctx.act else dy.contiguous()
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = _ToPlanes.apply(dz, w.t(), None, ctx.scale, x.detach(), ctx.bias_ref)
        if (ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2])) and not _skip_grad_of(w):
            o, ts = _PlanesOuter.apply(dz, x, ctx.has_bias)
            dw = o.t() * ctx.scale
            db = ts if ctx.has_bias else None
        return dx, dw, db, None, None


class _ToPlanes(torch.autograd.Function):
    """out[b,k,p] = s * sum_c w[k][c] h[b,p,c] + bias[k]   (StyleGenerator.outs, net.py:186-191,270,289-290).
    `inj_x` / `inj_bias`: when this node is the input gradient of a fromRGB layer, its forward image and bias (for the
    adversarial injection, see adversarial_injection)."""

    @staticmethod
    def forward(ctx, h, w, bias, scale, inj_x=None, inj_bias=None):
        h = h.contiguous()
        ctx.scale, ctx.has_bias = scale, bias is not None
        ctx.bias_ref = bias
        ctx.inj_x, ctx.inj_bias = inj_x, inj_bias
        ctx.save_for_backward(h, w)
        return kernels.to_planes(h, w.contiguous(), bias.contiguous() if bias is not None else None, scale)

    @staticmethod
    def backward(ctx, dout):
        h, w = ctx.saved_tensors
        dout = dout.contiguous()
        dh = dw = db = None
        if ctx.needs_input_grad[0]:
            dh = _FromPlanes.apply(dout, w.t(), None, ctx.scale, False)
        if ctx.needs_input_grad[1] and not _skip_grad_of(w):
            operand = dout
            if _INJECT is not None and ctx.inj_x is not None:
                operand = kernels.axpy_rows_f32(dout, ctx.inj_x.detach().contiguous(), _INJECT)
                b = ctx.inj_bias
                if b is not None and not _skip_grad_of(b):
                    if not _direct_grad(b):
                        raise RuntimeError("adversarial injection needs bias gradients bound to the flat buffer")
                    kernels.colsum(h, out=b.grad, row_scale=_INJECT, rows_per_sample=h.shape[1] * h.shape[2])
            # the bias gradient sum_{b,p} dout rides in the same kernel (straight into the bound gradient buffer): a
            # torch reduction here is a multi-block kernel with a memset-initialised semaphore, and memset nodes are
            # unreliable on replays of a captured HIP graph (ROCm 7.2) -- found by the graph == eager step test
            bias = ctx.bias_ref
            want_b = ctx.has_bias and ctx.needs_input_grad[2] and not _skip_grad_of(bias)
            psum = None
            if want_b and _INJECT is None:
                psum = bias.grad if _direct_grad(bias) else torch.zeros(dout.shape[1], dtype=torch.float32,
                                                                        device=dout.device)
            o, _ = _PlanesOuter.apply(h, operand.contiguous(), False, psum)
            dw = o * ctx.scale
            if want_b and psum is not None and not _direct_grad(bias):
                db = psum
            elif want_b and psum is None:
                db = dout.sum(dim=(0, 2, 3))
        elif ctx.has_bias and ctx.needs_input_grad[2] and not _skip_grad_of(ctx.bias_ref):
            db = dout.sum(dim=(0, 2, 3))
        return dh, dw, db, None, None, None


class _PlanesOuter(torch.autograd.Function):
    @staticmethod
    def forward(ctx, t, planes, want_tsum, psum=None):
        o, ts = kernels.planes_outer(t.contiguous(), planes.contiguous(), want_tsum, psum)
        if ts is None:
            ts = o.new_empty(())          # placeholder output, never read (no fill launch)
        return o, ts

    @staticmethod
    def backward(ctx, go, gts):
        raise NotImplementedError("third-order derivatives through the 1x1 plane kernels are not supported")


def from_planes(x, w, bias, scale, act=True):
    return _FromPlanes.apply(x, w, bias, float(scale), bool(act))


def to_planes(h, w, bias, scale):
    return _ToPlanes.apply(h, w, bias, float(scale))


class _LinearAct(torch.autograd.Function):
    """y = act(c * x W^T + b) for a handful of rows (first-order only: generator side).  Weight and bias gradients are
    accumulated straight into the flat gradient buffer when one is bound.  Rows are processed in groups of 64."""

    @staticmethod
    def forward(ctx, x, w, bias, c, act):
        x = x.contiguous()
        w = w.contiguous()
        b = bias.contiguous() if bias is not None else None
        y = torch.cat([kernels.linear_fwd(x[i:i + 64], w, b, c, act) for i in range(0, x.shape[0], 64)]) \
            if x.shape[0] > 64 else kernels.linear_fwd(x, w, b, c, act)
        ctx.c, ctx.act = c, act
        ctx.save_for_backward(x, w, bias, y)
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dy):
        x, w, bias, y = ctx.saved_tensors
        dy = dy.contiguous()
        need_x, need_w = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        need_b = bias is not None and ctx.needs_input_grad[2]
        direct = need_w and _direct_grad(w) and (not need_b or _direct_grad(bias))
        dw = w.grad if direct else (torch.zeros_like(w) if need_w else None)
        db = (bias.grad if direct else torch.zeros_like(bias)) if need_b else None
        dxs = [kernels.linear_bwd(dy[i:i + 64], y[i:i + 64], x[i:i + 64], w, ctx.c, ctx.act, want_dx=need_x, dw=dw, db=db)
               for i in range(0, x.shape[0], 64)]
        dx = (dxs[0] if len(dxs) == 1 else torch.cat(dxs)) if need_x else None
        if direct:
            return dx, None, None, None, None
        return dx, dw, db, None, None


def linear_act(x, w, bias, c, act=True):
    """Equalized-LR linear (pggan.py:39-50) + optional leaky ReLU on a small batch of rows, one HIP launch."""
    return _LinearAct.apply(x, w, bias, float(c), bool(act))


class _MlpChain(torch.autograd.Function):
    """h <- lrelu(c * h W_l^T + b_l) for l = 0..L-1 as ONE launch forward and two backward (kernels.mlp_fwd / mlp_bwd): the
    mapping network (net.py:58-62).  First-order only (generator side); weight and bias gradients are accumulated straight
    into the flat gradient buffer when one is bound, like _LinearAct's."""

    @staticmethod
    def forward(ctx, x, c, *params):
        L = len(params) // 2
        ws, bs = [p.detach() for p in params[:L]], [p.detach() for p in params[L:]]
        x = x.contiguous()
        acts = kernels.mlp_fwd(x, ws, bs, c)
        ctx.c, ctx.L = c, L
        ctx.save_for_backward(x, acts, *params)
        return acts[L - 1]

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dy):
        x, acts = ctx.saved_tensors[:2]
        params = ctx.saved_tensors[2:]
        L = ctx.L
        ws, bs = list(params[:L]), list(params[L:])
        need = [ctx.needs_input_grad[2 + i] and not _skip_grad_of(params[i]) for i in range(2 * L)]
        direct = all(_direct_grad(p) for p, n in zip(params, need) if n)
        if direct:
            dws = [w.grad if n else None for w, n in zip(ws, need[:L])]
            dbs = [b.grad if n else None for b, n in zip(bs, need[L:])]
        else:
            dws = [torch.zeros_like(w) if n else None for w, n in zip(ws, need[:L])]
            dbs = [torch.zeros_like(b) if n else None for b, n in zip(bs, need[L:])]
        any_w = any(t is not None for t in dws + dbs)
        dx = kernels.mlp_bwd(dy.contiguous(), x, acts, [w.detach() for w in ws], ctx.c, dws if any_w else None,
                             dbs if any_w else None)
        grads = (None,) * (2 * L) if direct else tuple(dws + dbs)
        return (dx if ctx.needs_input_grad[0] else None, None) + grads


def mlp_chain(x, weights, biases, c):
    """The mapping network's eight layers (net.py:58-62) as one fused chain; falls back to one launch per layer for shapes the
    fused kernel does not cover (C other than 256 / 512)."""
    if kernels.mlp_supported(x.shape[0], x.shape[1], len(weights)) and all(w.shape == (x.shape[1], x.shape[1]) for w in weights):
        return _MlpChain.apply(x, float(c), *weights, *biases)
    h = x
    for w, b in zip(weights, biases):
        h = linear_act(h, w, b, c, act=True)
    return h


class _Dense(torch.autograd.Function):
    """y = act(c * x W^T + b) on <= 64 fp32 rows, differentiable TWICE (the discriminator's dense tail sits under the R1
    penalty, updater.py:414-422): the backward is built from _DenseDgrad, whose own backward is again two kernels.
    `w` is the master parameter in ANY shape with N leading rows (the 4x4-valid conv weight (co,ci,4,4) is used as a
    (co, ci*16) matrix without a copy); weight / bias gradients go straight into the bound flat gradient buffer."""

    @staticmethod
    def forward(ctx, x, w, bias, c, act):
        x = x.contiguous()
        w2 = w.detach().reshape(w.shape[0], -1)
        y = kernels.linear_fwd(x, w2, bias.detach() if bias is not None else None, c, act)
        ctx.c, ctx.act = c, act
        ctx.save_for_backward(x, w, bias, y)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, bias, y = ctx.saved_tensors
        dy = dy.contiguous()
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = _DenseDgrad.apply(dy, y.detach(), w, ctx.c, ctx.act)
        if ctx.needs_input_grad[1] and not _skip_grad_of(w):
            w2 = w.detach().reshape(w.shape[0], -1)
            want_b = bias is not None and ctx.needs_input_grad[2]
            if _direct_grad(w) and (not want_b or _direct_grad(bias)):
                kernels.linear_bwd(dy.detach(), y, x, w2, ctx.c, ctx.act, want_dx=False, dw=w.grad.view_as(w2),
                                   db=bias.grad if want_b else None)
            else:
                dw2 = torch.zeros_like(w2)
                db = torch.zeros_like(bias) if want_b else None
                kernels.linear_bwd(dy.detach(), y, x, w2, ctx.c, ctx.act, want_dx=False, dw=dw2, db=db)
                dw = dw2.view_as(w)
        return dx, dw, db, None, None


class _DenseDgrad(torch.autograd.Function):
    """dx = c * (dy * act'(y)) W.  Linear in dy: its backward is (c * ddx W^T) * act'(y) for dy and
    c * (dy * act'(y))^T ddx for W (third order is not supported)."""

    @staticmethod
    def forward(ctx, dy, y, w, c, act):
        dy = dy.contiguous()
        w2 = w.detach().reshape(w.shape[0], -1)
        ctx.c, ctx.act = c, act
        ctx.save_for_backward(dy, y, w)
        return kernels.linear_bwd(dy, y, None, w2, c, act, want_dx=True)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, ddx):
        dy, y, w = ctx.saved_tensors
        ddx = ddx.contiguous()
        w2 = w.reshape(w.shape[0], -1)
        g_dy = g_w = None
        if ctx.needs_input_grad[0]:
            g_dy = kernels.linear_fwd_masked(ddx, w2, y, ctx.c) if ctx.act else kernels.linear_fwd(ddx, w2, None, ctx.c, False)
        if ctx.needs_input_grad[2] and not _skip_grad_of(w):
            if _direct_grad(w):
                kernels.linear_bwd(dy, y, ddx, w2, ctx.c, ctx.act, want_dx=False, dw=w.grad.view_as(w2))
            else:
                dw2 = torch.zeros_like(w2)
                kernels.linear_bwd(dy, y, ddx, w2, ctx.c, ctx.act, want_dx=False, dw=dw2)
                g_w = dw2.view_as(w)
        return g_dy, None, g_w, None, None


def dense(x, w, bias, c, act=True):
    """Equalized-LR linear (pggan.py:39-50) [+ leaky ReLU] on <= 64 rows through the HIP linear kernels, twice
    differentiable."""
    return _Dense.apply(x, w, bias, float(c), bool(act))


class _NhwcToRows(torch.autograd.Function):
    """(B,H,W,C) bf16 -> (B, C*H*W) fp32 in (c,h,w) order (L.Linear's flattening of an NCHW array, net.py:372-377)."""

    @staticmethod
    def forward(ctx, h):
        ctx.shape = tuple(h.shape)
        return kernels.nhwc_to_rows(h.contiguous())

    @staticmethod
    def backward(ctx, g):
        return _RowsToNhwc.apply(g, ctx.shape)


class _RowsToNhwc(torch.autograd.Function):
    @staticmethod
    def forward(ctx, rows, shape):
        return kernels.rows_to_nhwc(rows.contiguous(), shape[1], shape[2], shape[3])

    @staticmethod
    def backward(ctx, g):
        return _NhwcToRows.apply(g), None


def nhwc_to_rows(h):
    return _NhwcToRows.apply(h)


def rows_to_nhwc(rows, H, W, C):
    """(B, C*H*W) fp32 rows in (c,h,w) order -> (B,H,W,C) bf16 (DCGAN input layer, net.py:713-719)."""
    return _RowsToNhwc.apply(rows, (rows.shape[0], H, W, C))


class _ConstInput(torch.autograd.Function):
    """SynthesisBlock 0 (net.py:130-153): lrelu(W + b0) broadcast over the batch as (B,4,4,C) bf16, one launch; the
    backward adds the batch-summed, masked gradient to W's and b0's bound gradient buffers (first order only)."""

    @staticmethod
    def forward(ctx, w, bias, B):
        ctx.save_for_backward(w, bias)
        return kernels.const_input_fwd(w.detach().contiguous(), bias.detach().contiguous(), B)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dh):
        w, bias = ctx.saved_tensors
        want_w = ctx.needs_input_grad[0] and not _skip_grad_of(w)
        want_b = ctx.needs_input_grad[1] and not _skip_grad_of(bias)
        direct = (not want_w or _direct_grad(w)) and (not want_b or _direct_grad(bias))
        dw = (w.grad if direct else torch.zeros_like(w)) if want_w else None
        db = (bias.grad if direct else torch.zeros_like(bias)) if want_b else None
        if want_w or want_b:
            kernels.const_input_bwd(dh.contiguous(), w.detach().contiguous(), bias.detach().contiguous(), dw, db)
        return (None, None, None) if direct else (dw, db, None)


def const_input(w, bias, B):
    return _ConstInput.apply(w, bias, int(B))


class _R1Penalty(torch.autograd.Function):
    """lambda * mean_b ||g_b||^2 (updater.py:416-418 with loss_functions.py:7-8: sqrt, then squared again) on the image
    gradient g of the R1 first-order pass; d/dg = 2 lambda / B * g feeds the double backward."""

    @staticmethod
    def forward(ctx, g, coef):
        g = g.contiguous()
        ctx.coef = coef
        ctx.save_for_backward(g)
        return kernels.r1_penalty_fwd(g, coef).reshape(())

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gl):
        g, = ctx.saved_tensors
        return kernels.scale_by_scalar(g, gl.reshape(1).float().contiguous(), 2.0 * ctx.coef / g.shape[0]), None


class _SoftplusMean(torch.autograd.Function):
    """mean softplus(sign y) * sigmoid(sign y)^gamma (loss_func_dcgan_gen / one term of loss_func_dcgan_dis,
    loss_functions.py:15-31) with its derivative from the same launch; first order (the logits' own backward goes on from dy)."""

    @staticmethod
    def forward(ctx, y, sign, gamma):
        loss, dy = kernels.softplus_mean(y, sign, gamma)
        ctx.save_for_backward(dy)
        return loss.reshape(())

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gl):
        dy, = ctx.saved_tensors
        return kernels.scale_by_scalar(dy, gl.reshape(1).float().contiguous(), 1.0), None, None


def softplus_mean(y, sign, gamma=0.0):
    return _SoftplusMean.apply(y, float(sign), float(gamma or 0.0))


def r1_penalty(grad_x, lambda_gp):
    return _R1Penalty.apply(grad_x, float(lambda_gp))


class _FadePlanes(torch.autograd.Function):
    """(1-a) * upscale2x(lo) + a * hi on NCHW fp32 planes (generator fade-in, net.py:283-290); first order."""

    @staticmethod
    def forward(ctx, lo, hi, alpha):
        ctx.alpha = alpha
        return kernels.fade_planes_fwd(lo.contiguous(), hi.contiguous(), alpha)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dout):
        dlo, dhi = kernels.fade_planes_bwd(dout.contiguous(), ctx.alpha, ctx.needs_input_grad[0], ctx.needs_input_grad[1])
        return dlo, dhi, None


def fade_planes(lo, hi, alpha):
    return _FadePlanes.apply(lo, hi, alpha)


class _Lerp(torch.autograd.Function):
    """(1-a) p + a q on NHWC bf16 (discriminator fade-in, net.py:490-497); linear, so with _LerpSplit closed under
    differentiation (R1 goes through it twice)."""

    @staticmethod
    def forward(ctx, p, q, alpha):
        ctx.alpha = alpha
        return kernels.lerp_bf16(p.contiguous(), q.contiguous(), alpha)

    @staticmethod
    def backward(ctx, g):
        gp, gq = _LerpSplit.apply(g, ctx.alpha)
        return gp, gq, None


class _LerpSplit(torch.autograd.Function):
    @staticmethod
    def forward(ctx, g, alpha):
        ctx.alpha = alpha
        return kernels.lerp_split_bf16(g.contiguous(), alpha)

    @staticmethod
    def backward(ctx, gp, gq):
        return _Lerp.apply(gp, gq, ctx.alpha), None


def lerp(p, q, alpha):
    return _Lerp.apply(p, q, alpha)


class _PoolPlanes(torch.autograd.Function):
    """2x2 average pooling of NCHW fp32 planes (downscale2x of the image in the discriminator's fade-in path,
    net.py:491) and its adjoint: each is the other's derivative."""

    @staticmethod
    def forward(ctx, x, adjoint):
        ctx.adjoint = adjoint
        return kernels.pool2_planes(x.contiguous(), adjoint)

    @staticmethod
    def backward(ctx, g):
        return _PoolPlanes.apply(g, not ctx.adjoint), None


def avg_pool2_planes(x):
    return _PoolPlanes.apply(x, False)


class _L2Norm(torch.autograd.Function):
    """F.normalize over channels (DCGANBlock, net.py:621-648) on NHWC bf16, first order."""

    @staticmethod
    def forward(ctx, x):
        x = x.contiguous()
        ctx.save_for_backward(x)
        return kernels.l2norm_fwd(x)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dy):
        x, = ctx.saved_tensors
        return kernels.l2norm_bwd(x, dy.contiguous())


def l2_normalize(x):
    return _L2Norm.apply(x)


class _Blur(torch.autograd.Function):
    """rescale.py:20-25 on NHWC bf16; linear and symmetric, so every order of its derivative is the same kernel family:
    mode 0 is self-adjoint, modes 1 (blur after the nearest upsample) and 2 (2x2 sums of the blur) are adjoint."""

    @staticmethod
    def forward(ctx, x, mode):
        ctx.mode = mode
        return kernels.blur3x3(x.contiguous(), mode)

    @staticmethod
    def backward(ctx, g):
        return _Blur.apply(g.contiguous(), {0: 0, 1: 2, 2: 1}[ctx.mode]), None


def blur(x):
    return _Blur.apply(x, 0)


def upsample_blur(x):
    return _Blur.apply(x, 1)


class _PixelNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = x.contiguous()
        ctx.save_for_backward(x)
        return kernels.pixelnorm(x)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dy):
        x, = ctx.saved_tensors
        return kernels.pixelnorm(x, dy.contiguous())


def pixel_norm(x):
    """pggan.py:7-10 (feature_vector_normalization) on (M,C) fp32 rows."""
    return _PixelNorm.apply(x)


class _DepthHead(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = x.contiguous()
        y = kernels.depth_head_fwd(x)
        ctx.save_for_backward(x, y)
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dy):
        x, y = ctx.saved_tensors
        return kernels.depth_head_bwd(x, y, dy.contiguous())


def depth_head(x):
    """net.py:296: (B,4,H,W) fp32 planes -> RGB unchanged, depth = 1 / (softplus(x3) + 1e-4)."""
    return _DepthHead.apply(x)


# ---- DeepVoxels layout folds (deepvoxels_generator.py): one launch forward, one backward; each op's backward is its adjoint
class _FoldDepthTaps(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, up):
        ctx.shape, ctx.up = tuple(x.shape), bool(up)
        return kernels.fold_depth_taps(x.contiguous(), ctx.up)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dy):
        return kernels.fold_depth_taps(dy.contiguous(), ctx.up, adjoint_shape=ctx.shape), None


class _Fold4x4s2(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        ctx.shape = tuple(x.shape)
        return kernels.fold_4x4s2(x.contiguous())

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dy):
        return kernels.fold_4x4s2(dy.contiguous(), adjoint_shape=ctx.shape)


class _PadLast(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, n):
        ctx.c0 = x.shape[-1]
        return kernels.pad_last(x.contiguous(), n)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dy):
        return kernels.pad_last(dy.contiguous(), ctx.c0), None


def fold_depth_taps(x, upsample_depth=False):
    """(B,D0,H,W,C) -> (B*D,H,W,3C): every depth slice next to its two zero-padded neighbours (behind a 2x depth repeat)."""
    return _FoldDepthTaps.apply(x, upsample_depth)


def fold_4x4s2(x):
    """(B,H,W,C) -> (B,H/2,W/2,16C): channel (ky*4 + kx)*C + c holds x_pad[2i+ky, 2j+kx, c] (pad 1)."""
    return _Fold4x4s2.apply(x)


def pad_last(x, n):
    """Zero-pad the last dimension to n entries."""
    return x if x.shape[-1] == n else _PadLast.apply(x, n)


class _FoldWeight(torch.autograd.Function):
    @staticmethod
    def forward(ctx, W, mode, Cop, Cip):
        ctx.args = (mode, W.shape[0], W.shape[1], W.shape[-1], Cop, Cip)
        ctx.set_materialize_grads(False)       # a deferred weight gradient reaches the master around autograd: no zeros here
        return kernels.fold_weight(W.contiguous(), *ctx.args)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        if g is None:
            return None, None, None, None
        return kernels.fold_weight(g.contiguous(), *ctx.args, adjoint=True), None, None, None


class _FoldView(torch.autograd.Function):
    """The persistent folded buffer of a grouped DerivedConvLayer as a differentiable function of its master (no launch:
    DerivedPackGroup.repack has already written it)."""

    @staticmethod
    def forward(ctx, W, layer):
        mode, cop, cip = layer.fold
        ctx.args = (mode, W.shape[0], W.shape[1], W.shape[-1], cop, cip)
        ctx.master_shape = tuple(W.shape)
        ctx.set_materialize_grads(False)
        return layer._fold_buf.view(layer._fold_buf.shape)     # (a fresh alias: the buffer itself is not an input)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        if g is None:
            return None, None
        return kernels.fold_weight(g.contiguous(), *ctx.args, adjoint=True).view(ctx.master_shape), None


def fold_weight(W, mode, Cop, Cip):
    """Master conv parameter -> the weight the 2-D conv engine packs (kernels.fold_weight), differentiable."""
    return _FoldWeight.apply(W, mode, Cop, Cip)
