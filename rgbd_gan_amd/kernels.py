"""Thin, typed wrappers: torch device tensors in, C-ABI calls out.  No arithmetic happens here."""
import contextlib
import ctypes
import os

import numpy as np

import torch

from . import _lib

BF16 = torch.bfloat16
F32 = torch.float32

# ---- optional per-launch timing (bench.py's roofline leg): HIP events on the launch stream around each kernel
_PROFILE = None


class launch_profile:
    """with kernels.launch_profile() as prof: ...  -> prof.summary() = {kernel: (launches, seconds, flops, bytes)}.
    Events are recorded on torch's current stream, which is the stream every kernel here is launched on."""

    def __enter__(self):
        global _PROFILE
        self.records = []
        # an event pair around nothing still measures a few microseconds (the event packets themselves): calibrate
        # that and take it off every launch, so the averages line up with a profiler's kernel durations
        pairs = []
        for _ in range(32):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); e1.record()
            pairs.append((e0, e1))
        torch.cuda.synchronize()
        gaps = sorted(a.elapsed_time(b) for a, b in pairs)
        self.overhead_s = gaps[len(gaps) // 2] * 1e-3
        _PROFILE = self.records
        return self

    def __exit__(self, *exc):
        global _PROFILE
        _PROFILE = None

    def summary(self):
        """An event pair brackets the host's launch call, so now and then one of them spans a host stall that has nothing to do with
        the kernel (observed: ONE pair of 50-64 ms among ~100 launches of a 60-700 us kernel, twice in round 6; the same launches
        under rocprofv3 show no such kernel).  A launch that reads more than 8x the median of the launches of the SAME kernel on
        the SAME shape (equal flops and bytes) is counted at that median; `self.outliers` says how many per kernel."""
        torch.cuda.synchronize()
        groups = {}
        for name, flops, nbytes, e0, e1 in self.records:
            groups.setdefault((name, flops, nbytes), []).append(max(e0.elapsed_time(e1) * 1e-3 - self.overhead_s, 1e-7))
        out, self.outliers = {}, {}
        for (name, flops, nbytes), ts in groups.items():
            med = sorted(ts)[len(ts) // 2]
            bad = [t for t in ts if t > 8.0 * med and len(ts) >= 3]
            if bad:
                self.outliers[name] = self.outliers.get(name, 0) + len(bad)
            total = sum(med if (t > 8.0 * med and len(ts) >= 3) else t for t in ts)
            n, t0, f, b = out.get(name, (0, 0.0, 0.0, 0.0))
            out[name] = (n + len(ts), t0 + total, f + flops * len(ts), b + nbytes * len(ts))
        return out


def _timed(name, flops, nbytes, fn):
    """name: a label, or a callable evaluated after the launch (the library reports which kernel its planner chose)."""
    if _PROFILE is None:
        return fn()
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    rc = fn()
    e1.record()
    _PROFILE.append((name() if callable(name) else name, flops, nbytes, e0, e1))
    return rc


def _conv_kernel_name(tag=None):
    """The kernel the library's planner launched last; with RGBD_PROFILE_SHAPES=1 the layer shape is appended (per-shape
    tables: scripts/step_conv_shapes.py)."""
    name = _lib.load().rgbd_last_conv_kernel().decode()
    return f"{name} {tag}" if tag and os.environ.get("RGBD_PROFILE_SHAPES") else name


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


# Compute-unit budgets of the persistent 3x3 launches, per STREAM (cu_budget): host policy, handed to the library as the `cus`
# argument of every conv launch -- the library itself keeps none (ABI 19).  Keyed by (device, stream handle): launches issued
# by autograd's worker threads run on the stream their forward ran on and find the same entry; two updaters on two devices
# (or two streams) never see each other's.
_STREAM_CUS = {}


def _stream_key():
    s = torch.cuda.current_stream()
    return (s.device_index, s.cuda_stream)


def _cus():
    return _STREAM_CUS.get(_stream_key(), 0) if _STREAM_CUS else 0


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def _chk(t, dtype, name):
    if t is None:
        return
    if not t.is_cuda:
        raise RuntimeError(f"{name}: expected a GPU tensor (the HIP path has no CPU fallback)")
    if t.dtype != dtype:
        raise RuntimeError(f"{name}: expected {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise RuntimeError(f"{name}: expected a contiguous tensor")


# ------------------------------------------------------------------ warp loss
def warp_loss_fwd(img, img_rot, coef, flags, lambda_geometric, max_depth=0.0, min_depth=0.0, debug=False,
                  hinge_lambda=0.0, hinge_min=0.0, want_zp=False):
    """img, img_rot (b,4,S,S) fp32; coef (b,24) fp32 -> loss (1,) [+ debug tensors].  hinge_lambda > 0: the depth-range
    hinge of updater.py:357-359 over both image sets is added in the same pass.  want_zp: also return the projected
    points (2,b,S*S,3) = [new_zp, new_zp_rot] (loss_functions.py:146)."""
    for t, n in ((img, "img"), (img_rot, "img_rot"), (coef, "coef")):
        _chk(t, F32, n)
    b, C, S, _ = img.shape
    if C != 4 or img_rot.shape != img.shape or coef.shape != (b, 24):
        raise RuntimeError(f"warp_loss_fwd: bad shapes {tuple(img.shape)} {tuple(img_rot.shape)} {tuple(coef.shape)}")
    N = b * S * S
    partials = torch.empty(6 * ((N + 255) // 256), dtype=F32, device=img.device)
    loss = torch.empty(1, dtype=F32, device=img.device)
    zp = warped = idx = None
    if debug or want_zp:
        zp = torch.empty(2, b, S * S, 3, dtype=F32, device=img.device)
    if debug:
        warped = torch.empty(2, N, 4, dtype=F32, device=img.device)
        idx = torch.empty(2, N, 4, dtype=torch.int32, device=img.device)
    rc = _lib.load().rgbd_warp_loss_fwd(_ptr(img), _ptr(img_rot), _ptr(coef), b, S, int(flags),
                                        float(lambda_geometric), float(max_depth), float(min_depth),
                                        float(hinge_lambda), float(hinge_min),
                                        _ptr(partials), _ptr(loss), _ptr(zp), _ptr(warped), _ptr(idx), _stream())
    _lib.check(rc, "rgbd_warp_loss_fwd")
    if debug:
        return loss, zp, warped, idx
    return (loss, zp) if want_zp else loss


def warp_loss_bwd(img, img_rot, coef, flags, lambda_geometric, max_depth, min_depth, grad_loss, hinge_lambda=0.0,
                  hinge_min=0.0, grad_scale=1.0, out=None):
    """-> (grad_img, grad_img_rot).  grad_loss: (1,) fp32 device tensor or None (= 1), times the host factor grad_scale.
    out = (gimg, gimg_rot): ACCUMULATE into these buffers instead of returning fresh ones."""
    _chk(grad_loss, F32, "grad_loss")
    b, _, S, _ = img.shape
    if out is None:
        gimg, gimg_rot, acc = torch.empty_like(img), torch.empty_like(img_rot), 0
    else:
        gimg, gimg_rot, acc = out[0], out[1], 1
        _chk(gimg, F32, "out[0]"); _chk(gimg_rot, F32, "out[1]")
        if gimg.shape != img.shape or gimg_rot.shape != img_rot.shape:
            raise RuntimeError("warp_loss_bwd: accumulation buffers must have the images' shape")
    lib = _lib.load()
    ws = torch.empty(lib.rgbd_warp_loss_bwd_workspace(b, S) // 8, dtype=torch.int64, device=img.device)
    rc = lib.rgbd_warp_loss_bwd(_ptr(img), _ptr(img_rot), _ptr(coef), b, S, int(flags),
                                float(lambda_geometric), float(max_depth), float(min_depth),
                                float(hinge_lambda), float(hinge_min), _ptr(grad_loss), float(grad_scale),
                                _ptr(gimg), _ptr(gimg_rot), acc, _ptr(ws), _stream())
    _lib.check(rc, "rgbd_warp_loss_bwd")
    return gimg, gimg_rot


def warp_loss_nc_fwd(img, img_rot, coef, flags, norm_l2, lambda_geometric, max_depth=0.0, min_depth=0.0):
    """The warp loss for any channel count (last channel = depth), L1 or L2 criterion -> loss (1,)."""
    for t, n in ((img, "img"), (img_rot, "img_rot"), (coef, "coef")):
        _chk(t, F32, n)
    b, C, S, _ = img.shape
    if C < 2 or img_rot.shape != img.shape or coef.shape != (b, 24):
        raise RuntimeError(f"warp_loss_nc_fwd: bad shapes {tuple(img.shape)} {tuple(img_rot.shape)} {tuple(coef.shape)}")
    N = b * S * S
    partials = torch.empty(6 * ((N + 255) // 256), dtype=F32, device=img.device)
    loss = torch.empty(1, dtype=F32, device=img.device)
    rc = _lib.load().rgbd_warp_loss_nc_fwd(_ptr(img), _ptr(img_rot), _ptr(coef), b, C, S, int(flags), int(bool(norm_l2)),
                                           float(lambda_geometric), float(max_depth), float(min_depth), _ptr(partials),
                                           _ptr(loss), _stream())
    _lib.check(rc, "rgbd_warp_loss_nc_fwd")
    return loss


def warp_loss_nc_bwd(img, img_rot, coef, flags, norm_l2, lambda_geometric, max_depth, min_depth, grad_loss, grad_scale=1.0):
    _chk(grad_loss, F32, "grad_loss")
    b, C, S, _ = img.shape
    gimg, gimg_rot = torch.empty_like(img), torch.empty_like(img_rot)
    lib = _lib.load()
    ws = torch.empty(lib.rgbd_warp_loss_nc_bwd_workspace(b, C, S) // 8, dtype=torch.int64, device=img.device)
    rc = lib.rgbd_warp_loss_nc_bwd(_ptr(img), _ptr(img_rot), _ptr(coef), b, C, S, int(flags), int(bool(norm_l2)),
                                   float(lambda_geometric), float(max_depth), float(min_depth), _ptr(grad_loss),
                                   float(grad_scale), _ptr(gimg), _ptr(gimg_rot), 0, _ptr(ws), _stream())
    _lib.check(rc, "rgbd_warp_loss_nc_bwd")
    return gimg, gimg_rot


def nonfinite_mask(scalars, mask):
    """scalars: up to 8 device fp32 scalars (0-dim or 1-element tensors); mask: (1,) int32 device, bit i OR-ed in when scalar i
    is NaN / Inf."""
    for t in scalars:
        _chk(t, F32, "scalar")
    if mask.dtype != torch.int32 or not mask.is_cuda:
        raise RuntimeError("nonfinite_mask: the mask must be an int32 GPU tensor")
    arr = (ctypes.c_void_p * len(scalars))(*[t.data_ptr() for t in scalars])
    _lib.check(_lib.load().rgbd_nonfinite_mask_f32(arr, len(scalars), _ptr(mask), _stream()), "rgbd_nonfinite_mask_f32")


# ------------------------------------------------------------------ conv engine
def pack_weights(w, scale, want_fprop=True, want_dgrad=True):
    """w (Cout,Cin,KH,KW) fp32 -> (w_fprop [T][Cout][Cin], w_dgrad [T][Cin][Cout]) bf16 with `scale` folded in."""
    _chk(w, F32, "w")
    co, ci, kh, kw = w.shape
    wf = torch.empty(kh * kw, co, ci, dtype=BF16, device=w.device) if want_fprop else None
    wd = torch.empty(kh * kw, ci, co, dtype=BF16, device=w.device) if want_dgrad else None
    rc = _lib.load().rgbd_pack_weights(_ptr(w), co, ci, kh, kw, float(scale), _ptr(wf), _ptr(wd), _stream())
    _lib.check(rc, "rgbd_pack_weights")
    return wf, wd


PACK_DESC = [("w", "<u8"), ("wf", "<u8"), ("wd", "<u8"), ("cout", "<i4"), ("cin", "<i4"), ("taps", "<i4"),
             ("scale", "<f4"), ("block_begin", "<i4"), ("fold", "<i4")]          # struct rgbd_pack_desc, 48 bytes


PACK_MAX_BLOCKS = int(os.environ.get("RGBD_PACK_MAX_BLOCKS", "1024"))     # workgroups per layer of a pack launch (256 / 512 / 1024:
#                                                                             the DeepVoxels generator's 26 M weights in 114 / 97 / 77 us)


def _pack_work_items(co, ci, taps, fold_mode):
    """What pack_weights_multi_kernel deals out over a layer's workgroups: tiles of 8 output channels x 64 (unfolded) / 16 / 32 / 64 (fold
    mode 0 / 1 / 2) master input channels where the tiled path applies (csrc/elementwise.hip), else runs of 256 elements."""
    groups, ct, tm = ((1, 64, taps) if fold_mode in (None, 2) else (3, 16, 27) if fold_mode == 0 else (16, 32, 16))
    cip = ci // groups
    if co % 8 == 0 and cip % ct == 0 and tm <= (9 if fold_mode in (None, 2) else 27):
        return max(1, (co // 8) * (cip // ct))
    return (co * ci * taps + 255) // 256


def pack_fold_code(mode, Co, Ci):
    """The `fold` field of rgbd_pack_desc: the packing kernel reads a reference-shaped master through fold_weight's mode."""
    if not (0 <= mode <= 2 and 0 < Co < 32768 and 0 < Ci < 32768):
        raise ValueError(f"pack_fold_code: mode {mode}, master channels ({Co}, {Ci})")
    code = (mode + 1) | (Co << 2) | (Ci << 17)
    return code - (1 << 32) if code >= 1 << 31 else code            # the struct's field is a signed 32-bit integer


def build_pack_table(entries):
    """entries: list of (w fp32 (co,ci,k,k), scale, wf, wd) -> (device descriptor table, n, total_blocks).
    A folded entry is (master fp32, scale, wf, wd, (cop, cin_folded, k, k), (mode, Co, Ci)): the images are those of the master's
    fold_weight rearrangement, which is never materialised."""
    import numpy as np
    tab = np.zeros(len(entries), dtype=PACK_DESC)
    assert tab.dtype.itemsize == 48
    blocks = 0
    for i, ent in enumerate(entries):
        w, scale, wf, wd = ent[:4]
        _chk(w, F32, "w"); _chk(wf, BF16, "wf"); _chk(wd, BF16, "wd")
        fold = 0
        if len(ent) > 4:
            (co, ci, kh, kw), (mode, Co, Ci) = ent[4], ent[5]
            want = (Co, Ci, 3, 3, 3) if mode == 0 else (Co, Ci, 4, 4) if mode == 1 else (Co, Ci, kh, kw)
            if tuple(w.shape)[:2] != want[:2] or w.numel() != int(np.prod(want)) or not w.is_contiguous():
                raise RuntimeError(f"build_pack_table: master {tuple(w.shape)} is not the contiguous {want} of fold mode {mode}")
            if (mode == 0 and (kh, kw, ci % 3) != (3, 3, 0)) or (mode == 1 and (kh, kw, ci % 16) != (1, 1, 0)) or co < Co:
                raise RuntimeError(f"build_pack_table: folded shape {(co, ci, kh, kw)} does not fit mode {mode}")
            fold = pack_fold_code(mode, Co, Ci)
        else:
            co, ci, kh, kw = w.shape
        if tuple(wf.shape) != (kh * kw, co, ci) or tuple(wd.shape) != (kh * kw, ci, co):
            raise RuntimeError(f"build_pack_table: images {tuple(wf.shape)} / {tuple(wd.shape)} for a ({co},{ci},{kh},{kw}) weight")
        tab[i] = (w.data_ptr(), wf.data_ptr(), wd.data_ptr(), co, ci, kh * kw, scale, blocks, fold)
        blocks += min(PACK_MAX_BLOCKS, _pack_work_items(co, ci, kh * kw, ent[5][0] if len(ent) > 4 else None))
    dev = torch.from_numpy(tab.view(np.uint8).copy()).to(entries[0][0].device)
    return dev, len(entries), blocks


def pack_weights_multi(table):
    dev, n, blocks = table
    _lib.check(_lib.load().rgbd_pack_weights_multi(_ptr(dev), n, blocks, _stream()), "rgbd_pack_weights_multi")


# ---- MXFP8 operands (BASELINE configuration 5; format: csrc/mxfp8.hip, include/rgbd_gan_hip.h)
U8 = torch.uint8
MX8_MIN_TILES = 64          # (tests set 0 to reach the fp8 kernel with oracle-sized problems)
MX8_EMIT = True             # producers write the MXFP8 copy of their output (False: every conv input goes through the stand-alone
                            # quantiser -- the A/B of bench.py --mx8-standalone-quantiser)


class Mx8Image:
    """A packed 3x3 weight image in both forms: `.bf16` ([9][N][K] bf16, what every conv wrapper takes) and its MXFP8 twin
    `.q` ([9][N][K] e4m3 bytes) + `.s` ([9][N][K/32] E8M0 bytes).  A conv wrapper handed one of these runs the block-scaled
    fp8 kernel when the launch's shape allows (rgbd_conv3x3_mxfp8_supported) and the bf16 kernel on `.bf16` otherwise."""
    __slots__ = ("bf16", "q", "s")

    def __init__(self, bf16, q, s):
        self.bf16, self.q, self.s = bf16, q, s

    @property
    def shape(self):
        return self.bf16.shape


def _mx8_split(wp, B, Hout, Wout, K=3, pad=1):
    """-> (bf16 image, Mx8Image or None): the MXFP8 twin when this launch can use it."""
    if not isinstance(wp, Mx8Image):
        return wp, None
    _, N, Kd = wp.bf16.shape
    # launches with fewer 16x16 x 128 (or x 64) tiles than this stay on the bf16 planner, which has a split-K kernel for them
    tiles = B * (Hout // 16) * (Wout // 16) * (N // (128 if N % 128 == 0 else 64))
    ok = wp.q is not None and K == 3 and pad == 1 and tiles >= MX8_MIN_TILES and \
        bool(_lib.load().rgbd_conv3x3_mxfp8_supported(int(B), int(Hout), int(Wout), int(Kd), int(N)))
    return wp.bf16, (wp if ok else None)


def quantize_mx8(x):
    """x (..., C) bf16, C % 128 == 0 -> (q (..., C) uint8 e4m3, s (..., C/32) uint8 E8M0): blocks of 32 channels."""
    _chk(x, BF16, "x")
    C = x.shape[-1]
    if C % 128:
        raise RuntimeError(f"quantize_mx8: the channel count must be a multiple of 128, got {C}")
    # a tensor read by several convolutions (a residual block's input: c0 and c_sc; its gradient dz1: both input gradients)
    # is quantised once: the result rides on the tensor OBJECT, so it dies with it, and is dropped if torch has seen an
    # in-place write since (no kernel of this library rewrites an activation tensor it has handed out)
    hit = getattr(x, "_mx8", None)
    if hit is not None and hit[2] == x._version:
        return hit[0], hit[1]
    q = torch.empty(x.shape, dtype=U8, device=x.device)
    sc = torch.empty(tuple(x.shape[:-1]) + (C // 32,), dtype=U8, device=x.device)
    rows = x.numel() // C
    rc = _timed("quantize_mx8_kernel", 0.0, 3.03125 * x.numel(),
                lambda: _lib.load().rgbd_quantize_mxfp8(_ptr(x), _ptr(q), _ptr(sc), rows, C, _stream()))
    _lib.check(rc, "rgbd_quantize_mxfp8")
    x._mx8 = (q, sc, x._version)
    return q, sc


class Conv3x3Desc(ctypes.Structure):
    """struct rgbd_conv3x3_desc (include/rgbd_gan_hip.h)"""
    _fields_ = [(n, ctypes.c_void_p) for n in ("x", "x_scales", "w", "w_scales", "bias", "residual", "act_y", "colsum",
                                                "row_scale", "y", "y_pooled", "y2", "row_scale2", "stats", "y_q", "y_s",
                                                "yp_q", "yp_s")] + \
               [(n, ctypes.c_int) for n in ("B", "Hin", "Win", "Cin", "Cout", "upsample", "pool_sum", "lrelu_channels")] + \
               [("slope", ctypes.c_float), ("cus", ctypes.c_int)]


def mx8_emittable(B, Hout, Wout, Cout):
    """Can a 3x3 launch with this output write an MXFP8 copy of it that a following convolution will use?  (the pipelined
    kernel's epilogue: output images multiples of 16x16; the consumer's reduction channels = Cout a multiple of 128)"""
    return MX8_EMIT and Hout % 16 == 0 and Wout % 16 == 0 and Cout % 128 == 0 and Hout >= 16 and \
        B * (Hout // 16) * (Wout // 16) * (Cout // 128) >= MX8_MIN_TILES


def _mx8_side(t, want):
    """(q, s) buffers for the MXFP8 copy of the bf16 tensor `t` a kernel is about to write, or (None, None); the caller hangs
    them on the tensor with _mx8_attach once the launch has been issued."""
    C = t.shape[-1]
    if not want or not MX8_EMIT or C % 128:
        return None, None
    return (torch.empty(t.shape, dtype=U8, device=t.device),
            torch.empty(tuple(t.shape[:-1]) + (C // 32,), dtype=U8, device=t.device))


def _mx8_attach(t, q, s):
    if q is not None:
        t._mx8 = (q, s, t._version)


def _conv3x3_ex(x, mx_x, wp, mx_w, y, B, H, W, Cin, Cout, bias=None, residual=None, act_y=None, colsum=None, row_scale=None,
                y_pooled=None, y2=None, row_scale2=None, lrelu_channels=0, slope=0.2, emit=False, emit_pooled=False):
    """One rgbd_conv3x3_ex launch (no upsample, no statistics); emit / emit_pooled: also write the MXFP8 copies and hang them
    on y / y_pooled (`_mx8`, what quantize_mx8 looks for)."""
    dev = x.device
    yq = ys = ypq = yps = None
    if emit:
        yq = torch.empty(y.shape, dtype=U8, device=dev)
        ys = torch.empty(tuple(y.shape[:-1]) + (Cout // 32,), dtype=U8, device=dev)
    if emit_pooled:
        ypq = torch.empty(y_pooled.shape, dtype=U8, device=dev)
        yps = torch.empty(tuple(y_pooled.shape[:-1]) + (Cout // 32,), dtype=U8, device=dev)
    p = lambda t: t.data_ptr() if t is not None else None
    d = Conv3x3Desc(p(mx_x[0]) if mx_x else p(x), p(mx_x[1]) if mx_x else None, p(mx_w.q) if mx_w else p(wp),
                    p(mx_w.s) if mx_w else None, p(bias), p(residual), p(act_y), p(colsum), p(row_scale), p(y), p(y_pooled), p(y2),
                    p(row_scale2), None, p(yq), p(ys), p(ypq), p(yps), B, H, W, Cin, Cout, 0, 0, int(lrelu_channels), float(slope), _cus())
    rc = _lib.load().rgbd_conv3x3_ex(ctypes.byref(d), _stream())
    if emit:
        y._mx8 = (yq, ys, y._version)
    if emit_pooled:
        y_pooled._mx8 = (ypq, yps, y_pooled._version)
    return rc


PACK_MX8_DESC = [("w", "<u8"), ("wf_q", "<u8"), ("wf_s", "<u8"), ("wd_q", "<u8"), ("wd_s", "<u8"), ("cout", "<i4"),
                 ("cin", "<i4"), ("scale", "<f4"), ("block_begin", "<i4")]             # struct rgbd_pack_mx8_desc, 56 bytes


def build_pack_table_mx8(entries):
    """entries: list of (w fp32 (co,ci,3,3), scale, wf_q, wf_s, wd_q, wd_s) (either image pair may be None) -> table."""
    import numpy as np
    tab = np.zeros(len(entries), dtype=PACK_MX8_DESC)
    assert tab.dtype.itemsize == 56
    blocks = 0
    for i, (w, scale, fq, fs, dq, ds) in enumerate(entries):
        _chk(w, F32, "w")
        for t in (fq, fs, dq, ds):
            _chk(t, U8, "mxfp8 image")
        co, ci, kh, kw = w.shape
        if (kh, kw) != (3, 3) or co % 32 or ci % 32 or (fq is not None and ci % 128) or (dq is not None and co % 128):
            raise RuntimeError(f"build_pack_table_mx8: unsupported weight {tuple(w.shape)}")
        ptr = lambda t: t.data_ptr() if t is not None else 0
        tab[i] = (w.data_ptr(), ptr(fq), ptr(fs), ptr(dq), ptr(ds), co, ci, scale, blocks)
        blocks += min(512, (co // 32) * (ci // 32))
    dev = torch.from_numpy(tab.view(np.uint8).copy()).to(entries[0][0].device)
    return dev, len(entries), blocks


def pack_weights_mx8_multi(table):
    dev, n, blocks = table
    _lib.check(_lib.load().rgbd_pack_weights_mxfp8_multi(_ptr(dev), n, blocks, _stream()), "rgbd_pack_weights_mxfp8_multi")


def pack_weights_mx8(w, scale, want_fprop=True, want_dgrad=True):
    """w (Cout,Cin,3,3) fp32 -> ((wf_q, wf_s) or None, (wd_q, wd_s) or None): the MXFP8 images with `scale` folded in; an image
    whose reduction dimension is not a multiple of 128 is None."""
    co, ci = w.shape[:2]
    dev = w.device
    f = (torch.empty(9, co, ci, dtype=U8, device=dev), torch.empty(9, co, ci // 32, dtype=U8, device=dev)) \
        if want_fprop and ci % 128 == 0 else None
    d = (torch.empty(9, ci, co, dtype=U8, device=dev), torch.empty(9, ci, co // 32, dtype=U8, device=dev)) \
        if want_dgrad and co % 128 == 0 else None
    if f is None and d is None:
        return None, None
    pack_weights_mx8_multi(build_pack_table_mx8([(w, scale) + (f or (None, None)) + (d or (None, None))]))
    return f, d


def _fprop_workspace(lib, B, H, W, Cin, Cout, KH, KW, pad, ups, device):
    """Scratch for the split-K path of the small layers (None when the shape is not split)."""
    nbytes = lib.rgbd_conv2d_fprop_workspace(B, H, W, Cin, Cout, KH, KW, pad, ups)
    if nbytes < 0:
        raise RuntimeError(f"conv2d: unsupported shape B={B} H={H} W={W} Cin={Cin} Cout={Cout} K={KH}x{KW} pad={pad}")
    return torch.empty(nbytes // 4, dtype=F32, device=device) if nbytes > 0 else None


def conv2d_fprop(x, wp, KH, KW, pad, bias=None, residual=None, upsample=False, lrelu_channels=0, slope=0.2,
                 avg_pool2=False, emit_mx8=False):
    """x (B,H,W,Cin) bf16, wp [KH*KW][Cout][Cin] bf16 -> y (B,Hout,Wout,Cout) bf16.
    avg_pool2: -> (y, 2x2 average of y at half resolution); the average comes out of the conv epilogue for 3x3 convs on
    images that are multiples of 16x16, out of a second pass (rgbd_pool2_masked) otherwise."""
    B, H, W, Cin = x.shape
    Hup, Wup = (2 * H, 2 * W) if upsample else (H, W)
    Hout, Wout = Hup + 2 * pad - KH + 1, Wup + 2 * pad - KW + 1
    wp, mx = _mx8_split(wp, B, Hout, Wout, KH if KH == KW else 0, pad)
    _chk(x, BF16, "x"); _chk(wp, BF16, "wp"); _chk(bias, F32, "bias"); _chk(residual, BF16, "residual")
    T, Cout, Cin2 = wp.shape
    if T != KH * KW or Cin2 != Cin:
        raise RuntimeError(f"conv2d_fprop: weights {tuple(wp.shape)} do not match x {tuple(x.shape)} K={KH}x{KW}")
    y = torch.empty(B, Hout, Wout, Cout, dtype=BF16, device=x.device)
    if residual is not None and residual.shape != y.shape:
        raise RuntimeError("conv2d_fprop: residual shape mismatch")
    lib = _lib.load()
    flops = 2.0 * B * Hout * Wout * Cout * Cin * KH * KW
    nbytes = 2.0 * (x.numel() + y.numel() + wp.numel() + (residual.numel() if residual is not None else 0))
    fuse_pool = bool(avg_pool2) and KH == 3 and KW == 3 and pad == 1 and Hout % 16 == 0 and Wout % 16 == 0
    yp = torch.empty(B, Hout // 2, Wout // 2, Cout, dtype=BF16, device=x.device) if fuse_pool else None
    # emit_mx8: the launch also leaves the MXFP8 copy of what the NEXT convolution will read -- of the pooled output when
    # there is one (the next residual block's input), else of y (c0 -> c1) -- where that convolution can use it
    emit_y = bool(emit_mx8) and not avg_pool2 and not upsample and KH == 3 and pad == 1 and mx8_emittable(B, Hout, Wout, Cout)
    emit_p = bool(emit_mx8) and fuse_pool and not upsample and mx8_emittable(B, Hout // 2, Wout // 2, Cout)
    if emit_y or emit_p:
        mx_x = quantize_mx8(x) if mx is not None else None
        rc = _timed(lambda: _conv_kernel_name(f"fprop {Hout}x{Wout} {Cin}->{Cout}{' pool' if fuse_pool else ''}"
                                              f"{' res' if residual is not None else ''} emit"), flops,
                    nbytes if mx is None else nbytes - 0.97 * (x.numel() + wp.numel()),
                    lambda: _conv3x3_ex(x, mx_x, wp, mx, y, B, H, W, Cin, Cout, bias=bias, residual=residual, y_pooled=yp,
                                        lrelu_channels=lrelu_channels, slope=slope, emit=emit_y, emit_pooled=emit_p))
        _lib.check(rc, "rgbd_conv3x3_ex")
        if avg_pool2:
            return y, yp
        return y
    if mx is not None:
        xq, xs = quantize_mx8(x)
        nbytes = 1.03 * (x.numel() + mx.q.numel()) + 2.0 * (y.numel() + (residual.numel() if residual is not None else 0))
        rc = _timed(lambda: _conv_kernel_name(f"fprop {Hout}x{Wout} {Cin}->{Cout}{' ups' if upsample else ''}"
                                              f"{' pool' if fuse_pool else ''}{' res' if residual is not None else ''}"),
                    flops, nbytes,
                    lambda: lib.rgbd_conv2d_fprop_mxfp8(_ptr(xq), _ptr(xs), _ptr(mx.q), _ptr(mx.s), _ptr(bias), _ptr(residual),
                                                        _ptr(y), _ptr(yp), B, H, W, Cin, Cout, int(bool(upsample)),
                                                        int(lrelu_channels), float(slope), _cus(), _stream()))
        _lib.check(rc, "rgbd_conv2d_fprop_mxfp8")
        if avg_pool2:
            return y, (yp if fuse_pool else pool2_masked(y))
        return y
    ws = None if fuse_pool else _fprop_workspace(lib, B, H, W, Cin, Cout, KH, KW, pad, int(bool(upsample)), x.device)
    rc = _timed(lambda: _conv_kernel_name(f"fprop {Hout}x{Wout} {Cin}->{Cout}{' ups' if upsample else ''}"
                                          f"{' pool' if fuse_pool else ''}{' res' if residual is not None else ''}"),
                flops, nbytes,
                lambda: lib.rgbd_conv2d_fprop_bf16(_ptr(x), _ptr(wp), _ptr(bias), _ptr(residual), _ptr(y), _ptr(yp), B, H,
                                                   W, Cin, Cout, KH, KW, pad, int(bool(upsample)), int(lrelu_channels),
                                                   float(slope), _ptr(ws), _cus(), _stream()))
    _lib.check(rc, "rgbd_conv2d_fprop_bf16")
    if avg_pool2:
        return y, (yp if fuse_pool else pool2_masked(y))
    return y


def conv2d_dgrad(dy, wd, K, pad, sum_pool2=False, residual=None):
    """dy (B,H,W,Cout) bf16, wd [K*K][Cin][Cout] bf16 (dgrad image of pack_weights) -> dx (B,H',W',Cin) bf16.
    sum_pool2: return the 2x2 sums of dx at half resolution (adjoint of a nearest-2x upsample in front of the conv);
    taken in the conv epilogue for 3x3 convs on images that are multiples of 16x16, in a second pass otherwise."""
    B, H, W, Cout = dy.shape
    wd, mx = _mx8_split(wd, B, H + 2 * (K - 1 - pad) - K + 1, W + 2 * (K - 1 - pad) - K + 1, K, K - 1 - pad)
    _chk(dy, BF16, "dy"); _chk(wd, BF16, "wd"); _chk(residual, BF16, "residual")
    T, Cin, Cout2 = wd.shape
    if T != K * K or Cout2 != Cout:
        raise RuntimeError(f"conv2d_dgrad: weights {tuple(wd.shape)} do not match dy {tuple(dy.shape)} K={K}")
    if residual is not None and (sum_pool2 or tuple(residual.shape) != (B, H + K - 1 - 2 * pad, W + K - 1 - 2 * pad, Cin)):
        raise RuntimeError("conv2d_dgrad: residual must have the shape of dx (and excludes sum_pool2)")
    pd = K - 1 - pad
    Ho, Wo = H + 2 * pd - K + 1, W + 2 * pd - K + 1
    fuse = bool(sum_pool2) and K == 3 and pd == 1 and Ho % 16 == 0 and Wo % 16 == 0
    dx = torch.empty((B, Ho // 2, Wo // 2, Cin) if fuse else (B, Ho, Wo, Cin), dtype=BF16, device=dy.device)
    lib = _lib.load()
    flops = 2.0 * B * Ho * Wo * Cout * Cin * K * K
    nbytes = 2.0 * (dy.numel() + dx.numel() + wd.numel() + (residual.numel() if residual is not None else 0))
    if mx is not None and (fuse or not sum_pool2):
        dq, dsc = quantize_mx8(dy)
        nbytes = 1.03 * (dy.numel() + mx.q.numel()) + 2.0 * (dx.numel() + (residual.numel() if residual is not None else 0))
        rc = _timed(lambda: _conv_kernel_name(f"dgrad {Ho}x{Wo} {Cout}->{Cin}{' sumpool' if fuse else ''}"
                                              f"{' res' if residual is not None else ''}"), flops, nbytes,
                    lambda: lib.rgbd_conv2d_dgrad_mxfp8(_ptr(dq), _ptr(dsc), _ptr(mx.q), _ptr(mx.s), _ptr(residual), _ptr(dx), B,
                                                        H, W, Cin, Cout, int(fuse), _cus(), _stream()))
        _lib.check(rc, "rgbd_conv2d_dgrad_mxfp8")
        return dx
    ws = None if fuse else _fprop_workspace(lib, B, H, W, Cout, Cin, K, K, pd, 0, dy.device)
    rc = _timed(lambda: _conv_kernel_name(f"dgrad {Ho}x{Wo} {Cout}->{Cin}{' sumpool' if fuse else ''}"
                                          f"{' res' if residual is not None else ''}"), flops, nbytes,
                lambda: lib.rgbd_conv2d_dgrad_bf16(_ptr(dy), _ptr(wd), _ptr(residual), _ptr(dx), B, H, W, Cin, Cout, K, pad,
                                                   int(fuse),
                                                   _ptr(ws), _cus(), _stream()))
    _lib.check(rc, "rgbd_conv2d_dgrad_bf16")
    if sum_pool2 and not fuse:
        out = torch.empty(B, Ho // 2, Wo // 2, Cin, dtype=BF16, device=dx.device)     # (a torch reduction here was the last one
        _lib.check(lib.rgbd_pool2_sum_bf16(_ptr(dx), _ptr(out), B, Ho, Wo, Cin, _stream()), "rgbd_pool2_sum_bf16")    # on the step)
        dx = out
    return dx


def conv3x3_actgrad_supported(B, H, W, Cin, Cout):
    """Shapes the fused (3x3 conv, activation gradient of the layer in front) launch covers: the pipelined kernel's."""
    return bool(_lib.load().rgbd_conv3x3_actgrad_supported(int(B), int(H), int(W), int(Cin), int(Cout)))


def conv3x3_actgrad(x, wp, act_y, residual=None, bias_grad=None, row_scale=None, slope=0.2, operand_scale=None,
                    emit_mx8=False):
    """(conv3x3_pad1(x, wp) + residual) * lrelu'(act_y) in one launch; x (B,H,W,Cin) bf16, wp [9][Cout][Cin] bf16 (the fprop
    image, or the dgrad image of a Cout->Cin convolution), act_y / residual (B,H,W,Cout) bf16.  bias_grad (Cout fp32,
    accumulated): += sum_b row_scale[b] * column sums of the result (row_scale None = 1).  operand_scale (B,) fp32:
    -> (y, y + operand_scale[b] * act_y), the second tensor being axpy_rows(y, act_y, operand_scale)."""
    B, H, W, Cin = x.shape
    wp, mx = _mx8_split(wp, B, H, W)
    _chk(x, BF16, "x"); _chk(wp, BF16, "wp"); _chk(act_y, BF16, "act_y"); _chk(residual, BF16, "residual")
    _chk(bias_grad, F32, "bias_grad"); _chk(row_scale, F32, "row_scale")
    T, Cout, Cin2 = wp.shape
    if T != 9 or Cin2 != Cin or tuple(act_y.shape) != (B, H, W, Cout):
        raise RuntimeError(f"conv3x3_actgrad: x {tuple(x.shape)}, weights {tuple(wp.shape)}, act_y {tuple(act_y.shape)}")
    if residual is not None and residual.shape != act_y.shape:
        raise RuntimeError("conv3x3_actgrad: residual shape mismatch")
    if bias_grad is not None and bias_grad.numel() != Cout or (row_scale is not None and row_scale.numel() != B):
        raise RuntimeError("conv3x3_actgrad: bias_grad needs Cout entries, row_scale B")
    _chk(operand_scale, F32, "operand_scale")
    if operand_scale is not None and operand_scale.numel() != B:
        raise RuntimeError("conv3x3_actgrad: operand_scale needs B entries")
    y = torch.empty(B, H, W, Cout, dtype=BF16, device=x.device)
    y2 = torch.empty_like(y) if operand_scale is not None else None
    lib = _lib.load()
    flops = 2.0 * B * H * W * Cout * Cin * 9
    nbytes = 2.0 * (x.numel() + (3 if y2 is not None else 2) * y.numel() + wp.numel() +
                    (residual.numel() if residual is not None else 0))
    if emit_mx8 and mx8_emittable(B, H, W, Cout):
        mx_x = quantize_mx8(x) if mx is not None else None
        rc = _timed(lambda: _conv_kernel_name(f"actgrad {H}x{W} {Cin}->{Cout}{' res' if residual is not None else ''} emit"),
                    flops, nbytes if mx is None else nbytes - 0.97 * (x.numel() + wp.numel()),
                    lambda: _conv3x3_ex(x, mx_x, wp, mx, y, B, H, W, Cin, Cout, residual=residual, act_y=act_y, colsum=bias_grad,
                                        row_scale=row_scale, y2=y2, row_scale2=operand_scale, slope=slope, emit=True))
        _lib.check(rc, "rgbd_conv3x3_ex")
        return y if y2 is None else (y, y2)
    if mx is not None:
        xq, xs = quantize_mx8(x)
        nbytes = nbytes - 0.97 * (x.numel() + wp.numel())
        rc = _timed(lambda: _conv_kernel_name(f"actgrad {H}x{W} {Cin}->{Cout}{' res' if residual is not None else ''}"),
                    flops, nbytes,
                    lambda: lib.rgbd_conv3x3_actgrad_mxfp8(_ptr(xq), _ptr(xs), _ptr(mx.q), _ptr(mx.s), _ptr(residual),
                                                           _ptr(act_y), float(slope), _ptr(bias_grad), _ptr(row_scale), _ptr(y),
                                                           _ptr(y2), _ptr(operand_scale), B, H, W, Cin, Cout, _cus(), _stream()))
        _lib.check(rc, "rgbd_conv3x3_actgrad_mxfp8")
        return y if y2 is None else (y, y2)
    rc = _timed(lambda: _conv_kernel_name(f"actgrad {H}x{W} {Cin}->{Cout}{' res' if residual is not None else ''}"),
                flops, nbytes,
                lambda: lib.rgbd_conv3x3_actgrad_bf16(_ptr(x), _ptr(wp), _ptr(residual), _ptr(act_y), float(slope),
                                                      _ptr(bias_grad), _ptr(row_scale), _ptr(y), _ptr(y2),
                                                      _ptr(operand_scale), B, H, W, Cin, Cout, _cus(), _stream()))
    _lib.check(rc, "rgbd_conv3x3_actgrad_bf16")
    return y if y2 is None else (y, y2)


class _StatsPool:
    """Zeroed int64 scratch for the instance-norm statistics of the `stats` conv epilogues (integer atomics ADD into them).
    A training step clears the pool with the launch that clears its gradient buffers (begin_step: one more pointer in
    kernels.zero_multi) and the convolutions of the step take consecutive slices -- instead of one torch fill launch per
    layer (eight per step, the last torch `zeros` on it).  The take sequence of a step is fixed, so a captured step finds the
    same slices on every replay.  ONLY a step's own takes count: a caller outside begin_step .. end_step (the preview
    sampler, tests, a generator called by hand) gets a fresh torch.zeros and leaves no trace -- a run that replays its steps
    from graphs for 100 000 iterations and renders previews in between must not grow the pool.  A pool that is still too
    small (a configuration's first step) also falls back to torch.zeros and the pool grows to exactly that step's demand at
    the next begin_step."""

    def __init__(self):
        self.buf, self.off, self.used, self.in_step, self._retired = None, 0, 0, False, []

    def take(self, n, device):
        if not self.in_step:
            return None
        self.used += n
        if self.buf is None or self.buf.device != device or self.off + n > self.buf.numel():
            return None
        v = self.buf[self.off:self.off + n]
        self.off += n
        return v

    def begin_step(self, device, defer):
        """Start of a step (a prep phase, before any conv of the step): grow to what the last step asked for, hand the pool
        to the caller's zero launch, rewind."""
        device = torch.device(device)
        if self.buf is None or self.buf.device != device or self.used > self.buf.numel():
            if self.buf is not None:
                self._retired.append(self.buf)      # steps captured as HIP graphs keep clearing and using the pool they saw
            self.buf = torch.empty(max(-(-self.used // 4096) * 4096, 1 << 14), dtype=torch.int64, device=device)
        self.off, self.used, self.in_step = 0, 0, True
        defer.append(self.buf.view(F32))

    def end_step(self):
        self.in_step = False


STATS_POOL = _StatsPool()


def conv2d_fprop_stats(x, wp, bias, upsample=False, lrelu_channels=0, slope=0.2):
    """3x3 pad-1 conv (+ nearest-2x upsample in front) + bias + leaky ReLU like conv2d_fprop, and the per-(sample, channel)
    (sum y, sum y^2) of the stored values as (B,Cout,2) int64 in units of 2^-32 (for adain_apply_fixed).  Output images must
    be multiples of 16x16 (conv3x3_actgrad_supported(B, Hout, Wout, Cin, Cout))."""
    B, H, W, Cin = x.shape
    Hout, Wout = (2 * H, 2 * W) if upsample else (H, W)
    wp, mx = _mx8_split(wp, B, Hout, Wout)
    _chk(x, BF16, "x"); _chk(wp, BF16, "wp"); _chk(bias, F32, "bias")
    T, Cout, Cin2 = wp.shape
    if T != 9 or Cin2 != Cin:
        raise RuntimeError(f"conv2d_fprop_stats: weights {tuple(wp.shape)} do not match x {tuple(x.shape)}")
    y = torch.empty(B, Hout, Wout, Cout, dtype=BF16, device=x.device)
    stats = STATS_POOL.take(B * Cout * 2, x.device)
    stats = stats.view(B, Cout, 2) if stats is not None else torch.zeros(B, Cout, 2, dtype=torch.int64, device=x.device)
    lib = _lib.load()
    flops = 2.0 * B * Hout * Wout * Cout * Cin * 9
    nbytes = 2.0 * (x.numel() + y.numel() + wp.numel())
    if mx is not None:
        xq, xs = quantize_mx8(x)
        nbytes = 1.03 * (x.numel() + mx.q.numel()) + 2.0 * y.numel()
        rc = _timed(lambda: _conv_kernel_name(f"fprop {Hout}x{Wout} {Cin}->{Cout}{' ups' if upsample else ''} stats"),
                    flops, nbytes,
                    lambda: lib.rgbd_conv2d_fprop_stats_mxfp8(_ptr(xq), _ptr(xs), _ptr(mx.q), _ptr(mx.s), _ptr(bias), _ptr(y),
                                                              _ptr(stats), B, H, W, Cin, Cout, int(bool(upsample)),
                                                              int(lrelu_channels), float(slope), _cus(), _stream()))
        _lib.check(rc, "rgbd_conv2d_fprop_stats_mxfp8")
        return y, stats
    rc = _timed(lambda: _conv_kernel_name(f"fprop {Hout}x{Wout} {Cin}->{Cout}{' ups' if upsample else ''} stats"),
                flops, nbytes,
                lambda: lib.rgbd_conv2d_fprop_stats_bf16(_ptr(x), _ptr(wp), _ptr(bias), _ptr(y), _ptr(stats), B, H, W, Cin,
                                                         Cout, int(bool(upsample)), int(lrelu_channels), float(slope),
                                                         _cus(), _stream()))
    _lib.check(rc, "rgbd_conv2d_fprop_stats_bf16")
    return y, stats


def adain_apply_fixed(x, stats, scale, shift=None, eps=1e-5, col_off=0, emit_mx8=False):
    """adain_fwd with the statistics given (conv2d_fprop_stats) instead of reduced from x -> y, mean, rstd."""
    _chk(x, BF16, "x"); _chk(scale, F32, "scale"); _chk(shift, F32, "shift")
    B, H, W, C = x.shape
    if stats.dtype != torch.int64 or tuple(stats.shape) != (B, C, 2) or not stats.is_contiguous():
        raise RuntimeError(f"adain_apply_fixed: statistics {tuple(stats.shape)} {stats.dtype} do not match x {tuple(x.shape)}")
    fused = shift is None
    if fused:
        if scale.dim() != 2 or scale.shape[0] != B or col_off % 4 or col_off + 2 * C > scale.shape[1]:
            raise RuntimeError(f"adain_apply_fixed: window [{col_off},{col_off + 2 * C}) outside {tuple(scale.shape)}")
        ld = scale.shape[1]
    elif scale.shape != (B, C):
        raise RuntimeError(f"adain_apply_fixed: scale {tuple(scale.shape)} does not match x {tuple(x.shape)}")
    y = torch.empty_like(x)
    mean = torch.empty(B, C, dtype=F32, device=x.device)
    rstd = torch.empty(B, C, dtype=F32, device=x.device)
    yq, ysc = _mx8_side(y, emit_mx8)
    rc = _lib.load().rgbd_adain_apply_fixed(_ptr(x), _off(scale, col_off) if fused else _ptr(scale),
                                            _off(scale, col_off + C) if fused else _ptr(shift), _ptr(y), _ptr(stats),
                                            _ptr(mean), _ptr(rstd), B, H * W, C, ld if fused else C, float(eps), _ptr(yq),
                                            _ptr(ysc), _stream())
    _lib.check(rc, "rgbd_adain_apply_fixed")
    _mx8_attach(y, yq, ysc)
    return y, mean, rstd


def conv2d_wgrad(x, dy, K, scale, out=None, accumulate=False, upsample=False):
    """x (B,H,W,Cin) bf16, dy (B,H,W,Cout) bf16 -> dW (Cout,Cin,K,K) fp32 = scale * sum dy (x) x.
    upsample: x is (B,H/2,W/2,Cin) and stands for its nearest-2x upsampling (read through the index map, not copied)."""
    _chk(x, BF16, "x"); _chk(dy, BF16, "dy")
    B, H, W, Cin = x.shape
    if upsample:
        H, W = 2 * H, 2 * W
    Cout = dy.shape[3]
    if tuple(dy.shape[:3]) != (B, H, W):
        raise RuntimeError(f"conv2d_wgrad: spatial mismatch {tuple(x.shape)} vs {tuple(dy.shape)} (upsample={upsample})")
    lib = _lib.load()
    ws_bytes = lib.rgbd_conv2d_wgrad_workspace(B, H, W, Cin, Cout, K)
    if ws_bytes < 0:
        raise RuntimeError(f"conv2d_wgrad: unsupported shape x={tuple(x.shape)} dy={tuple(dy.shape)} K={K}")
    ws = torch.empty(ws_bytes // 4, dtype=F32, device=x.device)
    if out is None:
        out = torch.empty(Cout, Cin, K, K, dtype=F32, device=x.device)
        accumulate = False
    flops = 2.0 * B * H * W * Cout * Cin * K * K
    nbytes = 2.0 * (x.numel() + dy.numel()) + 2.0 * ws_bytes
    rc = _timed(f"conv_wgrad_kernel<{K * K}>+reduce", flops, nbytes,
                lambda: lib.rgbd_conv2d_wgrad_bf16(_ptr(x), _ptr(dy), _ptr(ws), _ptr(out), B, H, W, Cin, Cout, K,
                                                   float(scale), int(bool(accumulate)), int(bool(upsample)),
                                                   _stream()))
    _lib.check(rc, "rgbd_conv2d_wgrad_bf16")
    return out


# ------------------------------------------------------------------ fused elementwise / 1x1
def lrelu_bwd(dy, y, act_channels, slope=0.2, bias_grad=None, row_scale=None):
    """dz = dy * (y > 0 ? 1 : slope) on channels [0, act_channels) of NHWC bf16 tensors.
    bias_grad (C,) fp32: if given, the column sums of dz are ADDED to it in the same pass, every row weighted by
    row_scale[sample] (B,) fp32 when that is given."""
    _chk(dy, BF16, "dy"); _chk(y, BF16, "y"); _chk(bias_grad, F32, "bias_grad"); _chk(row_scale, F32, "row_scale")
    C = y.shape[-1]
    dz = torch.empty_like(y)
    rps = y.numel() // C // y.shape[0] if row_scale is not None else 0
    rc = _lib.load().rgbd_lrelu_bwd(_ptr(dy), _ptr(y), _ptr(dz), y.numel() // C, C, int(act_channels), float(slope),
                                    _ptr(bias_grad), _ptr(row_scale), rps, _stream())
    _lib.check(rc, "rgbd_lrelu_bwd")
    return dz


def colsum(x, out=None, row_scale=None, rows_per_sample=0):
    """(.., C) bf16 -> (C,) fp32 column sums (added to `out` when given); with row_scale (B,) fp32 every row is weighted
    by the entry of its sample (rows_per_sample consecutive rows per sample)."""
    _chk(x, BF16, "x"); _chk(out, F32, "out"); _chk(row_scale, F32, "row_scale")
    C = x.shape[-1]
    acc = out is not None
    if out is None:
        out = torch.empty(C, dtype=F32, device=x.device)
    rc = _lib.load().rgbd_colsum_bf16(_ptr(x), _ptr(out), x.numel() // C, C, int(acc), _ptr(row_scale),
                                      int(rows_per_sample), _stream())
    _lib.check(rc, "rgbd_colsum_bf16")
    return out


WGRAD_REDUCE_DESC = [("workspace", "<u8"), ("dw", "<u8"), ("nsplit", "<i4"), ("taps", "<i4"), ("cout", "<i4"),
                     ("cin", "<i4"), ("scale", "<f4"), ("accumulate", "<i4")]      # struct rgbd_wgrad_reduce_desc, 40 bytes


class _WgradProblem(ctypes.Structure):          # struct rgbd_wgrad_problem (include/rgbd_gan_hip.h)
    _fields_ = [("x", ctypes.c_void_p), ("dy", ctypes.c_void_p), ("workspace", ctypes.c_void_p), ("B", ctypes.c_int32),
                ("H", ctypes.c_int32), ("W", ctypes.c_int32), ("Cin", ctypes.c_int32), ("Cout", ctypes.c_int32),
                ("K", ctypes.c_int32), ("upsample", ctypes.c_int32), ("nsplit", ctypes.c_int32)]


WGRAD_MULTI_MAX = 24
# Persistent workgroups the batched weight-gradient launch deals out over its problems: 0 = one per compute unit.  A host
# that runs a second stream beside the launch leaves compute units for it (wgrad_workgroups; RGBDUpdater does, for the
# launches of its side stream: the launch fills the chip for ~0.5 ms, during which nothing of the other stream would start).
WGRAD_WORKGROUPS = 0


@contextlib.contextmanager
def cu_budget(n):
    """Conv launches issued inside ON THE STREAM THAT IS CURRENT AT ENTRY size the persistent 3x3 kernels' grids for `n`
    compute units instead of all (the `cus` argument of the conv entry points; 0 / None = all).  Enter it where the stream
    the launches go to is already current (inside a graph capture: the capture stream)."""
    key = _stream_key()
    prev = _STREAM_CUS.get(key)
    if n:
        _STREAM_CUS[key] = int(n)
    else:
        _STREAM_CUS.pop(key, None)
    try:
        yield
    finally:
        if prev is None:
            _STREAM_CUS.pop(key, None)
        else:
            _STREAM_CUS[key] = prev


@contextlib.contextmanager
def wgrad_workgroups(n):
    global WGRAD_WORKGROUPS
    saved, WGRAD_WORKGROUPS = WGRAD_WORKGROUPS, int(n or 0)
    try:
        yield
    finally:
        WGRAD_WORKGROUPS = saved


def _multi_ok(H, W, K):
    return K == 3 and H >= 8 and W >= 16 and H & (H - 1) == 0 and W & (W - 1) == 0


def _multi_small_ok(H, W, K):
    return K == 3 and H >= 4 and W >= 4 and not (H >= 8 and W >= 16) and H & (H - 1) == 0 and W & (W - 1) == 0


def conv2d_wgrad_batch(items):
    """Weight gradients of several convs, `items` = [(x, dy, target dW fp32, K, scale, upsample)], accumulated into their
    targets.  The 3x3 convs on images of 8x16 and larger share ONE partial-sum launch per 24 of them (the chip's
    workgroups dealt out over the layers by work: 38 MB of slab traffic per launch instead of per layer); the 3x3 convs on
    smaller images share another one; 1x1 convs get a launch each; ONE slab-reduction launch per 32 gradients finishes
    all of them."""
    import numpy as np
    if not items:
        return
    lib = _lib.load()
    tab = np.zeros(len(items), dtype=WGRAD_REDUCE_DESC)
    assert tab.dtype.itemsize == 40
    keep, multi, multi_small, accs = [], [], [], {}
    for i, item in enumerate(items):
        x, dy, target, K, scale, ups = item[:6]
        acc_flag = int(bool(item[6])) if len(item) > 6 else 1
        _chk(x, BF16, "x"); _chk(dy, BF16, "dy"); _chk(target, F32, "target")
        B, H, W, Cin = x.shape
        if ups:
            H, W = 2 * H, 2 * W
        Cout = dy.shape[3]
        if tuple(dy.shape[:3]) != (B, H, W) or tuple(target.shape) != (Cout, Cin, K, K):
            raise RuntimeError(f"conv2d_wgrad_batch: shape mismatch x={tuple(x.shape)} dy={tuple(dy.shape)} "
                               f"dW={tuple(target.shape)} upsample={ups}")
        if _multi_ok(H, W, K) and len(items) > 1:
            multi.append((i, x, dy, B, H, W, Cin, Cout, bool(ups)))
            accs[i] = acc_flag
            continue
        if _multi_small_ok(H, W, K) and len(items) > 1:
            multi_small.append((i, x, dy, B, H, W, Cin, Cout, bool(ups)))
            accs[i] = acc_flag
            continue
        ws_bytes = lib.rgbd_conv2d_wgrad_workspace(B, H, W, Cin, Cout, K)
        if ws_bytes < 0:
            raise RuntimeError(f"conv2d_wgrad_batch: unsupported shape x={tuple(x.shape)} dy={tuple(dy.shape)} K={K}")
        ws = torch.empty(ws_bytes // 4, dtype=F32, device=x.device)
        keep.append(ws)
        flops = 2.0 * B * H * W * Cout * Cin * K * K
        nbytes = 2.0 * (x.numel() + dy.numel()) + 2.0 * ws_bytes
        rc = _timed(f"conv_wgrad_kernel<{K * K}>+reduce", flops, nbytes,
                    lambda: lib.rgbd_conv2d_wgrad_partial_bf16(_ptr(x), _ptr(dy), _ptr(ws), B, H, W, Cin, Cout, K,
                                                               int(bool(ups)), _stream()))
        _lib.check(rc, "rgbd_conv2d_wgrad_partial_bf16")
        tab[i] = (ws.data_ptr(), target.data_ptr(), ws_bytes // (4 * K * K * Cout * Cin), K * K, Cout, Cin, float(scale), acc_flag)
    groups = [multi[g0:g0 + WGRAD_MULTI_MAX] for g0 in range(0, len(multi), WGRAD_MULTI_MAX)] + \
             [multi_small[g0:g0 + WGRAD_MULTI_MAX] for g0 in range(0, len(multi_small), WGRAD_MULTI_MAX)]
    for group in groups:
        probs = (_WgradProblem * len(group))()
        for q, (i, x, dy, B, H, W, Cin, Cout, ups) in zip(probs, group):
            q.x, q.dy, q.workspace = x.data_ptr(), dy.data_ptr(), 0
            q.B, q.H, q.W, q.Cin, q.Cout, q.K, q.upsample, q.nsplit = B, H, W, Cin, Cout, 3, int(ups), 0
        _lib.check(lib.rgbd_conv2d_wgrad_multi_plan(probs, len(group), WGRAD_WORKGROUPS), "rgbd_conv2d_wgrad_multi_plan")
        sizes = [q.nsplit * 9 * q.Cout * q.Cin for q in probs]
        ws = torch.empty(sum(sizes), dtype=F32, device=group[0][1].device)
        keep.append(ws)
        off = flops = nbytes = 0
        for q, n, (i, x, dy, B, H, W, Cin, Cout, ups) in zip(probs, sizes, group):
            q.workspace = ws.data_ptr() + 4 * off
            off += n
            flops += 2.0 * B * H * W * Cout * Cin * 9
            nbytes += 2.0 * (x.numel() + dy.numel()) + 8.0 * n
            tab[i] = (q.workspace, items[i][2].data_ptr(), q.nsplit, 9, Cout, Cin, float(items[i][4]), accs[i])
        rc = _timed("conv_wgrad_kernel<9>+reduce", flops, nbytes,
                    lambda: lib.rgbd_conv2d_wgrad_partial_multi_bf16(probs, len(group), _stream()))
        _lib.check(rc, "rgbd_conv2d_wgrad_partial_multi_bf16")
    rc = lib.rgbd_wgrad_reduce_multi(tab.ctypes.data, len(items), _stream())
    _lib.check(rc, "rgbd_wgrad_reduce_multi")


def axpy_rows(a, x, s):
    """a + s[b] * x for (B, ...) bf16 tensors, s (B,) fp32."""
    _chk(a, BF16, "a"); _chk(x, BF16, "x"); _chk(s, F32, "s")
    if a.shape != x.shape or s.numel() != a.shape[0]:
        raise RuntimeError(f"axpy_rows: shapes {tuple(a.shape)} {tuple(x.shape)} {tuple(s.shape)}")
    out = torch.empty_like(a)
    rc = _lib.load().rgbd_axpy_rows_bf16(_ptr(a), _ptr(x), _ptr(s), _ptr(out), a.shape[0], a.numel() // a.shape[0],
                                         _stream())
    _lib.check(rc, "rgbd_axpy_rows_bf16")
    return out


def unpool2_lrelu_bwd(dp, y, shape, slope=0.2, bias_grad=None, row_scale=None, bias_grad2=None, emit_mx8=False):
    """dz (B,H,W,C) = 0.25 * upsample2(dp) * lrelu'(y); y may be None (plain average-pool backward).  bias_grad /
    row_scale as in lrelu_bwd; bias_grad2 receives the same sums as bias_grad (shortcut bias of a residual block)."""
    _chk(dp, BF16, "dp"); _chk(y, BF16, "y"); _chk(bias_grad, F32, "bias_grad"); _chk(row_scale, F32, "row_scale")
    _chk(bias_grad2, F32, "bias_grad2")
    B, H, W, C = shape
    dz = torch.empty(B, H, W, C, dtype=BF16, device=dp.device)
    dzq, dzs = _mx8_side(dz, emit_mx8)
    rc = _lib.load().rgbd_unpool2_lrelu_bwd(_ptr(dp), _ptr(y), _ptr(dz), B, H, W, C, float(slope), _ptr(bias_grad),
                                            _ptr(bias_grad2), _ptr(row_scale), _ptr(dzq), _ptr(dzs), _stream())
    _lib.check(rc, "rgbd_unpool2_lrelu_bwd")
    _mx8_attach(dz, dzq, dzs)
    return dz


def pool2_masked(x, y=None, slope=0.2):
    """out (B,H/2,W/2,C) = 0.25 * sum_{2x2} x * lrelu'(y); y None: plain 2x2 average pooling."""
    _chk(x, BF16, "x"); _chk(y, BF16, "y")
    B, H, W, C = x.shape
    out = torch.empty(B, H // 2, W // 2, C, dtype=BF16, device=x.device)
    rc = _lib.load().rgbd_pool2_masked(_ptr(x), _ptr(y), _ptr(out), B, H, W, C, float(slope), _stream())
    _lib.check(rc, "rgbd_pool2_masked")
    return out


def from_planes(x, w, bias, wscale, act, slope=0.2):
    """x (B,KP,H,W) fp32, w (C,KP) fp32 -> (B,H,W,C) bf16 = act(wscale * w x + bias)."""
    _chk(x, F32, "x"); _chk(w, F32, "w"); _chk(bias, F32, "bias")
    B, KP, H, W = x.shape
    C = w.shape[0]
    y = torch.empty(B, H, W, C, dtype=BF16, device=x.device)
    rc = _lib.load().rgbd_from_planes(_ptr(x), _ptr(w), _ptr(bias), _ptr(y), B, H * W, KP, C, float(wscale),
                                      int(bool(act)), float(slope), _stream())
    _lib.check(rc, "rgbd_from_planes")
    return y


def to_planes(h, w, bias, wscale):
    """h (B,H,W,C) bf16, w (KP,C) fp32 -> (B,KP,H,W) fp32 = wscale * w h + bias."""
    _chk(h, BF16, "h"); _chk(w, F32, "w"); _chk(bias, F32, "bias")
    B, H, W, C = h.shape
    KP = w.shape[0]
    out = torch.empty(B, KP, H, W, dtype=F32, device=h.device)
    rc = _lib.load().rgbd_to_planes(_ptr(h), _ptr(w), _ptr(bias), _ptr(out), B, H * W, KP, C, float(wscale), _stream())
    _lib.check(rc, "rgbd_to_planes")
    return out


def planes_outer(t, planes, want_tsum=False, psum=None):
    """t (B,H,W,C) bf16, planes (B,KP,H,W) fp32 -> o (KP,C) fp32 [, tsum (C,) fp32]; psum (KP,) fp32, if given, has the
    plane sums sum_{b,p} planes[b,k,p] ADDED to it."""
    _chk(t, BF16, "t"); _chk(planes, F32, "planes"); _chk(psum, F32, "psum")
    B, H, W, C = t.shape
    KP = planes.shape[1]
    o = torch.empty(KP, C, dtype=F32, device=t.device)
    ts = torch.empty(C, dtype=F32, device=t.device) if want_tsum else None
    rc = _lib.load().rgbd_planes_outer(_ptr(t), _ptr(planes), _ptr(o), _ptr(ts), _ptr(psum), B, H * W, KP, C, _stream())
    _lib.check(rc, "rgbd_planes_outer")
    return o, ts


# ------------------------------------------------------------------ small linears
def linear_fwd(x, w, bias, c, act, slope=0.2):
    """x (M,K), w (N,K) fp32 -> y (M,N) = act(c * x w^T + bias), M <= 64."""
    _chk(x, F32, "x"); _chk(w, F32, "w"); _chk(bias, F32, "bias")
    M, K = x.shape
    N = w.shape[0]
    y = torch.empty(M, N, dtype=F32, device=x.device)
    lib = _lib.load()
    nws = lib.rgbd_linear_fwd_workspace(M, K, N)
    ws = torch.empty(nws, dtype=F32, device=x.device) if nws > 0 else None
    rc = lib.rgbd_linear_fwd(_ptr(x), _ptr(w), _ptr(bias), _ptr(y), M, K, N, float(c), int(bool(act)),
                             float(slope), _ptr(ws), _stream())
    _lib.check(rc, "rgbd_linear_fwd")
    return y


def _ptr_array(tensors):
    """HOST array of device pointers (NULL for None) for the entry points that take `const float* const*`."""
    arr = (ctypes.c_void_p * len(tensors))()
    for i, t in enumerate(tensors):
        arr[i] = t.data_ptr() if t is not None else None
    return arr


def mlp_supported(M, C, L):
    return C in (256, 512) and 1 <= L <= 8 and M >= 1


def mlp_fwd(x, ws, bs, c, slope=0.2):
    """The chain  h <- lrelu(c * h W_l^T + b_l), l = 0..L-1  (mapping network, net.py:58-62) in ONE launch.
    x (M,C) fp32, ws L x (C,C), bs L x (C) -> acts (L,M,C): every layer's output (acts[-1] is the result)."""
    _chk(x, F32, "x")
    for t in list(ws) + list(bs):
        _chk(t, F32, "mlp parameter")
    M, C = x.shape
    L = len(ws)
    acts = torch.empty(L, M, C, dtype=F32, device=x.device)
    rc = _lib.load().rgbd_mlp_fwd(_ptr(x), _ptr_array(ws), _ptr_array(bs), L, M, C, float(c), float(slope), _ptr(acts),
                                  _stream())
    _lib.check(rc, "rgbd_mlp_fwd")
    return acts


def mlp_bwd(dy, x, acts, ws, c, dws=None, dbs=None, slope=0.2):
    """Backward of mlp_fwd in two launches: the dgrad chain (-> dx) and ONE weight / bias gradient launch for all layers, which
    ACCUMULATES into dws[l] (C,C) / dbs[l] (C) (None entries are skipped; dws None: no second launch)."""
    _chk(dy, F32, "dy"); _chk(x, F32, "x"); _chk(acts, F32, "acts")
    L, M, C = acts.shape
    dz = torch.empty_like(acts)
    dx = torch.empty(M, C, dtype=F32, device=dy.device)
    for t in (dws or []) + (dbs or []):
        _chk(t, F32, "mlp gradient")
    rc = _lib.load().rgbd_mlp_bwd(_ptr(dy), _ptr(x), _ptr(acts), _ptr_array(ws), _ptr_array(dws) if dws else None,
                                  _ptr_array(dbs) if dbs else None, L, M, C, float(c), float(slope), _ptr(dz), _ptr(dx),
                                  _stream())
    _lib.check(rc, "rgbd_mlp_bwd")
    return dx


def linear_fwd_masked(x, w, mask_y, c, slope=0.2):
    """(c * x w^T) * lrelu'(mask_y): x (M,K), w (N,K), mask_y (M,N) an activation OUTPUT -> (M,N)."""
    _chk(x, F32, "x"); _chk(w, F32, "w"); _chk(mask_y, F32, "mask_y")
    M, K = x.shape
    N = w.shape[0]
    y = torch.empty(M, N, dtype=F32, device=x.device)
    lib = _lib.load()
    nws = lib.rgbd_linear_fwd_workspace(M, K, N)
    ws = torch.empty(nws, dtype=F32, device=x.device) if nws > 0 else None
    rc = lib.rgbd_linear_fwd_masked(_ptr(x), _ptr(w), _ptr(mask_y), _ptr(y), M, K, N, float(c), float(slope), _ptr(ws),
                                    _stream())
    _lib.check(rc, "rgbd_linear_fwd_masked")
    return y


def linear_bwd(dy, y, x, w, c, act, want_dx=True, dw=None, db=None, slope=0.2):
    """Returns dx (or None); accumulates into dw / db when given.  x may be None when dw is."""
    for t, n in ((dy, "dy"), (y, "y"), (x, "x"), (w, "w"), (dw, "dw"), (db, "db")):
        _chk(t, F32, n)
    M = dy.shape[0]
    N, K = w.shape
    dx = torch.empty(M, K, dtype=F32, device=dy.device) if want_dx else None
    rc = _lib.load().rgbd_linear_bwd(_ptr(dy), _ptr(y), _ptr(x), _ptr(w), _ptr(dx), _ptr(dw), _ptr(db), M, K, N,
                                     float(c), int(bool(act)), float(slope), 0, _stream())
    _lib.check(rc, "rgbd_linear_bwd")
    return dx


# ------------------------------------------------------------------ AdaIN
def _adain_workspace(B, HW, C, device):
    """Strip-partial sums of the AdaIN kernels (ceil(HW/1024), B, C, 2): plain stores, summed in index order by the second
    launch -- no atomics and no clearing, so the statistics are bit-reproducible."""
    return torch.empty(_lib.load().rgbd_adain_workspace(B, HW, C), dtype=F32, device=device)


def _off(t, nfloats):
    return ctypes.c_void_p(t.data_ptr() + 4 * nfloats)


def adain_fwd(x, scale, shift=None, eps=1e-5, col_off=0, emit_mx8=False, c_live=None):
    """x (B,H,W,C) bf16; scale, shift (B,C) fp32 -- or shift None and scale = (B,Wtot) fp32 whose columns
    [col_off, col_off + 2C) hold [scale | shift] (the fused style-affine output, possibly of several style blocks)
    -> y, mean, rstd.  c_live < C (fused form only): channels [c_live, C) of x are zero padding, the window is
    [col_off, col_off + 2 c_live) and the padding channels of y stay zero."""
    _chk(x, BF16, "x"); _chk(scale, F32, "scale"); _chk(shift, F32, "shift")
    B, H, W, C = x.shape
    fused = shift is None
    live = C if c_live is None else int(c_live)
    if live != C and (not fused or live % 8 or not 0 < live < C):
        raise RuntimeError(f"adain_fwd: c_live={c_live} needs the fused form and a multiple of 8 below C={C}")
    if fused:
        if scale.dim() != 2 or scale.shape[0] != B or col_off % 4 or col_off + 2 * live > scale.shape[1]:
            raise RuntimeError(f"adain_fwd: window [{col_off},{col_off + 2 * live}) outside {tuple(scale.shape)}")
        ld = scale.shape[1]
    elif scale.shape != (B, C):
        raise RuntimeError(f"adain_fwd: scale {tuple(scale.shape)} does not match x {tuple(x.shape)}")
    y = torch.empty_like(x)
    sums = _adain_workspace(B, H * W, C, x.device)
    mean = torch.empty(B, C, dtype=F32, device=x.device)
    rstd = torch.empty(B, C, dtype=F32, device=x.device)
    yq, ysc = _mx8_side(y, emit_mx8)
    rc = _lib.load().rgbd_adain_fwd(_ptr(x), _off(scale, col_off) if fused else _ptr(scale),
                                    _off(scale, col_off + live) if fused else _ptr(shift), _ptr(y), _ptr(sums),
                                    _ptr(mean), _ptr(rstd), B, H * W, C, live, ld if fused else C, float(eps), _ptr(yq),
                                    _ptr(ysc), _stream())
    _lib.check(rc, "rgbd_adain_fwd")
    _mx8_attach(y, yq, ysc)
    return y, mean, rstd


def adain_bwd(x, dy, scale, mean, rstd, fused=False, col_off=0, out=None, lrelu_slope=0.0, bias_grad=None, emit_mx8=False,
              c_live=None):
    """-> dx, dscale, dshift; with fused (scale = (B,Wtot), window [col_off, col_off + 2C) = [scale | shift]):
    dx, d[scale | shift] written into the same window of `out` (B,Wtot) (allocated when None), None.
    lrelu_slope > 0: x is a leaky-ReLU output and dx also carries that activation's gradient; bias_grad (C) fp32 then
    accumulates the column sums of dx."""
    _chk(x, BF16, "x"); _chk(dy, BF16, "dy"); _chk(scale, F32, "scale"); _chk(out, F32, "out")
    _chk(bias_grad, F32, "bias_grad")
    B, H, W, C = x.shape
    live = C if c_live is None else int(c_live)       # (see adain_fwd: a window of 2 c_live columns, padding channels stay 0)
    if live != C and (not fused or live % 8 or not 0 < live < C):
        raise RuntimeError(f"adain_bwd: c_live={c_live} needs the fused form and a multiple of 8 below C={C}")
    dx = torch.empty_like(x)
    sums = _adain_workspace(B, H * W, C, x.device)
    if fused:
        ld = scale.shape[1]
        dss = out if out is not None else torch.empty(B, ld, dtype=F32, device=x.device)
        if dss.shape != scale.shape:
            raise RuntimeError("adain_bwd: gradient buffer shape mismatch")
        sc, dscale, dshift = _off(scale, col_off), _off(dss, col_off), _off(dss, col_off + live)
    else:
        ds = torch.empty(B, C, dtype=F32, device=x.device)
        dsh = torch.empty(B, C, dtype=F32, device=x.device)
        sc, dscale, dshift, ld = _ptr(scale), _ptr(ds), _ptr(dsh), C
    dxq, dxs = _mx8_side(dx, emit_mx8)
    rc = _lib.load().rgbd_adain_bwd(_ptr(x), _ptr(dy), sc, _ptr(mean), _ptr(rstd), _ptr(dx), dscale,
                                    dshift, _ptr(sums), B, H * W, C, live, ld, float(lrelu_slope), _ptr(bias_grad),
                                    _ptr(dxq), _ptr(dxs), _stream())
    _lib.check(rc, "rgbd_adain_bwd")
    _mx8_attach(dx, dxq, dxs)
    return (dx, dss, None) if fused else (dx, ds, dsh)


# ------------------------------------------------------------------ small pointwise ops
def pixelnorm(x, dy=None, eps=1e-8):
    """x (M,C) fp32 -> x * rsqrt(mean_c x^2 + eps); with dy: the input gradient instead."""
    _chk(x, F32, "x"); _chk(dy, F32, "dy")
    M, C = x.shape
    out = torch.empty_like(x)
    lib = _lib.load()
    if dy is None:
        rc = lib.rgbd_pixelnorm_fwd(_ptr(x), _ptr(out), M, C, float(eps), _stream())
    else:
        rc = lib.rgbd_pixelnorm_bwd(_ptr(x), _ptr(dy), _ptr(out), M, C, float(eps), _stream())
    _lib.check(rc, "rgbd_pixelnorm")
    return out


def gan_logit_heads(y):
    """y: logits (any shape, fp32) -> (losses (2,) = [mean softplus(-y), mean softplus(y)], seed_neg, seed_pos, ratio),
    the three vectors shaped like y: derivatives of the two means w.r.t. y (at max(y, -60)) and their quotient."""
    _chk(y, F32, "y")
    y = y.contiguous()
    losses = torch.empty(2, dtype=F32, device=y.device)
    sn, sp, ratio = torch.empty_like(y), torch.empty_like(y), torch.empty_like(y)
    rc = _lib.load().rgbd_gan_logit_heads(_ptr(y), y.numel(), _ptr(losses), _ptr(sn), _ptr(sp), _ptr(ratio), _stream())
    _lib.check(rc, "rgbd_gan_logit_heads")
    return losses, sn, sp, ratio


def softplus_mean(y, sign, gamma=0.0):
    """y: logits (any shape, fp32) -> (loss (1,) = mean softplus(sign y) sigmoid(sign y)^gamma, d loss / d y shaped like y)."""
    _chk(y, F32, "y")
    y = y.contiguous()
    loss = torch.empty(1, dtype=F32, device=y.device)
    dy = torch.empty_like(y)
    rc = _lib.load().rgbd_softplus_mean(_ptr(y), y.numel(), float(sign), float(gamma), _ptr(loss), _ptr(dy), _stream())
    _lib.check(rc, "rgbd_softplus_mean")
    return loss, dy


def depth_head_fwd(x):
    """x (B,4,H,W) fp32 -> [x0, x1, x2, 1 / (softplus(x3) + 1e-4)]."""
    _chk(x, F32, "x")
    B, C, H, W = x.shape
    if C != 4:
        raise RuntimeError(f"depth_head_fwd: expected 4 planes, got {C}")
    y = torch.empty_like(x)
    _lib.check(_lib.load().rgbd_depth_head_fwd(_ptr(x), _ptr(y), B, H * W, _stream()), "rgbd_depth_head_fwd")
    return y


def depth_head_bwd(x, y, dy):
    _chk(x, F32, "x"); _chk(y, F32, "y"); _chk(dy, F32, "dy")
    B, _, H, W = x.shape
    dx = torch.empty_like(x)
    _lib.check(_lib.load().rgbd_depth_head_bwd(_ptr(x), _ptr(y), _ptr(dy), _ptr(dx), B, H * W, _stream()),
               "rgbd_depth_head_bwd")
    return dx


def ema_update(dst, src, tau):
    """dst = (1 - tau) * dst + tau * src over flat fp32 buffers (soft_copy_param)."""
    _chk(dst, F32, "dst"); _chk(src, F32, "src")
    if dst.numel() != src.numel():
        raise RuntimeError("ema_update: size mismatch")
    _lib.check(_lib.load().rgbd_ema_update(_ptr(dst), _ptr(src), dst.numel(), float(tau), _stream()), "rgbd_ema_update")


# ------------------------------------------------------------------ fused step ops (csrc/step_ops.hip)
def real_batch(data_u8, idx, size, alpha=None, out=None):
    """data (N,C,H,W) uint8, idx (B) int64 on the device -> (B,C,size,size) fp32: block means of data[idx]/127.5 - 1
    (downsize_real, even stage); alpha (device scalar tensor or float): the fade-in blend of an odd stage."""
    if data_u8.dtype != torch.uint8 or not data_u8.is_cuda or not data_u8.is_contiguous():
        raise RuntimeError("real_batch: expected a contiguous uint8 GPU tensor")
    _chk(idx, torch.int64, "idx")
    N, C, H, W = data_u8.shape
    B = idx.numel()
    if out is None:
        out = torch.empty(B, C, size, size, dtype=F32, device=data_u8.device)
    _chk(out, F32, "out")
    if tuple(out.shape) != (B, C, size, size):
        raise RuntimeError(f"real_batch: out {tuple(out.shape)} != {(B, C, size, size)}")
    fade = alpha is not None
    a_dev = alpha if torch.is_tensor(alpha) else None
    _chk(a_dev, F32, "alpha")
    rc = _lib.load().rgbd_real_batch_u8(_ptr(data_u8), _ptr(idx), _ptr(out), B, C, H, W, int(size), int(fade), _ptr(a_dev),
                                        float(alpha) if fade and a_dev is None else 0.0, _stream())
    _lib.check(rc, "rgbd_real_batch_u8")
    return out


def zero_multi(tensors):
    """Clear up to 8 contiguous fp32 tensors per launch."""
    tensors = [t for t in tensors if t is not None and t.numel() > 0]
    lib = _lib.load()
    for i in range(0, len(tensors), 8):
        group = tensors[i:i + 8]
        for t in group:
            _chk(t, F32, "zero_multi")
        ptrs = (ctypes.c_void_p * len(group))(*[t.data_ptr() for t in group])
        counts = (ctypes.c_int64 * len(group))(*[t.numel() for t in group])
        _lib.check(lib.rgbd_zero_multi_f32(ptrs, counts, len(group), _stream()), "rgbd_zero_multi_f32")


def hidden_normalize(z, ch, copies=1):
    """z (M,C) fp32 N(0,1) draws -> (copies*M, C): z / sqrt(sum_c z^2 / ch + 1e-8), repeated `copies` times."""
    _chk(z, F32, "z")
    M, C = z.shape
    out = torch.empty(copies * M, C, dtype=F32, device=z.device)
    _lib.check(_lib.load().rgbd_hidden_normalize(_ptr(z), _ptr(out), M, C, float(ch), int(copies), _stream()),
               "rgbd_hidden_normalize")
    return out


def new_hidden_rng_state(device, seed=None):
    """A latent generator's state on `device`: uint32[4] {seed lo, seed hi, launch number, ticket} for rgbd_hidden_draw,
    seeded from `seed` or from torch's seed of this moment.  One per generator OBJECT, created at its first draw: a model built
    after torch.manual_seed(s) draws the same latent sequence every time, and the launch number advances on the device (also
    under graph replay)."""
    seed = (torch.initial_seed() if seed is None else int(seed)) & 0xFFFFFFFFFFFFFFFF
    return torch.tensor([seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF, 0, 0], dtype=torch.int64).to(torch.int32).to(device)


def hidden_draw(state, M, C, ch, copies=1):
    """make_hidden in one launch: (copies*M, C) fp32 = normalised N(0,1) rows (Philox + Box-Muller on the device), each row
    written `copies` times (rows m and m + M are the same latent); `state` from new_hidden_rng_state."""
    if state.dtype != torch.int32 or state.numel() != 4 or not state.is_cuda:
        raise RuntimeError("hidden_draw: state must be the int32[4] device tensor of new_hidden_rng_state")
    out = torch.empty(copies * M, C, dtype=F32, device=state.device)
    _lib.check(_lib.load().rgbd_hidden_draw(_ptr(state), _ptr(out), M, C, float(ch), int(copies), _stream()),
               "rgbd_hidden_draw")
    return out


def r1_penalty_fwd(g, coef):
    """g (B, ...) fp32 -> (1,) = coef * mean_b (sqrt(sum g_b^2))^2."""
    _chk(g, F32, "g")
    B = g.shape[0]
    ws = torch.empty(16 * B, dtype=F32, device=g.device)
    loss = torch.empty(1, dtype=F32, device=g.device)
    _lib.check(_lib.load().rgbd_r1_penalty_fwd(_ptr(g), B, g.numel() // B, float(coef), _ptr(ws), _ptr(loss), _stream()),
               "rgbd_r1_penalty_fwd")
    return loss


def axpy_rows_f32(a, x, s=None, out=None):
    """a + s[r] * x row by row (fp32, contiguous, trailing dimensions flattened); s None = 1; out=a accumulates in place."""
    _chk(a, F32, "a"); _chk(x, F32, "x"); _chk(s, F32, "s")
    if a.shape != x.shape:
        raise RuntimeError(f"axpy_rows_f32: shapes {tuple(a.shape)} vs {tuple(x.shape)}")
    rows = int(s.numel()) if s is not None else 1
    row_len = a.numel() // rows
    if rows * row_len != a.numel() or row_len % 4:
        raise RuntimeError(f"axpy_rows_f32: {a.numel()} elements do not split into {rows} rows of a multiple of 4")
    if out is None:
        out = torch.empty_like(a)
    _lib.check(_lib.load().rgbd_axpy_rows_f32(_ptr(a), _ptr(x), _ptr(s), _ptr(out), rows, row_len, _stream()),
               "rgbd_axpy_rows_f32")
    return out


def scale_by_scalar(x, scalar, k):
    """(scalar[0] * k) * x with `scalar` a device tensor (or None = 1)."""
    _chk(x, F32, "x"); _chk(scalar, F32, "scalar")
    out = torch.empty_like(x)
    _lib.check(_lib.load().rgbd_scale_by_scalar_f32(_ptr(x), _ptr(scalar), float(k), _ptr(out), x.numel(), _stream()),
               "rgbd_scale_by_scalar_f32")
    return out


def image_grad_init(gx, ratio, planes_out):
    """gx (B,KP,H,W) fp32, ratio (B,) or None -> (B,planes_out,H,W): ratio[b] * gx on the first KP planes, zeros after."""
    _chk(gx, F32, "gx"); _chk(ratio, F32, "ratio")
    B, KP, H, W = gx.shape
    out = torch.empty(B, planes_out, H, W, dtype=F32, device=gx.device)
    _lib.check(_lib.load().rgbd_image_grad_init(_ptr(gx), _ptr(ratio), _ptr(out), B, KP, planes_out, H * W, _stream()),
               "rgbd_image_grad_init")
    return out


def const_input_fwd(w, bias, B, slope=0.2):
    """w (C,H,W) [or (C,D,H,W): any trailing positions], bias (C) fp32 -> (B,H,W,C) bf16 = lrelu(w + bias) for every sample."""
    _chk(w, F32, "w"); _chk(bias, F32, "bias")
    C, pos = w.shape[0], tuple(w.shape[1:])
    out = torch.empty((B,) + pos + (C,), dtype=BF16, device=w.device)
    _lib.check(_lib.load().rgbd_const_input_fwd(_ptr(w), _ptr(bias), _ptr(out), B, int(np.prod(pos)), C, float(slope), _stream()),
               "rgbd_const_input_fwd")
    return out


def const_input_bwd(dh, w, bias, dw, db, slope=0.2):
    """Accumulates into dw (C,H,W) / db (C) (either may be None)."""
    _chk(dh, BF16, "dh"); _chk(w, F32, "w"); _chk(bias, F32, "bias"); _chk(dw, F32, "dw"); _chk(db, F32, "db")
    B, C = dh.shape[0], dh.shape[-1]
    _lib.check(_lib.load().rgbd_const_input_bwd(_ptr(dh), _ptr(w), _ptr(bias), _ptr(dw), _ptr(db), B, int(np.prod(dh.shape[1:-1])), C,
                                                float(slope), _stream()), "rgbd_const_input_bwd")


def _alpha_args(alpha):
    """alpha: 0-dim / 1-element fp32 device tensor, or a Python float -> (device pointer or NULL, host value)."""
    if torch.is_tensor(alpha):
        _chk(alpha.reshape(1), F32, "alpha")
        return _ptr(alpha), 0.0
    return ctypes.c_void_p(0), float(alpha)


def fade_planes_fwd(lo, hi, alpha):
    """(1-a) * upscale2x(lo) + a * hi on NCHW fp32 planes; lo (B,C,H/2,W/2), hi (B,C,H,W)."""
    _chk(lo, F32, "lo"); _chk(hi, F32, "hi")
    B, C, H, W = hi.shape
    if tuple(lo.shape) != (B, C, H // 2, W // 2):
        raise RuntimeError(f"fade_planes_fwd: shapes {tuple(lo.shape)} {tuple(hi.shape)}")
    out = torch.empty_like(hi)
    ap, ah = _alpha_args(alpha)
    _lib.check(_lib.load().rgbd_fade_planes_fwd(_ptr(lo), _ptr(hi), _ptr(out), B * C, H, W, ap, ah, _stream()),
               "rgbd_fade_planes_fwd")
    return out


def fade_planes_bwd(dout, alpha, want_lo=True, want_hi=True):
    _chk(dout, F32, "dout")
    B, C, H, W = dout.shape
    dlo = torch.empty(B, C, H // 2, W // 2, dtype=F32, device=dout.device) if want_lo else None
    dhi = torch.empty_like(dout) if want_hi else None
    ap, ah = _alpha_args(alpha)
    _lib.check(_lib.load().rgbd_fade_planes_bwd(_ptr(dout), _ptr(dlo), _ptr(dhi), B * C, H, W, ap, ah, _stream()),
               "rgbd_fade_planes_bwd")
    return dlo, dhi


def lerp_bf16(p, q, alpha):
    """(1-a) p + a q on bf16 tensors of one shape."""
    _chk(p, BF16, "p"); _chk(q, BF16, "q")
    if p.shape != q.shape:
        raise RuntimeError("lerp_bf16: shape mismatch")
    out = torch.empty_like(p)
    ap, ah = _alpha_args(alpha)
    _lib.check(_lib.load().rgbd_lerp_bf16(_ptr(p), _ptr(q), _ptr(out), None, p.numel(), 0, ap, ah, _stream()), "rgbd_lerp_bf16")
    return out


def lerp_split_bf16(g, alpha):
    """-> ((1-a) g, a g): the adjoint of lerp_bf16."""
    _chk(g, BF16, "g")
    o1, o2 = torch.empty_like(g), torch.empty_like(g)
    ap, ah = _alpha_args(alpha)
    _lib.check(_lib.load().rgbd_lerp_bf16(_ptr(g), None, _ptr(o1), _ptr(o2), g.numel(), 1, ap, ah, _stream()), "rgbd_lerp_bf16")
    return o1, o2


def pool2_planes(x, adjoint=False):
    """NCHW fp32: 2x2 average pooling (adjoint False) or its adjoint 0.25 * nearest upsampling (adjoint True)."""
    _chk(x, F32, "x")
    B, C, H, W = x.shape
    if adjoint:
        H, W = 2 * H, 2 * W
    out = torch.empty((B, C, H, W) if adjoint else (B, C, H // 2, W // 2), dtype=F32, device=x.device)
    _lib.check(_lib.load().rgbd_pool2_planes(_ptr(x), _ptr(out), B * C, H, W, int(bool(adjoint)), _stream()), "rgbd_pool2_planes")
    return out


def l2norm_fwd(x, eps=1e-5):
    """(.., C) bf16 -> x / (||x||_2 + eps) over the last dim (fp32 norm)."""
    _chk(x, BF16, "x")
    y = torch.empty_like(x)
    C = x.shape[-1]
    _lib.check(_lib.load().rgbd_l2norm_fwd(_ptr(x), _ptr(y), x.numel() // C, C, float(eps), _stream()), "rgbd_l2norm_fwd")
    return y


def l2norm_bwd(x, dy, eps=1e-5):
    _chk(x, BF16, "x"); _chk(dy, BF16, "dy")
    dx = torch.empty_like(x)
    C = x.shape[-1]
    _lib.check(_lib.load().rgbd_l2norm_bwd(_ptr(x), _ptr(dy), _ptr(dx), x.numel() // C, C, float(eps), _stream()),
               "rgbd_l2norm_bwd")
    return dx


def blur3x3(x, mode=0):
    """NHWC bf16.  mode 0: blur(x); mode 1: blur(upscale2x(x)) -> (B,2H,2W,C); mode 2: 2x2 sums of blur(x) -> (B,H/2,W/2,C)."""
    _chk(x, BF16, "x")
    B, H, W, C = x.shape
    if mode == 1:
        H, W = 2 * H, 2 * W
    out = torch.empty((B, H // 2, W // 2, C) if mode == 2 else (B, H, W, C), dtype=BF16, device=x.device)
    _lib.check(_lib.load().rgbd_blur3x3_bf16(_ptr(x), _ptr(out), B, H, W, C, int(mode), _stream()), "rgbd_blur3x3_bf16")
    return out


def nhwc_to_rows(h):
    """(B,H,W,C) bf16 -> (B, C*H*W) fp32 rows in (c,h,w) order."""
    _chk(h, BF16, "h")
    B, H, W, C = h.shape
    out = torch.empty(B, C * H * W, dtype=F32, device=h.device)
    _lib.check(_lib.load().rgbd_nhwc_to_rows_f32(_ptr(h), _ptr(out), B, H * W, C, _stream()), "rgbd_nhwc_to_rows_f32")
    return out


def rows_to_nhwc(rows, H, W, C):
    _chk(rows, F32, "rows")
    B = rows.shape[0]
    out = torch.empty(B, H, W, C, dtype=BF16, device=rows.device)
    _lib.check(_lib.load().rgbd_rows_to_nhwc_bf16(_ptr(rows), _ptr(out), B, H * W, C, _stream()), "rgbd_rows_to_nhwc_bf16")
    return out


# ------------------------------------------------------------------ optimizer
def adam_clip_multi(p, g, m, v, seg_begin, seg_alpha, beta1, beta2, eps, clip, grad_scale, step, workspace,
                    norm_out=None):
    """step: int32 device tensor (1,) = chainer's update counter t, incremented on the device by this call."""
    for t, n in ((p, "p"), (g, "g"), (m, "m"), (v, "v"), (workspace, "workspace")):
        _chk(t, F32, n)
    _chk(step, torch.int32, "step")
    n = p.numel()
    nseg = len(seg_alpha)
    sb = (ctypes.c_int64 * (nseg + 1))(*[int(s) for s in seg_begin])
    sa = (ctypes.c_float * nseg)(*[float(a) for a in seg_alpha])
    rc = _lib.load().rgbd_adam_clip_multi(_ptr(p), _ptr(g), _ptr(m), _ptr(v), n, nseg, sb, sa, float(beta1),
                                          float(beta2), float(eps), float(clip), float(grad_scale), _ptr(step),
                                          _ptr(workspace), _ptr(norm_out), _stream())
    _lib.check(rc, "rgbd_adam_clip_multi")


# ------------------------------------------------------------------ DeepVoxels layout folds
def fold_depth_taps(x, upsample_depth=False, adjoint_shape=None):
    """(B,D0,H,W,C) bf16 -> (B*D,H,W,3C) with the depth taps -1, 0, +1 folded into channels (D = 2 D0 behind a nearest depth
    repeat).  adjoint_shape = (B,D0,H,W,C): x is the gradient (B*D,H,W,3C), the result the gradient w.r.t. the input."""
    _chk(x, BF16, "x")
    lib = _lib.load()
    if adjoint_shape is None:
        B, D0, H, W, C = x.shape
        D = 2 * D0 if upsample_depth else D0
        y = torch.empty(B * D, H, W, 3 * C, dtype=BF16, device=x.device)
        adj = 0
    else:
        B, D0, H, W, C = adjoint_shape
        y = torch.empty(B, D0, H, W, C, dtype=BF16, device=x.device)
        adj = 1
    _lib.check(lib.rgbd_fold_depth_taps_bf16(_ptr(x), _ptr(y), B, D0, H, W, C, int(bool(upsample_depth)), adj, _stream()),
               "rgbd_fold_depth_taps_bf16")
    return y


def fold_4x4s2(x, adjoint_shape=None):
    """(B,H,W,C) bf16 -> (B,H/2,W/2,16C): the 16 taps of a 4x4 stride-2 pad-1 window folded into channels; adjoint_shape =
    (B,H,W,C): the backward gather."""
    _chk(x, BF16, "x")
    if adjoint_shape is None:
        B, H, W, C = x.shape
        y = torch.empty(B, H // 2, W // 2, 16 * C, dtype=BF16, device=x.device)
        adj = 0
    else:
        B, H, W, C = adjoint_shape
        y = torch.empty(B, H, W, C, dtype=BF16, device=x.device)
        adj = 1
    _lib.check(_lib.load().rgbd_fold_4x4s2_bf16(_ptr(x), _ptr(y), B, H, W, C, adj, _stream()), "rgbd_fold_4x4s2_bf16")
    return y


def pad_last(x, C1):
    """Last dimension C0 -> C1 (zero tail when C1 > C0, slice when C1 < C0); bf16 or fp32, contiguous."""
    if not x.is_cuda or not x.is_contiguous() or x.dtype not in (BF16, F32):
        raise RuntimeError("pad_last: expected a contiguous bf16 / fp32 GPU tensor")
    C0 = x.shape[-1]
    y = torch.empty(*x.shape[:-1], C1, dtype=x.dtype, device=x.device)
    _lib.check(_lib.load().rgbd_pad_last(_ptr(x), _ptr(y), x.numel() // C0, C0, C1, x.element_size(), _stream()), "rgbd_pad_last")
    return y


class _FoldDesc(ctypes.Structure):          # struct rgbd_fold_desc (include/rgbd_gan_hip.h)
    _fields_ = [("src", ctypes.c_void_p), ("dst", ctypes.c_void_p)] + \
               [(n, ctypes.c_int32) for n in ("mode", "Co", "Ci", "K", "Cop", "Cip", "adjoint", "reserved")]


def fold_weight_multi(items):
    """fold_weight for several layers in ONE launch.  items: (src, dst, mode, Co, Ci, K, Cop, Cip, adjoint) with dst the
    contiguous destination (folded buffer of a forward fold; master-shaped gradient, ACCUMULATED into, of an adjoint)."""
    if not items:
        return
    descs = (_FoldDesc * len(items))()
    for d, (src, dst, mode, Co, Ci, K, Cop, Cip, adjoint) in zip(descs, items):
        _chk(src, F32, "src"); _chk(dst, F32, "dst")
        folded = (Cop, 3 * Cip, 3, 3) if mode == 0 else (Cop, 16 * Cip, 1, 1) if mode == 1 else (Cop, Cip, K, K)
        master = (Co, Ci, 3, 3, 3) if mode == 0 else (Co, Ci, K, K)
        if tuple(src.shape) != (folded if adjoint else master) or tuple(dst.shape) != (master if adjoint else folded) \
                or not dst.is_contiguous():
            raise RuntimeError(f"fold_weight_multi: mode {mode}: src {tuple(src.shape)} dst {tuple(dst.shape)}")
        d.src, d.dst = src.data_ptr(), dst.data_ptr()
        d.mode, d.Co, d.Ci, d.K, d.Cop, d.Cip, d.adjoint = mode, Co, Ci, K, Cop, Cip, (2 if adjoint else 0)
    _lib.check(_lib.load().rgbd_fold_weight_multi_f32(descs, len(items), _stream()), "rgbd_fold_weight_multi_f32")


def fold_weight(src, mode, Co, Ci, K, Cop, Cip, adjoint=False, out=None):
    """mode 0: (Co,Ci,3,3,3) -> (Cop,3*Cip,3,3); 1: (Co,Ci,4,4) -> (Cop,16*Cip,1,1); 2: (Co,Ci,K,K) -> (Cop,Cip,K,K); fp32.
    adjoint: src is the folded weight's gradient, the result the master's; with `out` (master-shaped, contiguous) the adjoint is
    ADDED to it.  Forward with `out` (folded-shaped): written there."""
    _chk(src, F32, "src"); _chk(out, F32, "out")
    folded = (Cop, 3 * Cip, 3, 3) if mode == 0 else (Cop, 16 * Cip, 1, 1) if mode == 1 else (Cop, Cip, K, K)
    master = (Co, Ci, 3, 3, 3) if mode == 0 else (Co, Ci, K, K)
    if tuple(src.shape) != (folded if adjoint else master):
        raise RuntimeError(f"fold_weight: mode {mode} expects {folded if adjoint else master}, got {tuple(src.shape)}")
    if out is not None and (tuple(out.shape) != (master if adjoint else folded) or not out.is_contiguous()):
        raise RuntimeError("fold_weight: `out` is the contiguous destination: master-shaped accumulation target of the "
                           "adjoint, or the folded weight's (persistent) buffer of the forward fold")
    dst = out if out is not None else torch.empty(master if adjoint else folded, dtype=F32, device=src.device)
    flag = (2 if out is not None else 1) if adjoint else 0
    _lib.check(_lib.load().rgbd_fold_weight_f32(_ptr(src), _ptr(dst), mode, Co, Ci, K, Cop, Cip, flag, _stream()),
               "rgbd_fold_weight_f32")
    return dst
