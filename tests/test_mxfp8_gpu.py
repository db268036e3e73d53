"""MXFP8 convolution path (BASELINE configuration 5) against oracle/mxfp8.py, through the C ABI.  Needs an MI355X.

Tolerances.  The quantisers are compared BIT FOR BIT with the oracle (bytes and scale bytes).  The convolution accumulates
exact products in fp32 and stores bf16: against the oracle's float64 sum over the same dequantised operands the bound is one
bf16 rounding of the output (2^-8 relative) plus fp32 accumulation noise (1e-6 of the sum of magnitudes); against the bf16
engine on the unquantised operands it is the format's own noise, stated where asserted."""
import numpy as np
import pytest
import torch

from oracle import mxfp8

pytestmark = pytest.mark.gpu


def dev():
    return torch.device("cuda:0")


def _bf16(t):
    return t.to(torch.bfloat16)


@pytest.fixture(autouse=True)
def _mx8_on_small_problems(monkeypatch):
    from rgbd_gan_amd import kernels
    monkeypatch.setattr(kernels, "MX8_MIN_TILES", 0)


def _wide_range(shape, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(shape, generator=g) * torch.exp2(torch.randint(-12, 6, shape[:-1] + (shape[-1] // 32, 1), generator=g)
                                                     .float().expand(shape[:-1] + (shape[-1] // 32, 32)).reshape(shape))
    x[..., 5:9] = 0
    x[0, ..., :32] = 0                                  # an all-zero block
    x[-1, ..., 32:64] *= 1e-30                            # bf16 subnormals / underflow
    return _bf16(x)


@pytest.mark.parametrize("shape", [(3, 128), (2, 16, 16, 256), (1, 7, 5, 512)])
def test_quantize_is_bit_exact(shape):
    from rgbd_gan_amd import kernels
    x = _wide_range(shape, 3)
    q, s = kernels.quantize_mx8(x.to(dev()))
    rq, rs = mxfp8.quantize(x.float().numpy())
    assert np.array_equal(s.cpu().numpy(), rs)
    assert np.array_equal(q.cpu().numpy(), rq)
    # and the bytes mean what torch's float8_e4m3fn says they mean
    back = q.cpu().view(torch.float8_e4m3fn).float().numpy()
    assert np.array_equal(back, mxfp8.E4M3_DECODE[rq])


def test_quantize_propagates_nan():
    from rgbd_gan_amd import kernels
    x = torch.ones(2, 128, dtype=torch.bfloat16)
    x[1, 40] = float("nan")
    q, s = kernels.quantize_mx8(x.to(dev()))
    assert int(q[1, 40]) & 0x7F == 0x7F and int(q[1, 41]) == 0x78 and int(q[0, 40]) == 0x78     # 1.0 = 256 * 2^(119 - 127)
    assert int(s[0, 0]) == 119 and int(s[1, 1]) == 119


@pytest.mark.parametrize("co,ci", [(128, 128), (64, 256), (256, 64), (96, 160)])
def test_pack_weights_is_bit_exact(co, ci):
    from rgbd_gan_amd import kernels
    g = torch.Generator().manual_seed(co + ci)
    w = torch.randn(co, ci, 3, 3, generator=g) * torch.exp2(torch.randint(-3, 3, (co, 1, 1, 1), generator=g).float())
    scale = float(np.sqrt(2.0 / (ci * 9)))
    if ci % 128 and co % 128:
        # neither image is usable by the convolution kernels; the packer itself only needs multiples of 32
        tab = kernels.build_pack_table_mx8
        with pytest.raises(RuntimeError):
            tab([(w.to(dev()), scale, torch.empty(9, co, ci, dtype=torch.uint8, device=dev()),
                  torch.empty(9, co, ci // 32, dtype=torch.uint8, device=dev()), None, None)])
        return
    f, d = kernels.pack_weights_mx8(w.to(dev()), scale)
    (rfq, rfs), (rdq, rds) = mxfp8.pack_weights(w.numpy(), scale)
    if ci % 128 == 0:
        assert np.array_equal(f[1].cpu().numpy(), rfs) and np.array_equal(f[0].cpu().numpy(), rfq)
    else:
        assert f is None
    if co % 128 == 0:
        assert np.array_equal(d[1].cpu().numpy(), rds) and np.array_equal(d[0].cpu().numpy(), rdq)
    else:
        assert d is None


def _image(w, scale, fprop=True):
    from rgbd_gan_amd import kernels
    wd = w.to(dev())
    wf, wdg = kernels.pack_weights(wd, scale)
    f, d = kernels.pack_weights_mx8(wd, scale)
    pair, bf = (f, wf) if fprop else (d, wdg)
    return kernels.Mx8Image(bf, pair[0], pair[1])


# Accumulation: allowed error per unit of the products' summed magnitudes.  The block-scaled MFMA does not add its 128 products
# one fp32 rounding at a time: measured 4e-6 .. 8e-6 of the summed magnitudes on every shape below (about 2^-17: the products
# of an instruction are aligned to the largest one before they are summed), against ~2e-7 for a chain of fp32 adds.  Bound: 2e-5.
ACC_NOISE = 2e-5


def _close_to_ref(got, ref, mag):
    """one bf16 rounding of the output + fp32 accumulation noise (proportional to the summed magnitudes `mag` of the products)"""
    ref, mag = torch.as_tensor(ref), torch.as_tensor(mag)
    err = (got.double() - ref).abs()
    tol = ref.abs() * 2.0 ** -8 + ACC_NOISE * mag + 1e-30
    print(f"accumulation noise: worst |err| / mag where the bf16 rounding is negligible = "
          f"{float((err / mag.clamp_min(1e-30))[ref.abs() < 1e-3 * mag].max()):.2e}")
    bad = err > tol
    assert not bool(bad.any()), f"{int(bad.sum())} of {bad.numel()} outside one bf16 rounding; worst " \
                                f"{float((err / tol).max()):.2f} x tol"


MX_CASES = [
    # B, H, W, Cin, Cout, ups, bias, lrelu_ch, resid, pool      (layer shapes of the 256 px networks, small batches)
    (2, 32, 32, 128, 128, False, False, 0, False, False),
    (1, 32, 32, 256, 256, False, True, 256, False, False),      # two channel slices, wide tiles
    (4, 16, 16, 512, 64, False, True, 64, False, False),        # four slices, narrow tiles
    (1, 64, 64, 128, 64, False, True, 64, True, False),         # residual + lrelu, narrow
    (2, 32, 32, 256, 128, False, True, 128, True, True),        # D block main conv: residual, lrelu, pooled second output
    (2, 16, 16, 256, 128, True, True, 128, False, False),       # folded upsample 16 -> 32
    (1, 32, 32, 128, 64, True, False, 0, False, False),         # folded upsample 32 -> 64, narrow
    (1, 64, 64, 512, 512, False, True, 512, False, False),      # configuration 5's widest layer at its own size (four slices)
    (1, 128, 128, 256, 256, False, True, 256, True, False),     # ... and its 128^2 block conv with the residual (one image)
]


@pytest.mark.parametrize("case", MX_CASES)
def test_conv_fprop_mxfp8_matches_oracle(case):
    from rgbd_gan_amd import kernels, _lib
    B, H, W, Cin, Cout, ups, use_bias, lrelu_ch, use_res, pool = case
    g = torch.Generator().manual_seed(hash(case) % 1000)
    x = _bf16(torch.randn(B, H, W, Cin, generator=g) * torch.exp2(torch.randint(-2, 3, (B, H, W, 1), generator=g).float()))
    w = torch.randn(Cout, Cin, 3, 3, generator=g)
    scale = float(np.sqrt(2.0 / (Cin * 9)))
    bias = torch.randn(Cout, generator=g) if use_bias else None
    Ho, Wo = (2 * H, 2 * W) if ups else (H, W)
    res = _bf16(torch.randn(B, Ho, Wo, Cout, generator=g)) if use_res else None
    ref = torch.from_numpy(mxfp8.conv3x3_fprop_ref(x.float().numpy(), w.numpy(), scale, upsample=ups))
    mag = torch.from_numpy(mxfp8.conv3x3_fprop_ref(x.float().numpy(), w.numpy(), scale, upsample=ups, magnitude=True))
    if use_bias:
        ref = ref + bias.double()
    if use_res:
        ref = ref + res.double()
    if lrelu_ch:
        ref = torch.where(ref > 0, ref, 0.2 * ref)
    img = _image(w, scale)
    assert _lib.load().rgbd_conv3x3_mxfp8_supported(B, Ho, Wo, Cin, Cout)
    out = kernels.conv2d_fprop(x.to(dev()), img, 3, 3, 1, bias=bias.to(dev()) if use_bias else None,
                               residual=res.to(dev()) if use_res else None, upsample=ups, lrelu_channels=lrelu_ch,
                               avg_pool2=pool)
    assert b"mxfp8" in _lib.load().rgbd_last_conv_kernel()
    y = out[0] if pool else out
    _close_to_ref(y.float().cpu(), ref, mag)
    if pool:
        pooled = y.float().view(B, Ho // 2, 2, Wo // 2, 2, Cout).mean(dim=(2, 4)).to(torch.bfloat16)
        torch.testing.assert_close(out[1].float(), pooled.float(), atol=1e-2, rtol=1e-2)
    # against the bf16 engine on the unquantised operands: the format's noise.  Both operands carry 3 mantissa bits
    # (relative error uniform within +-2^-4 => rms 3.6 % each, largely independent per term): measured 3-5 % relative L2
    yb = kernels.conv2d_fprop(x.to(dev()), img.bf16, 3, 3, 1, bias=bias.to(dev()) if use_bias else None,
                              residual=res.to(dev()) if use_res else None, upsample=ups, lrelu_channels=lrelu_ch)
    rel = float((y.float() - yb.float()).norm() / yb.float().norm())
    assert rel < 8e-2, rel


@pytest.mark.parametrize("B,H,Cin,Cout,ups", [(2, 32, 64, 128, False), (1, 32, 128, 256, False), (2, 32, 128, 128, True),
                                               (1, 64, 64, 128, False)])
def test_conv_dgrad_mxfp8_matches_oracle(B, H, Cin, Cout, ups):
    """dgrad: blocks along Cout for dy and for the flipped image; `ups`: 2x2 sums of the input gradient (the adjoint of the
    upsample folded into the forward conv)."""
    from rgbd_gan_amd import kernels, _lib
    g = torch.Generator().manual_seed(B * 100 + Cin)
    dy = _bf16(torch.randn(B, H, H, Cout, generator=g) * 1e-3 * torch.exp2(torch.randint(-4, 4, (B, H, H, 1), generator=g).float()))
    w = torch.randn(Cout, Cin, 3, 3, generator=g)
    scale = float(np.sqrt(2.0 / (Cin * 9)))
    ref = torch.from_numpy(mxfp8.conv3x3_dgrad_ref(dy.float().numpy(), w.numpy(), scale))
    mag = torch.from_numpy(mxfp8.conv3x3_dgrad_ref(dy.float().numpy(), w.numpy(), scale, magnitude=True))
    if ups:
        ref = ref.view(B, H // 2, 2, H // 2, 2, Cin).sum(dim=(2, 4))
        mag = mag.view(B, H // 2, 2, H // 2, 2, Cin).sum(dim=(2, 4))
    img = _image(w, scale, fprop=False)
    dx = kernels.conv2d_dgrad(dy.to(dev()), img, 3, 1, sum_pool2=ups)
    assert b"mxfp8" in _lib.load().rgbd_last_conv_kernel()
    _close_to_ref(dx.float().cpu(), ref, mag)


def test_actgrad_and_stats_epilogues_on_mxfp8_operands():
    """The fused epilogues are the bf16 kernel's own code behind another main loop: the masked (activation-gradient) form and
    the statistics form must equal their compositions from the plain MXFP8 launch."""
    from rgbd_gan_amd import kernels, _lib
    g = torch.Generator().manual_seed(11)
    B, H, C = 2, 32, 128
    x = _bf16(torch.randn(B, H, H, C, generator=g)).to(dev())
    act = _bf16(torch.randn(B, H, H, C, generator=g)).to(dev())
    res = _bf16(torch.randn(B, H, H, C, generator=g)).to(dev())
    w = torch.randn(C, C, 3, 3, generator=g)
    scale = float(np.sqrt(2.0 / (C * 9)))
    img = _image(w, scale)
    plain = kernels.conv2d_fprop(x, img, 3, 3, 1, residual=res)
    bg = torch.zeros(C, device=dev())
    rs = torch.rand(B, generator=g).to(dev())
    y, y2 = kernels.conv3x3_actgrad(x, img, act, residual=res, bias_grad=bg, row_scale=rs, operand_scale=rs)
    assert b"actgrad,mxfp8" in _lib.load().rgbd_last_conv_kernel()
    # the fused epilogue masks the fp32 sum, the composition the bf16-rounded one: one rounding apart at most
    want = torch.where(act.float() > 0, plain.float(), 0.2 * plain.float())
    torch.testing.assert_close(y.float(), want.to(torch.bfloat16).float(), atol=2.0 ** -7 * float(want.abs().max()), rtol=0)
    cs = (y.float() * rs.view(B, 1, 1, 1)).sum(dim=(0, 1, 2))
    torch.testing.assert_close(bg, cs, atol=1e-3 * float(cs.abs().max()), rtol=1e-3)
    torch.testing.assert_close(y2.float(), (y.float() + rs.view(B, 1, 1, 1) * act.float()).to(torch.bfloat16).float(),
                               atol=2.0 ** -7 * float(y2.float().abs().max()), rtol=0)
    bias = torch.randn(C, generator=g).to(dev())
    ys, stats = kernels.conv2d_fprop_stats(x, img, bias, lrelu_channels=C)
    assert b"stats,mxfp8" in _lib.load().rgbd_last_conv_kernel()
    yp = kernels.conv2d_fprop(x, img, 3, 3, 1, bias=bias, lrelu_channels=C)
    assert torch.equal(ys, yp)
    s1 = stats[..., 0].double() / 2.0 ** 32
    s2 = stats[..., 1].double() / 2.0 ** 32
    torch.testing.assert_close(s1, ys.double().sum(dim=(1, 2)), atol=1e-3, rtol=1e-5)
    torch.testing.assert_close(s2, (ys.double() ** 2).sum(dim=(1, 2)), atol=1e-3, rtol=1e-5)


def test_mxfp8_launches_are_bit_reproducible_and_race_free():
    """The hand-ordered LDS-DMA staging of the MXFP8 main loop (data + scale pieces, counted waits): 24 launches of three
    shapes, with and without a host synchronisation in between, must give identical bytes."""
    from rgbd_gan_amd import kernels
    g = torch.Generator().manual_seed(5)
    for (B, H, Cin, Cout, ups) in [(4, 32, 256, 256, False), (8, 16, 512, 128, False), (2, 32, 128, 128, True)]:
        x = _bf16(torch.randn(B, H, H, Cin, generator=g)).to(dev())
        img = _image(torch.randn(Cout, Cin, 3, 3, generator=g), float(np.sqrt(2.0 / (Cin * 9))))
        first = kernels.conv2d_fprop(x, img, 3, 3, 1, upsample=ups).clone()
        for i in range(8):
            y = kernels.conv2d_fprop(x, img, 3, 3, 1, upsample=ups)
            if i % 2:
                torch.cuda.synchronize()
            assert torch.equal(y, first)


def test_mxfp8_adjoint_identity_at_benchmark_size():
    """<fprop(x; W), dy> = <x, dgrad(dy; W)> on operands that quantise exactly (small integers times powers of two), at the
    256 px network's own layer size (B = 16, 64 x 64, 512 -> 512: out of the CPU oracle's reach)."""
    from rgbd_gan_amd import kernels
    g = torch.Generator().manual_seed(9)
    B, H, C = 16, 64, 512
    ints = lambda *s: torch.randint(-3, 4, s, generator=g).float()
    x = _bf16(ints(B, H, H, C) * torch.exp2(torch.randint(-3, 3, (B, H, H, 1), generator=g).float())).to(dev())
    dy = _bf16(ints(B, H, H, C) * torch.exp2(torch.randint(-9, -5, (B, H, H, 1), generator=g).float())).to(dev())
    w = ints(C, C, 3, 3) * torch.exp2(torch.randint(-2, 2, (C, 1, 1, 1), generator=g).float())
    wdev = w.to(dev())
    wf, wdg = kernels.pack_weights(wdev, 0.125)
    f, d = kernels.pack_weights_mx8(wdev, 0.125)
    y = kernels.conv2d_fprop(x, kernels.Mx8Image(wf, *f), 3, 3, 1)
    dx = kernels.conv2d_dgrad(dy, kernels.Mx8Image(wdg, *d), 3, 1)
    lhs = float((y.double() * dy.double()).sum())
    rhs = float((x.double() * dx.double()).sum())
    bound = float(y.double().norm() * dy.double().norm())
    assert abs(lhs - rhs) < 2e-4 * bound, (lhs, rhs, bound)
    # exactly representable operands: the fp8 product sum equals the bf16 engine's up to the output rounding
    yb = kernels.conv2d_fprop(x, wf, 3, 3, 1)
    torch.testing.assert_close(y.float(), yb.float(), atol=2.0 ** -7 * float(yb.float().abs().max()), rtol=0)


@pytest.mark.parametrize("operands", ["bf16", "mxfp8"])
def test_epilogue_emits_the_quantisers_bytes(operands):
    """rgbd_conv3x3_ex: the MXFP8 copies a launch writes of its outputs (y; the pooled second output; the masked form's y) are
    bit for bit rgbd_quantize_mxfp8 of the bf16 tensors it stores, whatever the operand type of the launch itself -- and the
    bf16 outputs are the non-emitting launch's."""
    from rgbd_gan_amd import kernels
    g = torch.Generator().manual_seed(17)
    B, H, Cin, Cout = 2, 32, 128, 256
    x = _bf16(torch.randn(B, H, H, Cin, generator=g) * torch.exp2(torch.randint(-3, 3, (B, H, H, 1), generator=g).float())).to(dev())
    res = _bf16(torch.randn(B, H, H, Cout, generator=g)).to(dev())
    act = _bf16(torch.randn(B, H, H, Cout, generator=g)).to(dev())
    bias = torch.randn(Cout, generator=g).to(dev())
    img = _image(torch.randn(Cout, Cin, 3, 3, generator=g), float(np.sqrt(2.0 / (Cin * 9))))
    wp = img if operands == "mxfp8" else img.bf16

    def check(t):
        q, s, _ = t._mx8
        t2 = t.clone()
        rq, rs = kernels.quantize_mx8(t2)
        assert torch.equal(q, rq) and torch.equal(s, rs)
    def same(a, b):      # (a small bf16 problem without a copy to emit goes to the split-K gather kernel: another summation
        if operands == "mxfp8":      # order, one bf16 rounding apart; the MXFP8 launches are the same kernel either way)
            assert torch.equal(a, b)
        else:
            torch.testing.assert_close(a.float(), b.float(), atol=2.0 ** -7 * float(b.float().abs().max()), rtol=0)
    y = kernels.conv2d_fprop(x, wp, 3, 3, 1, bias=bias, lrelu_channels=Cout, emit_mx8=True)
    same(y, kernels.conv2d_fprop(x, wp, 3, 3, 1, bias=bias, lrelu_channels=Cout))
    check(y)
    y, yp = kernels.conv2d_fprop(x, wp, 3, 3, 1, bias=bias, residual=res, lrelu_channels=Cout, avg_pool2=True, emit_mx8=True)
    y0, yp0 = kernels.conv2d_fprop(x, wp, 3, 3, 1, bias=bias, residual=res, lrelu_channels=Cout, avg_pool2=True)
    same(y, y0), same(yp, yp0)
    assert getattr(y, "_mx8", None) is None
    check(yp)
    bg, bg0 = torch.zeros(Cout, device=dev()), torch.zeros(Cout, device=dev())
    dz = kernels.conv3x3_actgrad(x, wp, act, bias_grad=bg, emit_mx8=True)
    same(dz, kernels.conv3x3_actgrad(x, wp, act, bias_grad=bg0))
    torch.testing.assert_close(bg, bg0, rtol=1e-5, atol=1e-5 * float(bg0.abs().max()))
    check(dz)
    # the consumer takes the copy that rides on the tensor: no quantiser launch
    with kernels.launch_profile() as prof:
        kernels.conv2d_fprop(dz, _image(torch.randn(128, Cout, 3, 3, generator=g), 0.02), 3, 3, 1)
    assert "quantize_mx8_kernel" not in prof.summary()


def test_elementwise_producers_emit_the_quantisers_bytes():
    """The AdaIN apply pass (both forms), the AdaIN backward and the fused unpool + activation-gradient pass write the MXFP8
    copy of their bf16 output for the convolution behind them: bit for bit rgbd_quantize_mxfp8 of what they stored, and the
    bf16 tensors are those of the plain calls."""
    from rgbd_gan_amd import kernels
    g = torch.Generator().manual_seed(23)
    B, H, C = 2, 32, 256

    def check(t):
        q, s, _ = t._mx8
        rq, rs = kernels.quantize_mx8(t.clone())
        assert torch.equal(q, rq) and torch.equal(s, rs)
    x = _bf16(torch.randn(B, H, H, C, generator=g) * 3).to(dev())
    ss = torch.randn(B, 2 * C, generator=g).to(dev())
    y, mean, rstd = kernels.adain_fwd(x, ss, emit_mx8=True)
    y0, _, _ = kernels.adain_fwd(x, ss)
    assert torch.equal(y, y0)
    check(y)
    dy = _bf16(torch.randn(B, H, H, C, generator=g) * 1e-3).to(dev())
    for slope in (0.0, 0.2):
        dx, _, _ = kernels.adain_bwd(x, dy, ss, mean, rstd, fused=True, lrelu_slope=slope, emit_mx8=True)
        dx0, _, _ = kernels.adain_bwd(x, dy, ss, mean, rstd, fused=True, lrelu_slope=slope)
        assert torch.equal(dx, dx0)
        check(dx)
    dp = _bf16(torch.randn(B, H // 2, H // 2, C, generator=g) * 1e-2).to(dev())
    bg = torch.zeros(C, device=dev())
    dz = kernels.unpool2_lrelu_bwd(dp, x, (B, H, H, C), bias_grad=bg, emit_mx8=True)
    assert torch.equal(dz, kernels.unpool2_lrelu_bwd(dp, x, (B, H, H, C)))
    check(dz)
    wf, _ = kernels.pack_weights(torch.randn(C, C, 3, 3, generator=g).to(dev()), 0.02)
    yc, stats = kernels.conv2d_fprop_stats(x, wf, torch.zeros(C, device=dev()), lrelu_channels=C)
    out, _, _ = kernels.adain_apply_fixed(yc, stats, ss, emit_mx8=True)
    assert torch.equal(out, kernels.adain_apply_fixed(yc, stats, ss)[0])
    check(out)
