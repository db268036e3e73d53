"""CPU restatement of the MXFP8 operand format of the fp8 convolution path (TEST INFRASTRUCTURE ONLY: nothing under
rgbd_gan_amd/ imports this; tests/, __graft_entry__.smoke() and bench.py's checker legs do).

What it restates: rgbd_gan_amd/csrc/mxfp8.hip (rgbd_quantize_mxfp8, rgbd_pack_weights_mxfp8_multi) and the arithmetic of the
block-scaled matrix instruction the convolution kernels use (rgbd_gan_amd/csrc/conv.hip, conv3x3_sp_kernel<.., MX>):

    OCP microscaling, element type E4M3 (e4m3fn: bias 7, no infinities, 0x7f / 0xff = NaN, largest finite 448), one E8M0
    scale byte per 32 consecutive elements of the reduction index:
        value = e4m3(q) * 2^(s - 127)
        s     = max(E - 8 + (mantissa > 1.75), 0),  E / mantissa = biased fp32 exponent / significand of the block's amax
        q     = e4m3_rne(clamp(x * 2^(127 - s), -448, 448))          (round to nearest even; amax lands in (224, 448])
    and a product sum over dequantised values accumulated in fp32 (every product of two scaled e4m3 numbers is exact in fp32,
    so the only rounding is the accumulation's and the final bf16 store).

The reference has no fp8 path -- the 256x256 blocks are commented out at /root/reference/net.py:181-183,192-194 and it
computes in fp32 throughout -- so this module is pinned by the format's own known answers (tests/test_mxfp8_cpu.py: every
e4m3 code decodes and re-encodes to itself, ties go to even, saturation, torch.float8_e4m3fn agrees on every representable
value and midpoint) and, through tests/test_mxfp8_gpu.py, compared bit for bit with what the HIP kernels emit.
"""
import numpy as np

BLOCK = 32
E4M3_MAX = 448.0


def _decode_table():
    t = np.zeros(256, dtype=np.float32)
    for c in range(256):
        s, e, m = c >> 7, (c >> 3) & 15, c & 7
        if e == 15 and m == 7:
            v = np.nan
        elif e == 0:
            v = m * 2.0 ** -9
        else:
            v = (1.0 + m / 8.0) * 2.0 ** (e - 7)
        t[c] = -v if s else v
    return t


E4M3_DECODE = _decode_table()


def e4m3_encode(v):
    """float32 array, already clamped to [-448, 448] (NaN allowed) -> uint8 e4m3 codes, round to nearest even."""
    v = np.asarray(v, dtype=np.float32)
    a = np.abs(np.where(np.isnan(v), np.float32(0), v)).astype(np.float64)
    sign = (np.signbit(v)).astype(np.uint8) << 7
    with np.errstate(divide="ignore", invalid="ignore"):
        e = np.floor(np.log2(np.where(a > 0, a, 1.0))).astype(np.int64)
    e = np.clip(e, -6, 8)
    # subnormal range (a < 2^-6): steps of 2^-9; normal: 8 steps per binade
    sub = a < 2.0 ** -6
    m_sub = np.rint(a * 2.0 ** 9)                                   # 0..8 (8 = the smallest normal, code 0x08)
    m_nor = np.rint((a / np.exp2(e.astype(np.float64)) - 1.0) * 8.0)   # 0..8
    carry = m_nor == 8
    e_n = np.where(carry, e + 1, e)
    m_n = np.where(carry, 0, m_nor)
    code = np.where(sub, m_sub, ((e_n + 7) * 8 + m_n)).astype(np.int64)
    code = np.minimum(code, 0x7E)
    out = (code.astype(np.uint8) | sign).astype(np.uint8)
    return np.where(np.isnan(v), np.uint8(0x7F), out).astype(np.uint8)


def block_scale(amax):
    """float32 block maxima -> E8M0 bytes."""
    bits = np.asarray(amax, dtype=np.float32).view(np.uint32)
    E = ((bits >> 23) & 0xFF).astype(np.int64) + ((bits & 0x7FFFFF) > 0x600000)
    return np.maximum(E - 8, 0).astype(np.uint8)


def quantize(x):
    """x (..., C) float32 (values as the kernel sees them: bf16-representable activations, fp32 inv_c * W), C % 32 == 0
    -> (q (..., C) uint8, s (..., C // 32) uint8), blocks of 32 along the last axis."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    C = x.shape[-1]
    assert C % BLOCK == 0
    xb = x.reshape(x.shape[:-1] + (C // BLOCK, BLOCK))
    amax = np.max(np.abs(np.where(np.isnan(xb), 0, xb)), axis=-1)
    s = block_scale(amax)
    inv = np.ldexp(np.float32(1.0), 127 - s.astype(np.int64)).astype(np.float32)
    with np.errstate(over="ignore", under="ignore"):
        v = (xb * inv[..., None]).astype(np.float32)
    v = np.where(v > E4M3_MAX, np.float32(E4M3_MAX), v)
    v = np.where(v < -E4M3_MAX, np.float32(-E4M3_MAX), v)
    return e4m3_encode(v).reshape(x.shape), s


def dequantize(q, s):
    """-> float32 (..., C)."""
    q = np.asarray(q, dtype=np.uint8)
    C = q.shape[-1]
    v = E4M3_DECODE[q].reshape(q.shape[:-1] + (C // BLOCK, BLOCK))
    sc = np.ldexp(np.float32(1.0), np.asarray(s, dtype=np.int64) - 127).astype(np.float32)
    return (v * sc[..., None]).reshape(q.shape).astype(np.float32)


def fake_quantize(x):
    """quantize + dequantize: the values the matrix instruction multiplies."""
    return dequantize(*quantize(x))


def pack_weights(w, scale):
    """Master weights (Cout, Cin, 3, 3) fp32 -> the two MXFP8 images of csrc/mxfp8.hip:
        fprop: q [9][Cout][Cin], s [9][Cout][Cin/32]              (blocks along Cin; None unless Cin % 32 == 0)
        dgrad: q [9][Cin][Cout], s [9][Cin][Cout/32], taps flipped (blocks along Cout)
    `scale` (inv_c) is multiplied in fp32 first, as the kernel does."""
    w = np.asarray(w, dtype=np.float32)
    co, ci = w.shape[:2]
    ws = (w * np.float32(scale)).astype(np.float32)
    taps = ws.reshape(co, ci, 9)
    f = quantize(np.ascontiguousarray(taps.transpose(2, 0, 1)))                          # [tap][co][ci]
    d = quantize(np.ascontiguousarray(taps[:, :, ::-1].transpose(2, 1, 0)))              # [8 - tap][ci][co]
    return f, d


def conv3x3_fprop_ref(x_nhwc, w, scale, upsample=False, magnitude=False):
    """The product sum the MXFP8 fprop kernel accumulates, in float64: x (B,H,W,Cin) values as stored (bf16-representable),
    w (Cout,Cin,3,3) fp32 master.  -> (B,Ho,Wo,Cout) float64 = conv3x3_pad1(dequant(quant(x)) [nearest 2x], dequant(quant(w))).
    magnitude: the sum of the products' magnitudes instead (what fp32 accumulation noise is proportional to)."""
    import torch
    import torch.nn.functional as F
    x = np.asarray(x_nhwc, dtype=np.float32)
    xq = fake_quantize(x)                                                                  # blocks along channels
    (fq, fs), _ = pack_weights(w, scale)
    co, ci = w.shape[:2]
    wq = dequantize(fq, fs).reshape(3, 3, co, ci).transpose(2, 3, 0, 1)                    # (co, ci, kh, kw)
    if magnitude:
        xq, wq = np.abs(xq), np.abs(wq)
    xt = torch.from_numpy(xq).permute(0, 3, 1, 2).double()
    if upsample:
        xt = xt.repeat_interleave(2, dim=2).repeat_interleave(2, dim=3)
    y = F.conv2d(xt, torch.from_numpy(np.ascontiguousarray(wq)).double(), None, padding=1)
    return y.permute(0, 2, 3, 1).numpy()


def conv3x3_dgrad_ref(dy_nhwc, w, scale, magnitude=False):
    """The MXFP8 dgrad product sum in float64: dy (B,H,W,Cout), blocks along Cout for both operands."""
    import torch
    import torch.nn.functional as F
    dy = fake_quantize(np.asarray(dy_nhwc, dtype=np.float32))
    _, (dq, ds) = pack_weights(w, scale)
    co, ci = w.shape[:2]
    wd = dequantize(dq, ds).reshape(3, 3, ci, co).transpose(2, 3, 0, 1)                    # (ci, co, kh', kw'), taps flipped
    if magnitude:
        dy, wd = np.abs(dy), np.abs(wd)
    y = F.conv2d(torch.from_numpy(dy).permute(0, 3, 1, 2).double(), torch.from_numpy(np.ascontiguousarray(wd)).double(), None,
                 padding=1)
    return y.permute(0, 2, 3, 1).numpy()
