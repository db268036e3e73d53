"""Known answers that pin oracle/mxfp8.py (the MXFP8 operand format of the fp8 convolution path) on the CPU box."""
import numpy as np
import torch

from oracle import mxfp8


def test_every_code_round_trips_and_ties_go_to_even():
    codes = np.arange(256, dtype=np.uint8)
    vals = mxfp8.E4M3_DECODE[codes]
    finite = ~np.isnan(vals)
    assert finite.sum() == 254 and np.isnan(vals[0x7F]) and np.isnan(vals[0xFF])
    assert vals[0x7E] == 448.0 and vals[0x01] == 2.0 ** -9 and vals[0x08] == 2.0 ** -6 and vals[0x38] == 1.0
    back = mxfp8.e4m3_encode(vals[finite])
    keep = codes[finite] != 0x80                          # -0 encodes as 0x80 too
    assert np.array_equal(back[keep], codes[finite][keep])
    # midpoints between neighbours go to the even code
    pos = np.arange(0, 0x7E, dtype=np.uint8)
    mid = (vals[pos].astype(np.float64) + vals[pos + 1].astype(np.float64)) / 2
    enc = mxfp8.e4m3_encode(mid.astype(np.float32))
    want = np.where(pos % 2 == 0, pos, pos + 1)
    assert np.array_equal(enc, want)
    assert mxfp8.e4m3_encode(np.float32([448.0, -448.0, 0.0, 2.0 ** -10, 1.5 * 2.0 ** -10])).tolist() == [0x7E, 0xFE, 0, 0, 1]
    assert mxfp8.e4m3_encode(np.float32([np.nan]))[0] == 0x7F


def test_encode_agrees_with_torch_float8_e4m3fn():
    rng = np.random.RandomState(0)
    x = np.concatenate([rng.uniform(-448, 448, 20000), rng.normal(0, 1, 20000), rng.normal(0, 0.01, 20000),
                        mxfp8.E4M3_DECODE[np.arange(0x7F)]]).astype(np.float32)
    got = mxfp8.e4m3_encode(x)
    ref = torch.from_numpy(x).to(torch.float8_e4m3fn).view(torch.uint8).numpy()
    assert np.array_equal(got, ref)


def test_block_scale_known_answers():
    # amax = m * 2^k -> s = k + 127 - 8 (+ 1 when m > 1.75): the scaled block tops out in (224, 448]
    for amax, s in ((1.0, 119), (1.75, 119), (1.76, 120), (1.99, 120), (2.0, 120), (448.0, 127), (449.0, 128), (300.0, 127),
                    (512.0, 128), (2.0 ** -20, 99), (0.0, 0), (2.0 ** -126, 0), (2.0 ** -119, 0), (2.0 ** -118, 1)):
        assert int(mxfp8.block_scale(np.float32(amax))) == s, (amax, s)


def test_quantize_blocks_are_independent_and_never_saturate():
    x = np.zeros((2, 64), dtype=np.float32)
    x[0, :32] = np.linspace(-1, 1, 32)
    x[0, 32:] = 1000.0 * np.linspace(-1, 1, 32)
    x[1, 5] = 3.0
    x[1, 40] = 480.0                     # amax 480 = 1.875 * 2^8 -> s = 128: 240 * 2, exact
    q, s = mxfp8.quantize(x)
    assert s.tolist() == [[119, 129], [120, 128]]
    d = mxfp8.dequantize(q, s)
    assert d[1, 5] == 3.0 and d[1, 40] == 480.0 and np.all(d[1, :5] == 0)
    assert np.abs(mxfp8.E4M3_DECODE[q]).max() <= 448.0
    # relative error of every non-tiny element <= 2^-4 (3 mantissa bits): nothing saturates
    big = np.abs(x[0]) > np.abs(x[0]).reshape(2, 32).max(1).repeat(32) / 16
    assert np.all(np.abs(d[0][big] - x[0][big]) <= np.abs(x[0][big]) * 2.0 ** -4 + 1e-12)


def test_pack_weights_layout_and_refs_are_adjoint():
    rng = np.random.RandomState(1)
    co, ci = 128, 128
    w = rng.normal(size=(co, ci, 3, 3)).astype(np.float32)
    scale = np.float32(np.sqrt(2.0 / (ci * 9)))
    (fq, fs), (dq, ds) = mxfp8.pack_weights(w, scale)
    assert fq.shape == (9, co, ci) and fs.shape == (9, co, ci // 32) and dq.shape == (9, ci, co) and ds.shape == (9, ci, co // 32)
    wf = mxfp8.dequantize(fq, fs)
    wd = mxfp8.dequantize(dq, ds)
    ws = (w * scale).reshape(co, ci, 9)
    assert np.abs(wf[4] - ws[:, :, 4]).max() <= np.abs(ws).max() * 2.0 ** -4
    assert np.abs(wd[8 - 2].T - ws[:, :, 2]).max() <= np.abs(ws).max() * 2.0 ** -4
    # with operands that quantise exactly (small integers, few nonzeros) fprop and dgrad are exact adjoints
    w2 = np.round(rng.normal(size=(co, ci, 3, 3))).astype(np.float32)
    x = np.round(rng.normal(size=(1, 16, 16, ci))).astype(np.float32)
    dy = np.round(rng.normal(size=(1, 16, 16, co))).astype(np.float32)
    y = mxfp8.conv3x3_fprop_ref(x, w2, 1.0)
    dx = mxfp8.conv3x3_dgrad_ref(dy, w2, 1.0)
    assert abs(float((y * dy).sum()) - float((x * dx).sum())) < 1e-6 * float(np.abs(y * dy).sum())
