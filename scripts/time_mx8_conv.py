"""MXFP8 against bf16 on the 3x3 layer shapes of the 256x256 networks (ch = 512, per-GPU batch 16): HIP-event timing of the
conv launch alone, of the activation quantiser alone, and of both (what a layer costs until the quantiser is fused into the
producer's epilogue)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from rgbd_gan_amd import _lib, kernels

B = int(os.environ.get("B", "16"))
dev = "cuda:0"
# H, Cin, Cout, upsample        (generator c0 / c1 and discriminator c0 / c1 / c_sc of the blocks >= 16x16)
shapes = [(16, 512, 512, False), (32, 512, 512, False), (32, 512, 512, True), (64, 512, 512, False), (64, 512, 256, False),
          (64, 256, 256, False), (128, 256, 256, False), (128, 256, 128, False), (128, 128, 128, False), (256, 128, 128, False),
          (256, 128, 64, False), (64, 512, 256, True), (128, 256, 128, True)]


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


lib = _lib.load()
print(f"B={B}")
for H, Cin, Cout, ups in shapes:
    Hi = H // 2 if ups else H
    x = torch.randn(B, Hi, Hi, Cin, device=dev).to(torch.bfloat16)
    w = torch.randn(Cout, Cin, 3, 3, device=dev)
    bias = torch.zeros(Cout, device=dev)
    scale = float(np.sqrt(2.0 / (Cin * 9)))
    wf, _ = kernels.pack_weights(w, scale)
    f, _ = kernels.pack_weights_mx8(w, scale)
    xq, xs = kernels.quantize_mx8(x)
    y = torch.empty(B, H, H, Cout, dtype=torch.bfloat16, device=dev)
    st = kernels._stream
    fl = 2.0 * B * H * H * Cin * Cout * 9
    tb = timeit(lambda: kernels.conv2d_fprop(x, wf, 3, 3, 1, bias=bias, lrelu_channels=Cout, upsample=ups))
    tm = timeit(lambda: _lib.check(lib.rgbd_conv2d_fprop_mxfp8(kernels._ptr(xq), kernels._ptr(xs), kernels._ptr(f[0]),
                                                               kernels._ptr(f[1]), kernels._ptr(bias), None, kernels._ptr(y),
                                                               None, B, Hi, Hi, Cin, Cout, int(ups), Cout, 0.2, 0, st()), "mx"))
    tq = timeit(lambda: (setattr(x, "_mx8", None), kernels.quantize_mx8(x)))
    print(f"H={H:4d} {Cin:4d}->{Cout:4d}{' ups' if ups else '    '}  bf16 {tb:7.1f} us {fl / tb / 1e6:6.0f} TF | mxfp8 {tm:7.1f} us "
          f"{fl / tm / 1e6:6.0f} TF ({tb / tm:4.2f}x) | quantise {tq:6.1f} us {3.03 * x.numel() / tq / 1e3:5.0f} GB/s | "
          f"conv+quantise {tb / (tm + tq):4.2f}x")
w = [torch.randn(512, 512, 3, 3, device=dev) for _ in range(12)]
ent = []
for t in w:
    ent.append((t, 0.02, torch.empty(9, 512, 512, dtype=torch.uint8, device=dev), torch.empty(9, 512, 16, dtype=torch.uint8, device=dev),
                torch.empty(9, 512, 512, dtype=torch.uint8, device=dev), torch.empty(9, 512, 16, dtype=torch.uint8, device=dev)))
tab = kernels.build_pack_table_mx8(ent)
tp = timeit(lambda: kernels.pack_weights_mx8_multi(tab))
print(f"pack_weights_mx8_multi, 12 x (512,512,3,3): {tp:.1f} us ({12 * 512 * 512 * 9 * 6.1 / tp / 1e3:.0f} GB/s)")
