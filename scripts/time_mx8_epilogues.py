"""The fused-epilogue forms of the MXFP8 kernel (activation gradient, instance-norm statistics) with 128- and 64-channel
output tiles, beside the plain form and the bf16 kernels: the wide fused forms spill (B-row ring + epilogue state > 256
VGPRs), is the narrow tile faster?  Uses the debug library's planner switch (variant 2 = 64-channel tiles everywhere)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rgbd_gan_amd import _lib, kernels

lib = _lib.debug_library().__enter__()
kernels.MX8_MIN_TILES = 0
B = int(os.environ.get("B", "16"))
dev = "cuda:0"
def timeit(fn, n=10):
    for _ in range(3): fn()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for H, Cin, Cout in [(64, 512, 512), (64, 256, 256), (128, 256, 256), (128, 128, 128), (256, 128, 128)]:
    x = torch.randn(B, H, H, Cin, device=dev).to(torch.bfloat16)
    act = torch.randn(B, H, H, Cout, device=dev).to(torch.bfloat16)
    w = torch.randn(Cout, Cin, 3, 3, device=dev)
    bias = torch.zeros(Cout, device=dev); bg = torch.zeros(Cout, device=dev)
    scale = float(np.sqrt(2.0 / (Cin * 9)))
    wf, _ = kernels.pack_weights(w, scale)
    f, _ = kernels.pack_weights_mx8(w, scale)
    img = kernels.Mx8Image(wf, *f)
    kernels.quantize_mx8(x)                      # cached on x: the timings below are the conv launches alone
    fl = 2.0 * B * H * H * Cin * Cout * 9
    row = f"H={H:4d} {Cin:4d}->{Cout:4d}"
    for name, fn in (("plain", lambda im: kernels.conv2d_fprop(x, im, 3, 3, 1, bias=bias, lrelu_channels=Cout)),
                     ("actgrad", lambda im: kernels.conv3x3_actgrad(x, im, act, bias_grad=bg)),
                     ("stats", lambda im: kernels.conv2d_fprop_stats(x, im, bias, lrelu_channels=Cout))):
        res = []
        for variant in (0, 2):
            lib.rgbd_debug_conv_variant(variant)
            tb = timeit(lambda: fn(wf)); tm = timeit(lambda: fn(img))
            res.append(f"{'wide' if variant == 0 else 'narrow'}: bf16 {fl / tb / 1e6:5.0f} mx {fl / tm / 1e6:5.0f} TF")
        lib.rgbd_debug_conv_variant(0)
        row += f" | {name:7s} " + ", ".join(res)
    print(row)
