"""A/B the 3x3 weight-gradient kernels on the layer shapes of the 128x128 stage in ONE process: variant 0 = default
(all taps per wave, LDS-DMA), variant 3 = the tap-split register-staged body.  Times partial + reduction together."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rgbd_gan_amd import kernels, _lib

B = int(os.environ.get("B", "32"))
REPS = int(os.environ.get("REPS", "20"))
lib = _lib.debug_library().__enter__()     # the A/B reference kernels live in the debug library (build --debug)
shapes = [(128, 64, 64, 0), (128, 64, 128, 0), (128, 128, 128, 0), (64, 128, 256, 0), (64, 256, 256, 0), (32, 256, 256, 0),
          (16, 256, 256, 0), (64, 256, 128, 1), (128, 128, 64, 1)]       # (H, Cin, Cout, x is read through the 2x upsample)
for H, Cin, Cout, ups in shapes:
    Hx = H // 2 if ups else H
    x = torch.randn(B, Hx, Hx, Cin, device="cuda").to(torch.bfloat16)
    dy = torch.randn(B, H, H, Cout, device="cuda").to(torch.bfloat16)
    flops = 2.0 * B * H * H * Cout * Cin * 9
    outs, line = {}, f"H={H:3d} Cin={Cin:3d} Cout={Cout:3d} ups={ups}:"
    for rnd in range(2):
        for v in [int(t) for t in os.environ.get("VARIANTS", "3,0").split(",")]:
            lib.rgbd_debug_conv_variant(v)
            for _ in range(3):
                dw = kernels.conv2d_wgrad(x, dy, 3, 1.0, upsample=bool(ups))
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(REPS):
                dw = kernels.conv2d_wgrad(x, dy, 3, 1.0, upsample=bool(ups))
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / REPS
            outs[v] = dw
            if rnd == 1:
                line += f"  v{v} {us:7.1f} us {flops / us * 1e-6:7.1f} TF"
    err = float((outs[0] - outs[3]).abs().max() / outs[3].abs().max()) if 3 in outs and 0 in outs else float("nan")
    print(line, f"  max rel diff {err:.2e}")
lib.rgbd_debug_conv_variant(0)
