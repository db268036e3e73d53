"""A/B the 3x3 conv kernels on the layer shapes of the 128x128 stage in ONE process (same device, same clocks):
variant 0 = default (software-pipelined, LDS-DMA), 2 = the same with 64-channel tiles, variant 1 = register-staged halo-patch kernel.  Prints TFLOP/s per shape and
checks that both variants give identical bytes.

    python scripts/ab_conv.py            # B=32
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rgbd_gan_amd import kernels, _lib

B = int(os.environ.get("B", "32"))
REPS = int(os.environ.get("REPS", "30"))
dev = "cuda:0"
lib = _lib.debug_library().__enter__()     # the A/B reference kernels live in the debug library (build --debug)
shapes = [(128, 64, 64, 0), (128, 128, 64, 1), (128, 64, 128, 0), (64, 128, 128, 0), (64, 128, 256, 0), (64, 256, 256, 0),
          (32, 256, 256, 0), (16, 256, 256, 0), (32, 256, 256, 1), (64, 256, 128, 1)]          # (Hout, Cin, Cout, upsample)
variants = [int(v) for v in os.environ.get("VARIANTS", "1,0").split(",")]
for H, Cin, Cout, ups in shapes:
    Hin = H // 2 if ups else H
    x = torch.randn(B, Hin, Hin, Cin, device=dev).to(torch.bfloat16)
    w = torch.randn(Cout, Cin, 3, 3, device=dev)
    bias = torch.randn(Cout, device=dev)
    wf, wd = kernels.pack_weights(w, float(np.sqrt(2.0 / (Cin * 9))))
    flops = 2.0 * B * H * H * Cout * Cin * 9
    res = torch.randn(B, H, H, Cout, device=dev).to(torch.bfloat16) if os.environ.get("RES") else None
    outs, line = {}, f"H={H:3d} Cin={Cin:3d} Cout={Cout:3d} ups={ups}:"
    for rnd in range(2):
        for v in variants:
            lib.rgbd_debug_conv_variant(v)
            for _ in range(3):
                y = kernels.conv2d_fprop(x, wf, 3, 3, 1, bias=bias, residual=res, lrelu_channels=Cout, upsample=bool(ups))
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(REPS):
                y = kernels.conv2d_fprop(x, wf, 3, 3, 1, bias=bias, residual=res, lrelu_channels=Cout, upsample=bool(ups))
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / REPS
            outs[v] = y
            if rnd == 1:
                line += f"  v{v} {us:7.1f} us {flops / us * 1e-6:7.1f} TF"
    same = all(torch.equal(outs[variants[0]], outs[v]) for v in variants[1:])
    print(line, " identical" if same else "  DIFFERENT")
lib.rgbd_debug_conv_variant(0)
