"""A/B of the dual-workgroup 3x3 kernel (conv3x3_dw_kernel: two 4-wave workgroups per CU) against the 8-wave pipelined kernel
(conv3x3_sp_kernel) on the step's layer shapes, ONE process, debug library.  Variants: 5 = 8-wave kernel, 0 = dual-workgroup
(planner's tile width), 7 / 8 = dual-workgroup forced to 128- / 64-channel tiles, 6 = dual-workgroup without its epilogue (timing
only).  Checks every variant against the 8-wave kernel's bytes (different sum order: bf16 ulps) and, at B <= 4, an fp32 convolution.

    python scripts/ab_conv_dw.py            # B=32
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rgbd_gan_amd import kernels, _lib

B = int(os.environ.get("B", "32"))
REPS = int(os.environ.get("REPS", "30"))
dev = "cuda:0"
lib = _lib.debug_library().__enter__()
shapes = [(128, 64, 64, 0), (128, 64, 128, 0), (128, 128, 128, 0), (128, 128, 64, 1), (64, 128, 128, 0), (64, 128, 256, 0),
          (64, 256, 256, 0), (64, 256, 128, 1), (32, 256, 256, 0), (32, 256, 256, 1), (16, 256, 256, 0)]   # (Hout, Cin, Cout, upsample)
variants = [int(v) for v in os.environ.get("VARIANTS", "5,0,7,8").split(",")]
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
    ref = outs[variants[0]].float()
    for v in variants[1:]:
        if v == 6:
            continue
        d = (outs[v].float() - ref).abs()
        nz = (d > 0).float().mean().item()
        rel = (d / (ref.abs() + 1e-3)).max().item()
        line += f"  | v{v} vs v{variants[0]}: differ {nz:.4f} max rel {rel:.4f}"
    print(line, flush=True)
lib.rgbd_debug_conv_variant(0)
