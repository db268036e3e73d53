"""Run the hot conv kernels alone on layer shapes of the 128x128 stage (for rocprofv3 --pmc / --kernel-trace)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rgbd_gan_amd import kernels, _lib
lib = _lib.debug_library().__enter__()     # the A/B reference kernels live in the debug library (build --debug)
B = int(os.environ.get("B", "32"))
reps = int(os.environ.get("REPS", "5"))
dev = "cuda:0"
shapes = [(128, 64, 128), (128, 128, 128), (64, 128, 256), (64, 256, 256), (32, 256, 256)]   # (H, Cin, Cout)
which = os.environ.get("WHICH", "fprop,wgrad").split(",")
for H, Cin, Cout in shapes:
    x = torch.randn(B, H, H, Cin, device=dev).to(torch.bfloat16)
    dy = torch.randn(B, H, H, Cout, device=dev).to(torch.bfloat16)
    w = torch.randn(Cout, Cin, 3, 3, device=dev)
    bias = torch.zeros(Cout, device=dev)
    wf, wd = kernels.pack_weights(w, float(np.sqrt(2.0 / (Cin * 9))))
    # both generations of each kernel in the same pass (same box, same counters): the register-staged references
    # (variants 1 / 3) first, then the shipped LDS-DMA kernels (variant 0)
    for variant in (1, 3, 7, 0):          # 7: round 5's dual-workgroup A/B kernel (128-channel tiles), same pass as the shipped one
        lib.rgbd_debug_conv_variant(variant)
        for _ in range(reps):
            if "fprop" in which and variant != 3:
                kernels.conv2d_fprop(x, wf, 3, 3, 1, bias=bias, lrelu_channels=Cout)
            if "wgrad" in which and variant not in (1, 7):
                kernels.conv2d_wgrad(x, dy, 3, 1.0)
        torch.cuda.synchronize()
    lib.rgbd_debug_conv_variant(0)
print("done")
