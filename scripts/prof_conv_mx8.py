"""Run the MXFP8 form of the pipelined 3x3 kernel, its bf16 twin and the activation quantiser alone on layer shapes of the
256x256 networks (ch = 512, per-GPU batch 16), for rocprofv3 --pmc / --kernel-trace (scripts/collect_profiles.sh)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from rgbd_gan_amd import kernels

B = int(os.environ.get("B", "16"))
reps = int(os.environ.get("REPS", "5"))
dev = "cuda:0"
shapes = [(32, 512, 512), (64, 512, 512), (64, 256, 256), (128, 256, 256), (128, 128, 128), (256, 128, 128)]   # (H, Cin, Cout)
kernels.MX8_MIN_TILES = 0
for H, Cin, Cout in shapes:
    x = torch.randn(B, H, H, Cin, device=dev).to(torch.bfloat16)
    w = torch.randn(Cout, Cin, 3, 3, device=dev)
    bias = torch.zeros(Cout, device=dev)
    scale = float(np.sqrt(2.0 / (Cin * 9)))
    wf, _ = kernels.pack_weights(w, scale)
    f, _ = kernels.pack_weights_mx8(w, scale)
    img = kernels.Mx8Image(wf, *f)
    for _ in range(reps):
        kernels.conv2d_fprop(x, wf, 3, 3, 1, bias=bias, lrelu_channels=Cout)          # bf16 twin, same pass
        x._mx8 = None
        kernels.conv2d_fprop(x, img, 3, 3, 1, bias=bias, lrelu_channels=Cout)         # quantiser + MXFP8 kernel
    torch.cuda.synchronize()
print("done")
