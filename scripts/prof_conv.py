"""Run the hot conv kernels alone on layer shapes of the 128x128 stage (for rocprofv3 --pmc / --kernel-trace)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rgbd_gan_amd import kernels
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
    for _ in range(reps):
        if "fprop" in which:
            kernels.conv2d_fprop(x, wf, 3, 3, 1, bias=bias, lrelu_channels=Cout)
        if "wgrad" in which:
            kernels.conv2d_wgrad(x, dy, 3, 1.0)
    torch.cuda.synchronize()
print("done")
