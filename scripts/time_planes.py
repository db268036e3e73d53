"""Bandwidth of the 1x1 plane convs (fromRGB / toRGB and their weight gradient) at the benchmark's sizes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rgbd_gan_amd import kernels
B, H, C = 32, 128, 64
x3 = torch.randn(B, 3, H, H, device="cuda"); x4 = torch.randn(B, 4, H, H, device="cuda")
w3 = torch.randn(C, 3, device="cuda"); w4 = torch.randn(4, C, device="cuda"); b = torch.randn(C, device="cuda"); b4 = torch.randn(4, device="cuda")
h = torch.randn(B, H, H, C, device="cuda").to(torch.bfloat16)
def t(fn, nbytes, name, n=50):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / n
    print(f"{name:28s} {us:7.1f} us  {nbytes / us * 1e-6:6.2f} TB/s")
act = B * H * H * C * 2
t(lambda: kernels.from_planes(x3, w3, b, 0.5, True), act + x3.numel() * 4, "from_planes<3>")
t(lambda: kernels.to_planes(h, w4, b4, 0.5), act + x4.numel() * 4, "to_planes<4>")
t(lambda: kernels.planes_outer(h, x3, True), act + x3.numel() * 4, "planes_outer<3> (+tsum)")
t(lambda: kernels.planes_outer(h, x4, False), act + x4.numel() * 4, "planes_outer<4>")
ps = torch.zeros(4, device="cuda")
t(lambda: kernels.planes_outer(h, x4, False, ps), act + x4.numel() * 4, "planes_outer<4> (+psum)")
w4t = torch.randn(C, 4, device="cuda")
t(lambda: kernels.from_planes(x4, w4t, None, 0.5, False), act + x4.numel() * 4, "from_planes<4>")
t(lambda: kernels.to_planes(h, torch.randn(3, C, device="cuda"), None, 0.5), act + x3.numel() * 4, "to_planes<3>")
