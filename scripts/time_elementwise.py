"""Achieved HBM bandwidth of the step's elementwise passes at their largest shapes (HIP-graph replay), next to a plain
read+write pass of the same size (torch add_) as the ceiling of this box."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rgbd_gan_amd import kernels

dev = "cuda:0"
B = 32


def run(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    g.replay(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (5 * n) * 1e3


def row(name, us, nbytes):
    print(f"{name:44s} {us:8.1f} us  {nbytes / us / 1e6:6.2f} TB/s  ({nbytes / 1e6:.0f} MB)")


SHAPES = ((128, 64), (128, 128), (64, 256), (32, 256), (16, 256))
if os.environ.get("ONLY"):
    SHAPES = tuple(tuple(int(v) for v in t.split("x")) for t in os.environ["ONLY"].split(","))
for H, C in SHAPES:
    t = lambda: torch.randn(B, H, H, C, device=dev).to(torch.bfloat16)
    n = B * H * H * C * 2
    x, y, dy = t(), t(), t()
    dp = torch.randn(B, H // 2, H // 2, C, device=dev).to(torch.bfloat16)
    bg = torch.zeros(C, device=dev)
    ss = torch.randn(B, 2 * C, device=dev)
    s = torch.randn(B, device=dev)
    print(f"--- {H}x{H} x {C}")
    row("torch add_ (read + write)", run(lambda: x.add_(1.0)), 2 * n)
    row("lrelu_bwd + colsum", run(lambda: kernels.lrelu_bwd(dy, y, C, bias_grad=bg)), 3 * n)
    row("lrelu_bwd", run(lambda: kernels.lrelu_bwd(dy, y, C)), 3 * n)
    row("unpool2_lrelu_bwd + colsum", run(lambda: kernels.unpool2_lrelu_bwd(dp, y, (B, H, H, C), bias_grad=bg)), 2.25 * n)
    row("pool2_masked", run(lambda: kernels.pool2_masked(x, y)), 2.25 * n)
    row("axpy_rows", run(lambda: kernels.axpy_rows(x, y, s)), 3 * n)
    out, mean, rstd = kernels.adain_fwd(x, ss)
    row("adain_fwd (reduce + apply)", run(lambda: kernels.adain_fwd(x, ss)), 3 * n)
    row("adain_bwd (reduce + apply, lrelu, bias)", run(lambda: kernels.adain_bwd(x, dy, ss, mean, rstd, fused=True, lrelu_slope=0.2, bias_grad=bg)), 5 * n)
