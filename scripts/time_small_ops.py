"""Alone-on-the-chip time of the step's latency-bound launches (HIP events over back-to-back launches): the small linears of
the mapping MLP / style affines, the instance-norm reductions of the 4x4 / 8x8 layers, the plane convs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rgbd_gan_amd import kernels

dev = "cuda"
def t(fn, name, n=200):
    for _ in range(10): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    print(f"{name:44s} {e0.elapsed_time(e1) * 1e3 / n:7.1f} us")

for M, K, N in [(64, 256, 256), (32, 256, 256), (32, 256, 1024), (32, 265, 256), (32, 256, 512), (32, 4096, 256)]:
    x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev); b = torch.randn(N, device=dev)
    dy = torch.randn(M, N, device=dev)
    t(lambda: kernels.linear_fwd(x, w, b, 0.1, True), f"linear_fwd {M}x{K}->{N}")
    y = kernels.linear_fwd(x, w, b, 0.1, True)
    dw, db = torch.zeros_like(w), torch.zeros_like(b)
    t(lambda: kernels.linear_bwd(dy, y, x, w, 0.1, True, True, dw, db), f"linear_bwd (dgrad + wgrad) {M}x{K}->{N}")
for S, C in [(4, 256), (8, 256), (16, 256), (32, 256)]:
    x = torch.randn(32, S, S, C, device=dev).to(torch.bfloat16); ss = torch.randn(32, 2 * C, device=dev)
    t(lambda: kernels.adain_fwd(x, ss), f"adain_fwd (reduce + apply) {S}x{S}x{C}")
