"""GPU time of the small-layer conv paths (HIP-graph replay, so the host launch cost is out of the picture)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rgbd_gan_amd import kernels
dev = "cuda:0"
B = 32
def bench(H, Cin, Cout, n=20):
    x = torch.randn(B, H, H, Cin, device=dev).to(torch.bfloat16)
    w = torch.randn(Cout, Cin, 3, 3, device=dev)
    bias = torch.zeros(Cout, device=dev)
    wf, _ = kernels.pack_weights(w, float(np.sqrt(2.0 / (Cin * 9))))
    for _ in range(3):
        kernels.conv2d_fprop(x, wf, 3, 3, 1, bias=bias, lrelu_channels=Cout)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            y = kernels.conv2d_fprop(x, wf, 3, 3, 1, bias=bias, lrelu_channels=Cout)
    g.replay(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (5 * n) * 1e3
for H in (4, 8, 16, 32):
    fl = 2.0 * B * H * H * 256 * 256 * 9
    t = bench(H, 256, 256)
    print(f"KSPLIT={os.environ.get('RGBD_DEBUG_KSPLIT','auto'):>4s} H={H:3d}  {t:7.1f} us  {fl/t/1e6:7.0f} TF")
