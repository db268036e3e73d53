"""Per-shape timing of the conv kernels (HIP events), layer shapes of the 128x128 stage at B=32."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rgbd_gan_amd import kernels
B = int(os.environ.get("B", "32"))
dev = "cuda:0"
shapes = [(128, 64, 64), (128, 64, 128), (128, 128, 128), (128, 128, 64), (64, 128, 256), (64, 256, 256), (64, 256, 128),
          (32, 256, 256), (16, 256, 256), (8, 256, 256)]
def timeit(fn, n=10):
    for _ in range(3): fn()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for H, Cin, Cout in shapes:
    x = torch.randn(B, H, H, Cin, device=dev).to(torch.bfloat16)
    dy = torch.randn(B, H, H, Cout, device=dev).to(torch.bfloat16)
    w = torch.randn(Cout, Cin, 3, 3, device=dev)
    bias = torch.zeros(Cout, device=dev)
    wf, wd = kernels.pack_weights(w, float(np.sqrt(2.0 / (Cin * 9))))
    fl = 2.0 * B * H * H * Cin * Cout * 9
    tf = timeit(lambda: kernels.conv2d_fprop(x, wf, 3, 3, 1, bias=bias, lrelu_channels=Cout))
    tw = timeit(lambda: kernels.conv2d_wgrad(x, dy, 3, 1.0))
    print(f"H={H:4d} Cin={Cin:4d} Cout={Cout:4d}  fprop {tf:7.1f} us {fl/tf/1e6:7.0f} TF   wgrad {tw:7.1f} us {fl/tw/1e6:7.0f} TF")
