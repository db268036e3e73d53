"""In-kernel timeline of the ping-pong 3x3 conv kernel (diagnostic build, variant 15): s_memtime stamps of workgroup
0's waves 0 and 4 around the segments of the first 36 K steps.  Prints per-step durations in cycles."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rgbd_gan_amd import kernels, _lib
lib = _lib.load()
B, H, Cin, Cout = 32, int(os.environ.get("H", 64)), int(os.environ.get("CIN", 256)), int(os.environ.get("COUT", 256))
x = torch.randn(B, H, H, Cin, device="cuda").to(torch.bfloat16)
w = torch.randn(Cout, Cin, 3, 3, device="cuda")
wf, wd = kernels.pack_weights(w, 0.02)
lib.rgbd_debug_conv_variant(15)
for _ in range(20):
    kernels.conv2d_fprop(x, wf, 3, 3, 1, lrelu_channels=Cout)
torch.cuda.synchronize()
buf = np.zeros(2 * 36 * 8, dtype=np.uint64)
rc = lib.rgbd_debug_conv_stamps(buf.ctypes.data_as(ctypes.c_void_p))
lib.rgbd_debug_conv_variant(0)
s = buf.reshape(2, 36, 8).astype(np.int64)
t0 = s[0, 0, 0]
names = ["dma", "reads", "vmcnt", "lgkm", "barrier", "mfma", "barrier2"]
print("step | G0: start  " + " ".join(f"{n:>7s}" for n in names) + " | G1: start  " + " ".join(f"{n:>7s}" for n in names))
for k in range(36):
    row = f"{k:4d} |"
    for g in range(2):
        row += f" {s[g, k, 0] - t0:9d}  " + " ".join(f"{s[g, k, i + 1] - s[g, k, i]:7d}" for i in range(7)) + " |"
    print(row)
d = s[0, 9:36, 0]
print("G0 cycles per step (steps 9..35):", float(np.diff(d).mean()))
