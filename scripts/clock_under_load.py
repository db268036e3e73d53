"""Shader clock the GPU sustains while one kernel runs back to back (rocm-smi polled from a child process)."""
import os, subprocess, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rgbd_gan_amd import kernels
which = sys.argv[1] if len(sys.argv) > 1 else "patch"
B, H, Cin, Cout = 32, 128, 128, 128
x = torch.randn(B, H, H, Cin, device="cuda:0").to(torch.bfloat16)
w = torch.randn(Cout, Cin, 3, 3, device="cuda:0")
wf, _ = kernels.pack_weights(w, float(np.sqrt(2.0 / (Cin * 9))))
bias = torch.zeros(Cout, device="cuda:0")
a = torch.randn(8192, 8192, device="cuda:0").to(torch.bfloat16)
b = torch.randn(8192, 8192, device="cuda:0").to(torch.bfloat16)
big = torch.empty(1 << 28, device="cuda:0")
def work():
    if which == "patch":
        kernels.conv2d_fprop(x, wf, 3, 3, 1, bias=bias, lrelu_channels=Cout)
    elif which == "gemm":
        torch.matmul(a, b)
    elif which == "copy":
        big.mul_(1.0)
    else:
        time.sleep(0.001)
for _ in range(20): work()
torch.cuda.synchronize()
poll = subprocess.Popen("for i in 1 2 3 4 5 6; do rocm-smi --showclocks 2>/dev/null | grep -i 'sclk\\|mclk' | head -2; rocm-smi --showpower 2>/dev/null | grep -i 'power' | head -1; sleep 0.4; done",
                        shell=True, stdout=subprocess.PIPE, text=True)
t0 = time.time(); n = 0
while time.time() - t0 < 3.0:
    for _ in range(50): work()
    torch.cuda.synchronize(); n += 50
dt = time.time() - t0
print(which, f"{dt / n * 1e6:.1f} us per call")
print(poll.communicate()[0])
