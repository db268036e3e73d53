import torch, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rgbd_gan_amd import kernels
x = torch.zeros(64, device="cuda")
big = torch.zeros(32*128*128*64, device="cuda", dtype=torch.bfloat16)
def run(fn, n=200):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): fn()
    g.replay(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (5 * n) * 1e3
print("tiny add_ in graph: %.2f us/launch" % run(lambda: x.add_(1.0)))
y = torch.zeros(64, device="cuda")
def two():
    x.add_(1.0); y.add_(1.0)
print("two independent tiny: %.2f us/pair" % run(two))
print("big bf16 add_ 67MB rw: %.2f us" % run(lambda: big.add_(1.0), 20))
