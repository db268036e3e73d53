"""Does a memcpy node inside a captured graph serialise it against a graph replayed on another stream?
Stream B: 40 conv launches with / without a device-to-device hipMemcpyAsync (torch copy_ of a contiguous tensor) after
every 10th; stream A: 400 tiny kernels.  One graph per stream."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from rgbd_gan_amd import kernels

dev = "cuda:0"
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
x = torch.randn(1 << 16, device=dev)
B = 32
xx = torch.randn(B, 64, 64, 256, device=dev).to(torch.bfloat16)
w = torch.randn(256, 256, 3, 3, device=dev)
wf, _ = kernels.pack_weights(w, 0.02)
src, dst = torch.randn(1 << 20, device=dev), torch.empty(1 << 20, device=dev)


def chain_a(n=400):
    y = x
    for _ in range(n):
        y = y * 1.0001 + 0.5
    return y


def chain_b(mode):
    def run():
        for i in range(40):
            kernels.conv2d_fprop(xx, wf, 3, 3, 1)
            if i % 10 == 9:
                if mode == "memcpy":
                    dst.copy_(src)                       # contiguous same-dtype copy: hipMemcpyAsync -> memcpy node
                elif mode == "kernel copy":
                    torch.add(src, 0.0, out=dst)         # the same bytes moved by a kernel
    return run


def timed(fa, fb, reps=5):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if fa:
            fa()
        if fb:
            fb()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best * 1e3


def graphed(stream, fn):
    with torch.cuda.stream(stream):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=stream):
        fn()
    def run():
        with torch.cuda.stream(stream):
            g.replay()
    return run


ra = graphed(sa, chain_a)
for mode in ("none", "kernel copy", "memcpy"):
    rb = graphed(sb, chain_b(mode))
    ta, tb, tba = timed(ra, None), timed(None, rb), timed(rb, ra)
    print(f"B with {mode:12s}: A {ta:6.3f} ms  B {tb:6.3f} ms  B then A launched {tba:6.3f} ms  overlap {(ta + tb - tba) / min(ta, tb):5.2f}",
          flush=True)
