"""As stream_overlap_probe.py with the library's own kernels on stream B: chains of conv3x3_sp_kernel launches (persistent,
one 512-thread workgroup per CU, ~150 KB of LDS), of the weight-gradient kernel, of an elementwise pass -- next to a chain of
tiny kernels on stream A.  One captured graph per stream."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from rgbd_gan_amd import kernels

dev = "cuda:0"
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
x = torch.randn(1 << 16, device=dev)
B = 32


def t(*shape):
    return torch.randn(*shape, device=dev).to(torch.bfloat16)


def chain_a(n=400):
    y = x
    for _ in range(n):
        y = y * 1.0001 + 0.5
    return y


def make(name):
    if name == "sp 64^2 256->256":
        xx = t(B, 64, 64, 256); w = torch.randn(256, 256, 3, 3, device=dev); wf, _ = kernels.pack_weights(w, 0.02)
        return lambda: [kernels.conv2d_fprop(xx, wf, 3, 3, 1) for _ in range(40)]
    if name == "sp 16^2 256->256":
        xx = t(B, 16, 16, 256); w = torch.randn(256, 256, 3, 3, device=dev); wf, _ = kernels.pack_weights(w, 0.02)
        return lambda: [kernels.conv2d_fprop(xx, wf, 3, 3, 1) for _ in range(200)]
    if name == "wgrad 64^2 256->256":
        xx, dy = t(B, 64, 64, 256), t(B, 64, 64, 256)
        return lambda: [kernels.conv2d_wgrad(xx, dy, 3, 0.02) for _ in range(20)]
    if name == "lrelu_bwd 128^2 x64":
        xx, dy = t(B, 128, 128, 64), t(B, 128, 128, 64)
        return lambda: [kernels.lrelu_bwd(dy, xx, 64) for _ in range(100)]
    if name.startswith("sp 128^2"):
        ci, co = {"sp 128^2 64->64": (64, 64), "sp 128^2 64->128": (64, 128), "sp 128^2 128->128": (128, 128)}[name]
        xx = t(B, 128, 128, ci); w = torch.randn(co, ci, 3, 3, device=dev); wf, _ = kernels.pack_weights(w, 0.02)
        bias = torch.zeros(co, device=dev)
        return lambda: [kernels.conv2d_fprop(xx, wf, 3, 3, 1, bias=bias, lrelu_channels=co) for _ in range(30)]
    if name == "gather 8^2 256->256":
        xx = t(B, 8, 8, 256); w = torch.randn(256, 256, 3, 3, device=dev); wf, _ = kernels.pack_weights(w, 0.02)
        return lambda: [kernels.conv2d_fprop(xx, wf, 3, 3, 1) for _ in range(200)]
    if name == "axpy_rows 128^2 x128":
        a, b2, sc = t(B, 128, 128, 128), t(B, 128, 128, 128), torch.randn(B, device=dev)
        return lambda: [kernels.axpy_rows(a, b2, sc) for _ in range(60)]
    if name == "pool2_masked 128^2 x128":
        a = t(B, 128, 128, 128)
        return lambda: [kernels.pool2_masked(a) for _ in range(60)]
    if name == "lrelu_bwd+colsum 128^2 x64":
        xx, dy = t(B, 128, 128, 64), t(B, 128, 128, 64); bg = torch.zeros(64, device=dev)
        return lambda: [kernels.lrelu_bwd(dy, xx, 64, bias_grad=bg) for _ in range(100)]
    if name == "torch add 64 MB":
        a = torch.randn(1 << 24, device=dev)
        return lambda: [a.add_(1.0) for _ in range(100)]
    raise SystemExit(name)


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
for name in ("sp 64^2 256->256", "sp 128^2 64->64", "sp 128^2 64->128", "sp 128^2 128->128", "gather 8^2 256->256",
             "wgrad 64^2 256->256", "lrelu_bwd 128^2 x64", "lrelu_bwd+colsum 128^2 x64", "axpy_rows 128^2 x128",
             "pool2_masked 128^2 x128", "torch add 64 MB"):
    rb = graphed(sb, make(name))
    ta, tb, tab, tba = timed(ra, None), timed(None, rb), timed(ra, rb), timed(rb, ra)
    print(f"B = {name:22s}: A {ta:7.3f} ms  B {tb:7.3f} ms  A then B launched {tab:7.3f} ms  B then A launched {tba:7.3f} ms  "
          f"overlap {(ta + tb - min(tab, tba)) / min(ta, tb):5.2f}", flush=True)
