"""Do two HIP streams overlap?  Stream A: a chain of small latency-bound kernels; stream B: a chain of compute kernels of a
given granularity (many short ones or few long ones, same total work).  Times A alone, B alone and both together, with
eager launches and with one captured graph per stream.  overlap = (tA + tB - tAB) / min(tA, tB): 1 = fully hidden, 0 = serial.
  python scripts/hw/stream_overlap_probe.py"""
import time
import torch

dev = "cuda:0"
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
x = torch.randn(1 << 16, device=dev)


def chain_a(n=400):
    y = x
    for _ in range(n):
        y = y * 1.0001 + 0.5          # tiny elementwise kernels: launch/latency bound
    return y


def make_b(size, count):
    a = torch.randn(size, size, device=dev, dtype=torch.bfloat16)
    def run():
        y = a
        for _ in range(count):
            y = a @ a
        return y
    return run


def timed(fn_a, fn_b, reps=5):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if fn_a:
            fn_a()
        if fn_b:
            fn_b()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best * 1e3


def on(stream, fn):
    def run():
        with torch.cuda.stream(stream):
            fn()
    return run


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


for size, count in ((8192, 8), (2048, 400), (1024, 2000)):
    fb = make_b(size, count)
    for mode in ("eager", "graph"):
        if mode == "eager":
            ra, rb = on(sa, chain_a), on(sb, fb)
        else:
            ra, rb = graphed(sa, chain_a), graphed(sb, fb)
        ta, tb, tab = timed(ra, None), timed(None, rb), timed(ra, rb)
        print(f"B = {count:5d} x matmul {size}^3 ({tb / count * 1e3:8.1f} us each)  {mode:5s}:  A {ta:7.3f} ms  B {tb:7.3f} ms  "
              f"A||B {tab:7.3f} ms  overlap {(ta + tb - tab) / min(ta, tb):5.2f}", flush=True)
