"""Do the branches of a captured fork / join run concurrently on replay (ROCm 7.2 HIP graphs)?  Two independent chains of small-grid
kernels (each far from filling the chip), captured (a) back to back on one stream, (b) forked onto a second stream and joined."""
import time
import torch

dev = torch.device("cuda", 0)
a = torch.randn(64, 4096, device=dev)
b = torch.randn(64, 4096, device=dev)
w = torch.randn(4096, 4096, device=dev)


def chain(x, n=20):
    for _ in range(n):
        x = torch.tanh(x @ w) * 0.5         # 64 x 4096 x 4096: a few workgroups, ~10 us
    return x


def timed(g, reps=50):
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps * 1e3


cap = torch.cuda.Stream()
side = torch.cuda.Stream()
chain(a); chain(b); torch.cuda.synchronize()
g1 = torch.cuda.CUDAGraph()
with torch.cuda.graph(g1, stream=cap):
    ya = chain(a)
    yb = chain(b)
g2 = torch.cuda.CUDAGraph()
with torch.cuda.graph(g2, stream=cap):
    ev = torch.cuda.Event(); ev.record(cap)
    side.wait_event(ev)
    with torch.cuda.stream(side):
        yb2 = chain(b)
        ev2 = torch.cuda.Event(); ev2.record(side)
    ya2 = chain(a)
    cap.wait_event(ev2)
g3 = torch.cuda.CUDAGraph()
with torch.cuda.graph(g3, stream=cap):
    ya3 = chain(a)
print("serial capture  : %.3f ms" % timed(g1))
print("fork/join       : %.3f ms" % timed(g2))
print("one chain alone : %.3f ms" % timed(g3))
print("results equal:", bool(torch.equal(ya, ya2)), bool(torch.equal(yb, yb2)))
