"""Long repeat screen of the LDS-DMA kernels (hand-counted vmcnt / barrier ordering): N launches per shape, every output
compared bit for bit with the first launch (and, for the 3x3 conv, with the register-staged reference kernel).
    N=300 python scripts/dma_race_screen.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rgbd_gan_amd import kernels, _lib
N = int(os.environ.get("N", "300"))
lib = _lib.debug_library().__enter__()     # the A/B reference kernels live in the debug library (build --debug)
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(1)
bad = 0
for B, H, Cin, Cout, ups, res in [(32, 64, 256, 256, 0, 1), (32, 128, 64, 128, 0, 0), (32, 128, 128, 64, 1, 0), (32, 32, 256, 256, 0, 0),
                                  (32, 16, 256, 256, 0, 1), (7, 64, 128, 192, 0, 0), (32, 64, 256, 128, 1, 0)]:
    Hin = H // 2 if ups else H
    x = torch.randn(B, Hin, Hin, Cin, device=dev, generator=g).to(torch.bfloat16)
    w = torch.randn(Cout, Cin, 3, 3, device=dev, generator=g)
    bias = torch.randn(Cout, device=dev, generator=g)
    r = torch.randn(B, H, H, Cout, device=dev, generator=g).to(torch.bfloat16) if res else None
    wf, _ = kernels.pack_weights(w, float(np.sqrt(2.0 / (Cin * 9))))
    lib.rgbd_debug_conv_variant(1)
    ref = kernels.conv2d_fprop(x, wf, 3, 3, 1, bias=bias, residual=r, upsample=bool(ups), lrelu_channels=Cout)
    lib.rgbd_debug_conv_variant(0)
    n_bad = 0
    for i in range(N):
        y = kernels.conv2d_fprop(x, wf, 3, 3, 1, bias=bias, residual=r, upsample=bool(ups), lrelu_channels=Cout)
        if i % 10 == 9 or i == N - 1:
            n_bad += int(not torch.equal(y, ref))
        else:
            ok = torch.equal(y, ref)            # synchronises every launch: different timing from the back-to-back runs
            n_bad += int(not ok)
    print(f"conv  B={B} H={H} {Cin}->{Cout} ups={ups} res={res}: {n_bad} of {N} launches differ", flush=True)
    bad += n_bad
    dy = torch.randn(B, H, H, Cout, device=dev, generator=g).to(torch.bfloat16)
    first = kernels.conv2d_wgrad(x, dy, 3, 1.0, upsample=bool(ups))
    n_bad = 0
    for i in range(N // 2):
        n_bad += int(not torch.equal(kernels.conv2d_wgrad(x, dy, 3, 1.0, upsample=bool(ups)), first))
    print(f"wgrad B={B} H={H} {Cin}->{Cout} ups={ups}: {n_bad} of {N // 2} launches differ", flush=True)
    bad += n_bad
# back-to-back without host synchronisation in between (the graph's timing): outputs kept, compared at the end
x = torch.randn(32, 64, 64, 256, device=dev, generator=g).to(torch.bfloat16)
w = torch.randn(256, 256, 3, 3, device=dev, generator=g)
wf, _ = kernels.pack_weights(w, 0.02)
outs = [kernels.conv2d_fprop(x, wf, 3, 3, 1, lrelu_channels=256) for _ in range(40)]
torch.cuda.synchronize()
n_bad = sum(int(not torch.equal(o, outs[0])) for o in outs)
print(f"conv back-to-back: {n_bad} of 40 differ")
bad += n_bad
print("TOTAL differing launches:", bad)
