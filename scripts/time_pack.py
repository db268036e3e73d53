"""rgbd_pack_weights_multi on the DeepVoxels generator's weight set (configuration 4) and per layer class: us per launch and GB/s on
(4 B read + 2 x 2 B written) per folded element.    python scripts/time_pack.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rgbd_gan_amd import kernels   # noqa: E402

dev = "cuda"
LAYERS = [("voxel 3x3x3 64->64 (mode 0) x4", [((64, 64, 3, 3, 3), (64, 192, 3, 3), (0, 64, 64))] * 4),
          ("voxel 3x3x3 32->32 (mode 0, padded) x3", [((32, 32, 3, 3, 3), (64, 192, 3, 3), (0, 32, 32))] * 3),
          ("renderer c1 4x4s2 512->1024 (mode 1)", [((1024, 512, 4, 4), (1024, 8192, 1, 1), (1, 1024, 512))]),
          ("renderer c4 3x3 1024->1024 (plain)", [((1024, 1024, 3, 3), None, None)]),
          ("renderer c5 3x3 1024->512 (plain)", [((512, 1024, 3, 3), None, None)]),
          ("renderer c6 3x3 1024->256 (plain)", [((256, 1024, 3, 3), None, None)]),
          ("D 3x3 256->256 (plain) x10", [((256, 256, 3, 3), None, None)] * 10)]


def entries(specs):
    out = []
    for shape, folded, fold in specs:
        w = torch.randn(*shape, device=dev)
        co, ci, kh, kw = folded if folded else shape
        wf = torch.empty(kh * kw, co, ci, dtype=torch.bfloat16, device=dev)
        wd = torch.empty(kh * kw, ci, co, dtype=torch.bfloat16, device=dev)
        out.append((w, 0.5, wf, wd) + ((folded, fold) if folded else ()))
    return out


def timed(table, reps=30):
    for _ in range(3):
        kernels.pack_weights_multi(table)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        kernels.pack_weights_multi(table)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


allent = []
for name, specs in LAYERS:
    ent = entries(specs)
    if not name.startswith("D "):
        allent += ent
    n = sum(e[2].numel() for e in ent)
    us = timed(kernels.build_pack_table(ent))
    print(f"{name:46s} {n / 1e6:6.2f} M elements  {us:7.1f} us  {n * 8 / us / 1e3:7.1f} GB/s")
n = sum(e[2].numel() for e in allent)
us = timed(kernels.build_pack_table(allent))
print(f"{'all of the generator in one launch':46s} {n / 1e6:6.2f} M elements  {us:7.1f} us  {n * 8 / us / 1e3:7.1f} GB/s")
