"""The frustum resampling's backward at configuration 4's size (B = 10, 32 features, 56 x 64 x 64 frustum, 32^3 grid), alone on the
device: the row-wise list kernel (rgbd_trilinear_bwd_fm) against the sorted-brick kernel (rgbd_trilinear_bwd_frustum), HIP-event
time per launch over 20 launches each, results compared.
    python scripts/time_trilinear_bwd.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import deepvoxels as odv                                     # noqa: E402  (geometry constants only)
from rgbd_gan_amd.deepvoxel import deepvoxel as dv                        # noqa: E402
from rgbd_gan_amd.deepvoxel.projection import ProjectionHelper           # noqa: E402
from rgbd_gan_amd.updater import get_camera_matries                      # noqa: E402

B, F = 10, 32
fr = odv.Frustum()
rng = np.random.RandomState(0)
th = np.zeros((B, 6), dtype="float32")
th[:, 0] = rng.uniform(-0.3, 0.3, B)
th[:, 1] = rng.uniform(-1.0, 1.0, B)
cams = get_camera_matries(th)
K = np.array([[128., 0, 32., 0], [0, 128., 32., 0], [0, 0, 1, 0], [0, 0, 0, 1]])
ph = ProjectionHelper(K, K, [64, 64], [64, 64], 0.0, 1.0, [32, 32, 32], fr.voxel_size, fr.near_plane, fr.depth)
idx, coords, counts = ph.compute_proj_idcs_batch(cams)
print("in-grid samples per camera:", counts.cpu().tolist(), "of", 64 * 64 * fr.depth)
g = torch.Generator().manual_seed(0)
grid = torch.randn(B, 32, 32, 32, F, generator=g).cuda()
dout = torch.randn(B, F, fr.depth, 64, 64, generator=g).cuda()
res = {}
for bricks in (False, True, False, True):
    dv.TRILINEAR_BWD_BRICKS = bricks
    gg = grid.clone().requires_grad_(True)
    out = dv.interpolate_trilinear_batch(gg, idx, coords, counts, [64, 64], fr.depth, feature_minor=True)
    times = []
    for _ in range(22):
        gg.grad = None
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        out.backward(dout, retain_graph=True)
        e1.record()
        torch.cuda.synchronize()
        times.append(e0.elapsed_time(e1) * 1e3)
    res[bricks] = gg.grad.clone()
    t = sorted(times[2:])
    alg = 4.0 * B * F * (32 ** 3 + 64 * 64 * fr.depth)
    print(f"{'sorted bricks' if bricks else 'row-wise list'}: median {t[len(t) // 2]:.1f} us, min {t[0]:.1f} us per backward (zero fill + scatter); "
          f"algorithmic {alg / 1e6:.1f} MB -> {alg / t[len(t) // 2] / 1e3:.0f} GB/s")
d = (res[True] - res[False]).abs().max().item()
print(f"max |difference| between the two forms {d:.3e} at gradient scale {res[False].abs().max().item():.3e}")
