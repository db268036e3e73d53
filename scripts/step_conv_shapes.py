"""Per-shape table of the conv launches of one training step (eager, HIP events per launch): which layers carry the
time of the 3x3 kernels.  RGBD_PROFILE_SHAPES=1 python scripts/step_conv_shapes.py"""
import os, sys
os.environ["RGBD_PROFILE_SHAPES"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from rgbd_gan_amd import kernels
from rgbd_gan_amd.training import DeviceImageIterator, build_training
from rgbd_gan_amd.utils import yaml_utils

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
config = yaml_utils.load(os.path.join(ROOT, "configs", os.environ.get("CONFIG", "stylegan_shapenet_car.yml")))    # CONFIG / B: another workload
if os.environ.get("B"):
    config.batchsize = int(os.environ["B"])
device = torch.device("cuda", 0)
images = np.random.RandomState(0).randint(0, 256, (256, 3, 128, 128)).astype("uint8")
it = DeviceImageIterator(images, config.batchsize, device, seed=0)
gen, dis, opt, upd = build_training(config, device, None, iterator=it, nan_check_interval=0)
upd.iteration = int(os.environ.get("ITERATION", "200000"))
upd.use_graphs = False
for _ in range(3):
    upd.update()
with kernels.launch_profile() as prof:
    for _ in range(2):
        upd.update()
rows = sorted(prof.summary().items(), key=lambda kv: -kv[1][1])
tot = sum(v[1] for _, v in rows)
print(f"{'kernel / layer':70s} {'n':>4s} {'ms/step':>8s} {'avg us':>8s} {'TF':>7s}")
for k, (n, t, f, b) in rows:
    print(f"{k:70s} {n // 2:4d} {t * 1e3 / 2:8.3f} {t / n * 1e6:8.1f} {f / t / 1e12:7.1f}")
print("total ms/step", tot * 1e3 / 2)
