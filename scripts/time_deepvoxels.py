"""Step time of configs/deepvoxels_shapenet_car.yml (BASELINE config 4: B=10, 64x64, DeepVoxelsUpdater, eager)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rgbd_gan_amd import kernels
from rgbd_gan_amd.training import DeviceImageIterator, build_training
from rgbd_gan_amd.utils import yaml_utils
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cfg = yaml_utils.load(os.path.join(root, "configs", "deepvoxels_shapenet_car.yml"))
images = np.random.RandomState(0).randint(0, 256, (64, 3, 128, 128)).astype("uint8")
it = DeviceImageIterator(images, cfg.batchsize, "cuda:0", seed=0)
gen, dis, opt, upd = build_training(cfg, "cuda:0", iterator=it, nan_check_interval=0)
upd.iteration = 100
for _ in range(4):
    upd.update()
torch.cuda.synchronize(); t0 = time.perf_counter()
n = 10
for _ in range(n):
    upd.update()
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"deepvoxels step: {1e3 * (t2 - t0) / n:.1f} ms ({cfg.batchsize * n / (t2 - t0):.1f} img/s), host enqueue {1e3 * (t1 - t0) / n:.1f} ms")
with kernels.launch_profile() as prof:
    upd.update()
s = prof.summary()
print({k: (v[0], round(v[1] * 1e3, 2)) for k, v in s.items()})
