"""Conv-engine launches per phase of one eager training step (checks that no pass runs twice)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rgbd_gan_amd import kernels
from rgbd_gan_amd.training import DeviceImageIterator, build_training
from rgbd_gan_amd.utils import yaml_utils
cfg = yaml_utils.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "configs", "stylegan_shapenet_car.yml"))
images = np.random.RandomState(0).randint(0, 256, (16, 3, 128, 128)).astype("uint8")
it = DeviceImageIterator(images, 4, "cuda:0", seed=0)
gen, dis, opt, upd = build_training(cfg, "cuda:0", iterator=it, use_graphs=False, nan_check_interval=0)
upd.iteration = 200000
upd.update()
orig_gen, orig_dis = upd._gen_phase, upd._dis_phase
def wrap(name, fn):
    def f(st):
        with kernels.launch_profile() as prof:
            fn(st)
        print(name, {k: v[0] for k, v in prof.summary().items()})
    return f
upd._gen_phase = wrap("gen", orig_gen)
upd._dis_phase = wrap("dis", orig_dis)
upd.update()
