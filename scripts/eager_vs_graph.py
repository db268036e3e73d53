"""Step time of the four arrangements of one step on this box: {graph replay, eager launches} x {one stream, two streams}.
Eager launches are issued from Python (~400 per step): whether they keep the GPU busy depends on the host core."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rgbd_gan_amd.training import DeviceImageIterator, build_training
from rgbd_gan_amd.utils import yaml_utils

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
config = yaml_utils.load(os.path.join(ROOT, "configs", "stylegan_shapenet_car.yml"))
device = torch.device("cuda", 0)
images = np.random.RandomState(0).randint(0, 256, (256, 3, 128, 128)).astype("uint8")
for graphs, conc in ((True, False), (True, True), (False, False), (False, True)):
    if True:
        it = DeviceImageIterator(images, config.batchsize, device, seed=0)
        gen, dis, opt, upd = build_training(config, device, None, iterator=it, nan_check_interval=0, use_graphs=graphs,
                                            concurrent_phases=conc)
        upd.iteration = 200000
        for _ in range(8):
            upd.update()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(30):
            upd.update()
        t_host = time.perf_counter() - t0
        torch.cuda.synchronize()
        t = time.perf_counter() - t0
        print(f"graphs={graphs!s:5} two_streams={conc!s:5}: {t / 30 * 1e3:7.3f} ms/step (host enqueue {t_host / 30 * 1e3:7.3f} ms)", flush=True)
        del gen, dis, opt, upd
