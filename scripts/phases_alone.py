"""Each captured phase of the step replayed ALONE (nothing on the other stream), timed with events: the phases' own cost,
against which the overlapped step (scripts/phase_timeline.py) is read.  CONFIG=... B=... RES256=[fp8] select another workload; the
updater's switches (RGBD_SIDE_CUS, RGBD_SIDE_WGRAD_WGS, RGBD_DFW_WGRAD_WGS, RGBD_CONCURRENT_PHASES) apply."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rgbd_gan_amd.training import DeviceImageIterator, build_training
from rgbd_gan_amd.utils import yaml_utils
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cfg = yaml_utils.load(os.path.join(root, "configs", os.environ.get("CONFIG", "stylegan_shapenet_car.yml")))   # CONFIG / B: another workload
images = np.random.RandomState(0).randint(0, 256, (256, 3, 128, 128)).astype("uint8")
it = DeviceImageIterator(images, int(os.environ.get("B", "32")), "cuda:0", seed=0)
cfg.batchsize = int(os.environ.get("B", "32"))
extra = {}
if os.environ.get("RES256"):            # BASELINE configuration 5 as bench.py --res256 [--fp8] builds it (B = 16)
    from rgbd_gan_amd import kernels
    cfg.ch, cfg.max_resolution, cfg.max_stage = 512, 256, 13
    extra = {"fixed_stage": 12.0}
    images = np.random.RandomState(0).randint(0, 256, (64, 3, 256, 256)).astype("uint8")
    it = DeviceImageIterator(images, int(os.environ.get("B", "16")), "cuda:0", seed=0)
    if os.environ.get("RES256") == "fp8":
        cfg.conv_dtype = "mxfp8"
        kernels.MX8_EMIT = True
gen, dis, opt, upd = build_training(cfg, "cuda:0", iterator=it, nan_check_interval=0, **extra)
upd.iteration = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
for i in range(8):
    upd.update()
torch.cuda.synchronize()
reps, total = 20, 0.0
for gkey, entry in upd._graphs.items():
    g = entry["graph"]
    for _ in range(3):
        g.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    total += ms
    print(f"{gkey[-1]:8s} {ms:7.3f} ms")
print(f"sum      {total:7.3f} ms")
