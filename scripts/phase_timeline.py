"""GPU timeline of the captured phases of a few steady-state steps (start - end in ms after the step's first phase)."""
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
marks, orig = [], upd._run_phase
def rp(name, fn, st, key, stream=None, **kw):
    s = stream if stream is not None else torch.cuda.current_stream()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(s); orig(name, fn, st, key, stream, **kw); e1.record(s)
    marks.append((name, e0, e1))
upd._run_phase = rp
steps = 4
for i in range(steps):
    marks.append(("|", None, None)); upd.update()
torch.cuda.synchronize()
row, t0 = [], None
for name, a, b in marks + [("|", None, None)]:
    if name == "|":
        if row:
            print(" ".join(row))
        row, t0 = [], None
        continue
    t0 = t0 or a
    row.append(f"{name}: {t0.elapsed_time(a):.2f}-{t0.elapsed_time(b):.2f}")
