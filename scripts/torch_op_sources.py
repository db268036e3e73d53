"""Which torch (non-extension) ops still run inside a training step, with shapes and the calling line.

Runs one eager step under a TorchDispatchMode and prints every aten op whose output has >= MIN elements, grouped by
(op, shape, dtype, innermost rgbd_gan_amd frame).  Used to find elementwise glue worth fusing into the HIP kernels."""
import os, sys, traceback, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from torch.utils._python_dispatch import TorchDispatchMode
from rgbd_gan_amd.training import DeviceImageIterator, build_training
from rgbd_gan_amd.utils import yaml_utils

MIN = int(os.environ.get("MIN_NUMEL", 65536))
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cfg = yaml_utils.load(os.path.join(root, "configs", "stylegan_shapenet_car.yml"))
images = np.random.RandomState(0).randint(0, 256, (256, 3, 128, 128)).astype("uint8")
it = DeviceImageIterator(images, 32, "cuda:0", seed=0)
gen, dis, opt, upd = build_training(cfg, "cuda:0", iterator=it, nan_check_interval=0)
upd.iteration = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
upd.graph_phases = ()
for i in range(2):
    upd.update()
torch.cuda.synchronize()
log = collections.Counter()


class Log(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        t = out[0] if isinstance(out, (tuple, list)) and out else out
        n = max([a.numel() for a in list(args) + [t] if isinstance(a, torch.Tensor)] or [0])
        if n >= MIN:
            where = "<autograd>"
            for fr in reversed(traceback.extract_stack()[:-1]):
                if "rgbd_gan_amd" in fr.filename:
                    where = f"{os.path.basename(fr.filename)}:{fr.lineno}"
                    break
            shape = tuple(t.shape) if isinstance(t, torch.Tensor) else None
            dt = str(t.dtype).replace("torch.", "") if isinstance(t, torch.Tensor) else ""
            log[(str(func), shape, dt, where)] += 1
        return out


with Log():
    upd.update()
torch.cuda.synchronize()
for (op, shape, dt, where), c in sorted(log.items(), key=lambda kv: -kv[1] * (np.prod(kv[0][1]) if kv[0][1] else 1)):
    print(f"{c:3d} x {op:40s} {str(shape):28s} {dt:9s} {where}")
