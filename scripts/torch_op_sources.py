"""Which torch (non-extension) ops still launch kernels inside a training step, with shapes and the calling line.

Runs one eager step under a TorchDispatchMode and prints every aten op that is not a pure view / allocation, grouped by
(op, shape, dtype, innermost rgbd_gan_amd frame).  Used to find glue worth fusing into the HIP kernels.
    python scripts/torch_op_sources.py [iteration] [config]"""
import os, sys, traceback, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from torch.utils._python_dispatch import TorchDispatchMode
from rgbd_gan_amd.training import DeviceImageIterator, build_training
from rgbd_gan_amd.utils import yaml_utils

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cfg = yaml_utils.load(sys.argv[2] if len(sys.argv) > 2 else os.path.join(root, "configs", "stylegan_shapenet_car.yml"))
images = np.random.RandomState(0).randint(0, 256, (256, 3, 128, 128)).astype("uint8")
it = DeviceImageIterator(images, cfg.batchsize, "cuda:0", seed=0)
gen, dis, opt, upd = build_training(cfg, "cuda:0", iterator=it, nan_check_interval=0)
upd.iteration = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
upd.use_graphs = False
for i in range(2):
    upd.update()
torch.cuda.synchronize()
log = collections.Counter()
NO_KERNEL = ("view", "slice", "detach", "empty", "as_strided", "expand", "alias", "t.default", "permute", "select",
             "unsqueeze", "squeeze", "transpose", "_unsafe_view", "reshape", "unbind", "split", "_local_scalar_dense",
             "is_same_size", "record_stream", "lift_fresh", "_reshape_alias", "set_", "resize_", "sym_")


class Log(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = str(func)
        if any(k in name for k in NO_KERNEL):
            return out
        t = out[0] if isinstance(out, (tuple, list)) and out else out
        where = "<autograd>"
        for fr in reversed(traceback.extract_stack()[:-1]):
            if "rgbd_gan_amd" in fr.filename:
                where = f"{os.path.basename(fr.filename)}:{fr.lineno}"
                break
        shape = tuple(t.shape) if isinstance(t, torch.Tensor) else None
        dt = str(t.dtype).replace("torch.", "") if isinstance(t, torch.Tensor) else ""
        log[(name, shape, dt, where)] += 1
        return out


with Log():
    upd.update()
torch.cuda.synchronize()
print(f"{sum(log.values())} kernel-launching aten calls in one step")
for (op, shape, dt, where), c in sorted(log.items(), key=lambda kv: (kv[0][3], kv[0][0])):
    print(f"{c:3d} x {op:36s} {str(shape):24s} {dt:9s} {where}")
