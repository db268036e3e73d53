"""Repeat one training step from identical weights and inputs; report how much the gradients move run to run."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import camera, nets
from rgbd_gan_amd.net import Discriminator, StyleGANGenerator
from rgbd_gan_amd.optimizer import FlatAdam
from rgbd_gan_amd.updater import CameraParamPrior, RGBDUpdater
from rgbd_gan_amd.utils.yaml_utils import Config
CH = 256
gp = nets.init_stylegan(CH, seed=2); dp = nets.init_discriminator(CH, seed=3)
for i in range(6):
    gp[f"gen/outs/{i}/c/W"][-1] = torch.randn(gp[f"gen/outs/{i}/c/W"][-1].shape) * 0.3
gen = StyleGANGenerator(CH, rgbd=True); dis = Discriminator(CH, res=True)
rng = np.random.RandomState(7)
zh = nets.make_hidden(2, CH, rng); z = np.concatenate([zh, zh])
np.random.seed(8); thetas = camera.PosePrior(0.3054, 1.0472, 0).sample(4)
x_real = (rng.randint(0, 256, (4, 3, 128, 128)).astype("float32") / 127.5 - 1)
cfg = Config(dict(generator_architecture="stylegan", stage_interval="0,0,0,0,0,0,0,100000,150000,160000,180000,300000",
                  max_stage=11, start_rotation=2000, start_occlusion_aware=2000, lambda_depth=10, depth_min=1.0,
                  x_rotate=0.3054, y_rotate=1.0472, z_rotate=0, x_translate=0, y_translate=0, z_translate=0, bigan=False))
names_g = ["blocks/5/c1/c/W", "blocks/5/c0/c/W", "blocks/3/c1/c/W", "outs/5/c/W"]
names_d = ["blocks/5/c0/c/W", "blocks/4/c_sc/c/W", "blocks/1/c1/c/W", "blocks/5/c1/c/b", "ins/5/c/W"]
ref = None
def cos(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a @ b) / (a.norm() * b.norm() + 1e-30))
for it in range(int(os.environ.get("N", "12"))):
    gen.load_state_dict(gp); dis.load_state_dict(dp)
    opt = {"map": FlatAdam(gen.mapping.store, 1e-5), "gen": FlatAdam(gen.gen.store, 1e-3), "dis": FlatAdam(dis.store, 3e-3)}
    upd = RGBDUpdater(models=[gen, dis], config=cfg, optimizer=opt, iterator=None, lambda_gp=1.0, smoothing=0.999,
                      total_gpu=1, prior=CameraParamPrior(cfg), fixed_stage=10.0, use_graphs=False)
    upd.iteration = 200000
    upd.update_core(batch=torch.from_numpy(x_real), z_fake_data=torch.from_numpy(z), thetas=thetas)
    cur = {"g/" + n: gen.gen.store[n].grad.clone() for n in names_g}
    cur.update({"d/" + n: dis.store[n].grad.clone() for n in names_d})
    obs = {k: float(v) for k, v in upd.observation.items() if k.startswith(("gen/", "dis/"))}
    if ref is None:
        ref = cur
        print("run 0", obs)
    else:
        print(f"run {it}", " ".join(f"{k}={cos(cur[k], ref[k]):.4f}" for k in cur), {k: round(v, 4) for k, v in obs.items()})
