"""Diagnostic (GPU): per-parameter gradient agreement engine vs oracle for one training step."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import camera, nets, step
from tests.test_model_gpu import _models, _inputs, CFG, cosine
from rgbd_gan_amd.optimizer import FlatAdam
from rgbd_gan_amd.updater import CameraParamPrior, RGBDUpdater
from rgbd_gan_amd.utils.yaml_utils import Config

gp, dp, gen, dis = _models(seed=2)
z, thetas, x_real = _inputs(4, seed=7)
for i in range(6):
    gp[f"gen/outs/{i}/c/W"][-1] = torch.randn(gp[f"gen/outs/{i}/c/W"][-1].shape) * 0.3
gen.load_state_dict(gp)
gpl = {k: v.clone().requires_grad_(True) for k, v in gp.items()}
dpl = {k: v.clone().requires_grad_(True) for k, v in dp.items()}
omap = {k: v for k, v in gpl.items() if k.startswith("mapping/")}
ogen = {k: v for k, v in gpl.items() if k.startswith("gen/")}
oopt = {"map": step.ChainerAdam(omap, 1e-5), "gen": step.ChainerAdam(ogen, 1e-3), "dis": step.ChainerAdam(dpl, 3e-3)}
ref = step.rgbd_step(gpl, dpl, oopt, x_real, z, thetas, 10.0, CFG, 200000)
cfg = Config(dict(generator_architecture="stylegan", stage_interval="0,0,0,0,0,0,0,100000,150000,160000,180000,300000",
                  max_stage=11, start_rotation=2000, start_occlusion_aware=2000, lambda_depth=10, depth_min=1.0,
                  x_rotate=0.3054, y_rotate=1.0472, z_rotate=0, x_translate=0, y_translate=0, z_translate=0, bigan=False))
opt = {"map": FlatAdam(gen.mapping.store, 1e-5), "gen": FlatAdam(gen.gen.store, 1e-3), "dis": FlatAdam(dis.store, 3e-3)}
upd = RGBDUpdater(models=[gen, dis], config=cfg, optimizer=opt, iterator=None, lambda_gp=1.0, smoothing=0.999,
                  total_gpu=1, prior=CameraParamPrior(cfg), fixed_stage=10.0)
upd.iteration = 200000
upd.update_core(batch=torch.from_numpy(x_real), z_fake_data=torch.from_numpy(z), thetas=thetas)
print({k: float(v) for k, v in upd.observation.items()})
print({k: v for k, v in ref.items() if k != "x_fake"})
rows = []
for store, prefix, src in ((gen.mapping.store, "mapping/", gpl), (gen.gen.store, "gen/", gpl), (dis.store, "", dpl)):
    for n in store.names:
        b = src[prefix + n].grad
        if b is None or float(b.abs().sum()) == 0:
            continue
        a = store[n].grad.cpu()
        rows.append((cosine(a, b), float(a.norm() / b.norm()), prefix + n))
rows.sort()
for r in rows[:25]:
    print("%.4f  ratio %.3f  %s" % r)
print("median cosine", np.median([r[0] for r in rows]), "n", len(rows))
