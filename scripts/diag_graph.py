"""Diagnostic (GPU): eager vs eager vs graph-replayed steps from identical seeds."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rgbd_gan_amd.training import DeviceImageIterator, build_training
from rgbd_gan_amd.utils.yaml_utils import Config
cfg = dict(generator_architecture="stylegan", ch=256, stage_interval="0,0,0,0,0,0,0,100000,150000,160000,180000,300000",
           max_stage=11, start_rotation=2000, start_occlusion_aware=2000, lambda_depth=10, depth_min=1.0,
           x_rotate=0.3054, y_rotate=1.0472, z_rotate=0, x_translate=0, y_translate=0, z_translate=0, bigan=False,
           adam_alpha_g=0.001, adam_alpha_d=0.003, adam_beta1=0.0, adam_beta2=0.999, lambda_gp=1.0, smoothing=0.999,
           res_dis=True, sn=False, enable_blur=False)
images = np.random.RandomState(0).randint(0, 256, (16, 3, 128, 128)).astype("uint8")
def run(use_graphs, nsteps=5):
    np.random.seed(11); torch.manual_seed(11); torch.cuda.manual_seed(11)
    it = DeviceImageIterator(images, 4, "cuda:0", seed=3)
    gen, dis, opt, upd = build_training(Config(cfg), "cuda:0", iterator=it, fixed_stage=8.0, use_graphs=use_graphs,
                                        graph_warmup=2, nan_check_interval=0)
    upd.iteration = 200000
    zgen = torch.Generator().manual_seed(5)
    hist = []
    for _ in range(nsteps):
        zh = torch.randn(2, 512, 1, 1, generator=zgen)
        zh = zh / torch.sqrt((zh * zh).sum(dim=1, keepdim=True) / 256 + 1e-8)
        upd.update_core(z_fake_data=torch.cat([zh, zh])); upd.iteration += 1
        torch.cuda.synchronize()
        hist.append({k: round(float(v), 4) for k, v in upd.observation.items() if "/" in k} |
                    {"gn": [round(float(opt[k].grad_norm), 3) for k in ("map", "gen", "dis")]})
    return hist
a = run(False); b = run(False); c = run(True)
for i in range(5):
    print("step", i); print("  eagerA", a[i]); print("  eagerB", b[i]); print("  graph ", c[i])

print("---- locate non-finite / huge gradients in graph mode")
np.random.seed(11); torch.manual_seed(11); torch.cuda.manual_seed(11)
it = DeviceImageIterator(images, 4, "cuda:0", seed=3)
gen, dis, opt, upd = build_training(Config(cfg), "cuda:0", iterator=it, fixed_stage=8.0, use_graphs=True,
                                    graph_warmup=2, nan_check_interval=0)
upd.iteration = 200000
zgen = torch.Generator().manual_seed(5)
for step in range(5):
    zh = torch.randn(2, 512, 1, 1, generator=zgen)
    zh = zh / torch.sqrt((zh * zh).sum(dim=1, keepdim=True) / 256 + 1e-8)
    upd.update_core(z_fake_data=torch.cat([zh, zh])); upd.iteration += 1
    torch.cuda.synchronize()
    rows = []
    for n in dis.store.names:
        g = dis.store[n].grad
        rows.append((float(g.abs().max()), int((~torch.isfinite(g)).sum()), n))
    rows.sort(reverse=True, key=lambda r: (r[1], r[0]))
    print("step", step, "norm", float(opt["dis"].grad_norm), rows[:4])
