"""How close does the bf16-emulating oracle get to the engine?  Generator output / discriminator logits per stage:
rel-L2 (engine vs fp32 oracle) against rel-L2 (engine vs emulating oracle), plus the run-to-run noise of the engine."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import camera, nets
from rgbd_gan_amd.net import Discriminator, StyleGANGenerator

CH = 256
def rel(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a - b).norm() / (b.norm() + 1e-30))

gp = nets.init_stylegan(CH, seed=2)
dp = nets.init_discriminator(CH, seed=3)
torch.manual_seed(0)
for i in range(6):
    gp[f"gen/outs/{i}/c/W"][-1] = torch.randn(gp[f"gen/outs/{i}/c/W"][-1].shape) * 0.1
gen = StyleGANGenerator(CH, rgbd=True); dis = Discriminator(CH, res=True)
gen.load_state_dict(gp); dis.load_state_dict(dp)
rng = np.random.RandomState(7)
zh = nets.make_hidden(2, CH, rng); z = np.concatenate([zh, zh])
np.random.seed(8)
t9 = camera.theta9(camera.PosePrior(0.3054, 1.0472, 0).sample(4))
for stage in (2.0, 4.0, 6.0, 8.0, 9.5, 10.0):
    with torch.no_grad():
        x_e = gen(z, stage, t9).cpu()
        x_e2 = gen(z, stage, t9).cpu()
        x_f = nets.stylegan_generator(gp, z, stage, t9)
        with nets.bf16_emulation():
            x_m = nets.stylegan_generator(gp, z, stage, t9)
        y_e = dis(x_e[:, :3].cuda().contiguous(), stage).cpu()
        y_f = nets.discriminator(dp, x_e[:, :3], stage)
        with nets.bf16_emulation():
            y_m = nets.discriminator(dp, x_e[:, :3], stage)
    print(f"stage {stage}: G rgb  engine-vs-fp32 {rel(x_e[:, :3], x_f[:, :3]):.2e}  engine-vs-emul {rel(x_e[:, :3], x_m[:, :3]):.2e}  "
          f"emul-vs-fp32 {rel(x_m[:, :3], x_f[:, :3]):.2e}  run-to-run {rel(x_e, x_e2):.2e} | "
          f"depth e-vs-emul {rel(x_e[:, 3], x_m[:, 3]):.2e} | D logits (same input) engine-vs-fp32 {rel(y_e, y_f):.2e} "
          f"engine-vs-emul {rel(y_e, y_m):.2e}  y={y_e.flatten()[:2].tolist()} {y_m.flatten()[:2].tolist()}", flush=True)
