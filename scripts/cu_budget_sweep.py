"""The side stream's weight-gradient workgroup rule (RGBDUpdater._side_wgrad_pair: a curve through three measured optima)
against measurements at shapes it was NOT fitted on: for every (stage, batch) the set-up measurement
(RGBDUpdater.autotune_side_budget: the rule's pair, neighbours at -32 / +32 / +64 / best -16 / best +16, two more for the second
launch) on this device, and how far the rule's pair is from the fastest one.

    python scripts/cu_budget_sweep.py [--shapes 8:32,9:32,9.5:32,10:16,10:32] [--measure-steps 30] [--repeat 2]

One line per shape: `stage S B=b (HxW): rule d/f = X ms | best d/f = Y ms | rule is +Z % | all: ...`."""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shapes", default="8:32,9:32,9.5:32,10:16,10:32")
    ap.add_argument("--measure-steps", type=int, default=30)
    ap.add_argument("--repeat", type=int, default=2)
    ap.add_argument("--config", default=os.path.join(ROOT, "configs", "stylegan_shapenet_car.yml"))
    args = ap.parse_args()
    from rgbd_gan_amd.training import DeviceImageIterator, build_training
    from rgbd_gan_amd.utils import yaml_utils
    device = torch.device("cuda", 0)
    images = np.random.RandomState(0).randint(0, 256, (256, 3, 128, 128)).astype("uint8")
    for shape in args.shapes.split(","):
        stage, B = float(shape.split(":")[0]), int(shape.split(":")[1])
        config = yaml_utils.load(args.config)
        np.random.seed(0)
        torch.manual_seed(0)
        it = DeviceImageIterator(images, B, device, seed=0)
        gen, dis, opt, upd = build_training(config, device, None, iterator=it, nan_check_interval=0, fixed_stage=stage)
        upd.iteration = 200000
        merged = {}
        for _ in range(args.repeat):                              # the landscape is flat near its optimum: keep the minimum of two passes
            upd.autotune_side_budget(measure_steps=args.measure_steps)
            for k, v in upd.side_budget_tuning["ms_per_step"].items():
                merged[k] = min(v, merged.get(k, v))
            upd._side_wgrad_tuned.clear()
        t = upd.side_budget_tuning
        rule = f"{t['rule'][0]}/{t['rule'][1]}"
        best = min(merged, key=merged.get)
        print(f"stage {stage:g} B={B} ({t['shape'][1]}x{t['shape'][2]}): rule {rule} = {merged[rule]:.3f} ms | best {best} = "
              f"{merged[best]:.3f} ms | rule is {100 * (merged[rule] / merged[best] - 1):+.1f} % | all: "
              + "  ".join(f"{k} {v:.3f}" for k, v in sorted(merged.items(), key=lambda kv: [int(x) for x in kv[0].split('/')])),
              flush=True)
        del gen, dis, opt, upd, it
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
