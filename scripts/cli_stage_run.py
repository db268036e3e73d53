"""A training run of the CLI through three progressive-growing stages with the measured side-stream budgets on: 32x32 (stage 6.x, 60
iterations) -> fade-in to 64x64 (stage 7.x) -> 64x64 (stage 8.x), on a procedural image set, `python train_rgbd.py --config_path <patched ffhq
config>`; prints the log rows, the tuner's reports ("side stream budget at iteration ...") and the run's wall time.  What it shows:
the in-loop measurement (SideBudgetTuner) starting at every new image size, being dropped and restarted by a stage change that
arrives in the middle of one, triggers (log / preview / snapshot) firing at every interval throughout, finite losses.
    python scripts/cli_stage_run.py [--iterations 900] [--batch 8]"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np
import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rgbd_gan_amd.utils.synthetic import procedural_images      # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--iterations", type=int, default=900)
ap.add_argument("--batch", type=int, default=8)
ap.add_argument("--full", action="store_true", help="the whole schedule compressed: 32x32 -> fade -> 64x64 -> fade -> 128x128, a sixth of the "
                "iterations each stage (and a third for the last)")
a = ap.parse_args()
tmp = tempfile.mkdtemp(prefix="rgbd_cli_")
os.makedirs(os.path.join(tmp, "data"))
np.save(os.path.join(tmp, "data", "images.npy"), procedural_images(256, 128, seed=0))
cfg = yaml.safe_load(open(os.path.join(ROOT, "configs", "ffhq_stylegan_occlusion.yml")))
n = a.iterations
# the schedule of updater.py:252-256: stage 6.x (32x32) for the first 60 iterations -- the boundary arrives INSIDE the first image size's
# 120-step measurement, which is dropped -- then the fade-in 7.x to 64x64 (its own measurement) until n/2, then stage 8.x (64x64: the same
# batch and image size, so the measured pair carries over)
si = [0] * 7 + [60, n // 2, 10 * n, 11 * n, 12 * n]
if a.full:
    si = [0] * 7 + [n // 6, n // 3, n // 2, 2 * n // 3, n]
cfg.update(dataset_path=os.path.join(tmp, "data"), out=os.path.join(tmp, "out"), iteration=n, batchsize=a.batch,
           stage_interval=",".join(str(x) for x in si), snapshot_interval=n // 3, display_interval=max(10, n // 18),
           evaluation_sample_interval=n // 3, start_rotation=20, start_occlusion_aware=20)
path = os.path.join(tmp, "cfg.yml")
yaml.safe_dump(cfg, open(path, "w"))
t0 = time.time()
r = subprocess.run([sys.executable, os.path.join(ROOT, "train_rgbd.py"), "-g", "0", "--config_path", path], capture_output=True, text=True)
dt = time.time() - t0
print(f"train_rgbd.py: return code {r.returncode}, {n} iterations at batch {a.batch} in {dt:.1f} s wall (incl. start-up); stage_interval {cfg['stage_interval']}")
if r.returncode != 0:
    print(r.stderr[-3000:])
    sys.exit(1)
for ln in r.stdout.splitlines():
    if ln.startswith("side stream budget"):
        print(ln)
    elif ln.startswith("{"):
        e = json.loads(ln)
        print("  it %5d  stage %.3f  %3dx%-3d  loss_rotate %s  gen/adv %.3f  dis/adv %.3f  elapsed %.1f s" % (
            e["iteration"], e["stage"], e["image_size"], e["image_size"],
            ("%.4f" % e["gen/loss_rotate"]) if "gen/loss_rotate" in e else "  -   ", e["gen/loss_adv"], e["dis/loss_adv"], e["elapsed_time"]))
print("files:", sorted(os.listdir(os.path.join(tmp, "out")))[:14], "...")
print("previews:", sorted(os.listdir(os.path.join(tmp, "out", "preview"))))
