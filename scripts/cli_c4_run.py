"""configs/deepvoxels_shapenet_car.yml through train_rgbd.py for a few hundred iterations on procedural images, previews and snapshots
on the way, with the DeepVoxels step's early forward / split backward (default) and without (RGBD_DV_PREFETCH=0): both runs must stay
finite (the updater's per-step watch raises otherwise) and end in the same place statistically.
    python scripts/cli_c4_run.py [iterations] > profiles/r06/cli_c4_run.txt"""
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
from rgbd_gan_amd.utils.synthetic import procedural_images   # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 400
tmp = tempfile.mkdtemp()
os.makedirs(os.path.join(tmp, "data"))
np.save(os.path.join(tmp, "data", "images.npy"), procedural_images(200, 128, seed=0))
print(f"python scripts/cli_c4_run.py {iters}: train_rgbd.py --config_path <deepvoxels_shapenet_car.yml, {iters} iterations, batch 10, "
      f"200 procedural 128x128 images, preview every 50, snapshot every 100>")
for name, env in (("early forward + split backward (default)", {}), ("RGBD_DV_PREFETCH=0", {"RGBD_DV_PREFETCH": "0"})):
    cfg = yaml.safe_load(open(os.path.join(ROOT, "configs", "deepvoxels_shapenet_car.yml")))
    out = os.path.join(tmp, "out_" + ("a" if not env else "b"))
    cfg.update(dataset_path=os.path.join(tmp, "data"), out=out, iteration=iters, snapshot_interval=100, display_interval=50,
               evaluation_sample_interval=50)
    path = os.path.join(tmp, "cfg.yml")
    yaml.safe_dump(cfg, open(path, "w"))
    t = time.time()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "train_rgbd.py"), "--config_path", path], capture_output=True, text=True,
                       env=dict(os.environ, **env))
    dt = time.time() - t
    print(f"\n== {name}: exit code {r.returncode}, {dt:.1f} s wall")
    if r.returncode != 0:
        print(r.stderr[-1500:])
        continue
    log = json.load(open(os.path.join(out, "log")))
    for e in log:
        print("  it %4d  " % e["iteration"] + "  ".join(f"{k}={e[k]:.4g}" for k in ("gen/loss_adv", "gen/loss_rotate", "dis/loss_adv", "dis/loss_gp")))
    print("  files:", sorted(f for f in os.listdir(out) if f.endswith(".npz"))[:6], "...")
