"""Soak of BASELINE configuration 5 (256x256, ch 512, per-GPU batch 16, stage 12): N training steps with the MXFP8 convs
and the SAME run (same seeds, same data, same pose draws) with the bf16 convs, each in its own process, graphs and the
default two-stream arrangement, finiteness checked every 25 steps.  Prints the loss trajectories side by side: three
mantissa bits must not change how the run behaves (nothing non-finite, losses of the same size and trend), they do change
the digits.
    python scripts/soak_fp8.py [N=400]"""
import json, os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
N = int(sys.argv[1]) if len(sys.argv) > 1 else 400
if os.environ.get("SOAK_CHILD"):
    import numpy as np, torch
    from rgbd_gan_amd.training import DeviceImageIterator, build_training
    from rgbd_gan_amd.utils import yaml_utils
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    config = yaml_utils.load(os.path.join(ROOT, "configs", "stylegan_shapenet_car.yml"))
    config.ch, config.max_resolution, config.max_stage = 512, 256, 13
    config.conv_dtype = os.environ["SOAK_CHILD"]
    device = torch.device("cuda", 0)
    np.random.seed(0); torch.manual_seed(0)
    images = np.random.RandomState(0).randint(0, 256, (64, 3, 256, 256)).astype("uint8")
    it = DeviceImageIterator(images, 16, device, seed=0)
    gen, dis, opt, upd = build_training(config, device, None, iterator=it, nan_check_interval=25, fixed_stage=12.0)
    upd.iteration = 200000
    traj = []
    for i in range(N):
        upd.update()
        upd.iteration += 1
        if i < 8 or i % 50 == 49:
            row = {k: float(v) for k, v in upd.observation.items() if k.startswith(("gen/", "dis/"))}
            row["step"] = i
            row.update({f"|g| {k}": float(o.grad_norm) for k, o in opt.items() if hasattr(o, "grad_norm")})
            traj.append(row)
    torch.cuda.synchronize()
    upd._check_finite()
    for name, link in (("gen", gen), ("dis", dis)):
        for _, store in link.stores:
            assert bool(torch.isfinite(store.flat).all()), f"non-finite parameter in {name}"
    print("TRAJ " + json.dumps(traj))
    sys.exit(0)
out = {}
for dtype in ("mxfp8", "bf16"):
    r = subprocess.run([sys.executable, __file__, str(N)], env=dict(os.environ, SOAK_CHILD=dtype), capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    out[dtype] = json.loads([l for l in r.stdout.splitlines() if l.startswith("TRAJ ")][0][5:])
print(f"{N} steps each, all finite (losses, parameters).  value = mxfp8 / bf16")
for a, b in zip(out["mxfp8"], out["bf16"]):
    print(f"step {a['step']:4d}  " + "  ".join(f"{k}={a[k]:.4f}/{b[k]:.4f}" for k in sorted(a) if k != "step"))
