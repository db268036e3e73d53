"""Soak: N training steps at stage 10 (graphs, default arrangement) with a finiteness check every 25 steps, then the same
start with the register-staged reference kernels (RGBD_CONV_VARIANT=1 in a child process): the loss trajectories of the
first steps must agree (same math, different kernels), and nothing may go non-finite.
    python scripts/soak.py [N=400]"""
import json, os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
N = int(sys.argv[1]) if len(sys.argv) > 1 else 400
if os.environ.get("SOAK_CHILD"):
    import numpy as np, torch
    from rgbd_gan_amd.training import DeviceImageIterator, build_training
    from rgbd_gan_amd.utils import yaml_utils
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    config = yaml_utils.load(os.path.join(ROOT, "configs", "stylegan_shapenet_car.yml"))
    device = torch.device("cuda", 0)
    np.random.seed(0); torch.manual_seed(0)
    images = np.random.RandomState(0).randint(0, 256, (256, 3, 128, 128)).astype("uint8")
    it = DeviceImageIterator(images, config.batchsize, device, seed=0)
    gen, dis, opt, upd = build_training(config, device, None, iterator=it, nan_check_interval=25)
    upd.iteration = 200000
    traj = []
    for i in range(N):
        upd.update()
        upd.iteration += 1
        if i < 12 or i % 50 == 49:
            traj.append({k: float(v) for k, v in upd.observation.items() if k.startswith(("gen/", "dis/"))})
    torch.cuda.synchronize()
    upd._check_finite()
    print("TRAJ " + json.dumps(traj))
    sys.exit(0)
out = {}
for variant in ("0", "1"):
    env = dict(os.environ, SOAK_CHILD="1", RGBD_CONV_VARIANT=variant)
    if variant != "0":       # the reference kernels live in the debug library (python -m rgbd_gan_amd.build --debug)
        env["RGBD_LIB_PATH"] = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "rgbd_gan_amd",
                                            "librgbdgan_hip_debug.so")
    r = subprocess.run([sys.executable, __file__, str(N if variant == "0" else 12)], env=env, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    out[variant] = json.loads([l for l in r.stdout.splitlines() if l.startswith("TRAJ ")][0][5:])
for i in range(12):
    a, b = out["0"][i], out["1"][i]
    print(i, " ".join(f"{k}={a[k]:.4f}/{b[k]:.4f}" for k in sorted(a)))
print("late:", out["0"][-1])
