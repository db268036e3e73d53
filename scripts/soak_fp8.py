"""Does `conv_dtype: mxfp8` train?  Long runs of the training step with the MXFP8 convolutions beside the SAME runs on bf16, over
several seeds and several fp8 coverages (functional.apply_mx8_coverage), each run in its own process, graphs and the default
two-stream arrangement.

    python scripts/soak_fp8.py [--steps 2000] [--seeds 0,1,2] [--variants bf16,all,fprop_only,...] [--size c128|c5]
                               [--window 100] [--out profiles/r06/soak_fp8_ablation.txt]

  --size c128 (default): 128x128, ch 256, B = 16, stage 10, MX8_MIN_TILES = 0 (the fp8 kernels run from 16x16 images up): the
         size the GPU budget affords for 2000 steps x seeds x variants;   c5: BASELINE configuration 5 (256x256, ch 512, B = 16)
  real images: PROCEDURAL (no data set in this container, and uniform noise -- what round 5's soak fed the discriminator -- is a
         game D wins outright: gen/loss_adv ~ 18 from step 50 on): shaded ellipsoid "bodies" on a two-colour gradient ground,
         random pose / size / colours, 256 of them, so that D has something to model and the generator something to match
  a seed fixes the networks' initialisation, the latent stream, the pose draws and the batch order; the data set is the same for
  every run

Reports, per variant, window means (over `window` steps, read every step) of gen/loss_rotate, gen/loss_adv, dis/loss_adv,
dis/loss_gp and the optimizers' gradient norms at a few points of the run, per seed and as mean / min / max over seeds, and
the verdict asked for in VERDICT round 5 item 3: does a variant's final loss_rotate lie inside the bf16 seed spread?"""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
KEYS = ("gen/loss_rotate", "gen/loss_adv", "dis/loss_adv", "dis/loss_gp", "|g| gen", "|g| dis", "|g| map")


from rgbd_gan_amd.utils.synthetic import procedural_images   # noqa: E402


def child(args):
    import numpy as np
    import torch
    from rgbd_gan_amd import kernels
    from rgbd_gan_amd.training import DeviceImageIterator, build_training
    from rgbd_gan_amd.utils import yaml_utils
    variant, seed = os.environ["SOAK_VARIANT"], int(os.environ["SOAK_SEED"])
    config = yaml_utils.load(os.path.join(ROOT, "configs", "stylegan_shapenet_car.yml"))
    if args.size == "c5":
        config.ch, config.max_resolution, config.max_stage = 512, 256, 13
        side, stage = 256, 12.0
    else:
        side, stage = 128, 10.0
        kernels.MX8_MIN_TILES = 0
    if variant != "bf16":
        config.conv_dtype, config.mxfp8_coverage = "mxfp8", variant
    device = torch.device("cuda", 0)
    np.random.seed(seed)
    torch.manual_seed(seed)
    images = procedural_images(256, side, seed=0)
    it = DeviceImageIterator(images, 16, device, seed=seed)
    gen, dis, opt, upd = build_training(config, device, None, iterator=it, nan_check_interval=0, fixed_stage=stage)
    upd.iteration = 200000
    sums, rows = {k: 0.0 for k in KEYS}, []
    for i in range(args.steps):
        upd.update()
        obs = upd.observation
        for k in KEYS:
            v = opt[k[4:]].grad_norm if k.startswith("|g| ") else obs[k]
            sums[k] += float(v)
        if (i + 1) % args.window == 0:
            row = {k: sums[k] / args.window for k in KEYS}
            row["step"] = i + 1
            if not all(np.isfinite(v) for v in row.values()):
                print("TRAJ " + json.dumps({"rows": rows, "nonfinite_at": i + 1}))
                sys.exit(0)
            rows.append(row)
            sums = {k: 0.0 for k in KEYS}
    torch.cuda.synchronize()
    finite = all(bool(torch.isfinite(store.flat).all()) for link in (gen, dis) for _, store in link.stores)
    print("TRAJ " + json.dumps({"rows": rows, "nonfinite_at": None if finite else args.steps}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--seeds", default="0,1,2")
    ap.add_argument("--variants", default="bf16,all,fprop_only,dis_only,gen_fprop_only,gen_skip_last2")
    ap.add_argument("--size", default="c128", choices=("c128", "c5"))
    ap.add_argument("--window", type=int, default=100)
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    if os.environ.get("SOAK_VARIANT"):
        return child(args)
    import numpy as np
    seeds = [int(s) for s in args.seeds.split(",")]
    variants = args.variants.split(",")
    runs = {}
    for v in variants:
        for s in seeds:
            r = subprocess.run([sys.executable, os.path.abspath(__file__)] + sys.argv[1:],
                               env=dict(os.environ, SOAK_VARIANT=v, SOAK_SEED=str(s)), capture_output=True, text=True)
            if r.returncode != 0:
                runs[(v, s)] = {"rows": [], "error": r.stderr[-800:]}
                continue
            runs[(v, s)] = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("TRAJ ")][0][5:])
    lines = [f"python scripts/soak_fp8.py {' '.join(sys.argv[1:])}",
             f"{args.steps} steps, size {args.size}, seeds {seeds}; every number is a mean over a {args.window}-step window ending at "
             f"`step`; real images procedural (shaded ellipsoids), B = 16"]
    marks = sorted({args.window, args.steps // 4 // args.window * args.window, args.steps // 2 // args.window * args.window,
                    args.steps // args.window * args.window} - {0})
    final = {}
    for v in variants:
        lines.append(f"\n== {v}")
        for s in seeds:
            run = runs[(v, s)]
            if run.get("error"):
                lines.append(f"  seed {s}: FAILED {run['error']!r}")
                continue
            if run["nonfinite_at"]:
                lines.append(f"  seed {s}: NON-FINITE at step {run['nonfinite_at']}")
            for row in run["rows"]:
                if row["step"] in marks:
                    lines.append(f"  seed {s} step {row['step']:5d}  " + "  ".join(f"{k}={row[k]:.4g}" for k in KEYS))
        ends = [runs[(v, s)]["rows"][-1] for s in seeds if runs[(v, s)]["rows"] and not runs[(v, s)].get("nonfinite_at")]
        if ends:
            final[v] = {k: [e[k] for e in ends] for k in KEYS}
            lines.append(f"  final window over seeds (mean [min, max]):  " + "  ".join(
                f"{k}={np.mean(final[v][k]):.4g} [{min(final[v][k]):.4g}, {max(final[v][k]):.4g}]" for k in KEYS))
    if "bf16" in final:
        lo, hi = min(final["bf16"]["gen/loss_rotate"]), max(final["bf16"]["gen/loss_rotate"])
        lines.append(f"\nbf16 seed spread of the final gen/loss_rotate: [{lo:.4g}, {hi:.4g}]")
        for v in variants:
            if v == "bf16" or v not in final:
                continue
            vals = final[v]["gen/loss_rotate"]
            inside = sum(lo * 0.999 <= x <= hi * 1.001 for x in vals)
            lines.append(f"  {v:28s} final loss_rotate {['%.4g' % x for x in vals]}  mean {np.mean(vals):.4g} = "
                         f"{np.mean(vals) / np.mean(final['bf16']['gen/loss_rotate']):.2f} x bf16's mean; {inside} of {len(vals)} seeds inside "
                         f"the bf16 spread; max {max(vals):.4g} {'<=' if max(vals) <= hi * 1.001 else '>'} bf16 max")
    text = "\n".join(lines)
    print(text)
    if args.out:
        with open(os.path.join(ROOT, args.out) if not os.path.isabs(args.out) else args.out, "w") as f:
            f.write(text + "\n")


if __name__ == "__main__":
    main()
