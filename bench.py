#!/usr/bin/env python3
"""Throughput of the RGBD-GAN training step (G + D + 3D-consistency loss) on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--config configs/stylegan_shapenet_car.yml]

A "step" is one RGBDUpdater.update_core(): generator step (forward, D forward with frozen weights, warp loss,
backward, clipped Adam on mapping + synthesis) and discriminator step (fake + real forward, R1 double backward,
backward, clipped Adam), on synthetic data already resident in HBM.  One process per GPU, gradients all-reduced
over RCCL.  Two ways to get N ranks (train_rgbd.py:103-121 of the reference: one MPI process per GPU):
  * `python bench.py --gpus N` with no WORLD_SIZE in the environment: this process stays off the GPU, starts N
    children of itself (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set), relays rank 0's JSON line and exits non-zero
    if any rank fails;
  * under `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N` every process is a rank.
A WORLD_SIZE that disagrees with --gpus is an error, not a silent one-rank run.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# algorithmic conv+linear FLOPs per generated image of one step at stage 10, ch=256:
# 3 F_G + 11 F_D with F_G = 10.45, F_D = 24.11 GFLOP is the step as the reference executes it minus D's weight
# gradients in the generator step.  This engine evaluates and back-propagates D(x_fake) once instead of twice (the
# reference's second forward recomputes identical values, and its two input-gradient passes differ by a per-sample
# scalar), and folds the adversarial backward on the reals into the R1 double backward (again a per-sample scalar),
# so it EXECUTES 3 F_G + 7 F_D; the whole-step roofline fraction is reported on executed work.
STEP_GFLOP_PER_IMAGE = 296.6 - 4 * 24.11
MFMA_BF16_PEAK_TFLOPS = 2500.0
MFMA_FP8_PEAK_TFLOPS = 5000.0   # dense, block-scaled K = 128 form (MI355X_MICROARCH.md, Matrix cores)
HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md: HBM3E ~8 TB/s


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=None, help="ranks (one per GPU); default: WORLD_SIZE or 1")
    ap.add_argument("--steps", type=int, default=100)      # 20 warm-up + 100 timed steps
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=None, help="per-GPU batch (default: the config's batchsize, 32)")
    ap.add_argument("--config", default=os.path.join(ROOT, "configs", "stylegan_shapenet_car.yml"))
    ap.add_argument("--iteration", type=int, default=200000, help="steady state: stage 10, rotation + occlusion on")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--res256", action="store_true",
                    help="BASELINE configuration 5's networks (ch=512, the 256x256 block, stage 12, per-GPU batch 16); with "
                         "--fp8 on its MXFP8 conv kernels, else on the bf16 ones; never the default line")
    ap.add_argument("--fp8", action="store_true",
                    help="conv_dtype: mxfp8 -- fprop / dgrad of the 3x3 convolutions on block-scaled fp8 operands "
                         "(v_mfma_scale_f32_16x16x128_f8f6f4), weight gradients on the bf16 kernels")
    ap.add_argument("--fp8-coverage", default=None,
                    help="with --fp8: which launches take fp8 operands (YAML key mxfp8_coverage, functional.MX8_COVERAGES); default: "
                         "the engine's default coverage")
    ap.add_argument("--arrangements", action="store_true",
                    help="also time the single-stream arrangement on the same box (extra key, never `value`)")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="the default command also runs short lines of configurations 3 (per-GPU shape), 4 and 5 (bf16 / fp8) "
                         "into `other_configs`; this switches them off")
    ap.add_argument("--stage", type=float, default=None,
                    help="pin the progressive-growing stage (default: the schedule's stage at --iteration, 10 = 128x128 without "
                         "fade-in); 9.5 = the fade-in arrangement between 64x64 and 128x128 (two output heads, blended reals)")
    ap.add_argument("--autotune", action="store_true", help="(kept for old command lines: the measurement is the default now)")
    ap.add_argument("--no-tune", action="store_true",
                    help="take the rule of thumb for the side stream's workgroup counts instead of measuring them on this device "
                         "before the warm-up (RGBDUpdater(tune_side_budget=True) + finish_tuning: ~120 ordinary steps)")
    ap.add_argument("--other-steps", type=int, default=30)
    ap.add_argument("--other-only", type=str, default="", help="comma list: run only these of the other configurations")
    ap.add_argument("--mx8-standalone-quantiser", action="store_true",
                    help="A/B aid for --fp8: every conv input through rgbd_quantize_mxfp8 instead of the producers' epilogues")
    return ap.parse_args()


def kernel_source_sha16():
    """Identity of the kernel sources a profile was taken with (there is no .git on the GPU box)."""
    import hashlib
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "rgbd_gan_amd", "csrc")
    for name in sorted(os.listdir(csrc)):
        if name.endswith((".hip", ".h")):
            with open(os.path.join(csrc, name), "rb") as f:
                h.update(f.read())
    return h.hexdigest()[:16]


def workload_key(args):
    """Name of the workload a command line runs: profiles are keyed by it (a PMC pass of one workload says nothing about
    the layer shapes and batch of another)."""
    cfg = os.path.basename(args.config)
    if args.res256:
        return "res256_fp8" if args.fp8 else "res256"
    if cfg == "deepvoxels_shapenet_car.yml":
        return "c4"
    if cfg == "ffhq_stylegan_occlusion.yml" and args.batch == 8:
        return "c3_b8"
    if cfg == "stylegan_shapenet_car.yml" and args.batch in (None, 32) and not args.fp8 and args.stage == 9.5:
        return "c2_fade"
    if cfg == "stylegan_shapenet_car.yml" and args.batch in (None, 32) and not args.fp8 and args.stage in (None, 10.0):
        return "default"
    return None


def pmc_traffic(kernel, workload):
    """HBM bytes per launch of `kernel` from the newest committed PMC passes of THIS workload
    (profiles/rNN/bench_pmc_traffic_<workload>.json, written by scripts/collect_profiles.sh: rocprofv3 --pmc FETCH_SIZE /
    WRITE_SIZE in separate passes of the same bench.py command, corrected as MI355X_MICROARCH.md prescribes:
    2 * FETCH_SIZE + WRITE_SIZE, KB).  Returns (bytes, provenance): bytes is None when no profile of this workload holds the
    kernel OR the profile was taken with different kernel sources than the ones running now."""
    import glob
    if workload is None:
        return None, {"profile": None, "note": "no PMC profile is kept for this command line"}
    paths = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", f"bench_pmc_traffic_{workload}.json")))
    if not paths:
        return None, {"profile": None, "workload": workload}
    path = paths[-1]
    with open(path) as fjson:
        prof = json.load(fjson)
    prov = {"profile": os.path.relpath(path, ROOT), "workload": workload, "profile_workload": prof.get("workload"),
            "profile_source_sha16": prof.get("source_sha16"),
            "profile_commit": prof.get("commit"), "running_source_sha16": kernel_source_sha16()}
    if prov["profile_workload"] != workload:
        prov["note"] = "the profile was recorded on another workload: traffic withheld"
        return None, prov
    if prov["profile_source_sha16"] != prov["running_source_sha16"]:
        prov["note"] = "kernel sources changed since the PMC passes: traffic withheld"
        return None, prov
    ks = prof["kernels"]
    # launch-profile labels of the DeepVoxels path that stand for one or several kernels of the library
    alias = {"trilinear_bwd_kernel": ("trilinear_bwd_brick_kernel", "trilinear_bwd_scatter_kernel"), "trilinear_fwd_kernel": ("trilinear_fwd_frustum_kernel", "trilinear_fwd_fm_kernel"),
             "occlusion_accum_fwd_kernel": ("occ_fwd_fused_kernel<32>", "occ_score_kernel", "occ_scan_kernel", "occ_compose_kernel"),
             "occlusion_accum_bwd_kernel": ("occ_bwd_scan_kernel", "occ_bwd_mlp_kernel", "occ_bwd_mlp4_kernel", "occ_bwd_dw_kernel", "occ_bwd_params_kernel")}
    if kernel in alias:
        per_launch = [ks[n]["hbm_bytes_per_launch"] for n in alias[kernel] if n in ks]
        return (int(sum(per_launch)) if per_launch else None), prov
    base = kernel.split("<")[0]
    tags = kernel.split("<")[1].rstrip(">").split(",") if "<" in kernel else [""]
    width = tags[0]
    # the launch-profile labels fold the upsampling variant into one name and tag the forms of the 3x3 kernels
    # (conv3x3_sp_kernel<BN, UPS, KO, EPI, MX, EMIT>, conv3x3_dw_kernel<BN, UPS, EPI, KO>: "<128>" = EPI 0,
    # "<128,actgrad>" = 1, "<128,stats>" = 2, ",mxfp8" = MX)
    want_epi = "1" if "actgrad" in tags else "2" if "stats" in tags else "0"
    want_mx = "true" if "mxfp8" in tags else "false"
    tot = cnt = 0
    if base == "conv_wgrad_kernel":            # "conv_wgrad_kernel<9>+reduce": the batched body's launches (the reduction's
        base, width = "conv_wgrad_multi_kernel", width.split(">")[0]     # slab traffic is a separate, small kernel)
    for name, v in ks.items():
        if not name.startswith(base):
            continue
        if base == "conv3x3_sp_kernel":
            targs = [a.strip() for a in name.split("<", 1)[1].rstrip(">").split(",")]
            mx = targs[4] if len(targs) > 4 else "false"
            if targs[0] != width or targs[2] != "0" or targs[3] != want_epi or mx != want_mx:
                continue
        elif base == "conv3x3_dw_kernel":
            targs = [a.strip() for a in name.split("<", 1)[1].rstrip(">").split(",")]
            if targs[0] != width or targs[2] != want_epi or (len(targs) > 3 and targs[3] != "0"):
                continue
        elif not (name.startswith(base + "<" + width) or name == kernel):
            continue
        tot += v["hbm_bytes_per_launch"] * v["launches"]
        cnt += v["launches"]
    return (round(tot / cnt) if cnt else None), prov


def cpu_baseline(batch=8, threads=None, budget_s=12.0, max_steps=12):
    """The CPU restatement of the same step (oracle/step.py; Chainer is unavailable), timed on the host cores on a
    bounded sample: update_core at 128x128, ch=256, `batch` images, repeated for about `budget_s` seconds."""
    import numpy as np
    import torch
    from oracle import camera, nets, step
    # a bounded sample on a bounded thread count: torch-CPU convolutions at this batch size stop scaling (and
    # collapse with hundreds of threads), so use at most 16 host cores and say so in `cores`
    threads = threads or min(os.cpu_count() or 1, 16)
    torch.set_num_threads(threads)
    ch = 256
    gp = {k: v.requires_grad_(True) for k, v in nets.init_stylegan(ch, seed=0).items()}
    dp = {k: v.requires_grad_(True) for k, v in nets.init_discriminator(ch, seed=1).items()}
    omap = {k: v for k, v in gp.items() if k.startswith("mapping/")}
    ogen = {k: v for k, v in gp.items() if k.startswith("gen/")}
    opt = {"map": step.ChainerAdam(omap, 1e-5), "gen": step.ChainerAdam(ogen, 1e-3), "dis": step.ChainerAdam(dp, 3e-3)}
    rng = np.random.RandomState(0)
    zh = nets.make_hidden(batch // 2, ch, rng)
    z = np.concatenate([zh, zh])
    np.random.seed(2)
    thetas = camera.PosePrior(0.3054, 3.1415, 0).sample(batch)
    x_real = rng.randint(0, 256, (batch, 3, 128, 128)).astype("float32") / 127.5 - 1
    cfg = dict(lambda_gp=1.0, lambda_depth=10, depth_min=0.6, lambda_geometric=2, lambda_rotate=None,
               start_rotation=2000, start_occlusion_aware=2000)
    t0 = time.time()
    steps = 0
    while steps < max_steps and (steps == 0 or time.time() - t0 < budget_s):
        step.rgbd_step(gp, dp, opt, x_real, z, thetas, 10.0, cfg, 200000 + steps)
        steps += 1
    dt = time.time() - t0
    return {"value": batch * steps / dt, "unit": "img/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{steps} update_core, batch {batch}, 128x128, ch=256, fp32 torch-CPU restatement of the reference "
                      f"path (Chainer unavailable), {dt:.1f} s"}


class VerifyingComm:
    """Proxy of a Communicator for the `dp.allreduce_verified` screen: every all-reduce it starts is repeated afterwards on a
    QUIET device from a copy of the same inputs, and the two results are compared bit for bit.  The first one runs where the
    step runs it -- on RCCL's stream beside the step's MFMA kernels on both compute streams -- which is the situation in which
    packed-fp32 arithmetic was found to return wrong lanes on this chip (the library is built without
    those instructions, RCCL's reduction kernels are not ours to build)."""

    def __init__(self, comm):
        self._comm, self.records = comm, []

    def __getattr__(self, name):
        return getattr(self._comm, name)

    def allreduce_async(self, flat):
        pre = flat.clone()                             # on the stream the gradients just became final on
        self.records.append([flat, pre, None])
        return (self._comm.allreduce_async(flat), len(self.records) - 1)

    def wait(self, handle):
        if handle is None:
            return
        work, i = handle
        self._comm.wait(work)
        self.records[i][2] = self.records[i][0].clone()        # what the optimizer is about to read

    def check(self):
        """Call on every rank at the same point, after a device-wide synchronisation.  -> (all equal, buffers compared)"""
        import torch
        import torch.distributed as dist
        ok, n = True, 0
        for flat, pre, post in self.records:
            if post is None:
                continue
            dist.all_reduce(pre, op=dist.ReduceOp.SUM)
            torch.cuda.synchronize()
            ok = ok and bool(torch.equal(pre.view(torch.int32), post.view(torch.int32)))
            n += 1
        self.records = []
        return ok, n


def dp_verify(upd, opt, steps=3):
    """`steps` ordinary training steps with every optimizer's communicator wrapped in VerifyingComm."""
    import torch
    proxies = {}
    for o in opt.values():
        proxies.setdefault(id(o.comm), VerifyingComm(o.comm))
        o.comm = proxies[id(o.comm)]
    ok, n = True, 0
    try:
        for _ in range(steps):
            upd.update()
            torch.cuda.synchronize()
            for p in proxies.values():
                a, b = p.check()
                ok, n = ok and a, n + b
    finally:
        for o in opt.values():
            o.comm = o.comm._comm
    return ok, n


def dp_exposed(upd, steps=5):
    """Event-timed gaps the collectives leave in the two compute streams, mean over `steps` steps (ms): on the main stream from
    the end of gen_b (the generator's gradients final, its all-reduces enqueued) to the start of its optimizer phase; on the side
    stream from the end of dfw + merge to the start of D's optimizer phase.  The main stream's gap is on the step's critical path;
    the side stream's is exposed only as far as it outlasts the generator's backward."""
    import torch
    gaps = {"gen": [], "dis": [], "side_end_before_gen_b_end": []}
    for _ in range(steps):
        upd.timeline = {}
        upd.update()
        torch.cuda.synchronize()
        t = upd.timeline
        if not all(k in t for k in ("gen_b_end", "opt_g_start", "side_end", "opt_d_start")):
            upd.timeline = None
            return None
        gaps["gen"].append(t["gen_b_end"].elapsed_time(t["opt_g_start"]))
        gaps["dis"].append(t["side_end"].elapsed_time(t["opt_d_start"]))
        gaps["side_end_before_gen_b_end"].append(t["side_end"].elapsed_time(t["gen_b_end"]))
    upd.timeline = None
    return {k: sum(v) / len(v) for k, v in gaps.items()}


def launch_ranks(n):
    """`python bench.py --gpus N` outside any launcher: start N ranks of this script (one per GPU) and relay rank 0's
    JSON line.  This parent never imports torch and never touches a GPU (a process that has initialised HIP must not
    fork/exec workers on this pool); the children get what torch.distributed.run would give them."""
    import signal
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    threads = max(1, (os.cpu_count() or n) // n)
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RGBD_BENCH_CHILD="1")
        env.setdefault("OMP_NUM_THREADS", str(threads))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, stderr=sys.stderr, text=True,
                                      start_new_session=True))
    failed = None
    while failed is None and any(p.poll() is None for p in procs):
        for r, p in enumerate(procs):
            if p.poll() not in (None, 0):
                failed = (r, p.returncode)
        time.sleep(0.2)
    for r, p in enumerate(procs):
        if failed is None and p.returncode != 0:
            failed = (r, p.returncode)
    if failed is not None:
        for p in procs:                        # the exact process groups started above, nothing matched by pattern
            if p.poll() is None:
                try:
                    os.killpg(p.pid, signal.SIGTERM)
                except ProcessLookupError:
                    pass
        print(f"bench.py: rank {failed[0]} of {n} exited with code {failed[1]}", file=sys.stderr, flush=True)
        sys.exit(1)
    out = procs[0].stdout.read()
    rows = [ln for ln in out.splitlines() if ln.startswith('{"metric"')]
    if len(rows) != 1 or json.loads(rows[0]).get("n_gpus") != n:
        print(f"bench.py: expected one JSON line with n_gpus={n} from rank 0, got: {out[-2000:]!r}", file=sys.stderr)
        sys.exit(1)
    print(rows[0], flush=True)


def run_workload(args, comm, device):
    """One bench line: build the trainer of `args`' workload, W untimed + K timed steps between barriers, the roofline leg.
    Returns the dict that becomes the JSON line (rank 0 fills the roofline keys)."""
    import numpy as np
    import torch
    from rgbd_gan_amd import kernels
    from rgbd_gan_amd.training import DeviceImageIterator, build_training
    from rgbd_gan_amd.utils import yaml_utils

    config = yaml_utils.load(args.config)
    deepvoxels = config.generator_architecture == "deepvoxels"      # BASELINE config 4: DeepVoxelsUpdater, 64x64, B=10
    B = args.batch or config.batchsize
    side, extra = 128, {}
    if args.res256:
        config.ch, config.max_resolution, config.max_stage = 512, 256, 13
        B = args.batch or 16
        side, extra = 256, {"fixed_stage": 12.0}
    if args.stage is not None and not deepvoxels:
        extra["fixed_stage"] = float(args.stage)
    if args.fp8:
        config.conv_dtype = "mxfp8"
        if args.fp8_coverage:
            config.mxfp8_coverage = args.fp8_coverage
        kernels.MX8_EMIT = not args.mx8_standalone_quantiser
    np.random.seed(2 + comm.rank)
    torch.manual_seed(comm.rank)
    images = np.random.RandomState(comm.rank).randint(0, 256, (256 if side == 128 else 64, 3, side, side)).astype("uint8")
    it = DeviceImageIterator(images, B, device, seed=comm.rank)
    gen, dis, opt, upd = build_training(config, device, comm if comm.active else None, iterator=it,
                                        nan_check_interval=0, **extra)
    upd.iteration = args.iteration
    tuning = None
    if not args.no_tune and not comm.active and getattr(upd, "concurrent_phases", False) and hasattr(upd, "finish_tuning") \
            and not deepvoxels:
        # set-up, before the W warm-up and K timed steps: the side stream's weight-gradient workgroup counts measured on this
        # device at this shape, as the training CLI does in its first ~120 iterations at every new image size (SideBudgetTuner:
        # ordinary training steps).  (N > 1: the rule of thumb + the data-parallel lead -- the re-captures beside RCCL have never
        # run on more than one device.)
        upd.tune_side_budget = True
        upd.finish_tuning()
        tuning = getattr(upd, "side_budget_tuning", None)
        upd.tune_side_budget = False
        upd.iteration = args.iteration

    px = side                     # the step's image side: the progressive stage's (--stage), not the data set's
    if not deepvoxels and hasattr(upd, "_net_max_stage"):
        from rgbd_gan_amd.updater import downsized_size
        px = downsized_size(upd.stage, upd._net_max_stage)

    def sync():
        torch.cuda.synchronize()
        comm.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        upd.update()
    dp_checked = None
    if comm.active and not deepvoxels:
        # data parallel: screen the collectives before anything is timed (three more untimed steps on every rank)
        sync()
        dp_checked = dp_verify(upd, opt)
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        upd.update()
    t_enqueue = time.perf_counter() - t0          # host time to enqueue the steps (GPU still running)
    sync()
    own = time.perf_counter() - t0
    elapsed = comm.max_over_ranks(own)
    fastest = -comm.max_over_ranks(-own)
    upd._check_finite()

    # host cost of enqueueing a step, measured on a short burst right after a sync: over the long timed loop the launch
    # thread runs into the back-pressure of the bounded HIP queues, so its wall time per step converges to the GPU's
    # step time whatever the host really costs
    burst, burst_each = 3, []
    t1 = time.perf_counter()
    for _ in range(burst):
        t2 = time.perf_counter()
        upd.update()
        burst_each.append(time.perf_counter() - t2)
    t_burst = time.perf_counter() - t1
    sync()

    ms = elapsed / args.steps * 1e3
    value = B * comm.size * args.steps / elapsed
    line = {
        "metric": "img/s (G+D+3D-loss step) at 128x128", "value": round(value, 2), "unit": "img/s",
        "n_gpus": comm.size, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "fp8" if args.fp8 else "bf16",
        "data": "synthetic",
        "config": {"workload": (f"{os.path.basename(args.config)} stage {upd.stage:.2f} (64x64), DeepVoxelsUpdater.update_core, "
                                f"voxel generator 32^3 x 32 features -> frustum 56x64x64 -> occlusion compositing -> 2-D "
                                f"renderer, rotation loss on, R1 on") if deepvoxels else
                               (f"{os.path.basename(args.config)} stage {upd.stage:.2f} ({px}x{px}), RGBDUpdater.update_core, "
                                f"StyleGAN ch={config.ch}, rotation+occlusion loss on, R1 on"
                                + (", max_resolution=256 (BASELINE configuration 5's networks)" if args.res256 else "")
                                + (", conv_dtype mxfp8: 3x3 fprop / dgrad on e4m3 operands with E8M0 scales per 32 channels "
                                   "(fp32 accumulate, bf16 activations in HBM), weight gradients and the layers the fp8 kernel "
                                   "does not cover on bf16" if args.fp8 else "")),
                   "per_gpu_batch": B, "global_batch": B * comm.size, "parallelism": f"dp{comm.size}",
                   "arrangement": ("two streams, one graph per phase"
                                   + (", the next step's generator forward on the side stream under this step's dis_fake phase "
                                      f"(sized for {upd.forward_cu_budget} CUs)" if getattr(upd, "prefetch_forward", False) else "")
                                   + (", the renderer's weight gradients on the side stream under the backward pass below the renderer"
                                      if getattr(upd, "split_backward", False) else ""))
                   if getattr(upd, "concurrent_phases", False) else "one stream",
                   # True only if the timed steps were replays of captured HIP graphs (a refused capture is fatal in the
                   # updater: graph_fallback is off)
                   "graphs": bool(getattr(upd, "graphs_in_use", upd.use_graphs)),
                   # compute units the side stream's chip-filling launches are sized for: the
                   # weight-gradient workgroup count measured at set-up on this device (rule of thumb -> chosen), or None
                   "side_stream_budget": ({"cus_3x3": getattr(upd, "side_cu_budget", None),
                                           "wgrad_workgroups_dis_dfw": list(upd._side_wgrad_pair({"B": B, "x_real": torch.empty(0, 0, px, px)}))
                                           if hasattr(upd, "_side_wgrad_pair") and getattr(upd, "side_wgrad_workgroups", 0) is None else
                                           getattr(upd, "side_wgrad_workgroups", None), "measured": tuning}
                                          if getattr(upd, "concurrent_phases", False) else None)},
        "host_enqueue_ms_per_step": round(min(burst_each) * 1e3, 3),
        "host_enqueue_burst_ms": [round(t * 1e3, 3) for t in burst_each],
        "host_enqueue_note": f"wall time of the launch thread for one step: the fastest of a {burst}-step burst after a sync (the "
                             f"first step of a burst meets empty queues; a later one can already wait for a free slot in a "
                             f"bounded hardware queue when a step is long -- its time then is the GPU's, not the host's); burst mean "
                             f"{t_burst / burst * 1e3:.3f} ms; over the timed loop {t_enqueue / args.steps * 1e3:.3f} ms",
    }
    if deepvoxels:
        line["metric"] = "img/s (G+D+3D-loss step) at 64x64, DeepVoxels generator"
    elif args.res256:
        line["metric"] = "img/s (G+D+3D-loss step) at 256x256"
    elif args.stage not in (None, 10.0):
        if px != 128:                   # (the whole-step FLOP count below is stage 10's)
            line["metric"] = f"img/s (G+D+3D-loss step) at {px}x{px} (progressive stage {upd.stage:g})"
    else:
        line["step_tflops_algorithmic"] = round(value * STEP_GFLOP_PER_IMAGE / 1e3, 2)
        line["mfma_roofline_frac_whole_step"] = round(value * STEP_GFLOP_PER_IMAGE / 1e3 / (MFMA_BF16_PEAK_TFLOPS * comm.size), 4)
    if comm.active and not deepvoxels:
        # the data-parallel job describes itself (train_rgbd.py:103-121,154-156): what moved, what of it the step waited for,
        # and whether the sums that arrived are the sums a quiet device computes
        exposed = dp_exposed(upd) if getattr(upd, "concurrent_phases", False) else None
        sync()
        if exposed is not None:
            exposed = {k: round(comm.max_over_ranks(v), 4) for k, v in exposed.items()}
        line["dp"] = {
            "backend": torch.distributed.get_backend(), "world_size": torch.distributed.get_world_size(),
            "allreduce_bytes": {k: int(o.store.grad.numel() * o.store.grad.element_size()) for k, o in opt.items()},
            "allreduce_exposed_ms": exposed["gen"] if exposed else None,
            "allreduce_wait_ms": ({"main_stream_gen": exposed["gen"], "side_stream_dis": exposed["dis"],
                                   "side_end_before_gen_b_end": exposed["side_end_before_gen_b_end"]} if exposed else None),
            "allreduce_exposed_note": "HIP events, mean over 5 steps behind the timed region, max over ranks: the gap between the "
                                      "end of a stream's last compute phase (its all-reduces are enqueued there) and the start of "
                                      "its optimizer phase; the main stream's gap is the step's exposed communication",
            "ms_per_step_rank_spread": [round(fastest / args.steps * 1e3, 3), round(elapsed / args.steps * 1e3, 3)],
            "allreduce_verified": bool(dp_checked[0]) if dp_checked and dp_checked[1] else None,
            "allreduce_verified_buffers": dp_checked[1] if dp_checked else 0,
            "allreduce_verified_note": "3 untimed steps before the timed region: every all-reduce that ran beside the step's MFMA "
                                       "kernels repeated on a quiet device from a copy of its inputs, results compared bit for bit",
            "budgets": {"reserve_cus": getattr(upd, "dp_reserve_cus", None),
                        "side_lead_workgroups": getattr(upd, "dp_side_lead_workgroups", None)}}

    if comm.rank == 0 and not args.no_roofline:
        # per-launch HIP-event timing of the conv kernels over extra (untimed) steps, on the launch stream
        # (eager launches: graph replays bypass the Python wrappers that record the events; one stream: with the
        # two-stream phase overlap an event pair would also time the other stream's kernels sharing the CUs)
        # (data parallel: EVERY rank runs these two steps in this arrangement, see the elif below -- the arrangements issue
        # their all-reduces in different orders, and a communicator matches collectives by order)
        was_graphs, was_concurrent = upd.use_graphs, getattr(upd, "concurrent_phases", False)
        upd.use_graphs, upd.concurrent_phases = False, False
        upd.update()        # one unprofiled step in this arrangement first: whatever switching to it costs once (allocations of
        #                     the eager one-stream step, lazily loaded code) must not land in a kernel's event pair -- it did:
        #                     one 50 ms "launch" of the first conv kernel of the leg, profiles/r06/roofline_leg_outlier.txt
        with kernels.launch_profile() as prof:
            for _ in range(2):
                upd.update()
        summ = prof.summary()
        upd.use_graphs, upd.concurrent_phases = was_graphs, was_concurrent
        table = {k: {"launches": n, "ms": round(t * 1e3, 3), "tflops": round(f / t / 1e12, 1),
                     "avg_us": round(t / n * 1e6, 1), "gbps": round(b / t / 1e9, 1)} for k, (n, t, f, b) in summ.items()}
        timing = ("HIP events on the launch stream around every launch of this kernel in 2 extra eager single-stream "
                  "steps after the timed region (behind one unprofiled step in that arrangement)")
        if deepvoxels:
            # the path's own kernels are the HBM-bound frustum resampling / occlusion compositing (SURVEY.md section 8(d):
            # 4.2 MB read + 29.4 MB written per sample, ~2 x 29.4 MB per sample); the conv stack is shared with config 2
            hbm = {k: v for k, v in summ.items() if k.startswith(("trilinear", "occlusion"))}
            dom = max(hbm, key=lambda k: hbm[k][1])
            n, t, f, b = hbm[dom]
            traffic, provenance = pmc_traffic(dom, workload_key(args))
            line["roofline"] = {"kernel": dom, "bound": "hbm", "achieved": round(b / t / 1e9, 1), "peak": HBM_PEAK_GBPS,
                                "unit": "GB/s", "frac": round(b / t / 1e9 / HBM_PEAK_GBPS, 4), "traffic": traffic,
                                "traffic_provenance": provenance, "launches": n, "avg_launch_us": round(t / n * 1e6, 2),
                                "bytes_per_launch_avg": b / n, "timing": timing}
        else:
            dom = max(summ, key=lambda k: summ[k][1])
            n, t, f, b = summ[dom]
            traffic, provenance = pmc_traffic(dom, workload_key(args))
            peak = MFMA_FP8_PEAK_TFLOPS if "mxfp8" in dom else MFMA_BF16_PEAK_TFLOPS
            line["roofline"] = {"kernel": dom, "bound": "mfma", "achieved": round(f / t / 1e12, 2),
                                "peak": peak, "unit": "TFLOP/s",
                                "frac": round(f / t / 1e12 / peak, 4),
                                "traffic": traffic, "traffic_provenance": provenance, "launches": n,
                                "avg_launch_us": round(t / n * 1e6, 2), "flops_per_launch_avg": f / n, "timing": timing}
            mx = {k: v for k, v in summ.items() if "mxfp8" in k}
            if mx and "mxfp8" not in dom:
                # --fp8: with fprop / dgrad on the fp8 kernels the step's dominant kernel can be the bf16 weight-gradient
                # kernel; the dominant MXFP8 kernel is priced as well, against the fp8 peak
                d8 = max(mx, key=lambda k: mx[k][1])
                n8, t8, f8, b8 = mx[d8]
                tr8, pv8 = pmc_traffic(d8, workload_key(args))
                line["roofline_fp8"] = {"kernel": d8, "bound": "mfma", "achieved": round(f8 / t8 / 1e12, 2),
                                        "peak": MFMA_FP8_PEAK_TFLOPS, "unit": "TFLOP/s",
                                        "frac": round(f8 / t8 / 1e12 / MFMA_FP8_PEAK_TFLOPS, 4), "traffic": tr8,
                                        "traffic_provenance": pv8, "launches": n8, "avg_launch_us": round(t8 / n8 * 1e6, 2),
                                        "flops_per_launch_avg": f8 / n8, "timing": timing}
        line["kernels"] = table
        line["kernels_note"] = "per-kernel totals over the 2 extra eager steps of the roofline leg, not per step"
        if getattr(prof, "outliers", None):
            line["kernels_timing_outliers"] = {"replaced_by_their_shape_median": prof.outliers,
                                               "note": "event pairs > 8x the median of the same kernel on the same shape: a host stall "
                                                       "between the two event records, not a kernel duration (kernels.launch_profile)"}
    elif comm.size > 1 and not args.no_roofline:
        # keep the ranks in lock-step with rank 0's extra steps, in rank 0's arrangement: on one stream the all-reduces go
        # map, gen, dis; on two streams dis, map, gen -- ranks in different arrangements would pair different buffers
        was_graphs, was_concurrent = upd.use_graphs, getattr(upd, "concurrent_phases", False)
        upd.use_graphs, upd.concurrent_phases = False, False
        for _ in range(3):
            upd.update()
        upd.use_graphs, upd.concurrent_phases = was_graphs, was_concurrent
    if comm.rank == 0 and comm.size == 1 and args.arrangements and not deepvoxels:
        # the single-stream arrangement (RGBD_CONCURRENT_PHASES=0), timed beside the default two-stream one on the same
        # box (never `value`)
        it2 = DeviceImageIterator(images, B, device, seed=1)
        _, _, _, upd2 = build_training(config, device, None, iterator=it2, nan_check_interval=0,
                                       concurrent_phases=False)
        upd2.iteration = args.iteration
        n2 = max(10, min(40, args.steps))
        for _ in range(8):
            upd2.update()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        for _ in range(n2):
            upd2.update()
        torch.cuda.synchronize()
        ms2 = (time.perf_counter() - t2) / n2 * 1e3
        line["other_arrangement"] = {
            "name": "one stream (RGBD_CONCURRENT_PHASES=0)", "ms_per_step": round(ms2, 3),
            "img_per_s": round(B / ms2 * 1e3, 1), "steps": n2,
            "note": "same step, same kernels, generator and discriminator phases back to back instead of on two streams"}
        del upd2
    line["_deepvoxels"] = deepvoxels
    del gen, dis, opt, upd, it
    torch.cuda.empty_cache()
    return line


# The other BASELINE configurations, run by the DEFAULT command behind its timed region (never `value`): short runs so that
# every performance claim of DESIGN.md has a driver-visible line.  (name, command-line overrides)
OTHER_CONFIGS = [
    # configuration 2 in a FADE-IN stage (the reference spends 20 000 of its first 180 000 iterations in stage 9.x,
    # updater.py:252-256): two output / input heads, blended reals, 64x64 trunk + the 128x128 block
    ("c2_fade", {"stage": 9.5}),
    ("c3_b8", {"config": os.path.join(ROOT, "configs", "ffhq_stylegan_occlusion.yml"), "batch": 8}),
    ("c4", {"config": os.path.join(ROOT, "configs", "deepvoxels_shapenet_car.yml")}),
    ("c5_bf16", {"res256": True}),
    ("c5_fp8", {"res256": True, "fp8": True}),
]


def main():
    args = parse()
    world_env = os.environ.get("WORLD_SIZE")
    if world_env is None and (args.gpus or 1) > 1:
        return launch_ranks(args.gpus)
    if world_env is not None and args.gpus is not None and int(world_env) != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world_env}: refusing to report a mislabelled run",
              file=sys.stderr, flush=True)
        sys.exit(2)
    # stdout carries exactly ONE line (the JSON): everything else that writes to file descriptor 1 -- RCCL prints its
    # version banner there when a process group is created -- is sent to stderr
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    import torch
    from rgbd_gan_amd import kernels
    from rgbd_gan_amd.dist import Communicator

    world = int(world_env or "1")
    args.gpus = world
    local = 0 if (world == 1 or os.environ.get("RGBD_SHARE_DEVICE")) else int(os.environ.get("LOCAL_RANK", "0"))
    if local >= torch.cuda.device_count():
        print(f"bench.py: rank needs cuda:{local} but only {torch.cuda.device_count()} device(s) are visible",
              file=sys.stderr, flush=True)
        sys.exit(3)
    torch.cuda.set_device(local)
    comm = Communicator()
    device = torch.device("cuda", local)
    if comm.size != world:
        raise RuntimeError(f"process group reports {comm.size} ranks, WORLD_SIZE={world}")

    line = run_workload(args, comm, device)
    deepvoxels = line.pop("_deepvoxels")
    if comm.rank == 0 and comm.size == 1 and not args.no_cpu_baseline and not deepvoxels:
        line["cpu_baseline"] = cpu_baseline()
    if (comm.size == 1 and workload_key(args) == "default" and not args.no_other_configs and not args.arrangements
            and not args.mx8_standalone_quantiser):
        import copy
        line["other_configs"] = {}
        for name, over in OTHER_CONFIGS:
            if args.other_only and name not in args.other_only.split(","):
                continue
            a2 = copy.copy(args)
            for k, v in over.items():
                setattr(a2, k, v)
            a2.steps, a2.warmup, a2.no_cpu_baseline, a2.no_other_configs = args.other_steps, 12, True, True
            try:
                l2 = run_workload(a2, comm, device)
            except Exception as e:      # a broken side line must not take the headline number with it; it says so instead
                line["other_configs"][name] = {"error": f"{type(e).__name__}: {e}"}
                continue
            finally:
                kernels.MX8_EMIT = True
                from rgbd_gan_amd import functional as Fn
                Fn.set_conv_dtype("bf16")
            l2.pop("_deepvoxels")
            keep = ("metric", "value", "unit", "ms_per_step", "steps", "warmup", "dtype", "config", "roofline", "roofline_fp8",
                    "host_enqueue_ms_per_step", "host_enqueue_burst_ms")
            line["other_configs"][name] = {k: l2[k] for k in keep if k in l2}
            line["other_configs"][name]["command"] = "python bench.py " + " ".join(
                (f"--{k.replace('_', '-')}" if v is True else f"--{k.replace('_', '-')} {os.path.relpath(v, ROOT) if k == 'config' else v}")
                for k, v in over.items()) + f" --steps {a2.steps} --warmup {a2.warmup}"
        line["other_configs_note"] = ("short runs of the other BASELINE configurations by the same process, after the timed "
                                      "region of the headline workload; each with its own ms_per_step / dtype / roofline; "
                                      "never `value`")
    if comm.rank == 0:
        print(json.dumps(line), file=json_out, flush=True)
    comm.barrier()
    comm.close()


if __name__ == "__main__":
    main()
