#!/usr/bin/env python3
"""Training entry point with the reference's command line (train_rgbd.py:261-470 of nogu-atsu/RGBD-GAN):

    python train_rgbd.py [-g GPU] --config_path configs/ffhq_stylegan_occlusion.yml
    python -m torch.distributed.run --nproc-per-node 8 train_rgbd.py --config_path ...   (config: use_mpi: True)

`--config` is accepted as an alias (the reference's README spells it that way).  One process per GPU; under
torch.distributed the gradients of each optimizer are averaged with one RCCL all-reduce per step.
Outputs: {out}/Generator_{iter}.npz, Discriminator_{iter}.npz, snapshot_iter_{iter}.npz, *_latest.npz, {out}/log.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from rgbd_gan_amd.dist import Communicator
from rgbd_gan_amd.training import (DeviceImageIterator, build_training, load_iterator_state, make_dataset,
                                   save_iterator_state)
from rgbd_gan_amd.common.utils import trainer_snapshot
from rgbd_gan_amd.utils import yaml_utils


def save_npz(path, link):
    np.savez(path, **link.state_dict())


def main():
    parser = argparse.ArgumentParser()
    parser.add_argument("--gpu", "-g", type=int, default=0)
    parser.add_argument("--config_path", "--config", type=str, default="configs/ffhq_stylegan_occlusion.yml")
    args = parser.parse_args()
    config = yaml_utils.load(args.config_path)
    config.gpu = args.gpu
    print(config.stage_interval)

    comm = Communicator() if (config.use_mpi or int(os.environ.get("WORLD_SIZE", "1")) > 1) else None
    device_index = comm.intra_rank if comm is not None and comm.size > 1 else config.gpu
    torch.cuda.set_device(device_index)
    device = torch.device("cuda", device_index)
    is_master = comm is None or comm.rank == 0

    images = make_dataset(config.dataset_path, config.image_path)       # uint8 (N,3,128,128), images.npy cache
    # every rank shuffles the whole data set with its own seed (the reference: no scatter_dataset, per-process RNG)
    iterator = DeviceImageIterator(images, config.batchsize, device,
                                   seed=None if config.seed is None else int(config.seed) + (comm.rank if comm else 0))
    if config.nvprof or config.enable_cuda_profiling:
        config.iteration = 10
    generator, discriminator, optimizer, updater = build_training(
        config, device, comm if comm is not None and comm.size > 1 else None, iterator=iterator,
        nan_check_interval=config.display_interval or 100)
    if config.nvprof or config.enable_cuda_profiling:
        # the reference wraps trainer.run() in cupy.cuda.profile() (train_rgbd.py:462-464); here the step's phases carry
        # roctx ranges for rocprofv3 (--marker-trace), and the sub-phases run eagerly so each is its own range
        updater.profile_ranges = True
    models = [("Generator", generator), ("Discriminator", discriminator)]
    if updater.smoothed_gen is not None:                               # train_rgbd.py:293-295
        models.append(("SmoothedGenerator", updater.smoothed_gen))
    if config.generator_architecture == "deepvoxels":                  # train_rgbd.py:374-377,456-459
        models.append(("Map", generator.mapping))

    out = config.out
    if is_master:
        os.makedirs(out, exist_ok=True)
    # resume (train_rgbd.py:405-459): explicit iteration or the newest complete set in auto_resume_dir
    log_resumed, elapsed_resumed = [], 0.0
    resume = config.get_model_from_interation or ""
    explicit = bool(resume)            # an explicit iteration is always read from config.out (train_rgbd.py:406-415)
    if not resume and config.auto_resume:
        d = config.auto_resume_dir or out
        cands = []
        for f in os.listdir(d) if os.path.isdir(d) else []:
            if f.startswith("Generator_") and f.endswith(".npz") and f[10:-4].isdigit():
                it = f[10:-4]
                if all(os.path.exists(f"{d}/{n}_{it}.npz") for n, _ in models) and \
                        os.path.exists(f"{d}/snapshot_iter_{it}.npz"):
                    cands.append(int(it))
        resume = str(max(cands)) if cands else ""
    if resume:
        d = out if explicit else (config.auto_resume_dir or out)
        print(f"Resume from {resume}")
        for name, m in models:
            m.load_state_dict(dict(np.load(f"{d}/{name}_{resume}.npz")), strict=False)
        # the trainer snapshot, in the reference's key layout (rgbd_gan_amd/common/utils/trainer_snapshot.py): iteration,
        # per-parameter Adam moments, the iterator (position, epoch, order) and LogReport's entries; also reads the flat
        # layout this engine wrote before round 4.  Missing keys are skipped (the reference loads with strict=False)
        with np.load(f"{d}/snapshot_iter_{resume}.npz", allow_pickle=True) as snap_file:
            snap = trainer_snapshot.unpack(snap_file, optimizer)
        from rgbd_gan_amd import functional as Fn
        Fn.bump_weight_epoch()          # (unpack also restores the optimizers' target weights: packed images are stale)
        updater.iteration = snap["iteration"]
        if snap["log"] is not None:
            log_resumed, elapsed_resumed = snap["log"], snap["elapsed_time"]
        # -- per rank: every rank shuffles with its own seed, so each resumes ITS position / order / generator state
        how = load_iterator_state(d, resume, comm.rank if comm is not None else 0, iterator,
                                  {f"iterator/{k}": v for k, v in (snap["iterator"] or {}).items()})
        if how == "fresh" and comm is not None and comm.rank > 0:
            print(f"rank {comm.rank}: no iterator state of its own in {d}; continuing with a fresh rank-seeded shuffle")

    previews = []
    if is_master and config.evaluation_sample_interval:                # train_rgbd.py:386-396
        from rgbd_gan_amd.common.utils.save_images import PreviewSampler
        previews.append(PreviewSampler(generator, out, config, rows=8, cols=8))
        if updater.smoothed_gen is not None:
            previews.append(PreviewSampler(updater.smoothed_gen, out, config, rows=8, cols=8, subdir="preview_smoothed"))

    # The side stream's compute-unit budget is MEASURED on this device at every image size the run meets, inside the ordinary
    # iterations of the loop below (rgbd_gan_amd.updater.SideBudgetTuner: ~120 steps per size, triggers fire as always; one GPU,
    # two streams, graphs; RGBD_TUNE_SIDE_BUDGET=0 keeps the rule of thumb).  DESIGN.md section 3.
    if comm is None or comm.size == 1:
        updater.tune_side_budget = os.environ.get("RGBD_TUNE_SIDE_BUDGET") not in ("", "0") and hasattr(updater, "finish_tuning") \
            and config.generator_architecture != "deepvoxels"
    reported_tuning = None
    log, t0 = log_resumed, time.time() - elapsed_resumed       # a resumed run appends to the log it left (LogReport)
    while updater.iteration < config.iteration:
        updater.update()
        it = updater.iteration
        tuned = getattr(updater, "side_budget_tuning", None)
        if is_master and tuned is not None and tuned is not reported_tuning:
            reported_tuning = tuned
            print(f"side stream budget at iteration {it}: {tuned}")
        if is_master and it % (config.display_interval or 100) == 0:
            obs = {k: (float(v) if torch.is_tensor(v) else v) for k, v in updater.observation.items()}
            entry = {"iteration": it, "elapsed_time": time.time() - t0, **obs}
            log.append(entry)
            print(json.dumps(entry))
            with open(f"{out}/log", "w") as f:
                json.dump(log, f, indent=1)
        if previews and it % config.evaluation_sample_interval == 0:
            for pv in previews:
                pv(updater.stage, it)
        if it % (config.snapshot_interval or 10000) == 0:
            updater.assert_finite_so_far()      # the newest step's non-finite flag, before anything of it is written out
        if it % (config.snapshot_interval or 10000) == 0 and comm is not None and comm.size > 1:
            save_iterator_state(out, it, comm.rank, iterator)
        if is_master and it % (config.snapshot_interval or 10000) == 0:
            for name, m in models:
                save_npz(f"{out}/{name}_{it}.npz", m)
            snap = trainer_snapshot.pack(it, optimizer, iterator.state_dict(), log, time.time() - t0,
                                         config.display_interval or 100)
            np.savez(f"{out}/snapshot_iter_{it}.npz", **snap)
    updater.assert_finite_so_far()
    if is_master:
        for name, m in models:
            save_npz(f"{out}/{name}_latest.npz", m)
    if comm is not None:
        comm.close()


if __name__ == "__main__":
    main()
