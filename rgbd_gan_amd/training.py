"""Model / optimizer / data set-up shared by train_rgbd.py and bench.py (train_rgbd.py:172-258,306-361 of the reference)."""
import glob
import os

import numpy as np
import torch

from .net import DCGANGenerator, Discriminator, StyleGANGenerator
from .optimizer import FlatAdam
from .updater import CameraParamPrior, RGBDUpdater, RGBUpdater


def setup_generator(config, device, seed=0):
    """train_rgbd.py:220-246."""
    rgbd = False if config.rgb else True
    arch = config.generator_architecture
    if arch == "stylegan":
        return StyleGANGenerator(config.ch, enable_blur=config.enable_blur, rgbd=rgbd,
                                 rotate_conv_input=config.rotate_conv_input, use_encoder=config.bigan,
                                 use_occupancy_net=config.use_occupancy_net_loss, initial_depth=config.initial_depth,
                                 device=device, seed=seed, max_resolution=config.max_resolution or 128)
    if arch == "dcgan":
        # NB the reference passes config.ch as in_ch and leaves ch at its default 512 (train_rgbd.py:230)
        return DCGANGenerator(config.ch, enable_blur=config.enable_blur, rgbd=rgbd, use_encoder=config.bigan,
                              use_occupancy_net=config.use_occupancy_net_loss, initial_depth=config.initial_depth,
                              device=device, seed=seed)
    if arch == "deepvoxels":
        from .deepvoxels_generator import Generator
        if config.rendernet_projection:
            occlusion_type = "rendernet"
        elif config.occlusion_type:
            occlusion_type = config.occlusion_type
        else:
            occlusion_type = "deepvoxels"
        return Generator(config.ch, occlusion_type=occlusion_type, background_generator=config.background_generator,
                         config=config, device=device, seed=seed)
    raise AssertionError(f"{arch} is not supported")


def setup_discriminator(config, device):
    """train_rgbd.py:249-258."""
    if config.bigan:
        raise AssertionError("bigan is not supported")
    # max_resolution: optional YAML key of this engine (default 128 = the reference's six blocks); 256 builds the block the
    # reference keeps commented out (net.py:181,192,437-452) -- BASELINE configuration 5
    return Discriminator(ch=config.ch, enable_blur=config.enable_blur, sn=config.sn, res=config.res_dis, device=device,
                         max_resolution=config.max_resolution or 128)


def make_optimizers(config, generator, discriminator, comm=None):
    """train_rgbd.py:314-343: Adam(beta1, beta2) + GradientClipping(5); map runs at alpha_g/100, and so do gen.l1/l2."""
    b1, b2 = config.adam_beta1, config.adam_beta2
    if config.generator_architecture == "stylegan":
        opt = {"map": FlatAdam(generator.mapping.store, config.adam_alpha_g / 100, b1, b2, comm=comm),
               "gen": FlatAdam(generator.gen.store, config.adam_alpha_g, b1, b2, comm=comm),
               "dis": FlatAdam(discriminator.store, config.adam_alpha_d, b1, b2, comm=comm)}
        if not config.rgb:
            for n in ("l1/c/W", "l2/c/W", "l1/c/b", "l2/c/b"):
                opt["gen"].set_alpha(n, config.adam_alpha_g / 100)
        return opt
    if config.generator_architecture == "deepvoxels":
        return {"map": FlatAdam(generator.mapping.store, config.adam_alpha_g / 100, b1, b2, comm=comm),
                "gen": FlatAdam(generator.store, config.adam_alpha_g, b1, b2, comm=comm),
                "dis": FlatAdam(discriminator.store, config.adam_alpha_d, b1, b2, comm=comm)}
    return {"gen": FlatAdam(generator.store, config.adam_alpha_g, b1, b2, comm=comm),
            "dis": FlatAdam(discriminator.store, config.adam_alpha_d, b1, b2, comm=comm)}


def make_dataset(dataset_path, image_path):
    """train_rgbd.py:172-184: `images.npy` cache (uint8, N x 3 x H x W), built from the glob on first use."""
    cache = f"{dataset_path}/images.npy"
    if os.path.exists(cache):
        return np.load(cache)
    from PIL import Image
    imgs = [np.array(Image.open(p)).transpose(2, 0, 1) for p in sorted(glob.glob(image_path))]
    imgs = np.array(imgs, dtype="uint8")
    np.save(cache, imgs)
    return imgs


class DeviceImageIterator:
    """SerialIterator + TransformDataset(x / 127.5 - 1) of train_rgbd.py:308-310 with the uint8 data set resident
    in HBM (an FFHQ-sized 70k x 3 x 128 x 128 set is 3.4 GB of the 288 GB): a batch is one gather + one normalise on
    the device instead of a Python list and a host-to-device copy per step (updater.py:259-268)."""

    def __init__(self, images_u8, batch_size, device, shuffle=True, seed=None):
        self.data = torch.as_tensor(images_u8).to(device)
        self.batch_size, self.shuffle = batch_size, shuffle
        # chainer's SerialIterator shuffles with the process's own entropy-seeded NumPy RNG, so the ranks of a
        # data-parallel job draw DIFFERENT real batches (train_rgbd.py:306-310: no scatter_dataset).  A fresh
        # torch.Generator has a fixed default seed: without one given, seed from the OS so ranks and runs differ.
        self.seed = int(seed) if seed is not None else int.from_bytes(os.urandom(8), "little") >> 1
        self.gen = torch.Generator(device="cpu")
        self.gen.manual_seed(self.seed)
        self.epoch, self._pos = 0, 0
        self._order = self._new_order()

    def _new_order(self):
        """One permutation per epoch, drawn on the host and uploaded ONCE: slicing a batch's indices out of it is then
        a device view, and next() enqueues no host-to-device copy (a pageable upload per step made the launch thread
        wait for the stream to drain every iteration)."""
        n = self.data.shape[0]
        order = torch.randperm(n, generator=self.gen) if self.shuffle else torch.arange(n)
        return order.to(self.data.device)

    def next(self):
        return self.data[self.next_indices()].to(torch.float32) / 127.5 - 1

    def next_indices(self):
        """Indices (device int64) of the next batch: the updater gathers, normalises and down-sizes them in one kernel
        (rgbd_real_batch_u8) instead of materialising the fp32 batch here."""
        n = self.data.shape[0]
        idx = self._order[self._pos:self._pos + self.batch_size]
        self._pos += self.batch_size
        if self._pos >= n:                                    # wrap like chainer's SerialIterator(repeat=True)
            self.epoch += 1
            rest = self._pos - n
            self._order = self._new_order()
            if rest > 0:
                idx = torch.cat([idx, self._order[:rest]])
            self._pos = rest
        return idx

    __next__ = next

    def state_dict(self):
        """What chainer's SerialIterator.serialize keeps (current_position, epoch, the epoch's order), plus the state of
        the generator that draws the next permutations: a resumed run continues the same sample sequence."""
        return {"pos": np.int64(self._pos), "epoch": np.int64(self.epoch), "order": self._order.cpu().numpy(),
                "rng_state": self.gen.get_state().numpy(), "seed": np.int64(self.seed)}

    def load_state_dict(self, sd):
        order = torch.as_tensor(np.asarray(sd["order"]))
        if order.numel() != self.data.shape[0]:
            raise ValueError(f"iterator snapshot holds an order over {order.numel()} samples, the data set has "
                             f"{self.data.shape[0]}")
        self._pos, self.epoch = int(sd["pos"]), int(sd["epoch"])
        self._order = order.to(self.data.device)
        if sd.get("rng_state") is not None:        # (a trainer snapshot written by the reference has no generator state: the
            self.gen.set_state(torch.as_tensor(np.asarray(sd["rng_state"]), dtype=torch.uint8))   # next epochs' permutations
            if sd.get("seed") is not None:                                                         # come from this run's seed)
                self.seed = int(sd["seed"])


def iterator_state_path(directory, iteration, rank):
    return os.path.join(directory, f"iterator_rank{rank}_iter_{iteration}.npz")


def save_iterator_state(directory, iteration, rank, iterator):
    """EVERY rank keeps its own iterator state next to the master's snapshot: ranks shuffle the whole data set with their own
    seeds (train_rgbd.py of the reference: no scatter_dataset, one NumPy RNG per process), so rank 0's position / order /
    generator state is not theirs."""
    path = iterator_state_path(directory, iteration, rank)
    tmp = path + ".tmp.npz"
    try:
        # every rank makes the directory for itself (a node-local `out` exists on the master's node only) and replaces the
        # file in one step: a crash in the middle of the write must not leave a truncated file that a resume would prefer
        os.makedirs(directory, exist_ok=True)
        np.savez(tmp, **iterator.state_dict())
        os.replace(tmp, path)
    except OSError as e:
        # a rank that cannot write its file must not die here and leave the other ranks in their next collective: a missing
        # file already degrades to "fresh" on resume (load_iterator_state)
        if rank == 0:
            raise
        print(f"[rank {rank}] could not write {path}: {e}; a resume will start this rank's iterator fresh", flush=True)
        try:
            os.remove(tmp)
        except OSError:
            pass


def load_iterator_state(directory, iteration, rank, iterator, master_snapshot=None):
    """Resume `iterator` as rank `rank` of a run snapshotted at `iteration`.  Its own file if there is one; else, on rank 0
    only, the copy inside the master snapshot (snapshots written before per-rank files existed); any other rank keeps the
    fresh, rank-seeded state it was built with -- loading rank 0's would make every rank draw the same reals for the rest of
    the run.  -> "own" | "master" | "fresh"."""
    path = iterator_state_path(directory, iteration, rank)
    if os.path.exists(path):
        try:
            with np.load(path) as f:
                sd = {k: f[k] for k in f.files}
            iterator.load_state_dict(sd)
            return "own"
        except (OSError, ValueError, KeyError, EOFError) as e:      # zipfile.BadZipFile is an OSError-free Exception: see below
            print(f"[rank {rank}] unreadable iterator state {path}: {e}", flush=True)
        except Exception as e:                                      # a truncated archive (BadZipFile) and the like
            print(f"[rank {rank}] unreadable iterator state {path}: {type(e).__name__}: {e}", flush=True)
    if rank == 0 and master_snapshot is not None and "iterator/pos" in master_snapshot:
        iterator.load_state_dict({k[len("iterator/"):]: master_snapshot[k] for k in master_snapshot
                                  if k.startswith("iterator/")})
        return "master"
    return "fresh"


def build_training(config, device, comm=None, iterator=None, updater_class=None, **updater_kwargs):
    # `conv_dtype: mxfp8` (BASELINE configuration 5; no such key in the reference's YAMLs, which compute in fp32): the 3x3
    # convolutions' fprop / dgrad on block-scaled fp8 operands.  Process-wide, like chainer.global_config.dtype.
    from . import functional as Fn
    Fn.set_conv_dtype(config.conv_dtype or "bf16")
    generator = setup_generator(config, device)
    discriminator = setup_discriminator(config, device)
    Fn.apply_mx8_coverage(generator, discriminator, config.mxfp8_coverage)
    optimizer = make_optimizers(config, generator, discriminator, comm)
    # the persistent weight images (and, with conv_dtype mxfp8, their fp8 twins and descriptor tables: allocations and a
    # host-to-device copy) exist before the first step -- a first build inside a HIP-graph capture (graph_warmup = 0, or the
    # conv dtype switched late) would be an illegal synchronous copy there
    for net in (getattr(generator, "gen", generator), discriminator):
        group = getattr(net, "pack_group", None)
        if group is not None and torch.device(device).type == "cuda":
            group.repack()
    models = [generator, discriminator]
    if config.keep_smoothed_gen and (comm is None or comm.rank == 0):       # train_rgbd.py:124-125,288-295
        models.append(setup_generator(config, device, seed=1000))    # its own random init, like the reference
    if config.generator_architecture == "deepvoxels":             # train_rgbd.py:355-356
        from .updater_deepvoxels import DeepVoxelsUpdater as Updater
        for k in ("fixed_stage", "concurrent_phases", "dp_split_body"):
            updater_kwargs.pop(k, None)
    elif config.rgb:                                               # train_rgbd.py:357-358
        Updater = RGBUpdater
    else:
        Updater = RGBDUpdater
    if updater_class is not None:                                  # tests: a subclass with one phase overridden
        Updater = updater_class
    updater = Updater(models=models, config=config, optimizer=optimizer, iterator=iterator,
                      lambda_gp=config.lambda_gp, smoothing=config.smoothing,
                      total_gpu=comm.size if comm is not None else 1, prior=CameraParamPrior(config),
                      **updater_kwargs)
    return generator, discriminator, optimizer, updater
