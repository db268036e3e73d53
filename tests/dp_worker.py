"""Helper process of tests/test_multirank_gpu.py (not a test module): runs RGBDUpdater.update_core on FIXED inputs and
dumps what the optimizers saw, so that arrangements of the same step can be compared tensor by tensor.

    python tests/dp_worker.py OUT.npz --calls K [--batch B] [--eager] [--sequential] [--logit-shift S]

Single process: the whole batch.  Under RANK / WORLD_SIZE (test: two ranks sharing cuda:0 over gloo): rank r takes the
view pairs r, r + W, ... of the same batch, i.e. the N-rank job and the 1-rank job see the same samples.

--virtual-rank R --virtual-world W [--peer-grads F.npz ...] [--dump-reduce-inputs F.npz]: ONE process plays rank R of a
W-rank job through a loop-back communicator (same half-batch, same kernels, same graphs as the real rank): its all-reduce
adds the gradients the peers dumped (--dump-reduce-inputs of their own run) instead of talking to them.  The real 2-rank
job has to reproduce this to fp32 rounding: only the transport differs.

Protocol (makes a multi-step comparison well-posed although a GAN step is chaotic): after every call except the last
the master weights are put back to their initial values, so every call differentiates at the SAME weights with the
SAME inputs while Adam's moments and step counter advance as usual; the last call is then a replayed-graph step (calls
1-2 run eagerly, call 3 is captured) whose gradients, moments and weight update are dumped.  A data-parallel job's
first call is ChainerMN's broadcast-only update (train_rgbd.py:154-156), so it makes one call more.
"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

CFG = dict(generator_architecture="stylegan", ch=256, stage_interval="0,0,0,0,0,0,0,100000,150000,160000,180000,300000",
           max_stage=11, start_rotation=2000, start_occlusion_aware=2000, lambda_depth=10, depth_min=1.0,
           x_rotate=0.3054, y_rotate=1.0472, z_rotate=0, x_translate=0, y_translate=0, z_translate=0, bigan=False,
           adam_alpha_g=0.001, adam_alpha_d=0.003, adam_beta1=0.0, adam_beta2=0.999, lambda_gp=1.0, smoothing=0.999,
           res_dis=True, sn=False, enable_blur=False)


def fixed_inputs(B, seed=7):
    from rgbd_gan_amd.updater import CameraParamPrior
    from rgbd_gan_amd.utils.yaml_utils import Config
    rng = np.random.RandomState(seed)
    zh = rng.randn(B // 2, 512, 1, 1).astype("float32")
    zh = zh / np.sqrt((zh * zh).sum(axis=1, keepdims=True) / 256 + 1e-8)
    z = np.concatenate([zh, zh])
    np.random.seed(seed + 1)
    thetas = CameraParamPrior(Config(CFG)).sample(B)
    x_real = rng.randint(0, 256, (B, 3, 128, 128)).astype("float32") / 127.5 - 1
    return z, thetas, x_real


class LoopbackComm:
    """A W-rank communicator played by one process: all-reduce(sum) adds the peers' recorded buffers (the three flat
    gradient buffers are told apart by their sizes)."""

    def __init__(self, rank, size, peer_files, dump_path):
        self.rank, self.size, self.intra_rank, self.active = rank, size, 0, True
        self.peers = [np.load(f) for f in peer_files]
        self.dump_path, self.seen, self.names = dump_path, {}, {}

    def allreduce_async(self, flat):
        name = self.names[flat.numel()]
        torch.cuda.current_stream().synchronize()
        self.seen[name] = flat.detach().cpu().numpy().copy()          # what this rank contributes (last call wins)
        for peer in self.peers:
            assert peer[name].shape == tuple(flat.shape), name
            flat.add_(torch.from_numpy(peer[name]).to(flat.device))
        return "done"                                                 # a non-None work handle (FlatAdam._pending)

    def wait(self, work):
        pass

    def broadcast(self, flat, src=0):
        pass                                                          # every virtual rank builds rank 0's weights

    def barrier(self):
        pass

    def close(self):
        if self.dump_path:
            np.savez(self.dump_path, **self.seen)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("out")
    ap.add_argument("--calls", type=int, default=4)
    ap.add_argument("--batch", type=int, default=8, help="GLOBAL batch")
    ap.add_argument("--stage", type=float, default=10.0)
    ap.add_argument("--eager", action="store_true")
    ap.add_argument("--sequential", action="store_true")
    ap.add_argument("--concurrent", action="store_true", help="two-stream phase overlap (the default on even stages)")
    ap.add_argument("--no-dp-split", action="store_true", help="data parallel, one stream: one body graph, then all three "
                    "all-reduces (default: body split where G's gradients are final)")
    ap.add_argument("--direct-seed", action="store_true", help="back-propagate the generator's OWN seed through D(x_fake) "
                    "instead of rescaling the discriminator-seeded pass (reference for the seed-ratio chain)")
    ap.add_argument("--virtual-rank", type=int, default=None)
    ap.add_argument("--virtual-world", type=int, default=2)
    ap.add_argument("--peer-grads", nargs="*", default=[])
    ap.add_argument("--dump-reduce-inputs", default=None)
    ap.add_argument("--sync-restore", action="store_true", help="host-synchronise after putting the weights back")
    ap.add_argument("--logit-shift", type=float, default=0.0,
                    help="added to the discriminator's output bias: y_fake ~ shift (seed-ratio chain at the clamp)")
    args = ap.parse_args()

    from rgbd_gan_amd import functional as Fn
    from rgbd_gan_amd.dist import Communicator
    from rgbd_gan_amd.training import build_training
    from rgbd_gan_amd.utils.yaml_utils import Config
    from rgbd_gan_amd.updater import RGBDUpdater
    if args.virtual_rank is not None:
        comm = LoopbackComm(args.virtual_rank, args.virtual_world, args.peer_grads, args.dump_reduce_inputs)
    else:
        comm = Communicator()
    torch.cuda.set_device(0 if os.environ.get("RGBD_SHARE_DEVICE") else comm.intra_rank)
    device = torch.device("cuda", torch.cuda.current_device())
    W, r = comm.size, comm.rank
    z, thetas, x_real = fixed_inputs(args.batch)
    half = args.batch // 2
    pairs = np.arange(r, half, W)                      # this rank's view pairs
    sel = np.concatenate([pairs, pairs + half])
    z, thetas, x_real = z[sel], thetas[sel], x_real[sel]

    kw = dict(fixed_stage=args.stage, use_graphs=not args.eager, graph_warmup=2, nan_check_interval=0)
    if args.sequential:
        kw["concurrent_phases"] = False
    if args.concurrent:
        kw["concurrent_phases"] = True
    if args.no_dp_split:
        kw["dp_split_body"] = False
    if args.direct_seed:
        class DirectSeed(RGBDUpdater):
            def _gen_seeds(self, st, x_d, y_fake, seed_g, seed_d, ratio):
                with Fn.weight_grads_frozen(self.dis):
                    gx, = torch.autograd.grad([y_fake], [x_d], [seed_g])
                return gx, None
        kw["updater_class"] = DirectSeed
    gen, dis, opt, upd = build_training(Config(CFG), device, comm if comm.active else None, iterator=None, **kw)
    torch.manual_seed(3)
    with torch.no_grad():                               # give the depth head some signal (as the step parity test does)
        for i in range(6):
            w = gen.gen.store[f"outs/{i}/c/W"]
            w[-1] = torch.randn(w[-1].shape, device=device) * 0.1
        if args.logit_shift:
            dis.store["blocks/0/l2/c/b"].add_(args.logit_shift)
    Fn.bump_weight_epoch()
    stores = {"map": gen.mapping.store, "gen": gen.gen.store, "dis": dis.store}
    if isinstance(comm, LoopbackComm):
        comm.names = {s.flat.numel(): k for k, s in stores.items()}
        assert len(comm.names) == 3
    w0 = {k: s.flat.clone() for k, s in stores.items()}
    upd.iteration = 200000
    calls = args.calls + (1 if comm.active else 0)      # data parallel: the first call only broadcasts
    xb, zb = torch.from_numpy(x_real), torch.from_numpy(z)
    for c in range(calls):
        upd.update_core(batch=xb, z_fake_data=zb, thetas=thetas)
        upd.iteration += 1
        if c < calls - 1:
            with torch.no_grad():
                for k, s in stores.items():
                    s.flat.copy_(w0[k])
            Fn.bump_weight_epoch()
            if args.sync_restore:
                torch.cuda.synchronize()
    torch.cuda.synchronize()
    out = {"n_graphs": len(upd._graphs), "world": W, "rank": r}
    scale = 1.0 / W                                     # the all-reduce sums; 1/N is folded into the Adam kernel
    for k, s in stores.items():
        out[f"{k}/grad"] = (s.grad * scale).cpu().numpy()
        out[f"{k}/delta"] = (s.flat - w0[k]).cpu().numpy()
        out[f"{k}/m"] = opt[k].m.cpu().numpy()
        out[f"{k}/v"] = opt[k].v.cpu().numpy()
        out[f"{k}/norm"] = float(opt[k].grad_norm)
        out[f"{k}/t"] = opt[k].t
    for k, s in stores.items():                        # parameter table: name -> (offset, numel) in the flat buffers
        out[f"{k}/names"] = np.array(s.names)
        out[f"{k}/offsets"] = np.array([s.offsets[n] for n in s.names], dtype=np.int64)
        out[f"{k}/sizes"] = np.array([int(np.prod(s.shapes[n])) for n in s.names], dtype=np.int64)
    for k, v in upd.observation.items():
        out["obs/" + k] = float(v)
    np.savez(args.out, **out)
    comm.barrier()
    comm.close()


if __name__ == "__main__":
    main()
