"""Data-parallel communicator: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI on ROCm;
"gloo" on CPU for tests).  Replaces chainermn.create_communicator('pure_nccl') (train_rgbd.py:103-121).

The only collectives the path needs: one parameter broadcast per optimizer at start and one
gradient all-reduce(sum) per optimizer per step on a flat fp32 buffer.  No dataset scatter, no barrier.
"""
import os

import torch
import torch.distributed as dist


class Communicator:
    def __init__(self, backend=None):
        self.size = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.intra_rank = int(os.environ.get("LOCAL_RANK", "0"))
        if os.environ.get("RGBD_SHARE_DEVICE"):          # every rank on cuda:0 (single-GPU test boxes)
            self.intra_rank = 0
        self.owns_group = False
        # RGBD_DEBUG_FORCE_COLLECTIVES=1: run the broadcast / all-reduce calls even with a single rank (a one-rank RCCL
        # group), so the data-parallel code path -- collectives next to graph replays and the two compute streams --
        # can be exercised on a one-GPU box
        self.force = bool(os.environ.get("RGBD_DEBUG_FORCE_COLLECTIVES"))
        if (self.size > 1 or self.force) and not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29500")
            if backend is None:
                # RGBD_DIST_BACKEND=gloo lets several ranks share one GPU (tests of the multi-process path on a
                # single-GPU box; RCCL refuses duplicate devices)
                backend = os.environ.get("RGBD_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
            if backend == "nccl":
                torch.cuda.set_device(self.intra_rank)
            dist.init_process_group(backend=backend, rank=self.rank, world_size=self.size)
            self.owns_group = True

    def allreduce_async(self, flat):
        return dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=True)

    def wait(self, work):
        if work is not None:
            work.wait()

    def broadcast(self, flat, src=0):
        dist.broadcast(flat, src=src)

    @property
    def active(self):
        """True when collectives have to run (several ranks, or forced for testing)."""
        return self.size > 1 or self.force

    def barrier(self):
        if self.size > 1:
            dist.barrier()

    def max_over_ranks(self, value):
        if self.size == 1:
            return value
        dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
        t = torch.tensor([value], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def close(self):
        if self.owns_group and dist.is_initialized():
            dist.destroy_process_group()
