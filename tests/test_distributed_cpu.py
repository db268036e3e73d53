"""World-size-2 (gloo, CPU) test of the data-parallel optimizer logic: ChainerMN semantics restated in
rgbd_gan_amd/optimizer.py -- first update() only broadcasts rank 0's parameters, later updates all-reduce the flat
gradient buffer and every rank applies the same clipped Adam step on the MEAN gradient.

The HIP Adam kernel cannot run on CPU, so this test swaps the kernel wrapper for a torch restatement (a test double
of the C ABI call, same arguments); everything above it (flat buffers, segments, collectives, ordering) is the
product code."""
import os
import socket

import numpy as np
import torch
import torch.multiprocessing as mp


def _cpu_adam(p, g, m, v, seg_begin, seg_alpha, beta1, beta2, eps, clip, grad_scale, step, workspace, norm_out=None):
    gs = g * grad_scale
    norm = float(torch.sqrt((gs.double() ** 2).sum()))
    rate = min(1.0, clip / norm) if norm > 0 else 1.0
    step += 1
    t = int(step.item())
    corr = np.sqrt(1 - beta2 ** t) / (1 - beta1 ** t)
    for i, a in enumerate(seg_alpha):
        sl = slice(int(seg_begin[i]), int(seg_begin[i + 1]))
        gr = gs[sl] * rate
        m[sl] += (1 - beta1) * (gr - m[sl])
        v[sl] += (1 - beta2) * (gr * gr - v[sl])
        p[sl] -= a * corr * m[sl] / (torch.sqrt(v[sl]) + eps)
    if norm_out is not None:
        norm_out.fill_(norm)


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from rgbd_gan_amd import kernels
    from rgbd_gan_amd.dist import Communicator
    from rgbd_gan_amd.optimizer import FlatAdam
    from rgbd_gan_amd.params import ParamStore
    kernels.adam_clip_multi = _cpu_adam
    comm = Communicator(backend="gloo")
    store = ParamStore([("a/W", (5, 3), "normal"), ("a/b", (3,), "zeros")], "cpu", seed=100 + rank)  # ranks differ
    opt = FlatAdam(store, alpha=1e-2, comm=comm)
    opt.set_alpha("a/b", 1e-4)
    start = store.flat.clone()
    # 1st update: broadcast only
    store.grad.fill_(1.0)
    opt.update()
    after_bcast = store.flat.clone()
    # 2nd update: rank-dependent gradients -> mean
    g = torch.arange(store.numel, dtype=torch.float32) * (rank + 1)
    store.grad.copy_(g)
    opt.start_allreduce()
    opt.update()
    q.put((rank, start.numpy(), after_bcast.numpy(), store.flat.clone().numpy(), int(opt.step.item()),
           float(opt.grad_norm)))
    comm.close()


def test_two_rank_allreduce_and_first_update_broadcast():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, start0, b0, end0, t0, n0), (r1, start1, b1, end1, t1, n1) = res
    assert not np.allclose(start0, start1)                  # different initial weights
    np.testing.assert_array_equal(b0, start0)               # first update(): broadcast of rank 0, no step ...
    np.testing.assert_array_equal(b1, start0)
    assert t0 == t1 == 1                                    # ... and Adam's t counted only the real step
    np.testing.assert_array_equal(end0, end1)               # identical step on every rank
    # reference: single process on the mean gradient
    n = len(start0)
    gmean = torch.arange(n, dtype=torch.float32) * 1.5
    p = torch.from_numpy(start0.copy())
    m, v, stp = torch.zeros(n), torch.zeros(n), torch.zeros(1, dtype=torch.int32)
    _cpu_adam(p, gmean, m, v, [0, 16, n], [1e-2, 1e-4], 0.0, 0.999, 1e-8, 5.0, 1.0, stp, None)
    np.testing.assert_allclose(end0, p.numpy(), rtol=1e-6, atol=1e-7)
    assert abs(n0 - float(torch.sqrt((gmean.double() ** 2).sum()))) < 1e-3


def _iter_worker(rank, q):
    from rgbd_gan_amd.training import DeviceImageIterator
    images = np.arange(64, dtype="uint8").reshape(64, 1, 1, 1).repeat(3, axis=1)       # pixel value = sample index
    it = DeviceImageIterator(images, 8, "cpu")                                          # no seed given, as train_rgbd.py
    first = ((it.next()[:, 0, 0, 0] + 1) * 127.5).round().to(torch.int64).tolist()
    q.put((rank, first, it.seed))


def test_ranks_draw_different_real_batches():
    """train_rgbd.py:306-310 of the reference: no scatter_dataset, every process shuffles the whole data set with its
    own RNG -- so the ranks of a data-parallel job must not see the same real batch (a fresh torch.Generator has a
    fixed default seed; DeviceImageIterator seeds from the OS when no seed is given)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_iter_worker, args=(r, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, b0, s0), (_, b1, s1) = res
    assert s0 != s1 and b0 != b1
    assert len(set(b0)) == 8 and all(0 <= i < 64 for i in b0)
    # explicit seeds: reproducible, and one epoch is a permutation
    from rgbd_gan_amd.training import DeviceImageIterator
    images = np.arange(64, dtype="uint8").reshape(64, 1, 1, 1).repeat(3, axis=1)
    a, b = DeviceImageIterator(images, 8, "cpu", seed=5), DeviceImageIterator(images, 8, "cpu", seed=5)
    seen = []
    for _ in range(8):
        xa, xb = a.next(), b.next()
        assert torch.equal(xa, xb)
        seen += ((xa[:, 0, 0, 0] + 1) * 127.5).round().to(torch.int64).tolist()
    assert sorted(seen) == list(range(64)) and a.epoch == 1


def test_bench_refuses_a_mislabelled_world():
    """`--gpus 8` under WORLD_SIZE=1 must not print an n_gpus=1 line (it exits before touching torch or a GPU)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode != 0 and "WORLD_SIZE" in r.stderr and '"metric"' not in r.stdout


def test_bench_parent_fails_when_a_rank_fails():
    """`python bench.py --gpus 2` starts two ranks of itself; here (no GPU) each rank exits non-zero, and so must the
    parent, without printing a JSON line."""
    import subprocess
    import sys
    import torch as _t
    if _t.cuda.is_available():
        import pytest
        pytest.skip("CPU-only property")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode != 0 and '"metric"' not in r.stdout
    assert "exited with code" in r.stderr


def test_every_rank_resumes_its_own_sample_sequence(tmp_path):
    """Resume under data parallelism (round-3 advisor finding): ranks shuffle the whole data set with their own seeds, so a
    resumed rank must continue ITS sequence -- rank 0's state loaded everywhere makes all ranks draw the same reals for the
    rest of the run.  Each rank writes / reads its own iterator file; with only the master's snapshot present (files written
    before per-rank states existed) rank 0 takes that copy and the other ranks keep a fresh rank-seeded shuffle."""
    from rgbd_gan_amd.training import DeviceImageIterator, load_iterator_state, save_iterator_state
    images = np.random.RandomState(0).randint(0, 256, (40, 3, 8, 8)).astype("uint8")
    make = lambda rank: DeviceImageIterator(images, 8, "cpu", seed=11 + rank)
    its = [make(0), make(1)]
    for it in its:
        for _ in range(7):                                  # past an epoch boundary: a second permutation has been drawn
            it.next_indices()
    for r, it in enumerate(its):
        save_iterator_state(str(tmp_path), 1000, r, it)
    want = [[it.next_indices().tolist() for _ in range(6)] for it in its]
    assert want[0] != want[1]
    resumed = [make(0), make(1)]
    assert [load_iterator_state(str(tmp_path), 1000, r, it) for r, it in enumerate(resumed)] == ["own", "own"]
    got = [[it.next_indices().tolist() for _ in range(6)] for it in resumed]
    assert got == want                                      # every rank continues its own sequence ...
    assert got[0] != got[1]                                 # ... and they are different sequences
    # only the master's copy: rank 0 resumes from it, rank 1 must NOT adopt it
    its = [make(0), make(1)]
    for it in its:
        for _ in range(3):
            it.next_indices()
    master = {f"iterator/{k}": v for k, v in its[0].state_dict().items()}
    want0 = [its[0].next_indices().tolist() for _ in range(4)]
    fresh = [make(0), make(1)]
    assert load_iterator_state(str(tmp_path), 2000, 0, fresh[0], master) == "master"
    assert load_iterator_state(str(tmp_path), 2000, 1, fresh[1], master) == "fresh"
    assert [fresh[0].next_indices().tolist() for _ in range(4)] == want0
    assert [fresh[1].next_indices().tolist() for _ in range(4)] != want0
