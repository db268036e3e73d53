"""`snapshot_iter_{iteration}.npz` in the key layout of the reference's trainer snapshot.

The reference saves `extensions.snapshot(filename='snapshot_iter_{.updater.iteration}.npz')` (train_rgbd.py:378-381) and
resumes with `chainer.serializers.load_npz(..., trainer, strict=False)` (train_rgbd.py:405-459).  `save_npz(trainer)` walks
Trainer -> StandardUpdater -> {iterators, optimizers, optimizer targets} and the extensions; the resulting flat keys are
(Chainer v7, restated from memory -- Chainer cannot be installed here, so this layout is UNVERIFIED against a file written by
the reference; everything is read back with strict=False semantics, i.e. absent keys are skipped, like the reference does):

    updater/iteration
    updater/iterator:main/{current_position, epoch, is_new_epoch, order, previous_epoch_detail}
    updater/optimizer:{map,gen,dis}/{t, epoch}
    updater/optimizer:{name}/{param path}/{t, m, v}        one update rule per parameter (Adam: first / second moment)
    updater/model:{name}/{param path}                      the optimizer's target link (generator.mapping / .gen / discriminator)
    extensions/LogReport/_log                              json of the report entries
    extensions/LogReport/_trigger/{_previous_iteration, _previous_epoch_detail}
    _snapshot_elapsed_time

Parameter paths are Chainer's (`l/0/c/W`, `blocks/3/c0/c/W`, ...), which are this engine's ParamStore names.  Keys of this
engine's own (the shuffle generator's state, which Chainer does not keep) sit under `rgbd_gan_amd/`.
"""
import json

import numpy as np


def pack(iteration, optimizers, iterator_state=None, log=None, elapsed_time=0.0, display_interval=100):
    """optimizers: {'map': FlatAdam, 'gen': ..., 'dis': ...} -> dict of arrays for np.savez."""
    out = {"updater/iteration": np.int64(iteration), "_snapshot_elapsed_time": np.float64(elapsed_time)}
    for name, opt in optimizers.items():
        store = opt.store
        sd = opt.state_dict()
        base = f"updater/optimizer:{name}/"
        out[base + "t"] = np.int64(sd["t"])
        out[base + "epoch"] = np.int64(0 if iterator_state is None else int(iterator_state["epoch"]))
        flat = store.flat.detach().cpu().numpy()
        for pname in store.names:
            off, n = store.offsets[pname], int(np.prod(store.shapes[pname]))
            shape = store.shapes[pname]
            out[f"{base}{pname}/t"] = np.int64(sd["t"])
            out[f"{base}{pname}/m"] = sd["m"][off:off + n].reshape(shape).copy()
            out[f"{base}{pname}/v"] = sd["v"][off:off + n].reshape(shape).copy()
            out[f"updater/model:{name}/{pname}"] = flat[off:off + n].reshape(shape).copy()
    if iterator_state is not None:
        base = "updater/iterator:main/"
        n = len(iterator_state["order"])
        pos, epoch = int(iterator_state["pos"]), int(iterator_state["epoch"])
        out[base + "current_position"] = np.int64(pos)
        out[base + "epoch"] = np.int64(epoch)
        out[base + "is_new_epoch"] = np.bool_(pos == 0 and epoch > 0)
        out[base + "order"] = np.asarray(iterator_state["order"], dtype=np.int64)
        out[base + "previous_epoch_detail"] = np.float64(epoch + pos / max(n, 1))
        out["rgbd_gan_amd/iterator:main/rng_state"] = np.asarray(iterator_state["rng_state"])
        out["rgbd_gan_amd/iterator:main/seed"] = np.int64(iterator_state["seed"])
    if log is not None:
        out["extensions/LogReport/_log"] = np.asarray(json.dumps(log))
        out["extensions/LogReport/_trigger/_previous_iteration"] = np.int64(iteration // display_interval * display_interval)
        out["extensions/LogReport/_trigger/_previous_epoch_detail"] = out.get("updater/iterator:main/previous_epoch_detail",
                                                                              np.float64(0.0))
    return out


def _normalised(f):
    """Chainer's DictionarySerializer strips slashes around every path component: index a file's keys by their '//'-free
    form once (a linear scan per miss was O(parameters x keys) on a reference-written file)."""
    out = {}
    for k in f:
        nk = k
        while "//" in nk:
            nk = nk.replace("//", "/")
        out.setdefault(nk.strip("/"), f[k])
    return out


def _get(f, key):
    return f.get(key)


def _as_flat_tensor(a, like):
    import torch
    return torch.as_tensor(np.asarray(a, dtype=np.float32).reshape(-1)).to(like.device)


def unpack(f, optimizers):
    """f: a mapping key -> array (np.load of a snapshot written by pack(), by the reference's trainer, or by this engine
    before round 4).  Loads what is there into `optimizers` (strict=False semantics) and returns
    {"iteration", "iterator" (state dict for DeviceImageIterator.load_state_dict, or None), "log" (list or None),
     "elapsed_time"}."""
    keys = set(f.keys()) if hasattr(f, "keys") else set(f.files)
    f = _normalised({k: f[k] for k in keys})
    if "iteration" in f and "updater/iteration" not in f:          # this engine's layout of rounds 1-3
        for k, o in optimizers.items():
            if f"{k}/t" in f:
                # flat moments are positional: they only mean something under the flat parameter layout they were written
                # with.  The DeepVoxels generator's layout changed in round 5 (style W's / b's regrouped, same total size), so a
                # rounds-1-3 snapshot of it would load silently misaligned Adam moments: refused, as is any size mismatch.
                names = getattr(o.store, "names", ())
                if any(n.startswith(("voxel_gen/", "style_generator/")) for n in names):
                    raise ValueError(f"snapshot in the flat layout of rounds 1-3: the Adam moments of optimizer '{k}' are positional "
                                     "and the DeepVoxels generator's flat parameter layout has changed since; resume from the "
                                     "per-parameter snapshot format (rounds 4+) or restart the optimizer state")
                if int(np.asarray(f[f"{k}/m"]).size) != int(o.store.numel):
                    raise ValueError(f"snapshot in the flat layout of rounds 1-3: optimizer '{k}' holds {int(np.asarray(f[f'{k}/m']).size)} "
                                     f"moments, the model has {int(o.store.numel)} parameters")
                o.load_state_dict({"t": f[f"{k}/t"], "m": f[f"{k}/m"], "v": f[f"{k}/v"]})
        it = {k[len("iterator/"):]: f[k] for k in f if k.startswith("iterator/")} or None
        return {"iteration": int(f["iteration"]), "iterator": it,
                "log": json.loads(str(f["log"])) if "log" in f else None,
                "elapsed_time": float(f["elapsed_time"]) if "elapsed_time" in f else 0.0}
    for name, opt in optimizers.items():
        store = opt.store
        base = f"updater/optimizer:{name}/"
        t = _get(f, base + "t")
        sd = opt.state_dict()
        for pname in store.names:
            off, n = store.offsets[pname], int(np.prod(store.shapes[pname]))
            for mom in ("m", "v"):
                a = _get(f, f"{base}{pname}/{mom}")
                if a is not None:
                    sd[mom][off:off + n] = np.asarray(a, dtype=np.float32).reshape(-1)
            if t is None:
                t = _get(f, f"{base}{pname}/t")
        sd["t"] = int(t) if t is not None else sd["t"]
        opt.load_state_dict(sd)
        # the optimizer's target link travels in the trainer snapshot too (updater/model:{name}/...): a reference-style resume
        # from the snapshot alone finds the weights there; train_rgbd.py loads the separate Generator_*.npz files first and these
        # agree with them
        flat = store.flat
        for pname in store.names:
            a = _get(f, f"updater/model:{name}/{pname}")
            if a is not None and tuple(np.shape(a)) == tuple(store.shapes[pname]):
                off, n = store.offsets[pname], int(np.prod(store.shapes[pname]))
                flat[off:off + n].copy_(_as_flat_tensor(a, flat))
    it = None
    base = "updater/iterator:main/"
    if _get(f, base + "current_position") is not None and _get(f, base + "order") is not None:
        it = {"pos": np.int64(_get(f, base + "current_position")), "epoch": np.int64(_get(f, base + "epoch")),
              "order": np.asarray(_get(f, base + "order"), dtype=np.int64)}
        rng = _get(f, "rgbd_gan_amd/iterator:main/rng_state")
        if rng is not None:                                            # (a Chainer-written file has no generator state)
            it["rng_state"] = rng
            seed = _get(f, "rgbd_gan_amd/iterator:main/seed")
            if seed is not None:                                       # (never None in a file pack() wrote; a hand-merged one)
                it["seed"] = seed
    log = _get(f, "extensions/LogReport/_log")
    el = _get(f, "_snapshot_elapsed_time")
    return {"iteration": int(_get(f, "updater/iteration")), "iterator": it,
            "log": json.loads(str(log)) if log is not None else None, "elapsed_time": float(el) if el is not None else 0.0}
