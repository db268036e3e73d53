"""copy_param / soft_copy_param of the reference (common/utils/copy_param.py) over flat parameter stores.

The smoothed (EMA) generator of `keep_smoothed_gen` (updater.py:397-400): target = (1 - tau) * target + tau * source
for every parameter.  With all parameters of a link in one flat fp32 buffer this is ONE launch of rgbd_ema_update per
store instead of two elementwise kernels per parameter tensor.  (No BatchNormalization statistics exist on this
path -- copy_param.py:33-41 is a no-op for these models.)
"""
import torch

from ... import functional, kernels


def _store_pairs(target_link, source_link):
    t, s = dict(target_link.stores), dict(source_link.stores)
    if set(t) != set(s):
        raise ValueError("copy_param: links have different parameter stores")
    for k in s:
        if t[k].names != s[k].names or t[k].numel != s[k].numel:
            raise ValueError("copy_param: links have different parameters")
        yield t[k], s[k]


@torch.no_grad()
def copy_param(target_link, source_link):
    for t, s in _store_pairs(target_link, source_link):
        t.flat.copy_(s.flat)
    functional.bump_weight_epoch()


@torch.no_grad()
def soft_copy_param(target_link, source_link, tau, layers_in_use=None):
    if layers_in_use is None:
        for t, s in _store_pairs(target_link, source_link):
            kernels.ema_update(t.flat, s.flat, tau)
    else:
        target = dict(target_link.namedparams())
        for name, p in source_link.namedparams():
            if any(name.startswith(prefix) for prefix in layers_in_use):
                kernels.ema_update(target[name].detach().view(-1), p.detach().view(-1), tau)
    functional.bump_weight_epoch()
