"""Does any kernel read memory nobody wrote?  Every torch.empty / empty_like / new_empty the Python host makes is filled
with NaN (floating dtypes) before the kernels run; an eager single-stream step then shows NaN wherever an output depends
on uninitialised scratch.  (Sequential replays hide such reads: the stale content is last iteration's identical values.)
    python scripts/poison_empty.py [--stage S] [--batch B]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch

_empty, _empty_like, _new_empty = torch.empty, torch.empty_like, torch.Tensor.new_empty
def _poison(t):
    if t.is_cuda and t.is_floating_point() and t.numel() > 0:
        t.fill_(float("nan"))
    return t
torch.empty = lambda *a, **k: _poison(_empty(*a, **k))
torch.empty_like = lambda *a, **k: _poison(_empty_like(*a, **k))
torch.Tensor.new_empty = lambda self, *a, **k: _poison(_new_empty(self, *a, **k))

sys.argv = [sys.argv[0], "/tmp/poison.npz", "--calls", "2", "--eager", "--sequential"] + sys.argv[1:]
import runpy
runpy.run_path(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "dp_worker.py"), run_name="__main__")
L = np.load("/tmp/poison.npz")
for k in ("map", "gen", "dis"):
    g = L[f"{k}/grad"]
    bad = []
    for n, o, sz in zip(L[f"{k}/names"], L[f"{k}/offsets"], L[f"{k}/sizes"]):
        if not np.isfinite(g[o:o + sz]).all():
            bad.append(str(n))
    print(k, "non-finite gradient entries:", int((~np.isfinite(g)).sum()), bad[:12])
print({key: float(L[key]) for key in L.files if key.startswith("obs/")})
