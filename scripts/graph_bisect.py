"""Which captured phase makes the replayed step differ from the eager one?  tests/dp_worker.py with one phase captured at
a time (the others eager), each compared with the eager sequential run."""
import os, subprocess, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
W = os.path.join(ROOT, "tests", "dp_worker.py")
tmp = tempfile.mkdtemp()
def run(name, *flags):
    r = subprocess.run([sys.executable, W, f"{tmp}/{name}.npz", "--calls", "4"] + list(flags), capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    return np.load(f"{tmp}/{name}.npz")
def rel(a, b):
    return np.linalg.norm(a.astype("f8") - b.astype("f8")) / (np.linalg.norm(b.astype("f8")) + 1e-30)
ref = run("seq", "--eager", "--sequential")
cases = [("eager2", ["--eager"])] + [(p, ["--graph-phases", p]) for p in ("prep", "dis", "gen_a", "dfw", "gen_b", "join", "opt")] + \
        [("all", []), ("all_again", []), ("all_seq", ["--sequential"]), ("dis+dfw", ["--graph-phases", "dis,dfw"]),
         ("gen_a+gen_b", ["--graph-phases", "gen_a,gen_b"])]
for name, flags in cases:
    L = run(name, *flags)
    print(f"{name:12s} graphs {int(L['n_graphs'])} | " + " | ".join(
        f"{k}: grad rel {rel(L[f'{k}/grad'], ref[f'{k}/grad']):.1e} v rel {rel(L[f'{k}/v'], ref[f'{k}/v']):.1e}" for k in ("map", "gen", "dis")), flush=True)
