"""Noise floor of the step: the same fixed-input step (tests/dp_worker.py) in several arrangements, every pair compared
per optimizer (cosine / rel-L2 of the flat gradient buffers, weight-update mismatch fraction)."""
import os, socket, subprocess, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
W = os.path.join(ROOT, "tests", "dp_worker.py")
tmp = tempfile.mkdtemp()
extra = sys.argv[1:]
def run(name, *flags, env=None):
    r = subprocess.run([sys.executable, W, f"{tmp}/{name}.npz", "--calls", "4"] + list(flags) + extra, env=env or dict(os.environ),
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
def run2(name):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ps = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   RGBD_DIST_BACKEND="gloo", RGBD_SHARE_DEVICE="1")
        ps.append(subprocess.Popen([sys.executable, W, f"{tmp}/{name}{r}.npz", "--calls", "4"] + extra, env=env,
                                   stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    for p in ps:
        out, _ = p.communicate()
        assert p.returncode == 0, out[-2000:]
run("seqA", "--eager", "--sequential"); run("seqB", "--eager", "--sequential")
run("two", "--eager"); run("twoB", "--eager"); run("graph"); run("graphB"); run("gseq", "--sequential"); run2("rank")
def cos(a, b):
    a, b = a.astype("f8").ravel(), b.astype("f8").ravel()
    return a @ b / (np.linalg.norm(a) * np.linalg.norm(b) + 1e-30)
def rel(a, b):
    return np.linalg.norm(a.astype("f8") - b.astype("f8")) / (np.linalg.norm(b.astype("f8")) + 1e-30)
L = {k: np.load(f"{tmp}/{k}.npz") for k in ("seqA", "seqB", "two", "twoB", "graph", "graphB", "gseq", "rank0")}
for a, b in (("seqB", "seqA"), ("two", "seqA"), ("twoB", "two"), ("graph", "seqA"), ("graphB", "graph"), ("gseq", "seqA"),
             ("rank0", "seqA"), ("rank0", "gseq")):
    row = []
    for k in ("map", "gen", "dis"):
        da, db = L[a][f"{k}/delta"], L[b][f"{k}/delta"]
        mm = float((np.abs(da - db) > 0.05 * np.abs(db).max()).mean())
        row.append(f"{k}: 1-cos {1 - cos(L[a][f'{k}/grad'], L[b][f'{k}/grad']):.2e} rel {rel(L[a][f'{k}/grad'], L[b][f'{k}/grad']):.2e} "
                   f"v-rel {rel(L[a][f'{k}/v'], L[b][f'{k}/v']):.2e} norm {float(L[a][f'{k}/norm']) / float(L[b][f'{k}/norm']) - 1:+.1e} upd-mismatch {mm:.1e}")
    obs = " ".join(f"{key.split('/')[-1]} {float(L[a][key]) - float(L[b][key]):+.1e}" for key in ("obs/gen/loss_adv", "obs/gen/loss_rotate", "obs/dis/loss_adv", "obs/dis/loss_gp"))
    print(f"{a:7s} vs {b:7s} | " + " | ".join(row) + " | " + obs, flush=True)

def per_param(a, b, k, top=12):
    rows = []
    for n, o, sz in zip(L[a][f"{k}/names"], L[a][f"{k}/offsets"], L[a][f"{k}/sizes"]):
        ga, gb = L[a][f"{k}/grad"][o:o + sz], L[b][f"{k}/grad"][o:o + sz]
        nb = np.linalg.norm(gb)
        rows.append((float(rel(ga, gb)) if nb > 0 else float(np.linalg.norm(ga)), str(n), float(np.linalg.norm(ga)), float(nb)))
    for r in sorted(rows, reverse=True)[:top]:
        print(f"   {k}/{r[1]:28s} rel {r[0]:.2e}  |a| {r[2]:.3e} |b| {r[3]:.3e}")
for a, b in (("seqB", "seqA"), ("two", "seqA"), ("twoB", "two"), ("graph", "seqA"), ("gseq", "seqA")):
    print(f"per-parameter worst, {a} vs {b}")
    for k in ("gen", "dis"):
        per_param(a, b, k, top=5)
