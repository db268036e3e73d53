"""Screen of the two-stream phase overlap (the default since its hazard was cured; RGBD_DEBUG_WARP_LDS=0
brings the hazard back): the replayed step against the eager single-stream step on fixed inputs (tests/dp_worker.py), N runs each:
rel-L2 of the flat gradient buffers; with RGBD_DEBUG_DUMP=1 also of intermediates of the generator phase.
    NRUNS=16 python scripts/graph_race.py --stage 10 --batch 16"""
import os, subprocess, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
W = os.path.join(ROOT, "tests", "dp_worker.py")
tmp = tempfile.mkdtemp()
def run(name, flags=(), env=None):
    e = dict(os.environ); e.update(env or {})
    r = subprocess.run([sys.executable, W, f"{tmp}/{name}.npz", "--calls", "4"] + list(flags), capture_output=True, text=True, env=e)
    assert r.returncode == 0, r.stderr[-2000:]
    return np.load(f"{tmp}/{name}.npz")
def rel(a, b):
    return np.linalg.norm(a.astype("f8") - b.astype("f8")) / (np.linalg.norm(b.astype("f8")) + 1e-30)
def worst(L, ref, k, top=6):
    rows = []
    for n, o, sz in zip(L[f"{k}/names"], L[f"{k}/offsets"], L[f"{k}/sizes"]):
        a, b = L[f"{k}/grad"][o:o + sz].astype("f8"), ref[f"{k}/grad"][o:o + sz].astype("f8")
        if np.linalg.norm(b) == 0:
            continue
        rows.append((rel(a, b), str(n), float(a @ b / (b @ b))))
    return " ".join(f"{n}:{r:.1e}(x{p:.2f})" for r, n, p in sorted(rows, reverse=True)[:top])
EXTRA = sys.argv[1:]
ref = run("seq", ["--eager", "--sequential"] + EXTRA)
N = int(os.environ.get("NRUNS", "12"))
if os.environ.get("HYBRID"):         # generator phase from graphs, discriminator phase eager on the side stream
    cases = [(f"hybrid {os.environ['HYBRID']} two streams {i}", ["--hybrid", os.environ["HYBRID"]], {}) for i in range(N)]
elif os.environ.get("EAGER_TWO"):      # screen the EAGER two-stream arrangement instead (launches from Python, no graphs)
    cases = [(f"eager two streams {i}", ["--eager", "--concurrent"], {}) for i in range(N)]
else:
    cases = [(f"one stream {i}", ["--sequential"], {}) for i in range(max(2, N // 4))] + \
            [(f"two streams {i}", ["--concurrent"], {}) for i in range(N)]
for i, (name, flags, env) in enumerate(cases):
    try:
        L = run(f"case{i}", list(flags) + EXTRA, env)
    except AssertionError as exc:
        print(name, "FAILED", str(exc)[-300:]); continue
    rels = {k: rel(L[f"{k}/grad"], ref[f"{k}/grad"]) for k in ("map", "gen", "dis")}
    if max(rels["map"], rels["gen"]) < 2e-2 and rels["dis"] < 2e-5:
        print(f"{name}: clean", flush=True)
        continue
    print(f"{name:28s} graphs {int(L['n_graphs'])} | " + " | ".join(f"{k}: {v:.1e}" for k, v in rels.items()) +
          f" | loss_adv {float(L['obs/gen/loss_adv']) - float(ref['obs/gen/loss_adv']):+.1e}", flush=True)
    dk = [k for k in L.files if k.startswith("dbg/")]
    if dk and os.environ.get("RGBD_SAVE_BAD"):
        os.makedirs(os.environ["RGBD_SAVE_BAD"], exist_ok=True)
        np.savez_compressed(os.path.join(os.environ["RGBD_SAVE_BAD"], f"case{i}.npz"), **{k[4:]: L[k] for k in dk},
                            **{"ref_" + k[4:]: ref[k] for k in dk})
    if dk:
        print("      " + "  ".join(f"{k[4:]}: {rel(L[k], ref[k]):.1e}" for k in sorted(dk)), flush=True)
    for k, v in rels.items():
        if os.environ.get("WORST") and (v > 1e-2 or (k == "dis" and v > 1e-5)):
            print(f"      {k}: {worst(L, ref, k)}", flush=True)
