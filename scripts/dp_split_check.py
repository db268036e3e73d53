"""2 ranks sharing cuda:0 (gloo): the split-body arrangement against the whole-body one, pairwise over repeated runs."""
import itertools, os, socket, subprocess, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, "tests", "dp_worker.py")


def two_ranks(tmp, tag, extra):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    if os.environ.get("TURNS"):
        extra = dict(extra, RGBD_SHARE_DEVICE_LOCK=os.path.join(tmp, "turn.lock"))
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), RGBD_DIST_BACKEND="gloo", RGBD_SHARE_DEVICE="1", **extra)
        procs.append(subprocess.Popen([sys.executable, WORKER, os.path.join(tmp, f"{tag}{r}.npz"), "--calls", "4",
                                       "--stage", os.environ.get("STAGE", "10.0")] + sys.argv[1:], env=env))
    for p in procs:
        assert p.wait(timeout=150) == 0
    return np.load(os.path.join(tmp, f"{tag}0.npz"))


def rel(a, b):
    a, b = a.astype("float64"), b.astype("float64")
    return float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30))


with tempfile.TemporaryDirectory() as tmp:
    runs = {}
    for rep in range(int(os.environ.get("REPS", "2"))):
        runs[f"whole{rep}"] = two_ranks(tmp, f"whole{rep}", {"RGBD_DP_NO_SPLIT": "1"})
        runs[f"split{rep}"] = two_ranks(tmp, f"split{rep}", {})
    for a, b in itertools.combinations(sorted(runs), 2):
        print(f"{a} vs {b}: " + "  ".join(f"{k} {rel(runs[a][k + '/grad'], runs[b][k + '/grad']):.2e}" for k in ("map", "gen", "dis")))

    # per-parameter picture of every outlier against a clean run
    names = sorted(runs)
    def score(n):
        return sorted(rel(runs[n]["gen/grad"], runs[m]["gen/grad"]) for m in names if m != n)[len(names) // 2]
    clean = min(names, key=score)
    for n in names:
        if rel(runs[n]["gen/grad"], runs[clean]["gen/grad"]) < 0.02:
            continue
        print(f"--- outlier {n} against {clean}")
        for k in ("gen", "map", "dis"):
            a, b = runs[n], runs[clean]
            for nm, off, sz in zip(a[f"{k}/names"], a[f"{k}/offsets"], a[f"{k}/sizes"]):
                ga, gb = a[f"{k}/grad"][off:off + sz], b[f"{k}/grad"][off:off + sz]
                r = rel(ga, gb)
                if r > 0.02:
                    print(f"   {k}/{nm:28s} rel {r:9.3e}  |a| {np.linalg.norm(ga):9.3e} |b| {np.linalg.norm(gb):9.3e}")
        for key in sorted(a.files):
            if key.startswith("dbg/"):
                d = a[key].astype("float64") - b[key].astype("float64")
                bad = np.flatnonzero(np.abs(d).ravel() > 1e-6 * np.abs(b[key]).max())
                print(f"   {key:14s} rel {rel(a[key], b[key]):9.3e}  entries off {bad.size} of {d.size}"
                      + (f"  first {bad[:6].tolist()} last {bad[-3:].tolist()}" if bad.size else ""))
                if bad.size and key == "dbg/gout1":
                    fa, fb, f0 = a[key].ravel(), b[key].ravel(), a["dbg/gout0"].ravel()
                    xf = a["dbg/x_fake"].ravel()
                    for e in bad[:24]:
                        print(f"      [{e}] b,c,i,j = {np.unravel_index(e, a[key].shape)}  bad {fa[e]:.6e}  clean {fb[e]:.6e}  "
                              f"pre {f0[e]:.6e}  x_fake {xf[e]:.5f}")
                    pre = np.mean(np.abs(fa[bad] - f0[bad]) < 1e-12)
                    print(f"      bad entries equal to the pre-atomic value: {pre:.3f}; runs of consecutive indices: "
                          f"{np.sum(np.diff(bad) > 1) + 1}; shape {a[key].shape}")
        for key in a.files:
            if key.startswith("obs/"):
                print(f"   {key:24s} {float(a[key]):.6f} {float(b[key]):.6f}")
