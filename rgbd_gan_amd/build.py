"""Build librgbdgan_hip.so (gfx950) in-tree with hipcc.  No torch involved: the library is a plain C ABI.

    python -m rgbd_gan_amd.build [--force]
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "librgbdgan_hip.so")
ARCH = "gfx950"

# (source, extra flags).  warp_loss.hip must not contract a*b+c into FMAs: its index math is specified
# unfused so that it is bit-exact against the CPU oracle.
SOURCES = [
    ("elementwise.hip", []),
    ("warp_loss.hip", ["-ffp-contract=off"]),
    ("conv.hip", []),
    ("deepvoxels.hip", ["-ffp-contract=off"]),
    ("step_ops.hip", []),
]
COMMON = ["-O3", "-fPIC", "-std=c++17", f"--offload-arch={ARCH}", "-Wall", "-Wno-unused-function"]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    return "hipcc"


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    hipcc = _hipcc()
    headers = [os.path.join(CSRC, "common.h"), os.path.join(HERE, "..", "include", "rgbd_gan_hip.h")]
    objs = []
    for src, extra in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(CSRC, src.replace(".hip", ".o"))
        if force or _stale(o, [s] + headers):
            cmd = [hipcc] + COMMON + extra + ["-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
        objs.append(o)
    if force or _stale(OUT, objs):
        cmd = [hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", OUT] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(OUT)
