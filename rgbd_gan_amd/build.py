"""Build librgbdgan_hip.so (gfx950) in-tree with hipcc.  No torch involved: the library is a plain C ABI.

    python -m rgbd_gan_amd.build [--force]
    python -m rgbd_gan_amd.build --debug            # librgbdgan_hip_debug.so: + the A/B reference kernels and planner switches
    python -m rgbd_gan_amd.build --packed-fp32 --out /tmp/librgbdgan_pk.so     # A/B library for scripts/hw/ (never shipped)
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "librgbdgan_hip.so")
# The debug library = the same sources with -DRGBD_DEBUG_BUILD: it additionally carries round 1's register-staged 3x3 kernel,
# the tap-split weight-gradient body, the timing knock-outs of the pipelined kernel and the two process-wide planner switches
# that select them (csrc/rgbd_debug.h).  Tests and scripts/ load it for cross-checks and A/B timing; the training path never.
DEBUG_OUT = os.path.join(HERE, "librgbdgan_hip_debug.so")
ARCH = "gfx950"

# (source, extra flags).  warp_loss.hip must not contract a*b+c into FMAs: its index math is specified
# unfused so that it is bit-exact against the CPU oracle.
SOURCES = [
    ("elementwise.hip", []),
    ("warp_loss.hip", ["-ffp-contract=off"]),
    ("conv.hip", ["-Wno-inline-asm"]),      # the M0 clobber of lds_dma16 (reserved register: see the comment there)
    ("deepvoxels.hip", ["-ffp-contract=off"]),
    ("step_ops.hip", []),
    ("mxfp8.hip", []),
]
# NO packed-fp32 VALU instructions (v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32) anywhere in the library: on MI355X a wave
# that executes them while ANOTHER wave on the same SIMD issues MFMAs gets wrong results in lanes 48-63 now and then
# (found with scripts/hw/atomic_share_stress.py: the warp-loss backward next to the convolution kernels of a second stream
# or process, 36-56 % of its launches wrong; the same source compiled without the feature: 0 of 560 000.  DESIGN.md
# section 3).  hipcc vectorises pairs of fp32 operations into these instructions by default on gfx90a and later; the
# feature is switched off for the device compilation (the host half of the same command prints a note that it does not
# know the feature, which _run() drops).  tests/test_isa_cpu.py checks the built code objects.
NO_PACKED_FP32 = ["-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"]
COMMON = ["-O3", "-fPIC", "-std=c++17", f"--offload-arch={ARCH}", "-Wall", "-Wno-unused-function"] + NO_PACKED_FP32


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


def _run(cmd):
    r = subprocess.run(cmd, stderr=subprocess.PIPE, text=True)
    err = "\n".join(ln for ln in r.stderr.splitlines() if "is not a recognized feature for this target" not in ln)
    if err.strip():
        print(err, file=sys.stderr, flush=True)
    if r.returncode != 0:
        raise subprocess.CalledProcessError(r.returncode, cmd)


def build(force=False, verbose=True, out=OUT, packed_fp32=False, debug=False):
    """packed_fp32=True (with another `out`): the compiler's default code generation, for the A/B of DESIGN.md section 3
    (scripts/hw/atomic_share_stress.py with RGBD_LIB_PATH); objects go to a directory next to `out`.
    debug=True: the debug library (DEBUG_OUT unless `out` says otherwise)."""
    hipcc = _hipcc()
    if debug and out == OUT:
        out = DEBUG_OUT
    common = [f for f in COMMON if packed_fp32 is False or f not in NO_PACKED_FP32] + (["-DRGBD_DEBUG_BUILD"] if debug else [])
    objdir = CSRC if out == OUT else os.path.splitext(out)[0] + "_obj"
    if packed_fp32 and out == OUT:
        raise ValueError("the shipped library is built without packed-fp32 instructions; give another --out")
    os.makedirs(objdir, exist_ok=True)
    headers = [os.path.join(CSRC, "common.h"), os.path.join(CSRC, "rgbd_debug.h"),
               os.path.join(HERE, "..", "include", "rgbd_gan_hip.h")]
    objs = []
    for src, extra in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(objdir, src.replace(".hip", ".o"))
        if force or _stale(o, [s] + headers):
            cmd = [hipcc] + common + extra + ["-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd), flush=True)
            _run(cmd)
        objs.append(o)
    if force or _stale(out, objs):
        cmd = [hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", out] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return out


if __name__ == "__main__":
    target = sys.argv[sys.argv.index("--out") + 1] if "--out" in sys.argv else OUT
    print(build(force="--force" in sys.argv, out=target, packed_fp32="--packed-fp32" in sys.argv, debug="--debug" in sys.argv))
