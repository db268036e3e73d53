"""Where does conv3x3_patch_kernel spend its time?  Builds knock-out variants of csrc/conv.hip (textual edits, results
are WRONG by construction -- timing only) into scripts/_probe/ and times one layer shape with each.

    python scripts/patch_probe.py --build      (CPU container: hipcc cross-compiles the variants)
    python scripts/patch_probe.py              (GPU box: times them)
"""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CSRC = os.path.join(ROOT, "rgbd_gan_amd", "csrc")
OUT = os.path.join(ROOT, "scripts", "_probe")

MFMA = "acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], brow[rj][s2], acc[i][j], 0, 0, 0);"
RD_ROWS = "brow[r][s2] = *reinterpret_cast<const bf16x8*>(pbuf + boff[kw][s2] + r * HPW * 128);"
RD_A = "af[i] = *reinterpret_cast<const bf16x8*>(wbuf + aoff[s2] + i * 16 * 128);"
ST_W = "            store_w((t + 1) % 3, Wr[(t + 1) % 3]);\n"
LD_W = "            load_w(t + 3 >= 9 ? c_next : c, 3 * (((t + 3) % 9) % 3) + ((t + 3) % 9) / 3, Wr[t % 3]);   // Wr[t % 3] went to LDS one step ago\n"
ST_P = "            if (t == 6) store_patch((g + 1) & 1);\n"
LD_P = "                load_patch(min(g + 2, g_total - 1));\n"
BAR = "            __syncthreads();\n        }\n        if (c_next == 0) {"
NOBAR = "        }\n        if (c_next == 0) {"
# operands that are not re-read: taken once from the accumulators' initial zeros, opaque to the optimiser
no_reads = [(RD_ROWS, "if (pbuf == patch_lds && kw == 0) { brow[r][s2] = *reinterpret_cast<const bf16x8*>(pbuf + boff[kw][s2]); }"),
            (RD_A, "af[i] = brow[i % NR][s2];")]
no_traffic = [(ST_W, ""), (LD_W, ""), (ST_P, ""), (LD_P, "")]
VARIANTS = {
    "base": [],
    "mfma_only": no_reads + no_traffic + [(BAR, NOBAR)],
    "mfma+barrier": no_reads + no_traffic,
    "mfma+reads": no_traffic + [(BAR, NOBAR)],
    "mfma+reads+barrier": no_traffic,
    "mfma+traffic+barrier": no_reads,
    "no_barrier": [(BAR, NOBAR)],
    "no_weight_traffic": [(ST_W, ""), (LD_W, "")],
    "no_patch_traffic": [(ST_P, ""), (LD_P, "")],
    "no_epilogue_store": [("*reinterpret_cast<u32x4*>(a.y + o + 8 * h) = out;", 'asm volatile("" :: "v"(out));')],
}


REVISIONS = [r for r in os.environ.get("PROBE_REVS", "").split(",") if r]   # A/B against committed versions of conv.hip


def build():
    os.makedirs(OUT, exist_ok=True)
    src = open(os.path.join(CSRC, "conv.hip")).read()
    only = os.environ.get("PROBE_ONLY")
    todo = {k: v for k, v in VARIANTS.items() if not only or k in only.split(",")}
    for r in REVISIONS:
        todo["rev_" + r] = subprocess.check_output(["git", "show", f"{r}:rgbd_gan_amd/csrc/conv.hip"], cwd=ROOT, text=True)
    for name, edits in todo.items():
        s = src
        if isinstance(edits, str):
            s, edits = edits, []
        for old, new in edits:
            assert old in s, (name, old)
            s = s.replace(old, new)
        path = os.path.join(OUT, f"conv_{name}.hip")
        open(path, "w").write(s)
        obj = path.replace(".hip", ".o")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-fPIC", "-std=c++17", "--offload-arch=gfx950",
                               "-Wno-unused-function", "-I", CSRC, "-I", os.path.join(ROOT, "include"), "-c", path, "-o", obj])
        others = [os.path.join(CSRC, f) for f in ("elementwise.o", "warp_loss.o", "deepvoxels.o")]
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-shared", "-fPIC", "--offload-arch=gfx950", "-o",
                               os.path.join(OUT, f"lib_{name}.so"), obj] + others)
        os.remove(path); os.remove(obj)
        print("built", name, flush=True)


def run():
    import numpy as np, torch
    from rgbd_gan_amd import _lib, kernels
    shapes = [(128, 128, 128), (64, 256, 256), (128, 64, 128), (64, 128, 256), (32, 256, 256), (128, 128, 64)]
    B = 32
    for H, Cin, Cout in shapes:
        x = torch.randn(B, H, H, Cin, device="cuda:0").to(torch.bfloat16)
        w = torch.randn(Cout, Cin, 3, 3, device="cuda:0")
        bias = torch.zeros(Cout, device="cuda:0")
        fl = 2.0 * B * H * H * Cin * Cout * 9
        names = sorted(f[4:-3] for f in os.listdir(OUT) if f.startswith("lib_") and f.endswith(".so"))
        for name in names:
            _lib.LIB_PATH = os.path.join(OUT, f"lib_{name}.so")
            _lib._lib = None
            wf, _ = kernels.pack_weights(w, float(np.sqrt(2.0 / (Cin * 9))))
            fn = lambda: kernels.conv2d_fprop(x, wf, 3, 3, 1, bias=bias, lrelu_channels=Cout)
            for _ in range(3): fn()
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): fn()
            e1.record(); torch.cuda.synchronize()
            t = e0.elapsed_time(e1) / 10 * 1e3
            print(f"H={H} {Cin}->{Cout} {name:18s} {t:7.1f} us  ({fl / t / 1e6:6.0f} TF-equivalent)", flush=True)


if __name__ == "__main__":
    build() if "--build" in sys.argv else run()
