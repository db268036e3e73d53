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
VARIANTS = {
    "base": [],
    "no_mfma": [(MFMA, 'asm volatile("" :: "v"(af[i]), "v"(brow[rj][s2]));')],
    "no_lds_reads": [
        ("af[i] = *reinterpret_cast<const bf16x8*>(wbuf + aoff[s2] + i * 16 * 128);",
         "af[i] = __builtin_bit_cast(bf16x8, acc[i][0]);"),
        ("brow[r][s2] = *reinterpret_cast<const bf16x8*>(pbuf + boff[kw][s2] + r * HPW * 128);",
         "brow[r][s2] = __builtin_bit_cast(bf16x8, acc[0][r % TPX]);")],
    "no_barrier": [("            __syncthreads();\n        }\n        if (c_next == 0) {", "        }\n        if (c_next == 0) {")],
    "no_weight_traffic": [
        ("            store_w((t + 1) % 3, Wr[(t + 1) % 3]);\n", ""),
        ("            load_w(t + 3 >= 9 ? c_next : c, 3 * (((t + 3) % 9) % 3) + ((t + 3) % 9) / 3, Wr[t % 3]);", "")],
    "no_patch_traffic": [
        ("            if (t == 6) store_patch((g + 1) & 1);\n", ""),
        ("                load_patch(min(g + 2, g_total - 1));\n", "")],
    "no_epilogue_store": [("*reinterpret_cast<u32x4*>(a.y + o + 8 * h) = out;", 'asm volatile("" :: "v"(out));')],
}


def build():
    os.makedirs(OUT, exist_ok=True)
    src = open(os.path.join(CSRC, "conv.hip")).read()
    for name, edits in VARIANTS.items():
        s = src
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
    shapes = [(128, 128, 128), (64, 256, 256), (128, 64, 128)]
    B = 32
    for H, Cin, Cout in shapes:
        x = torch.randn(B, H, H, Cin, device="cuda:0").to(torch.bfloat16)
        w = torch.randn(Cout, Cin, 3, 3, device="cuda:0")
        bias = torch.zeros(Cout, device="cuda:0")
        fl = 2.0 * B * H * H * Cin * Cout * 9
        for name in VARIANTS:
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
