"""Where and when the workgroups of one conv3x3_dw_kernel launch ran (debug library): CU of every workgroup, how many
workgroups were resident per CU at the same time, start / end spread.   python scripts/dw_census.py [H Cin Cout]"""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rgbd_gan_amd import kernels, _lib
lib = _lib.debug_library().__enter__()
H, Cin, Cout = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (128, 64, 128)
B = int(os.environ.get("B", "32"))
lib.rgbd_debug_dw_census.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_int]
lib.rgbd_debug_dw_census(1, None, 0)
lib.rgbd_debug_conv_variant(int(os.environ.get("VARIANT", "7")))
x = torch.randn(B, H, H, Cin, device="cuda").to(torch.bfloat16)
w = torch.randn(Cout, Cin, 3, 3, device="cuda")
wf, wd = kernels.pack_weights(w, 0.05)
for _ in range(3):
    y = kernels.conv2d_fprop(x, wf, 3, 3, 1, lrelu_channels=Cout)
torch.cuda.synchronize()
n = 512
out = np.zeros((n, 8), dtype=np.uint32)
lib.rgbd_debug_dw_census(1, out.ctypes.data, n)
t0 = out[:, 2].astype(np.uint64) | (out[:, 3].astype(np.uint64) << 32)
t1 = out[:, 4].astype(np.uint64) | (out[:, 5].astype(np.uint64) << 32)
hw, xcc = out[:, 0], out[:, 1] & 0xf
cu = (hw >> 8) & 0xf; sh = (hw >> 12) & 1; se = (hw >> 13) & 0x7      # HW_ID: wave 3:0 simd 5:4 pipe 7:6 cu 11:8 sh 12 se 15:13
key = xcc.astype(np.int64) * 4096 + se * 256 + sh * 16 + cu
live = t1 > 0
if os.environ.get("VARIANT") in ("31", "32"):
    live[:] = False
print("workgroups that reported:", int(live.sum()), " distinct CUs:", len(set(key[live])))
if not live.any():
    per = {}
else:
    pass
base = t0[live].min() if live.any() else 0
s0 = (t0 - base) / 100.0; s1 = (t1 - base) / 100.0      # s_memrealtime ticks at 100 MHz -> us
if live.any():
    print("start us: min %.1f max %.1f   end us: min %.1f max %.1f" % (s0[live].min(), s0[live].max(), s1[live].min(), s1[live].max()))
from collections import defaultdict
per = defaultdict(list)
for i in range(n):
    if live[i]: per[int(key[i])].append((s0[i], s1[i], i))
ov = 0
for k, v in per.items():
    v.sort()
    for a in range(len(v)):
        for b in range(a + 1, len(v)):
            if v[b][0] < v[a][1] - 0.5: ov += 1
print("workgroups per CU:", sorted(set(len(v) for v in per.values())), " overlapping pairs on a CU:", ov)
for k in list(per)[:4]:
    print(" CU", hex(k), [(round(a, 1), round(b, 1), i) for a, b, i in per[k]])
if os.environ.get("VARIANT") == "30":          # in-kernel stamps: cycles per phase, per wave
    big = np.zeros((4096, 8), dtype=np.uint32)
    lib.rgbd_debug_dw_census(1, big.ctypes.data, 4096)
    st = big.reshape(-1)[8 * 1024:8 * 1024 + 512 * 16].reshape(512, 4, 4).astype(np.float64)
    for nm, sel in (("older WGs", slice(0, 256)), ("younger WGs", slice(256, 512))):
        m = st[sel].mean(axis=0)
        for w in range(4):
            print(f"{nm} wave {w} ({'W' if w < 2 else 'H'}): quarters 0-2 {m[w,0]:9.0f}  wait+barrier {m[w,1]:9.0f}  quarter 3 + DMA {m[w,2]:9.0f}  epilogue {m[w,3]:9.0f}  cycles; sum {m[w].sum():9.0f}")
if os.environ.get("VARIANT") in ("31", "32"):          # the 8-wave kernel's stamps: cycles per phase, per wave (roles: 0,1,6,7 weights; 2-5 halo)
    big = np.zeros((4096, 8), dtype=np.uint32)
    lib.rgbd_debug_dw_census(1, big.ctypes.data, 4096)
    st = big.reshape(-1)[8 * 1024:8 * 1024 + 256 * 8 * 5].reshape(256, 8, 5).astype(np.float64)
    m = st.mean(axis=0)
    for w in range(8):
        role = "W" if w in (0, 1, 6, 7) else "H"
        print(f"wave {w} ({role}): block 1 {m[w,0]:9.0f}  wait+barrier {m[w,1]:9.0f}  block 2 + DMA, steps with halo DMA {m[w,2]:9.0f} / without {m[w,3]:9.0f}  epilogue {m[w,4]:9.0f}  cycles; sum {m[w].sum():9.0f}")
