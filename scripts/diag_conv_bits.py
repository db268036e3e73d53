"""Is the conv engine's fp32 accumulation the CPU's?  Same bf16 operands through the engine and through torch-CPU fp32
conv2d (and float64): fraction of outputs whose bf16 bits differ, and the engine's distance to the exact value in units
of half a bf16 ulp (<= 1 means 'correctly rounded result of the exact sum')."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.nn.functional as F
from rgbd_gan_amd import kernels

torch.manual_seed(0)
for (B, H, Cin, Cout) in ((2, 16, 256, 256), (2, 8, 256, 256), (2, 64, 128, 128), (1, 128, 64, 64)):
    x = torch.randn(B, H, H, Cin).to(torch.bfloat16)
    w = torch.randn(Cout, Cin, 3, 3)
    inv_c = float(np.sqrt(2) / np.sqrt(Cin * 9))
    wf, _ = kernels.pack_weights(w.cuda(), inv_c)
    y = kernels.conv2d_fprop(x.cuda(), wf, 3, 3, 1).float().cpu().permute(0, 3, 1, 2)
    wr = (w * inv_c).to(torch.bfloat16)
    xin = x.float().permute(0, 3, 1, 2)
    ref32 = F.conv2d(xin, wr.float(), padding=1)
    ref64 = F.conv2d(xin.double(), wr.double(), padding=1)
    r32 = ref32.to(torch.bfloat16).float()
    r64 = ref64.float().to(torch.bfloat16).float()
    ulp = (2.0 ** (torch.floor(torch.log2(ref64.abs().clamp_min(1e-30))) - 7)).float()      # bf16 spacing at the value
    d_engine = ((y.double() - ref64).abs() / (0.5 * ulp.double())).float()
    d_cpu32 = ((r32.double() - ref64).abs() / (0.5 * ulp.double())).float()
    print(f"B{B} H{H} {Cin}->{Cout}: bits differ engine-vs-cpu32 {float((y != r32).float().mean()):.2e}  engine-vs-exact "
          f"{float((y != r64).float().mean()):.2e}  cpu32-vs-exact {float((r32 != r64).float().mean()):.2e} | max distance to exact in "
          f"half-ulps: engine {float(d_engine.max()):.3f} cpu32 {float(d_cpu32.max()):.3f} | fp32-level rel err of cpu32 conv "
          f"{float(((ref32.double() - ref64).abs().max()) / ref64.abs().max()):.1e}", flush=True)
