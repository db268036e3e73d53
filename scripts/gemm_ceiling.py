"""What the vendor GEMM (hipBLASLt through torch.matmul) reaches on bf16 GEMMs of the conv layers' shapes: a practical
ceiling for the implicit-GEMM conv kernels next to the 2.5 PFLOP/s datasheet peak."""
import torch
dev = "cuda:0"
def bench(M, N, K, n=20):
    a = torch.randn(M, K, device=dev).to(torch.bfloat16); b = torch.randn(K, N, device=dev).to(torch.bfloat16)
    for _ in range(5): a @ b
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): a @ b
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / n * 1e-3
    return 2.0 * M * N * K / t / 1e12
for name, (M, N, K) in {"128^2 64->128": (32 * 128 * 128, 128, 576), "128^2 128->128": (32 * 128 * 128, 128, 1152),
                        "64^2 128->256": (32 * 64 * 64, 256, 1152), "64^2 256->256": (32 * 64 * 64, 256, 2304),
                        "32^2 256->256": (32 * 32 * 32, 256, 2304), "square 8192": (8192, 8192, 8192)}.items():
    print(f"{name:16s} M={M:7d} N={N:5d} K={K:5d}  {bench(M, N, K):7.0f} TFLOP/s")
