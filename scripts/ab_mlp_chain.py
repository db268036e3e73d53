"""A/B of the fused mapping-network chain (rgbd_mlp_fwd / rgbd_mlp_bwd: one launch per pass) against the per-layer launches it
replaced (8 forward + 16 backward), same box, same process, alternating: bench.py's own workload runner at the 8-GPU job's
per-GPU shape (B = 8) and at the default shape (B = 32), side budgets measured for each arm.

    python scripts/ab_mlp_chain.py [--steps 60] [--rounds 2]"""
import argparse
import copy
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--rounds", type=int, default=2)
    a = ap.parse_args()
    import torch
    import bench
    from rgbd_gan_amd import kernels
    from rgbd_gan_amd.dist import Communicator
    sys.argv = [sys.argv[0]]
    base = bench.parse()
    base.steps, base.warmup, base.no_cpu_baseline, base.no_roofline, base.no_other_configs = a.steps, 12, True, True, True
    comm = Communicator()
    device = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    fused_rule = kernels.mlp_supported
    shapes = [("B=8 (ffhq_stylegan_occlusion.yml)", {"config": os.path.join(ROOT, "configs", "ffhq_stylegan_occlusion.yml"), "batch": 8}),
              ("B=32 (default)", {})]
    for name, over in shapes:
        rows = {"fused": [], "per-layer": []}
        for _ in range(a.rounds):
            for arm in ("fused", "per-layer"):
                kernels.mlp_supported = fused_rule if arm == "fused" else (lambda *args: False)
                args = copy.copy(base)
                for k, v in over.items():
                    setattr(args, k, v)
                line = bench.run_workload(args, comm, device)
                rows[arm].append((line["value"], line["ms_per_step"]))
        kernels.mlp_supported = fused_rule
        print(f"{name}: " + "  |  ".join(f"{arm}: " + ", ".join(f"{v:.0f} img/s ({ms:.3f} ms)" for v, ms in r) for arm, r in rows.items()),
              flush=True)


if __name__ == "__main__":
    main()
