"""Stress of the warp-loss backward while ANOTHER PROCESS keeps the same GPU busy.
  python scripts/hw/atomic_share_stress.py [--seconds 20] [--partner step|k_gather|k_sp|k_patch|k_wgrad|k_lrelu|conv|warp|idle]
What it found (round 3): built with hipcc's default code generation the kernel goes wrong in 36-56 % of its launches next to
conv_fprop_kernel / conv3x3_sp_kernel (any kernel whose MFMA-issuing waves leave room for it on their SIMD) -- in lanes 48-63,
in values computed by v_pk_mul_f32 / v_pk_add_f32 -- and never when it is built without packed-fp32 instructions, which is
how the library is built now.  To reproduce the fault:
  python -m rgbd_gan_amd.build --packed-fp32 --out /tmp/librgbdgan_pk.so
  RGBD_LIB_PATH=/tmp/librgbdgan_pk.so python scripts/hw/atomic_share_stress.py --partner k_gather
Every process repeats rgbd_warp_loss_bwd on fixed inputs into a freshly initialised buffer and counts the repetitions
whose result differs from its own first one -- by a single bit for the warp-loss backward (its scatter accumulates in
64-bit fixed point with integer atomics since round 3), by more than the rounding of re-ordered fp32 atomics for the
fp32-atomic victims (--victim scatter / bias).
Round 4: victims that are NOT this library's code -- `add_` (torch's in-place fp32 vector add, what gradient accumulation
launches) and `sum2` (two 34 MB fp32 buffers summed into a third: the reduction step of an all-reduce, which under data
parallelism runs beside the other stream's MFMA waves by design) -- built by torch with the compiler's defaults, i.e.
possibly with packed-fp32 instructions; log: profiles/r04/hazard_foreign_fp32_kernels.log."""
import argparse, os, subprocess, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def worker(args):
    import numpy as np, torch
    from rgbd_gan_amd import kernels
    from rgbd_gan_amd.common.loss_functions import LossFuncRotate
    from rgbd_gan_amd.updater import get_camera_matries
    dev = "cuda:0"
    rng = np.random.RandomState(5 + args.seed)
    b, S = args.half, 128
    th = rng.uniform(-0.3, 0.3, (2 * b, 6)).astype("float32")
    th[:, 2] = 0; th[:, 3:] *= 0.1
    cams = get_camera_matries(th)
    lf = LossFuncRotate(np, lambda_geometric=3.0)
    coef = torch.from_numpy(np.asarray(lf.coefficients_for_size(S, cams[:b], cams[b:]), dtype="float32")).to(dev)
    x = torch.from_numpy(rng.uniform(-1, 1, (2 * b, 4, S, S)).astype("float32"))
    x[:, 3] = torch.from_numpy(rng.uniform(0.7, 1.3, (2 * b, S, S)).astype("float32"))
    x = x.to(dev)
    g0 = torch.from_numpy(rng.uniform(-1e-5, 1e-5, (2 * b, 4, S, S)).astype("float32")).to(dev)
    if args.role in ("step", "step_eager"):      # a partner that runs the training step (stylegan config, stage 10, B=8)
        from rgbd_gan_amd.training import DeviceImageIterator, build_training
        from rgbd_gan_amd.utils import yaml_utils
        root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        config = yaml_utils.load(os.path.join(root, "configs", "stylegan_shapenet_car.yml"))
        config.batchsize = 8
        images = np.random.RandomState(0).randint(0, 256, (64, 3, 128, 128)).astype("uint8")
        it = DeviceImageIterator(images, config.batchsize, torch.device(dev), seed=0)
        gen, dis, opt, upd = build_training(config, torch.device(dev), None, iterator=it, nan_check_interval=0)
        upd.iteration = 200000
        upd.use_graphs = args.role == "step"
        t0, n = time.time(), 0
        while time.time() - t0 < args.seconds + 3:
            upd.update(); n += 1
        torch.cuda.synchronize()
        print(f"[{args.role} pid {os.getpid()}] {n} training steps", flush=True)
        return
    if args.role.startswith("k_"):      # a partner that repeats ONE of the library's kernels
        B = 32
        def t(*shape):
            return torch.randn(*shape, device=dev).to(torch.bfloat16)
        name = args.role[2:]
        if name in ("sp", "patch"):
            if name == "patch" or args.variant:
                from rgbd_gan_amd import _lib
                _lib.debug_library().__enter__().rgbd_debug_conv_variant(args.variant or 1)
            xx = t(B, 64, 64, 256); w = torch.randn(256, 256, 3, 3, device=dev)
            wf, wd = kernels.pack_weights(w, 0.02)
            fn = lambda: kernels.conv2d_fprop(xx, wf, 3, 3, 1)
        elif name == "gather":
            xx = t(B, 8, 8, 256); w = torch.randn(256, 256, 3, 3, device=dev)
            wf, wd = kernels.pack_weights(w, 0.02)
            fn = lambda: kernels.conv2d_fprop(xx, wf, 3, 3, 1)
        elif name == "wgrad":
            xx, dy = t(B, 64, 64, 256), t(B, 64, 64, 256)
            fn = lambda: kernels.conv2d_wgrad(xx, dy, 3, 0.02)
        elif name == "lrelu":
            xx, dy = t(B, 128, 128, 64), t(B, 128, 128, 64); bg = torch.zeros(64, device=dev)
            fn = lambda: kernels.lrelu_bwd(dy, xx, 64, bias_grad=bg)
        elif name == "adain":
            xx = t(B, 64, 64, 256); ss = torch.randn(B, 512, device=dev)
            fn = lambda: kernels.adain_fwd(xx, ss)
        else:
            raise SystemExit("unknown kernel " + name)
        t0, n = time.time(), 0
        while time.time() - t0 < args.seconds + 2:
            for _ in range(20):
                fn()
            torch.cuda.synchronize(); n += 20
        print(f"[{args.role} pid {os.getpid()}] {n} launches", flush=True)
        return
    if args.role == "conv":      # a partner that only keeps the GPU busy with matrix work
        a = torch.randn(4096, 4096, device=dev, dtype=torch.bfloat16)
        t0 = time.time()
        while time.time() - t0 < args.seconds:
            for _ in range(50):
                a @ a
            torch.cuda.synchronize()
        return
    def once():
        out = g0.clone()
        kernels.warp_loss_bwd(x[:b], x[b:], coef, 1, 3.0, 0.0, 0.0, None, hinge_lambda=1.0, hinge_min=0.9, grad_scale=2.0,
                              out=(out[:b], out[b:]))
        return out
    if args.victim == "warpfwd":   # the forward kernel: the same projection arithmetic and gathers, NO atomics
        def once():
            loss, zp, warped, idx = kernels.warp_loss_fwd(x[:b], x[b:], coef, 1, 3.0, debug=True, hinge_lambda=1.0, hinge_min=0.9)
            return torch.cat([zp.reshape(-1), warped.reshape(-1), idx.reshape(-1).float(), loss.reshape(-1)])
        g0 = torch.zeros_like(once())
    elif args.victim == "bias":    # the library's other atomics: per-block column sums added to a bias gradient
        yv = torch.randn(8, 64, 64, 128, device=dev).to(torch.bfloat16)
        dv = torch.randn(8, 64, 64, 128, device=dev).to(torch.bfloat16)
        pv = torch.randn(8, 32, 32, 128, device=dev).to(torch.bfloat16)
        def once():
            bg, bg2 = torch.zeros(128, device=dev), torch.zeros(128, device=dev)
            dz = kernels.lrelu_bwd(dv, yv, 128, bias_grad=bg)
            dz2 = kernels.unpool2_lrelu_bwd(pv, yv, (8, 64, 64, 128), bias_grad=bg2)
            return torch.cat([bg, bg2, dz.float().sum().reshape(1), dz2.float().sum().reshape(1)])
        g0 = torch.zeros_like(once())
    elif args.victim != "warp":      # torch-only victims: which kind of memory operation is the one that gets hurt?
        n_el = 1 << 20
        src = torch.randn(n_el, device=dev)
        idx = torch.randint(0, n_el // 4, (n_el,), device=dev)
        base = torch.randn(n_el // 4, device=dev)
        if args.victim == "scatter":        # fp32 atomics (index_add_)
            once = lambda: base.clone().index_add_(0, idx, src)
        elif args.victim == "gather":       # scattered reads, plain coalesced writes
            once = lambda: src[idx] + 1.0
        elif args.victim == "copy":         # streaming read-modify-write
            once = lambda: src * 2.0 + 1.0
        elif args.victim == "div":          # IEEE division (v_div_scale / v_rcp / v_div_fmas / v_div_fixup)
            den = src.abs() + 0.5
            once = lambda: src / den
        elif args.victim == "trans":        # transcendental unit
            once = lambda: torch.exp(src) * torch.rsqrt(src.abs() + 1.0) + torch.log(src.abs() + 1.0)
        elif args.victim == "add_":         # what autograd's AccumulateGrad launches: an in-place fp32 vector add (torch's
            n_big = 34 * (1 << 20) // 4     #   vectorised elementwise kernel; 34 MB = the discriminator's flat gradient buffer)
            ga, gb = torch.randn(n_big, device=dev), torch.randn(n_big, device=dev)
            once = lambda: ga.clone().add_(gb)
        elif args.victim == "sum2":         # an all-reduce's reduction step as RCCL runs it beside the other stream's MFMA waves:
            n_big = 34 * (1 << 20) // 4     #   two 34 MB fp32 buffers -> one (out-of-place sum)
            ga, gb = torch.randn(n_big, device=dev), torch.randn(n_big, device=dev)
            gc = torch.empty_like(ga)
            once = lambda: torch.add(ga, gb, out=gc).clone()
        elif args.victim == "fma":          # plain VALU arithmetic, many operations per element
            def once():
                y = src
                for _ in range(6):
                    y = y * 1.0001 + 0.5 * src
                return y
        g0 = torch.zeros_like(once())
    ref = once()
    scale = float((ref - g0).abs().max())
    bad, worst, t0, n = 0, 0.0, time.time(), 0
    shown = 0
    while time.time() - t0 < args.seconds:
        for _ in range(100):
            got = once()
            d = float((got - ref).abs().max())
            n += 1
            if d > 0 and shown < 3 and args.victim == "warp":
                shown += 1
                idx = torch.nonzero((got != ref).reshape(-1)).reshape(-1)
                gi, ri, zi = got.reshape(-1)[idx].cpu(), ref.reshape(-1)[idx].cpu(), g0.reshape(-1)[idx].cpu()
                i = idx.cpu()
                img, ch, pix = i // (4 * S * S), (i // (S * S)) % 4, i % (S * S)
                print(f"  rep {n}: {len(i)} elements differ; got == initial value in {int((gi == zi).sum())}; "
                      f"channels {sorted(set(ch.tolist()))} images {sorted(set(img.tolist()))}", flush=True)
                for k in range(min(12, len(i))):
                    print(f"    img {int(img[k])} ch {int(ch[k])} row {int(pix[k]) // S} col {int(pix[k]) % S}: got {float(gi[k]):.6e} "
                          f"want {float(ri[k]):.6e} init {float(zi[k]):.6e}", flush=True)
            if d > (1e-3 * scale if args.victim in ('scatter', 'bias') else 0.0):   # the warp-loss backward is bit-reproducible (integer scatter)
                bad += 1
                worst = max(worst, d / scale)
    print(f"[{args.role} pid {os.getpid()}] {n} repetitions, {bad} differ from the first by > 1e-3 of the largest "
          f"contribution (worst {worst:.3g}x)", flush=True)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--procs", type=int, default=2)
    ap.add_argument("--seconds", type=float, default=20.0)
    ap.add_argument("--half", type=int, default=2)
    ap.add_argument("--partner", default="warp", help="warp | conv | idle | step | step_eager | k_sp | k_patch | k_gather | k_wgrad | k_lrelu | k_adain")
    ap.add_argument("--role", default=None)
    ap.add_argument("--victim", default="warp", choices=["warp", "warpfwd", "bias", "scatter", "gather", "copy", "div", "trans", "fma", "add_", "sum2"])
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--variant", type=int, default=0, help="partner k_sp: rgbd_debug_conv_variant (11-16: knock-outs)")
    ap.add_argument("--same-seed", action="store_true", help="every process works on identical data")
    args = ap.parse_args()
    if args.role:
        worker(args)
    else:
        roles = ["warp"] + ([] if args.partner == "idle" else [args.partner] * (args.procs - 1))
        ps = [subprocess.Popen([sys.executable, __file__, "--role", r, "--seconds", str(args.seconds), "--half", str(args.half),
                                "--seed", str(0 if args.same_seed else i), "--victim", args.victim, "--variant", str(args.variant)]) for i, r in enumerate(roles)]
        sys.exit(max(p.wait() for p in ps))
