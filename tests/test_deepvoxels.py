"""DeepVoxels frustum path: oracle known answers on CPU, HIP parity on GPU."""
import numpy as np
import pytest
import torch

from oracle import camera, deepvoxels as odv


def _cams(n, seed=0):
    rng = np.random.RandomState(seed)
    th = rng.uniform(-1, 1, (n, 6)).astype("float32") * np.array([0.3054, 3.1415, 0, 0, 0, 0], "float32")
    return camera.camera_matrices(th)


def test_frustum_constants_and_fractional_row():
    fr = odv.Frustum()
    assert fr.depth == 56 and fr.n == 229376
    assert abs(fr.voxel_size - 0.0171875) < 1e-12 and abs(fr.near_plane - 0.4330127) < 1e-7
    # n = 65: tmp = 65, yc0 = 65 / 64 = 1.015625 (TRUE division: fractional), xc0 = 1
    n = np.array([65], "int32")
    tmp = n - ((n // 4096).astype("float32") * 64 * 64).astype("int32")
    assert float(tmp / 64) == 1.015625 and int(tmp % 64) == 1
    np.testing.assert_allclose(odv.depth_coords(fr)[[0, 28, 55]], [-0.5, 0.0, 27 / 56], atol=1e-7)


def test_proj_idcs_oracle_properties():
    fr = odv.Frustum()
    lin, v = odv.proj_idcs_np(_cams(1)[0], fr)
    assert lin.dtype == np.int32 and v.dtype == np.float32 and v.shape == (3, len(lin))
    assert (np.diff(lin) > 0).all()                          # ordered compaction
    assert (v >= 0).all() and (v < 32).all()
    assert 0.2 * fr.n < len(lin) < 0.9 * fr.n
    far = np.eye(4, dtype="float32")
    far[2, 3] = 100.0
    assert odv.proj_idcs_np(far, fr) is None                 # nothing inside -> None, as the reference


@pytest.mark.gpu
def test_hip_proj_idcs_bit_exact():
    from rgbd_gan_amd.deepvoxel.projection import ProjectionHelper
    fr = odv.Frustum()
    K = np.array([[128., 0, 32., 0], [0, 128., 32., 0], [0, 0, 1, 0], [0, 0, 0, 1]])
    ph = ProjectionHelper(K, K, [64, 64], [64, 64], 0.0, 32 * fr.voxel_size + fr.near_plane, [32, 32, 32],
                          fr.voxel_size, fr.near_plane, fr.depth)
    cams = _cams(5, seed=3)
    idx, coords, counts = ph.compute_proj_idcs_batch(cams)
    for b in range(5):
        lin, v = odv.proj_idcs_np(cams[b], fr)
        m = int(counts[b])
        assert m == len(lin)
        np.testing.assert_array_equal(idx[b, :m].cpu().numpy(), lin)
        np.testing.assert_array_equal(coords[b, :, :m].cpu().numpy().view(np.uint32), v.view(np.uint32))
    one = ph.compute_proj_idcs(cams[2])
    np.testing.assert_array_equal(one[0].cpu().numpy(), odv.proj_idcs_np(cams[2], fr)[0])
    far = np.eye(4, dtype="float32")
    far[2, 3] = 100.0
    assert ph.compute_proj_idcs(far) is None


@pytest.mark.gpu
def test_hip_trilinear_forward_backward():
    from rgbd_gan_amd.deepvoxel import deepvoxel as dv
    fr = odv.Frustum()
    cams = _cams(2, seed=5)
    g = torch.Generator().manual_seed(0)
    grid = torch.randn(2, 8, 32, 32, 32, generator=g)
    outs, grads = [], []
    dout = torch.randn(2, 8, fr.depth, 64, 64, generator=g)
    for b in range(2):
        lin, v = odv.proj_idcs_np(cams[b], fr)
        gb = grid[b:b + 1].clone().requires_grad_(True)
        o = odv.trilinear_torch(gb, lin, v, fr)
        o.backward(dout[b:b + 1])
        outs.append(o.detach())
        grads.append(gb.grad)
    from rgbd_gan_amd.deepvoxel.projection import ProjectionHelper
    K = np.array([[128., 0, 32., 0], [0, 128., 32., 0], [0, 0, 1, 0], [0, 0, 0, 1]])
    ph = ProjectionHelper(K, K, [64, 64], [64, 64], 0.0, 1.0, [32, 32, 32], fr.voxel_size, fr.near_plane, fr.depth)
    idx, coords, counts = ph.compute_proj_idcs_batch(cams)
    gd = grid.cuda().requires_grad_(True)
    od = dv.interpolate_trilinear_batch(gd, idx, coords, counts, [64, 64], fr.depth)
    od.backward(dout.cuda())
    torch.testing.assert_close(od.detach().cpu(), torch.cat(outs), atol=1e-5, rtol=1e-5)
    torch.testing.assert_close(gd.grad.cpu(), torch.cat(grads), atol=1e-3, rtol=1e-4)
    # feature-minor grid (B,G,G,G,F), the voxel generator's own layout: bit-identical forward (same term order), the same
    # gradient in the transposed layout
    gfm = grid.permute(0, 2, 3, 4, 1).contiguous().cuda().requires_grad_(True)
    ofm = dv.interpolate_trilinear_batch(gfm, idx, coords, counts, [64, 64], fr.depth, feature_minor=True)
    assert torch.equal(ofm.detach(), od.detach())
    ofm.backward(dout.cuda())
    torch.testing.assert_close(gfm.grad.permute(0, 4, 1, 2, 3).cpu(), torch.cat(grads), atol=1e-3, rtol=1e-4)
    # reference signature, one sample
    lin, v = odv.proj_idcs_np(cams[0], fr)
    o1 = dv.interpolate_trilinear(grid[:1].cuda(), torch.from_numpy(lin).cuda(), torch.from_numpy(v).cuda(), [64, 64], fr.depth)
    torch.testing.assert_close(o1.cpu(), outs[0], atol=1e-5, rtol=1e-5)
    # straight from the cameras (ProjectionHelper.frustum -> rgbd_trilinear_fwd_frustum / rgbd_trilinear_bwd_frustum): no index list, no
    # zero fill in front of the forward -- every element written, bit-identical to the list forward, zeros outside the grid
    fru = ph.frustum(cams)
    assert dv.frustum_kernels_apply(fru, 8, 2)
    gfr = grid.permute(0, 2, 3, 4, 1).contiguous().cuda().requires_grad_(True)
    ofr = dv.interpolate_trilinear_frustum(gfr, fru)
    assert torch.equal(ofr.detach(), ofm.detach())
    ofr.backward(dout.cuda())
    torch.testing.assert_close(gfr.grad.permute(0, 4, 1, 2, 3).cpu(), torch.cat(grads), atol=1e-3, rtol=1e-4)
    # the two forms of the feature-minor backward at the step's size (32 features): the row-wise list kernel (one line atomic
    # per run of equal voxels along a pixel row) and the sorted bricks (rgbd_trilinear_bwd_frustum: voxel coordinates recomputed
    # from the cameras, one line atomic per distinct voxel of a 16 x 8 x 2 brick) add up the same contributions
    assert getattr(idx, "_frustum", None) is not None and dv.TRILINEAR_BWD_BRICKS
    g32 = torch.randn(2, 32, 32, 32, 32, generator=g).cuda()
    d32 = torch.randn(2, 32, fr.depth, 64, 64, generator=g).cuda()
    got = {}
    for bricks in (True, False):
        dv.TRILINEAR_BWD_BRICKS = bricks
        try:
            gg = g32.clone().requires_grad_(True)
            dv.interpolate_trilinear_batch(gg, idx, coords, counts, [64, 64], fr.depth, feature_minor=True).backward(d32)
            got[bricks] = gg.grad
        finally:
            dv.TRILINEAR_BWD_BRICKS = True
    scale = float(got[False].abs().max())
    assert scale > 1.0
    torch.testing.assert_close(got[True], got[False], atol=2e-5 * scale, rtol=1e-4)
    untouched = got[False] == 0                     # voxels outside every camera's frustum stay exactly zero in both
    assert bool((got[True][untouched] == 0).all()) and int(untouched.sum()) > 0


@pytest.mark.gpu
def test_hip_accumulative_occlusion_forward_backward():
    from rgbd_gan_amd.deepvoxel import deepvoxel as dv
    fr = odv.Frustum()
    g = torch.Generator().manual_seed(1)
    B, F = 2, 32
    vol = (torch.randn(B, F, fr.depth, 64, 64, generator=g) * 0.5)
    W1 = torch.randn(4, F + 1, generator=g)
    b1 = torch.randn(4, generator=g) * 0.1
    W2 = torch.randn(1, 4, generator=g) * 2
    b2 = torch.full((1,), 3.0)                              # around the threshold so the cumulative sum saturates mid-ray
    dfeat = torch.randn(B, F, 64, 64, generator=g)
    ddepth = torch.randn(B, 1, 64, 64, generator=g)
    ref_f, ref_d, ref_g = [], [], []
    params = [t.clone().requires_grad_(True) for t in (W1, b1, W2, b2)]
    for b in range(B):
        vb = vol[b:b + 1].clone().requires_grad_(True)
        f, d, w = odv.occlusion_torch(vb, *params, fr=fr)
        (f * dfeat[b:b + 1]).sum().backward(retain_graph=True)
        (d * ddepth[b:b + 1]).sum().backward()
        ref_f.append(f.detach()); ref_d.append(d.detach()); ref_g.append(vb.grad)
    vd = vol.cuda().requires_grad_(True)
    pd = [t.clone().cuda().requires_grad_(True) for t in (W1, b1, W2, b2)]
    f, d, w = dv.accumulative_occlusion(vd, *pd, threshold=4.0, voxel_size=fr.voxel_size, near_plane=fr.near_plane)
    ((f * dfeat.cuda()).sum() + (d * ddepth.cuda()).sum()).backward()
    torch.testing.assert_close(f.detach().cpu(), torch.cat(ref_f), atol=1e-4, rtol=1e-4)
    torch.testing.assert_close(d.detach().cpu(), torch.cat(ref_d), atol=1e-5, rtol=1e-5)
    wsum = w.sum(dim=2)
    assert float(wsum.max()) <= 1.0 + 1e-5 and float(wsum.min()) >= 0.0
    torch.testing.assert_close(vd.grad.cpu(), torch.cat(ref_g), atol=2e-4, rtol=1e-3)
    for got, ref in zip(pd, params):
        scale = float(ref.grad.abs().max())
        torch.testing.assert_close(got.grad.cpu(), ref.grad, atol=2e-3 * scale, rtol=2e-3)


@pytest.mark.gpu
@pytest.mark.parametrize("B,D,S", [(2, 32, 64), (3, 8, 24)])
def test_single_pass_occlusion_forward_is_the_three_kernel_forward_bit_for_bit(B, D, S):
    """F = 32 grids take occ_fwd_fused_kernel (one pass over the volume); the debug library's switch sends them through the
    score / scan / compose kernels the other widths use.  Same expressions in the same order: every output equal as bits
    (ragged ray counts included: 3 x 24 x 24 is not a multiple of the block)."""
    from rgbd_gan_amd import _lib
    from rgbd_gan_amd.deepvoxel import deepvoxel as dv
    g = torch.Generator().manual_seed(B * 100 + D)
    F = 32
    vol = (torch.randn(B, F, D, S, S, generator=g) * 0.5).cuda()
    W1 = torch.randn(4, F + 1, generator=g).cuda()
    b1 = (torch.randn(4, generator=g) * 0.1).cuda()
    W2 = (torch.randn(1, 4, generator=g) * 2).cuda()
    b2 = torch.full((1,), 3.0).cuda()
    outs = []
    with _lib.debug_library() as lib:
        for unfused in (0, 1):
            lib.rgbd_debug_occ_unfused(unfused)
            try:
                with torch.no_grad():
                    f, d, w = dv.accumulative_occlusion(vol, W1, b1, W2, b2, threshold=4.0, voxel_size=0.05, near_plane=0.5)
                torch.cuda.synchronize()
            finally:
                lib.rgbd_debug_occ_unfused(0)
            outs.append((f.clone(), d.clone(), w.clone()))
    for a, b in zip(*outs):
        assert torch.equal(a.view(torch.int32), b.view(torch.int32))
    assert float(outs[0][2].abs().sum()) > 0


# ---- layout folds of the DeepVoxels networks (deepvoxels_generator.py): HIP ops against the torch formulations
@pytest.mark.gpu
@pytest.mark.parametrize("B,D0,H,C,up", [(2, 4, 4, 64, True), (3, 8, 8, 64, False), (1, 16, 16, 128, True), (2, 32, 32, 64, False)])
def test_fold_depth_taps_matches_torch_and_its_backward_is_the_adjoint(B, D0, H, C, up):
    from rgbd_gan_amd import functional as Fn
    from rgbd_gan_amd.deepvoxels_generator import fold_depth_taps
    g = torch.Generator().manual_seed(B + D0 + C)
    x = torch.randn(B, D0, H, H, C, generator=g).to(torch.bfloat16)
    xr = x.float().requires_grad_(True)
    xs = xr.unsqueeze(2).expand(B, D0, 2, H, H, C).reshape(B, 2 * D0, H, H, C) if up else xr
    ref = fold_depth_taps(xs)
    dy = torch.randn(ref.shape, generator=g).to(torch.bfloat16)
    ref.backward(dy.float())
    xd = x.cuda().requires_grad_(True)
    got = Fn.fold_depth_taps(xd, up)
    assert torch.equal(got.cpu().float(), ref.detach())                      # a pure rearrangement: exact
    got.backward(dy.cuda())
    torch.testing.assert_close(xd.grad.float().cpu(), xr.grad.to(torch.bfloat16).float(), rtol=1e-2, atol=1e-2)


@pytest.mark.gpu
@pytest.mark.parametrize("B,H,C", [(2, 8, 64), (3, 64, 32), (1, 32, 512)])
def test_fold_4x4s2_matches_torch_and_its_backward_is_the_adjoint(B, H, C):
    from rgbd_gan_amd import functional as Fn
    from rgbd_gan_amd.deepvoxels_generator import fold_4x4s2
    g = torch.Generator().manual_seed(B + H + C)
    x = torch.randn(B, H, H, C, generator=g).to(torch.bfloat16)
    xr = x.float().requires_grad_(True)
    ref = fold_4x4s2(xr)
    dy = torch.randn(ref.shape, generator=g).to(torch.bfloat16)
    ref.backward(dy.float())
    xd = x.cuda().requires_grad_(True)
    got = Fn.fold_4x4s2(xd)
    assert torch.equal(got.cpu().float(), ref.detach())
    got.backward(dy.cuda())
    torch.testing.assert_close(xd.grad.float().cpu(), xr.grad.to(torch.bfloat16).float(), rtol=1e-2, atol=1e-2)


@pytest.mark.gpu
@pytest.mark.parametrize("shape,n,dtype", [((5, 32), 64, torch.float32), ((2, 4, 4, 4, 32), 64, torch.bfloat16),
                                           ((7, 3), 64, torch.bfloat16), ((1, 288), 320, torch.float32)])
def test_pad_last_and_its_backward(shape, n, dtype):
    from rgbd_gan_amd import functional as Fn
    x = torch.randn(*shape).to(dtype)
    xd = x.cuda().requires_grad_(True)
    y = Fn.pad_last(xd, n)
    assert y.shape == shape[:-1] + (n,) and torch.equal(y[..., :shape[-1]].cpu(), x)
    assert float(y[..., shape[-1]:].abs().max()) == 0.0
    dy = torch.randn(*y.shape).to(dtype).cuda()
    y.backward(dy)
    assert torch.equal(xd.grad, dy[..., :shape[-1]])


@pytest.mark.gpu
@pytest.mark.parametrize("mode,shape", [(0, (32, 64, 3, 3, 3)), (0, (64, 64, 3, 3, 3)), (1, (512, 32, 4, 4)), (2, (3, 288, 3, 3)),
                                        (2, (32, 32, 1, 1))])
def test_fold_weight_matches_torch_and_its_backward_is_the_adjoint(mode, shape):
    from rgbd_gan_amd import deepvoxels_generator as dg
    g = torch.Generator().manual_seed(mode + shape[0])
    W = torch.randn(*shape, generator=g)
    Wr = W.clone().requires_grad_(True)
    ref = dg._fold_w(Wr, mode)                                  # CPU: the torch formulations
    dy = torch.randn(ref.shape, generator=g)
    ref.backward(dy)
    Wd = W.cuda().requires_grad_(True)
    got = dg._fold_w(Wd, mode)
    assert torch.equal(got.cpu(), ref.detach())
    got.backward(dy.cuda())
    assert torch.equal(Wd.grad.cpu(), Wr.grad)


@pytest.mark.gpu
def test_fold_weight_multi_matches_the_single_launches():
    """rgbd_fold_weight_multi_f32 (all folds of a network / all adjoints of a backward pass in one launch) against
    rgbd_fold_weight_f32 layer by layer: identical bytes forward, identical accumulated gradients backward."""
    from rgbd_gan_amd import kernels
    g = torch.Generator().manual_seed(11)
    layers = [(0, (32, 64, 3, 3, 3), 64, 64), (0, (64, 64, 3, 3, 3), 64, 64), (1, (512, 32, 4, 4), 512, 64),
              (2, (3, 288, 3, 3), 64, 320), (2, (32, 32, 1, 1), 64, 64), (0, (64, 32, 3, 3, 3), 64, 64)]
    fwd, adj, refs_f, refs_a = [], [], [], []
    for mode, shape, cop, cip in layers:
        W = torch.randn(*shape, generator=g).cuda()
        Co, Ci, K = shape[0], shape[1], shape[-1]
        ref = kernels.fold_weight(W, mode, Co, Ci, K, cop, cip)
        out = torch.full_like(ref, float("nan"))
        fwd.append((W, out, mode, Co, Ci, K, cop, cip, False))
        refs_f.append(ref)
        dy = torch.randn(ref.shape, generator=g).cuda()
        base = torch.randn(*shape, generator=g).cuda()
        acc_ref = kernels.fold_weight(dy, mode, Co, Ci, K, cop, cip, adjoint=True, out=base.clone())
        acc = base.clone()
        adj.append((dy, acc, mode, Co, Ci, K, cop, cip, True))
        refs_a.append(acc_ref)
    kernels.fold_weight_multi(fwd)
    kernels.fold_weight_multi(adj)
    for it, ref in zip(fwd, refs_f):
        assert torch.equal(it[1], ref), it[2:8]
    for it, ref in zip(adj, refs_a):
        assert torch.equal(it[1], ref), it[2:8]


@pytest.mark.gpu
def test_packing_through_the_fold_equals_fold_then_pack():
    """rgbd_pack_desc.fold (ABI 20): the bf16 weight images packed straight from reference-shaped master parameters -- the fold done
    in the packing read -- against fold (rgbd_fold_weight_f32) then pack (rgbd_pack_weights), bytes equal; every mode, padded channel
    counts, the tiled and the element-wise path of the packing kernel, a 5-D 1x1x1 master, all in ONE table (= one launch), next
    to an ordinary entry."""
    from rgbd_gan_amd import kernels
    g = torch.Generator().manual_seed(13)
    layers = [(0, (32, 64, 3, 3, 3), 64, 64), (0, (256, 128, 3, 3, 3), 256, 128), (1, (512, 32, 4, 4), 512, 32),
              (1, (20, 3, 4, 4), 20, 3), (2, (3, 288, 3, 3), 64, 320), (2, (32, 32, 1, 1, 1), 64, 64), (2, (24, 40, 3, 3), 64, 64)]
    entries, refs = [], []
    for mode, shape, cop, cip in layers:
        W = torch.randn(*shape, generator=g).cuda()
        Co, Ci, K = shape[0], shape[1], shape[-1]
        W4 = W if mode == 0 or W.dim() == 4 else W.view(Co, Ci, K, K)
        folded = kernels.fold_weight(W4, mode, Co, Ci, K, cop, cip)
        refs.append(kernels.pack_weights(folded, 0.37))
        co, ci, kh, kw = folded.shape
        wf = torch.full((kh * kw, co, ci), float("nan"), dtype=torch.bfloat16, device="cuda")
        wd = torch.full((kh * kw, ci, co), float("nan"), dtype=torch.bfloat16, device="cuda")
        entries.append((W, 0.37, wf, wd, tuple(folded.shape), (mode, Co, Ci)))
    plain = torch.randn(128, 64, 3, 3, generator=g).cuda()
    refs.append(kernels.pack_weights(plain, 0.5))
    entries.append((plain, 0.5, torch.empty_like(refs[-1][0]), torch.empty_like(refs[-1][1])))
    kernels.pack_weights_multi(kernels.build_pack_table(entries))
    for ent, (wf0, wd0), lay in zip(entries, refs, layers + [("plain",)]):
        assert torch.equal(ent[2].view(torch.int16), wf0.view(torch.int16)), lay
        assert torch.equal(ent[3].view(torch.int16), wd0.view(torch.int16)), lay
    with pytest.raises(RuntimeError, match="master"):
        kernels.build_pack_table([(torch.zeros(32, 64, 3, 3, device="cuda"), 1.0, entries[0][2], entries[0][3], (64, 192, 3, 3),
                                  (0, 32, 64))])
