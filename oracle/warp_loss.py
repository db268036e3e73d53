"""3D-consistency (warp) loss of the reference, restated for the CPU.

Test infrastructure only (see oracle/__init__.py).  PARITY UNPINNED.

Two restatements of common/loss_functions.py:31-228 live here:

* ``forward_np``  -- NumPy fp32, every operation spelled out in the evaluation
  order that SURVEY.md section 8(c) defines as "bit-exact" (left-to-right, no FMA
  contraction, IEEE division, truncating int cast).  This is the checker for the
  integer outputs (u0, v0, mask) and the fp32 forward values of the HIP kernel.
* ``loss_torch``  -- the same math in differentiable torch-CPU ops, used for the
  gradients (autograd reproduces Chainer's backward: gather -> scatter-add,
  clip gradient 1 inside the interval, sign() for the L1 terms).
"""
import numpy as np
import torch

F32 = np.float32


def intrinsics(size, K=None):
    """loss_functions.py:39-61 (init_params) for a first call at `size`.

    Returns K, inv_K (3,3 float32) and p (3, size*size) with p[0]=column j,
    p[1]=row i, p[2]=1 at flat index n = i*size + j.
    """
    if K is not None:
        K = np.array(np.asarray(K)[:3, :3], "float32")
        K[:2] *= size / K[0, 2] / 2
    else:
        K = np.array([[size * 2, 0, size / 2],
                      [0, size * 2, size / 2],
                      [0, 0, 1]], dtype="float32")
    inv_K = np.linalg.inv(K).astype("float32")
    jj, ii = np.meshgrid(np.arange(size), np.arange(size))
    p = np.asarray([jj, ii, np.ones((size, size))], dtype="float32").reshape(3, -1)
    return K, inv_K, p


def relative_pose(cam, cam_rot):
    """loss_functions.py:85-91: R = R2^T R1, t = R1^T (t2 - t1), float32."""
    R1 = cam[:, :3, :3]
    R2 = cam_rot[:, :3, :3]
    t1 = cam[:, :3, -1:]
    t2 = cam_rot[:, :3, -1:]
    R = np.matmul(R2.transpose(0, 2, 1), R1).astype("float32")
    t = np.matmul(R1.transpose(0, 2, 1), t2 - t1).astype("float32")
    return R, t


def warp_coefficients(K, inv_K, R, t):
    """The per-pair 3x3 / 3x1 constants of warp and inv_warp.

    loss_functions.py:174  A  = (K R) K^-1,  c  = (K R) t       (zp' = A (z p) - c)
    loss_functions.py:181  A' = (K R^T) K^-1, c' = K t          (zp'_rot = A'(z_rot p) + c')
    Formed with np.matmul in exactly that association, float32.
    """
    KR = np.matmul(K, R)
    A = np.matmul(KR, inv_K).astype(F32)
    c = np.matmul(KR, t).astype(F32)[:, :, 0]
    inv_R = R.transpose(0, 2, 1)
    A2 = np.matmul(np.matmul(K, inv_R), inv_K).astype(F32)
    c2 = np.matmul(K, t).astype(F32)[:, :, 0]
    return A, c, A2, c2


def _project(A, c, sign, z, p):
    """zp_k = ((A_k0*(z p0) + A_k1*(z p1)) + A_k2*(z p2)) -/+ c_k, all fp32, unfused."""
    zp0 = z * p[0][None]
    zp1 = z * p[1][None]
    zp2 = z * p[2][None]
    out = []
    for k in range(3):
        s = (A[:, k, 0, None] * zp0 + A[:, k, 1, None] * zp1) + A[:, k, 2, None] * zp2
        s = s - c[:, k, None] if sign < 0 else s + c[:, k, None]
        out.append(s.astype(F32))
    return np.stack(out, axis=2)  # (b, hw, 3)


def _bilinear_np(img, zp):
    """loss_functions.py:185-228 (bilinear) in NumPy.

    Returns warped (b*hw, C), mask (b*hw,) bool, and the masked integer taps.
    Reproduces line 219 (``u1 = u0 * mask`` -- the "+1" row taps read row u0).
    """
    b, hw, _ = zp.shape
    _, C, h, w = img.shape
    zpf = zp.reshape(-1, 3)
    den = np.clip(zpf[:, 2], F32(1e-4), F32(10000)).astype(F32)
    x = (zpf[:, 0] / den).astype(F32)
    y = (zpf[:, 1] / den).astype(F32)
    u, v = y, x                                   # line 202: swap
    u0 = u.astype("int32")
    v0 = v.astype("int32")
    u1 = u0 + 1
    v1 = v0 + 1
    u0f, u1f, v0f, v1f = (a.astype(F32) for a in (u0, u1, v0, v1))
    w1 = (u1f - u) * (v1f - v)
    w2 = (u - u0f) * (v1f - v)
    w3 = (u1f - u) * (v - v0f)
    w4 = (u - u0f) * (v - v0f)
    mask = (u >= 0) & (u < h - 1) & (v >= 0) & (v < w - 1) & (zpf[:, 2] > F32(1e-4))
    u0m = u0 * mask
    u1m = u0m * mask                              # line 219
    v0m = v0 * mask
    v1m = v1 * mask
    mf = mask.astype(F32)
    w1, w2, w3, w4 = w1 * mf, w2 * mf, w3 * mf, w4 * mf
    bi = np.arange(b * hw) // hw
    warped = (w1[:, None] * img[bi, :, u0m, v0m] + w2[:, None] * img[bi, :, u1m, v0m] +
              w3[:, None] * img[bi, :, u0m, v1m] + w4[:, None] * img[bi, :, u1m, v1m]).astype(F32)
    return warped, mask, (u0m.astype("int32"), v0m.astype("int32"), v1m.astype("int32"))


def _mae(a, b):
    """chainer F.mean_absolute_error: sum |a-b| / size (accumulated in float64 here
    so the checker's own summation error is negligible)."""
    d = (a - b).astype(F32)
    return np.abs(d).astype(np.float64).sum() / d.size


def forward_np(img, cam, img_rot, cam_rot, occlusion_aware=False, lambda_geometric=3.0,
               K=None, max_depth=None, min_depth=None):
    """loss_functions.py:63-146 (LossFuncRotate.__call__, norm='l1') in NumPy.

    img, img_rot: (b,4,S,S) float32; cam, cam_rot: (b,4,4) float32.
    Returns a dict with the scalar loss (float64 accumulate) and every
    intermediate the HIP debug outputs expose.
    """
    img = np.asarray(img, F32)
    img_rot = np.asarray(img_rot, F32)
    b, C, S, _ = img.shape
    Kmat, inv_K, p = intrinsics(S, K)
    R, t = relative_pose(np.asarray(cam, F32), np.asarray(cam_rot, F32))
    A, c, A2, c2 = warp_coefficients(Kmat, inv_K, R, t)
    z = img[:, -1].reshape(b, -1)
    z_rot = img_rot[:, -1].reshape(b, -1)
    zp = _project(A, c, -1, z, p)
    zp_rot = _project(A2, c2, +1, z_rot, p)
    warped, mask, taps = _bilinear_np(img_rot, zp)
    warped_rot, mask_rot, taps_rot = _bilinear_np(img, zp_rot)

    def target(src, zpx, m):
        rgb = src[:, :-1].transpose(0, 2, 3, 1).reshape(-1, C - 1)
        return (np.concatenate([rgb, zpx[:, :, 2].reshape(-1, 1)], axis=1) * m[:, None]).astype(F32)

    tgt = target(img, zp, mask)
    tgt_rot = target(img_rot, zp_rot, mask_rot)
    vis = np.ones(b * S * S, bool)
    vis_rot = np.ones(b * S * S, bool)
    if occlusion_aware:
        vis = warped[:, -1] > zp[:, :, 2].reshape(-1)
        vis_rot = warped_rot[:, -1] > zp_rot[:, :, 2].reshape(-1)
    if max_depth is not None:
        vis = vis & (z.reshape(-1) < max_depth)
        vis_rot = vis_rot & (z_rot.reshape(-1) < max_depth)
    if min_depth is not None:
        vis = vis & (z.reshape(-1) > min_depth)
        vis_rot = vis_rot & (z_rot.reshape(-1) > min_depth)
    wv = warped * vis[:, None]
    tv = tgt * vis[:, None]
    wv_rot = warped_rot * vis_rot[:, None]
    tv_rot = tgt_rot * vis_rot[:, None]
    loss = _mae(wv[:, :-1], tv[:, :-1]) + _mae(wv_rot[:, :-1], tv_rot[:, :-1])
    loss += _mae(wv[:, -1], tv[:, -1]) * lambda_geometric + _mae(wv_rot[:, -1], tv_rot[:, -1]) * lambda_geometric
    return dict(loss=loss, A=A, c=c, A2=A2, c2=c2, zp=zp, zp_rot=zp_rot,
                warped=warped, warped_rot=warped_rot, mask=mask, mask_rot=mask_rot,
                u0=taps[0], v0=taps[1], v1=taps[2], u0_rot=taps_rot[0], v0_rot=taps_rot[1], v1_rot=taps_rot[2],
                vis=vis, vis_rot=vis_rot)


# ---------------------------------------------------------------- differentiable (torch-CPU) restatement

def _bilinear_torch(img, zp):
    b, hw, _ = zp.shape
    _, C, h, w = img.shape
    zpf = zp.reshape(-1, 3)
    den = torch.clamp(zpf[:, 2], 1e-4, 10000)
    x = zpf[:, 0] / den
    y = zpf[:, 1] / den
    u, v = y, x
    u0 = u.detach().to(torch.int32)
    v0 = v.detach().to(torch.int32)
    u1 = u0 + 1
    v1 = v0 + 1
    w1 = (u1 - u) * (v1 - v)
    w2 = (u - u0) * (v1 - v)
    w3 = (u1 - u) * (v - v0)
    w4 = (u - u0) * (v - v0)
    ud, vd = u.detach(), v.detach()
    mask = (ud >= 0) & (ud < h - 1) & (vd >= 0) & (vd < w - 1) & (zpf[:, 2].detach() > 1e-4)
    mi = mask.to(torch.int64)
    u0m = u0.to(torch.int64) * mi
    u1m = u0m * mi
    v0m = v0.to(torch.int64) * mi
    v1m = v1.to(torch.int64) * mi
    mf = mask.to(img.dtype)
    w1, w2, w3, w4 = w1 * mf, w2 * mf, w3 * mf, w4 * mf
    bi = torch.arange(b * hw) // hw
    warped = (w1[:, None] * img[bi, :, u0m, v0m] + w2[:, None] * img[bi, :, u1m, v0m] +
              w3[:, None] * img[bi, :, u0m, v1m] + w4[:, None] * img[bi, :, u1m, v1m])
    return warped, mask


def loss_torch(img, cam, img_rot, cam_rot, occlusion_aware=False, lambda_geometric=3.0, K=None,
               max_depth=None, min_depth=None, norm="l1"):
    """Differentiable restatement of LossFuncRotate.__call__ (loss_functions.py:63-146).

    img, img_rot: torch (b,4,S,S), may require grad; cam, cam_rot: numpy (b,4,4).
    """
    b, C, S, _ = img.shape
    Kmat, inv_K, p = intrinsics(S, K)
    R, t = relative_pose(np.asarray(cam, F32), np.asarray(cam_rot, F32))
    A, c, A2, c2 = (torch.from_numpy(a).to(img.dtype) for a in warp_coefficients(Kmat, inv_K, R, t))
    pt = torch.from_numpy(p).to(img.dtype)
    z = img[:, -1:].reshape(b, 1, -1)
    z_rot = img_rot[:, -1:].reshape(b, 1, -1)
    zp = (torch.matmul(A, z * pt) - c[:, :, None]).transpose(1, 2)
    zp_rot = (torch.matmul(A2, z_rot * pt) + c2[:, :, None]).transpose(1, 2)
    warped, mask = _bilinear_torch(img_rot, zp)
    warped_rot, mask_rot = _bilinear_torch(img, zp_rot)
    mf = mask.to(img.dtype)[:, None]
    mf_rot = mask_rot.to(img.dtype)[:, None]
    tgt = torch.cat([img[:, :-1].permute(0, 2, 3, 1).reshape(-1, C - 1), zp[:, :, 2].reshape(-1, 1)], 1) * mf
    tgt_rot = torch.cat([img_rot[:, :-1].permute(0, 2, 3, 1).reshape(-1, C - 1),
                         zp_rot[:, :, 2].reshape(-1, 1)], 1) * mf_rot
    if occlusion_aware:
        vis = (warped[:, -1:].detach() > zp[:, :, 2].reshape(-1, 1).detach()).to(img.dtype)
        vis_rot = (warped_rot[:, -1:].detach() > zp_rot[:, :, 2].reshape(-1, 1).detach()).to(img.dtype)
        warped, warped_rot, tgt, tgt_rot = warped * vis, warped_rot * vis_rot, tgt * vis, tgt_rot * vis_rot
    if max_depth is not None:
        sd = (z.detach().transpose(1, 2).reshape(-1, 1) < max_depth).to(img.dtype)
        sdr = (z_rot.detach().transpose(1, 2).reshape(-1, 1) < max_depth).to(img.dtype)
        warped, tgt, warped_rot, tgt_rot = warped * sd, tgt * sd, warped_rot * sdr, tgt_rot * sdr
    if min_depth is not None:
        ld = (z.detach().transpose(1, 2).reshape(-1, 1) > min_depth).to(img.dtype)
        ldr = (z_rot.detach().transpose(1, 2).reshape(-1, 1) > min_depth).to(img.dtype)
        warped, tgt, warped_rot, tgt_rot = warped * ld, tgt * ld, warped_rot * ldr, tgt_rot * ldr

    def mae(a, bb):          # loss_functions.py:137-140: F.mean_absolute_error for norm == "l1", else F.mean_squared_error
        if norm != "l1":
            return ((a - bb) ** 2).sum() / a.numel()
        return (a - bb).abs().sum() / a.numel()

    loss = mae(warped[:, :-1], tgt[:, :-1]) + mae(warped_rot[:, :-1], tgt_rot[:, :-1])
    loss = loss + mae(warped[:, -1], tgt[:, -1]) * lambda_geometric + \
        mae(warped_rot[:, -1], tgt_rot[:, -1]) * lambda_geometric
    return loss, torch.cat([zp, zp_rot], 0)
