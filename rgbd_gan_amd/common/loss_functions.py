"""Losses with the reference's surface (common/loss_functions.py), running on the HIP warp-loss kernels.

    loss_l2, loss_func_dcgan_gen, loss_func_dcgan_dis                      loss_functions.py:7-28
    LossFuncRotate(xp, K=None, norm="l1", lambda_geometric=3)              loss_functions.py:31-146
        __call__(img, theta, img_rot, theta_rot, occlusion_aware=False, debug=False, max_depth=None, min_depth=None)

The per-pair 3x3 / 3x1 constants (K R K^-1, K R t, K R^T K^-1, K t) are formed on the host in NumPy fp32 in the
reference's association order (loss_functions.py:174,181) -- they are 24 floats per view pair -- and everything per
pixel (projection, truncation, masks, gathers, L1 terms, and all of the backward) runs in
rgbd_gan_amd/csrc/warp_loss.hip.
"""
import numpy as np
import torch
import torch.nn.functional as F

from .. import functional as Fn
from .. import kernels

WARP_OCCLUSION, WARP_MAX_DEPTH, WARP_MIN_DEPTH = 1, 2, 4


def loss_l2(h, t):
    return torch.sum((h - t) ** 2) / h.numel()


def loss_func_dcgan_gen(y_fake, focal_loss_gamma=0.):
    if focal_loss_gamma is None:
        focal_loss_gamma = 0.
    if focal_loss_gamma == 0.:
        return torch.sum(F.softplus(-y_fake)) / y_fake.numel()
    return torch.sum(F.softplus(-y_fake) * torch.sigmoid(-y_fake) ** focal_loss_gamma) / y_fake.numel()


def loss_func_dcgan_dis(y_fake, y_real):
    if isinstance(y_fake, tuple):
        loss = 0
        for _f, _r in zip(y_fake, y_real):
            loss = loss + torch.sum(F.softplus(_f)) / _f.numel() + torch.sum(F.softplus(-_r)) / _r.numel()
        return loss
    return torch.sum(F.softplus(y_fake)) / y_fake.numel() + torch.sum(F.softplus(-y_real)) / y_real.numel()


def _to_numpy(a):
    if torch.is_tensor(a):
        return a.detach().cpu().numpy()
    return np.asarray(a)


class LossFuncRotate:
    def __init__(self, xp=None, K=None, norm="l1", lambda_geometric=3):
        self.xp = xp
        self.size = None
        self.K = K
        self.norm = norm
        self.lambda_geometric = lambda_geometric
        self.inv_K = None

    def init_params(self, xp=None, size=4):
        """loss_functions.py:39-61, state machine included: the first call builds K (or crops/rescales a given
        one); later size changes rescale rows 0-1 in place."""
        if self.size is None:
            if self.K is not None:
                self.K = np.array(np.asarray(_to_numpy(self.K))[:3, :3], "float32")
                self.K[:2] *= size / self.K[0, 2] / 2
            else:
                self.K = np.array([[size * 2, 0, size / 2], [0, size * 2, size / 2], [0, 0, 1]], dtype="float32")
            self.size = size
        else:
            self.size = size
            self.K[:2] *= size / self.K[0, 2] / 2
        self.inv_K = np.linalg.inv(self.K).astype("float32")

    def coefficients(self, theta, theta_rot):
        """(b,24) fp32: A, c, A', c' for warp / inv_warp (loss_functions.py:85-94,174,181)."""
        theta = _to_numpy(theta).astype("float32", copy=False)
        theta_rot = _to_numpy(theta_rot).astype("float32", copy=False)
        R1, R2 = theta[:, :3, :3], theta_rot[:, :3, :3]
        t1, t2 = theta[:, :3, -1:], theta_rot[:, :3, -1:]
        R = np.matmul(R2.transpose(0, 2, 1), R1).astype("float32")
        inv_R = R.transpose(0, 2, 1)
        t = np.matmul(R1.transpose(0, 2, 1), t2 - t1).astype("float32")
        KR = np.matmul(self.K, R)
        A = np.matmul(KR, self.inv_K)
        c = np.matmul(KR, t)
        A2 = np.matmul(np.matmul(self.K, inv_R), self.inv_K)
        c2 = np.matmul(self.K, t)
        b = len(theta)
        return np.concatenate([A.reshape(b, 9), c.reshape(b, 3), A2.reshape(b, 9), c2.reshape(b, 3)],
                              axis=1).astype("float32")

    def coefficients_for_size(self, size, theta, theta_rot):
        """Host half of __call__: (re)build the intrinsics for `size` like the reference does on a size change,
        then return the (b,24) warp constants.  Lets a caller upload them itself (HIP-graph replays)."""
        if self.size != size:
            self.init_params(self.xp, size=size)
        return self.coefficients(theta, theta_rot)

    def loss_from_coefficients(self, img, img_rot, coef, occlusion_aware=False, max_depth=None, min_depth=None):
        """Device half of __call__: the fused warp-loss kernels on constants already resident on the device."""
        flags = (WARP_OCCLUSION if occlusion_aware else 0) | (WARP_MAX_DEPTH if max_depth is not None else 0) | \
                (WARP_MIN_DEPTH if min_depth is not None else 0)
        return Fn.warp_loss(img, img_rot, coef, flags, self.lambda_geometric,
                            0.0 if max_depth is None else float(max_depth),
                            0.0 if min_depth is None else float(min_depth))[0]

    @staticmethod
    def _projected_points(img, img_rot, coef):
        """(2b, hw, 3): [new_zp, new_zp_rot] (loss_functions.py:94-95,146) from the depth channels and the warp constants
        -- the second return value for inputs the 4-channel kernels do not take; same association as the kernels
        ((A0 zp0 + A1 zp1) + A2 zp2) -/+ c, torch ops (not differentiable here: detached)."""
        with torch.no_grad():
            b, _, S, _ = img.shape
            jj = torch.arange(S, device=img.device, dtype=torch.float32).repeat(S)
            ii = torch.arange(S, device=img.device, dtype=torch.float32).repeat_interleave(S)
            out = []
            for x, off, sign in ((img, 0, -1.0), (img_rot, 12, 1.0)):
                z = x[:, -1].reshape(b, -1)
                a0, a1, a2 = z * jj, z * ii, z * 1.0
                cf = coef[:, off:off + 12]
                rows = []
                for k in range(3):
                    v = (cf[:, 3 * k:3 * k + 1] * a0 + cf[:, 3 * k + 1:3 * k + 2] * a1) + cf[:, 3 * k + 2:3 * k + 3] * a2
                    rows.append(v + sign * cf[:, 9 + k:10 + k])
                out.append(torch.stack(rows, dim=2))
            return torch.cat(out, dim=0)

    def __call__(self, img, theta, img_rot, theta_rot, occlusion_aware=False, debug=False, max_depth=None,
                 min_depth=None):
        if self.size != img.shape[-1]:
            self.init_params(self.xp, size=img.shape[-1])
        coef = torch.from_numpy(self.coefficients(theta, theta_rot)).to(img.device)
        flags = (WARP_OCCLUSION if occlusion_aware else 0) | (WARP_MAX_DEPTH if max_depth is not None else 0) | \
                (WARP_MIN_DEPTH if min_depth is not None else 0)
        mx = 0.0 if max_depth is None else float(max_depth)
        mn = 0.0 if min_depth is None else float(min_depth)
        if self.norm != "l1" or img.shape[1] != 4:
            # loss_functions.py:137-140 (criteria = F.mean_squared_error unless norm == "l1") and inputs other than RGB-D
            # (updater.py:345-354 feeds 257-channel features): the generic kernels, fp32 atomic scatter
            if debug:
                raise NotImplementedError("debug=True is served by the 4-channel kernels only")
            loss = Fn.warp_loss_nc(img, img_rot, coef, flags, self.norm != "l1", self.lambda_geometric, mx, mn)
            zp = self._projected_points(img, img_rot, coef)
            return loss, zp
        if debug:
            b, _, S, _ = img.shape
            _, zp, warped, idx = kernels.warp_loss_fwd(img.detach().contiguous(), img_rot.detach().contiguous(), coef,
                                                       flags, self.lambda_geometric, mx, mn, debug=True)
            mask = idx[..., 3].bool()
            return warped[0], mask[0], zp[0], warped[1], mask[1], zp[1]
        loss, zp = Fn.warp_loss(img, img_rot, coef, flags, self.lambda_geometric, mx, mn, want_zp=True)
        # loss_functions.py:146: F.concat([new_zp, new_zp_rot], axis=0) -> (2b, hw, 3); a by-product of the fused forward
        # here, detached (no caller of the reference differentiates it)
        return loss, zp.reshape(-1, zp.shape[2], 3)
