"""DeepVoxels frustum path of the reference (config 4), restated for the CPU.

Test infrastructure only (see oracle/__init__.py).  PARITY UNPINNED.

* ``proj_idcs_np``      deepvoxel/projection.py:48-105 (compute_proj_idcs): frustum -> voxel coordinates, compaction
* ``trilinear_torch``   deepvoxel/deepvoxel.py:388-428 (interpolate_trilinear), differentiable
* ``occlusion_torch``   deepvoxel/deepvoxel.py:574-587 (AccumulativeOcclusionNet.forward) + the compositing of
                        DeepVoxels.forward :886-889 and the depth rescale :903-904, differentiable

Evaluation order for the bit-exact index math: every operation in float32 (the Chainer-era NumPy /
CuPy value-based casting keeps float32 arrays float32 against Python / float64 scalars), left to right, unfused:
    zc = float(n // (W*H)) * f32(voxel) + f32(near);   tmp = n - int(float(n // (W*H)) * W * H)
    yc = (float(tmp) / W - cy) / fy * zc  (tmp / W is a TRUE division: fractional row coordinate);   xc likewise
    g_k = ((C_k0*xc + C_k1*yc) + C_k2*zc) + C_k3;     v_k = g_k / f32(voxel) + grid/2
"""
import numpy as np
import torch

F32 = np.float32


class Frustum:
    """Constants of deepvoxels_generator.py:230-253."""

    def __init__(self, grid_dim=32, img=64, scale=0.5):
        self.grid_dim = grid_dim
        self.W = self.H = img
        self.near_plane = np.sqrt(3) / 4
        self.voxel_size = (1. / grid_dim) * 1.1 * scale
        self.depth = int(np.ceil(np.sqrt(3) * grid_dim))
        self.fx = self.fy = img * 2.0
        self.cx = self.cy = img / 2.0
        self.n = self.W * self.H * self.depth


def proj_idcs_np(cam2world, fr=None):
    """Returns (lin_ind int32 (M,), voxel_coords float32 (3,M)) or None when nothing is inside the grid."""
    fr = fr or Frustum()
    C = np.asarray(cam2world, F32)
    n = np.arange(0, fr.n).astype("int32")
    zc = (n // (fr.W * fr.H)).astype(F32)
    tmp = n - ((zc * F32(fr.W)) * F32(fr.H)).astype("int32")
    yc = (tmp.astype(F32) / F32(fr.W)).astype(F32)          # true division, exact (W is a power of two)
    xc = (tmp % fr.W).astype(F32)
    zc = (zc * F32(fr.voxel_size)).astype(F32)
    zc = (zc + F32(fr.near_plane)).astype(F32)
    xc = ((xc - F32(fr.cx)) / F32(fr.fx)).astype(F32)
    yc = ((yc - F32(fr.cy)) / F32(fr.fy)).astype(F32)
    xc = (xc * zc).astype(F32)
    yc = (yc * zc).astype(F32)
    v = []
    for k in range(3):
        g = ((C[k, 0] * xc + C[k, 1] * yc) + C[k, 2] * zc) + C[k, 3]
        v.append((g.astype(F32) / F32(fr.voxel_size) + F32(fr.grid_dim / 2)).astype(F32))
    v = np.stack(v, 0)
    mask = np.all(v >= 0, axis=0) & (v[0] < fr.grid_dim) & (v[1] < fr.grid_dim) & (v[2] < fr.grid_dim)
    if not mask.any():
        return None
    return n[mask], v[:, mask]


def trilinear_torch(grid, lin_ind, voxel_coords, fr=None):
    """grid (1,F,G,G,G) torch; lin_ind (M,), voxel_coords (3,M) numpy -> (1,F,depth,H,W)."""
    fr = fr or Frustum()
    _, Fch, Hh, Ww, Dd = grid.shape
    vc = torch.as_tensor(voxel_coords)
    xi, yi, zi = vc[2], vc[1], vc[0]
    x0, y0, z0 = xi.to(torch.int64), yi.to(torch.int64), zi.to(torch.int64)   # truncation
    x1 = torch.clamp(x0 + 1, 0, Ww - 1)
    y1 = torch.clamp(y0 + 1, 0, Hh - 1)
    z1 = torch.clamp(z0 + 1, 0, Dd - 1)
    x, y, z = xi - x0, yi - y0, zi - z0
    g = grid
    added = (g[:, :, x0, y0, z0] * (1 - x) * (1 - y) * (1 - z) + g[:, :, x1, y0, z0] * x * (1 - y) * (1 - z) +
             g[:, :, x0, y1, z0] * (1 - x) * y * (1 - z) + g[:, :, x0, y0, z1] * (1 - x) * (1 - y) * z +
             g[:, :, x1, y0, z1] * x * (1 - y) * z + g[:, :, x0, y1, z1] * (1 - x) * y * z +
             g[:, :, x1, y1, z0] * x * y * (1 - z) + g[:, :, x1, y1, z1] * x * y * z)
    out = torch.zeros(1, Fch, fr.n, dtype=grid.dtype)
    idx = torch.as_tensor(lin_ind).to(torch.int64)
    out = out.index_add(2, idx, added)
    return out.reshape(1, Fch, fr.depth, fr.H, fr.W)


def depth_coords(fr):
    """deepvoxel.py:568-569."""
    return (np.arange(-fr.depth // 2, fr.depth // 2) / fr.depth).astype("float32")


def occlusion_torch(vol, W1, b1, W2, b2, fr=None, threshold=4.0):
    """vol (1,F,D,H,W); W1 (nf,F+1) b1 (nf,) W2 (1,nf) b2 (1,) = the two 1x1x1 equalized convs
    (inv_c = sqrt(2/in_ch)).  Returns (features (1,F,H,W), depth (1,1,H,W) rescaled, weights (1,1,D,H,W))."""
    fr = fr or Frustum()
    Fch = vol.shape[1]
    dc = torch.from_numpy(depth_coords(fr)).to(vol.dtype)
    dvol = dc.reshape(1, 1, fr.depth, 1, 1).expand(1, 1, fr.depth, fr.H, fr.W)
    x = torch.cat([dvol, vol], dim=1)                                           # depth coordinate is channel 0
    c1 = float(np.sqrt(2.0 / (Fch + 1)))
    h = torch.einsum("oc,bcdhw->bodhw", W1, x * c1) + b1.reshape(1, -1, 1, 1, 1)
    h = torch.nn.functional.leaky_relu(h, 0.2)
    c2 = float(np.sqrt(2.0 / W1.shape[0]))
    s = torch.einsum("oc,bcdhw->bodhw", W2, h * c2) + b2.reshape(1, -1, 1, 1, 1)
    s = torch.sigmoid(s - threshold)
    cs = torch.clamp(torch.cumsum(s, dim=2), 0, 1)
    cs = torch.cat([torch.zeros_like(cs[:, :, :1]), cs], dim=2)
    w = cs[:, :, 1:] - cs[:, :, :-1]
    depth = torch.sum(dvol * w, dim=2)
    feat = torch.sum(w * vol, dim=2)
    depth = (depth + 0.5) * fr.depth * fr.voxel_size + fr.near_plane           # deepvoxel.py:903-904
    return feat, depth, w
