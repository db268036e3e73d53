"""ProjectionHelper with the reference's constructor and compute_proj_idcs (deepvoxel/projection.py:5-105), plus a
batched form that keeps everything on the device.

The per-element index math (frustum element -> voxel coordinates -> in-grid mask -> ordered compaction) runs in
rgbd_gan_amd/csrc/deepvoxels.hip for the WHOLE batch in three launches; the reference loops over the batch in Python
(deepvoxels_generator.py:288-289).
"""
import ctypes

import numpy as np
import torch

from .. import _lib
from ..kernels import _ptr, _stream


class ProjectionHelper:
    def __init__(self, lifting_intrinsic, projection_intrinsic, projection_image_dims, lifting_image_dims, depth_min,
                 depth_max, grid_dims, voxel_size, near_plane, frustrum_depth, device=None):
        self.grid_dims = grid_dims
        self.projection_intrinsic = np.asarray(projection_intrinsic)
        self.lifting_intrinsic = np.asarray(lifting_intrinsic)
        self.depth_min, self.depth_max = depth_min, depth_max
        self.projection_image_dims = projection_image_dims
        self.lifting_image_dims = lifting_image_dims
        self.voxel_size = voxel_size
        self.device = torch.device(device or "cuda:0")
        self.near_plane = near_plane
        self.frustrum_depth = int(frustrum_depth)
        assert grid_dims[0] == grid_dims[1] == grid_dims[2]

    @property
    def num_frust_elements(self):
        return self.projection_image_dims[0] * self.projection_image_dims[1] * self.frustrum_depth

    def frustum(self, cam2world):
        """What the frustum kernels that work straight from the cameras need (rgbd_trilinear_{fwd,bwd}_frustum): the (B,16) camera
        matrices on the device + the frustum's constants -- no index list, no compaction."""
        cams = torch.as_tensor(np.asarray(cam2world, dtype="float32") if not torch.is_tensor(cam2world) else cam2world)
        cams = cams.to(self.device, torch.float32).reshape(-1, 16).contiguous()
        K = self.projection_intrinsic
        W, H = self.projection_image_dims[0], self.projection_image_dims[1]
        return (cams, int(W), int(H), int(self.frustrum_depth), int(self.grid_dims[2]), float(self.voxel_size),
                float(self.near_plane), float(K[0][0]), float(K[1][1]), float(K[0][2]), float(K[1][2]))

    def compute_proj_idcs_batch(self, cam2world):
        """cam2world (B,4,4) -> idx (B,N) int32, coords (B,3,N) fp32 (compacted in order), counts (B,) int32."""
        cams = torch.as_tensor(np.asarray(cam2world, dtype="float32") if not torch.is_tensor(cam2world) else cam2world)
        cams = cams.to(self.device, torch.float32).reshape(-1, 16).contiguous()
        B, N = cams.shape[0], self.num_frust_elements
        W, H = self.projection_image_dims[0], self.projection_image_dims[1]
        idx = torch.empty(B, N, dtype=torch.int32, device=self.device)
        coords = torch.empty(B, 3, N, dtype=torch.float32, device=self.device)
        counts = torch.empty(B, dtype=torch.int32, device=self.device)
        ws = torch.empty(B * ((N + 255) // 256), dtype=torch.int32, device=self.device)
        K = self.projection_intrinsic
        rc = _lib.load().rgbd_proj_idcs(_ptr(cams), B, W, H, self.frustrum_depth, self.grid_dims[2],
                                        float(self.voxel_size), float(self.near_plane), float(K[0][0]), float(K[1][1]),
                                        float(K[0][2]), float(K[1][2]), _ptr(idx), _ptr(coords), _ptr(counts), _ptr(ws),
                                        _stream())
        _lib.check(rc, "rgbd_proj_idcs")
        # what the index list was computed FROM travels with it: the backward of the resampling recomputes the voxel coordinates
        # brick by brick from the cameras instead of walking the compacted list (deepvoxel._TrilinearFM.backward)
        idx._frustum = (cams, int(W), int(H), int(self.frustrum_depth), int(self.grid_dims[2]), float(self.voxel_size),
                        float(self.near_plane), float(K[0][0]), float(K[1][1]), float(K[0][2]), float(K[1][2]))
        return idx, coords, counts

    def compute_proj_idcs(self, cam2world, grid2world=None):
        """Reference signature: one (4,4) camera -> (lin_ind_frustrum (M,), voxel_coords (3,M)) or None."""
        if grid2world is not None:
            raise NotImplementedError("grid2world is never passed by the reference's training path")
        idx, coords, counts = self.compute_proj_idcs_batch(np.asarray(cam2world, dtype="float32")[None])
        m = int(counts[0].item())
        if m == 0:
            print('error: nothing in frustum bounds')
            return None
        return idx[0, :m], coords[0, :, :m]
