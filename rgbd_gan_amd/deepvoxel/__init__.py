"""DeepVoxels frustum path (reference deepvoxel/projection.py, deepvoxel/deepvoxel.py) on the HIP kernels."""
