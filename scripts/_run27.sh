mkdir -p gpurun_out/r3aa
RGBD_CONCURRENT_PHASES=0 python scripts/step_conv_shapes.py 2>&1 | grep -v "amdgpu.ids\|Warning\|warn\|return Variable" | tee gpurun_out/r3aa/shapes.log
