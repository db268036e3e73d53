mkdir -p gpurun_out/r3ae
python scripts/torch_op_sources.py 2>&1 | grep -v "amdgpu.ids\|Warning\|warn\|return Variable" | tee gpurun_out/r3ae/torch_ops.log | tail -60
