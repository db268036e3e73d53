mkdir -p gpurun_out/r3ai
O=gpurun_out/r3ai
python -c "import torch; torch.zeros(1).cuda()" 2>/dev/null
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "small_image or conv_fprop_matches or gather" 2>&1 | tail -12 | tee $O/tests.log
python scripts/time_small_conv.py 2>&1 | grep -v amdgpu | tee $O/time.log
