#!/bin/bash
# Collect the round's evidence on the GPU box (run through gpurun): per-kernel time of the default bench command, HBM
# traffic counters (separate --pmc passes, no trace domains other than --kernel-trace) and SQ / LDS / L2 counters of the hot
# conv kernels on the step's layer shapes; summaries into profiles/$1 (stamped with the kernel sources' hash).
#   scripts/collect_profiles.sh r02 [commit]
set -u
R=${1:-r06}
export RGBD_COMMIT=${2:-unknown}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/$R profiles/$R
# (--no-tune under the profiler: the 120 measuring steps of the side-budget tuner -- eager steps, re-captures, other workgroup counts --
#  would be averaged into the per-kernel table; the profiled step uses the rule of thumb's counts)
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$R/stats -o b -- python3 bench.py --no-tune --no-cpu-baseline --no-other-configs > gpurun_out/$R/stats.log 2>&1
cp gpurun_out/$R/stats/b_kernel_stats.csv profiles/$R/bench_kernel_stats.csv
grep '"metric"' gpurun_out/$R/stats.log > profiles/$R/bench_line_under_rocprof.json
rm -f gpurun_out/$R/stats/b_kernel_trace.csv
# HBM traffic per kernel, ONE profile per workload (bench.py:workload_key): bench.py quotes a profile only on the workload it was
# recorded on
pmc_traffic() {   # name, bench.py arguments
  local W=$1; shift
  for C in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $C --output-format csv -d gpurun_out/$R/pmc_${W}_$C -o p -- python3 bench.py "$@" --no-tune --steps 3 --warmup 3 --no-cpu-baseline --no-roofline --no-other-configs > gpurun_out/$R/pmc_${W}_$C.log 2>&1
  done
  python3 scripts/pmc_summary.py gpurun_out/$R/pmc_${W}_FETCH_SIZE gpurun_out/$R/pmc_${W}_WRITE_SIZE profiles/$R/bench_pmc_traffic_$W.json \
    "rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE (separate passes) -- python3 bench.py $* --no-tune --steps 3 --warmup 3 --no-cpu-baseline --no-roofline --no-other-configs" $W
  rm -rf gpurun_out/$R/pmc_${W}_FETCH_SIZE gpurun_out/$R/pmc_${W}_WRITE_SIZE
}
pmc_traffic default
pmc_traffic c2_fade --stage 9.5
pmc_traffic res256 --res256
pmc_traffic res256_fp8 --res256 --fp8
pmc_traffic c3_b8 --config configs/ffhq_stylegan_occlusion.yml --batch 8
pmc_traffic c4 --config configs/deepvoxels_shapenet_car.yml
# SQ / LDS / L2 counters of the conv kernels alone (scripts/prof_conv.py: the step's layer shapes at B=32), a few counters per pass
P=0
for CS in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" \
          "SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU" \
          "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE" "SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_MFMA" "FETCH_SIZE" "WRITE_SIZE"; do
  P=$((P+1))
  REPS=3 rocprofv3 --kernel-trace --pmc $CS --output-format csv -d gpurun_out/$R/kpmc_$P -o k -- python3 scripts/prof_conv.py > gpurun_out/$R/kpmc_$P.log 2>&1
done
python3 scripts/pmc_kernels.py profiles/$R gpurun_out/$R/kpmc_*
rm -rf gpurun_out/$R/kpmc_*/
# ---- BASELINE configuration 5 (256x256, ch = 512, per-GPU batch 16): the MXFP8 line and its bf16 twin on the same box, the
#      per-kernel time of the fp8 command, and the counters of the MXFP8 kernel beside its bf16 twin (scripts/prof_conv_mx8.py)
python3 bench.py --res256 --fp8 --no-cpu-baseline > profiles/$R/bench_res256_fp8.json 2> gpurun_out/$R/res256_fp8.err
python3 bench.py --res256 --no-cpu-baseline > profiles/$R/bench_res256_bf16.json 2> gpurun_out/$R/res256_bf16.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$R/stats256 -o b -- python3 bench.py --res256 --fp8 --no-tune --steps 40 --no-cpu-baseline --no-roofline > gpurun_out/$R/stats256.log 2>&1
cp gpurun_out/$R/stats256/b_kernel_stats.csv profiles/$R/bench_res256_fp8_kernel_stats.csv
rm -f gpurun_out/$R/stats256/b_kernel_trace.csv
P=0
for CS in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" \
          "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  P=$((P+1))
  REPS=3 rocprofv3 --kernel-trace --pmc $CS --output-format csv -d gpurun_out/$R/mpmc_$P -o k -- python3 scripts/prof_conv_mx8.py > gpurun_out/$R/mpmc_$P.log 2>&1
done
python3 scripts/pmc_kernels.py profiles/$R/mx8 gpurun_out/$R/mpmc_*
rm -rf gpurun_out/$R/mpmc_*/
# ---- the other configurations' lines: configuration 3's per-GPU shape (batch 64 over 8 GPUs) and configuration 4
python3 bench.py --config configs/ffhq_stylegan_occlusion.yml --batch 8 --no-cpu-baseline > profiles/$R/bench_c3_b8.json 2> gpurun_out/$R/c3.err
python3 bench.py --config configs/deepvoxels_shapenet_car.yml --no-cpu-baseline > profiles/$R/bench_c4.json 2> gpurun_out/$R/c4.err
RGBD_CONCURRENT_PHASES=0 python3 bench.py --config configs/deepvoxels_shapenet_car.yml --no-cpu-baseline --no-roofline > profiles/$R/bench_c4_one_stream.json 2> gpurun_out/$R/c4_1s.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$R/stats_c4 -o b -- python3 bench.py --config configs/deepvoxels_shapenet_car.yml --steps 40 --no-cpu-baseline --no-roofline > gpurun_out/$R/stats_c4.log 2>&1
cp gpurun_out/$R/stats_c4/b_kernel_stats.csv profiles/$R/bench_c4_kernel_stats.csv
rm -f gpurun_out/$R/stats_c4/b_kernel_trace.csv
# ---- what lies under what: a kernel trace of the default command through scripts/trace_overlap.py (small kernels that the
#      per-kernel averages show at 5-10x their stand-alone time are stretched under the other queue's chip-filling kernels)
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/$R/trace -o t -- python3 bench.py --no-tune --steps 6 --warmup 10 --no-cpu-baseline --no-roofline --no-other-configs > gpurun_out/$R/trace.log 2>&1
T=$(find gpurun_out/$R/trace -name 't_kernel_trace.csv' | head -1)
python3 scripts/trace_overlap.py $T 'planes_outer_kernel<4>' 'from_planes_kernel<4>' 'linear_fwd' 'adain_reduce' 'warp_loss_bwd_kernel' > profiles/$R/trace_overlap.txt 2>&1
python3 scripts/timeline.py $T > profiles/$R/trace_timeline.txt 2>&1
rm -f $T
python3 scripts/time_planes.py > profiles/$R/time_planes_alone.txt 2>/dev/null
python3 scripts/time_small_ops.py > profiles/$R/time_small_ops_alone.txt 2>/dev/null
# (the figures of time_small_ops.py are host-paced launch pairs; the kernels' own durations of the same script under the tracer:)
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$R/smallops -o s -- python3 scripts/time_small_ops.py > gpurun_out/$R/smallops.log 2>&1
cp $(find gpurun_out/$R/smallops -name 's_kernel_stats.csv' | head -1) profiles/$R/time_small_ops_kernel_stats.csv
rm -rf gpurun_out/$R/smallops
# ---- the phases' timeline from events (no tracer), and which torch ops still launch kernels in a step (default config, config 4)
python3 scripts/phase_timeline.py > profiles/$R/phase_timeline.txt 2>/dev/null
python3 scripts/phases_alone.py > profiles/$R/phases_alone.txt 2>/dev/null
CONFIG=ffhq_stylegan_occlusion.yml B=8 python3 scripts/phase_timeline.py > profiles/$R/phase_timeline_c3_b8.txt 2>/dev/null
CONFIG=ffhq_stylegan_occlusion.yml B=8 python3 scripts/phases_alone.py > profiles/$R/phases_alone_c3_b8.txt 2>/dev/null
python3 scripts/step_conv_shapes.py 2>/dev/null | grep -v amdgpu.ids > profiles/$R/step_conv_shapes.txt
CONFIG=deepvoxels_shapenet_car.yml ITERATION=100 python3 scripts/step_conv_shapes.py 2>/dev/null | grep -v amdgpu.ids > profiles/$R/step_conv_shapes_c4.txt
# the two-stream step without the side stream's compute-unit budgets, same box (profiles/r05/cu_budget_sweep.txt)
RGBD_SIDE_CUS=0 RGBD_SIDE_WGRAD_WGS=0 python3 bench.py --no-tune --no-cpu-baseline --no-roofline --no-other-configs > profiles/$R/bench_default_no_cu_budget.json 2>/dev/null
# ... and with the rule of thumb instead of the measured workgroup counts
python3 bench.py --no-tune --no-cpu-baseline --no-roofline --no-other-configs > profiles/$R/bench_default_rule_of_thumb.json 2>/dev/null
python3 scripts/time_trilinear_bwd.py 2>/dev/null | grep -v amdgpu.ids > profiles/$R/time_trilinear_bwd.txt
python3 -m pytest tests -q -m gpu 2>&1 | tail -3 > profiles/$R/gpu_tests_same_box.txt
python3 scripts/torch_op_sources.py 2>/dev/null | grep -v amdgpu.ids > profiles/$R/torch_op_sources.txt
python3 scripts/torch_op_sources.py 200000 configs/deepvoxels_shapenet_car.yml 2>/dev/null | grep -v amdgpu.ids > profiles/$R/torch_op_sources_c4.txt
python3 bench.py > profiles/$R/bench_default.json 2> gpurun_out/$R/default.err
RGBD_DV_PREFETCH=0 python3 bench.py --config configs/deepvoxels_shapenet_car.yml --no-cpu-baseline --no-roofline > profiles/$R/bench_c4_no_early_forward.json 2>/dev/null
CONFIG=deepvoxels_shapenet_car.yml B=10 python3 scripts/phase_timeline.py > profiles/$R/phase_timeline_c4.txt 2>/dev/null
cp profiles/$R/*.csv profiles/$R/*.json profiles/$R/*.txt gpurun_out/$R/ 2>/dev/null
mkdir -p gpurun_out/$R/mx8 && cp profiles/$R/mx8/* gpurun_out/$R/mx8/ 2>/dev/null
ls -la profiles/$R
