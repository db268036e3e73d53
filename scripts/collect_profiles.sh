#!/bin/bash
# Collect the round's evidence on the GPU box (run through gpurun): per-kernel time of the default bench command and
# HBM traffic counters (separate --pmc passes, no trace domains other than --kernel-trace), summaries into profiles/$1.
set -u
R=${1:-r01}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/$R profiles/$R
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$R/stats -o b -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/$R/stats.log 2>&1
cp gpurun_out/$R/stats/b_kernel_stats.csv profiles/$R/bench_kernel_stats.csv
grep '"metric"' gpurun_out/$R/stats.log > profiles/$R/bench_line_under_rocprof.json
rm -f gpurun_out/$R/stats/b_kernel_trace.csv
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d gpurun_out/$R/pmc_$C -o p -- python3 bench.py --steps 3 --warmup 3 --no-cpu-baseline --no-roofline > gpurun_out/$R/pmc_$C.log 2>&1
done
python3 scripts/pmc_summary.py gpurun_out/$R/pmc_FETCH_SIZE gpurun_out/$R/pmc_WRITE_SIZE profiles/$R/bench_pmc_traffic.json \
  "rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE (separate passes) -- python3 bench.py --steps 3 --warmup 3 --no-cpu-baseline --no-roofline"
rm -rf gpurun_out/$R/pmc_FETCH_SIZE gpurun_out/$R/pmc_WRITE_SIZE
cp profiles/$R/*.csv profiles/$R/*.json gpurun_out/$R/ 2>/dev/null
ls -la profiles/$R
