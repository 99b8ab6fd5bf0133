#!/bin/bash
# the bench part of tools/gpu_final.sh: default bench line (with CPU baseline) + rocprofv3 kernel stats of the same command
set -u
export PYTHONUNBUFFERED=1
export TMPDIR=/tmp
mkdir -p gpurun_out/final
timeout 900 python bench.py 2>&1 | tail -1 | tee gpurun_out/final/bench_n1.json | cut -c1-400
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/final/stats -o stats -- python bench.py --no-cpu > gpurun_out/final/stats.log 2>&1
head -3 gpurun_out/final/stats/stats_kernel_stats.csv
tail -1 gpurun_out/final/stats.log | cut -c1-200
