#!/bin/bash
# rocprofv3 kernel statistics of the secondary paths (segmented scan, scans with actions, long-string counting, SlowScanner)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/extra
LONG_NS=1 LONG_TOTAL_LOG2=30 PYTHONPATH=. timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/extra/long -o p -- python tools/long_case.py > gpurun_out/extra/long.log 2>&1
PYTHONPATH=. timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/extra/actions -o p -- python tools/actions_case.py > gpurun_out/extra/actions.log 2>&1
PYTHONPATH=. timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/extra/longhf -o p -- python tools/long_half_final.py > gpurun_out/extra/longhf.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/extra/slow -o p -- python bench.py --set slow_x40_utf8 --log2-strings 18 --len 16384 --steps 5 --warmup 1 --no-cpu > gpurun_out/extra/slow.log 2>&1
for d in long actions longhf slow; do echo "== $d"; f=$(find gpurun_out/extra/$d -name "p_kernel_stats.csv" | head -1); head -8 "$f" | cut -c1-200; done
