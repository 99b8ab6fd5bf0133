#!/bin/bash
set -u
export PYTHONUNBUFFERED=1 TMPDIR=/tmp PYTHONPATH=. PIRE_HIP_LIB=tools/ab/libpire_hip_tuning.so
OUT=gpurun_out/r02exp7
mkdir -p $OUT
P='import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(sys.argv[1], "avg", r["kernel_avg_ms"], "min", r["kernel_min_ms"], d["match_counts"]["final"])'
for rep in 1 2; do
for knobs in "X=1" "PIRE_HIP_DEBUG_TASKMAP=1" "PIRE_HIP_DEBUG_NOSTEP=1" "PIRE_HIP_DEBUG_NOSTEP=1 PIRE_HIP_DEBUG_TASKMAP=1"; do
    env $knobs timeout 300 python bench.py --no-cpu --steps 20 --warmup 5 2>&1 | tail -1 | python -c "$P" "$knobs" | tee -a $OUT/taskmap.log
done
done
