#!/bin/bash
set -u
export PYTHONUNBUFFERED=1 TMPDIR=/tmp PYTHONPATH=.
OUT=gpurun_out/r02eq
mkdir -p $OUT
P='import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(sys.argv[1], d["value"], "avg", r["kernel_avg_ms"], "min", r["kernel_min_ms"], "frac", r["frac"], d["match_counts"]["final"])'
for rep in 1 2 3 4 5; do
for v in 0 15; do
  env PIRE_HIP_TILED_VARIANT=$v timeout 300 python bench.py --no-cpu --steps 20 --warmup 5 2>&1 | tail -1 | python -c "$P" "2^20 v$v" | tee -a $OUT/ab.log
done
done
env PIRE_HIP_LIB=tools/ab/libpire_hip_tuning.so PIRE_HIP_DEBUG_STAMPS=1 PIRE_HIP_TILED_VARIANT=15 timeout 300 python bench.py --no-cpu --steps 4 --warmup 2 2>&1 | grep "pire_hip stamps" | tail -3 | tee -a $OUT/ab.log
