#!/bin/bash
set -u
export PYTHONUNBUFFERED=1 TMPDIR=/tmp PYTHONPATH=.
OUT=gpurun_out/r02exp4
mkdir -p $OUT
P='import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(sys.argv[1], d["value"], r["kernel"], "avg", r["kernel_avg_ms"], "min", r["kernel_min_ms"], "frac", r["frac"], "cold", d.get("value_before_adapt"), d["match_counts"]["final"], d["config"]["table"]["lds_dense_rows"])'
env PIRE_HIP_MAX_HOT=112 PIRE_HIP_LIB=tools/ab/libpire_hip_tuning.so PIRE_HIP_DEBUG_LAUNCH=1 PIRE_HIP_TILED_VARIANT=11 timeout 300 python bench.py --no-cpu --steps 2 --warmup 1 2>&1 | grep "pire_hip:" | sort | uniq -c | tee $OUT/launch.log
for rep in 1 2; do
for v in 0 11 12; do
  env PIRE_HIP_MAX_HOT=112 PIRE_HIP_TILED_VARIANT=$v timeout 300 python bench.py --no-cpu --steps 30 --warmup 10 2>&1 | tail -1 | python -c "$P" "set_a hot112 v$v" | tee -a $OUT/variants.log
  env PIRE_HIP_TILED_VARIANT=$v timeout 300 python bench.py --set c2_single --no-cpu --steps 30 --warmup 10 2>&1 | tail -1 | python -c "$P" "c2 v$v" | tee -a $OUT/variants.log
done
done
