#!/bin/bash
# Round 2, experiment: two 10-wave blocks per CU (20 chains per CU) with the dense rows inside 64 KiB of LDS per block
set -u
export PYTHONUNBUFFERED=1 TMPDIR=/tmp PYTHONPATH=.
OUT=gpurun_out/r02exp2
mkdir -p $OUT
P='import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(sys.argv[1], d["value"], r["kernel"], "avg", r["kernel_avg_ms"], "min", r["kernel_min_ms"], "frac", r["frac"], "cold", d.get("value_before_adapt"), d["match_counts"]["final"], d["config"]["table"]["lds_dense_rows"])'
for rep in 1 2; do
for v in 0 9; do
  env PIRE_HIP_MAX_HOT=246 PIRE_HIP_TILED_VARIANT=$v timeout 300 python bench.py --no-cpu --steps 30 --warmup 10 2>&1 | tail -1 | python -c "$P" "set_a hot246 v$v" | tee -a $OUT/variants.log
  env PIRE_HIP_TILED_VARIANT=$v timeout 300 python bench.py --set c2_single --no-cpu --steps 30 --warmup 10 2>&1 | tail -1 | python -c "$P" "c2 v$v" | tee -a $OUT/variants.log
done
done
env PIRE_HIP_MAX_HOT=120 PIRE_HIP_TILED_VARIANT=9 timeout 300 python bench.py --no-cpu --steps 30 --warmup 10 2>&1 | tail -1 | python -c "$P" "set_a hot120 v9" | tee -a $OUT/variants.log
env PIRE_HIP_MAX_HOT=120 PIRE_HIP_TILED_VARIANT=0 timeout 300 python bench.py --no-cpu --steps 30 --warmup 10 2>&1 | tail -1 | python -c "$P" "set_a hot120 v0" | tee -a $OUT/variants.log
