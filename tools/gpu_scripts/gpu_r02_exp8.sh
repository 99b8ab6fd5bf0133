#!/bin/bash
set -u
export PYTHONUNBUFFERED=1 TMPDIR=/tmp PYTHONPATH=.
OUT=gpurun_out/r02exp8
mkdir -p $OUT
P='import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(sys.argv[1], d["value"], r["kernel"], "avg", r["kernel_avg_ms"], "min", r["kernel_min_ms"], "frac", r["frac"], d["match_counts"]["final"], d.get("cpu_baseline",{}).get("parity_vs_gpu"))'
PIRE_HIP_TILED_VARIANT=3 timeout 300 python bench.py --steps 20 --warmup 5 --cpu-sample-log2 16 2>&1 | tail -1 | python -c "$P" "v3 parity" | tee -a $OUT/trmap.log
for rep in 1 2 3; do
for v in 0 3; do
  env PIRE_HIP_TILED_VARIANT=$v timeout 300 python bench.py --no-cpu --steps 30 --warmup 10 2>&1 | tail -1 | python -c "$P" "v$v" | tee -a $OUT/trmap.log
done
done
for v in 0 3; do
  env PIRE_HIP_LIB=tools/ab/libpire_hip_tuning.so PIRE_HIP_DEBUG_NOSTEP=1 PIRE_HIP_TILED_VARIANT=$v timeout 300 python bench.py --no-cpu --steps 30 --warmup 10 2>&1 | tail -1 | python -c "$P" "loads only v$v" | tee -a $OUT/trmap.log
done
