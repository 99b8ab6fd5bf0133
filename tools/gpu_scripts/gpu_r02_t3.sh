#!/bin/bash
# tiled kernel with three tiles per wave (two AGPR landing slots), variant 30, against the default; alternating, one box
set -u
export PYTHONUNBUFFERED=1 TMPDIR=/tmp PYTHONPATH=.
OUT=gpurun_out/r02t3
mkdir -p $OUT
PIRE_HIP_TILED_VARIANT=30 timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x 2>&1 | tail -3 | tee $OUT/parity.log
P='import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(sys.argv[1], d["value"], "avg", r["kernel_avg_ms"], "min", r["kernel_min_ms"], "frac", r["frac"], d["match_counts"]["final"])'
for rep in 1 2 3; do
for v in 0 30; do
  env PIRE_HIP_TILED_VARIANT=$v timeout 300 python bench.py --no-cpu --steps 20 --warmup 5 2>&1 | tail -1 | python -c "$P" "C3 v$v" | tee -a $OUT/ab.log
  env PIRE_HIP_TILED_VARIANT=$v timeout 300 python bench.py --set c2_single --no-cpu --steps 20 --warmup 5 2>&1 | tail -1 | python -c "$P" "c2 v$v" | tee -a $OUT/ab.log
  env PIRE_HIP_TILED_VARIANT=$v timeout 300 python bench.py --corpus cxx --no-cpu --steps 20 --warmup 5 2>&1 | tail -1 | python -c "$P" "cxx v$v" | tee -a $OUT/ab.log
done
done
