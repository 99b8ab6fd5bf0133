#!/bin/bash
# the two-chains-per-lane tiled kernel of commit b8c697b (tools/ab/libpire_hip_twochain.so) against its own one-chain kernel
set -u
export PYTHONUNBUFFERED=1 TMPDIR=/tmp PYTHONPATH=. PIRE_HIP_LIB=tools/ab/libpire_hip_twochain.so
OUT=gpurun_out/r02twochain
mkdir -p $OUT
P='import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(sys.argv[1], d["value"], r["kernel"], "avg", r["kernel_avg_ms"], "min", r["kernel_min_ms"], "frac", r["frac"], d["match_counts"]["final"])'
for rep in 1 2 3; do
for v in 0 3 4; do
  env PIRE_HIP_TILED_VARIANT=$v timeout 300 python bench.py --no-cpu --steps 20 --warmup 5 2>&1 | tail -1 | python -c "$P" "2^20 v$v" | tee -a $OUT/ab.log
done
done
for v in 0 3; do
  env PIRE_HIP_TILED_VARIANT=$v timeout 300 python bench.py --no-cpu --steps 20 --warmup 5 --strings 1179648 2>&1 | tail -1 | python -c "$P" "1179648 strings (3 full rounds of 24 chains) v$v" | tee -a $OUT/ab.log
  env PIRE_HIP_TILED_VARIANT=$v timeout 300 python bench.py --no-cpu --steps 5 --warmup 2 --log2-strings 23 2>&1 | tail -1 | python -c "$P" "2^23 v$v" | tee -a $OUT/ab.log
  env PIRE_HIP_TILED_VARIANT=$v timeout 300 python bench.py --set c2_single --no-cpu --steps 20 --warmup 5 2>&1 | tail -1 | python -c "$P" "c2 v$v" | tee -a $OUT/ab.log
done
