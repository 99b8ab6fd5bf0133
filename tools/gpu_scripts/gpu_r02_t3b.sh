#!/bin/bash
# three tiles per wave: loads only / walk only, against the two-slot kernel (tuning build)
set -u
export PYTHONUNBUFFERED=1 TMPDIR=/tmp PYTHONPATH=. PIRE_HIP_LIB=tools/ab/libpire_hip_tuning.so
P='import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(sys.argv[1], d["value"], "avg", r["kernel_avg_ms"], "min", r["kernel_min_ms"])'
for rep in 1 2; do
for v in 0 30; do
  env PIRE_HIP_TILED_VARIANT=$v PIRE_HIP_DEBUG_NOSTEP=1 timeout 300 python bench.py --no-cpu --no-adapt --steps 20 --warmup 5 2>&1 | tail -1 | python -c "$P" "loads only v$v"
  env PIRE_HIP_TILED_VARIANT=$v PIRE_HIP_DEBUG_NOLOAD=1 timeout 300 python bench.py --no-cpu --no-adapt --steps 20 --warmup 5 2>&1 | tail -1 | python -c "$P" "walk only v$v"
  env PIRE_HIP_TILED_VARIANT=$v timeout 300 python bench.py --no-cpu --steps 20 --warmup 5 2>&1 | tail -1 | python -c "$P" "whole v$v"
done
done
