#!/bin/bash
set -u
export PYTHONUNBUFFERED=1 TMPDIR=/tmp PYTHONPATH=.
OUT=gpurun_out/r02prog2
mkdir -p $OUT
P='import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(sys.argv[1], d["value"], "avg", r["kernel_avg_ms"], "min", r["kernel_min_ms"], "frac", r["frac"])'
for rep in 1 2 3 4 5 6 7 8; do
for v in 0 14; do
  env PIRE_HIP_TILED_VARIANT=$v timeout 300 python bench.py --no-cpu --steps 20 --warmup 5 2>&1 | tail -1 | python -c "$P" "2^20 v$v" | tee -a $OUT/ab.log
done
done
for rep in 1 2 3; do
for v in 0 14; do
  env PIRE_HIP_TILED_VARIANT=$v timeout 300 python bench.py --no-cpu --steps 10 --warmup 3 --log2-strings 21 2>&1 | tail -1 | python -c "$P" "2^21 v$v" | tee -a $OUT/ab.log
  env PIRE_HIP_TILED_VARIANT=$v timeout 300 python bench.py --no-cpu --steps 5 --warmup 2 --log2-strings 23 2>&1 | tail -1 | python -c "$P" "2^23 v$v" | tee -a $OUT/ab.log
  env PIRE_HIP_TILED_VARIANT=$v timeout 300 python bench.py --set set_b --len 16384 --log2-strings 18 --no-cpu --steps 10 --warmup 3 2>&1 | tail -1 | python -c "$P" "set_b 2^18x16K v$v" | tee -a $OUT/ab.log
  env PIRE_HIP_TILED_VARIANT=$v timeout 300 python bench.py --corpus cxx --no-cpu --steps 20 --warmup 5 2>&1 | tail -1 | python -c "$P" "cxx v$v" | tee -a $OUT/ab.log
done
done
