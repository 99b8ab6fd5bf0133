#!/bin/bash
# Round 2, ablation C (tuning build): walk only (stale registers, no loads) and stream only (loads + transpose, no walk)
# for the one-chain and the two-chain tiled kernels.
set -u
export PYTHONUNBUFFERED=1 TMPDIR=/tmp PIRE_HIP_LIB=tools/ab/libpire_hip_tuning.so
OUT=gpurun_out/r02c
mkdir -p $OUT
P='import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], d["roofline"]["kernel"], "avg", d["roofline"]["kernel_avg_ms"], "min", d["roofline"]["kernel_min_ms"])'
for v in 3 0 4; do
  for knob in none PIRE_HIP_DEBUG_NOLOAD PIRE_HIP_DEBUG_NOSTEP PIRE_HIP_DEBUG_NOHIST; do
    env PIRE_HIP_TILED_VARIANT=$v $knob=1 timeout 300 python bench.py --no-cpu --no-adapt --steps 20 --warmup 5 2>&1 | tail -1 | python -c "$P" "v$v $knob" | tee -a $OUT/ablation.log
  done
done
