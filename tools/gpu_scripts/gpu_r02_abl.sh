#!/bin/bash
# Round 2, ablation of the tiled kernel (tuning build: results are WRONG by design, only times matter)
set -u
export PYTHONUNBUFFERED=1 TMPDIR=/tmp PYTHONPATH=. PIRE_HIP_LIB=tools/ab/libpire_hip_tuning.so
OUT=gpurun_out/r02abl
mkdir -p $OUT
P='import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(sys.argv[1], "avg", r["kernel_avg_ms"], "min", r["kernel_min_ms"])'
for knobs in "X=1" "PIRE_HIP_DEBUG_NOLOAD=1" "PIRE_HIP_DEBUG_NOSTEP=1" "PIRE_HIP_DEBUG_NOLOAD=1 PIRE_HIP_DEBUG_NOTRANSPOSE=1" "PIRE_HIP_DEBUG_NOTRANSPOSE=1" "PIRE_HIP_DEBUG_NOSTEP=1 PIRE_HIP_DEBUG_NOTRANSPOSE=1" "PIRE_HIP_DEBUG_NOHIST=1" "PIRE_HIP_DEBUG_NOTRAP=1" "PIRE_HIP_DEBUG_NOLOAD=1 PIRE_HIP_DEBUG_NOTRANSPOSE=1 PIRE_HIP_DEBUG_NOHIST=1 PIRE_HIP_DEBUG_NOTRAP=1" "X=2"; do
  env $knobs timeout 300 python bench.py --no-cpu --no-adapt --steps 30 --warmup 10 2>&1 | tail -1 | python -c "$P" "$knobs" | tee -a $OUT/ablation.log
done
