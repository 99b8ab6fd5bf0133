#!/bin/bash
# which half of the tiled kernel degrades with several blocks per CU: walk only (NOLOAD) or loads only (NOSTEP)?
set -u
export PYTHONUNBUFFERED=1 TMPDIR=/tmp PYTHONPATH=. PIRE_HIP_LIB=tools/ab/libpire_hip_tuning.so
OUT=gpurun_out/r02exp6
mkdir -p $OUT
P='import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(sys.argv[1], "avg", r["kernel_avg_ms"], "min", r["kernel_min_ms"])'
for cfg in "0 1" "11 4" "12 2" "11 5" "11 3" "13 1"; do
  set -- $cfg
  for knobs in "X=1" "PIRE_HIP_DEBUG_NOLOAD=1" "PIRE_HIP_DEBUG_NOSTEP=1" "PIRE_HIP_DEBUG_NOLOAD=1 PIRE_HIP_DEBUG_NOTRANSPOSE=1"; do
    env $knobs PIRE_HIP_BLOCKS_PER_CU=$2 PIRE_HIP_TILED_VARIANT=$1 timeout 300 python bench.py --set c2_single --no-cpu --no-adapt --steps 20 --warmup 5 2>&1 | tail -1 | python -c "$P" "variant $1 blocks/CU $2 $knobs" | tee -a $OUT/map.log
  done
done
