#!/bin/bash
# waves per CU x blocks per CU map of the tiled kernel (c2_single: 11 states, the LDS never limits residency)
set -u
export PYTHONUNBUFFERED=1 TMPDIR=/tmp PYTHONPATH=.
OUT=gpurun_out/r02exp5
mkdir -p $OUT
P='import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(sys.argv[1], d["value"], r["kernel"], "avg", r["kernel_avg_ms"], "min", r["kernel_min_ms"], "frac", r["frac"])'
for cfg in "11 1" "11 2" "11 3" "11 4" "11 5" "12 1" "12 2" "13 1" "0 1" "9 1" "9 2"; do
  set -- $cfg
  env PIRE_HIP_BLOCKS_PER_CU=$2 PIRE_HIP_TILED_VARIANT=$1 timeout 300 python bench.py --set c2_single --no-cpu --steps 20 --warmup 5 2>&1 | tail -1 | python -c "$P" "c2 variant $1 blocks/CU $2" | tee -a $OUT/map.log
done
