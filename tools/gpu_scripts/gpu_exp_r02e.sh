#!/bin/bash
set -u
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
OUT=gpurun_out/r02e
mkdir -p $OUT
P='import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], d["value"], d["roofline"]["kernel"], "avg", d["roofline"]["kernel_avg_ms"], "min", d["roofline"]["kernel_min_ms"], "frac", d["roofline"]["frac"])'
for st in 4096 4224 4352 4112 8192 4096; do
  env PIRE_HIP_TILED_VARIANT=3 timeout 300 python bench.py --no-cpu --steps 30 --warmup 10 --stride $st 2>&1 | tail -1 | python -c "$P" "stride $st" | tee -a $OUT/stride.log
done
