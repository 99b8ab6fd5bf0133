#!/bin/bash
set -u
export PYTHONUNBUFFERED=1 TMPDIR=/tmp PYTHONPATH=.
OUT=gpurun_out/r03slow; mkdir -p $OUT
timeout 900 python -m pytest tests/test_slow.py -m gpu -x -q 2>&1 | tail -5 | tee $OUT/pytest_slow.log
for st in slow_x300 slow_x400_utf8; do
  timeout 400 python bench.py --set $st --log2-strings 16 --len 4096 --steps 5 --warmup 2 --cpu-sample-log2 10 2>&1 | tail -1 | cut -c1-1800
  PIRE_HIP_SLOW_NO_LIST=1 timeout 400 python bench.py --set $st --log2-strings 16 --len 4096 --steps 2 --warmup 1 --no-cpu 2>&1 | tail -1 | cut -c1-400
done | tee $OUT/bench_slow_wide.jsonl
timeout 600 python bench.py --set slow_x40_utf8 --len 16384 --log2-strings 18 --steps 5 --warmup 1 --no-cpu 2>&1 | tail -1 | cut -c1-1200 | tee $OUT/bench_c5b_small.json
