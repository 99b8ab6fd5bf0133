#!/bin/bash
set -u
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
OUT=gpurun_out/r02d
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "two_chain or tiled or big_sets or full_size or cold or resume" 2>&1 | tail -3 | tee $OUT/pytest.log
P='import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], d["value"], d["roofline"]["kernel"], "avg", d["roofline"]["kernel_avg_ms"], "min", d["roofline"]["kernel_min_ms"], "frac", d["roofline"]["frac"], "cold", d.get("value_before_adapt"))'
for v in 3 0 3 0; do
  env PIRE_HIP_TILED_VARIANT=$v timeout 300 python bench.py --no-cpu --steps 30 --warmup 10 2>&1 | tail -1 | python -c "$P" "v$v" | tee -a $OUT/variants.log
done
for v in 3 0; do
  env PIRE_HIP_TILED_VARIANT=$v timeout 300 python bench.py --no-cpu --steps 5 --warmup 2 --log2-strings 23 2>&1 | tail -1 | python -c "$P" "v$v 2^23" | tee -a $OUT/variants.log
done
