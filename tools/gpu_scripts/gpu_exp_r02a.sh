#!/bin/bash
# Round 2, experiment A: the two-chains-per-lane tiled kernel -- parity first, then A/B against the one-chain kernel.
set -u
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
mkdir -p gpurun_out/r02a
echo "== parity (tiled paths)"
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "two_chain or tiled or big_sets or full_size or cold or resume" 2>&1 | tail -5 | tee gpurun_out/r02a/pytest.log
for v in 0 3 4; do
  echo "== variant $v, 2^20 strings"
  PIRE_HIP_TILED_VARIANT=$v timeout 300 python bench.py --no-cpu --steps 30 --warmup 10 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['kernel'], d['roofline']['kernel_avg_ms'], d['roofline']['kernel_min_ms'], d['roofline']['frac'])" | tee -a gpurun_out/r02a/variants.log
done
echo "== variant 0 and 3 at 1179648 strings (3 full rounds of 24 chains)"
for v in 0 3; do
  PIRE_HIP_TILED_VARIANT=$v timeout 300 python bench.py --no-cpu --steps 30 --warmup 10 --strings 1179648 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['kernel'], d['roofline']['kernel_avg_ms'], d['roofline']['kernel_min_ms'], d['roofline']['frac'])" | tee -a gpurun_out/r02a/variants.log
done
echo "== C4-size shard 2^23 strings, variant 0 and 3"
for v in 0 3; do
  PIRE_HIP_TILED_VARIANT=$v timeout 300 python bench.py --no-cpu --steps 5 --warmup 2 --log2-strings 23 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['kernel'], d['roofline']['kernel_avg_ms'], d['roofline']['kernel_min_ms'], d['roofline']['frac'])" | tee -a gpurun_out/r02a/variants.log
done
