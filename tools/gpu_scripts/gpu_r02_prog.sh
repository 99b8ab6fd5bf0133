#!/bin/bash
# progressive refill (variant 14) against the standard tiled kernel: parity, then A/B
set -u
export PYTHONUNBUFFERED=1 TMPDIR=/tmp PYTHONPATH=.
OUT=gpurun_out/r02prog
mkdir -p $OUT
PIRE_HIP_TILED_VARIANT=14 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_random_scanners.py tests/test_segmented.py -m gpu -q -x 2>&1 | tail -3 | tee $OUT/pytest.log
P='import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(sys.argv[1], d["value"], r["kernel"], "avg", r["kernel_avg_ms"], "min", r["kernel_min_ms"], "frac", r["frac"], d["match_counts"]["final"], d.get("cpu_baseline",{}).get("parity_vs_gpu"))'
PIRE_HIP_TILED_VARIANT=14 timeout 300 python bench.py --steps 20 --warmup 5 --cpu-sample-log2 18 2>&1 | tail -1 | python -c "$P" "v14 parity" | tee -a $OUT/ab.log
for rep in 1 2 3; do
for v in 0 14; do
  env PIRE_HIP_TILED_VARIANT=$v timeout 300 python bench.py --no-cpu --steps 30 --warmup 10 2>&1 | tail -1 | python -c "$P" "v$v" | tee -a $OUT/ab.log
done
done
for v in 0 14; do
  env PIRE_HIP_TILED_VARIANT=$v timeout 300 python bench.py --set c2_single --no-cpu --steps 30 --warmup 10 2>&1 | tail -1 | python -c "$P" "c2 v$v" | tee -a $OUT/ab.log
  env PIRE_HIP_TILED_VARIANT=$v timeout 300 python bench.py --no-cpu --steps 5 --warmup 2 --log2-strings 22 2>&1 | tail -1 | python -c "$P" "2^22 v$v" | tee -a $OUT/ab.log
done
