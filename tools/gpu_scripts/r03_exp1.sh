#!/bin/bash
# round 3, experiment 1: shadowed transpose (+ trap tests off the chain) against round 2's kernel, same box, alternating
set -u
export PYTHONUNBUFFERED=1 TMPDIR=/tmp PYTHONPATH=.
OUT=gpurun_out/r03exp1
mkdir -p $OUT
echo "== parity subset"
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "tiled or work_counters or empty_fixed or golden or strided or checked or cold" 2>&1 | tail -5 | tee $OUT/pytest_subset.log
summ() { python -c "
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r=d['roofline']
print(sys.argv[2], 'value', d['value'], 'before_adapt', d['value_before_adapt'], 'kernel avg', r['kernel_avg_ms'], 'min', r['kernel_min_ms'], 'frac', r['frac'], r['kernel'])
" $1 $2; }
for rep in 1 2 3; do
  for v in 0 22 21; do
    PIRE_HIP_TILED_VARIANT=$v timeout 300 python bench.py --no-cpu --steps 20 --warmup 5 2>/dev/null | tail -1 > $OUT/bench_v${v}_r${rep}.json
    summ $OUT/bench_v${v}_r${rep}.json "variant=$v rep=$rep"
  done
done | tee $OUT/ab.log
echo "== no compact tier (ramp)"
for v in 0 21; do
  PIRE_HIP_NO_COMPACT=1 PIRE_HIP_TILED_VARIANT=$v timeout 300 python bench.py --no-cpu --steps 20 --warmup 5 2>/dev/null | tail -1 > $OUT/bench_nocompact_v${v}.json
  summ $OUT/bench_nocompact_v${v}.json "nocompact variant=$v"
done | tee -a $OUT/ab.log
echo "== defaults (50 after 20), 2^23"
for v in 0 21; do
  PIRE_HIP_TILED_VARIANT=$v timeout 300 python bench.py --no-cpu 2>/dev/null | tail -1 > $OUT/bench_def_v${v}.json; summ $OUT/bench_def_v${v}.json "defaults variant=$v"
  PIRE_HIP_TILED_VARIANT=$v timeout 300 python bench.py --no-cpu --log2-strings 23 --steps 5 --warmup 2 2>/dev/null | tail -1 > $OUT/bench_23_v${v}.json; summ $OUT/bench_23_v${v}.json "2^23 variant=$v"
  PIRE_HIP_TILED_VARIANT=$v timeout 300 python bench.py --no-cpu --set c2_single --steps 20 --warmup 5 2>/dev/null | tail -1 > $OUT/bench_c2_v${v}.json; summ $OUT/bench_c2_v${v}.json "c2 variant=$v"
  PIRE_HIP_TILED_VARIANT=$v timeout 300 python bench.py --no-cpu --corpus cxx --steps 20 --warmup 5 2>/dev/null | tail -1 > $OUT/bench_cxx_v${v}.json; summ $OUT/bench_cxx_v${v}.json "cxx variant=$v"
done | tee -a $OUT/ab.log
