#!/bin/bash
# round 4: counting scanners with entries that are LDS addresses (CountingRowKernel) against the 16-bit entries
set -u
export PYTHONUNBUFFERED=1 TMPDIR=/tmp PYTHONPATH=.
OUT=gpurun_out/r04c1
mkdir -p $OUT
timeout 600 python -m pytest tests/test_counting.py -m gpu -x -q > $OUT/tests.log 2>&1; tail -3 $OUT/tests.log
for name in count_glued3_advanced count0_advanced count0_basic; do
  for v in 1 0; do
    PIRE_HIP_COUNTING_VARIANT=$v timeout 300 python tools/counting_case.py $name 2>&1 | head -2 | sed "s/^/variant=$v /"
  done
done | tee $OUT/counting.log
