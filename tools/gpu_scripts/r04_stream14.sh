#!/bin/bash
# round 4, experiment 14: the first sub-task's positions staged before the table copy (A = the library before, B = with it)
set -u
export PYTHONUNBUFFERED=1 TMPDIR=/tmp PYTHONPATH=.
OUT=gpurun_out/r04s14
mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "stream or ragged_kernel" > $OUT/parity.log 2>&1; tail -1 $OUT/parity.log
for round in 1 2; do
  for c in urls loglines uniform2k; do
    for l in tools/ab/libpire_hip_prev.so ""; do
      PIRE_HIP_LIB=$l timeout 120 python tools/ragged_case.py $c 3 2>&1 | tail -1 | sed "s|^|${l:-NEW} |" | cut -c1-40,100-200
    done
  done
done | tee $OUT/ab.log
