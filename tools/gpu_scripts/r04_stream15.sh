#!/bin/bash
# round 4, experiment 15: the B path's two selects under a scalar branch (most steps of a chunk have no boundary), masks as lo[b] & hi[w]
set -u
export PYTHONUNBUFFERED=1 TMPDIR=/tmp PYTHONPATH=.
OUT=gpurun_out/r04s15
mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "stream or ragged_kernel" > $OUT/parity.log 2>&1; tail -1 $OUT/parity.log
for round in 1 2; do
  for c in urls loglines uniform2k; do
    for l in tools/ab/libpire_hip_prev.so ""; do
      PIRE_HIP_LIB=$l timeout 120 python tools/ragged_case.py $c 3 2>&1 | tail -1 | sed "s|^|${l:-NEW} |"
    done
  done
done > $OUT/ab.log; grep -o "^[^ ]* stream [a-z0-9]*\|mean [0-9.]* ms -> [0-9.]* GB/s" $OUT/ab.log | paste - -
