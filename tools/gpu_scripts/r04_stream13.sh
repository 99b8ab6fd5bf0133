#!/bin/bash
# round 4, experiment 13: the stream kernel after the stream_common.h cut, with and without lanes levelling their pieces out
set -u
export PYTHONUNBUFFERED=1 TMPDIR=/tmp PYTHONPATH=.
OUT=gpurun_out/r04s13
mkdir -p $OUT
timeout 120 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1 || { echo "SMOKE FAILED"; tail -5 $OUT/smoke.log; exit 1; }
tail -1 $OUT/smoke.log
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "stream or ragged_kernel" > $OUT/parity.log 2>&1; tail -2 $OUT/parity.log
for rr in 0 2 4 8; do
  for c in urls loglines uniform2k loglines_x4; do
    PIRE_HIP_STREAM_RELAX=$rr timeout 120 python tools/ragged_case.py $c 3 2>&1 | tail -1 | sed "s/^/relax=$rr /"
  done
done | tee $OUT/relax.log
