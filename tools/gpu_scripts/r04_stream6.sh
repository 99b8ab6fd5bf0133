#!/bin/bash
set -u
export PYTHONUNBUFFERED=1 TMPDIR=/tmp PYTHONPATH=.
OUT=gpurun_out/r04s6
mkdir -p $OUT
for c in urls loglines; do
  PIRE_HIP_LIB=tools/ab/libpire_hip_tuning.so PIRE_HIP_DEBUG_STREAM_CLOCKS=1 timeout 120 python tools/ragged_case.py $c 3 2>&1 | grep "stream clocks\|^stream\|fault" | tail -3
done | tee $OUT/stream_clocks.log
