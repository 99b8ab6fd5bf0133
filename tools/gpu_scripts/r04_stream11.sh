#!/bin/bash
set -u
export PYTHONUNBUFFERED=1 TMPDIR=/tmp PYTHONPATH=.
OUT=gpurun_out/r04s11
mkdir -p $OUT
for c in urls; do
  echo -n "product                     : "; timeout 90 python tools/ragged_case.py $c 3 2>&1 | grep "^stream\|fault" | tail -1
  echo -n "boundaries not processed (4): "; NO_OUT=1 PIRE_HIP_LIB=tools/ab/libpire_hip_exp4.so timeout 90 python tools/ragged_case.py $c 3 2>&1 | grep "^stream\|fault" | tail -1
  echo -n "plain step everywhere    (5): "; NO_OUT=1 PIRE_HIP_LIB=tools/ab/libpire_hip_exp5.so timeout 90 python tools/ragged_case.py $c 3 2>&1 | grep "^stream\|fault" | tail -1
done | tee $OUT/stream_variants2.log
