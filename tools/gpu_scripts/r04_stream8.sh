#!/bin/bash
# round 4, experiment 8: what the stream kernel's stages cost in the PRODUCT build -- variants without the flush, with
# the 64-at-a-time flush, without the walk, and without result arrays
set -u
export PYTHONUNBUFFERED=1 TMPDIR=/tmp PYTHONPATH=.
OUT=gpurun_out/r04s8
mkdir -p $OUT
for c in urls loglines; do
  echo -n "product        : "; timeout 90 python tools/ragged_case.py $c 3 2>&1 | grep "^stream\|fault" | tail -1
  echo -n "no result arrays: "; NO_OUT=1 timeout 90 python tools/ragged_case.py $c 3 2>&1 | grep "^stream\|fault" | tail -1
  echo -n "no flush (exp1): "; PIRE_HIP_LIB=tools/ab/libpire_hip_exp1.so timeout 90 python tools/ragged_case.py $c 3 2>&1 | grep "^stream\|fault" | tail -1
  echo -n "flush 64 (exp2): "; PIRE_HIP_LIB=tools/ab/libpire_hip_exp2.so timeout 90 python tools/ragged_case.py $c 3 2>&1 | grep "^stream\|fault" | tail -1
  echo -n "no walk  (exp3): "; PIRE_HIP_LIB=tools/ab/libpire_hip_exp3.so timeout 90 python tools/ragged_case.py $c 3 2>&1 | grep "^stream\|fault" | tail -1
done | tee $OUT/stream_variants.log
