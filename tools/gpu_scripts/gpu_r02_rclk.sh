#!/bin/bash
# ragged kernel: shader-clock time per section of an iteration (tuning build)
export PYTHONPATH=.
for c in urls loglines uniform2k; do
  for m in 0 8; do
    echo "== $c knobs $m"
    PIRE_HIP_LIB=tools/ab/libpire_hip_tuning.so PIRE_HIP_DEBUG_RAGGED=$m python tools/ragged_case.py $c 3 2>&1 | grep "^ragged"
    PIRE_HIP_LIB=tools/ab/libpire_hip_tuning.so PIRE_HIP_DEBUG_RAGGED=$m PIRE_HIP_DEBUG_RAGGED_CLOCKS=1 python tools/ragged_case.py $c 1 2>&1 | grep "clocks" | tail -1
  done
done
