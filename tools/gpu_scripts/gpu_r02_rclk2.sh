#!/bin/bash
# ragged kernel A/B: strings to lanes in column order (new) against lane order (old); alternating, one box
export PYTHONPATH=.
for rep in 1 2; do
for c in urls loglines uniform2k uniform8k fixed4096; do
  for v in laneorder colorder; do
    echo -n "$v: "
    PIRE_HIP_LIB=tools/ab/libpire_hip_tuning_$v.so python tools/ragged_case.py $c 5 2>&1 | grep "^ragged"
  done
done
done
for c in urls loglines; do
  for v in laneorder colorder; do
    PIRE_HIP_LIB=tools/ab/libpire_hip_tuning_$v.so PIRE_HIP_DEBUG_RAGGED_CLOCKS=1 python tools/ragged_case.py $c 1 2>&1 | grep "clocks" | tail -1 | sed "s/^/$v $c: /"
  done
done
