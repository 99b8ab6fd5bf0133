#!/bin/bash
# ragged kernel A/B: window addresses through DPP instead of ds_bpermute; chunks nobody will look at not fetched
export PYTHONPATH=.
PIRE_HIP_LIB=tools/ab/libpire_hip_tuning_both.so python -m pytest tests/test_gpu_parity.py tests/test_ragged_actions.py tests/test_poisoned_surroundings.py -q -m gpu -x 2>&1 | tail -2
for rep in 1 2; do
for c in urls loglines uniform2k uniform8k fixed4096; do
  for v in base skip dpp both; do
    echo -n "$v: "
    PIRE_HIP_LIB=tools/ab/libpire_hip_tuning_$v.so python tools/ragged_case.py $c 5 2>&1 | grep "^ragged"
  done
done
done
for c in urls; do
  for v in base skip dpp both; do
    PIRE_HIP_LIB=tools/ab/libpire_hip_tuning_$v.so PIRE_HIP_DEBUG_RAGGED_CLOCKS=1 python tools/ragged_case.py $c 1 2>&1 | grep "clocks" | tail -1 | sed "s/^/$v $c: /"
  done
done
