#!/bin/bash
# ragged kernel: the next window requested before the current one is waited for; A/B against the previous order
export PYTHONPATH=.
python -m pytest tests/test_gpu_parity.py tests/test_ragged_actions.py tests/test_poisoned_surroundings.py tests/test_segmented.py tests/test_half_final.py -q -m gpu -x 2>&1 | tail -2
for rep in 1 2; do
for c in urls loglines uniform2k uniform8k fixed4096; do
  for v in _issuelate ""; do
    echo -n "lib$v: "
    PIRE_HIP_LIB=tools/ab/libpire_hip_tuning$v.so python tools/ragged_case.py $c 5 2>&1 | grep "^ragged"
  done
done
done
for c in urls loglines; do
  for v in _issuelate ""; do
    PIRE_HIP_LIB=tools/ab/libpire_hip_tuning$v.so PIRE_HIP_DEBUG_RAGGED_CLOCKS=1 python tools/ragged_case.py $c 1 2>&1 | grep "clocks" | tail -1 | sed "s/^/lib$v $c: /"
  done
done
