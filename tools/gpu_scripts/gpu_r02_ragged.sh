#!/bin/bash
# ragged kernel: group loads + transpose (new) against per-lane loads (tools/ab/libpire_hip_lane_loads.so), parity first
set -u
export PYTHONUNBUFFERED=1 TMPDIR=/tmp PYTHONPATH=.
OUT=gpurun_out/r02ragged
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_ragged_actions.py tests/test_half_final.py tests/test_prefix.py tests/test_segmented.py tests/test_random_scanners.py -m gpu -q -x 2>&1 | tail -3 | tee $OUT/pytest.log
for c in urls loglines uniform2k uniform8k fixed4096; do
  timeout 120 python tools/ragged_case.py $c 3 2>&1 | grep "^ragged" | sed 's/^/group loads: /' | tee -a $OUT/cases.log
  PIRE_HIP_LIB=tools/ab/libpire_hip_lane_loads.so timeout 120 python tools/ragged_case.py $c 3 2>&1 | grep "^ragged" | sed 's/^/lane loads:  /' | tee -a $OUT/cases.log
done
timeout 200 python tools/actions_case.py 2>&1 | grep -v amdgpu.ids | tail -6 | cut -c1-220 | tee $OUT/actions.log
