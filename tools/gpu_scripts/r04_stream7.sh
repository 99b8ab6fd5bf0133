#!/bin/bash
# round 4, experiment 7: stream kernel v3 (one-round-trip staging, balanced sub-tasks, batched flush)
set -u
export PYTHONUNBUFFERED=1 TMPDIR=/tmp PYTHONPATH=.
OUT=gpurun_out/r04s7
mkdir -p $OUT
echo "== parity"
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "stream or ragged_kernel" 2>&1 | tail -4 | tee $OUT/pytest_stream.log
if ! grep -q " passed" $OUT/pytest_stream.log || grep -q "failed\|error" $OUT/pytest_stream.log; then echo "PARITY FAILED"; exit 1; fi
echo "== timings"
for c in urls loglines uniform2k uniform8k fixed4096 urls_x4 loglines_x4 urls_64k urls_256k; do
  timeout 90 python tools/ragged_case.py $c 3 2>&1 | grep "^ragged\|^stream\|fault\|Error\|error" | tail -1
done | tee $OUT/ragged_cases_stream.log
echo "== stage clocks (tuning build)"
for c in urls loglines urls_x4; do
  PIRE_HIP_LIB=tools/ab/libpire_hip_tuning.so PIRE_HIP_DEBUG_STREAM_CLOCKS=1 timeout 120 python tools/ragged_case.py $c 3 2>&1 | grep "stream clocks\|fault" | tail -2
done | tee $OUT/stream_clocks.log
