#!/bin/bash
# round 4, experiment 10: the restart of a boundary chunk as a select of the v_perm selector (start state = dense id 0)
set -u
export PYTHONUNBUFFERED=1 TMPDIR=/tmp PYTHONPATH=.
OUT=gpurun_out/r04s10
mkdir -p $OUT
echo "== parity"
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -4 | tee $OUT/pytest_parity.log
if ! grep -q " passed" $OUT/pytest_parity.log || grep -q "failed\|error" $OUT/pytest_parity.log; then echo "PARITY FAILED"; exit 1; fi
echo "== timings"
for c in urls loglines uniform2k uniform8k fixed4096 urls_x4 loglines_x4 urls_64k urls_256k; do
  timeout 90 python tools/ragged_case.py $c 3 2>&1 | grep "^ragged\|^stream\|fault\|Error\|error" | tail -1
done | tee $OUT/ragged_cases_stream.log
