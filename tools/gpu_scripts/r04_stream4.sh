#!/bin/bash
# round 4, experiment 4: where the stream kernel's time goes -- wall-clock stamps per stage and wave (tuning build)
set -u
export PYTHONUNBUFFERED=1 TMPDIR=/tmp PYTHONPATH=.
OUT=gpurun_out/r04s4
mkdir -p $OUT
for c in urls urls_64k urls_256k loglines uniform2k urls_x4; do
  echo "== $c"
  PIRE_HIP_LIB=tools/ab/libpire_hip_tuning.so PIRE_HIP_DEBUG_STREAM_CLOCKS=1 timeout 120 python tools/ragged_case.py $c 1 2>&1 | grep "stream clocks\|^stream\|fault" | tail -3
done | tee $OUT/stream_clocks.log
echo "== lambda"
for lam in 1 16 64; do
  for c in urls loglines; do
    echo -n "lambda=$lam "; PIRE_HIP_LIB=tools/ab/libpire_hip_tuning.so PIRE_HIP_STREAM_LAMBDA=$lam timeout 120 python tools/ragged_case.py $c 2 2>&1 | grep "^stream\|fault" | tail -1
  done
done | tee $OUT/stream_lambda.log
echo "== default config + RCCL tests"
timeout 600 python -m pytest tests/test_default_config.py tests/test_multi_gpu.py -m gpu -x -q 2>&1 | tail -6 | tee $OUT/pytest_default_multi.log
