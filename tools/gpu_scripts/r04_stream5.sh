#!/bin/bash
# round 4, experiment 5: the stream kernel's stage clocks at settled clocks (accumulated, read once), default-config tests
set -u
export PYTHONUNBUFFERED=1 TMPDIR=/tmp PYTHONPATH=.
OUT=gpurun_out/r04s5
mkdir -p $OUT
for c in urls urls_64k loglines uniform2k urls_x4 fixed4096; do
  PIRE_HIP_LIB=tools/ab/libpire_hip_tuning.so PIRE_HIP_DEBUG_STREAM_CLOCKS=1 timeout 120 python tools/ragged_case.py $c 3 2>&1 | grep "stream clocks\|^stream\|fault" | tail -2
done | tee $OUT/stream_clocks.log
echo "== default config + RCCL tests"
timeout 600 python -m pytest tests/test_default_config.py tests/test_multi_gpu.py -m gpu -x -q 2>&1 | tail -6 | tee $OUT/pytest_default_multi.log
