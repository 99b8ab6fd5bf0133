#!/bin/bash
# round 4: CapturingScanner on whole text lines (CaptureRowKernel) against the dense-row kernel
set -u
export PYTHONUNBUFFERED=1 TMPDIR=/tmp PYTHONPATH=.
OUT=gpurun_out/r04cap1
mkdir -p $OUT
timeout 600 python -m pytest tests/test_capture.py -m gpu -x -q > $OUT/tests.log 2>&1; tail -3 $OUT/tests.log
for v in 1 0; do
  PIRE_HIP_COUNTING_VARIANT=$v timeout 600 python tools/capture_case.py 2>&1 | grep -v amdgpu.ids | sed "s/^/variant=$v /"
done | tee $OUT/capture.log
