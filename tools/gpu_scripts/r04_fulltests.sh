#!/bin/bash
set -u
export PYTHONUNBUFFERED=1 TMPDIR=/tmp PYTHONPATH=.
OUT=gpurun_out/r04full
mkdir -p $OUT
timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | tail -8 | tee $OUT/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee $OUT/smoke.log
tests/cpp/bin/shim_test 2>&1 | tail -3 | tee $OUT/shim.log
