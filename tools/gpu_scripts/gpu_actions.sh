#!/bin/bash
# ragged kernel with actions: parity tests, then throughput against the exact kernels
set -u
export PYTHONUNBUFFERED=1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_ragged_actions.py -m gpu -x -q 2>&1 | tail -5
PYTHONPATH=. timeout 600 python tools/actions_case.py 2>&1 | tee gpurun_out/actions_case.log | tail -60
