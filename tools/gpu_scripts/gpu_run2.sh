#!/bin/bash
set -u
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
echo "== pytest -m gpu"
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee gpurun_out/pytest_gpu.log
echo "== micro_lds"
timeout 300 ./tools/micro_lds 2>&1 | tee gpurun_out/micro_lds_r01.log
echo "== prof"
timeout 1500 ./tools/gpu_prof.sh 2>&1 | tail -80
