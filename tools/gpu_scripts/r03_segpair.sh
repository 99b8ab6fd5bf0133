#!/bin/bash
# the segmented scan with two modes fused into one pass of the pair kernel: parity tests, then the long-string cases
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_segmented.py tests/test_pair.py tests/test_half_final.py tests/test_random_scanners.py -m gpu -x -q 2>&1 | tail -15 > gpurun_out/segpair_tests.log
PYTHONPATH=. LONG_TOTAL_LOG2=30 LONG_NS=1,8,64,1024 timeout 600 python tools/long_case.py set_a > gpurun_out/segpair_long.log 2>&1
PYTHONPATH=. timeout 600 python tools/stress_more.py 2>&1 | tail -5 > gpurun_out/segpair_stress.log
cat gpurun_out/segpair_stress.log gpurun_out/segpair_tests.log gpurun_out/segpair_long.log
