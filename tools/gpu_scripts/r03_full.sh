#!/bin/bash
# the whole GPU suite, the randomised differential runs, the small-call latencies
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > gpurun_out/full_tests.log
PYTHONPATH=. timeout 600 python tools/stress_more.py 2>&1 | tail -6 > gpurun_out/full_stress.log
PYTHONPATH=. timeout 600 python tools/stress_random.py 2>&1 | tail -3 >> gpurun_out/full_stress.log
PYTHONPATH=. timeout 300 python tools/host_call_latency.py > gpurun_out/full_latency.log 2>&1
cat gpurun_out/full_tests.log gpurun_out/full_stress.log gpurun_out/full_latency.log
