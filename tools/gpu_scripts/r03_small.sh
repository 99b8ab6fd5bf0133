#!/bin/bash
# small host-pointer calls after the faster table copy: parity of the exact kernels, latencies, and the big cases for regressions
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_host_staging.py tests/test_slow.py tests/test_shim_cpp.py tests/test_prefix.py tests/test_suffix.py tests/test_half_final.py tests/test_gpu_parity.py tests/test_counting.py tests/test_capture.py -m gpu -x -q 2>&1 | tail -5 > gpurun_out/small_tests.log
PYTHONPATH=. timeout 300 python tools/host_call_latency.py > gpurun_out/small_latency.log 2>&1
for c in prefix_case half_final_case; do PYTHONPATH=. timeout 300 python tools/$c.py 2>&1 | tail -12 > gpurun_out/small_$c.log; done
cat gpurun_out/small_tests.log gpurun_out/small_latency.log gpurun_out/small_prefix_case.log gpurun_out/small_half_final_case.log
