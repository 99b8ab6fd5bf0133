#!/bin/bash
# round 3: the host-pointer entry points staged through the stream-ordered pool ("failed in 2 runs of 3")
set -u
export PYTHONUNBUFFERED=1 TMPDIR=/tmp PYTHONPATH=.
for i in 1 2 3; do PIRE_HIP_HOST_STAGING=2 tests/cpp/bin/shim_test 2>&1 | grep -v "GB/s\|amdgpu.ids" | tail -3; done
for i in 1; do PIRE_HIP_HOST_STAGING=2 timeout 900 python -m pytest tests -m gpu -x -q -k "half_final or prefix or suffix or counting or capture or slow or poisoned or shim or actions" 2>&1 | tail -2; done
python tools/host_call_latency.py 2>&1 | grep -v amdgpu | tail -12
echo "== default staging (cached blocks)"
for i in 1 2 3; do tests/cpp/bin/shim_test 2>&1 | grep -v "GB/s\|amdgpu.ids" | tail -2; done
timeout 900 python -m pytest tests -m gpu -x -q -k "half_final or prefix or suffix or counting or capture or slow or poisoned or shim or actions" 2>&1 | tail -2
