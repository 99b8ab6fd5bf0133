#!/bin/bash
# the segmented scan after the one-mode fusion, derived modes, 1 KiB segment floor and blocks sized to the task count: parity, then one string of 64 MiB .. 1 GiB
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export PYTHONPATH=.
timeout 600 python -m pytest tests/test_segmented.py tests/test_half_final.py tests/test_pair.py -m gpu -x -q 2>&1 | tail -4
timeout 300 python tools/stress_more.py 2>&1 | grep segmented
for lg in 18 16 14; do timeout 300 python bench.py --corpus cxx --one-string --log2-strings $lg --steps 10 --warmup 3 --settle 10 --no-cpu 2>&1 | tail -1 | cut -c90-260; done
for lg in 30 28 26; do LONG_TOTAL_LOG2=$lg LONG_NS=1,64 timeout 200 python tools/long_case.py 2>&1 | grep -v "amdgpu\|pire_hip segm" | cut -c1-200; done
timeout 200 python tools/long_half_final.py 2>&1 | grep -v amdgpu | cut -c1-200
