#!/bin/bash
# two modes as one walk of their product automaton: parity, stress, sizes
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export PYTHONPATH=.
timeout 600 python -m pytest tests/test_segmented.py tests/test_half_final.py tests/test_glue.py -m gpu -x -q 2>&1 | tail -4
timeout 300 python tools/stress_more.py 2>&1 | grep segmented
for lg in 30 28; do LONG_TOTAL_LOG2=$lg LONG_NS=1,64 timeout 200 python tools/long_case.py 2>&1 | grep -v "amdgpu" | cut -c1-330; done
