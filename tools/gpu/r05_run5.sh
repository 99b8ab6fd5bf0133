#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export PYTHONPATH=. PYTHONUNBUFFERED=1 TMPDIR=/tmp
O=gpurun_out
timeout 2400 python -m pytest tests -m gpu -q > $O/r05_pytest_gpu.log 2>&1; echo "pytest gpu rc=$?"; tail -12 $O/r05_pytest_gpu.log | cut -c1-200
for lg in 18 20 21; do PREFIX_LOG2_STRINGS=$lg PREFIX_SETTLE=30 timeout 600 python tools/prefix_case.py 2>&1 | grep -v "^adapt\|amdgpu.ids"; done > $O/r05_prefix_sizes_after.log 2>&1; grep Prefix $O/r05_prefix_sizes_after.log | cut -c1-160
