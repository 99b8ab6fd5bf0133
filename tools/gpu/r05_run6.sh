#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export PYTHONPATH=. PYTHONUNBUFFERED=1 TMPDIR=/tmp
O=gpurun_out
for lg in 20 21; do PREFIX_LOG2_STRINGS=$lg PREFIX_SETTLE=30 timeout 600 python tools/prefix_case.py ${PREFIX_MODE:-adapt_by_prefix} 2>&1 | grep -v "amdgpu.ids"; done > $O/r05_prefix_sizes_adapted.log 2>&1; grep "Prefix\|adapt" $O/r05_prefix_sizes_adapted.log | cut -c1-160
