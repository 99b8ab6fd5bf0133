#!/bin/bash
# round 4, last run: the GPU suite, smoke, shim and the row-kernel comparisons at HEAD (the bench line, its PMC file and
# the other configs are tools/gpu_final_r04.sh's; nothing they measure changed since)
set -u
export PYTHONUNBUFFERED=1 TMPDIR=/tmp PYTHONPATH=.
OUT=gpurun_out/final_r04
mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -4 | tee $OUT/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee $OUT/smoke.log
tests/cpp/bin/shim_test 2>&1 | tail -2 | tee $OUT/shim.log
for v in 1 0; do PIRE_HIP_COUNTING_VARIANT=$v timeout 400 python tools/capture_case.py 2>&1 | grep "^capture" | sed "s/^/variant=$v: /"; done | tee $OUT/capture_variants.log | cut -c1-220
timeout 300 python tools/capture_case.py 2>&1 | grep -v amdgpu.ids | tee $OUT/capture.log | cut -c1-200
timeout 120 python bench.py --steps 20 --warmup 5 --no-cpu 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench', d['value'], d['roofline']['frac'], 'traffic', d['roofline']['traffic'])"
