#!/bin/bash
# Runs on the GPU box (via gpurun): parity tests, smoke, and bench variants.  Logs to gpurun_out/.
set -u
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
echo "== pytest -m gpu" 
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 | tee gpurun_out/pytest_gpu.log
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5 | tee gpurun_out/smoke.log
echo "== bench variants"
for v in 0 1 2 3 4 5; do
  echo "-- variant $v"
  PIRE_HIP_TILED_VARIANT=$v timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu 2>&1 | tail -1 | python -c "
import sys, json
l = sys.stdin.read().strip()
try:
    d = json.loads(l); r = d['roofline']
    print('variant $v value', d['value'], 'GB/s  kernel', r['kernel_avg_ms'], 'ms min', r['kernel_min_ms'], ' achieved', r['achieved'], 'frac', r['frac'])
except Exception as e:
    print('variant $v FAILED', l[-300:])
" | tee -a gpurun_out/bench_variants.log
done
echo "== full bench (default variant, with cpu baseline)"
timeout 600 python bench.py 2>&1 | tail -1 | tee gpurun_out/bench_default.json
