#!/bin/bash
set -u
export PYTHONUNBUFFERED=1
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
run() { local label="$1"; shift
  env "$@" timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu 2>&1 | tail -1 | python -c "
import sys, json
l = sys.stdin.read().strip()
try:
    d = json.loads(l); r = d['roofline']
    print('%-28s value %8.1f GB/s  kernel avg %.4f ms min %.4f frac %.3f' % ('$label', d['value'], r['kernel_avg_ms'], r['kernel_min_ms'], r['frac']))
except Exception as e:
    print('$label FAILED', l[-300:])
"
}
for i in 1 2 3; do
run "v0 norot nt fused" PIRE_HIP_TILED_VARIANT=0
done
run "v1 rot nt fused" PIRE_HIP_TILED_VARIANT=1
for s in c2_single set_d; do
timeout 300 python bench.py --set $s --steps 10 --warmup 3 --no-cpu 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip()); r = d['roofline']
print('$s: value %8.1f kernel avg %.4f min %.4f' % (d['value'], r['kernel_avg_ms'], r['kernel_min_ms']))"
done
timeout 300 python bench.py --set set_b --len 16384 --steps 10 --warmup 3 --no-cpu 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip()); r = d['roofline']
print('set_b 16K: value %8.1f kernel avg %.4f min %.4f frac %.3f' % (d['value'], r['kernel_avg_ms'], r['kernel_min_ms'], r['frac']))"
