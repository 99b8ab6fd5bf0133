#!/bin/bash
set -u
export PYTHONUNBUFFERED=1
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
for cfg in "set_a 4096" "set_a 4096" "c2_single 4096" "set_b 16384" "set_d 4096"; do
  set -- $cfg
  timeout 600 python bench.py --set $1 --len $2 --steps 20 --warmup 3 --no-cpu ${3:-} 2>&1 | tail -1 | python -c "
import sys, json
l = sys.stdin.read().strip()
try:
    d = json.loads(l); r = d['roofline']
    print('$1 len $2 ${3:-}: value %8.1f GB/s  kernel avg %.4f ms min %.4f frac %.3f promoted %s' % (d['value'], r['kernel_avg_ms'], r['kernel_min_ms'], r['frac'], d['config']['table']['rows_promoted_by_adapt']))
except Exception as e:
    print('$1 FAILED', l[-400:])
"
done
