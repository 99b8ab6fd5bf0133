#!/bin/bash
# the headline table on small fixed-length batches (dense rows): kernel time per launch from bench.py's HIP events
export PYTHONPATH=. PYTHONUNBUFFERED=1
for lg in 12 14 15 16 17 18 20; do
  timeout 200 python bench.py --log2-strings $lg --steps 20 --warmup 5 --settle 20 --no-cpu --cold-launches 0 2>/dev/null | grep '^{' | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']
print('2^$lg strings x 4 KiB: %.1f GB/s, kernel avg %.4f ms (min %.4f)' % (d['value'], r['kernel_avg_ms'], r['kernel_min_ms']))"
done
