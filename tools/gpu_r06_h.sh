#!/bin/bash
export PYTHONUNBUFFERED=1 TMPDIR=/tmp PYTHONPATH=.
for k in 1 2 3; do timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu --cold-launches 0 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('headline', d['value'], d['roofline']['kernel_avg_ms'], d['roofline']['frac'])"; done
for s in set_d; do timeout 300 python bench.py --set $s --steps 20 --warmup 5 --no-cpu --cold-launches 0 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('set_d', d['value'], d['roofline']['frac'])"; done
timeout 400 python tools/ranking_quality_records.py blacklist_1k urls256 1 dense 2>&1 | grep "after [1248] x" | cut -c1-260
timeout 400 python tools/ranking_quality.py blacklist_1k dense 2>&1 | grep "^after [1248] x" | cut -c1-200
for c in urls loglines; do for v in 1 0; do PIRE_HIP_RAGGED_VARIANT=$v timeout 200 python tools/ragged_case.py $c 3 2>&1 | grep "GB/s" | cut -c1-200; done; done
