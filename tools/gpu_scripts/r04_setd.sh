#!/bin/bash
# round 4: set_d with the rotated-column rows (tiled_variant 23) against the plain rows (24), same box; set_a as control
set -u
export PYTHONUNBUFFERED=1 TMPDIR=/tmp PYTHONPATH=.
OUT=gpurun_out/r04setd
mkdir -p $OUT
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "rotated_columns or tiled_kernel_shapes" 2>&1 | tail -3 | tee $OUT/pytest.log
summ() { python -c "
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r=d['roofline']
print(sys.argv[2], 'value', d['value'], 'before_adapt', d['value_before_adapt'], 'kernel avg', r['kernel_avg_ms'], 'frac', r['frac'], r['kernel'], 'traps', d['traps']['cold_lane_chunk_share'])
" $1 "$2"; }
for rep in 1 2; do
  for st in set_d set_a set_b c2_single; do
    for v in 24 23 0; do
      PIRE_HIP_TILED_VARIANT=$v timeout 300 python bench.py --set $st --no-cpu --steps 20 --warmup 5 --cold-launches 0 $( [ $st = set_b ] && echo "--len 16384 --steps 10 --settle 20" ) 2>/dev/null | tail -1 > $OUT/bench_${st}_v${v}_r${rep}.json
      summ $OUT/bench_${st}_v${v}_r${rep}.json "$st variant=$v rep=$rep"
    done
  done
done | tee $OUT/ab.log
