#!/bin/bash
# round 3, experiment 2: bench.py with the clock-settling phase, held-out-seed ranking, fresh-table cold leg
set -u
export PYTHONUNBUFFERED=1 TMPDIR=/tmp PYTHONPATH=.
OUT=gpurun_out/r03exp2
mkdir -p $OUT
summ() { python -c "
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r=d['roofline']
print(sys.argv[2], 'value', d['value'], 'ms/step', d['ms_per_step'], 'before_adapt', d['value_before_adapt'], 'kernel avg', r['kernel_avg_ms'], 'min', r['kernel_min_ms'], 'frac', r['frac'], r['kernel'], 'promoted', d['config']['table']['rows_promoted_by_adapt'])
" $1 "$2"; }
for rep in 1 2; do
  for v in 0 22; do
    PIRE_HIP_TILED_VARIANT=$v timeout 300 python bench.py --no-cpu --steps 20 --warmup 5 2>$OUT/err.log | tail -1 > $OUT/bench_v${v}_r${rep}.json
    summ $OUT/bench_v${v}_r${rep}.json "variant=$v rep=$rep" || tail -5 $OUT/err.log
  done
done | tee $OUT/ab.log
timeout 300 python bench.py --no-cpu --steps 20 --warmup 5 --settle 0 2>/dev/null | tail -1 > $OUT/bench_nosettle.json; summ $OUT/bench_nosettle.json "settle=0" | tee -a $OUT/ab.log
for st in c2_single set_d set_b; do timeout 300 python bench.py --no-cpu --steps 20 --warmup 5 --set $st 2>/dev/null | tail -1 > $OUT/bench_$st.json; summ $OUT/bench_$st.json "$st" | tee -a $OUT/ab.log; done
timeout 300 python bench.py --no-cpu --steps 20 --warmup 5 --corpus cxx 2>/dev/null | tail -1 > $OUT/bench_cxx.json; summ $OUT/bench_cxx.json "cxx" | tee -a $OUT/ab.log
timeout 300 python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > $OUT/bench_full.json; summ $OUT/bench_full.json "full (cpu leg)" | tee -a $OUT/ab.log
python -c "
import json; d=json.loads(open('$OUT/bench_full.json').read().strip().splitlines()[-1]); print(d['cpu_baseline']['value'], d['cpu_baseline']['parity_vs_gpu'], d['match_counts'])"
