#!/bin/bash
# round 4: the SlowScanner list kernel with the step costing per active group of four slots
set -u
export PYTHONUNBUFFERED=1 TMPDIR=/tmp PYTHONPATH=.
OUT=gpurun_out/r04slow2
mkdir -p $OUT
timeout 900 python -m pytest tests/test_slow.py tests/test_length_order.py -m gpu -x -q > $OUT/tests.log 2>&1; tail -3 $OUT/tests.log
timeout 300 python bench.py --set slow_x40_utf8 --steps 10 --warmup 3 2>&1 | tail -1 > $OUT/bench_c5b.json; python -c "
import json; d=json.load(open('$OUT/bench_c5b.json')); print('C5b', d['value'], d['unit'], d['ms_per_step'], 'parity', d.get('parity'), d['roofline']['frac'])"
timeout 300 python tools/slow_ragged_case.py 2>&1 | grep -v amdgpu.ids | tail -4
bash tools/gpu_scripts/r04_slow_pmc.sh > /dev/null 2>&1; python tools/summarize_pmc.py gpurun_out/r04slow/pmc | grep -A17 "SlowListKernel" | head -18 > $OUT/pmc_after.txt; grep "INSTS_VALU\|INSTS_LDS\|INSTS_SALU\|INSTS_BRANCH" $OUT/pmc_after.txt
