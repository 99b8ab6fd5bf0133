#!/bin/bash
# round 5: counters of the wide walk (and of the dense walk on the same corpus) -- separate --pmc passes, kernel trace only
cd "$GRAFT_REPO_ROOT" || exit 1
export PYTHONUNBUFFERED=1 TMPDIR=/tmp PYTHONPATH=.
SET=${1:-dict_1k}; CORPUS=${2:-k128}; TAG=${3:-r05_pmc_wide}
OUT=gpurun_out/$TAG
mkdir -p $OUT
i=0
for set in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD" "SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_SMEM SQ_BUSY_CYCLES SQ_WAVES" "FETCH_SIZE"; do
  i=$((i+1))
  for walk in 2 1; do
    timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/w$walk/p$i -o pmc -- python bench.py --set $SET --corpus $CORPUS --walk $walk --steps 5 --warmup 1 --settle 10 --no-cpu --cold-launches 0 > $OUT/pmc_w${walk}_$i.log 2>&1 || echo "pmc walk $walk pass $i failed"
  done
done
for walk in 2 1; do echo "== $SET $CORPUS walk_variant $walk"; python tools/summarize_pmc.py $OUT/w$walk; tail -1 $OUT/pmc_w${walk}_1.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('bench line: value', d['value'], 'kernel', d['roofline']['kernel'], 'kernel_avg_ms', d['roofline']['kernel_avg_ms'], 'traps', d.get('traps'))"; done > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
find $OUT -name "*.csv" -size +1M -delete; find $OUT -name "*.db" -delete
