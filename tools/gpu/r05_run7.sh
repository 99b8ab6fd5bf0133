#!/bin/bash
# round 5: the LDS ceiling of the dense walk for set_d / set_a (micro_lds), the counters of the counting row kernel
cd "$GRAFT_REPO_ROOT" || exit 1
export PYTHONPATH=. PYTHONUNBUFFERED=1 TMPDIR=/tmp
O=gpurun_out
for st in set_d set_a c2_single; do timeout 300 python tools/micro_lds.py $st --waves 16 --steps 1024 --reps 64 2>&1 | grep -v amdgpu.ids; done > $O/r05_micro_lds.log 2>&1; cat $O/r05_micro_lds.log
timeout 300 python bench.py --set set_d --steps 20 --warmup 5 --no-cpu 2>/dev/null | grep '^{' | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('set_d bench: value', d['value'], 'kernel_avg_ms', d['roofline']['kernel_avg_ms'], 'frac', d['roofline']['frac'])" | tee -a $O/r05_micro_lds.log
OUT=$O/r05_pmc_counting; mkdir -p $OUT
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_WR SQ_BUSY_CYCLES SQ_WAVES" "FETCH_SIZE WRITE_SIZE"; do
  i=$((i+1))
  for c in count_glued3_advanced; do
    timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/$c/p$i -o pmc -- python tools/counting_case.py $c > $OUT/${c}_$i.log 2>&1 || echo "pmc $c pass $i failed"
  done
done
python tools/summarize_pmc.py $OUT/count_glued3_advanced --last 3 > $OUT/summary.txt 2>&1; cat $OUT/summary.txt | head -40; grep -h "GB/s" $OUT/count_glued3_advanced_1.log | cut -c1-200 | head -3
find $OUT -name "*.csv" -size +1M -delete; find $OUT -name "*.db" -delete
