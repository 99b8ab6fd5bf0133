#!/bin/bash
# round 4: issue accounting of the SlowScanner list kernel on BASELINE C5b (bench.py --set slow_x40_utf8)
set -u
export PYTHONUNBUFFERED=1 TMPDIR=/tmp PYTHONPATH=.
OUT=gpurun_out/r04slow
mkdir -p $OUT
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VALU" \
           "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_BRANCH SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_LDS SQ_THREAD_CYCLES_VALU"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/pmc/p$i -o pmc -- python bench.py --set slow_x40_utf8 --steps 3 --warmup 1 > $OUT/pmc_$i.log 2>&1 || { echo "pmc pass $i failed"; tail -3 $OUT/pmc_$i.log; }
done
python tools/summarize_pmc.py $OUT/pmc > $OUT/pmc_summary.txt 2>&1
grep -A20 "SlowListKernel" $OUT/pmc_summary.txt | head -24
find $OUT -name "*.csv" -size +1M -delete; find $OUT -name "*.db" -delete
