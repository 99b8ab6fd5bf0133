#!/bin/bash
# round 4, experiment 12: issue accounting of the stream kernel (URL batch)
set -u
export PYTHONUNBUFFERED=1 TMPDIR=/tmp PYTHONPATH=.
OUT=gpurun_out/r04s12
mkdir -p $OUT
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_IFETCH SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_BRANCH SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU"; do
  i=$((i+1))
  timeout 90 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/pmc_urls/p$i -o pmc -- python tools/ragged_case.py urls 1 > $OUT/pmc_$i.log 2>&1 || { echo "pmc pass $i failed"; tail -3 $OUT/pmc_$i.log; }
done
python tools/summarize_pmc.py $OUT/pmc_urls > $OUT/pmc_summary_urls.txt 2>&1
grep -A30 "ScanStream" $OUT/pmc_summary_urls.txt | head -34
find $OUT -name "*.csv" -size +1M -delete; find $OUT -name "*.db" -delete
