#!/bin/bash
# counters of the zipped walk's timed launches: tools/zip_pmc.sh <set:corpus> <leg> <outdir>
set -u
export PYTHONUNBUFFERED=1 TMPDIR=/tmp PYTHONPATH=.
PT=$1; LEG=$2; OUT=$3
mkdir -p $OUT
i=0
for set in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD" \
           "SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_WAVES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" \
           "TA_TA_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  WIDE_CASE_LEGS=$LEG timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/pmc$i -o pmc -- python tools/wide_case.py --log2-strings 20 --reps 5 --points $PT > $OUT/pmc$i.log 2>&1 || echo "pmc pass $i failed"
done
python tools/summarize_pmc.py $OUT --last 5 > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
