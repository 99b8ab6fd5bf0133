#!/bin/bash
# ragged kernel ablation (tuning build: WRONG results by design) + PMC of the new kernel on URLs / log lines
set -u
export PYTHONUNBUFFERED=1 TMPDIR=/tmp PYTHONPATH=.
OUT=gpurun_out/r02ragged2
mkdir -p $OUT
for c in urls loglines; do
for m in 0 1 2 3 4 8 24 27; do
  PIRE_HIP_LIB=tools/ab/libpire_hip_tuning.so PIRE_HIP_DEBUG_RAGGED=$m timeout 120 python tools/ragged_case.py $c 3 2>&1 | grep "^ragged" | sed "s/^/knobs $m: /" | tee -a $OUT/ablation.log
done
done
for c in urls loglines; do
  i=0
  for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
             "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_VMEM SQ_IFETCH SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS" \
             "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_BUSY_avr GRBM_GUI_ACTIVE" \
             "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_ACCESSES_sum" \
             "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    timeout 120 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/rprof/$c/pmc$i -o pmc -- python tools/ragged_case.py $c 2 > $OUT/rprof_$c.pmc$i.log 2>&1 || echo "pass $i failed"
  done
  python tools/summarize_pmc.py $OUT/rprof/$c > $OUT/ragged_pmc_$c.txt 2>&1
done
grep -A30 "ScanRagged" $OUT/ragged_pmc_urls.txt | head -32
find $OUT -name "*.csv" -size +1M -delete; find $OUT -name "*.db" -delete
