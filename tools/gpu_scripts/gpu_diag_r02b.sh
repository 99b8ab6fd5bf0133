#!/bin/bash
# Round 2, diagnosis B: why is the two-chain tiled kernel slower?  PMC passes for both kernels + clock / power samples.
set -u
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
OUT=gpurun_out/r02b
mkdir -p $OUT
CMD="python bench.py --steps 5 --warmup 2 --no-cpu"
for v in 3 0; do
  export PIRE_HIP_TILED_VARIANT=$v
  i=0
  for set in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU" \
             "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INST_CYCLES_SALU SQ_IFETCH SQ_INSTS_VMEM_RD" \
             "GRBM_GUI_ACTIVE GRBM_COUNT" "SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_THREAD_CYCLES_VALU SQ_LDS_ADDR_CONFLICT"; do
    i=$((i+1))
    timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/v$v/pmc$i -o pmc -- $CMD > $OUT/v$v.pmc$i.log 2>&1 || echo "pass $i failed: $(tail -3 $OUT/v$v.pmc$i.log)"
  done
  python tools/summarize_pmc.py $OUT/v$v 2>&1 | grep -A40 "ScanTiled" > $OUT/pmc_summary_v$v.txt
  cat $OUT/pmc_summary_v$v.txt
  echo "== clocks under sustained load, variant $v"
  (python bench.py --steps 4000 --warmup 5 --no-cpu > $OUT/sustained_v$v.json 2>&1 &)
  sleep 9
  for k in 1 2 3 4; do rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|mclk\|power" | head -4; sleep 0.4; done | tee $OUT/clocks_v$v.txt
  sleep 4
  tail -c 600 $OUT/sustained_v$v.json | head -c 300; echo
done
