#!/bin/bash
# rocprofv3 passes for bench.py on the GPU box: kernel trace/stats, then PMC counter passes (separate runs).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/prof
export TMPDIR=/tmp
V="${PIRE_HIP_TILED_VARIANT:-0}"
CMD="python bench.py --steps 5 --warmup 1 --no-cpu"
echo "== kernel trace / stats (variant $V)"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/stats -o stats -- $CMD > gpurun_out/prof/stats.log 2>&1
tail -2 gpurun_out/prof/stats.log
find gpurun_out/prof/stats -name "*kernel_stats*" | head -3
for f in $(find gpurun_out/prof/stats -name "*kernel_stats.csv" | head -1); do head -8 "$f"; done
i=0
for set in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_RD SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL" \
           "SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_IFETCH SQ_INSTS_SALU SQ_INSTS_SMEM SQ_BUSY_CU_CYCLES" \
           "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE GRBM_COUNT" "TCC_HIT_sum TCC_MISS_sum" \
           "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum" "TA_ADDR_STALLED_BY_TD_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
           "TD_TD_BUSY_sum TD_TC_STALL_sum" "TD_SPI_STALL_sum TD_LOAD_WAVEFRONT_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
           "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum" \
           "TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TD_TCP_STALL_CYCLES_sum TCP_LFIFO_STALL_CYCLES_sum"; do
  i=$((i+1))
  echo "== pmc pass $i: $set"
  timeout 600 rocprofv3 --pmc $set --kernel-trace --output-format csv -d gpurun_out/prof/pmc$i -o pmc -- $CMD > gpurun_out/prof/pmc$i.log 2>&1 || echo "pass $i failed: $(tail -3 gpurun_out/prof/pmc$i.log)"
done
python tools/summarize_pmc.py gpurun_out/prof 2>&1 | tee gpurun_out/prof/pmc_summary.txt
