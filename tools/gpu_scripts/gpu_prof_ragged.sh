#!/bin/bash
# PMC passes over single ragged cases (tools/ragged_case.py)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp PYTHONPATH=.
rm -rf gpurun_out/rprof; mkdir -p gpurun_out/rprof
for c in "$@"; do
  i=0
  for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
             "SQ_INSTS_SMEM SQ_INSTS_FLAT SQ_WAIT_ANY SQ_INSTS_BRANCH SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_WR SQ_WAVES" \
             "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_VMEM SQ_IFETCH SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC"; do
    i=$((i+1))
    timeout 120 rocprofv3 --pmc $set --kernel-trace --output-format csv -d gpurun_out/rprof/$c/pmc$i -o pmc -- python tools/ragged_case.py $c 2 > gpurun_out/rprof/$c.pmc$i.log 2>&1 || echo "pass $i failed"
  done
  tail -1 gpurun_out/rprof/$c.pmc1.log
  python tools/summarize_pmc.py gpurun_out/rprof/$c > gpurun_out/rprof/$c.summary.txt 2>&1
done
find gpurun_out/rprof -name "*.csv" -size +2M -delete
