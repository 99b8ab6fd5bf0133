#!/bin/bash
# counters of the prefix search on the ragged kernel with actions against the plain scan of the same kind of batch
cd "$GRAFT_REPO_ROOT" || exit 1
export PYTHONPATH=. PYTHONUNBUFFERED=1 TMPDIR=/tmp
OUT=gpurun_out/r05_pmc_prefix; mkdir -p $OUT
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_BUSY_CYCLES SQ_WAVES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SMEM"; do
  i=$((i+1))
  PREFIX_LOG2_STRINGS=20 PREFIX_SETTLE=3 timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/prefix/p$i -o pmc -- python tools/prefix_case.py > $OUT/prefix_$i.log 2>&1 || echo "prefix pass $i failed"
  PIRE_HIP_RAGGED_VARIANT=1 timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/plain/p$i -o pmc -- python tools/ragged_case.py loglines 1 > $OUT/plain_$i.log 2>&1 || echo "plain pass $i failed"
done
for w in prefix plain; do echo "== $w"; python tools/summarize_pmc.py $OUT/$w --last 5; done > $OUT/summary.txt 2>&1; cat $OUT/summary.txt
grep -h "GB/s" $OUT/prefix_1.log $OUT/plain_1.log | cut -c1-200
find $OUT -name "*.csv" -size +1M -delete; find $OUT -name "*.db" -delete
