#!/bin/bash
# the headline's counters, bench line and trace again (tiled.hip changed: bench.py refuses a traffic file of other kernel sources)
set -u
export PYTHONUNBUFFERED=1 TMPDIR=/tmp PYTHONPATH=.
OUT=gpurun_out/final_r05
mkdir -p $OUT
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD" \
           "TA_TA_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rm -rf $OUT/pmc$i
  timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/pmc$i -o pmc -- python bench.py --steps 5 --warmup 1 --settle 10 --no-cpu --no-adapt --cold-launches 0 > $OUT/pmc$i.log 2>&1 || echo "pmc pass $i failed"
done
python tools/summarize_pmc.py $OUT --last 5 2>&1 > $OUT/pmc_summary.txt; grep -A16 ScanTiled $OUT/pmc_summary.txt | head -18
python tools/make_pmc_json.py $OUT/pmc_summary.txt profiles/r05_pmc_traffic.json "set_a 2^20 x 4096" "python bench.py --steps 5 --warmup 1 --settle 10 --no-cpu --no-adapt --cold-launches 0; tools/gpu/r05_refresh_headline.sh" > /dev/null && cp profiles/r05_pmc_traffic.json $OUT/pmc_traffic.json
timeout 900 python bench.py --steps 20 --warmup 5 2>&1 | tail -1 > $OUT/bench_n1.json; cut -c1-200 $OUT/bench_n1.json
timeout 900 python bench.py --no-cpu 2>&1 | tail -1 > $OUT/bench_n1_defaults.json; cut -c1-200 $OUT/bench_n1_defaults.json
rm -rf $OUT/stats
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o stats -- python bench.py --steps 20 --warmup 5 --no-cpu --cold-launches 0 > $OUT/stats.log 2>&1
python tools/summarize_trace.py $OUT/stats 20 | tee $OUT/trace_timed_region.txt
find $OUT -name "*.csv" -size +1M -delete; find $OUT -name "*.db" -delete
