#!/bin/bash
# Round 2, call 1: state of the tree on the GPU -- LDS micro-benchmark (u8+v_perm against u16+v_dot4 chains), headline
# bench + rocprofv3 kernel stats, config C5 at its stated size (C5a set_b, C5b SlowScanner), ragged-kernel PMC passes,
# multi-GPU tests.
set -u
export PYTHONUNBUFFERED=1 TMPDIR=/tmp PYTHONPATH=.
OUT=gpurun_out/r02base
mkdir -p $OUT
echo "== micro_lds"
timeout 120 tools/micro_lds > $OUT/micro_lds.log 2>&1; grep -A22 "realistic" $OUT/micro_lds.log | head -40
echo "== bench (driver style: --steps 20 --warmup 5)"
timeout 600 python bench.py --steps 20 --warmup 5 2>&1 | tail -1 > $OUT/bench_n1_driver_style.json; cut -c1-400 $OUT/bench_n1_driver_style.json
echo "== bench (default)"
timeout 600 python bench.py --no-cpu 2>&1 | tail -1 > $OUT/bench_n1.json; cut -c1-300 $OUT/bench_n1.json
echo "== rocprofv3 kernel stats"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o stats -- python bench.py --no-cpu > $OUT/stats.log 2>&1
head -4 $OUT/stats/stats_kernel_stats.csv
echo "== C5a at the stated size: set_b, 2^20 x 16 KiB"
timeout 900 python bench.py --set set_b --len 16384 --log2-strings 20 --steps 10 --warmup 3 --cpu-sample-log2 14 2>&1 | tail -1 > $OUT/bench_c5a.json; cut -c1-300 $OUT/bench_c5a.json
echo "== C5b at the stated size: SlowScanner, 2^20 x 16 KiB"
timeout 900 python bench.py --set slow_x40_utf8 --len 16384 --log2-strings 20 --steps 5 --warmup 1 2>&1 | tail -1 > $OUT/bench_c5b.json; cut -c1-300 $OUT/bench_c5b.json
echo "== multi-GPU tests + tiled parity"
timeout 900 python -m pytest tests/test_multi_gpu.py tests/test_gpu_parity.py tests/test_fuzz_blobs.py tests/test_glue.py -m gpu -q -x 2>&1 | tail -4 | tee $OUT/pytest_subset.log
echo "== ragged PMC (urls, loglines)"
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
  tail -1 $OUT/rprof_$c.pmc1.log
  python tools/summarize_pmc.py $OUT/rprof/$c > $OUT/ragged_pmc_$c.txt 2>&1
  grep -A40 "ScanRagged" $OUT/ragged_pmc_$c.txt | head -45
done
find $OUT -name "*.csv" -size +1M -delete
find $OUT -name "*.db" -delete
du -sh $OUT
