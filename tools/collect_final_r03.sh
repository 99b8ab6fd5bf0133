#!/bin/bash
# copies what tools/gpu_final_r03.sh left under gpurun_out/final_r03 into profiles/ (the tracked evidence)
set -eu
S=gpurun_out/final_r03
P=profiles
cp $S/pytest_gpu.log $P/r03_final_pytest_gpu.log
cp $S/smoke.log $P/r03_final_smoke.log
cp $S/bench_n1.json $P/r03_bench_n1.json
cp $S/bench_n1_defaults.json $P/r03_bench_n1_defaults.json
cp $S/bench_n1_settle0.json $P/r03_bench_n1_settle0.json
cp $S/stats/stats_kernel_stats.csv $P/r03_bench_kernel_stats.csv
cp $S/trace_timed_region.txt $P/r03_bench_trace_timed_region.txt
cp $S/pmc_summary.txt $P/r03_bench_pmc_summary.txt
python tools/make_pmc_json.py $P/r03_bench_pmc_summary.txt $P/r03_pmc_traffic.json "set_a 2^20 x 4096" "python bench.py --steps 5 --warmup 1 --settle 10 --no-cpu --no-adapt; tools/gpu_final_r03.sh" > /dev/null
for f in c2 c5a c5b set_d c4_shard cxx_records cxx_one_string 2ranks_gloo; do cp $S/bench_$f.json $P/r03_bench_$f.json; done
cp $S/bench_slow_wide.jsonl $P/r03_bench_slow_wide.jsonl
cp $S/ragged_cases.log $P/r03_final_ragged_cases.log
for f in prefix half_final counting actions long_strings long_half_final capture pair host_mode host_call_latency shim warmup_curve; do cp $S/$f.log $P/r03_final_$f.log; done
cp $S/micro_smallcall.log $P/r03_micro_smallcall.log
cp $S/small_call_timeline.log $P/r03_small_call_timeline.log
