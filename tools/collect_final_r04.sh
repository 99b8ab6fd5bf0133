#!/bin/bash
# copies what tools/gpu_final_r04.sh left under gpurun_out/final_r04 into profiles/ (the tracked evidence)
set -eu
S=gpurun_out/final_r04
P=profiles
cp $S/pytest_gpu.log $P/r04_final_pytest_gpu.log
cp $S/smoke.log $P/r04_final_smoke.log
cp $S/bench_n1.json $P/r04_bench_n1.json
cp $S/bench_n1_defaults.json $P/r04_bench_n1_defaults.json
cp $S/stats/stats_kernel_stats.csv $P/r04_bench_kernel_stats.csv
cp $S/trace_timed_region.txt $P/r04_bench_trace_timed_region.txt
cp $S/pmc_summary.txt $P/r04_bench_pmc_summary.txt
cp $S/pmc_traffic.json $P/r04_pmc_traffic.json
for f in c1_nonreloc c2 c5a c5b set_d c4_shard cxx_records cxx_one_string 2ranks_gloo 8ranks_gloo force_dist_nccl; do cp $S/bench_$f.json $P/r04_bench_$f.json; done
cp $S/bench_slow_wide.jsonl $P/r04_bench_slow_wide.jsonl
cp $S/ragged_cases.log $P/r04_ragged_cases.log
for c in urls loglines; do cp $S/ragged_pmc_${c}_v1.txt $P/r04_ragged_pmc_${c}.txt; cp $S/ragged_pmc_${c}_v0.txt $P/r04_stream_pmc_${c}.txt; done
for f in prefix half_final counting actions long_strings capture pair host_call_latency shim; do cp $S/$f.log $P/r04_final_$f.log; done
for f in counting_variants capture_variants half_final_variants counting_many_regexps slow_ragged slow_ragged_nostats; do cp $S/$f.log $P/r04_final_$f.log; done
cp $S/counting_kernel_stats.txt $P/r04_counting_kernel_stats.txt
cp gpurun_out/final_r04.log $P/r04_final_run.log
python tools/fill_design_tables.py
