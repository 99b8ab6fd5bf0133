#!/bin/bash
# copies what tools/gpu_final_r02.sh left under gpurun_out/final_r02 into profiles/ (the tracked evidence)
set -eu
S=gpurun_out/final_r02
P=profiles
cp $S/pytest_gpu.log $P/r02_final_pytest_gpu.log
cp $S/smoke.log $P/r02_final_smoke.log
cp $S/bench_n1.json $P/r02_bench_n1.json
cp $S/bench_n1_defaults.json $P/r02_bench_n1_defaults.json
cp $S/stats/stats_kernel_stats.csv $P/r02_bench_kernel_stats.csv
cp $S/pmc_summary.txt $P/r02_bench_pmc_summary.txt
for f in c2 c5a c5b set_d c4_shard cxx_records cxx_one_string 2ranks_gloo; do cp $S/bench_$f.json $P/r02_bench_$f.json; done
cp $S/bench_slow_wide.jsonl $P/r02_bench_slow_wide.jsonl
cp $S/ragged_cases.log $P/r02_final_ragged_cases.log
for f in prefix half_final counting actions long_strings long_half_final capture pair host_mode shim; do cp $S/$f.log $P/r02_final_$f.log; done
