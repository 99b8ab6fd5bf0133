#!/bin/bash
# copies what tools/gpu_final_r06.sh left under gpurun_out/final_r06 into profiles/ (the tracked evidence)
set -eu
S=gpurun_out/final_r06
P=profiles
cp $S/pytest_gpu.log $P/r06_final_pytest_gpu.log
cp $S/smoke.log $P/r06_final_smoke.log
cp $S/bench_n1.json $P/r06_bench_n1.json
cp $S/bench_n1_defaults.json $P/r06_bench_n1_defaults.json
cp $S/stats/stats_kernel_stats.csv $P/r06_bench_kernel_stats.csv
cp $S/trace_timed_region.txt $P/r06_bench_trace_timed_region.txt
cp $S/pmc_summary.txt $P/r06_bench_pmc_summary.txt
cp $S/pmc_traffic.json $P/r06_pmc_traffic.json
for f in c1_nonreloc c2 c5a c5b set_d c4_shard cxx_records cxx_one_string 2ranks_gloo 8ranks_gloo c4_2ranks_gloo force_dist_nccl force_dist_nccl_every_step \
         set_b_mix_mix dict_1k_k32 dict_1k_k128 dict_1k_k512 dict_1k_k1000 dict_10k_k32 dict_10k_k512 dict_10k_k2048 dict_10k_k10000 dict_utf8_1k_k32 dict_utf8_1k_k1000 dict_utf8_5k_k512 dict_utf8_5k_k5000 c5_dict_10k_16k dict_1k_k128_dense_rows dict_10k_k10000_plain_rows dict_1k_k1000_plain_rows; do cp $S/bench_$f.json $P/r06_bench_$f.json; done
cp $S/wide_curve.jsonl $P/r06_wide_curve.jsonl
cp $S/wide_pmc_fit.txt $P/r06_wide_pmc_dict_1k_k128.txt
cp $S/wide_pmc_cold.txt $P/r06_wide_pmc_dict_10k_k10000.txt
cp $S/wide_pmc_light.txt $P/r06_wide_pmc_dict_1k_k1000_zipped.txt
cp $S/stats_wide/stats_kernel_stats.csv $P/r06_wide_kernel_stats.csv
cp $S/trace_timed_region_wide.txt $P/r06_wide_trace_timed_region.txt
cp $S/micro_lds.log $P/r06_micro_lds.log
cp $S/bench_slow_wide.jsonl $P/r06_bench_slow_wide.jsonl
cp $S/ragged_cases.log $P/r06_ragged_cases.log
for c in urls loglines; do cp $S/ragged_pmc_${c}_v1.txt $P/r06_ragged_pmc_${c}.txt; cp $S/ragged_pmc_${c}_v0.txt $P/r06_stream_pmc_${c}.txt; done
for f in prefix suffix half_final counting actions long_strings capture pair host_call_latency shim; do cp $S/$f.log $P/r06_final_$f.log; done
for f in counting_variants capture_variants half_final_variants counting_many_regexps slow_ragged slow_ragged_nostats; do cp $S/$f.log $P/r06_final_$f.log; done
cp $S/counting_kernel_stats.txt $P/r06_counting_kernel_stats.txt
[ -s $S/actions_wide.jsonl ] && cp $S/actions_wide.jsonl $P/r06_actions_wide.jsonl
[ -s $S/selftest_cost.txt ] && cp $S/selftest_cost.txt $P/r06_selftest_cost.txt
[ -s $S/ranking_quality.txt ] && cp $S/ranking_quality.txt $P/r06_ranking_quality.txt
[ -s $S/sampler_probe.txt ] && cp $S/sampler_probe.txt $P/r06_sampler_probe.txt
[ -s $S/stress_dict_60.log ] && cp $S/stress_dict_60.log $P/r06_stress_dict_60.log
[ -f $S/tsan_summary.txt ] && cat $S/tsan_pytest.log $S/tsan_summary.txt > $P/r06_tsan_gpu_summary.txt
cp gpurun_out/final_r06.log $P/r06_final_run.log
python tools/fill_design_tables.py
