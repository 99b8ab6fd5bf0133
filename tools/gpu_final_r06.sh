#!/bin/bash
# Round-6 evidence run on the GPU box.  Outputs under gpurun_out/final_r06; tools/collect_final_r06.sh copies what is
# judged into profiles/.  Every step under its own timeout; the long parity samples are bounded (no 32 GiB host arrays).
set -u
export PYTHONUNBUFFERED=1 TMPDIR=/tmp PYTHONPATH=.
OUT=gpurun_out/final_r06
mkdir -p $OUT
echo "== pytest -m gpu"
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -4 | tee $OUT/pytest_gpu.log
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee $OUT/smoke.log
echo "== PMC passes of the bench command FIRST (bench.py refuses a traffic file of other kernel sources)"
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD" \
           "TA_TA_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/pmc$i -o pmc -- python bench.py --steps 5 --warmup 1 --settle 10 --no-cpu --no-adapt --cold-launches 0 > $OUT/pmc$i.log 2>&1 || echo "pmc pass $i failed"
done
python tools/summarize_pmc.py $OUT --last 5 2>&1 > $OUT/pmc_summary.txt; grep -A16 ScanTiled $OUT/pmc_summary.txt | head -18
python tools/make_pmc_json.py $OUT/pmc_summary.txt profiles/r06_pmc_traffic.json "set_a 2^20 x 4096" "python bench.py --steps 5 --warmup 1 --settle 10 --no-cpu --no-adapt --cold-launches 0; tools/gpu_final_r06.sh" > /dev/null && cp profiles/r06_pmc_traffic.json $OUT/pmc_traffic.json
echo "== bench, the driver's command"
timeout 900 python bench.py --steps 20 --warmup 5 2>&1 | tail -1 > $OUT/bench_n1.json; cut -c1-330 $OUT/bench_n1.json
python - <<'PY'
import json
d=json.load(open("gpurun_out/final_r06/bench_n1.json"))
print("roofline", d["roofline"]["frac"], "traffic", d["roofline"]["traffic"], "cold_start", d.get("cold_start"), "traps", d.get("traps"), "cpu parity", d["cpu_baseline"]["parity_vs_gpu"], d["cpu_baseline"]["value"])
PY
echo "== bench, defaults (50 steps after 20)"
timeout 900 python bench.py --no-cpu 2>&1 | tail -1 > $OUT/bench_n1_defaults.json; cut -c1-200 $OUT/bench_n1_defaults.json
echo "== rocprofv3 kernel trace + stats of the bench command"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o stats -- python bench.py --steps 20 --warmup 5 --no-cpu --cold-launches 0 > $OUT/stats.log 2>&1   # (no from-idle leg: the LAST 20 launches of the process are the timed region)
head -3 $OUT/stats/stats_kernel_stats.csv
python tools/summarize_trace.py $OUT/stats 20 | tee $OUT/trace_timed_region.txt
echo "== world of one rank over RCCL; 8 ranks over gloo (launch path)"
timeout 300 python bench.py --force-dist --backend nccl --steps 20 --warmup 5 --no-cpu 2>/dev/null | grep '^{' | tail -1 > $OUT/bench_force_dist_nccl.json; grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|"reduce_backend": "[^"]*"\|"counter_reduce_ms": [0-9.]*' $OUT/bench_force_dist_nccl.json | head -4
timeout 300 python bench.py --force-dist --backend nccl --steps 20 --warmup 5 --no-cpu --reduce-every-step 2>/dev/null | grep '^{' | tail -1 > $OUT/bench_force_dist_nccl_every_step.json; grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' $OUT/bench_force_dist_nccl_every_step.json | head -2
timeout 900 python bench.py --gpus 8 --backend gloo --steps 5 --warmup 2 --settle 5 --no-cpu --cold-launches 0 2>/dev/null | grep '^{' | tail -1 > $OUT/bench_8ranks_gloo.json; grep -o '"per_rank_[A-Za-z_]*": \[[^]]*\]' $OUT/bench_8ranks_gloo.json
timeout 900 python bench.py --gpus 2 --backend gloo --c4 --steps 3 --warmup 1 --settle 3 --no-cpu --cold-launches 0 2>/dev/null | grep '^{' | tail -1 > $OUT/bench_c4_2ranks_gloo.json; grep -o '"per_rank_[A-Za-z_]*": \[[^]]*\]' $OUT/bench_c4_2ranks_gloo.json
timeout 600 python bench.py --gpus 2 --backend gloo --log2-strings 18 --steps 5 --warmup 2 --no-cpu 2>/dev/null | tail -1 > $OUT/bench_2ranks_gloo.json; cut -c1-160 $OUT/bench_2ranks_gloo.json
echo "== other configs (CPU samples: the whole batch where it fits the host)"
timeout 300 python bench.py --set c1_nonreloc 2>/dev/null | tail -1 > $OUT/bench_c1_nonreloc.json; cut -c1-260 $OUT/bench_c1_nonreloc.json
timeout 600 python bench.py --set c2_single --steps 20 --warmup 5 --cpu-sample-log2 20 2>&1 | tail -1 > $OUT/bench_c2.json; cut -c1-200 $OUT/bench_c2.json
timeout 1200 python bench.py --set set_b --len 16384 --log2-strings 20 --steps 10 --warmup 3 --settle 20 --cpu-sample-log2 20 2>&1 | tail -1 > $OUT/bench_c5a.json; cut -c1-200 $OUT/bench_c5a.json
timeout 900 python bench.py --set slow_x40_utf8 --len 16384 --log2-strings 20 --steps 5 --warmup 1 --cpu-sample-log2 20 2>&1 | tail -1 > $OUT/bench_c5b.json; cut -c1-200 $OUT/bench_c5b.json
timeout 600 python bench.py --set set_d --steps 20 --warmup 5 --cpu-sample-log2 18 2>&1 | tail -1 > $OUT/bench_set_d.json; cut -c1-200 $OUT/bench_set_d.json
timeout 900 python bench.py --c4 --steps 10 --warmup 3 2>&1 | tail -1 > $OUT/bench_c4_shard.json; cut -c1-200 $OUT/bench_c4_shard.json
timeout 600 python bench.py --corpus cxx --steps 20 --warmup 5 --cpu-sample-log2 16 2>&1 | tail -1 > $OUT/bench_cxx_records.json; cut -c1-200 $OUT/bench_cxx_records.json
timeout 600 python bench.py --corpus cxx --one-string --log2-strings 18 --steps 10 --warmup 3 --settle 10 --no-cpu 2>&1 | tail -1 > $OUT/bench_cxx_one_string.json; cut -c1-200 $OUT/bench_cxx_one_string.json
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/final_r06/bench_c*.json"))+["gpurun_out/final_r06/bench_set_d.json"]:
    try:
        d=json.load(open(f)); c=d.get("cpu_baseline",{})
        print(f.split("/")[-1], d["value"], d.get("roofline",{}).get("frac"), "cpu parity", c.get("parity_vs_gpu"), c.get("sample","")[:60])
    except Exception as e: print(f, "unreadable", e)
PY
echo "== wide working sets (config 5 as north_star means it): bench lines, the curve, counters of the wide walk"
for pt in "set_b_mix mix" "dict_1k k32" "dict_1k k128" "dict_1k k512" "dict_1k k1000" "dict_10k k32" "dict_10k k512" "dict_10k k2048" "dict_10k k10000" "dict_utf8_1k k32" "dict_utf8_1k k1000" "dict_utf8_5k k512" "dict_utf8_5k k5000"; do set -- $pt
  timeout 600 python bench.py --set $1 --corpus $2 --steps 20 --warmup 5 2>/dev/null | grep '^{' | tail -1 > $OUT/bench_$1_$2.json
done
timeout 900 python bench.py --set dict_10k --corpus k10000 --len 16384 --log2-strings 20 --steps 10 --warmup 3 --settle 20 2>/dev/null | grep '^{' | tail -1 > $OUT/bench_c5_dict_10k_16k.json
timeout 600 python bench.py --set dict_1k --corpus k128 --walk 1 --steps 20 --warmup 5 --no-cpu 2>/dev/null | grep '^{' | tail -1 > $OUT/bench_dict_1k_k128_dense_rows.json
timeout 600 python bench.py --set dict_10k --corpus k10000 --zip 1 --steps 20 --warmup 5 --no-cpu 2>/dev/null | grep '^{' | tail -1 > $OUT/bench_dict_10k_k10000_plain_rows.json
timeout 600 python bench.py --set dict_1k --corpus k1000 --zip 1 --steps 20 --warmup 5 --no-cpu 2>/dev/null | grep '^{' | tail -1 > $OUT/bench_dict_1k_k1000_plain_rows.json
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/final_r06/bench_dict*.json")+glob.glob("gpurun_out/final_r06/bench_set_b_mix*.json")+glob.glob("gpurun_out/final_r06/bench_c5_dict*.json")):
    try:
        d=json.load(open(f)); w=d.get("working_set",{}); c=d.get("cpu_baseline",{})
        print(f.split("/")[-1], d["value"], d["roofline"]["kernel"].split("::")[-1][:40], "frac", d["roofline"]["frac"], "visited", w.get("distinct_states_visited"), "twice", d["traps"].get("wide_walk_wave_chunk_share_walked_twice"), "parity", c.get("parity_vs_gpu"), d.get("parity_of_repeats"))
    except Exception as e: print(f, "unreadable", e)
PY
timeout 1500 python tools/wide_case.py --log2-strings 20 --points set_b_mix:mix,dict_1k:k32,dict_1k:k128,dict_1k:k512,dict_1k:k1000,dict_10k:k32,dict_10k:k512,dict_10k:k2048,dict_10k:k10000,dict_utf8_1k:k32,dict_utf8_1k:k1000,dict_utf8_5k:k512,dict_utf8_5k:k5000,blacklist_1k:urls,blacklist_10k:urls --out $OUT/wide_curve.jsonl > $OUT/wide_curve.log 2>&1; echo "wide_case rc=$?"
i=0
for set in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD" "SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_WAVES" "FETCH_SIZE" "WRITE_SIZE" \
           "TA_TA_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/wpmc_fit/p$i -o pmc -- python bench.py --set dict_1k --corpus k128 --steps 5 --warmup 1 --settle 10 --no-cpu --cold-launches 0 > $OUT/wpmc_fit_$i.log 2>&1 || echo "wide pmc fit pass $i failed"
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/wpmc_light/p$i -o pmc -- python bench.py --set dict_1k --corpus k1000 --steps 5 --warmup 1 --settle 10 --no-cpu --cold-launches 0 > $OUT/wpmc_light_$i.log 2>&1 || echo "wide pmc light pass $i failed"
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/wpmc_cold/p$i -o pmc -- python bench.py --set dict_10k --corpus k10000 --steps 5 --warmup 1 --settle 10 --no-cpu --cold-launches 0 > $OUT/wpmc_cold_$i.log 2>&1 || echo "wide pmc cold pass $i failed"
done
for w in fit light cold; do python tools/summarize_pmc.py $OUT/wpmc_$w --last 5 > $OUT/wide_pmc_$w.txt 2>&1; done; grep -A16 ScanWide $OUT/wide_pmc_fit.txt | head -18
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_wide -o stats -- python bench.py --set dict_1k --corpus k128 --steps 20 --warmup 5 --no-cpu --cold-launches 0 > $OUT/stats_wide.log 2>&1
python tools/summarize_trace.py $OUT/stats_wide 20 ScanWide | tee $OUT/trace_timed_region_wide.txt
for st in set_d set_a c2_single; do timeout 300 python tools/micro_lds.py $st --waves 16 --steps 1024 --reps 300 --out $OUT 2>&1 | grep -v amdgpu.ids; done > $OUT/micro_lds.log 2>&1; rm -f $OUT/micro_trace.bin $OUT/micro_rows.bin; grep "indep. u8\|chain u8\|^set" $OUT/micro_lds.log | cut -c1-150
echo "== offset batches: ragged kernel (variant 1) against the default routing (stream kernel), same box"
for c in urls loglines uniform2k uniform8k fixed4096 urls_x4 loglines_x4 uniform2k_x4; do
  for v in 1 0; do
    echo -n "variant=$v: "; PIRE_HIP_RAGGED_VARIANT=$v timeout 120 python tools/ragged_case.py $c 3 2>&1 | grep "^ragged\|^stream\|fault\|rror" | tail -1
  done
done | tee $OUT/ragged_cases.log
echo "== PMC of both kernels on the URL and log-line batches"
i=0
for set in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum" \
           "SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY" \
           "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TA_TA_BUSY_sum GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  for c in urls loglines; do for v in 1 0; do
    PIRE_HIP_RAGGED_VARIANT=$v timeout 90 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/rpmc_${c}_v${v}/p$i -o pmc -- python tools/ragged_case.py $c 1 > $OUT/rpmc_${c}_v${v}_$i.log 2>&1 || echo "pmc pass $i $c $v failed"
  done; done
done
for c in urls loglines; do for v in 1 0; do python tools/summarize_pmc.py $OUT/rpmc_${c}_v${v} > $OUT/ragged_pmc_${c}_v${v}.txt 2>&1; done; done
echo "== secondary kernels"
{ for lg in 18 20 21; do echo "-- 2^$lg strings, table as created (never adapted)"; PREFIX_LOG2_STRINGS=$lg PREFIX_SETTLE=30 timeout 300 python tools/prefix_case.py 2>&1 | grep "Prefix\|adapt"; echo "-- 2^$lg strings, table adapted by the searches' own samples"; PREFIX_LOG2_STRINGS=$lg PREFIX_SETTLE=30 timeout 300 python tools/prefix_case.py adapt_by_prefix 2>&1 | grep "Prefix\|adapt"; done; } | tee $OUT/prefix.log | cut -c1-200
timeout 300 python tools/suffix_case.py 2>&1 | grep "Suffix" | tee $OUT/suffix.log | cut -c1-220
timeout 200 python tools/half_final_case.py half_5 2>&1 | grep "half_final\|reference" | tee $OUT/half_final.log | cut -c1-200
timeout 300 python tools/counting_case.py count_glued3_advanced 2>&1 | grep "counting\|reference" | tee $OUT/counting.log | cut -c1-220
timeout 300 python tools/capture_case.py 2>&1 | grep -v amdgpu.ids | tee $OUT/capture.log | cut -c1-220
timeout 200 python tools/actions_case.py 2>&1 | grep -v amdgpu.ids > $OUT/actions.log; tail -4 $OUT/actions.log | cut -c1-200
timeout 600 python tools/actions_wide_case.py 2>&1 | grep "^{" > $OUT/actions_wide.jsonl; wc -l $OUT/actions_wide.jsonl
{ LONG_TOTAL_LOG2=30 timeout 120 python tools/long_case.py 2>&1 | grep -v "amdgpu.ids\|pire_hip segm"; timeout 120 python tools/long_grep_case.py 2>&1 | grep "^grep"; } | tee $OUT/long_strings.log | cut -c1-200
timeout 200 python tools/pair_case.py 2>&1 | grep -v amdgpu.ids | tee $OUT/pair.log | cut -c1-200
for st in slow_x300 slow_x400_utf8; do timeout 400 python bench.py --set $st --log2-strings 16 --len 4096 --steps 5 --warmup 2 --cpu-sample-log2 10 2>&1 | tail -1 | cut -c1-1500; done > $OUT/bench_slow_wide.jsonl; cut -c1-200 $OUT/bench_slow_wide.jsonl
timeout 300 python tools/selftest_cost.py 2>&1 | grep "self-test" | tee $OUT/selftest_cost.txt | cut -c1-200
echo "== what adapt() learns: the ranking against the best one (oracle's lookup counts), rounds of scan + adapt()"
{ for a in "blacklist_1k" "blacklist_10k" "blacklist_1k dense"; do timeout 300 python tools/ranking_quality.py $a 2>&1 | grep "^after\|ideal"; done; } | cut -c1-200 | tee $OUT/ranking_quality.txt | grep "ideal\|after [18] x"
timeout 300 python tools/sampler_probe.py 2>&1 | grep "^states\|round" | cut -c1-260 > $OUT/sampler_probe.txt; tail -2 $OUT/sampler_probe.txt
echo "== random dictionaries through every kernel of the wide walk (tools/stress_dict.py, 60 seeds here; profiles/r06_stress_dict.log: 300)"
timeout 900 python tools/stress_dict.py 400 460 2>&1 | grep -v amdgpu.ids | tail -2 | tee $OUT/stress_dict_60.log
echo "== host-pointer mode"
timeout 300 python tools/host_call_latency.py 2>&1 | grep -v amdgpu.ids | tee $OUT/host_call_latency.log | tail -8
echo "== C++ shim and the pigrep example"
tests/cpp/bin/shim_test 2>&1 | tail -2 | tee $OUT/shim.log
echo "== counting / capturing scanners: round 3's kernels (variant 1) against the row kernels (default), same box; SlowScanner on ragged strings"
for name in count_glued3_advanced count0_advanced count0_basic; do
  for v in 1 0; do
    echo -n "variant=$v: "; PIRE_HIP_COUNTING_VARIANT=$v timeout 300 python tools/counting_case.py $name 2>&1 | grep "^counting\|parity" | tr '\n' ' ' | cut -c1-420; echo
  done
done | tee $OUT/counting_variants.log
for v in 1 0; do PIRE_HIP_COUNTING_VARIANT=$v timeout 400 python tools/capture_case.py 2>&1 | grep "^capture" | sed "s/^/variant=$v: /"; done | tee $OUT/capture_variants.log | cut -c1-220
for n in half_5 half_4 half_3 half_2; do for v in 1 0; do echo -n "variant=$v: "; PIRE_HIP_COUNTING_VARIANT=$v timeout 300 python tools/half_final_case.py $n 2>&1 | grep "^half_final\|parity" | tr '\n' ' ' | cut -c1-420; echo; done; done | tee $OUT/half_final_variants.log | cut -c1-200
timeout 300 python tools/debug/counting_rows_wide.py 2>&1 | grep -v amdgpu.ids | tee $OUT/counting_many_regexps.log | cut -c1-200
timeout 300 python tools/slow_ragged_case.py 2>&1 | grep "^slow" | tee $OUT/slow_ragged_nostats.log | cut -c1-220
PIRE_HIP_SLOW_STATS=1 timeout 300 python tools/slow_ragged_case.py 2>&1 | grep "^slow\|pire_hip slow" | tee $OUT/slow_ragged.log | cut -c1-220
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_counting -o stats -- python tools/counting_case.py count_glued3_advanced > /dev/null 2>&1; grep "Counting\|Order\|Length" $OUT/stats_counting/stats_kernel_stats.csv | cut -c1-200 | tee $OUT/counting_kernel_stats.txt
echo "== ThreadSanitizer build of the host side (make -C pire_amd/csrc tsan, shipped with the snapshot): the background adaptation under four host threads, the default policy's tests"
if [ -f pire_amd/libpire_hip_tsan.so ]; then
  mkdir -p $OUT/tsan
  LD_PRELOAD=/opt/rocm-7.2.0/lib/llvm/lib/clang/22/lib/linux/libclang_rt.tsan-x86_64.so PIRE_HIP_LIB=pire_amd/libpire_hip_tsan.so TSAN_OPTIONS="log_path=$OUT/tsan/report exitcode=0 report_signal_unsafe=0" timeout 1500 python -m pytest tests/test_background_adapt.py tests/test_default_config.py -m gpu -q -k "c_abi_alone or test_default_config" 2>&1 | tail -2 | tee $OUT/tsan_pytest.log
  python tools/summarize_tsan.py $OUT/tsan/report* 2>&1 | tail -3 | tee $OUT/tsan_summary.txt
  rm -rf $OUT/tsan
fi
find $OUT -name "*.csv" -size +1M -delete; find $OUT -name "*.db" -delete
du -sh $OUT
