#!/bin/bash
# round 5 (the same run as round 4, plus the wide walk and the self-test): the host side under UBSan and under ThreadSanitizer ON THE GPU BOX (ASan cannot: ROCm's ASan runtime intercepts
# hsa_amd_memory_pool_allocate and aborts in a process that initialises HIP, profiles/r04_asan_gpu_probe.log): the
# concurrency tests of tests/test_default_config.py (eight host threads while the table adapts; adapt() under scans), the
# ABI tests, host staging, the multi-device runner
set -u
export PYTHONUNBUFFERED=1 TMPDIR=/tmp PYTHONPATH=.
OUT=gpurun_out/sanitizers_r05
mkdir -p $OUT
D=$(ls -d /opt/rocm/lib/llvm/lib/clang/*/lib/linux | tail -1)
ls -la pire_amd/libpire_hip_ubsan.so pire_amd/libpire_hip_tsan.so | cut -c1-120
echo "== UBSan"
LD_PRELOAD=$D/libclang_rt.ubsan_standalone-x86_64.so UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=0:log_path=$OUT/ubsan_report PIRE_HIP_LIB=pire_amd/libpire_hip_ubsan.so \
  timeout 900 python -m pytest tests/test_default_config.py tests/test_abi.py tests/test_host_staging.py tests/test_multi_gpu.py tests/test_segmented.py tests/test_counting.py tests/test_capture.py tests/test_slow.py tests/test_wide.py tests/test_selftest.py -m gpu -x -q -p no:cacheprovider > $OUT/ubsan_gpu.log 2>&1; echo "rc=$?"
tail -4 $OUT/ubsan_gpu.log; cat $OUT/ubsan_report.* 2>/dev/null | grep "runtime error" | sort | uniq -c | sort -rn | head -20; echo "UBSan reports: $(cat $OUT/ubsan_report.* 2>/dev/null | grep -c 'runtime error:')"
echo "== ThreadSanitizer"
LD_PRELOAD=$D/libclang_rt.tsan-x86_64.so TSAN_OPTIONS="halt_on_error=0 report_signal_unsafe=0 history_size=4 second_deadlock_stack=1 log_path=$OUT/tsan_report" PIRE_HIP_LIB=pire_amd/libpire_hip_tsan.so \
  timeout 1200 python -m pytest tests/test_default_config.py tests/test_selftest.py -m gpu -x -q -p no:cacheprovider > $OUT/tsan_gpu.log 2>&1; echo "rc=$?"
tail -4 $OUT/tsan_gpu.log; cat $OUT/tsan_report.* > $OUT/tsan_reports_all.txt 2>/dev/null; echo "TSan warnings: $(grep -c 'WARNING: ThreadSanitizer' $OUT/tsan_reports_all.txt)"
grep "WARNING: ThreadSanitizer" $OUT/tsan_reports_all.txt | sort | uniq -c | sort -rn | head
python tools/summarize_tsan.py $OUT/tsan_reports_all.txt | cut -c1-220 | tee $OUT/tsan_summary.txt | head -60
echo "-- frames of ours in the reports:"; grep -o "in pirehip::[A-Za-z_:<>]*\|in pire_hip_[a-z_]*\|in [A-Za-z:]*Staging[A-Za-z:]*" $OUT/tsan_reports_all.txt | sort | uniq -c | sort -rn | head -30
head -c 60000 $OUT/tsan_reports_all.txt > $OUT/tsan_reports_head.txt; rm -f $OUT/tsan_report.* $OUT/tsan_reports_all.txt
