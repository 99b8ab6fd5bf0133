#!/bin/bash
# round 4, follow-up of the evidence run: the kernel trace of the bench command without the from-idle leg (so that the
# last 20 launches ARE the timed region), and the ASan / UBSan build of the host side on the GPU box
set -u
export PYTHONUNBUFFERED=1 TMPDIR=/tmp PYTHONPATH=.
OUT=gpurun_out/final_r04
mkdir -p $OUT
rm -rf $OUT/stats
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o stats -- python bench.py --steps 20 --warmup 5 --no-cpu --cold-launches 0 > $OUT/stats.log 2>&1
head -3 $OUT/stats/stats_kernel_stats.csv
python tools/summarize_trace.py $OUT/stats 20 | tee $OUT/trace_timed_region.txt
tail -1 $OUT/stats.log | cut -c1-200
echo "== bench again (box-to-box and run-to-run spread)"
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu 2>&1 | tail -1 > $OUT/bench_n1_again.json; python -c "
import json; d=json.load(open('$OUT/bench_n1_again.json')); print('value', d['value'], 'frac', d['roofline']['frac'], 'kernel avg', d['roofline']['kernel_avg_ms'], 'cold', d['cold_start']['value'])"
echo "== ASan on the GPU box"
make -C pire_amd/csrc -j16 asan > $OUT/asan_build.log 2>&1
RT=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | tail -1)
export ASAN_OPTIONS=detect_leaks=0:halt_on_error=1:protect_shadow_gap=0:verify_asan_link_order=0 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
LD_PRELOAD=$RT PIRE_HIP_LIB=pire_amd/libpire_hip_asan.so timeout 300 python -c "
import torch
print('torch cuda', torch.cuda.is_available())
import pire_amd
from pire_amd import binding as pb
print('lib', pb.lib_path(), 'devices', pire_amd.device_count())
" > $OUT/asan_probe.log 2>&1; echo "probe rc=$?"; tail -5 $OUT/asan_probe.log
LD_PRELOAD=$RT PIRE_HIP_LIB=pire_amd/libpire_hip_asan.so timeout 900 python -m pytest tests/test_default_config.py tests/test_abi.py tests/test_host_staging.py tests/test_multi_gpu.py -m gpu -x -q -p no:cacheprovider > $OUT/asan_gpu.log 2>&1; echo "asan pytest rc=$?"
tail -6 $OUT/asan_gpu.log; echo "sanitizer reports: $(grep -c 'AddressSanitizer\|runtime error:' $OUT/asan_gpu.log)"
