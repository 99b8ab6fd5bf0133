#!/bin/bash
# Round-3 evidence run on the GPU box: parity tests, smoke, bench (driver style) + rocprofv3 trace/stats of the same
# command + PMC passes, the other BASELINE configs at their stated sizes, the reference's corpus, ragged batches,
# secondary kernels, host mode, the C++ shim, 2 ranks.  Outputs under gpurun_out/final_r03; tools/collect_final_r03.sh
# copies what is judged into profiles/.
set -u
export PYTHONUNBUFFERED=1 TMPDIR=/tmp PYTHONPATH=.
OUT=gpurun_out/final_r03
mkdir -p $OUT
echo "== pytest -m gpu"
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -4 | tee $OUT/pytest_gpu.log
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee $OUT/smoke.log
echo "== bench, the driver's command"
timeout 900 python bench.py --steps 20 --warmup 5 2>&1 | tail -1 > $OUT/bench_n1.json; cut -c1-330 $OUT/bench_n1.json
echo "== bench, defaults (50 steps after 20)"
timeout 900 python bench.py --no-cpu 2>&1 | tail -1 > $OUT/bench_n1_defaults.json; cut -c1-200 $OUT/bench_n1_defaults.json
echo "== bench without the clock-settling phase (round 2's flow: the timed launches fall into the power transient)"
timeout 900 python bench.py --no-cpu --steps 20 --warmup 5 --settle 0 2>&1 | tail -1 > $OUT/bench_n1_settle0.json; cut -c1-200 $OUT/bench_n1_settle0.json
echo "== rocprofv3 kernel trace + stats of the bench command"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o stats -- python bench.py --steps 20 --warmup 5 --no-cpu > $OUT/stats.log 2>&1
head -3 $OUT/stats/stats_kernel_stats.csv
python tools/summarize_trace.py $OUT/stats 20 | tee $OUT/trace_timed_region.txt
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD" \
           "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/pmc$i -o pmc -- python bench.py --steps 5 --warmup 1 --settle 10 --no-cpu --no-adapt > $OUT/pmc$i.log 2>&1 || echo "pmc pass $i failed"
done
python tools/summarize_pmc.py $OUT 2>&1 > $OUT/pmc_summary.txt; grep -A20 ScanTiled $OUT/pmc_summary.txt | head -22
echo "== other configs"
timeout 600 python bench.py --set c2_single --steps 20 --warmup 5 --cpu-sample-log2 18 2>&1 | tail -1 > $OUT/bench_c2.json; cut -c1-200 $OUT/bench_c2.json
timeout 900 python bench.py --set set_b --len 16384 --log2-strings 20 --steps 10 --warmup 3 --settle 20 --cpu-sample-log2 16 2>&1 | tail -1 > $OUT/bench_c5a.json; cut -c1-200 $OUT/bench_c5a.json
timeout 900 python bench.py --set slow_x40_utf8 --len 16384 --log2-strings 20 --steps 5 --warmup 1 --cpu-sample-log2 20 2>&1 | tail -1 > $OUT/bench_c5b.json; cut -c1-200 $OUT/bench_c5b.json
timeout 600 python bench.py --set set_d --steps 20 --warmup 5 --no-cpu 2>&1 | tail -1 > $OUT/bench_set_d.json; cut -c1-200 $OUT/bench_set_d.json
timeout 600 python bench.py --log2-strings 23 --steps 5 --warmup 2 --settle 8 --cpu-sample-log2 20 2>&1 | tail -1 > $OUT/bench_c4_shard.json; cut -c1-200 $OUT/bench_c4_shard.json
echo "== the reference's corpus (tools/bench/test_file repeated): 4 KiB records, and the whole text as ONE string"
timeout 600 python bench.py --corpus cxx --steps 20 --warmup 5 --cpu-sample-log2 16 2>&1 | tail -1 > $OUT/bench_cxx_records.json; cut -c1-200 $OUT/bench_cxx_records.json
timeout 600 python bench.py --corpus cxx --one-string --log2-strings 18 --steps 10 --warmup 3 --settle 10 --no-cpu 2>&1 | tail -1 > $OUT/bench_cxx_one_string.json; cut -c1-200 $OUT/bench_cxx_one_string.json
echo "== two ranks on this box's GPU (gloo): the multi-rank control flow through the real kernel"
timeout 600 python bench.py --gpus 2 --backend gloo --log2-strings 18 --steps 5 --warmup 2 --no-cpu 2>&1 | tail -1 > $OUT/bench_2ranks_gloo.json; cut -c1-200 $OUT/bench_2ranks_gloo.json
echo "== ragged batches through pire_hip_run (offsets), set_a table, after two adapt() passes"
for c in urls loglines uniform2k uniform8k fixed4096; do timeout 120 python tools/ragged_case.py $c 3 2>&1 | grep "^ragged"; timeout 120 python tools/ragged_case.py $c 3 generic 2>&1 | grep "^generic"; done | tee $OUT/ragged_cases.log
echo "== secondary kernels"
timeout 200 python tools/prefix_case.py 2>&1 | grep "Prefix" | tee $OUT/prefix.log
timeout 200 python tools/half_final_case.py half_5 2>&1 | grep "half_final\|reference" | tee $OUT/half_final.log
timeout 200 python tools/counting_case.py count_glued3_advanced 2>&1 | grep "counting\|reference" | tee $OUT/counting.log
{ echo "== strings in the caller's order (pire_hip_config.no_length_order), the same batch:"; NO_LENGTH_ORDER=1 timeout 200 python tools/counting_case.py count_glued3_advanced 2>&1 | grep "^counting"; echo "== the 32-bit kernel alone, by length / in the caller's order:"; timeout 200 python tools/counting_case.py count_glued3_advanced generic 2>&1 | grep "^counting"; NO_LENGTH_ORDER=1 timeout 200 python tools/counting_case.py count_glued3_advanced generic 2>&1 | grep "^counting"; echo "== SlowScanner list kernel on a ragged batch:"; timeout 100 python tools/slow_ragged_case.py 2>&1 | grep "^slow"; } | tee -a $OUT/counting.log | cut -c1-200
timeout 200 python tools/actions_case.py 2>&1 | grep -v amdgpu.ids > $OUT/actions.log; tail -4 $OUT/actions.log | cut -c1-200
{ LONG_TOTAL_LOG2=30 timeout 120 python tools/long_case.py 2>&1 | grep -v "amdgpu.ids\|pire_hip segm"; timeout 120 python tools/long_grep_case.py 2>&1 | grep "^grep"; } | tee $OUT/long_strings.log | cut -c1-200
timeout 200 python tools/long_half_final.py 2>&1 | grep -v amdgpu.ids | tee $OUT/long_half_final.log | cut -c1-200
timeout 200 python tools/capture_case.py 2>&1 | grep -v amdgpu.ids | tee $OUT/capture.log | cut -c1-200
timeout 200 python tools/pair_case.py 2>&1 | grep -v amdgpu.ids | tee $OUT/pair.log | cut -c1-200
for st in slow_x300 slow_x400_utf8; do timeout 400 python bench.py --set $st --log2-strings 16 --len 4096 --steps 5 --warmup 2 --cpu-sample-log2 10 2>&1 | tail -1 | cut -c1-1500; done > $OUT/bench_slow_wide.jsonl; cut -c1-200 $OUT/bench_slow_wide.jsonl
echo "== host-pointer mode"
timeout 300 python tools/host_call_latency.py 2>&1 | grep -v amdgpu.ids | tee $OUT/host_call_latency.log
timeout 600 python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee $OUT/host_mode.log
import time, numpy as np
import pire_amd
from oracle import binding as ob
from tests import helpers as H
big=[b for b in H.big_sets() if b["name"]=="set_a"][0]
t=pire_amd.Table(H.load_blob(big["blob"])); t.upload()
for n,L in ((1<<16,4096),(1<<18,4096),(1<<19,4096)):
    data=ob.corpus_fill(0x5EED5EED,0,n,L,H.plants_for(big),threads=32)
    t.run_strided_host(data[:1024])
    best=1e9
    for _ in range(3):
        t0=time.perf_counter(); idx,fin=t.run_strided_host(data); dt=time.perf_counter()-t0; best=min(best,dt)
    print("host-pointer mode: %d x %d B (%.0f MiB pageable): %.1f ms -> %.2f GB/s" % (n,L,n*L/2**20,best*1e3,n*L/best/1e9))
PY
hipcc --offload-arch=gfx950 -O3 -o /tmp/micro_smallcall tools/micro_smallcall.hip && /tmp/micro_smallcall | tee $OUT/micro_smallcall.log | cut -c1-200
bash tools/gpu_scripts/r03_smalltrace.sh > /dev/null 2>&1; cp gpurun_out/small_call_timeline.log $OUT/small_call_timeline.log; rm -rf gpurun_out/smalltrace
echo "== C++ shim (host pointers, pinned, device-resident) and the pigrep example"
tests/cpp/bin/shim_test 2>&1 | tail -2 | tee $OUT/shim.log
examples/bin/pigrep_hip -i "lds.*bytes" DESIGN.md | head -2
echo "== power transient"
timeout 120 python tools/warmup_curve.py 300 2>&1 | grep -v amdgpu.ids | tee $OUT/warmup_curve.log
find $OUT -name "*.csv" -size +1M -delete; find $OUT -name "*.db" -delete
du -sh $OUT
