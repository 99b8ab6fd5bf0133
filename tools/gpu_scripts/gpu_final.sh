#!/bin/bash
# Round-end evidence run on the GPU box: parity tests, smoke, bench, rocprofv3 stats + PMC passes, PCIe-inclusive rate.
set -u
export PYTHONUNBUFFERED=1
export TMPDIR=/tmp
mkdir -p gpurun_out/final
echo "== pytest -m gpu"
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -6 | tee gpurun_out/final/pytest_gpu.log
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee gpurun_out/final/smoke.log
echo "== bench (default)"
timeout 900 python bench.py 2>&1 | tail -1 | tee gpurun_out/final/bench_n1.json
echo "== bench C2 single pattern (parity config, informational)"
timeout 600 python bench.py --set c2_single --steps 10 --warmup 2 --cpu-sample-log2 18 2>&1 | tail -1 | cut -c1-700 | tee gpurun_out/final/bench_c2.json
echo "== rocprofv3 kernel stats of the bench command"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/final/stats -o stats -- python bench.py --no-cpu > gpurun_out/final/stats.log 2>&1
cat gpurun_out/final/stats/stats_kernel_stats.csv | head -5
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD" \
           "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $set --kernel-trace --output-format csv -d gpurun_out/final/pmc$i -o pmc -- python bench.py --steps 5 --warmup 1 --no-cpu > gpurun_out/final/pmc$i.log 2>&1 || echo "pmc pass $i failed"
done
python tools/summarize_pmc.py gpurun_out/final 2>&1 | tee gpurun_out/final/pmc_summary.txt | grep -A20 ScanTiled
echo "== PCIe-inclusive (host-pointer mode)"
timeout 600 python - <<'PY' 2>&1 | tee gpurun_out/final/pcie.log
import time, numpy as np, torch
import pire_amd
from oracle import binding as ob
from tests import helpers as H
big=[b for b in H.big_sets() if b["name"]=="set_a"][0]
t=pire_amd.Table(H.load_blob(big["blob"])); t.upload()
n,L=1<<16,4096
data=ob.corpus_fill(0x5EED5EED,0,n,L,H.plants_for(big),threads=32)
t.run_strided_host(data[:1024])
best=1e9
for _ in range(3):
    t0=time.perf_counter(); idx,fin=t.run_strided_host(data); dt=time.perf_counter()-t0; best=min(best,dt)
print("PCIe-inclusive host-pointer mode: %d x %d B (%.0f MiB pageable host memory): %.1f ms -> %.2f GB/s" % (n,L,n*L/2**20,best*1e3,n*L/best/1e9))
PY
echo "== ragged batches through pire_hip_run (offsets), set_a table, after two adapt() passes"
for c in urls loglines uniform2k uniform8k fixed4096; do PYTHONPATH=. timeout 120 python tools/ragged_case.py $c 3 2>&1 | grep "^ragged"; PYTHONPATH=. timeout 120 python tools/ragged_case.py $c 3 generic 2>&1 | grep "^generic"; done | tee gpurun_out/final/ragged_cases.log
echo "== secondary kernels"
PYTHONPATH=. timeout 200 python tools/prefix_case.py 2>&1 | grep "Prefix" | tee gpurun_out/final/prefix.log
PYTHONPATH=. timeout 200 python tools/half_final_case.py half_5 2>&1 | grep "half_final\|reference" | tee gpurun_out/final/half_final.log
PYTHONPATH=. timeout 200 python tools/counting_case.py count_glued3_advanced 2>&1 | grep "counting\|reference" | tee gpurun_out/final/counting.log
PYTHONPATH=. timeout 200 python tools/actions_case.py 2>&1 | grep -v amdgpu.ids > gpurun_out/final/actions.log; tail -4 gpurun_out/final/actions.log | cut -c1-200
LONG_TOTAL_LOG2=30 PYTHONPATH=. timeout 120 python tools/long_case.py 2>&1 | grep -v "amdgpu.ids\|pire_hip segm" | tee gpurun_out/final/long_strings.log | cut -c1-200
PYTHONPATH=. timeout 200 python tools/long_half_final.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/final/long_half_final.log | cut -c1-200
PYTHONPATH=. timeout 200 python tools/capture_case.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/final/capture.log | cut -c1-200
timeout 300 python bench.py --set slow_x40_utf8 --log2-strings 18 --len 16384 --steps 5 --warmup 1 2>&1 | tail -1 | tee gpurun_out/final/bench_c5b.json | cut -c1-200
tests/cpp/bin/shim_test 2>&1 | tail -1 | tee gpurun_out/final/shim.log
echo "== pigrep example"
examples/bin/pigrep_hip -i "lds.*bytes" DESIGN.md | head -2
