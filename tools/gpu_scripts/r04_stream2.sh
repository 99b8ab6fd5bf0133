#!/bin/bash
# round 4, experiment 2: the stream kernel after the window-address fix -- a quick smoke under a short timeout first (a
# kernel that faults or hangs must not eat the budget again), then the parity tests, the A/B timings, PMC of both kernels
set -u
export PYTHONUNBUFFERED=1 TMPDIR=/tmp PYTHONPATH=.
OUT=gpurun_out/r04s2
mkdir -p $OUT
echo "== smoke"
timeout 150 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "stream_kernel_vs_oracle and set_a and urls" 2>&1 | tail -6 | tee $OUT/smoke.log
if ! grep -q " passed" $OUT/smoke.log || grep -q "failed\|error" $OUT/smoke.log; then echo "SMOKE FAILED: stopping"; exit 1; fi
echo "== parity: stream + ragged + routing tests"
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "stream or ragged_kernel" 2>&1 | tail -8 | tee $OUT/pytest_stream.log
echo "== timings (variant 1 = ragged kernel only, 0 = default routing)"
for c in urls loglines uniform2k uniform8k fixed4096 urls_x4 loglines_x4; do
  for v in 1 0; do
    echo -n "variant=$v: "; PIRE_HIP_RAGGED_VARIANT=$v timeout 90 python tools/ragged_case.py $c 3 2>&1 | grep "^ragged\|^stream\|fault\|Error\|error" | tail -2
  done
done | tee $OUT/ragged_cases_ab.log
if ! grep -q " passed" $OUT/pytest_stream.log || grep -q "failed" $OUT/pytest_stream.log; then echo "PARITY FAILED: no PMC"; exit 1; fi
echo "== PMC"
i=0
for set in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum" \
           "SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD" \
           "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TA_TA_BUSY_sum GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  for c in urls loglines; do for v in 1 0; do
    PIRE_HIP_RAGGED_VARIANT=$v timeout 90 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/pmc_${c}_v${v}/p$i -o pmc -- python tools/ragged_case.py $c 1 > $OUT/pmc_${c}_v${v}_$i.log 2>&1 || echo "pmc pass $i $c $v failed"
  done; done
done
for c in urls loglines; do for v in 1 0; do python tools/summarize_pmc.py $OUT/pmc_${c}_v${v} > $OUT/pmc_summary_${c}_v${v}.txt 2>&1; done; done
grep -A14 "ScanStream" $OUT/pmc_summary_urls_v0.txt | head -20
find $OUT -name "*.csv" -size +1M -delete; find $OUT -name "*.db" -delete
du -sh $OUT
