#!/bin/bash
# round 4, experiment 3: stream kernel v2 (lane masks from the scalar unit, every CU, ends read ahead) + RCCL on one GPU +
# the default configuration under concurrent adaptation + set_d's counters
set -u
export PYTHONUNBUFFERED=1 TMPDIR=/tmp PYTHONPATH=.
OUT=gpurun_out/r04s3
mkdir -p $OUT
echo "== smoke"
timeout 150 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "stream_kernel_vs_oracle and set_a and (urls or tiny)" 2>&1 | tail -4 | tee $OUT/smoke.log
if ! grep -q " passed" $OUT/smoke.log || grep -q "failed\|error" $OUT/smoke.log; then echo "SMOKE FAILED: stopping"; exit 1; fi
echo "== parity: stream + ragged + routing + adaptation tests"
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "stream or ragged_kernel or auto_adapt" 2>&1 | tail -6 | tee $OUT/pytest_stream.log
echo "== timings: stream kernel (default routing)"
for c in urls loglines uniform2k uniform8k fixed4096 urls_x4 loglines_x4 urls_16k urls_64k urls_256k loglines_64k; do
  timeout 90 python tools/ragged_case.py $c 3 2>&1 | grep "^ragged\|^stream\|fault\|Error\|error" | tail -1
done | tee $OUT/ragged_cases_stream.log
echo "== timings: ragged kernel on the small batches"
for c in urls_16k urls_64k urls_256k loglines_64k; do
  PIRE_HIP_RAGGED_VARIANT=1 timeout 90 python tools/ragged_case.py $c 3 2>&1 | grep "^ragged\|^stream\|fault\|Error\|error" | tail -1
done | tee $OUT/ragged_cases_small_v1.log
echo "== default configuration, concurrent adaptation; RCCL as a one-rank communicator"
timeout 600 python -m pytest tests/test_default_config.py tests/test_multi_gpu.py -m gpu -x -q 2>&1 | tail -6 | tee $OUT/pytest_default_multi.log
echo "== bench --force-dist (world of one rank over RCCL)"
timeout 300 python bench.py --force-dist --backend nccl --steps 10 --warmup 3 --no-cpu 2>$OUT/force_dist.err | tail -1 > $OUT/bench_force_dist_nccl.json; cut -c1-250 $OUT/bench_force_dist_nccl.json; grep -o '"reduce_backend": "[^"]*"' $OUT/bench_force_dist_nccl.json; tail -3 $OUT/force_dist.err
echo "== 8 ranks over gloo on this one GPU (launch path dry run)"
timeout 600 python bench.py --gpus 8 --backend gloo --log2-strings 12 --steps 3 --warmup 1 --settle 2 --no-cpu --cold-launches 0 2>$OUT/gloo8.err | tail -1 > $OUT/bench_8ranks_gloo.json; cut -c1-200 $OUT/bench_8ranks_gloo.json; grep -o '"per_rank_GBps": \[[^]]*\]' $OUT/bench_8ranks_gloo.json; tail -2 $OUT/gloo8.err
echo "== set_d: bench line + counters"
timeout 300 python bench.py --set set_d --steps 20 --warmup 5 --no-cpu 2>/dev/null | tail -1 > $OUT/bench_set_d.json; python - <<'PY'
import json
d=json.load(open("gpurun_out/r04s3/bench_set_d.json"))
print("set_d value", d["value"], "before_adapt", d["value_before_adapt"], "frac", d["roofline"]["frac"], "traps", d.get("traps"), "cold", d.get("cold_start"))
PY
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu 2>/dev/null | tail -1 > $OUT/bench_set_a.json; python - <<'PY'
import json
d=json.load(open("gpurun_out/r04s3/bench_set_a.json"))
print("set_a value", d["value"], "before_adapt", d["value_before_adapt"], "frac", d["roofline"]["frac"], "traps", d.get("traps"), "cold", d.get("cold_start"))
PY
i=0
for set in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD" "FETCH_SIZE"; do
  i=$((i+1))
  for st in set_d set_a; do
    timeout 120 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/pmc_$st/p$i -o pmc -- python bench.py --set $st --steps 5 --warmup 1 --settle 10 --no-cpu --cold-launches 0 > $OUT/pmc_${st}_$i.log 2>&1 || echo "pmc $st $i failed"
  done
done
for st in set_d set_a; do python tools/summarize_pmc.py $OUT/pmc_$st > $OUT/pmc_summary_$st.txt 2>&1; echo "-- $st"; grep -A12 "ScanTiled" $OUT/pmc_summary_$st.txt | head -12; done
find $OUT -name "*.csv" -size +1M -delete; find $OUT -name "*.db" -delete
du -sh $OUT
