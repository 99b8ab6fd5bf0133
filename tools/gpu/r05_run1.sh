#!/bin/bash
# round 5, GPU call 1: parity of the wide walk, the working-set curve, the headline line, three wide bench lines
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export PYTHONPATH=.
timeout 900 python -m pytest tests/test_wide.py -m gpu -x -q > gpurun_out/r05_wide_pytest.log 2>&1; echo "pytest wide rc=$?" | tee -a gpurun_out/r05_wide_pytest.log
tail -5 gpurun_out/r05_wide_pytest.log
timeout 900 python tools/wide_case.py --out gpurun_out/r05_wide_curve.jsonl > gpurun_out/r05_wide_curve.log 2>&1; echo "wide_case rc=$?"
cat gpurun_out/r05_wide_curve.jsonl | cut -c1-1500
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r05_bench_n1.json 2> gpurun_out/r05_bench_n1.err; echo "bench rc=$?"
cut -c1-600 gpurun_out/r05_bench_n1.json
for pt in "dict_1k k128" "dict_1k k1000" "dict_10k k10000"; do set -- $pt
  timeout 600 python bench.py --set $1 --corpus $2 --steps 20 --warmup 5 > gpurun_out/r05_bench_$1_$2.json 2> gpurun_out/r05_bench_$1_$2.err; echo "bench $1 $2 rc=$?"
  cut -c1-400 gpurun_out/r05_bench_$1_$2.json
done
