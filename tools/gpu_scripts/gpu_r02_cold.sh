#!/bin/bash
# the a-priori ranking on the GPU: value_before_adapt against value for the three big sets + C++ text; tiled parity tests
set -u
export PYTHONUNBUFFERED=1 TMPDIR=/tmp PYTHONPATH=.
OUT=gpurun_out/r02cold
mkdir -p $OUT
P='import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(sys.argv[1], "value", d["value"], "before adapt", d.get("value_before_adapt"), "kernel avg", r["kernel_avg_ms"], "before", r["kernel_avg_ms_before_adapt"], "promoted", d["config"]["table"]["rows_promoted_by_adapt"])'
for st in set_a set_b set_d c2_single; do
  timeout 300 python bench.py --set $st --no-cpu --steps 20 --warmup 5 2>&1 | tail -1 | python -c "$P" "$st synthetic 2^20x4096" | tee -a $OUT/cold.log
  timeout 300 python bench.py --set $st --corpus cxx --no-cpu --steps 20 --warmup 5 2>&1 | tail -1 | python -c "$P" "$st C++text 2^20x4096" | tee -a $OUT/cold.log
done
timeout 300 python bench.py --set set_b --len 16384 --log2-strings 18 --no-cpu --steps 10 --warmup 3 2>&1 | tail -1 | python -c "$P" "set_b synthetic 2^18x16384" | tee -a $OUT/cold.log
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_random_scanners.py -m gpu -q -x 2>&1 | tail -3 | tee $OUT/pytest.log
