#!/bin/bash
# refresh of the round-5 evidence that depends on wide.hip after its last change (visit samples behind the tile)
set -u
export PYTHONUNBUFFERED=1 TMPDIR=/tmp PYTHONPATH=.
OUT=gpurun_out/final_r05
mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -4 | tee $OUT/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee $OUT/smoke.log
timeout 600 python bench.py --steps 20 --warmup 5 2>&1 | tail -1 > $OUT/bench_n1.json; cut -c1-200 $OUT/bench_n1.json
for pt in "set_b_mix mix" "dict_1k k32" "dict_1k k128" "dict_1k k512" "dict_1k k1000" "dict_10k k32" "dict_10k k2048" "dict_10k k10000"; do set -- $pt
  timeout 600 python bench.py --set $1 --corpus $2 --steps 20 --warmup 5 2>/dev/null | grep '^{' | tail -1 > $OUT/bench_$1_$2.json
done
timeout 900 python bench.py --set dict_10k --corpus k10000 --len 16384 --steps 10 --warmup 3 --settle 20 2>/dev/null | grep '^{' | tail -1 > $OUT/bench_c5_dict_10k_16k.json
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/final_r05/bench_dict*.json")+glob.glob("gpurun_out/final_r05/bench_set_b_mix*.json")+glob.glob("gpurun_out/final_r05/bench_c5_dict*.json")):
    try:
        d=json.load(open(f)); w=d.get("working_set",{}); c=d.get("cpu_baseline",{})
        print(f.split("/")[-1], d["value"], d["roofline"]["kernel"].split("::")[-1][:40], "frac", d["roofline"]["frac"], "twice", d["traps"].get("wide_walk_wave_chunk_share_walked_twice"), "parity", c.get("parity_vs_gpu"), d.get("parity_of_repeats"))
    except Exception as e: print(f, "unreadable", e)
PY
timeout 900 python tools/wide_case.py --log2-strings 20 --out $OUT/wide_curve.jsonl > $OUT/wide_curve.log 2>&1; echo "wide_case rc=$?"
