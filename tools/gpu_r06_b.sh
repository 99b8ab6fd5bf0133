#!/bin/bash
# round 6: bench lines of the wide sets under the zipped image (held-out ranking, CPU baseline = the reference on all cores)
set -u
export PYTHONUNBUFFERED=1 TMPDIR=/tmp PYTHONPATH=.
OUT=gpurun_out/r06b
mkdir -p $OUT
for pt in "dict_1k k1000" "dict_10k k32" "dict_10k k2048" "dict_10k k10000" "dict_utf8_1k k1000" "dict_utf8_5k k5000" "dict_1k k128" "set_b_mix mix"; do set -- $pt
  timeout 600 python bench.py --set $1 --corpus $2 --steps 20 --warmup 5 ${EXTRA:-} 2>/dev/null | grep '^{' | tail -1 > $OUT/bench_$1_$2.json
  python - "$OUT/bench_$1_$2.json" <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); w=d["config"]["walk"]; c=d.get("cpu_baseline",{})
print(sys.argv[1].split("/")[-1], d["value"], "before", d["value_before_adapt"], d["roofline"]["kernel"][-40:], "frac", d["roofline"]["frac"], "tier", w["wide_rows"], "rows", w["states_with_a_row_of_their_own"], "outside", w["measured_share_outside_wide_rows"], "cpu", c.get("value"), c.get("parity_vs_gpu"))
PY
done
timeout 900 python bench.py --set dict_10k --corpus k10000 --len 16384 --log2-strings 20 --steps 10 --warmup 3 --settle 20 2>/dev/null | grep '^{' | tail -1 > $OUT/bench_c5_dict_10k_16k.json
python - <<'PY'
import json
d=json.load(open("gpurun_out/r06b/bench_c5_dict_10k_16k.json")); w=d["config"]["walk"]; c=d.get("cpu_baseline",{})
print("C5 shape", d["value"], d["roofline"]["kernel"][-40:], "frac", d["roofline"]["frac"], "tier", w["wide_rows"], "outside", w["measured_share_outside_wide_rows"], "cpu", c.get("value"), c.get("parity_vs_gpu"))
PY
