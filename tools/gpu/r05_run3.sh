#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export PYTHONPATH=. PYTHONUNBUFFERED=1 TMPDIR=/tmp
TAG=${1:-r05c}
timeout 900 python -m pytest tests/test_wide.py -m gpu -x -q 2>&1 | tail -3
timeout 900 python tools/wide_case.py --points "${2:-dict_1k:k128,dict_1k:k512,dict_1k:k1000,dict_10k:k32,dict_10k:k2048,dict_10k:k10000}" --log2-strings ${3:-20} --out gpurun_out/${TAG}_wide_curve.jsonl > gpurun_out/${TAG}_wide_curve.log 2>&1; echo "wide_case rc=$?"
python - <<PY
import json
for l in open("gpurun_out/${TAG}_wide_curve.jsonl"):
    d=json.loads(l)
    print(d["set"], d["corpus"], d.get("GiB", ""), "visited", d["distinct_states_visited_in_sample"], "rows", d["wide_rows"], "| " + " | ".join("%s %s %.0f GB/s twice %.4f out_dense %.4f out_wide %.4f par %s" % (k, d[k]["kernel"], d[k]["GBps"], d[k].get("wave_chunk_share_walked_twice_by_the_wide_walk", -1), d[k]["measured_share_outside_dense_rows"], d[k]["measured_share_outside_wide_rows"], d[k]["parity_all_strings"]) for k in ("dense","wide","wide2","auto") if k in d))
PY
