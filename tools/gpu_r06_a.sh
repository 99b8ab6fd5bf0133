#!/bin/bash
# round 6, first GPU pass: the zipped image -- parity tests, then the curve on the points VERDICT r5 names
set -u
export PYTHONUNBUFFERED=1 TMPDIR=/tmp PYTHONPATH=.
OUT=gpurun_out/r06a
mkdir -p $OUT
true
WIDE_CASE_LEGS=wide2,zip,zip2,auto timeout 1500 python tools/wide_case.py --log2-strings 20 --points dict_utf8_1k:k32,dict_utf8_1k:k1000,dict_utf8_5k:k512,dict_utf8_5k:k5000 --out $OUT/wide_curve.jsonl 2>&1 | tail -3 > $OUT/wide_curve.log
python - <<'PY'
import json
for l in open("gpurun_out/r06a/wide_curve.jsonl"):
    d=json.loads(l)
    print(d["set"],d["corpus"],{k:(d[k]["GBps"],d[k]["parity_all_strings"],d[k].get("tier_states"),d[k].get("states_with_a_row"),d[k]["measured_share_outside_wide_rows"],d[k]["symbol"][-28:]) for k in ("wide2","zip","zip2","auto") if k in d})
PY
