#!/bin/bash
# round 6, third batch: the stream kernel on the class-indexed walk -- parity, then the blacklist scanners' URL batches
export PYTHONUNBUFFERED=1 TMPDIR=/tmp PYTHONPATH=.
mkdir -p gpurun_out/r06c
timeout 1200 python -m pytest tests/test_wide.py tests/test_selftest.py -q -m gpu -x -k "stream or first_use" 2>&1 | tail -8
echo "== stage clocks (tuning build)"
PIRE_HIP_LIB=tools/ab/libpire_hip_tuning.so PIRE_HIP_DEBUG_STREAM_CLOCKS=1 WIDE_CASE_LEGS=wide,zip timeout 600 python tools/wide_case.py --points blacklist_1k:urls --log2-urls 23 2>&1 | grep "stream clocks" | cut -c1-400
echo "== URL batches"
timeout 900 python tools/wide_case.py --points blacklist_1k:urls,blacklist_10k:urls --log2-urls 23 2>&1 | tee gpurun_out/r06c/urls.jsonl | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['set'],d['corpus'],{k:(d[k]['kernel'],d[k]['GBps'],d[k]['parity_all_strings'],d[k].get('tier_states'),d[k].get('states_with_a_row'),d[k]['measured_share_outside_wide_rows']) for k in ('dense','wide_ragged','wide','zip_ragged','zip','auto') if k in d})
"
