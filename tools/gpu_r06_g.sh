#!/bin/bash
export PYTHONUNBUFFERED=1 TMPDIR=/tmp PYTHONPATH=.
mkdir -p gpurun_out/r06g
for s in blacklist_1k blacklist_10k; do timeout 400 python tools/ranking_quality.py $s 2>&1 | grep "^after\|ideal" | cut -c1-200; done | tee gpurun_out/r06g/ranking_quality.txt
timeout 900 python -m pytest tests/test_wide.py tests/test_wide_actions.py tests/test_zip.py -q -m gpu -x 2>&1 | tail -4 | cut -c1-300
WIDE_CASE_LEGS=wide,zip,auto timeout 900 python tools/wide_case.py --points blacklist_1k:urls,blacklist_10k:urls --log2-urls 23 2>&1 | grep "^{" | tee gpurun_out/r06g/urls.jsonl | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print(d['set'],d['corpus'],{k:(d[k]['kernel'],d[k]['GBps'],d[k]['parity_all_strings'],d[k].get('tier_states'),d[k]['measured_share_outside_wide_rows']) for k in ('wide','zip','auto') if k in d})
"
