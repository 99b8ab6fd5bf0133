#!/bin/bash
# A/B of an experiment build (make exp N=k) against the product on wide_case points: tools/gpu_exp.sh <k> <points> <legs>
export PYTHONUNBUFFERED=1 TMPDIR=/tmp PYTHONPATH=.
for lib in "" tools/ab/libpire_hip_exp$1.so; do
  echo "== ${lib:-product}"
  PIRE_HIP_LIB=$lib WIDE_CASE_LEGS=$3 timeout 900 python tools/wide_case.py --log2-strings 20 --points $2 2>&1 | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['set'],d['corpus'],{k:(d[k]['GBps'],d[k]['parity_all_strings'],d[k].get('tier_states'),d[k].get('states_with_a_row'),d[k]['measured_share_outside_wide_rows']) for k in ('wide2','zip','zip2','auto') if k in d})
"
done
