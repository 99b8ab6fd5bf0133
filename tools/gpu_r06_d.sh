#!/bin/bash
# round 6, fourth batch: the walks with actions on the class-indexed walk -- parity, then prefix / half-final on the wide sets
export PYTHONUNBUFFERED=1 TMPDIR=/tmp PYTHONPATH=.
mkdir -p gpurun_out/r06d
timeout 1500 python -m pytest tests/test_wide_actions.py tests/test_selftest.py tests/test_prefix.py tests/test_zip.py -q -m gpu -x 2>&1 | tail -12
echo "== prefix / half-final, dense rows against wide rows"
timeout 900 python tools/actions_wide_case.py 2>&1 | grep "^{" | tee gpurun_out/r06d/actions_wide.jsonl | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l)
    print(d['set'],d['corpus'],'with a prefix',d['share_of_strings_with_a_prefix'])
    for leg in ('dense','wide'):
        if leg in d:
            print('  ',leg,{k:(v['kernel'],v['GBps'],v['parity_all_strings']) for k,v in d[leg].items() if isinstance(v,dict)}, d[leg]['tier_states'], d[leg]['zipped'])
"
