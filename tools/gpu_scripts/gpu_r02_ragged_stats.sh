#!/bin/bash
# rocprofv3 kernel statistics of the offset entry point at HEAD: ragged batches (small and x4), actions
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp PYTHONPATH=.
OUT=gpurun_out/rstats; rm -rf $OUT; mkdir -p $OUT
for c in urls urls_x4 loglines loglines_x4 uniform2k_x4; do
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$c -o s -- python tools/ragged_case.py $c 5 > $OUT/$c.log 2>&1
  echo "== tools/ragged_case.py $c 5"; grep "^ragged" $OUT/$c.log; grep "ScanRagged" $OUT/$c/s_kernel_stats.csv
done > $OUT/summary.txt 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/actions -o s -- python tools/actions_case.py > $OUT/actions.log 2>&1
{ echo "== tools/actions_case.py"; grep "ScanRagged\|HalfFinalKernel\|PrefixKernel" $OUT/actions/s_kernel_stats.csv; } >> $OUT/summary.txt
cat $OUT/summary.txt
find $OUT -name "*.csv" -size +1M -delete; find $OUT -name "*.db" -delete
