#!/bin/bash
# Offset batches below the stream kernel's thresholds: KERNEL durations (rocprofv3 trace, not host-side events: a Python
# launch loop cannot keep a 20 us kernel busy) of the ragged kernel (variant 1) and the stream kernel (variant 2).
export PYTHONPATH=. PYTHONUNBUFFERED=1 TMPDIR=/tmp
OUT=gpurun_out/small_batches
mkdir -p $OUT
for c in urls_16k urls_64k urls_256k urls_1m loglines_16k loglines_64k loglines_256k; do
  for v in 1 2; do
    PIRE_HIP_RAGGED_VARIANT=$v timeout 120 rocprofv3 --kernel-trace --output-format csv -d $OUT/${c}_v$v -o t -- python tools/ragged_case.py $c 2 > $OUT/${c}_v$v.log 2>&1
    echo -n "$c variant=$v: "; grep "strings," $OUT/${c}_v$v.log | cut -c1-60 | tr '\n' ' '; python tools/summarize_trace.py $OUT/${c}_v$v 20 $([ $v = 1 ] && echo ScanRaggedKernel || echo ScanStreamKernel) | sed -n 3p
  done
done | tee $OUT/summary.txt
find $OUT -name "*.csv" -size +1M -delete
