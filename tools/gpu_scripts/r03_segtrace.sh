#!/bin/bash
# per-kernel durations of the segmented scan of one 1 GiB string (kernel trace)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
PYTHONPATH=. LONG_TOTAL_LOG2=30 LONG_NS=1 timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/segtrace -o seg --output-format csv -- python tools/long_case.py set_a > gpurun_out/segtrace.log 2>&1
tail -3 gpurun_out/segtrace.log
f=$(find gpurun_out/segtrace -name '*kernel_trace.csv' | head -1)
python - "$f" <<'PY' | tee gpurun_out/segmented_call_timeline.log
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last segmented call: print the last 40 kernels with start offsets
tail = [r for r in rows if "copyBuffer" not in r["Kernel_Name"]][-14:]
t0 = int(tail[0]["Start_Timestamp"])
for r in tail:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%9.1f us  +%7.1f us  %s" % ((s - t0) / 1e3, (e - s) / 1e3, r["Kernel_Name"][:90]))
PY
find gpurun_out/segtrace -name '*.csv' -size +2M -delete
