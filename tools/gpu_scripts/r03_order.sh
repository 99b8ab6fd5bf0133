#!/bin/bash
# the length order (order.hip): what the three launches cost and what the counting kernel gains
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp PYTHONPATH=.
{
echo "== fixed length 544, order on / off"
COUNTING_FIXED_LEN=544 python tools/counting_case.py 2>&1 | grep "^counting" | cut -c1-40,150-260
COUNTING_FIXED_LEN=544 NO_LENGTH_ORDER=1 python tools/counting_case.py 2>&1 | grep "^counting" | cut -c1-40,150-260
echo "== 64..1023 bytes, order on: kernel statistics"
rocprofv3 --kernel-trace --stats -d gpurun_out/ordertrace -o o --output-format csv -- python tools/counting_case.py > /dev/null 2>&1
python - <<'PY'
import csv, glob
for r in csv.DictReader(open(glob.glob("gpurun_out/ordertrace/*kernel_stats.csv")[0])):
    print("%-90s calls %5s avg %9.1f us" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
rm -rf gpurun_out/ordertrace
} 2>&1 | tee gpurun_out/length_order.log
