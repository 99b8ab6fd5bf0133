#!/bin/bash
# rocprofv3 kernel statistics of the secondary workloads at HEAD (the headline's are in the evidence run)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp PYTHONPATH=.
summ() {
python - "$1" "$2" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*kernel_stats.csv")
print("== " + sys.argv[2])
if f:
    for r in list(csv.DictReader(open(f[0])))[:8]:
        if "at::native" in r["Name"] or "rocclr" in r["Name"]:
            continue
        print("  %-96s calls %5s  avg %9.1f us  min %9.1f  max %9.1f" % (r["Name"][:96], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
}
{
for c in urls loglines; do rm -rf gpurun_out/ks; timeout 200 rocprofv3 --kernel-trace --stats -d gpurun_out/ks -o k --output-format csv -- python tools/ragged_case.py $c 3 > /dev/null 2>&1; summ gpurun_out/ks "tools/ragged_case.py $c (set_a, after adapt)"; done
rm -rf gpurun_out/ks; timeout 200 rocprofv3 --kernel-trace --stats -d gpurun_out/ks -o k --output-format csv -- python tools/counting_case.py > /dev/null 2>&1; summ gpurun_out/ks "tools/counting_case.py count_glued3_advanced (2^20 strings of 64..1023 B)"
rm -rf gpurun_out/ks; timeout 200 rocprofv3 --kernel-trace --stats -d gpurun_out/ks -o k --output-format csv -- python tools/capture_case.py > /dev/null 2>&1; summ gpurun_out/ks "tools/capture_case.py (three capture scanners)"
rm -rf gpurun_out/ks; LONG_TOTAL_LOG2=30 LONG_NS=1 timeout 200 rocprofv3 --kernel-trace --stats -d gpurun_out/ks -o k --output-format csv -- python tools/long_case.py > /dev/null 2>&1; summ gpurun_out/ks "tools/long_case.py, one 1 GiB string (learning call, product walk, pair kernel and two tiled passes for the A/B)"
rm -rf gpurun_out/ks; timeout 200 rocprofv3 --kernel-trace --stats -d gpurun_out/ks -o k --output-format csv -- python tools/long_grep_case.py > /dev/null 2>&1; summ gpurun_out/ks "tools/long_grep_case.py, one 1 GiB string under grep-like patterns (derived modes; product + passes for the A/B)"
rm -rf gpurun_out/ks
} 2>&1 | tee gpurun_out/kernel_stats_secondary.txt
