#!/bin/bash
# timeline of the small host-pointer calls (run / prefix / half-final, 10 strings): kernels and copies
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
rm -rf gpurun_out/smalltrace
PYTHONPATH=. timeout 600 rocprofv3 --kernel-trace --memory-copy-trace -d gpurun_out/smalltrace -o small --output-format csv -- python tools/host_call_latency.py > gpurun_out/smalltrace.log 2>&1
python - <<'PY' | tee gpurun_out/small_call_timeline.log
import csv, glob
ev = []
for r in csv.DictReader(open(glob.glob("gpurun_out/smalltrace/*kernel_trace.csv")[0])):
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60]))
for r in csv.DictReader(open(glob.glob("gpurun_out/smalltrace/*memory_copy_trace.csv")[0])):
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy " + r.get("Direction", "") ))
ev.sort()
# find the first PrefixKernel and HalfFinalKernel occurrences at host_staging=0 (first block of each), print 3 calls around them
def show(name, skip):
    idx = [i for i, e in enumerate(ev) if name in e[2]]
    i0 = idx[skip]
    t0 = ev[i0 - 4][0]
    for s, e, n in ev[i0 - 4:i0 + 12]:
        print("%9.1f us +%6.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, n))
    print()
show("PrefixKernel", 20)
show("HalfFinalKernel", 20)
show("ScanGenericKernel", 10)
PY
