#!/bin/bash
# kernel timeline of the segmented scan (tools/long_case.py) for a few string counts
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for n in ${PROF_NS:-8}; do
LONG_NS=$n LONG_TOTAL_LOG2=30 PYTHONPATH=. timeout 200 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/proflong$n -o p -- python tools/long_case.py > gpurun_out/proflong$n.log 2>&1
echo "== n=$n"; grep segmented gpurun_out/proflong$n.log | tail -2
python - <<PY
import csv,glob
f=glob.glob("gpurun_out/proflong$n/**/p_kernel_trace.csv",recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
idx=[i for i,r in enumerate(rows) if "SegmentFinish" in r["Kernel_Name"]]
# the last complete segmented call: from the previous finish to the last finish
lo=idx[-2]+1 if len(idx)>1 else 0
sel=rows[lo:idx[-1]+1]
t0=int(sel[0]["Start_Timestamp"])
for r in sel:
    print("  %9.1f us  +%8.1f us  %s"%((int(r["Start_Timestamp"])-t0)/1e3,(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3,r["Kernel_Name"][:70]))
PY
done
