#!/bin/bash
set -u
export PYTHONUNBUFFERED=1
export TMPDIR=/tmp
mkdir -p gpurun_out/abl
for mode in full noload nostep; do
  case $mode in full) E="";; noload) E="PIRE_HIP_DEBUG_NOLOAD=1";; nostep) E="PIRE_HIP_DEBUG_NOSTEP=1";; esac
  env $E timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES --kernel-trace --output-format csv -d gpurun_out/abl/$mode -o p -- python bench.py --steps 5 --warmup 1 --no-cpu > gpurun_out/abl/$mode.log 2>&1
  echo "== $mode"; python tools/summarize_pmc.py gpurun_out/abl/$mode | grep -A8 "ScanTiled"
  python - <<PY
import csv,glob
f=glob.glob("gpurun_out/abl/$mode/**/*kernel_trace.csv", recursive=True)[0]
d=[ (int(r["End_Timestamp"])-int(r["Start_Timestamp"])) for r in csv.DictReader(open(f)) if "ScanTiled" in r["Kernel_Name"]]
print("   kernel durations us:", [round(x/1000,1) for x in d])
PY
done
