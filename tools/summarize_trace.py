#!/usr/bin/env python3
"""rocprofv3 --kernel-trace CSV -> duration statistics of the LAST K dispatches of a kernel: bench.py's timed region is
the last K launches of the process, so this is the rocprof view of exactly the launches the bench line's
`roofline.kernel_avg_ms` averages (the whole-run --stats average mixes warm-up and cold-table launches in).
usage: summarize_trace.py <dir with *_kernel_trace.csv> <K> [kernel substring]"""
import csv
import glob
import os
import sys

root, k = sys.argv[1], int(sys.argv[2])
needle = sys.argv[3] if len(sys.argv) > 3 else "ScanTiledKernel"
rows = []
for f in glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            if needle in r.get("Kernel_Name", ""):
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0]))
rows.sort()
if not rows:
    sys.exit("no dispatch of %r under %s" % (needle, root))
d = [(e - s) / 1e6 for s, e, _ in rows]
last = d[-k:]
gaps = [(rows[i + 1][0] - rows[i][1]) / 1e3 for i in range(len(rows) - k, len(rows) - 1)]
print("kernel: %s" % rows[-1][2])
print("all %d dispatches of the process: avg %.4f ms, min %.4f, max %.4f" % (len(d), sum(d) / len(d), min(d), max(d)))
print("the last %d (the timed region): avg %.4f ms, min %.4f, max %.4f; gaps between them avg %.1f us, max %.1f us" % (
    len(last), sum(last) / len(last), min(last), max(last), sum(gaps) / max(len(gaps), 1), max(gaps) if gaps else 0))
print("per launch (ms):", " ".join("%.3f" % x for x in last))
