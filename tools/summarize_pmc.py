#!/usr/bin/env python3
"""Summarise rocprofv3 counter_collection CSVs: per-kernel average of each counter for the scan kernel."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/prof"
# --last K: only the K most recent dispatches of each kernel (the timed region of a bench.py run: the launches before it
# belong to the held-out corpus and to a never-adapted table, whose counters are another kernel's as far as traps go)
last = int(sys.argv[sys.argv.index("--last") + 1]) if "--last" in sys.argv else 0
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            k = row.get("Kernel_Name", "")
            if not any(w in k for w in ("Scan", "Corpus", "Slow", "Counting", "Capture")):
                continue
            acc[k.split("(")[0][:60]][row["Counter_Name"]].append((row.get("Dispatch_Id"), float(row["Counter_Value"])))
for k, ctrs in acc.items():
    print("kernel:", k)
    for c, vals in sorted(ctrs.items()):
        per = defaultdict(float)
        for d, v in vals:
            per[d] += v           # sum over XCDs / SEs of one dispatch
        xs = [per[d] for d in sorted(per, key=lambda d: int(d))]
        if last:
            xs = xs[-last:]
        print("  %-32s dispatches %3d  avg/dispatch %.6g" % (c, len(xs), sum(xs) / len(xs)))
