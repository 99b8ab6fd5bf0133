#!/usr/bin/env python3
"""profiles/rNN_bench_pmc_summary.txt (tools/summarize_pmc.py over the --pmc passes of bench.py) -> the JSON bench.py
reads its `roofline.traffic` and `lds_gather` from.  usage: make_pmc_json.py <pmc_summary.txt> <out.json> <workload> <command>"""
import hashlib
import json
import os
import re
import sys

src, out, workload, command = sys.argv[1:5]
vals, kernel, cur = {}, None, None
for line in open(src):
    m = re.match(r"kernel: (.*)", line)
    if m:
        cur = m.group(1).strip()
        continue
    m = re.match(r"\s+(\S+)\s+dispatches\s+\d+\s+avg/dispatch\s+(\S+)", line)
    if m and cur and "ScanTiled" in cur:
        kernel = cur
        vals[m.group(1)] = float(m.group(2))
fetch, write = vals["FETCH_SIZE"], vals["WRITE_SIZE"]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
h = hashlib.sha256()
for name in ("tiled.hip", "device_common.h", "internal.h"):   # the same three files bench.py kernel_sources_sha16() hashes
    h.update(open(os.path.join(root, "pire_amd", "csrc", name), "rb").read())
res = {
    "workload": workload, "kernel": kernel, "kernel_sources_sha16": h.hexdigest()[:16],
    "FETCH_SIZE_KB_per_launch": fetch, "WRITE_SIZE_KB_per_launch": write,
    "correction": "FETCH_SIZE x2 for 16 B/lane whole-line reads on gfx950 (MI355X_MICROARCH.md HBM section)"
                  + ("; cross-check: TCC_MISS_sum %.6g x 128 B = %.4g B" % (vals["TCC_MISS_sum"], vals["TCC_MISS_sum"] * 128)
                     if "TCC_MISS_sum" in vals else ""),
    "hbm_bytes_per_launch": round((2 * fetch + write) * 1024, 1),
    "source": f"{src} (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes, {command})",
}
if "SQ_LDS_IDX_ACTIVE" in vals:
    res.update({"lds_bank_conflict_cycles_per_launch": vals["SQ_LDS_BANK_CONFLICT"],
                "lds_idx_active_cycles_per_launch": vals["SQ_LDS_IDX_ACTIVE"],
                "lds_instructions_per_launch": vals["SQ_INSTS_LDS"]})
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res))
