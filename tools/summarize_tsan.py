#!/usr/bin/env python3
"""ThreadSanitizer reports of a run with the instrumented libpire_hip_tsan.so inside an otherwise uninstrumented process
(Python, torch, the HIP / HSA runtimes): most reports pair an access inside libhsa / libamdhip64 with a malloc of ours --
memory the runtime recycles behind TSan's back.  A report counts against the library only if one of its two racing
ACCESSES (the first frame that is not a sanitizer interceptor) is code of libpire_hip_tsan.so.
usage: summarize_tsan.py <report files...>"""
import re
import sys

text = "".join(open(f, errors="replace").read() for f in sys.argv[1:])
reports = [r for r in text.split("==================") if "WARNING: ThreadSanitizer" in r]
ours, kinds = [], {}
for r in reports:
    kind = re.search(r"WARNING: ThreadSanitizer: ([^(\n]*)", r).group(1).strip()
    kinds[kind] = kinds.get(kind, 0) + 1
    # the access blocks: "  <Read|Write|Atomic ...> of size N at ... by ...:" / "  Previous ... by ...:" followed by frames
    blocks = re.findall(r"\n  (?:Previous )?(?:[Aa]tomic )?(?:[Rr]ead|[Ww]rite) of size[^\n]*\n((?:    #\d+[^\n]*\n)+)", r)
    hit = False
    for b in blocks:
        frames = [l for l in b.splitlines() if "compiler-rt" not in l and "tsan_interceptors" not in l]
        if frames and "libpire_hip_tsan.so" in frames[0]:
            hit = True
    if hit:
        ours.append(r)
print("reports: %d %s; with a racing access inside libpire_hip_tsan.so: %d" % (len(reports), kinds, len(ours)))
for r in ours[:10]:
    print("==================" + r[:3000])
