#!/usr/bin/env python3
"""The ISA audits of the kernels that keep text on its way in registers, as ONE step of the build (round 5).

The tiled / wide / ragged / stream / pair kernels issue their text loads from inline asm and wait for them with
hand-counted s_waitcnt; CountingRowKernel / CaptureRowKernel land theirs in a0..a31.  Between a load and its wait the
registers belong to the memory system, and hipcc does not know: a spill, a live-range split or a reused register there
reads or clobbers data that has not arrived -- silently, and only when the data is late (three
such incidents in round 4 alone, one of them only under -O1).  Round 4 checked this in pytest; a library built by
`make` alone could fail the checks and ship.  Now `make` runs

    python tools/audit/build_audit.py <unit.hip> --stamp build/audit_<unit>.json

for every unit below before it links libpire_hip.so (one make target per unit, so `make -j` runs them side by side), the
link fails when an audit fails, and the result -- hipcc's version, the kernels looked at -- is compiled into the library
(pire_hip_build_info()).  tests/test_build_audit.py calls the same functions.

What is checked, per unit (the product's flags: -O3 --offload-arch=gfx950):
  tiled / wide / ragged / stream / pair   no scratch, no VGPR spills, <= 128 VGPRs (16 waves per CU); in the window loop no
                                          instruction names a tile register between its load and the wait that covers it,
                                          and on every way OUT of that loop none does before the loop's last loads are
                                          waited for (inflight_registers.py check / check_exits; round 6: both bugs of this
                                          round were loads still on their way when a loop was left)
  counting                                the row kernels: a0..a31 named by the loads and v_accvgpr_read only, loads of the
                                          form `global_load_dwordx4 a[..], v[..], off`, no scratch inside the window loop,
                                          VGPRs + AGPRs <= 128; every other kernel of the unit: no scratch
  exact / slow / segmented / order        no scratch (per-lane counter arrays stay in registers)
"""
import argparse
import importlib.util
import json
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "pire_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# The flags the audited ISA is compiled with ARE the build's: pire_amd/csrc/Makefile hands over its $(CFLAGS) (the
# audit compiled with flags of its own, so an overridden ARCH or added flags were never looked at); without the Makefile -- the
# tests, a call by hand -- the product's defaults.  The stamp records them, pire_hip_build_info() prints them.
FLAGS = (os.environ["PIRE_AUDIT_CFLAGS"].split() if os.environ.get("PIRE_AUDIT_CFLAGS") else ["--offload-arch=gfx950", "-O3", "-std=c++17"]) + ["-x", "hip"]

# unit -> (kernels whose loads are inline asm: substring of the mangled name, checked with the window-loop walker)
WINDOW = {"tiled.hip": ["ScanTiledKernel", "ScanTiledSegKernel"], "wide.hip": ["ScanWideKernel", "ScanWide2Kernel"], "ragged.hip": ["ScanRaggedKernel"],
          "stream.hip": ["ScanStreamKernel"], "pair.hip": ["ScanPairTiledKernel"]}
# the window-loop walker's reports are demanded empty for these (it was written for them; the pair / segment kernels'
# loops have shapes it does not follow, their pins are no scratch + no spills)
WALKED = {"tiled.hip": ["ScanTiledKernel"], "wide.hip": ["ScanWideKernel", "ScanWide2Kernel"], "ragged.hip": ["ScanRaggedKernel"],
          "stream.hip": ["ScanStreamKernel"]}
# ... whose INNER tile loop's exits are followed as well (the kernel that chains tasks through its ring of two tiles and leaves
# the tile loop early when a wave's strings are all absorbed: where round 5's wrong-result bug was)
INNER_EXITS = {"wide.hip": ["ScanWideKernel"]}
NO_SCRATCH = ["exact.hip", "slow.hip", "segmented.hip", "order.hip", "counting.hip"]
UNITS = sorted(set(WINDOW) | set(NO_SCRATCH))


def hipcc_version() -> str:
    out = subprocess.run([HIPCC, "--version"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True).stdout
    hip = re.search(r"HIP version: (\S+)", out)
    clang = re.search(r"clang version (\S+)", out)
    return "hipcc HIP %s clang %s" % (hip.group(1) if hip else "?", clang.group(1) if clang else "?")


def resources(unit, extra=()):
    """{kernel: {remark: value}} from -Rpass-analysis=kernel-resource-usage."""
    src = os.path.join(CSRC, unit)
    r = subprocess.run([HIPCC] + FLAGS + list(extra) + ["-c", src, "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=1800)
    if r.returncode != 0:
        raise RuntimeError(r.stdout[-3000:])
    kernels, cur = {}, None
    for line in r.stdout.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = kernels.setdefault(m.group(1), {})
            continue
        m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[bytes/lane\])?: (\d+)", line)
        if m and cur is not None:
            cur[m.group(1).strip()] = int(m.group(2))
    return kernels


def isa(unit, extra=()):
    src = os.path.join(CSRC, unit)
    r = subprocess.run([HIPCC] + FLAGS + list(extra) + ["--offload-device-only", "-S", src, "-o", "-"], stdout=subprocess.PIPE,
                       stderr=subprocess.DEVNULL, text=True, timeout=1800)
    if r.returncode != 0:
        raise RuntimeError("hipcc -S failed for " + unit)
    return r.stdout


def _inflight():
    spec = importlib.util.spec_from_file_location("inflight_registers", os.path.join(HERE, "inflight_registers.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def audit_window_unit(unit, extra=()):
    """Failures (strings) of a unit whose kernels keep a tile on its way in ordinary registers; and the kernels seen."""
    fails, seen = [], []
    res = resources(unit, extra)
    for want in WINDOW[unit]:
        mine = {k: v for k, v in res.items() if want in k}
        if not mine:
            fails.append("%s: no %s instantiation found" % (unit, want))
        for name, r in mine.items():
            seen.append(name)
            if r.get("ScratchSize", -1) != 0:
                fails.append("%s: scratch %s bytes per lane (a spilled tile register is read or clobbered while its load is in flight)" % (name, r.get("ScratchSize")))
            if r.get("VGPRs Spill", -1) != 0:
                fails.append("%s: %s VGPR spills" % (name, r.get("VGPRs Spill")))
            if r.get("VGPRs", 999) > 128:
                fails.append("%s: %s VGPRs (16 waves per CU need <= 128)" % (name, r.get("VGPRs")))
    if unit in WALKED:
        mod = _inflight()
        found = 0
        for name, body in mod.kernels(isa(unit, extra)):
            if any(w in name for w in WALKED[unit]):
                found += 1
                rep = mod.check(body)
                if rep is None:
                    fails.append("%s: no window loop found" % name)
                elif rep:
                    fails.append("%s: %d instructions name a tile register between its load and its wait, first: %s" % (name, len(rep), rep[0][1]))
                ex = mod.check_exits(body) or []   # (round 6: the ways OUT of the window loop, DESIGN.md 6 lessons 24 and 29)
                if any(w in name for w in INNER_EXITS.get(unit, [])):
                    ex = sorted(set(ex) | set(mod.check_exits(body, outermost=False) or []))
                if ex:
                    fails.append("%s: behind the window loop %d instructions name a tile register before the loop's last loads are waited for, first: %s"
                                 % (name, len(ex), ex[0][1]))
        if found < len(WALKED[unit]):
            fails.append("%s: %d kernel bodies in the ISA, at least %d expected (%s)" % (unit, found, len(WALKED[unit]), ", ".join(WALKED[unit])))
    return fails, seen


def audit_counting(extra=()):
    """The row kernels of counting.hip own a0..a31; every other kernel of the unit: no scratch."""
    fails, seen = [], []
    res = resources("counting.hip", extra)
    rows = {k: v for k, v in res.items() if "CountingRowKernel" in k or "CaptureRowKernel" in k}
    if len(rows) != 16:
        fails.append("counting.hip: %d row kernel instantiations, 16 expected" % len(rows))
    for name, r in res.items():
        if "pirehip" not in name:
            continue
        seen.append(name)
        if name in rows:
            if r["VGPRs"] + r.get("AGPRs", 0) > 128:
                fails.append("%s: %d + %d registers" % (name, r["VGPRs"], r.get("AGPRs", 0)))
            if r.get("ScratchSize", -1) > (96 if "Capture" in name else 64):
                fails.append("%s: %s bytes of scratch (a few per-pass values are expected, no array)" % (name, r.get("ScratchSize")))
        elif r.get("ScratchSize", -1) != 0:
            fails.append("%s: scratch" % name)
    lines = isa("counting.hip", extra).splitlines()
    body, count = None, 0
    for n, line in enumerate(lines):
        m = re.match(r"^(_ZN7pirehip\w*(?:CountingRowKernel|CaptureRowKernel)\S*):", line)
        if m:
            body, count = m.group(1), count + 1
            end = next(k for k in range(n, len(lines)) if lines[k].startswith(".Lfunc_end"))
            land = [k for k in range(n, end) if re.search(r"v_accvgpr_read_b32 v\d+, a0\b", lines[k])]
            wide = [k for k in range(n, end) if "ds_read_b64" in lines[k] or "ds_read_b128" in lines[k]]
            if not land or not wide:
                fails.append("%s: landing registers / row reads not found" % body)
            elif "Capture" not in body and [k for k in range(land[0], max(wide)) if "scratch_" in lines[k]]:
                fails.append("%s: scratch access inside the window loop" % body)
        elif line.startswith(".Lfunc_end"):
            body = None
        elif body:
            for m in re.finditer(r"\ba(\d+)\b|\ba\[(\d+):(\d+)\]", line.split(";")[0]):
                lo = int(m.group(1) if m.group(1) is not None else m.group(2))
                if lo >= 32:
                    continue
                op = line.split()[0]
                if op not in ("global_load_dwordx4", "v_accvgpr_read_b32"):
                    fails.append("%s: `%s` names a landing register" % (body, line.strip()))
                elif op == "global_load_dwordx4" and not re.search(r"global_load_dwordx4 a\[\d+:\d+\], v\[\d+:\d+\], off", line):
                    fails.append("%s: unexpected form of a landing load: %s" % (body, line.strip()))
    if count != 16:
        fails.append("counting.hip: %d row kernel bodies in the ISA, 16 expected" % count)
    return fails, seen


def audit_no_scratch(unit, extra=()):
    fails, seen = [], []
    for name, r in resources(unit, extra).items():
        if "pirehip" in name:     # segmented.hip also instantiates library (rocprim) scan kernels
            seen.append(name)
            if r.get("ScratchSize", -1) != 0:
                fails.append("%s: %s bytes of scratch per lane" % (name, r.get("ScratchSize")))
    return fails, seen


def audit(unit, extra=()):
    if unit in WINDOW:
        return audit_window_unit(unit, extra)
    if unit == "counting.hip":
        return audit_counting(extra)
    return audit_no_scratch(unit, extra)


def emit_header(path, stamps):
    """build_info.h: what pire_hip_build_info() returns -- the compiler the audits passed with and what they looked at."""
    units, hipcc, flags = [], set(), set()
    for st in sorted(stamps):
        with open(st) as f:
            d = json.load(f)
        hipcc.add(d["hipcc"])
        flags.add(d["flags"])
        for u, r in d["units"].items():
            assert not r["failures"], (st, r)
            units.append("%s (%d)" % (u, r["kernels"]))
    text = "libpire_hip: ISA audit passed (tools/audit/build_audit.py) with %s, device flags `%s`; units (kernels): %s" % (
        " / ".join(sorted(hipcc)), " / ".join(sorted(flags)), ", ".join(units))
    with open(path, "w") as f:
        f.write("// generated by pire_amd/csrc/Makefile from the audits' stamps\n#define PIRE_HIP_BUILD_INFO \"%s\"\n" % text.replace('"', "'"))
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("unit", nargs="*", help="units to audit (default: all of %s)" % ", ".join(UNITS))
    ap.add_argument("--stamp", default="", help="write the result (JSON) here when the audit passes")
    ap.add_argument("--device-flags", default="", help="extra hipcc flags, space separated (tests: an -O1 device build must FAIL)")
    ap.add_argument("--emit-header", default="", help="write build_info.h from the stamps given as `unit` arguments, audit nothing")
    args = ap.parse_args()
    if args.emit_header:
        return emit_header(args.emit_header, args.unit)
    extra = args.device_flags.split()
    bad = 0
    result = {"hipcc": hipcc_version(), "flags": " ".join(FLAGS + extra), "units": {}}
    for unit in args.unit or UNITS:
        unit = os.path.basename(unit)
        fails, seen = audit(unit, extra)
        result["units"][unit] = {"kernels": len(seen), "failures": fails}
        print("audit %-14s %3d kernels  %s" % (unit, len(seen), "ok" if not fails else "FAILED"))
        for f in fails:
            print("    " + f)
        bad += len(fails)
    if args.stamp and not bad:
        with open(args.stamp, "w") as f:
            json.dump(result, f)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
