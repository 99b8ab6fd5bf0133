"""Static audit of the compiled gfx950 kernels (no GPU needed; hipcc cross-compiles).

The tiled kernel issues its HBM loads from inline asm and waits for them with hand-counted s_waitcnt, so the
compiler does not know those registers are in flight.  If it ever spills (or otherwise copies) a tile register
between the load and the wait, the scan reads garbage silently.  This was observed with a three-slot ring
(DESIGN.md section 6).  The ragged kernel prefetches the same way.  Pin: every instantiated scan kernel has NO
scratch and NO spills."""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc"


def _audit():
    import importlib.util

    spec = importlib.util.spec_from_file_location("build_audit", os.path.join(ROOT, "tools", "audit", "build_audit.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def resources(unit):
    return _audit().resources(unit)


def audit_of(unit):
    """(failures, kernels seen) of a unit: from the stamp `make` left when it audited exactly these sources (the stamp is
    a make target that depends on the unit, the headers, the Makefile and the audit scripts), else by running the audit."""
    import glob
    import json

    csrc = os.path.join(ROOT, "pire_amd", "csrc")
    stamp = os.path.join(csrc, "build", "audit_%s.json" % unit)
    deps = [os.path.join(csrc, unit), os.path.join(csrc, "Makefile"), os.path.join(ROOT, "include", "pire_hip.h")] + \
        glob.glob(os.path.join(csrc, "*.h")) + glob.glob(os.path.join(ROOT, "tools", "audit", "*.py"))
    if os.path.exists(stamp) and all(os.path.getmtime(stamp) >= os.path.getmtime(d) for d in deps):
        with open(stamp) as f:
            r = json.load(f)["units"][unit]
        return r["failures"], ["from the build's stamp"] * r["kernels"]
    return _audit().audit(unit)


# The audits themselves live in tools/audit/build_audit.py and are a step of `make` since round 5 (a library whose kernels
# fail them is not linked); the tests call the same functions, unit by unit.
@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
@pytest.mark.parametrize("unit", ["tiled.hip", "wide.hip", "ragged.hip", "stream.hip", "pair.hip"])
def test_scan_kernels_have_no_scratch_and_no_spills(unit):
    """No scratch, no spills, <= 128 VGPRs, and in the window loop nothing names a tile between its load and its wait."""
    fails, seen = audit_of(unit)
    assert seen and not fails, fails


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
@pytest.mark.parametrize("unit", ["exact.hip", "slow.hip", "segmented.hip", "order.hip"])
def test_exact_kernels_have_no_scratch(unit):
    """Per-lane counter arrays must stay in registers (a runtime index once put HalfFinalKernel's into scratch)."""
    fails, seen = audit_of(unit)
    assert seen and not fails, fails


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_counting_row_kernel_owns_its_landing_registers():
    """CountingRowKernel and CaptureRowKernel land the text line that is on its way in a0..a31, named in its asm statements only.  The
    compiler may use accumulation registers as spill space: never those (a register written by it while the memory
    system still owes data to it, or the other way round, is a silently wrong count), and no scratch."""
    fails, seen = audit_of("counting.hip")
    assert len(seen) >= 16 and not fails, fails


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_the_build_refuses_device_code_compiled_at_O1():
    """Round 4's failure: at -O1 hipcc spills and re-uses the registers CaptureRowKernel lands its text in, and the kernel
    came out wrong.  The audit `make` runs in front of the link must say no to that build."""
    if os.path.exists("/tmp/pire_audit_o1.json"):
        os.remove("/tmp/pire_audit_o1.json")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "audit", "build_audit.py"), "counting.hip", "--device-flags",
                        "-Xarch_device -O1", "--stamp", "/tmp/pire_audit_o1.json"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode != 0 and ("landing register" in r.stdout or "scratch" in r.stdout), r.stdout[-2000:]
    assert not os.path.exists("/tmp/pire_audit_o1.json")   # no stamp, so make has nothing to link build_info.o against


def test_the_library_says_which_compiler_its_audits_passed_with():
    """pire_hip_build_info(): the product library carries the summary of the audits it was linked behind."""
    from pire_amd import binding as pb

    info = pb.build_info()
    assert "ISA audit passed" in info and "wide.hip" in info and "counting.hip" in info, info
    mk = open(os.path.join(ROOT, "pire_amd", "csrc", "Makefile")).read()
    assert "$(OBJDIR)/build_info.cpp.o: $(HERE)build_info.cpp $(OBJDIR)/build_info.h" in mk and "$(OBJDIR)/build_info.h: $(AUDITS)" in mk
    if os.path.exists(HIPCC):
        assert _audit().hipcc_version() in info, (info, _audit().hipcc_version())


def _inflight():
    import importlib.util

    spec = importlib.util.spec_from_file_location("inflight_registers", os.path.join(ROOT, "tools", "audit", "inflight_registers.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_inflight_register_check_sees_a_copied_tile():
    """The checker itself: a tile register copied between its load and the wait that covers it is reported (that is
    what register allocation did to the first form of CountingRowKernel), the same loop without the copy is not."""
    loop = [".LBB0_1:"] + ["\tglobal_load_dwordx4 v[%d:%d], v[40:41], off" % (4 * j, 4 * j + 3) for j in range(8)]
    tail = ["\ts_waitcnt vmcnt(0)", "\tv_add_u32_e32 v50, v0, v5", "\ts_cbranch_vccnz .LBB0_1"]
    mod = _inflight()
    assert mod.check(loop + tail) == []
    rep = mod.check(loop + ["\tv_mov_b32_e32 v60, v28"] + tail)
    assert rep and "v_mov_b32_e32 v60, v28" in rep[0][1]
    # eight younger loads may stay out: the tile issued before them has landed, theirs has not
    two = loop + ["\tglobal_load_dwordx4 v[%d:%d], v[40:41], off" % (64 + 4 * j, 64 + 4 * j + 3) for j in range(8)]
    assert mod.check(two + ["\ts_waitcnt vmcnt(8)", "\tv_mov_b32_e32 v60, v28"] + tail) == []
    assert mod.check(two + ["\ts_waitcnt vmcnt(8)", "\tv_mov_b32_e32 v60, v70"] + tail)


def test_inflight_register_check_follows_the_ways_out_of_the_window_loop():
    """Round 6 (lessons 24 and 29): a load issued in the loop's last trip is still on its way when the loop is
    left.  The checker follows every exit edge -- both arms of every branch -- until all loads are waited for: an epilogue that
    names a tile register first is reported (the ragged kernel's counter flush did: a memory fault once in a few hundred
    launches), one that waits first is not, and eight younger loads with vmcnt(8) retire the loop's (loads return in order)."""
    loop = [".LBB0_1:"] + ["\tglobal_load_dwordx4 v[%d:%d], v[40:41], off" % (4 * j, 4 * j + 3) for j in range(8)]
    loop += ["\ts_waitcnt vmcnt(8)", "\tv_add_u32_e32 v50, v50, v51", "\ts_cbranch_vccz .LBB0_3", "\ts_cbranch_scc1 .LBB0_1"]
    mod = _inflight()
    bad = loop + ["\ts_barrier", "\tv_mov_b32_e32 v3, 0", ".LBB0_3:", "\ts_waitcnt vmcnt(0)", "\ts_endpgm"]
    rep = mod.check_exits(bad)
    assert rep and "v_mov_b32_e32 v3, 0" in rep[0][1]
    good = loop + ["\ts_waitcnt vmcnt(0)", "\ts_barrier", "\tv_mov_b32_e32 v3, 0", ".LBB0_3:", "\ts_waitcnt vmcnt(0)", "\tv_mov_b32_e32 v4, 0", "\ts_endpgm"]
    assert mod.check_exits(good) == []
    # the branch out of the loop leads to code that names a tile register before it waits
    assert mod.check_exits(loop + ["\ts_waitcnt vmcnt(0)", "\ts_endpgm", ".LBB0_3:", "\tv_mov_b32_e32 v60, v28", "\ts_endpgm"])
    # the next task's request behind the loop: eight younger loads, vmcnt(8) -- the loop's own have landed
    nxt = ["\tglobal_load_dwordx4 v[%d:%d], v[40:41], off" % (64 + 4 * j, 64 + 4 * j + 3) for j in range(8)]
    assert mod.check_exits(loop + nxt + ["\ts_waitcnt vmcnt(8)", "\tv_mov_b32_e32 v60, v28", "\ts_endpgm", ".LBB0_3:", "\ts_endpgm"]) == []
    assert mod.check_exits(loop + nxt[:4] + ["\ts_waitcnt vmcnt(8)", "\tv_mov_b32_e32 v60, v28", "\ts_endpgm", ".LBB0_3:", "\ts_endpgm"])


def test_the_walker_takes_the_loop_with_the_asm_loads_not_the_table_copy():
    """Round 6: "the innermost loop with eight dwordx4 loads" was the block-wide table copy (the compiler's own loads) in the
    kernels where that is shorter than the window loop -- the walker looked at the wrong loop and a fault of the dense prefix
    instantiation went unseen.  A listing with both: the copy loop first, then a window loop whose loads stand between the asm
    markers and whose way out names a tile register before any wait."""
    copy = [".LBB0_1:"] + ["\tglobal_load_dwordx4 v[%d:%d], v[90:91], off" % (100 + 4 * j, 103 + 4 * j) for j in range(8)]
    copy += ["\ts_waitcnt vmcnt(0)", "\tds_write_b128 v99, v[100:103]", "\ts_cbranch_scc1 .LBB0_1"]
    window = [".LBB0_2:", "\t;;#ASMSTART"] + ["\tglobal_load_dwordx4 v[%d:%d], v[40:41], off" % (4 * j, 4 * j + 3) for j in range(8)] + ["\t;;#ASMEND"]
    window += ["\tv_add_u32_e32 v50, v50, v51", "\tv_add_u32_e32 v52, v50, v51", "\tv_add_u32_e32 v53, v50, v51", "\tv_add_u32_e32 v54, v50, v51",
               "\ts_cbranch_scc1 .LBB0_2"]
    tail = ["\tv_mov_b32_e32 v3, 0", "\ts_waitcnt vmcnt(0)", "\ts_endpgm"]
    mod = _inflight()
    rep = mod.check_exits(copy + window + tail)
    assert rep and "v_mov_b32_e32 v3, 0" in rep[0][1]
    assert mod.check_exits(copy + window + ["\ts_waitcnt vmcnt(0)"] + tail) == []
    # (and the in-loop check: a tile register named between its load and a wait, in the WINDOW loop)
    bad = window[:-1] + ["\tv_mov_b32_e32 v60, v28", "\ts_waitcnt vmcnt(0)", "\ts_cbranch_scc1 .LBB0_2"]
    assert mod.check(copy + bad + tail)


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_the_exit_check_sees_round_5s_wide_kernel_and_is_quiet_on_the_product():
    """check_exits on real ISA: the form of ScanWideKernel that round 5 shipped (kept behind PIRE_EXP == 2 for this test and for
    test_early_out_between_chained_tasks) waited for a tile behind the loop, where hipcc had copied the slot -- reported; the
    product's kernels are clean (the build demands it: build_audit.py).  And the walker must look at the WINDOW loop: the
    block-wide table copy in front of it is a loop of eight dwordx4 loads too (the compiler's own), and until round 6 that was
    the loop it took in the kernels where it is the shorter one -- a fault of the dense prefix instantiation went unseen."""
    import subprocess

    mod = _inflight()
    asm = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-DPIRE_EXP=2", "-x", "hip", "--offload-device-only", "-S",
                          os.path.join(ROOT, "pire_amd", "csrc", "wide.hip"), "-o", "-"], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL,
                         text=True, check=True).stdout
    seen = 0
    for name, body in mod.kernels(asm):
        if "ScanWideKernel" in name:
            seen += 1
            assert mod.check_exits(body, outermost=False), name   # (the ways out of the INNER tile loop: build_audit.py INNER_EXITS)
    assert seen >= 3


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_sanitizer_builds_carry_the_products_device_code():
    """The sanitizer libraries instrument the HOST side (-O1 -g); their kernels must be the product's, instruction for
    instruction: the audits above hold for the product flags only (at -O1 hipcc spills the registers CaptureRowKernel
    lands its text in -- found by the UBSan run of round 4).  counting.hip compiled with the flags of `make ubsan` and
    with the product's, device side only."""
    src = os.path.join(ROOT, "pire_amd", "csrc", "counting.hip")
    mk = open(os.path.join(ROOT, "pire_amd", "csrc", "Makefile")).read()
    assert mk.count("-O1 -g $(DEVOPT)") == 3 and "DEVOPT  := -Xarch_device -O3 -Xarch_device -g0" in mk

    def isa(flags):
        out = subprocess.run([HIPCC, "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-x", "hip", "--offload-device-only", "-S",
                              src, "-o", "-"] + flags, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True,
                             timeout=900).stdout
        return [l for l in out.splitlines() if not l.lstrip().startswith(";") and "__hip_cuid" not in l and ".ident" not in l]

    product = isa(["-O3"])
    sanitized = isa(["-O1", "-g", "-Xarch_device", "-O3", "-Xarch_device", "-g0", "-fsanitize=undefined", "-fno-gpu-sanitize"])
    assert len(product) > 10000 and product == sanitized
