"""Static audit of the compiled gfx950 kernels (no GPU needed; hipcc cross-compiles).

The tiled kernel issues its HBM loads from inline asm and waits for them with hand-counted s_waitcnt, so the
compiler does not know those registers are in flight.  If it ever spills (or otherwise copies) a tile register
between the load and the wait, the scan reads garbage silently.  This was observed with a three-slot ring
(DESIGN.md section 6).  The ragged kernel prefetches the same way.  Pin: every instantiated scan kernel has NO
scratch and NO spills."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc"


def resources(unit):
    src = os.path.join(ROOT, "pire_amd", "csrc", unit)
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-x", "hip", "-c", src, "-o", "/dev/null",
                        "-Rpass-analysis=kernel-resource-usage"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                       text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:]
    kernels = {}
    cur = None
    for line in r.stdout.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = kernels.setdefault(m.group(1), {})
            continue
        m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[bytes/lane\])?: (\d+)", line)
        if m and cur is not None:
            cur[m.group(1).strip()] = int(m.group(2))
    return kernels


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_scan_kernels_have_no_scratch_and_no_spills():
    tiled = {k: v for k, v in resources("tiled.hip").items() if "ScanTiledKernel" in k or "ScanTiledSegKernel" in k}
    assert any("ScanTiledKernel" in k for k in tiled), "no tiled kernel instantiation found"
    assert any("ScanTiledSegKernel" in k for k in tiled), "no segment form of the tiled kernel found"
    ragged = {k: v for k, v in resources("ragged.hip").items() if "ScanRaggedKernel" in k}
    assert ragged, "no ragged kernel found"
    pair = {k: v for k, v in resources("pair.hip").items() if "ScanPairTiledKernel" in k}
    assert pair, "no fused pair kernel found"
    stream = {k: v for k, v in resources("stream.hip").items() if "ScanStreamKernel" in k}
    assert stream, "no stream kernel found"
    for name, res in stream.items():
        # two line registers per wave, one of them in flight during the walk: a spill of either reads or clobbers a
        # register the compiler does not know is busy (round 4: named behind the window loop they went through scratch)
        assert res.get("ScratchSize", -1) == 0 and res.get("VGPRs Spill", -1) == 0, (name, res)
        assert res["VGPRs"] <= 128, (name, res)   # 16 waves per CU
    for name, res in pair.items():
        assert res["VGPRs"] <= 128, (name, res)   # 16 waves per CU, like the tiled kernel it shares the load path with
    for name, res in list(tiled.items()) + list(ragged.items()) + list(pair.items()):
        assert res.get("ScratchSize", -1) == 0, (name, res)
        assert res.get("VGPRs Spill", -1) == 0, (name, res)   # SGPR spills go to VGPR lanes, harmless
    for name, res in tiled.items():
        assert res["VGPRs"] <= 128, (name, res)   # 16 waves per CU = 4 per SIMD x 128 registers (the shadowed transpose
                                                  # of round 3 keeps four temporaries alive across lookups: 100)
    for name, res in ragged.items():
        assert res["VGPRs"] <= 128, (name, res)   # 16 waves per CU (one 1024-thread block)


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_exact_kernels_have_no_scratch():
    """Per-lane counter arrays must stay in registers (a runtime index once put HalfFinalKernel's into scratch)."""
    for unit in ("exact.hip", "counting.hip", "slow.hip", "segmented.hip", "order.hip"):
        for name, res in resources(unit).items():
            if "pirehip" in name and "CountingRowKernel" not in name and "CaptureRowKernel" not in name:    # segmented.hip also instantiates library (rocprim) scan kernels
                assert res.get("ScratchSize", -1) == 0, (unit, name, res)   # (the row kernels: the test below)


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_counting_row_kernel_owns_its_landing_registers():
    """CountingRowKernel and CaptureRowKernel land the text line that is on its way in a0..a31, named in its asm statements only.  The
    compiler may use accumulation registers as spill space: never those (a register written by it while the memory
    system still owes data to it, or the other way round, is a silently wrong count), and no scratch."""
    src = os.path.join(ROOT, "pire_amd", "csrc", "counting.hip")
    res = {k: v for k, v in resources("counting.hip").items() if "CountingRowKernel" in k or "CaptureRowKernel" in k}
    assert len(res) == 16, sorted(res)
    for name, r in res.items():
        assert r["VGPRs"] + r.get("AGPRs", 0) <= 128, (name, r)   # 16 waves per CU
        assert r.get("ScratchSize", -1) <= (96 if "Capture" in name else 64), (name, r)   # a few per-pass values, no array
    asm = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-x", "hip", "--offload-device-only", "-S", src,
                          "-o", "-"], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, timeout=900).stdout
    body, seen = None, 0
    lines = asm.splitlines()
    for n, line in enumerate(lines):
        m = re.match(r"^(_ZN7pirehip\w*(?:CountingRowKernel|CaptureRowKernel)\S*):", line)
        if m:
            body, seen = m.group(1), seen + 1
            # what hipcc spills per pass (a lane has 64 ordinary registers next to the 64 accumulation registers) goes
            # to a32.. and a few bytes of scratch: not inside the window loop, where its s_waitcnt for a reload would
            # wait for the line on its way as well
            end = next(k for k in range(n, len(lines)) if lines[k].startswith(".Lfunc_end"))
            land = next(k for k in range(n, end) if re.search(r"v_accvgpr_read_b32 v\d+, a0\b", lines[k]))
            last = max(k for k in range(n, end) if "ds_read_b64" in lines[k] or "ds_read_b128" in lines[k])
            # (CaptureRowKernel reloads one value per window there -- 52 bytes of scratch per lane, measured with it)
            assert "Capture" in body or not [lines[k] for k in range(land, last) if "scratch_" in lines[k]], body
        elif line.startswith(".Lfunc_end"):
            body = None
        elif body:
            # a0..a31: written by the eight loads, read by v_accvgpr_read_b32, touched by nothing else
            for m in re.finditer(r"\ba(\d+)\b|\ba\[(\d+):(\d+)\]", line.split(";")[0]):
                lo = int(m.group(1) if m.group(1) is not None else m.group(2))
                if lo >= 32:
                    continue
                op = line.split()[0]
                assert op in ("global_load_dwordx4", "v_accvgpr_read_b32"), (body, line)
                if op == "global_load_dwordx4":
                    assert re.search(r"global_load_dwordx4 a\[\d+:\d+\], v\[\d+:\d+\], off", line), (body, line)
    assert seen == 16


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


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
@pytest.mark.parametrize("unit,kernel", [("stream.hip", "ScanStreamKernel"), ("tiled.hip", "ScanTiledKernel"),
                                         ("ragged.hip", "ScanRaggedKernel")])
def test_no_instruction_touches_a_tile_that_is_on_its_way(unit, kernel):
    """The window loops of the kernels that keep the line on its way in ordinary registers (tools/audit/
    inflight_registers.py): nothing but the loads names a register between its load and the s_waitcnt that covers it."""
    mod = _inflight()
    asm = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-x", "hip", "--offload-device-only", "-S",
                          os.path.join(ROOT, "pire_amd", "csrc", unit), "-o", "-"], stdout=subprocess.PIPE,
                         stderr=subprocess.DEVNULL, text=True, timeout=900).stdout
    found = 0
    for name, body in mod.kernels(asm):
        if kernel in name:
            found += 1
            assert mod.check(body) == [], name
    assert found


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
