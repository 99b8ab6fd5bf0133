"""The C ABI without a GPU: the library loads, exports everything include/pire_hip.h declares, ingests
Scanner::Save() blobs exactly, and refuses to run without a device (no CPU fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import pire_amd
from pire_amd import binding as pb
from oracle import binding as ob
from tests import helpers as H
from tests.conftest import has_gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "pire_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(pire_hip_[a-z_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    names = declared_symbols()
    assert len(names) >= 15
    L = C.CDLL(pire_amd.lib_path())
    for n in names:
        assert hasattr(L, n), f"{n} declared in include/pire_hip.h but not exported"
    bound = {n for n, _, _ in pb.ABI}
    assert bound == set(names), f"python binding out of sync with the header: {bound ^ set(names)}"


def test_no_cpu_scan_symbols_in_product():
    """The product library must not link or embed the oracle."""
    import subprocess

    out = subprocess.run(["nm", "-D", "--defined-only", pire_amd.lib_path()], stdout=subprocess.PIPE, text=True).stdout
    assert "oracle_" not in out
    out = subprocess.run(["ldd", pire_amd.lib_path()], stdout=subprocess.PIPE, text=True).stdout
    assert "liboracle" not in out and "libpire_ref" not in out


@pytest.mark.parametrize("case", H.all_cases() + H.big_sets(), ids=lambda c: c["name"])
def test_table_ingest_matches_oracle(case):
    blob = H.load_blob(case["blob"])
    t = pb.Table(blob)
    o = ob.OracleScanner(blob)
    g = case["geometry"]
    assert t.Empty == g["empty"] == o.empty
    assert t.RegexpsCount == g["regexps"]
    if t.Empty:
        assert t.Size == 1 and not t.Final(0) and t.AcceptedRegexps(0) == []
        return
    assert (t.Size, t.LettersCount, t.initial) == (g["states"], g["letters"], g["initial"])
    assert t.info.ref_buf_size == g["bufsize"]
    assert t.info.row_stride == o.row_stride and t.info.header_size == o.header_size
    for ch in range(260):
        if ch != 257:
            assert t.letter_class(ch) == o.letter_class(ch)
    rng = np.random.RandomState(1)
    states = range(t.Size) if t.Size <= 300 else rng.randint(0, t.Size, 300).tolist()
    chars = list(range(0, 256, 7)) + [9, 10, 32, 65, 97, 255, 256, 258, 259]
    for s in states:
        assert t.Final(s) == o.final(s)
        assert t.Dead(s) == o.dead(s)
        assert t.AcceptedRegexps(s) == o.accepted(s)
        for ch in chars:
            assert t.Next(s, ch) == o.next(s, ch)


@pytest.mark.parametrize("case", H.big_sets() + [c for c in H.all_cases() if c["name"] in
                                                 ("survey_known_answer", "inline_glue3", "utf8_dot", "rep_dot_3_10")],
                         ids=lambda c: c["name"])
def test_device_layout_is_a_faithful_renumbering(case):
    """Simulate the kernel's hot-row/trap logic in numpy on the layout the library would upload and check it
    reproduces the oracle: the permutation is a bijection, every dense-row entry equals the exact transition or
    is the trap id, and the start state is hot."""
    blob = H.load_blob(case["blob"])
    t = pb.Table(blob)
    o = ob.OracleScanner(blob)
    orig_of_perm, hot = t.layout()
    N, Hn = t.Size, t.info.hot_states
    assert sorted(orig_of_perm.tolist()) == list(range(N))
    assert Hn == min(N, 255) and hot.shape == (Hn + 1, 256)
    perm_of_orig = np.empty(N, dtype=np.int64)
    perm_of_orig[orig_of_perm] = np.arange(N)
    assert (hot[Hn] == Hn).all(), "trap row must be absorbing"
    for pid in range(Hn):
        s = int(orig_of_perm[pid])
        exact = np.array([perm_of_orig[o.next(s, b)] for b in range(256)])
        row = hot[pid].astype(np.int64)
        assert ((row == exact) | ((row == Hn) & (exact >= Hn))).all()
    start = o.next(o.initial, 258)
    assert perm_of_orig[start] < Hn and perm_of_orig[o.initial] < Hn
    if "corpus" in case:
        # on the synthetic corpus nearly every step must be served by the dense rows (that is the design point)
        c = case["corpus"]
        data = ob.corpus_fill(c["seed"], 0, c["n"], c["len"], H.plants_for(case))
        st = np.full(c["n"], perm_of_orig[start])
        traps = 0
        nxt_cache = {}
        for pos in range(c["len"]):
            b = data[:, pos]
            e = np.where(st < Hn, hot[np.minimum(st, Hn), b], Hn).astype(np.int64)
            need = e == Hn
            traps += int(need.sum())
            for i in np.nonzero(need)[0]:
                key = (int(st[i]), int(b[i]))
                if key not in nxt_cache:
                    nxt_cache[key] = int(perm_of_orig[o.next(int(orig_of_perm[key[0]]), key[1])])
                e[i] = nxt_cache[key]
            st = e.astype(np.int64)
        end = [o.next(int(orig_of_perm[s]), 259) for s in st]
        assert end == c["idx"]
        # set_a ($-anchored bench patterns): ~every step is dense-row resident.  set_d has unanchored patterns:
        # after a planted match the walk moves into product states the byte model ranks low.
        assert traps / (c["n"] * c["len"]) < (0.01 if case["name"] == "set_a" else 0.10)


def test_bad_blobs_are_rejected_like_header_validate():
    blob = bytearray(H.load_blob(H.all_cases()[0]["blob"]))
    for mutate in (lambda b: b.__setitem__(0, b[0] ^ 0xFF),          # Magic
                   lambda b: b.__setitem__(4, 99),                   # Version
                   lambda b: b.__setitem__(8, 4),                    # PtrSize
                   lambda b: b.__setitem__(16, 2),                   # Type (SimpleScanner)
                   lambda b: b.__setitem__(20, 40)):                 # HdrSize
        bad = bytearray(blob)
        mutate(bad)
        with pytest.raises(pb.PireHipError) as e:
            pb.Table(bytes(bad))
        assert e.value.code == -2
    for cut in (0, 10, 30, 80, 500):
        with pytest.raises(pb.PireHipError):
            pb.Table(bytes(blob[:cut]))
    # a transition pointing outside the table
    bad = bytearray(blob)
    t = pb.Table(bytes(blob))
    pos = len(blob) - t.info.row_stride + t.info.header_size * 4   # first letter class of the last row
    bad[pos:pos + 4] = (0x7FFFFFF0).to_bytes(4, "little")
    with pytest.raises(pb.PireHipError):
        pb.Table(bytes(bad))


@pytest.mark.skipif(has_gpu(), reason="only meaningful on a box without a GPU")
def test_run_without_gpu_fails_loudly():
    t = pb.Table(H.load_blob(H.all_cases()[0]["blob"]))
    with pytest.raises(pb.PireHipError) as e:
        t.run_strings([b"abc"])
    assert e.value.code == -3
    assert "hip" in str(e.value).lower()


def test_config_struct_round_trip_and_partial_sizes():
    """pire_hip_config (runtime knobs through the ABI): get/set round trip, a caller compiled against
    a SHORTER struct only touches the fields it knows, bad sizes are rejected."""
    import ctypes as C

    old = pb.get_config()
    assert old.size == C.sizeof(pb.Config)
    try:
        pb.set_config(tiled_variant=22, segment_bytes=4096, segment_warmup=pb.NONE, auto_adapt=1)
        c = pb.get_config()
        assert (c.tiled_variant, c.segment_bytes, c.segment_warmup, c.auto_adapt) == (22, 4096, pb.NONE, 1)
        short = pb.Config()
        short.size = 12                      # size + tiled_variant + checked
        short.tiled_variant, short.checked = 2, 1
        short.segment_bytes = 999            # beyond `size`: must be ignored
        assert pb.lib().pire_hip_config_set(C.byref(short)) == 0
        c = pb.get_config()
        assert (c.tiled_variant, c.checked, c.segment_bytes) == (2, 1, 4096)
        part = pb.Config()
        part.size = 8
        assert pb.lib().pire_hip_config_get(C.byref(part)) == 0 and part.size == 8 and part.tiled_variant == 2
        bad = pb.Config()
        bad.size = 0
        assert pb.lib().pire_hip_config_set(C.byref(bad)) < 0 and pb.lib().pire_hip_config_get(C.byref(bad)) < 0
        assert pb.lib().pire_hip_config_set(None) < 0
    finally:
        assert pb.lib().pire_hip_config_set(C.byref(old)) == 0
    assert pb.get_config().tiled_variant == old.tiled_variant


def test_environment_only_seeds_the_config_once():
    """The PIRE_HIP_* environment variables are read when the library is loaded, never per launch: a child process
    started with them set sees them in pire_hip_config; changing os.environ afterwards changes nothing."""
    import subprocess
    import sys

    code = ("import os; from pire_amd import binding as pb; c = pb.get_config(); "
            "os.environ['PIRE_HIP_TILED_VARIANT'] = '2'; d = pb.get_config(); "
            "print(c.tiled_variant, c.no_segments, c.segment_warmup == pb.NONE, c.segment_bytes, d.tiled_variant)")
    env = dict(os.environ, PIRE_HIP_TILED_VARIANT="22", PIRE_HIP_NO_SEGMENTS="1", PIRE_HIP_SEGMENT_WARMUP="0",
               PIRE_HIP_SEGMENT_BYTES="512", PYTHONPATH=ROOT)
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-400:]
    assert out.stdout.split() == ["22", "1", "True", "512", "22"]


def test_no_launch_path_reads_the_environment():
    """getenv() only where the configuration is seeded (api.cpp) and in -DPIRE_HIP_TUNING blocks."""
    import re

    src = os.path.join(ROOT, "pire_amd", "csrc")
    for name in sorted(os.listdir(src)):
        if not name.endswith((".cpp", ".hip", ".h")):
            continue
        text = open(os.path.join(src, name)).read()
        # drop the tuning-build blocks
        text = re.sub(r"#ifdef PIRE_HIP_TUNING.*?#endif", "", text, flags=re.S)
        for m in re.finditer(r"getenv\(([^)]*)\)", text):
            assert name == "api.cpp" and ("name" in m.group(1) or "PIRE_HIP_SEGMENT_" in m.group(1)), (name, m.group(0))


def test_info_for_a_caller_built_against_an_older_header():
    """pire_hip_table_get_info_sized writes min(size, sizeof) bytes, pire_hip_abi_version() is the header's."""
    import ctypes as C
    import re

    import pire_amd
    from pire_amd import binding as pb

    L = pb.lib()
    with open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "pire_hip.h")) as f:
        assert int(re.search(r"#define PIRE_HIP_ABI_VERSION (\d+)", f.read()).group(1)) == L.pire_hip_abi_version() == pb.ABI_VERSION
    t = pire_amd.Table(H.load_blob("c2_single.blob"))
    full = t.refresh_info()
    old = (C.c_uint8 * 200)(*([0xAB] * 200))
    assert L.pire_hip_table_get_info_sized(t._h, old, 48) == 0
    assert bytes(old[:48]) == bytes(full)[:48] and all(b == 0xAB for b in old[48:])
    big = (C.c_uint8 * 400)()
    assert L.pire_hip_table_get_info_sized(t._h, big, 400) == 0 and bytes(big[:C.sizeof(full)]) == bytes(full)


def test_a_table_keeps_a_configuration_of_its_own():
    """pire_hip_table_config_set: the configuration calls on ONE table run under, next to the process-wide one."""
    t = pire_amd.Table(H.load_blob("c2_single.blob"))
    before = pb.get_config()
    assert t.get_config().walk_variant == before.walk_variant
    t.set_config(walk_variant=2, selftest=1)
    mine = t.get_config()
    assert (mine.walk_variant, mine.selftest) == (2, 1) and mine.ragged_variant == before.ragged_variant
    assert pb.get_config().walk_variant == before.walk_variant          # the process-wide configuration has not moved
    other = pire_amd.Table(H.load_blob("c2_single.blob"))
    assert other.get_config().walk_variant == before.walk_variant       # nor has another table's
    t.set_config()
    assert t.get_config().walk_variant == before.walk_variant
