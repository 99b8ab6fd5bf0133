"""The oracle (oracle/pire_oracle.c) against the reference: golden known answers + the live reference library."""
import os
import subprocess

import numpy as np
import pytest

from oracle import binding as ob
from tests import helpers as H

REF_PRESENT = os.path.exists("/root/reference/pire/run.h")


@pytest.mark.parametrize("case", H.all_cases(), ids=lambda c: c["name"])
def test_oracle_matches_golden(case):
    """Every golden case: geometry, StateIndex, Final and AcceptedRegexps are bit-exact, and the verdicts the
    reference's own unit tests assert (ACCEPTS/DENIES, pire_ut.cpp) hold."""
    o = ob.OracleScanner(H.load_blob(case["blob"]))
    g = case["geometry"]
    assert o.empty == g["empty"]
    assert o.regexps == g["regexps"]
    if not g["empty"]:
        assert (o.size, o.letters, o.initial) == (g["states"], g["letters"], g["initial"])
    strings = H.case_strings(case)
    idx, fin = o.run_strings(strings)
    assert idx.tolist() == case["idx"]
    assert fin.tolist() == case["final"]
    acc = [o.accepted(int(i)) for i in idx]
    assert acc == case["accepted"]
    for a, want in zip(acc, case.get("ref_expect", [])):
        if want is not None:
            assert (len(a) > 0) == want
    for a, want in zip(acc, case.get("ref_expect_accepted", [])):
        assert a == want
    # the production control flow (head/body/tail + exit-mask skipping) is result-neutral
    idx2, fin2 = o.run_strings(strings, shortcut=True)
    assert (idx2 == idx).all() and (fin2 == fin).all()


@pytest.mark.parametrize("big", H.big_sets(), ids=lambda b: b["name"])
def test_oracle_big_sets(big):
    o = ob.OracleScanner(H.load_blob(big["blob"]))
    g = big["geometry"]
    assert (o.size, o.letters, o.regexps, o.initial) == (g["states"], g["letters"], g["regexps"], g["initial"])
    c = big["corpus"]
    data = ob.corpus_fill(c["seed"], 0, c["n"], c["len"], H.plants_for(big))
    import hashlib

    assert hashlib.sha256(data.tobytes()).hexdigest() == c["sha256"]
    offs = np.arange(c["n"] + 1, dtype=np.uint64) * c["len"]
    for shortcut in (False, True):
        idx, fin = o.run(data.reshape(-1), offs, shortcut=shortcut)
        assert idx.tolist() == c["idx"] and fin.tolist() == c["final"]
    assert [o.accepted(int(i)) for i in idx] == c["accepted"]
    assert len(set(c["idx"])) >= (8 if o.regexps >= 8 else 2), "planted corpus must exercise several distinct end states"
    raw = [bytes.fromhex(h) for h in big["raw"]["strings_hex"]]
    idx, fin = o.run_strings(raw)
    assert idx.tolist() == big["raw"]["idx"] and fin.tolist() == big["raw"]["final"]
    idx, fin = o.run_strings(raw, shortcut=True)
    assert idx.tolist() == big["raw"]["idx"]


def test_oracle_threads_and_resume():
    big = H.big_sets()[0]
    o = ob.OracleScanner(H.load_blob(big["blob"]))
    data = ob.corpus_fill(7, 0, 50, 333, H.plants_for(big))
    offs = np.arange(51, dtype=np.uint64) * 333
    a = o.run(data.reshape(-1), offs)
    b = o.run(data.reshape(-1), offs, threads=4)
    assert (a[0] == b[0]).all() and (a[1] == b[1]).all()
    # chunked scan with resume == whole scan (the O(1)-state streaming property, run.h:368, 391-392)
    cut = 100
    offs1 = np.stack([np.arange(50) * 333, np.arange(50) * 333 + cut], 1).astype(np.uint64)
    text = data.reshape(-1)
    first = np.concatenate([text[int(b):int(e)] for b, e in offs1])
    second = np.concatenate([text[int(e):int(b) + 333] for b, e in offs1])
    i1, _ = o.run(first, np.arange(51, dtype=np.uint64) * cut, flags=ob.FLAG_BEGIN)
    i2, f2 = o.run(second, np.arange(51, dtype=np.uint64) * (333 - cut), flags=ob.FLAG_END, init_idx=i1)
    assert (i2 == a[0]).all() and (f2 == a[1]).all()


def test_oracle_rejects_bad_blobs():
    blob = bytearray(H.load_blob(H.all_cases()[0]["blob"]))
    with pytest.raises(ValueError):
        ob.OracleScanner(bytes(blob[:10]))
    bad = bytearray(blob)
    bad[0] ^= 0xFF   # magic
    with pytest.raises(ValueError):
        ob.OracleScanner(bytes(bad))
    bad = bytearray(blob)
    bad[4] = 99      # version
    with pytest.raises(ValueError):
        ob.OracleScanner(bytes(bad))
    with pytest.raises(ValueError):
        ob.OracleScanner(bytes(blob[:200]))   # truncated buffer


# ------------------------------------------------------------------ against the live reference library

needs_ref = pytest.mark.skipif(not ob.ref_available(), reason="oracle/_ref/libpire_ref.so not built")


@needs_ref
@pytest.mark.parametrize("patterns,opts", [
    (["hello\\s+w.+d$"], None),
    (["a.{4}b", "[0-9]+x", "foo|bar"], None),
    (["^abc", "abc$", "^x{3,6}$"], ["", "", ""]),
    (["\xD0\xB0.\xD0\xB1"], ["u"]),
    (["[a-c]+d", "d[a-c]+"], ["i", "n"]),
])
def test_oracle_vs_reference_random(patterns, opts):
    r = ob.RefScanner.compile(patterns, opts)
    o = ob.OracleScanner(r.save())
    assert (o.size, o.letters, o.regexps, o.initial) == (r.size, r.letters, r.regexps, r.initial)
    rng = np.random.RandomState(42)
    strings = H.random_strings(rng, 300, 200) + H.random_strings(rng, 300, 120, b"abcdxfo0123456789 hellowrd\xd0\xb0\xb1")
    ri, rf = r.run_strings(strings)
    for shortcut in (False, True):
        oi, of = o.run_strings(strings, shortcut=shortcut)
        assert (oi == ri).all() and (of == rf).all()
    rn, _ = r.run_strings(strings, kind=ob.RefScanner.NONRELOC)
    assert (rn == ri).all()
    for s in range(o.size):
        assert o.final(s) == r.final(s) and o.dead(s) == r.dead(s) and o.accepted(s) == r.accepted(s)
        for ch in (0, 65, 97, 100, 255, 256, 258, 259):
            assert o.next(s, ch) == r.next(s, ch)
    # flags: no Begin / no End / neither
    for flags in (0, ob.FLAG_BEGIN, ob.FLAG_END):
        oi, of = o.run_strings(strings[:100], flags=flags)
        ri2, rf2 = r.run_strings(strings[:100], flags=flags)
        assert (oi == ri2).all() and (of == rf2).all()


@needs_ref
def test_oracle_prefix_vs_reference():
    # patterns of the ScanBoundaries table (pire_ut.cpp:343-473) are compiled unsurrounded there
    for pat in ["a*", "a", "fixed", "a+b", "(abc|def)+", "aaa"]:
        r = ob.RefScanner.compile([pat], ["n"])
        o = ob.OracleScanner(r.save())
        rng = np.random.RandomState(3)
        strings = H.random_strings(rng, 200, 12, b"abcdefix") + [b"", b"aaab", b"fixed point", b"abcdefabc"]
        text, offs = H.pack(strings)
        for longest in (True, False):
            for tb, te in ((False, False), (True, True)):
                assert (o.prefix(text, offs, longest, tb, te) == r.prefix(text, offs, longest, tb, te)).all()


@pytest.mark.skipif(not REF_PRESENT, reason="/root/reference not present (GPU box)")
def test_reference_own_unit_tests_pass_with_standin_parser():
    """Acceptance gate of oracle/_ref: the reference's OWN tests (pire_ut.cpp + easy_ut.cpp) built against the
    same objects and our bison stand-in must pass."""
    here = os.path.join(H.GOLDEN, "..", "..", "oracle")
    subprocess.run(["make", "-C", here, "-j8", "reftest"], check=True, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    r = subprocess.run([os.path.join(here, "_ref", "pire_test")], stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:]
    assert "OK(50 tests)" in r.stdout          # pire_ut + easy_ut (26) + count_ut (13) + capture_ut (11)


@needs_ref
def test_baseline_config_c1_nonreloc_scanner_10k_x_256_on_the_reference_cpu_run():
    """BASELINE.json configs[0], at its stated shape: a single NonrelocScanner, pattern hello\\s+w.+d$, 10 000 x 256 B
    ASCII strings, the reference's own CPU Run() -- the plumbing case, no GPU.  The unmodified reference
    (oracle/_ref: Pire::NonrelocScanner, Runner(sc).Begin().Run(p, 256).End() per string) over the seeded corpus; the
    relocatable Scanner and the C restatement must return the same StateIndex / Final for every string, the planted
    witnesses must be found, and SURVEY 8(c)'s known answers hold for this very scanner object."""
    big = [b for b in H.big_sets() if b["name"] == "c2_single"][0]
    assert big["patterns"] == ["hello\\s+w.+d$"]
    blob = H.load_blob(big["blob"])
    r, o = ob.RefScanner.load(blob), ob.OracleScanner(blob)
    assert (r.size, r.letters, r.initial) == (11, 10, 8)          # 11 states x 10 letter classes, initial 8
    n, length = 10000, 256
    data = ob.corpus_fill(0x5EED5EED, 0, n, length, H.plants_for(big), threads=4)
    assert data.max() <= 0x7E                                     # ASCII (printable text, tabs in the whitespace runs)
    offs = np.arange(n + 1, dtype=np.uint64) * length
    ni, nf = r.run(data.reshape(-1), offs, kind=ob.RefScanner.NONRELOC, threads=1)   # the config: NonrelocScanner, CPU Run()
    si, sf = r.run(data.reshape(-1), offs, kind=0, threads=2)
    oi, of = o.run(data.reshape(-1), offs, threads=2)
    assert (ni == si).all() and (nf == sf).all() and (ni == oi).all() and (nf == of).all()
    assert 0 < int(nf.sum()) < n and len(np.unique(ni)) >= 2      # planted matches: parity is not vacuous
    known = [(b"hello world", 1, 1), (b"Hello world", 8, 0), (b"say hello   wod", 1, 1), (b"hello world!", 8, 0),
             (b"hello wd", 8, 0), (b"", 8, 0), (b"xxhello\tw--d", 1, 1)]
    ki, kf = r.run_strings([k[0] for k in known], kind=ob.RefScanner.NONRELOC)
    assert ki.tolist() == [k[1] for k in known] and kf.tolist() == [k[2] for k in known]
