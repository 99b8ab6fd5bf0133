"""LongestSuffix / ShortestSuffix (run.h:313-362): the reference's PrefixSuffix test (pire_ut.cpp:278-306) and its
ScanBoundaries table on reversed texts (pire_ut.cpp:343-473) as known answers, oracle vs reference, GPU vs oracle."""
import numpy as np
import pytest

from oracle import binding as ob
from tests import helpers as H
from tests.test_prefix import SCAN_BOUNDARIES

needs_ref = pytest.mark.skipif(not ob.ref_available(), reason="oracle/_ref/libpire_ref.so not built")
TEXT = b"1234567890 --> middle --> end"


@needs_ref
def test_prefix_suffix_known_answers():
    """pire_ut.cpp:278-306: rsc = Parse("-->", "n").Reverse(); the suffix that ends where a prefix search stopped."""
    r = ob.RefScanner.compile(["-->"], ["nr"])
    o = ob.OracleScanner(r.save())
    for end, begin in ((14, 11), (25, 22)):
        t, offs = H.pack([TEXT[:end]])
        for longest in (True, False):
            want = end - begin                           # LongestSuffix(rsc, end - 1, text - 1) + 1 == text + begin
            assert r.suffix(t, offs, longest)[0] == want
            assert o.suffix(t, offs, longest)[0] == want


@needs_ref
def test_scan_boundaries_reversed_known_answers():
    """pire_ut.cpp:466-471: the same scanners on the reversed text, from its end, give the prefix lengths."""
    for pat, text, shortest, longest in SCAN_BOUNDARIES:
        r = ob.RefScanner.compile([pat], ["n"])
        o = ob.OracleScanner(r.save())
        t, offs = H.pack([text.encode()[::-1]])
        assert r.suffix(t, offs, False)[0] == shortest and r.suffix(t, offs, True)[0] == longest, (pat, text)
        assert o.suffix(t, offs, False)[0] == shortest and o.suffix(t, offs, True)[0] == longest, (pat, text)


@needs_ref
def test_suffix_oracle_vs_reference_random():
    rng = np.random.RandomState(4)
    for pats, opts in ((["-->"], ["nr"]), (["a+b"], ["n"]), (["hello\\s+w.+d$"], [""]), (["[a-c]+d.{2,4}"], ["nr"]), (["aaa"], ["n"])):
        r = ob.RefScanner.compile(pats, opts)
        o = ob.OracleScanner(r.save())
        strings = H.random_strings(rng, 400, 60, b"abcd->hel wor") + [b"", b"a", b"-->"]
        t, offs = H.pack(strings)
        for longest in (True, False):
            for te, tb in ((False, False), (True, False), (False, True), (True, True)):
                assert (r.suffix(t, offs, longest, te, tb) == o.suffix(t, offs, longest, te, tb)).all(), (pats, longest, te, tb)


def _cases():
    return [[x for x in H.all_cases() + H.big_sets() if x["name"] == name][0]
            for name in ("survey_known_answer", "inline_glue3", "rep_dot_3_10", "set_d")]


@pytest.mark.gpu
@pytest.mark.parametrize("case", _cases(), ids=lambda c: c["name"])
def test_gpu_suffix_matches_oracle(case):
    import pire_amd

    blob = H.load_blob(case["blob"])
    t, o = pire_amd.Table(blob), ob.OracleScanner(blob)
    ref = ob.RefScanner.load(blob) if ob.ref_available() else None
    rng = np.random.RandomState(22)
    strings = (H.random_strings(rng, 1500, 120, b"abcdefhelo wrdxHTailnI0123 \t/.:fb") + [b""] * 3 +
               H.random_strings(rng, 300, 80) + [b"dlrow  olleh", b"x" * 300 + b"cba", b"baaa", b"a"])
    text, offs = H.pack(strings)
    for longest in (True, False):
        for te, tb in ((False, False), (True, False), (False, True), (True, True)):
            want = o.suffix(text, offs, longest, te, tb)
            got = t.suffix(text, offs, longest, te, tb)
            assert (got == want).all(), (longest, te, tb, np.nonzero(got != want)[0][:5])
            if ref is not None:   # ... and the unmodified reference itself
                assert (got == ref.suffix(text, offs, longest, te, tb)).all(), (longest, te, tb)


@pytest.mark.gpu
@needs_ref
def test_gpu_suffix_known_answers():
    import pire_amd

    r = ob.RefScanner.compile(["-->"], ["nr"])
    t = pire_amd.Table(r.save())
    for end, begin in ((14, 11), (25, 22)):
        tx, offs = H.pack([TEXT[:end]])
        assert t.suffix(tx, offs, True)[0] == end - begin and t.suffix(tx, offs, False)[0] == end - begin
    for pat, text, shortest, longest in SCAN_BOUNDARIES:
        t = pire_amd.Table(ob.RefScanner.compile([pat], ["n"]).save())
        tx, offs = H.pack([text.encode()[::-1]])
        assert t.suffix(tx, offs, False)[0] == shortest and t.suffix(tx, offs, True)[0] == longest, (pat, text)
    # strings that start in the middle of a buffer, at every alignment: only their own bytes may be looked at
    t = pire_amd.Table(ob.RefScanner.compile(["a+b"], ["n"]).save())
    o = ob.OracleScanner(ob.RefScanner.compile(["a+b"], ["n"]).save())
    strings = [b"b" * k + b"baaa" + b"b" * (k % 3) for k in range(40)]
    tx, offs = H.pack(strings)
    assert (t.suffix(tx, offs, True) == o.suffix(tx, offs, True)).all()


@pytest.mark.gpu
def test_gpu_suffix_whole_blocks_through_the_dense_rows():
    """Round 6: a whole 16-byte block of the string from a state with a dense row is walked top down through the dense rows
    alone (exact.hip SuffixKernel).  Strings of 0..700 bytes cut out of the benchmark corpus at every alignment -- blocks that
    reach a Final or a Dead state, blocks that leave the dense rows, first and last blocks that are not whole -- against the
    oracle and, where it is built, the unmodified reference; set_a's patterns are $-anchored, a table that dies early, one that
    never does."""
    import pire_amd

    big = [b for b in H.big_sets() if b["name"] == "set_a"][0]
    data = ob.corpus_fill(9, 0, 1024, 1024, H.plants_for(big), threads=4).reshape(-1)
    rng = np.random.RandomState(6)
    lens = rng.randint(0, 700, size=1400).astype(np.uint64)
    offs = np.zeros(len(lens) + 1, dtype=np.uint64)
    offs[1:] = np.cumsum(lens)
    text = data[:int(offs[-1])]
    for name in ("set_a", "set_d", "survey_known_answer"):
        case = [x for x in H.all_cases() + H.big_sets() if x["name"] == name][0]
        blob = H.load_blob(case["blob"])
        t, o = pire_amd.Table(blob), ob.OracleScanner(blob)
        ref = ob.RefScanner.load(blob) if ob.ref_available() else None
        for longest in (True, False):
            for te, tb in ((False, False), (True, True)):
                got = t.suffix(text, offs, longest, te, tb)
                assert (got == o.suffix(text, offs, longest, te, tb)).all(), (name, longest, te, tb)
                if ref is not None:
                    assert (got == ref.suffix(text, offs, longest, te, tb)).all(), (name, longest, te, tb)
