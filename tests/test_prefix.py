"""LongestPrefix / ShortestPrefix (run.h:277-311): the ScanBoundaries table of the reference's unit test
(pire_ut.cpp:343-473) as known answers, oracle vs reference, GPU vs oracle."""
import numpy as np
import pytest

from oracle import binding as ob
from tests import helpers as H

# (pattern, text, shortestPrefixLen, longestPrefixLen) -- a sample of the cases of pire_ut.cpp:351-457, compiled
# exactly as there: Lexer(pattern).Parse().Compile<Scanner>() (no Surround), no Begin/End marks.
SCAN_BOUNDARIES = [
    ("a*", "", 0, 0),
    ("a", "", -1, -1),
    ("fixed", "fixed prefix", 5, 5),
    ("fixed", "a fixed nonprefix", -1, -1),
    ("a*", "aaa", 0, 3),
    ("a+", "aaa", 1, 3),
    ("a+b", "aaab", 4, 4),
    ("aaa", "aaab", 3, 3),
]

needs_ref = pytest.mark.skipif(not ob.ref_available(), reason="oracle/_ref/libpire_ref.so not built")


@needs_ref
def test_scan_boundaries_known_answers():
    for pat, text, shortest, longest in SCAN_BOUNDARIES:
        r = ob.RefScanner.compile([pat], ["n"])
        o = ob.OracleScanner(r.save())
        t, offs = H.pack([text.encode()])
        assert r.prefix(t, offs, False)[0] == shortest, (pat, text)
        assert r.prefix(t, offs, True)[0] == longest, (pat, text)
        assert o.prefix(t, offs, False)[0] == shortest and o.prefix(t, offs, True)[0] == longest


def _cases():
    out = []
    for name in ("survey_known_answer", "inline_glue3", "rep_dot_3_10", "set_d"):
        c = [x for x in H.all_cases() + H.big_sets() if x["name"] == name][0]
        out.append(c)
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("case", _cases(), ids=lambda c: c["name"])
def test_gpu_prefix_matches_oracle(case):
    import pire_amd

    blob = H.load_blob(case["blob"])
    t, o = pire_amd.Table(blob), ob.OracleScanner(blob)
    ref = ob.RefScanner.load(blob) if ob.ref_available() else None
    rng = np.random.RandomState(21)
    strings = (H.random_strings(rng, 1500, 120, b"abcdefhelo wrdxHTailnI0123 \t/.:fb") + [b""] * 3 +
               H.random_strings(rng, 300, 80) + [b"hello  world", b"say hello   wod and more", b"aaab", b"xxabc"])
    text, offs = H.pack(strings)
    for longest in (True, False):
        for tb, te in ((False, False), (True, False), (False, True), (True, True)):
            want = o.prefix(text, offs, longest, tb, te)
            got = t.prefix(text, offs, longest, tb, te)
            assert (got == want).all(), (longest, tb, te, np.nonzero(got != want)[0][:5])
            if ref is not None:   # ... and the unmodified reference itself (the GPU searches were held against the restatement only)
                assert (got == ref.prefix(text, offs, longest, tb, te)).all(), (longest, tb, te)


@pytest.mark.gpu
@needs_ref
def test_gpu_prefix_scan_boundaries_and_termination():
    import pire_amd

    for pat, text, shortest, longest in SCAN_BOUNDARIES:
        r = ob.RefScanner.compile([pat], ["n"])
        t = pire_amd.Table(r.save())
        tx, offs = H.pack([text.encode()])
        assert t.prefix(tx, offs, False)[0] == shortest, (pat, text)
        assert t.prefix(tx, offs, True)[0] == longest, (pat, text)
    # ScanTermination, pire_ut.cpp:475-483: must stop at the first dead state
    r = ob.RefScanner.compile(["aaa"], ["n"])
    t = pire_amd.Table(r.save())
    tx, offs = H.pack([b"aaab\x00"])
    assert t.prefix(tx, offs, True)[0] == 3


@pytest.mark.gpu
def test_prefix_calls_alone_feed_the_adaptation(cfg):
    """A caller of pire_hip_prefix alone: the walks with actions leave visit samples of the states outside the dense rows
    (round 5; before, only the plain scans did), adapt() promotes them, the same searches then leave them no more --
    same answers before and after.  The table starts from a prior that knows nothing."""
    import pire_amd

    big = [b for b in H.big_sets() if b["name"] == "set_a"][0]
    blob = H.load_blob(big["blob"])
    cfg.set(prior_flat=1, ragged_act_always=1)
    t, o = pire_amd.Table(blob), ob.OracleScanner(blob)
    t.layout()
    cfg.set(prior_flat=0)
    rng = np.random.RandomState(3)
    data = ob.corpus_fill(5, 0, 4096, 1024, H.plants_for(big), threads=4).reshape(-1)
    lens = rng.randint(64, 700, size=6000).astype(np.uint64)
    offs = np.zeros(len(lens) + 1, dtype=np.uint64)
    offs[1:] = np.cumsum(lens)
    text = data[:int(offs[-1])]
    want = o.prefix(text, offs, True, False, True)
    assert (t.prefix(text, offs, True, False, True) == want).all()
    changed = t.adapt()
    first = t.info.last_trap_samples
    assert first > 0 and changed > 0, (first, changed)
    assert (t.prefix(text, offs, True, False, True) == want).all()
    t.adapt()
    assert t.info.last_trap_samples * 10 < first, (t.info.last_trap_samples, first)
    assert (t.prefix(text, offs, False, False, True) == o.prefix(text, offs, False, False, True)).all()
